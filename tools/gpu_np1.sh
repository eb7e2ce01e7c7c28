# Reduced-precision kernels: the new library against the old one (BDVCIL_LIB_PATH), alternating processes in one call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_old.so
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r = d['roofline']
print(d['value'], 'clips/s', d['ms_per_step'], 'ms  conv', r['conv_ms_per_step'], 'ms')
PY
}
timeout -k 10 900 python -m pytest tests/test_bf16_storage_gpu.py tests/test_bf16x1_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/pytest_bf16.log 2>&1
rc=$?; echo "[pytest bf16 + bf16x1 + ops + model] rc=$rc"; tail -n 4 gpurun_out/pytest_bf16.log | cut -c1-300
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/pytest_bf16.log | cut -c1-200; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/np1_new_bf16_$i.log 2>&1
echo -n "[new bf16 b64 #$i] rc=$? "; show gpurun_out/np1_new_bf16_$i.log
BDVCIL_LIB_PATH=$OLD timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/np1_old_bf16_$i.log 2>&1
echo -n "[old bf16 b64 #$i] rc=$? "; show gpurun_out/np1_old_bf16_$i.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/np1_new_default_$i.log 2>&1
echo -n "[new default #$i] rc=$? "; show gpurun_out/np1_new_default_$i.log
BDVCIL_LIB_PATH=$OLD timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/np1_old_default_$i.log 2>&1
echo -n "[old default #$i] rc=$? "; show gpurun_out/np1_old_default_$i.log
done
