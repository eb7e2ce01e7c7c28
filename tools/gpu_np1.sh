# Reduced-precision kernels: the new library against the old one (BDVCIL_LIB_PATH), alternating processes in one call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_old.so
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r = d['roofline']
print(d['value'], 'clips/s', d['ms_per_step'], 'ms  conv', r['conv_ms_per_step'], 'ms')
acc = {}
for k, v in r['all_conv_kernels'].items():
    key = k.split('_pl_')[0].replace('conv_', '') + ' ' + k.split('<')[1][:7] if '_pl_' in k else k[:24]
    acc[key] = acc.get(key, 0) + v['ms_per_step']
print('      ' + '  '.join(f'{k} {v:.2f}' for k, v in acc.items()))
PY
}
timeout -k 10 600 python -m pytest tests/test_bf16_storage_gpu.py tests/test_bf16x1_gpu.py -m gpu -q -x > gpurun_out/pytest_bf16.log 2>&1
rc=$?; echo "[pytest bf16 + bf16x1] rc=$rc"; tail -n 4 gpurun_out/pytest_bf16.log | cut -c1-300
if [ $rc -ne 0 ]; then tail -n 40 gpurun_out/pytest_bf16.log | cut -c1-200; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/np1_new_bf16_$i.log 2>&1
echo -n "[new bf16 b64 #$i] rc=$? "; show gpurun_out/np1_new_bf16_$i.log
BDVCIL_LIB_PATH=$OLD timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/np1_old_bf16_$i.log 2>&1
echo -n "[old bf16 b64 #$i] rc=$? "; show gpurun_out/np1_old_bf16_$i.log
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/np1_new_default.log 2>&1
echo -n "[new default] rc=$? "; show gpurun_out/np1_new_default.log
