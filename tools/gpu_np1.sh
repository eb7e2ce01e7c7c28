# NP = 1 plane kernels: occupancy rules of the new library against the old one, and forced tiles, alternating processes in one call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_old.so
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
r = d['roofline']
print(d['value'], 'clips/s', d['ms_per_step'], 'ms  conv', r['conv_ms_per_step'], 'ms')
for k, v in r['all_conv_kernels'].items():
    if 'fprop_pl' in k or 'dgrad_pl' in k: print('      ', k, v['launches_per_step'], v['ms_per_step'])
PY
}
timeout -k 10 600 python -m pytest tests/test_bf16_storage_gpu.py tests/test_bf16x1_gpu.py -m gpu -q -x > gpurun_out/pytest_bf16.log 2>&1
rc=$?; echo "[pytest bf16 + bf16x1] rc=$rc"; tail -n 4 gpurun_out/pytest_bf16.log | cut -c1-300
for i in 1 2; do
for arith in bf16 bf16x1; do
timeout -k 10 300 python bench.py --no-cpu-baseline --arith $arith --batch 64 > gpurun_out/np1_new_${arith}_$i.log 2>&1
echo -n "[new $arith b64 #$i] rc=$? "; show gpurun_out/np1_new_${arith}_$i.log
BDVCIL_LIB_PATH=$OLD timeout -k 10 300 python bench.py --no-cpu-baseline --arith $arith --batch 64 > gpurun_out/np1_old_${arith}_$i.log 2>&1
echo -n "[old $arith b64 #$i] rc=$? "; show gpurun_out/np1_old_${arith}_$i.log
done
done
for t in 4 3; do
BDVCIL_PL_TILE=$t timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/np1_tile$t.log 2>&1
echo -n "[new bf16 b64 forced tile $t] rc=$? "; show gpurun_out/np1_tile$t.log
done
