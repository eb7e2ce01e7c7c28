# bf16-storage mode: its tests, the neighbouring suites whose kernels were re-templated, and the config-5 bench lines.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bf16_storage_gpu.py -m gpu -q -x -s > gpurun_out/pytest_bf16.log 2>&1
rc=$?; echo "[pytest bf16 storage] rc=$rc"; tail -n 25 gpurun_out/pytest_bf16.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "kd or ops or eval or frontend" > gpurun_out/pytest_ops.log 2>&1
rc=$?; echo "[pytest ops/kd] rc=$rc"; tail -n 5 gpurun_out/pytest_ops.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/bench_bf16_$i.log 2>&1
echo "[bench bf16 storage b64 #$i] rc=$?"; tail -n 1 gpurun_out/bench_bf16_$i.log | cut -c100-240
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16x1 --batch 64 > gpurun_out/bench_bf16x1_$i.log 2>&1
echo "[bench bf16x1 b64 #$i] rc=$?"; tail -n 1 gpurun_out/bench_bf16x1_$i.log | cut -c100-240
done
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --workload cil --steps 8 --warmup 2 > gpurun_out/bench_cil_bf16.log 2>&1
echo "[bench cil bf16] rc=$?"; tail -n 1 gpurun_out/bench_cil_bf16.log | cut -c1-300
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --workload predict --steps 8 --warmup 2 > gpurun_out/bench_predict_bf16.log 2>&1
echo "[bench predict bf16] rc=$?"; tail -n 1 gpurun_out/bench_predict_bf16.log | cut -c1-300
