# bf16-storage mode: tests, then a kernel trace of the config-5 bench line (top kernels by total time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bf16_storage_gpu.py -m gpu -q -x > gpurun_out/pytest_bf16.log 2>&1
rc=$?; echo "[pytest bf16 storage] rc=$rc"; tail -n 25 gpurun_out/pytest_bf16.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
rm -rf gpurun_out/prof_bf16
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bf16 -- python3 bench.py --arith bf16 --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_bf16.json 2> gpurun_out/prof_bf16.err
echo "[prof] rc=$?"
find gpurun_out/prof_bf16 -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/prof_bf16 -name "*kernel_stats.csv" | head -n 1)
head -n 25 "$f" | cut -c1-200
