cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "[pytest_gpu] rc=$rc"; tail -n 5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench.log 2>&1
echo "[bench] rc=$?"; tail -n 1 gpurun_out/bench.log | cut -c1-200
BDVCIL_PREFETCH_PLANES=0 timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_nopf.log 2>&1
echo "[bench_nopf] rc=$?"; tail -n 1 gpurun_out/bench_nopf.log | cut -c1-200
