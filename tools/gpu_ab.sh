cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "[pytest_gpu] rc=$rc"; tail -n 5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
OLD=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_old.so
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_new$i.log 2>&1
echo "[bench_new$i] rc=$?"; tail -n 1 gpurun_out/bench_new$i.log | cut -c1-120
BDVCIL_LIB_PATH=$OLD timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_old$i.log 2>&1
echo "[bench_old$i] rc=$?"; tail -n 1 gpurun_out/bench_old$i.log | cut -c1-120
done
