# Final validation of a build in one call: build check, smoke, the -m gpu suite, the default bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "[smoke] rc=$?"; tail -n 1 gpurun_out/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "[pytest_gpu] rc=$rc"; tail -n 3 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
timeout -k 10 400 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "[bench] rc=$?"; tail -n 1 gpurun_out/bench.log | cut -c1-400
