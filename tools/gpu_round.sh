# One gpurun call: the -m gpu suite, then the round profile (bench line, rocprofv3 kernel stats, the two PMC passes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.log
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=10 ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "[pytest_gpu] rc=$rc"; tail -n 5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
R=${R:-r02} bash tools/run_profile.sh
