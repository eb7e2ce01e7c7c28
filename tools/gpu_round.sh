# One gpurun call: the -m gpu suite, the per-site conv table, a short bench line.
# A step that was killed at its limit ends the call (no further GPU step behind a hung one).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.log
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 5 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_gpu 1100 python -m pytest tests -m gpu -q --durations=10 ${PYTEST_ARGS:-}
step bench_conv_x3 200 python tools/bench_conv.py
step bench 400 python bench.py --steps 12 --warmup 4
