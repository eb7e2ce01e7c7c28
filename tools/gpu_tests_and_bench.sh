# One gpurun call: the -m gpu suite, the per-site conv table in both arithmetics, a short bench line.
# A step that was killed at its limit ends the call (no further GPU step behind a hung one).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 4 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_gpu 1000 python -m pytest tests -m gpu -q -x --durations=15
step bench_conv_x3 200 python tools/bench_conv.py
BDVCIL_CONV_F32MFMA=1 step bench_conv_f32 200 python tools/bench_conv.py
step bench 400 python bench.py --steps 12 --warmup 4
