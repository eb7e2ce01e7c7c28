cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 3 gpurun_out/$name.log | cut -c1-250
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_conv 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x -k "every_tile or dgrad"
FUSED=1 step tune_conv_fused 600 python tools/tune_conv.py
