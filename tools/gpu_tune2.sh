cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 2 gpurun_out/$name.log | cut -c1-250
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_conv 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_sites_gpu.py -m gpu -q -x
step bc_default 300 python tools/bench_conv.py
BDVCIL_WGRAD_2STAGE=0 step bc_1stage 300 python tools/bench_conv.py
BDVCIL_WGRAD_TILE=1 step bc_128x256 300 python tools/bench_conv.py
BDVCIL_WGRAD_TILE=2 step bc_256x128 300 python tools/bench_conv.py
