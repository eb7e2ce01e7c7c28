cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 4 gpurun_out/$name.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_conv 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_sites_gpu.py tests/test_i3d_gpu.py tests/test_bf16x1_gpu.py -m gpu -q -x
step bench_conv 300 python tools/bench_conv.py
step bench 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
BDVCIL_C4_X3=0 BDVCIL_PL_WGRAD64=0 step bench_old 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
