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
OLD=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_old.so
step pytest_conv 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_sites_gpu.py tests/test_ops_gpu.py -m gpu -q -x
step bench_conv_new 300 python tools/bench_conv.py
BDVCIL_LIB_PATH=$OLD step bench_conv_old 300 python tools/bench_conv.py
step bench_new 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
BDVCIL_LIB_PATH=$OLD step bench_old 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
step bench_new2 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
BDVCIL_LIB_PATH=$OLD step bench_old2 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
step find_fills 200 python tools/debug/find_fills.py
