cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 3 gpurun_out/$name.log | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_conv 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x
step bench_a 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline
BDVCIL_WGRAD_SIDE_STREAM=1 step bench_side 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing
step bench_b 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing
BDVCIL_WGRAD_SIDE_STREAM=1 step bench_side2 300 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing
rm -rf gpurun_out/prof_r2a
step prof 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2a -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
find gpurun_out/prof_r2a -name "*kernel_trace.csv" -size +8M -delete
