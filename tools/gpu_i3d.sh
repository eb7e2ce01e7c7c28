cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 6 gpurun_out/$name.log | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_new 900 python -m pytest tests/test_i3d_gpu.py tests/test_bf16x1_gpu.py tests/test_conv_gpu.py -m gpu -q
step bench_x1 400 python bench.py --arith bf16x1 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline
step bench_i3d 400 python bench.py --workload i3d --steps 6 --warmup 2
