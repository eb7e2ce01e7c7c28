cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 6 gpurun_out/$name.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_i3d 900 python -m pytest tests/test_i3d_gpu.py tests/test_conv_gpu.py -m gpu -q -x
step bench_i3d 400 python bench.py --workload i3d --steps 8 --warmup 2 --no-cpu-baseline
BDVCIL_STEM3D_FUSED=0 step bench_i3d_5pass 400 python bench.py --workload i3d --steps 8 --warmup 2 --no-cpu-baseline
step bench 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline
