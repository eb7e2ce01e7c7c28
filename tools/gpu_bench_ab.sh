cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 2 gpurun_out/$name.log | cut -c1-330
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
B="python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timing"
step bench_side 300 $B
BDVCIL_MAIN_HIGH_PRIORITY=1 step bench_prio 300 $B
BDVCIL_DS_SIDE=1 step bench_ds 300 $B
step bench_side2 300 $B
BDVCIL_MAIN_HIGH_PRIORITY=1 step bench_prio2 300 $B
BDVCIL_DS_SIDE=1 step bench_ds2 300 $B
BDVCIL_DS_SIDE=1 BDVCIL_MAIN_HIGH_PRIORITY=1 step bench_both 300 $B
step pytest_model 900 python -m pytest tests/test_model_gpu.py tests/test_task_loop_gpu.py tests/test_ddp_gpu.py -m gpu -q -x
BDVCIL_DS_SIDE=1 step pytest_model_ds 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "train_step or determin or full_size"
