# One gpurun call: the -m gpu suite (all of it, no -x), then the K-loop microbenchmark.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 6 gpurun_out/$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step pytest_gpu 1000 python -m pytest tests -m gpu -q --durations=15
step ubench_x3p 500 tools/ubench/gemm_x3p.bin
