cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
step() {  # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > gpurun_out/$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"
  tail -n 8 gpurun_out/$name.log | cut -c1-330
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] killed at its limit: stopping"; exit $rc; fi
}
step ab_step 400 python tools/ab_step.py 3 10
