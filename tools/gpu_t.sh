cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_chunked_apply.py > gpurun_out/chunked_apply.log 2>&1
echo "rc=$?"; grep -v amdgpu gpurun_out/chunked_apply.log | tail -8
