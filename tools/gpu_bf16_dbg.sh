cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python tools/debug/bf16_grad_diff.py 50 224 4 > gpurun_out/bf16_grad_diff4.log 2>&1
echo "[grad diff B=4] rc=$?"; grep -v amdgpu gpurun_out/bf16_grad_diff4.log | sed -n '1,5p;48,60p' | cut -c1-150
timeout -k 10 300 python tools/debug/bf16_grad_diff.py 50 224 16 > gpurun_out/bf16_grad_diff16.log 2>&1
echo "[grad diff B=16] rc=$?"; grep -v amdgpu gpurun_out/bf16_grad_diff16.log | sed -n '1,5p;48,60p' | cut -c1-150
