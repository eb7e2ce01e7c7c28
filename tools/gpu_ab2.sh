cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_conv_gpu.py -m gpu -q -x > gpurun_out/pytest_ops.log 2>&1
echo "[pytest_ops] rc=$?"; tail -n 2 gpurun_out/pytest_ops.log
timeout -k 10 500 python tools/ab_step.py 4 10 > gpurun_out/ab_step.log 2>&1; tail -5 gpurun_out/ab_step.log
