cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_task_loop_gpu.py tests/test_ddp_gpu.py tests/test_i3d_gpu.py -m gpu -q -x > gpurun_out/pytest_model.log 2>&1
echo "[pytest model/taskloop/ddp/i3d] rc=$?"; tail -n 3 gpurun_out/pytest_model.log | cut -c1-300
timeout -k 10 400 python tools/ab_step.py 4 10 > gpurun_out/ab_step.log 2>&1; tail -5 gpurun_out/ab_step.log
