cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; python -c "import os, torch; print('cpu_count', os.cpu_count(), 'torch threads', torch.get_num_threads(), 'affinity', len(os.sched_getaffinity(0)))"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/pytest_gpu.log 2>&1
rc=$?
echo "[pytest_gpu] rc=$rc"; tail -n 18 gpurun_out/pytest_gpu.log | cut -c1-200
