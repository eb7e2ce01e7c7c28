cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bf16_storage_gpu.py -m gpu -q -x > gpurun_out/pytest_bf16.log 2>&1
rc=$?; echo "[pytest bf16] rc=$rc"; tail -n 30 gpurun_out/pytest_bf16.log | cut -c1-250
