cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "producer_batchnorm or bit_identical" > gpurun_out/pytest_model.log 2>&1
echo "[pytest pre] rc=$?"; tail -n 3 gpurun_out/pytest_model.log | cut -c1-300
