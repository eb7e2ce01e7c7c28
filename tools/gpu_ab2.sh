cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OLD=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_old.so
timeout -k 10 900 python -m pytest tests/test_conv_gpu.py tests/test_conv_sites_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/pytest_model.log 2>&1
echo "[pytest conv+sites+model] rc=$?"; tail -n 3 gpurun_out/pytest_model.log | cut -c1-300
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_new$i.log 2>&1
echo "[new$i] rc=$?"; tail -n 1 gpurun_out/bench_new$i.log | cut -c58-110
BDVCIL_LIB_PATH=$OLD timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_old$i.log 2>&1
echo "[old$i] rc=$?"; tail -n 1 gpurun_out/bench_old$i.log | cut -c58-110
done
