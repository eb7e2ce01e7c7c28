cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
BDVCIL_FORCE_DIST=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i bench.py --gpus 1 --no-cpu-baseline > gpurun_out/bench_force_dist.log 2>&1
echo "[force dist under torch.distributed.run #$i] rc=$? $(tail -n 1 gpurun_out/bench_force_dist.log | cut -c58-100)"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_plain.log 2>&1
echo "[plain #$i] rc=$? $(tail -n 1 gpurun_out/bench_plain.log | cut -c58-100)"
GPU_MAX_HW_QUEUES=4 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_plain4.log 2>&1
echo "[plain, GPU_MAX_HW_QUEUES=4 #$i] rc=$? $(tail -n 1 gpurun_out/bench_plain4.log | cut -c58-100)"
done
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py tests/test_ops_gpu.py -m gpu -q -x > gpurun_out/pytest_ddp.log 2>&1
echo "[pytest ddp + ops] rc=$?"; tail -n 2 gpurun_out/pytest_ddp.log | cut -c1-200
