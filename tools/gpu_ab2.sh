cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_a$i.log 2>&1
echo "[default $i] rc=$?"; tail -n 1 gpurun_out/bench_a$i.log | cut -c58-110
BDVCIL_BN_NT=1 timeout -k 10 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench_b$i.log 2>&1
echo "[BN_NT $i] rc=$?"; tail -n 1 gpurun_out/bench_b$i.log | cut -c58-110
done
