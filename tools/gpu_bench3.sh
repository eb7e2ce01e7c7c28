# Three default bench lines of the current build (and one with every other workload flag untouched) in one call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_n$i.log 2> gpurun_out/bench_n$i.err
echo "[bench $i] rc=$?"; tail -n 1 gpurun_out/bench_n$i.log | cut -c58-110
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_nokt.log 2>&1
echo "[bench no-kernel-timing] rc=$?"; tail -n 1 gpurun_out/bench_nokt.log | cut -c58-110
