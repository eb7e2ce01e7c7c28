# The default bench line (with cpu_baseline) and the config-5 lines of the final build.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 400 python bench.py > gpurun_out/r02/bench.json 2> gpurun_out/r02/bench.err
echo "[bench] rc=$?"; tail -n 1 gpurun_out/r02/bench.json | cut -c1-200
timeout -k 10 300 python bench.py --arith bf16 --batch 64 --no-cpu-baseline > gpurun_out/r02/bench_bf16.json 2> gpurun_out/r02/bench_bf16.err
echo "[bench bf16] rc=$?"; tail -n 1 gpurun_out/r02/bench_bf16.json | cut -c100-220
timeout -k 10 300 python bench.py --arith bf16x1 --batch 64 --no-cpu-baseline > gpurun_out/r02/bench_bf16x1.json 2> gpurun_out/r02/bench_bf16x1.err
echo "[bench bf16x1] rc=$?"; tail -n 1 gpurun_out/r02/bench_bf16x1.json | cut -c100-220
