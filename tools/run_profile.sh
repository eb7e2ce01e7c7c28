# Round profile of the bench command (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE separately, without any trace domain, as the MI355X guide prescribes).
#   R=r02 bash tools/run_profile.sh        -> gpurun_out/$R/{bench.json,stats/,stats_overlap/,traffic.json}
# The kernel-stats pass that bench.py's rocprof_avg_us reads runs with BDVCIL_WGRAD_SIDE_STREAM=0: with the weight gradients on
# their own stream (the default) two kernels share the GPU and every duration in the trace is stretched by its neighbour; the
# same pass with the default streams is kept beside it (stats_overlap).
set -e
R=${R:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/$R
mkdir -p gpurun_out/$R
timeout -k 10 400 python bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
BDVCIL_WGRAD_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$R/stats_bench.json 2> gpurun_out/$R/stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats_overlap -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$R/stats_overlap_bench.json 2> gpurun_out/$R/stats_overlap.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/$R/pmc_fetch.json 2> gpurun_out/$R/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/$R/pmc_write.json 2> gpurun_out/$R/pmc_write.err
python tools/pmc_traffic.py gpurun_out/$R/pmc_fetch gpurun_out/$R/pmc_write > gpurun_out/$R/traffic.json
# the other workloads of BASELINE.json (not the metric)
timeout -k 10 300 python bench.py --workload cil --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/$R/bench_cil.json 2> gpurun_out/$R/bench_cil.err
timeout -k 10 300 python bench.py --workload predict --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench_predict.json 2> gpurun_out/$R/bench_predict.err
timeout -k 10 300 python bench.py --workload i3d --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/$R/bench_i3d.json 2> gpurun_out/$R/bench_i3d.err
timeout -k 10 300 python bench.py --arith bf16x2 --no-cpu-baseline > gpurun_out/$R/bench_bf16x2.json 2> gpurun_out/$R/bench_bf16x2.err
timeout -k 10 300 python bench.py --arith bf16x1 --batch 64 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench_bf16x1.json 2> gpurun_out/$R/bench_bf16x1.err
timeout -k 10 300 python bench.py --arith bf16 --batch 64 --no-cpu-baseline > gpurun_out/$R/bench_bf16.json 2> gpurun_out/$R/bench_bf16.err
timeout -k 10 300 python bench.py --arith bf16 --workload cil --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/$R/bench_cil_bf16.json 2> gpurun_out/$R/bench_cil_bf16.err
timeout -k 10 300 python bench.py --arith bf16 --workload predict --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench_predict_bf16.json 2> gpurun_out/$R/bench_predict_bf16.err
timeout -k 10 300 python bench.py --depth 34 --no-cpu-baseline > gpurun_out/$R/bench_r34.json 2> gpurun_out/$R/bench_r34.err
timeout -k 10 300 python bench.py --arith f32mfma --no-cpu-baseline > gpurun_out/$R/bench_f32mfma.json 2> gpurun_out/$R/bench_f32mfma.err
find gpurun_out/$R -name "*counter_collection.csv" -size +8M -delete
find gpurun_out/$R -name "*kernel_trace.csv" -delete
ls -la gpurun_out/$R gpurun_out/$R/stats/* | head -40
