# Round-3 evidence in one gpurun call: bench + rocprofv3 stats + PMC traffic (tools/run_profile.sh), SQ counters at six sites
# (tools/gpu_sq.sh), per-site tuning table, in-kernel stamps (diagnostic library), K-step / MFMA-rate microbenchmarks.
#   gpurun --timeout 1150 -- bash tools/gpu_round3.sh
export R=r03
bash tools/run_profile.sh > gpurun_out/r03_profile.log 2>&1
bash tools/gpu_sq.sh > gpurun_out/r03_sq.log 2>&1
python tools/pmc_derive.py gpurun_out/sq_a.tsv gpurun_out/sq_b.tsv > gpurun_out/r03/sq_counters.tsv 2>> gpurun_out/r03_sq.log
FUSED=1 timeout -k 10 300 python tools/tune_conv.py > gpurun_out/r03/tune_conv_fused.txt 2>&1
BDVCIL_LIB_PATH=background-debiased-video-cil_amd/csrc/libbdvcil_hip_stamps.so timeout -k 10 200 python tools/stamp_tiles.py > gpurun_out/r03/stamps.txt 2>&1
timeout -k 10 100 tools/ubench/kstep_parts.bin > gpurun_out/r03/ubench_kstep_parts.txt 2>&1
timeout -k 10 100 tools/ubench/mfma_rate.bin > gpurun_out/r03/ubench_mfma_rate.txt 2>&1
timeout -k 10 300 python bench.py --steps 64 --warmup 10 --no-cpu-baseline > gpurun_out/r03/bench_steps64.json 2> gpurun_out/r03/bench_steps64.err
ls gpurun_out/r03 | head -60
tail -c 400 gpurun_out/r03/bench.json
