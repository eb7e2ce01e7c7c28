set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
timeout -k 10 300 python bench.py > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-experimental > gpurun_out/r01/stats_bench.json 2> gpurun_out/r01/stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-experimental > gpurun_out/r01/pmc_fetch.json 2> gpurun_out/r01/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-experimental > gpurun_out/r01/pmc_write.json 2> gpurun_out/r01/pmc_write.err
python tools/pmc_traffic.py gpurun_out/r01/pmc_fetch gpurun_out/r01/pmc_write > gpurun_out/r01/traffic.json
find gpurun_out/r01 -name "*counter_collection.csv" -size +8M -delete
ls -la gpurun_out/r01 gpurun_out/r01/stats/* | head -40
