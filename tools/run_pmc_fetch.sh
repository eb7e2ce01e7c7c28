set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01b
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01b/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01b/pmc_fetch.json 2> gpurun_out/r01b/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/r01b/pmc_hit -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/r01b/pmc_hit.json 2> gpurun_out/r01b/pmc_hit.err
python tools/pmc_traffic.py gpurun_out/r01b/pmc_fetch gpurun_out/r01b/pmc_fetch > gpurun_out/r01b/traffic.json
