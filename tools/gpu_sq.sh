cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sq
bash tools/run_pmc_sq.sh || exit 1
python tools/pmc_summarize.py gpurun_out/sq/a > gpurun_out/sq_a.tsv
python tools/pmc_summarize.py gpurun_out/sq/b > gpurun_out/sq_b.tsv
rm -rf gpurun_out/sq
timeout -k 10 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/bench.log 2>&1 || exit 1
tail -n 1 gpurun_out/bench.log | cut -c1-300
