cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sq; mkdir -p gpurun_out/sq
bash tools/run_pmc_sq.sh || exit 1
python tools/pmc_summarize.py gpurun_out/sq/a > gpurun_out/sq_a.tsv
python tools/pmc_summarize.py gpurun_out/sq/b > gpurun_out/sq_b.tsv
rm -rf gpurun_out/sq
wc -l gpurun_out/sq_a.tsv gpurun_out/sq_b.tsv
