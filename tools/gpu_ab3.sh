# 64x128 dgrad tiles for the conv1 dgrads of the first stages only (BDVCIL_DGRAD_64X128=2 / 3) against the default, alternating.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
for m in 0 2 3; do
BDVCIL_DGRAD_64X128=$m timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab3_${m}_$i.log 2>&1
echo "[mode $m #$i] rc=$? $(tail -n 1 gpurun_out/ab3_${m}_$i.log | cut -c58-135)"
done
done
