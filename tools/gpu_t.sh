cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
for nt in 2 1 0; do
BDVCIL_BN_NT=$nt timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_nt${nt}_$i.log 2>&1
echo "[BDVCIL_BN_NT=$nt #$i] $(tail -n 1 gpurun_out/b_nt${nt}_$i.log | cut -c58-100)"
done
done
