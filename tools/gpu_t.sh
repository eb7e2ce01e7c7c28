cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
U2=$GRAFT_REPO_ROOT/background-debiased-video-cil_amd/csrc/libbdvcil_hip_u2.so
timeout -k 10 200 python tools/bench_bn.py > gpurun_out/bn_u1.log 2>&1; echo "[bench_bn U=1] $(tail -n 1 gpurun_out/bn_u1.log)"
BDVCIL_LIB_PATH=$U2 timeout -k 10 200 python tools/bench_bn.py > gpurun_out/bn_u2.log 2>&1; echo "[bench_bn U=2] $(tail -n 1 gpurun_out/bn_u2.log)"
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_u1_$i.log 2>&1
echo "[U=1 #$i] $(tail -n 1 gpurun_out/b_u1_$i.log | cut -c58-100)"
BDVCIL_LIB_PATH=$U2 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_u2_$i.log 2>&1
echo "[U=2 #$i] $(tail -n 1 gpurun_out/b_u2_$i.log | cut -c58-100)"
done
