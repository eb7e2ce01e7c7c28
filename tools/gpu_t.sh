cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "[pytest_gpu] rc=$rc"; tail -n 6 gpurun_out/pytest_gpu.log | cut -c1-200
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_new_$i.log 2>&1
echo "[default new #$i] $(tail -n 1 gpurun_out/b_new_$i.log | cut -c58-100)"
BDVCIL_EW_BLOCKS=4096 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b_old_$i.log 2>&1
echo "[default, bn passes capped at 4096 blocks #$i] $(tail -n 1 gpurun_out/b_old_$i.log | cut -c58-100)"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/b_bf16.log 2>&1; echo "[bf16 b64] $(tail -n 1 gpurun_out/b_bf16.log | cut -c100-150)"
timeout -k 10 300 python bench.py --no-cpu-baseline --workload cil --steps 8 --warmup 2 > gpurun_out/b_cil.log 2>&1; echo "[cil] $(tail -n 1 gpurun_out/b_cil.log | cut -c100-160)"
timeout -k 10 300 python bench.py --no-cpu-baseline --workload i3d --steps 8 --warmup 2 > gpurun_out/b_i3d.log 2>&1; echo "[i3d] $(tail -n 1 gpurun_out/b_i3d.log | cut -c90-150)"
