# Side-stream operand lifetime: FIFO + main-stream wait (BDVCIL_SIDE_LAG) against record_stream() (0): step time and
# the caching allocator's footprint, alternating runs in one call.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['value'], 'clips/s', d['ms_per_step'], 'ms', d['config']['hbm'])
PY
}
for i in 1 2 3; do
for lag in 64 0 24; do
BDVCIL_SIDE_LAG=$lag timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/lag_${lag}_$i.log 2>&1
echo -n "[default b32 lag=$lag #$i] rc=$? "; show gpurun_out/lag_${lag}_$i.log
done
done
for lag in 64 0 24; do
BDVCIL_SIDE_LAG=$lag timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16 --batch 64 > gpurun_out/lag64_$lag.log 2>&1
echo -n "[bf16 b64 lag=$lag] rc=$? "; show gpurun_out/lag64_$lag.log
BDVCIL_SIDE_LAG=$lag timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16x1 --batch 64 > gpurun_out/lag64x1_$lag.log 2>&1
echo -n "[bf16x1 b64 lag=$lag] rc=$? "; show gpurun_out/lag64x1_$lag.log
done
