cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --arith bf16x1 --batch 64 > gpurun_out/b64_$i.log 2>&1
echo "[bf16x1 b64 #$i] rc=$?"; python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/b64_$i.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config']['hbm'])
PY
done
rocm-smi --showpower --showclocks --showmeminfo vram 2>/dev/null | head -30
