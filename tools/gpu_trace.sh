cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
BDVCIL_WGRAD_SIDE_STREAM=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/trace/bench.json 2> gpurun_out/trace/err.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60] for r in rows]
# last step = last third
n = len(rows); start = n - n // 3
out = []
for i in range(start, n):
    nm = names[i]
    if 'copyBuffer' in nm or 'FillFunctor' in nm or 'elementwise' in nm:
        out.append(f"{i-start:5d} {nm:50s} dur {(int(rows[i]['End_Timestamp'])-int(rows[i]['Start_Timestamp']))/1e3:6.1f}us  prev: {names[i-1][:40]:40s} next: {names[i+1][:40] if i+1<n else ''}")
open('gpurun_out/trace_small_kernels.txt', 'w').write('\n'.join(out) + '\n')
print(len(out), 'small kernels in the last step of', n - start)
st = [int(r['Start_Timestamp']) for r in rows[start:]]; en = [int(r['End_Timestamp']) for r in rows[start:]]
busy = sum(e - s for s, e in zip(st, en)); span = max(en) - st[0]
gaps = sorted(((st[i + 1] - en[i]) / 1e3, names[start + i][:36], names[start + i + 1][:36]) for i in range(len(st) - 1))
print(f'last step (one stream): span {span / 1e6:.2f} ms, kernels {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms; largest gaps (us):', gaps[-8:])
PY
rm -rf gpurun_out/trace
