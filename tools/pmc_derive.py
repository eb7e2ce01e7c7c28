"""Derived SQ-counter columns per conv dispatch from the two --pmc passes of tools/run_pmc_sq.sh (tools/gpu_sq.sh):
    python tools/pmc_derive.py gpurun_out/sq_a.tsv gpurun_out/sq_b.tsv > profiles/rNN_sq_counters.tsv
mfma_util = SQ_INSTS_MFMA x (32 | 64 cycles for a bf16 | fp32 MFMA) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs);
valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA - 1; lds_inst_per_mfma = SQ_INSTS_LDS / SQ_INSTS_MFMA;
lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; wait / issue-stall / active = SQ_WAIT_ANY, SQ_WAIT_INST_ANY,
SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of a wave's life)."""
import sys


def read(path):
    rows, names = {}, None
    for line in open(path):
        f = line.rstrip('\n').split('\t')
        if names is None:
            names = f[1:]
            continue
        disp, rest = f[0].split(':', 1)
        rows[int(disp)] = (rest, dict(zip(names, map(float, f[1:]))))
    return rows


a, b = read(sys.argv[1]), read(sys.argv[2])
ka, kb = sorted(a), sorted(b)
print('# mfma_util = SQ_INSTS_MFMA x (32 | 64) / (GRBM_GUI_ACTIVE / 8 x 1024); valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA - 1; '
      'lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; parked / issue_stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / '
      'SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES')
cols = None
for da, db in zip(ka, kb):              # the two passes launch the same kernels in the same order
    (name, va), (name_b, vb) = a[da], b[db]
    if name.split(':')[0] != name_b.split(':')[0] or not (name.startswith('conv_') or name.startswith('wgrad_reduce')):
        continue
    v = dict(va)
    v.update(vb)
    mf = v.get('SQ_INSTS_MFMA', 0.0)
    cyc = 64.0 if ('_pl_' not in name and '_x3_' not in name) else 32.0
    gui = v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 * 1024.0
    wc = max(v.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    d = [('mfma_util', mf * cyc / gui if gui else 0.0), ('valu_per_mfma', v.get('SQ_INSTS_VALU', 0.0) / mf - 1 if mf else 0.0),
         ('lds_inst_per_mfma', v.get('SQ_INSTS_LDS', 0.0) / mf if mf else 0.0),
         ('lds_conflict', v.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(v.get('SQ_ACTIVE_INST_LDS', 0.0), 1.0)),
         ('parked', v.get('SQ_WAIT_ANY', 0.0) / wc), ('issue_stall', v.get('SQ_WAIT_INST_ANY', 0.0) / wc),
         ('active', v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc)]
    raw = sorted(v)
    if cols is None:
        cols = raw
        print('dispatch:kernel:grid', *[k for k, _ in d], *cols, sep='\t')
    print(f'{da}:{name}', *[f'{x:.3f}' for _, x in d], *[f'{v.get(c, 0):.4g}' for c in cols], sep='\t')
