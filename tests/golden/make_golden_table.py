"""Generate golden accuracy tables from the reference's own ``libs/utils.py`` (AverageMeter, print_mean_accuracy).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_table.py

``libs/utils.py`` imports ``mmcv.utils.config.Config`` for a type annotation only; an empty class of that name is placed
in ``sys.modules`` for the import.  Only the inputs (per-task accuracy values and sample counts) and the resulting
strings are written to ``tests/golden/table_golden.json``.
"""
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'table_golden.json')


def main():
    mmcv, mu, mc = types.ModuleType('mmcv'), types.ModuleType('mmcv.utils'), types.ModuleType('mmcv.utils.config')
    mc.Config = type('Config', (), {})
    sys.modules.update({'mmcv': mmcv, 'mmcv.utils': mu, 'mmcv.utils.config': mc})
    spec = importlib.util.spec_from_file_location('ref_utils', os.path.join(REF, 'libs/utils.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = []
    for updates, classes in [
        ([[(91.25, 160)], [(88.0, 160), (71.5, 40)], [(80.125, 160), (64.0, 40), (97.75, 37)]], [51, 5, 5]),
        ([[(100.0, 3)]], [26]),
        ([[(12.5, 8)], [(0.0, 8), (100.0 / 3, 9)]], [2, 2]),
    ]:
        meters = []
        for task in updates:
            m = mod.AverageMeter()
            for val, n in task:
                m.update(val, n)
            meters.append(m)
        cases.append({'updates': updates, 'classes': classes, 'table': mod.print_mean_accuracy(meters, classes),
                      'avg': [m.avg for m in meters], 'sum': [m.sum for m in meters], 'count': [m.count for m in meters]})
    with open(OUT, 'w') as f:
        json.dump(cases, f, indent=1)
    print('wrote', OUT)


if __name__ == '__main__':
    main()
