"""Host-side cost of one training step: wall time the Python thread needs to enqueue forward + backward + SGD after a
device synchronize (the device then runs ~72 ms).  If this approaches the device time the step becomes launch-bound as
soon as the host is shared (8 ranks on one node)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bdvcil_amd as bd  # noqa: E402
from bench import model_cfg  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model = bd.build_model(model_cfg(50, 101, 'SimpleLinear', 'CrossEntropyLoss', 0.5)).to(dev)
    opt = bd.build_optimizer(model, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                         paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
    engine = bd.TrainEngine(model, opt)
    batch = dict(imgs=torch.randn(32, 8, 3, 224, 224, device=dev), label=torch.randint(0, 101, (32, 1), device=dev))
    for _ in range(3):
        engine.step(batch)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.step(batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    parts = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        losses = bd.base_training_step(model, batch)
        t1 = time.perf_counter()
        losses['loss'].backward()
        t2 = time.perf_counter()
        opt.step()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        parts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    print('host ms  forward / backward / sgd / drain:')
    for p_ in parts:
        print('   ' + ' / '.join(f'{v:6.1f}' for v in p_))
    print('host enqueue ms per step:', ' '.join(f'{h:.1f}' for h in host))
    print('step ms (host + drain)  :', ' '.join(f'{t:.1f}' for t in total))
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)
    print(f'process CPU time so far: user {r.ru_utime:.1f} s, sys {r.ru_stime:.1f} s; threads {torch.get_num_threads()}')


if __name__ == '__main__':
    main()
