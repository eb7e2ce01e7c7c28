"""Two ranks on one MI355X (gloo over device tensors: RCCL refuses two ranks on the same device) driving the real
HIP training step through the bucketed gradient reducer, and the CIL task loop with its per-rank epoch partition.

What the eight-GPU run relies on and one rank cannot show: the reduced gradient equals the mean of the ranks' local
gradients, the weights stay bit-identical across ranks over several clip + SGD steps although BatchNorm statistics are
per rank, and only rank 0 writes the run's files."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bdvcil_amd as bd
    from test_task_loop_gpu import _config, _model_cfg
    dev = torch.device('cuda:0')

    # ---- one model, two ranks, different clips ----
    torch.manual_seed(100 + rank)                       # ranks start different: the broadcast must fix it
    mcfg = _model_cfg(5)
    mcfg['cls_head']['dropout_ratio'] = 0.0
    model = bd.build_model(mcfg).to(dev)
    bd.broadcast_parameters(model)
    twin = bd.build_model(mcfg).to(dev)                 # same weights, no reducer: the rank's local gradient
    twin.load_state_dict(model.state_dict())
    g = torch.Generator().manual_seed(7 + rank)
    batch = dict(imgs=torch.randn(3, 8, 3, 64, 64, generator=g).to(dev), label=torch.randint(0, 5, (3, 1), generator=g).to(dev))
    reducer = bd.GradAllReducer(model, bucket_cap_mb=4.0)
    opt = bd.build_optimizer(model, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                         paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
    twin(batch['imgs'], batch['label'])['loss_cls'].backward()
    local = [p.grad.detach().clone() for p in twin.parameters()]
    model(batch['imgs'], batch['label'])['loss_cls'].backward()
    reducer.finish()
    torch.cuda.synchronize()
    reduced = [(p.grad * reducer.grad_scale).detach().clone() for p in model.parameters()]
    engine = bd.TrainEngine(model, opt, grad_clip=1.0, reducer=reducer)
    for _ in range(3):
        engine.step(batch)
    torch.cuda.synchronize()
    torch.save({'local': [t.cpu() for t in local], 'reduced': [t.cpu() for t in reduced], 'buckets': len(reducer.buckets),
                'params': [p.detach().cpu() for p in model.parameters()],
                'running_mean': model.state_dict()['backbone.conv1.bn.running_mean'].cpu()}, os.path.join(out_dir, f'step_{rank}.pt'))
    reducer.remove()

    # ---- the task loop on two ranks ----
    import pathlib
    import bdvcil_amd.task_loop as TL
    tmp = pathlib.Path(out_dir) / 'run'
    if rank == 0:
        tmp.mkdir()
    dist.barrier()
    for r in range(world):                              # _config writes the annotation files: one rank at a time
        if r == rank:
            cfg = _config(tmp, ending_task=1, num_epochs_per_task=2)
        dist.barrier()
    torch.manual_seed(200 + rank)
    loop = TL.CILTaskLoop(cfg, TL.SyntheticClipLoader('cuda', size=64, seed=5), device='cuda', seed=3, log=lambda *a: None)
    writes = []
    real_save = torch.save
    torch.save = lambda obj, path, *a, **k: (writes.append(str(path)), real_save(obj, path, *a, **k))[1]
    try:
        hist = loop.train()
    finally:
        torch.save = real_save
    torch.cuda.synchronize()
    real_save({'params': [p.detach().cpu() for p in loop.current_model.parameters()],
               'buffers': [b.detach().cpu() for b in loop.current_model.buffers()],
               'exemplars': [[v['frame_dir'] for v in e.video_infos] for e in loop.exemplar_datasets],
               'writes': writes, 'cnn': [h['cnn'].values for h in hist], 'nme': [h['nme'].values for h in hist]},
              os.path.join(out_dir, f'loop_{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu(tmp_path):
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=420)
        assert p.exitcode == 0
    a, b = (torch.load(tmp_path / f'step_{r}.pt', weights_only=True) for r in range(2))
    assert a['buckets'] > 1
    worst = 0.0
    for ra, rb, la, lb in zip(a['reduced'], b['reduced'], a['local'], b['local']):
        assert torch.equal(ra, rb)                                             # both ranks hold the same reduced gradient
        want = 0.5 * (la.double() + lb.double())
        worst = max(worst, float((ra.double() - want).abs().max() / (want.abs().max() + 1e-12)))
    assert worst < 1e-5                                                        # ... and it is the mean of the local ones
    assert any(not torch.equal(la, lb) for la, lb in zip(a['local'], b['local']))
    assert all(torch.equal(pa, pb) for pa, pb in zip(a['params'], b['params']))  # weights identical after 3 clip+SGD steps
    assert not torch.equal(a['running_mean'], b['running_mean'])               # BatchNorm statistics are per rank (no SyncBN)

    la, lb = (torch.load(tmp_path / f'loop_{r}.pt', weights_only=True) for r in range(2))
    assert all(torch.equal(pa, pb) for pa, pb in zip(la['params'], lb['params']))
    assert all(torch.equal(pa, pb) for pa, pb in zip(la['buffers'], lb['buffers']))   # buffers follow rank 0 after each fit
    assert la['exemplars'] == lb['exemplars'] and len(la['exemplars']) == 2
    assert la['cnn'] == lb['cnn'] and la['nme'] == lb['nme']
    assert len(la['writes']) == 4 and lb['writes'] == []                       # 2 checkpoints + 2 class-mean files, rank 0 only
    assert sorted(os.listdir(tmp_path / 'run' / 'work' / 'exemplar')) == ['exemplar_task_0.txt', 'exemplar_task_1.txt', 'tmp_exemplars.txt']
