"""World-size-2 test of the bucketed gradient reducer on CPU (gloo): after finish(), every rank's p.grad holds
the SUM over ranks and, scaled by grad_scale = 1/world, equals the gradient of the mean loss over the global
batch (what DDP computes for the reference under ddp_spawn, libs/cil/cil.py:704-709)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.ReLU(), nn.Flatten(), nn.Linear(8 * 36, 5))


def _worker(rank, world, port, bucket_mb, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bdvcil_amd as bd
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _toy()
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                      # rank 1 starts different: broadcast must fix it
    bd.broadcast_parameters(model)
    model[0].weight.data = model[0].weight.data.contiguous(memory_format=torch.channels_last)
    reducer = bd.GradAllReducer(model, bucket_cap_mb=bucket_mb)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(4, 3, 8, 8, generator=g)
    y = torch.randint(0, 5, (4,), generator=g)
    xs, ys = x[rank * 2:(rank + 1) * 2], y[rank * 2:(rank + 1) * 2]
    out = {}
    for step in range(2):                        # two steps: buckets are reusable
        for p in model.parameters():
            p.grad = None
        nn.functional.cross_entropy(model(xs), ys).backward()
        reducer.finish()
        out[step] = [(p.grad * reducer.grad_scale).clone() for p in model.parameters()]
    q.put((rank, [t.numpy() for t in out[1]], len(reducer.buckets)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_mb', [25.0, 0.001])
def test_gloo_world2_allreduce_matches_full_batch(bucket_mb):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_mb, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # both ranks hold identical averaged gradients
    for a, b in zip(res[0][1], res[1][1]):
        assert (abs(a - b)).max() == 0
    if bucket_mb < 1:
        assert res[0][2] > 1                     # several buckets exercised
    # BN uses per-rank statistics (no SyncBN in the reference), so compare against the mean of per-rank losses
    model = _toy()
    g = torch.Generator().manual_seed(123)
    x = torch.randn(4, 3, 8, 8, generator=g)
    y = torch.randint(0, 5, (4,), generator=g)
    loss = 0.5 * (nn.functional.cross_entropy(model(x[:2]), y[:2]) + nn.functional.cross_entropy(model(x[2:]), y[2:]))
    loss.backward()
    for got, p in zip(res[0][1], model.parameters()):
        assert torch.allclose(torch.from_numpy(got), p.grad, rtol=1e-5, atol=1e-6)


def _unused_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bdvcil_amd as bd
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    a, b, c = nn.Linear(6, 6), nn.Linear(6, 6), nn.Linear(6, 6)
    model = nn.ModuleList([a, b, c])
    reducer = bd.GradAllReducer(model, bucket_cap_mb=25.0)
    x = torch.ones(2, 6) * (rank + 1)
    for step in range(2):
        for p in model.parameters():
            p.grad = None
        out = a(x)
        if rank == 0:
            out = out + b(x)              # b: used on rank 0 only; c: used by nobody
        out.sum().backward()
        reducer.finish()
    q.put((rank, None if b.weight.grad is None else b.weight.grad.clone().numpy(), c.weight.grad is None,
           a.weight.grad.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_unused_parameters_are_agreed_on_across_ranks():
    """A parameter one rank did not use gets the other rank's sum on BOTH ranks (so the weights cannot diverge); a parameter no
    rank used keeps ``.grad = None`` everywhere -- torch's DistributedDataParallel semantics (libs/cil/cil.py:704-709)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_unused_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, c0_none, a0), (_, b1, c1_none, a1) = res
    assert b0 is not None and b1 is not None and abs(b0 - b1).max() == 0 and abs(b0).max() > 0
    assert c0_none and c1_none
    assert abs(a0 - a1).max() == 0


def test_tail_bucket_split():
    """The bucket that closes last (earliest layers) is split so that only a small tail is reduced after backward."""
    import bdvcil_amd as bd
    torch.manual_seed(0)
    layers = [nn.Linear(64, 64, bias=False) for _ in range(10)]                   # 16 KB each, registration order = forward order
    model = nn.Sequential(*layers)
    r = bd.GradAllReducer(model, bucket_cap_mb=0.0625, tail_cap_mb=0.03125)       # 64 KB buckets, 32 KB tail
    sizes = [[p.numel() * 4 for p in b.params] for b in r.buckets]
    assert [sum(s) for s in sizes] == [65536, 65536, 32768]
    r.remove()
    r = bd.GradAllReducer(model, bucket_cap_mb=0.0625, tail_cap_mb=0.0234375)     # 24 KB tail: one 16 KB tensor fits
    assert [sum(p.numel() * 4 for p in b.params) for b in r.buckets] == [65536, 65536, 16384, 16384]
    # order: the tail bucket holds the FIRST layers of the model (their gradients arrive last)
    assert r.buckets[-1].params[0] is layers[0].weight
    covered = [p for b in r.buckets for p in b.params]
    assert len(covered) == 10 and len({id(p) for p in covered}) == 10
    r.remove()
    r = bd.GradAllReducer(model, bucket_cap_mb=0.0625, tail_cap_mb=0)             # disabled
    assert [sum(p.numel() * 4 for p in b.params) for b in r.buckets] == [65536, 65536, 32768]
    r.remove()
    r = bd.GradAllReducer(model, bucket_cap_mb=25.0, tail_cap_mb=0.001)           # nothing fits the tail: unchanged
    assert len(r.buckets) == 1
    r.remove()


def test_bench_self_launch_without_torchrun():
    """``python bench.py --gpus 2`` with no WORLD_SIZE in the environment starts its own ranks (the driver's N = 1 command
    line is a plain ``python bench.py``; the N > 1 record must not depend on an outer launcher).  The ranks run the
    launcher self-test: gloo rendezvous on 127.0.0.1, barrier-bracketed timing, MAX over ranks, one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--selftest-cpu'], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['value'] == 3.0 and rec['steps'] == 3      # all-reduce over both ranks: 1 + 2
    # the record proves itself: ranks / backend / buckets come from the communicator and the reducer, and the time the step spent
    # blocked in GradAllReducer.finish() is in it (bench.py writes the same keys into the GPU record)
    cfg = rec['config']
    assert cfg['parallelism'] == 'dp2' and cfg['rccl']['ranks'] == 2 and cfg['rccl']['backend'] == 'gloo'
    assert cfg['rccl']['buckets'] >= 2 and cfg['rccl']['payload_mb'] > 0 and len(cfg['rccl']['bucket_mb']) == cfg['rccl']['buckets']
    assert cfg['allreduce_exposed_ms'] is not None and cfg['allreduce_exposed_ms'] >= 0
    # a launcher that starts the wrong number of ranks is refused
    env2 = dict(env, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--selftest-cpu'], env=env2,
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE=3' in (bad.stderr + bad.stdout)
