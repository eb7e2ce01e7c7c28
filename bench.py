#!/usr/bin/env python
"""Headline benchmark: clips/s, forward + backward + fused SGD step, TSM-ResNet50, 8x3x224x224 clips,
batch 32 per GPU, fp32 tensors and results (BASELINE.json configs[1]; synthetic clips, random-init weights).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            # N > 1 without a launcher: starts torch.distributed.run itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
``roofline`` (dominant kernel = an implicit-GEMM conv kernel; algorithmic FLOP / HIP-event time measured inside the timed
region, priced against the peak of the MFMA instruction it issues) and ``cpu_baseline`` (the CPU oracle timed on this
box's host cores, N=1 only).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# before anything starts the HIP runtime (bdvcil_amd/__init__.py says why): enough hardware queues for the streams of the
# distributed path, so that the weight-gradient stream keeps a queue of its own
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

R50_FLOP_PER_CLIP = 194.29e9       # BASELINE.md section 2 (fwd+bwd, conv MACs only, stem dgrad excluded)
R50_KD_FLOP_PER_CLIP = 259.7e9     # + the frozen previous model's forward
R50_FWD_FLOP_PER_CLIP = 65.39e9    # forward only (2 x 8 x 4.0871 GMAC)
# BASELINE config 4: I3D-ResNet50 on 32 x 224 x 224 clips: 33.127 GMAC of convolutions per clip (oracle/i3d_oracle.py), of
# which 9.441 GMAC in the 5x7x7 stem (no input gradient): fwd + bwd = 6 x 33.127 - 2 x 9.441 GFLOP
I3D_FLOP_PER_CLIP = 6 * 33.127399424e9 - 2 * 9.44111616e9
PEAK_F32_MFMA = 157.3e12           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_BF16_MFMA = 2516.6e12         # v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16 on 1024 SIMDs at 2.4 GHz, dense
# An fp32 product formed from three bf16 pieces per operand costs six bf16 MFMA products: the MFMA-bound rate of the
# default conv kernels in fp32-equivalent FLOP (2*M*N*K per GEMM) is the bf16 dense peak / 6.
PEAK_BF16X3 = PEAK_BF16_MFMA / 6.0
PROFILE_ROUND = 'r03'


def model_cfg(depth, num_classes, head, loss, dropout):
    in_ch = 2048 if depth == 50 else 512
    inc = dict(type=head, out_features=num_classes)
    if head == 'LocalSimilarityClassifier':
        inc['nb_proxies'] = 1
    return dict(type='CILRecognizer2D',
                backbone=dict(type='ResNetTSM', pretrained=None, depth=depth, norm_eval=False, num_segments=8, shift_div=8),
                cls_head=dict(type='IncrementalTSMHead', num_classes=num_classes, in_channels=in_ch, inc_head_config=inc,
                              num_segments=8, loss_cls=dict(type=loss), spatial_type='avg',
                              consensus=dict(type='AvgConsensus', dim=1), dropout_ratio=dropout, init_std=0.001, is_shift=True),
                train_cfg=None, test_cfg=dict(average_clips='prob'))


class ConvTimer:
    """HIP-event timing of every conv launch on the stream it is launched on (torch's current stream).  An event pair
    spans one C-ABI call: the main kernel plus its K-split fix-up when the planner uses one.  The weight-gradient main
    kernels (``conv_wgrad_partial``) and their batched split-K reduction (``wgrad_reduce_batched``, one launch per stage)
    are separate calls and separate lines."""

    def __init__(self):
        self.records = []      # (tag, flops, start_event, end_event, bytes)
        self.enabled = False
        self.only = None       # when set: event pairs only around the launches of this kernel
        self._tags = {}

    def wrap(self, K):
        timer = self

        def kernel_tag(kind, g, x3):
            key = (kind, g.key(), bool(x3), K.PIECES)
            tag = timer._tags.get(key)
            if tag is None:
                tag = timer._tags[key] = K.conv_kernel_name(g, kind, x3).replace(' ', '')
            return tag

        def make(kind, fn, geom_pos, flag):
            def timed(*a, **kw):
                if not timer.enabled:
                    return fn(*a, **kw)
                g = a[geom_pos]
                g.act_dtype = 1 if a[0].dtype == torch.bfloat16 else 0      # (the call sets it too: the kernel name depends on it)
                x3 = kw.get('x3')
                x3 = getattr(K, flag) if x3 is None else x3
                tag = kernel_tag(kind, g, x3)
                if timer.only is not None and tag != timer.only:
                    return fn(*a, **kw)
                cin = 3 if g.Cin == 4 else g.Cin
                flops = 2.0 * g.N * g.Ho * g.Wo * g.Cout * g.R * g.S * max(g.Rt, 1) * cin
                # algorithmic bytes of the call: every operand and result tensor once (fp32), plus what the fused epilogues stream
                n_in = g.N * max(g.st_t, 1) * g.H * g.W * g.Cin if g.Rt > 1 else g.N * g.H * g.W * g.Cin
                n_out = g.N * g.Ho * g.Wo * g.Cout
                n_w = g.Cout * g.R * g.S * max(g.Rt, 1) * g.Cin
                nbytes = 4.0 * (n_in + n_out + n_w)
                if kind == 'dgrad':
                    nbytes += 4.0 * n_in * ((kw.get('add_src') is not None) + (kw.get('bn_stats') is not None))
                if kind == 'fprop' and kw.get('affine') is not None and kw['affine'][2] is not None:
                    nbytes += 4.0 * n_out
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **kw)
                e1.record()
                timer.records.append((tag, flops, e0, e1, nbytes))
                return out
            return timed

        def timed_reduce(fn):
            def timed(*a, **kw):
                if not timer.enabled or timer.only is not None:
                    return fn(*a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = fn(*a, **kw)
                e1.record()
                timer.records.append(('wgrad_reduce_batched_kernel', 0.0, e0, e1, 0.0))
                return out
            return timed

        K.conv_fprop = make('fprop', K.conv_fprop, 2, 'FPROP_X3')
        K.conv_dgrad = make('dgrad', K.conv_dgrad, 2, 'DGRAD_X3')
        # conv_wgrad (one-call form: the stem) runs the same main kernels; with the bf16-piece arithmetic it goes through
        # conv_wgrad_partial + wgrad_reduce_batched, which are timed below, so it is only wrapped for the fp32-MFMA path
        inner_wgrad = K.conv_wgrad
        timed_wgrad = make('wgrad', inner_wgrad, 2, 'WGRAD_X3')

        def conv_wgrad(*a, **kw):
            x3 = kw.get('x3')
            return inner_wgrad(*a, **kw) if (K.WGRAD_X3 if x3 is None else x3) else timed_wgrad(*a, **kw)
        K.conv_wgrad = conv_wgrad
        K.conv_wgrad_partial = make('wgrad', K.conv_wgrad_partial, 2, 'WGRAD_X3')
        K.wgrad_reduce_batched = timed_reduce(K.wgrad_reduce_batched)

    def summary(self, records=None):
        by = {}
        for tag, flops, e0, e1, nbytes in (self.records if records is None else records):
            ms = e0.elapsed_time(e1)
            d = by.setdefault(tag, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d['launches'] += 1
            d['ms'] += ms
            d['flops'] += flops
            d['bytes'] += nbytes
        return by


def pmc_traffic(tag):
    """HBM-side bytes per launch of kernel ``tag`` from the committed rocprofv3 --pmc passes of this same command
    (profiles/<round>_traffic.json, produced by tools/run_profile.sh + tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in
    separate passes, KiB units, FETCH_SIZE doubled on gfx950).  None when the file has no entry."""
    path = os.path.join(ROOT, 'profiles', f'{PROFILE_ROUND}_traffic.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        kernels = json.load(f).get('kernels', {})
    base = tag.replace(' ', '').rstrip('>')                # a tag may stop before trailing template flags (wgrad: INCR)
    tot, launches = 0.0, 0
    for name, v in kernels.items():
        n = name.replace(' ', '')
        if n.startswith(base) and n[len(base):len(base) + 1] in ('>', ','):
            tot += (v.get('hbm_bytes_per_launch') or 0) * v.get('launches', 1)
            launches += v.get('launches', 1)
    return round(tot / launches) if launches else None


def rocprof_avg_us(tag):
    """Average duration (us) of kernel ``tag`` in the committed ``rocprofv3 --kernel-trace --stats`` summary of this same
    command (profiles/<round>_kernel_stats.csv); None when absent.  Reported beside the live HIP-event time."""
    import csv
    path = os.path.join(ROOT, 'profiles', f'{PROFILE_ROUND}_kernel_stats.csv')
    if not os.path.exists(path):
        return None
    base = tag.replace(' ', '').rstrip('>')
    tot_ns, calls = 0.0, 0
    with open(path) as f:
        for row in csv.DictReader(f):
            n = row['Name'].replace(' ', '')
            k = n.find(base)
            if k >= 0 and n[k + len(base):k + len(base) + 1] in ('>', ','):
                tot_ns += float(row['TotalDurationNs'])
                calls += int(row['Calls'])
    return round(tot_ns / calls / 1e3, 2) if calls else None


def cpu_baseline(depth, num_classes, head, loss, budget_s=25.0):
    """The CPU oracle (pure-torch restatement of the reference path; the reference's own trainer needs mmaction2 /
    Lightning, which do not exist here) timed on the host cores: fwd + bwd + SGD step, fp32, bounded sample."""
    from oracle import tsm_oracle as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))     # a 1-GPU box owns a 16-CPU share of the host
    torch.manual_seed(0)
    cfg = O.r50_cfg(num_classes=num_classes, depth=depth, head=head, loss=loss, dropout_ratio=0.5)
    model = O.build_model(cfg)
    model.train()
    opt = O.build_sgd(model)
    B = 2
    g = torch.Generator().manual_seed(0)
    imgs = torch.randn(B, 8, 3, 224, 224, generator=g)
    labels = torch.randint(0, num_classes, (B, 1), generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        model(imgs, labels)['loss_cls'].backward()
        opt.step()

    step()                                   # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 3 or (time.perf_counter() - t0 < budget_s and n < 12):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return dict(value=round(B * n / dt, 3), unit='clips/s', cores=torch.get_num_threads(), kind='port',
                sample=f'TSM-R{depth} fwd+bwd+SGD, batch {B} x 8x3x224x224, {n} steps in {dt:.1f}s on {os.cpu_count()} host CPUs '
                       f'(torch {torch.__version__} CPU, fp32)')


_JSON_FD = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner when the first
    communicator comes up): from here on file descriptor 1 is the process's stderr, and ``emit`` writes the line to the real stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def self_launch(n, argv):
    """``python bench.py --gpus N`` without torchrun: run ``torch.distributed.run`` with N ranks on 127.0.0.1 as a child
    process (this process never initialises the GPU), pass the child's output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: required by RCCL on this driver
    return subprocess.run(cmd, env=env).returncode


def dist_record(reducer, world):
    """What proves an N > 1 record: the communicator's own view (ranks, backend, buckets, payload) -- not WORLD_SIZE -- and the time
    per step the compute stream spent blocked in ``GradAllReducer.finish()`` (HIP events around its waits; 0 when the collectives hid
    under backward).  Replaces the implicit DDP reducer of libs/cil/cil.py:704-709."""
    if reducer is None:
        return {'parallelism': f'dp{world}'}
    d = reducer.describe()
    exposed = reducer.exposed_ms()
    return {'parallelism': f'dp{d["ranks"]}', 'rccl': d, 'allreduce_exposed_ms': None if exposed is None else round(exposed, 4)}


def selftest_cpu(args, world, rank):
    """What the ranks started by ``self_launch`` do in the CPU test of the launcher: the rendezvous, the barrier-bracketed
    timing with the MAX over ranks, and rank 0 printing the one line -- over gloo, with no GPU and no kernels."""
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bdvcil_amd as bd
    torch.manual_seed(0)
    toy = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
    bd.broadcast_parameters(toy)
    reducer = bd.GradAllReducer(toy, bucket_cap_mb=0.001)
    reducer.timing = True
    dist.barrier()
    t0 = time.perf_counter()
    x = torch.ones(4) * (rank + 1)
    dist.all_reduce(x)
    for _ in range(args.steps):                  # the reducer's own path over gloo: bucketed all-reduce + finish()
        for p in toy.parameters():
            p.grad = None
        toy(torch.randn(4, 64)).sum().backward()
        reducer.finish()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        emit({'metric': 'launcher selftest', 'value': float(x[0]), 'n_gpus': world, 'steps': args.steps,
              'warmup': args.warmup, 'ms_per_step': float(t) * 1e3, 'config': dist_record(reducer, world)})
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU')
    ap.add_argument('--classes', type=int, default=101)
    ap.add_argument('--head', default='SimpleLinear', choices=['SimpleLinear', 'LocalSimilarityClassifier'])
    ap.add_argument('--loss', default='CrossEntropyLoss', choices=['CrossEntropyLoss', 'LSCLoss'])
    ap.add_argument('--dropout', type=float, default=0.5)
    ap.add_argument('--workload', default='ce', choices=['ce', 'cil', 'predict', 'i3d'],
                    help="'ce': BASELINE config 2 (the metric); 'cil': config 3 step = uint8 background-mix front-end, LSC head + "
                         "LSCLoss, feature-KD against a frozen previous model (task >= 1), clip 1.0, SGD; 'predict': eval forward "
                         "+ representations of BaseCIL.predict_step (SURVEY section 8(f) rank 1), no backward; 'i3d': BASELINE config 4, "
                         "I3D-ResNet50 fwd+bwd+SGD on 32x3x224x224 clips (default batch 16)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--arith', default='bf16x3', choices=['bf16x3', 'f32mfma', 'bf16x2', 'bf16x1', 'bf16'],
                    help="conv arithmetic: 'bf16x3' (default; fp32 products from three bf16 pieces per operand, six bf16 MFMA "
                         "products, fp32 accumulate), 'f32mfma' (v_mfma_f32_32x32x2_f32 kernels), or the REDUCED-PRECISION 'bf16x1' of BASELINE "
                         "config 5 (operands rounded to bf16, one MFMA product, fp32 accumulate and tensors; use with --batch 64; "
                         "reported with dtype bf16, never as the headline metric)")
    ap.add_argument('--no-alt-arith', action='store_true',
                    help="skip the second timed run of the same step in the 'bf16x2' arithmetic (two bf16 pieces per operand, three "
                         "MFMA products; reported under `alt_arith`, never as `value`)")
    ap.add_argument('--selftest-cpu', action='store_true', help=argparse.SUPPRESS)   # launcher test: gloo ranks, no GPU
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # Started without a launcher: this process has not touched the GPU (importing torch does not), so it may start
        # one fresh process per GPU -- what the reference's ddp_spawn does (libs/cil/cil.py:704-709) -- and relay rank 0's line.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start exactly --gpus ranks')
    quiet_stdout()
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # BDVCIL_FORCE_DIST=1 exercises the RCCL path (process group, bucketed all-reduce, barrier) on a single rank
    force_dist = world == 1 and os.environ.get('BDVCIL_FORCE_DIST', '0') != '0'
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import bdvcil_amd as bd
    from bdvcil_amd import kernels as K

    K.set_conv_arith(args.arith)
    timer = ConvTimer()
    if not args.no_kernel_timing:
        timer.wrap(K)

    cil = args.workload == 'cil'
    predict = args.workload == 'predict'
    i3d = args.workload == 'i3d'
    if i3d and args.batch == 32:
        args.batch = 16                       # BASELINE config 4: batch 16 per GPU
    if cil or predict:
        args.head, args.loss = 'LocalSimilarityClassifier', 'LSCLoss'
    torch.manual_seed(0)
    if i3d:
        cfg = dict(type='Recognizer3D',
                   backbone=dict(type='ResNet3d', pretrained2d=True, pretrained=None, depth=50, conv1_kernel=(5, 7, 7), conv1_stride_t=2,
                                 pool1_stride_t=2, conv_cfg=dict(type='Conv3d'), norm_eval=False,
                                 inflate=((1, 1, 1), (1, 0, 1, 0), (1, 0, 1, 0, 1, 0), (0, 1, 0)), zero_init_residual=False),
                   cls_head=dict(type='I3DHead', num_classes=args.classes, in_channels=2048, spatial_type='avg',
                                 dropout_ratio=args.dropout, init_std=0.01),
                   train_cfg=None, test_cfg=dict(average_clips='prob'))         # configs/_base_/models/i3d_r50.py:1-27
    else:
        cfg = model_cfg(args.depth, args.classes, args.head, args.loss, args.dropout)
    model = bd.build_model(cfg).to(dev)
    model.train()
    reducer = None
    if use_dist:
        bd.broadcast_parameters(model)
        reducer = bd.GradAllReducer(model, bucket_cap_mb=float(os.environ.get('BDVCIL_BUCKET_MB', '25')))
        reducer.timing = True
    opt = bd.build_optimizer(model, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                         paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9, weight_decay=1e-4))
    engine = bd.TrainEngine(model, opt, grad_clip=1.0 if cil else None, reducer=reducer)
    if predict:
        model.eval()
        predictor = bd.ReprPredictor(model)

        class _Predict:                       # same interface as TrainEngine.step for the timing loop below
            def step(self, b, _loss_fn=None):
                r = predictor.predict_step(b)
                return {'loss_cls': r['cls_score'].sum() * 0.0, 'repr_': r['repr_']}
        engine = _Predict()

    g = torch.Generator().manual_seed(1000 + rank)
    labels = torch.randint(0, args.classes, (args.batch, 1), generator=g).to(dev)
    loss_fn = None
    if cil:
        # uint8 frames + background + mix mask -> fused front-end; teacher = same
        # architecture (seed 1); KD modules / weights / scale of configs/ucf101/bgmix_plus_randAug/...py:88-89
        frames = torch.randint(0, 256, (args.batch, 8, 224, 224, 3), generator=g, dtype=torch.uint8).to(dev)
        bg = torch.randint(0, 256, (args.batch, 224, 224, 3), generator=g, dtype=torch.uint8).to(dev)
        mix = (torch.rand(args.batch, generator=torch.Generator().manual_seed(1)) < 0.25).to(dev)
        torch.manual_seed(1)
        prev = bd.build_model(model_cfg(args.depth, args.classes, args.head, args.loss, args.dropout)).to(dev)
        prev.eval()
        for q in prev.parameters():
            q.requires_grad_(False)
        names = ['backbone.layer1', 'backbone.layer2', 'backbone.layer3', 'backbone.layer4', 'cls_head.avg_pool']
        cur_hooks, prev_hooks = bd.OutputHook(model, names), bd.OutputHook(prev, names)
        front = bd.BackgroundMixFrontEnd(alpha=0.5)
        batch = dict(frames=frames, bg=bg, mix=mix, label=labels)

        def loss_fn(m, b):
            data = dict(imgs=front(b['frames'], b['bg'], b['mix']), label=b['label'])
            return bd.base_training_step(m, data, current_task=1, prev_model=prev, current_hooks=cur_hooks,
                                         prev_hooks=prev_hooks, kd_modules_names=names, kd_weight_by_module=[0.01] * 5,
                                         adaptive_scale_factors=[1.0, 3.3466401061363023])
    elif i3d:
        imgs = torch.randn(args.batch, 1, 3, 32, 224, 224, generator=g).to(dev)

        def loss_fn(m, b):
            out = m(b['imgs'], b['label'])
            out['loss'] = out['loss_cls']
            return out
        batch = dict(imgs=imgs, label=labels)
    else:
        imgs = torch.randn(args.batch, 8, 3, 224, 224, generator=g).to(dev)
        batch = dict(imgs=imgs, label=labels)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    from bdvcil_amd import functional as Fn
    side_default = Fn._SIDE['enabled']
    # The per-kernel table (`all_conv_kernels`) comes from ONE untimed step -- the last warm-up step, or an extra step when
    # --warmup 0 -- with HIP events around every conv call; on that step the weight gradients stay on the main stream (an event
    # pair must not span work of the other stream).  It names the dominant kernel; the timed region then carries event pairs
    # around that kernel's launches only (a few dozen instead of ~700 per sampled step), on one timed step in 16.
    probe = not args.no_kernel_timing
    for w in range(args.warmup - (1 if probe else 0)):
        engine.step(batch, loss_fn)
    table, dom = {}, None
    if probe:
        timer.enabled = True
        Fn.set_side_stream_enabled(False)
        engine.step(batch, loss_fn)
        Fn.set_side_stream_enabled(side_default)
        timer.enabled = False
        torch.cuda.synchronize()
        table = timer.summary()
        timer.records = []
        conv_rows = {k: v for k, v in table.items() if v['flops'] > 0}
        if conv_rows:
            dom = max(conv_rows, key=lambda k: conv_rows[k]['ms'])
            timer.only = dom
    sync()
    # one event per step boundary on the compute stream (a record costs about a microsecond): per-step times for the median the
    # survey's measurement definition asks for; `value` stays total clips / wall time of the K steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    sampled = 0
    for i in range(args.steps):
        # one timed step in 16 (the 9th, 25th, ...; the middle one of a short run) carries the event pairs, with the side streams
        # off for that step: a pair then brackets the kernel alone, not the kernel stretched by what shares the GPU with it
        sample = dom is not None and (i % 16 == 8 or (args.steps <= 8 and i == args.steps // 2))
        timer.enabled = sample
        Fn.set_side_stream_enabled(side_default and not sample)
        sampled += int(sample)
        out = engine.step(batch, loss_fn)
        marks[i + 1].record()
    Fn.set_side_stream_enabled(side_default)
    sync()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    timer.enabled = False
    loss_val = float(out['loss_cls'].item())

    # The same step once more in the 'bf16x2' arithmetic (16 significand bits per operand, three MFMA products instead of six):
    # a second record beside the headline, which stays the fp32-level arithmetic.  Same model configuration, batch, warm-up and
    # step count, same barrier + synchronize bracket, max over ranks.
    alt = None
    if args.arith == 'bf16x3' and not (cil or predict or i3d or args.no_alt_arith):
        K.set_conv_arith('bf16x2')
        torch.manual_seed(0)
        model2 = bd.build_model(cfg).to(dev)
        model2.train()
        reducer2 = None
        if use_dist:
            bd.broadcast_parameters(model2)
            reducer2 = bd.GradAllReducer(model2, bucket_cap_mb=float(os.environ.get('BDVCIL_BUCKET_MB', '25')))
        engine2 = bd.TrainEngine(model2, bd.build_optimizer(model2, dict(type='SGD', constructor='CILTSMOptimizerConstructorImprovised',
                                                                          paramwise_cfg=dict(fc_lr_scale_factor=5.0), lr=0.01, momentum=0.9,
                                                                          weight_decay=1e-4)), grad_clip=None, reducer=reducer2)
        for _ in range(args.warmup):
            engine2.step(batch, None)
        sync()
        marks2 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t1 = time.perf_counter()
        marks2[0].record()
        for i in range(args.steps):
            out2 = engine2.step(batch, None)
            marks2[i + 1].record()
        sync()
        dt2 = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([dt2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt2 = float(t.item())
        ms2 = sorted(marks2[i].elapsed_time(marks2[i + 1]) for i in range(args.steps))
        alt = {'conv_arith': 'bf16x2', 'value': round(args.batch * world * args.steps / dt2, 2), 'unit': 'clips/s',
               'ms_per_step': round(1000.0 * dt2 / args.steps, 3), 'ms_per_step_median': round(ms2[len(ms2) // 2], 3),
               'final_loss': round(float(out2['loss_cls'].item()), 5),
               'note': 'NOT the headline: the same workload, steps and warm-up with every conv operand cut to its two leading bf16 pieces '
                       '(16 significand bits) and three v_mfma_f32_32x32x16_bf16 products (hi*hi + hi*mid + mid*hi), fp32 accumulate, fp32 '
                       'tensors; dropped terms <= 2^-15 relative per product (fp32-level: 2^-24; TF32: 2^-11); plane kernels only, the stem '
                       'keeps three pieces; parity bars of this arithmetic: tests/test_bf16x2_gpu.py'}
        K.set_conv_arith(args.arith)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / dt
        flop_per_clip = (I3D_FLOP_PER_CLIP if i3d else R50_KD_FLOP_PER_CLIP if cil else R50_FWD_FLOP_PER_CLIP if predict else R50_FLOP_PER_CLIP) if args.depth == 50 else None
        res = {
            'metric': (f'clips/sec fwd+bwd TSM-R50 8x224^2 bs{args.batch}/GPU, bf16 MFMA tiles (BASELINE config 5; not the headline metric)'
                       if args.arith in ('bf16x1', 'bf16') and args.depth == 50 and not (cil or predict or i3d) else
                       f'clips/sec fwd+bwd TSM-R50 8x224^2 bs{args.batch}/GPU, two-piece bf16 conv products (not the headline metric)'
                       if args.arith == 'bf16x2' and args.depth == 50 and not (cil or predict or i3d) else
                       'clips/sec fwd+bwd I3D-R50 32x224^2 bs16/GPU (BASELINE config 4; not the headline metric)' if i3d else
                       'clips/sec fwd+bwd TSM-R50 8x224^2 bs32/GPU' if args.depth == 50 else f'clips/sec fwd+bwd TSM-R{args.depth}')
                      + (' (CIL step: bg-mix front-end + KD teacher + LSCLoss)' if cil else '')
                      + (' (predict_step: eval forward + representations, no backward)' if predict else ''),
            'value': round(value, 2), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1000.0 * dt / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if args.arith in ('bf16x1', 'bf16') else 'bf16x2' if args.arith == 'bf16x2' else 'f32', 'data': 'synthetic',
            'config': {'workload': ('CIL task-1 step (uint8 bg-mix front-end, frozen teacher forward, 5 feature-KD MSE terms, clip 1.0): ' if cil else '')
                                   + (f'I3D-ResNet50 (ResNet3d, 3x1x1 inflation) fwd+bwd+SGD step, synthetic {args.batch}x3x32x224x224 clips per GPU, ' if i3d else
                                      f'TSM-ResNet{args.depth} ' + ('eval forward + clip representations (predict_step)' if predict else 'fwd+bwd+SGD step') + f', synthetic {args.batch}x8x3x224x224 clips per GPU, ')
                                   + f'{args.classes} classes, {"I3DHead" if i3d else args.head}+{args.loss}, dropout {args.dropout}, random-init weights; '
                                   + ('fp32 tensors, accumulators and results; conv products: fp32 via 3xbf16 split, 6 MFMA products, fp32 accumulate '
                                      '(dropped terms <= 2^-24 relative)' if args.arith == 'bf16x3' else
                                      'fp32 tensors, accumulators and results; conv products from the TWO leading bf16 pieces of each operand (16 significand '
                                      'bits), 3 MFMA products, fp32 accumulate (dropped terms <= 2^-15 relative; plane kernels only, the stem keeps three pieces)'
                                      if args.arith == 'bf16x2' else
                                      'REDUCED PRECISION (BASELINE config 5): conv operands rounded to bf16, one bf16 MFMA product, fp32 accumulate, '
                                      'fp32 tensors' if args.arith == 'bf16x1' else
                                      'REDUCED PRECISION (BASELINE config 5): activations and their gradients STORED as bf16 between the stem max-pool and the '
                                      'average pool, one bf16 MFMA product per step, fp32 accumulate; fp32 BatchNorm statistics, weights, weight gradients, '
                                      'head and loss' if args.arith == 'bf16' else
                                      'fp32 tensors and results; conv products on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain)'),
                       'conv_arith': args.arith,
                       'clips_per_gpu': args.batch, 'global_batch': args.batch * world,
                       'final_loss': round(loss_val, 5),
                       # device time between step boundaries (HIP events): the median is insensitive to the one step in 16 that
                       # carries the per-kernel event pairs and runs with one stream
                       'ms_per_step_median': round(step_ms[len(step_ms) // 2], 3), 'ms_per_step_min': round(step_ms[0], 3),
                       'ms_per_step_max': round(step_ms[-1], 3),
                       'clips_per_s_at_median': round(args.batch * world * 1000.0 / step_ms[len(step_ms) // 2], 2)},
        }
        res['config'].update(dist_record(reducer, world))
        if alt is not None:
            res['alt_arith'] = alt
        if reducer is not None:
            res['n_gpus'] = reducer.describe()['ranks'] if world > 1 else world      # the communicator's count, not the launcher's
        ms = torch.cuda.memory_stats(dev)
        res['config']['hbm'] = {'peak_allocated_gb': round(ms.get('allocated_bytes.all.peak', 0) / 2**30, 2),
                                'peak_reserved_gb': round(ms.get('reserved_bytes.all.peak', 0) / 2**30, 2),
                                'device_mallocs': ms.get('num_device_alloc', 0), 'device_frees': ms.get('num_device_free', 0),
                                'alloc_retries': ms.get('num_alloc_retries', 0)}
        if flop_per_clip:
            res['config']['algorithmic_flop_per_clip'] = flop_per_clip
            res['config']['step_tflops'] = round(value / world * flop_per_clip / 1e12, 2)
            res['config']['step_frac_of_f32_mfma_peak'] = round(value / world * flop_per_clip / PEAK_F32_MFMA, 4)
        if dom is not None and timer.records:
            by = table
            timed_steps = 1
            d = timer.summary()[dom]                      # the dominant kernel over the timed region
            bf16_pieces = '_x3_' in dom or '_pl_' in dom
            two = args.arith == 'bf16x2' and dom.replace(' ', '').endswith(',2>')
            peak = (PEAK_BF16_MFMA if args.arith in ('bf16x1', 'bf16') and bf16_pieces else PEAK_BF16_MFMA / 3 if two else
                    PEAK_BF16X3 if bf16_pieces else PEAK_F32_MFMA)
            achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
            tot_ms = sum(v['ms'] for v in by.values())
            tot_fl = sum(v['flops'] for v in by.values())
            res['roofline'] = {
                'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak / 1e12, 1), 'unit': 'TFLOP/s',
                'frac': round(achieved * 1e12 / peak, 4), 'traffic': pmc_traffic(dom),
                'peak_note': ('FLOP (2*M*N*K) against the bf16 dense MFMA peak / 3: every product costs three v_mfma_f32_32x32x16_bf16 products' if two else
                              'fp32-equivalent FLOP (2*M*N*K) against the bf16 dense MFMA peak / 6: every fp32 product costs six '
                              'v_mfma_f32_32x32x16_bf16 products' if bf16_pieces else 'v_mfma_f32_32x32x2_f32 dense peak'),
                'frac_of_f32_mfma_peak': round(achieved * 1e12 / PEAK_F32_MFMA, 4),
                'traffic_note': f'HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of '
                                f'this command, profiles/{PROFILE_ROUND}_traffic.json); avg_launch_ms spans the whole C-ABI call '
                                '(main kernel + its K-split fix-up when the planner uses one)',
                'kernel': dom, 'launches_per_step': d['launches'] // max(sampled, 1),
                'avg_launch_ms': round(d['ms'] / d['launches'], 4),
                'rocprof_avg_us': rocprof_avg_us(dom),
                'algorithmic_gflop_per_launch': round(d['flops'] / d['launches'] / 1e9, 2),
                'algorithmic_bytes_per_launch': int(d['bytes'] / d['launches']),      # operands + results once (+ fused epilogue tensors): compare with `traffic`
                'all_conv_kernels': {k: {'launches_per_step': v['launches'] // timed_steps, 'ms_per_step': round(v['ms'] / timed_steps, 3),
                                         'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 2) if v['flops'] else None,
                                         'rocprof_avg_us': rocprof_avg_us(k)} for k, v in sorted(by.items())},
                'conv_ms_per_step': round(tot_ms / timed_steps, 3), 'conv_tflops': round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                'kernel_timed_steps': sampled,
                'kernel_timing_note': 'achieved / avg_launch_ms: HIP events around every launch of the dominant kernel on one timed step in 16, with the side streams off on that step (a pair brackets the kernel alone); all_conv_kernels / conv_ms_per_step: HIP events around every conv call on ONE untimed step (the last warm-up step), side streams off',
            }
        if world == 1 and not args.no_cpu_baseline and not cil and not predict and not i3d:
            res['cpu_baseline'] = cpu_baseline(args.depth, args.classes, args.head, args.loss)
        emit(res)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
