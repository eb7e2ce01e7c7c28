"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` backend ``nccl`` (= RCCL over
xGMI on ROCm; ``gloo`` in the CPU tests).

Replaces the implicit PL ``ddp_spawn`` DistributedDataParallel reducer (libs/cil/cil.py:704-709,752;
SURVEY section 2.1/2.2): parameters are grouped into flat fp32 buckets in reverse registration order (the order
backward produces gradients); a bucket is packed and all-reduced (SUM) on a side stream as soon as its last
gradient has been accumulated, overlapping the remaining backward.  The mean is applied inside the fused SGD
step (``FusedSGD.set_grad_scale(1/world)``), so no extra pass over the gradients.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): RCCL's ring moves 2*(R-1)/R of the payload over one link per
hop, so few, large buckets (default 4 x ~24 MB for the 94.9 MB R50 payload) beat many small ones; the last one is split
so that only <= 2 MB (stem + first block) are reduced after backward has ended.
Deliberate deviation from DDP defaults: BN buffers are not broadcast every iteration (per-GPU statistics, no
SyncBN in the reference).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

# BDVCIL_FORCE_DIST=1: issue the collectives even in a one-rank process group (exercises the RCCL path on one GPU)
_FORCE = os.environ.get('BDVCIL_FORCE_DIST', '0') != '0'


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """One-time parameter/buffer broadcast at construction (what DDP does in its constructor)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _FORCE):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
    from . import kernels as K
    K.bump_weight_epoch()            # written through .data: torch's version counters did not move


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat: Optional[torch.Tensor] = None
        self.views: List[torch.Tensor] = []
        self.pending = 0
        self.work = None
        self.event = None
        self.held = None            # the gradients the pack copy reads on the communication stream, kept until finish() has joined it
        self.no_grad: List[bool] = []
        self.flags: Optional[torch.Tensor] = None

    def ensure(self):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            # payload, then one "a gradient arrived on this rank" flag per parameter: summed with the gradients, so that every rank
            # can tell a parameter NO rank used (its .grad stays None, the optimizer skips it everywhere) from one that only this rank
            # did not use (the other ranks' sum is its gradient here too)
            self.flat = torch.zeros(self.numel + len(self.params), dtype=p0.dtype, device=p0.device)
            self.flags = self.flat[self.numel:]
            self.views, off = [], 0
            for p in self.params:
                # a view with the parameter's own (dense, possibly channels_last) strides
                v = torch.as_strided(self.flat, p.shape, p.stride(), storage_offset=off)
                self.views.append(v)
                off += p.numel()


_COMM_STREAMS = {}


def _comm_stream(device) -> 'torch.cuda.Stream':
    """ONE communication stream per device for every reducer of the process.  HIP maps a process's streams onto GPU_MAX_HW_QUEUES
    hardware queues and streams that share a queue run in order: a second reducer with a stream of its own (a second model in one
    process, e.g. bench.py's alt_arith run) pushed the weight-gradient stream onto a shared queue and cost 12 ms per step
    (profiles/r03_ab.txt item 9)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _COMM_STREAMS.get(idx)
    if st is None:
        st = _COMM_STREAMS[idx] = torch.cuda.Stream(device=device)
    return st


class GradAllReducer:
    def __init__(self, module: torch.nn.Module, bucket_cap_mb: float = 25.0, process_group=None, tail_cap_mb: float = 2.0):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters() if p.requires_grad]
        params.reverse()
        cap = int(bucket_cap_mb * 1024 * 1024 / 4)
        groups: List[List[torch.nn.Parameter]] = []
        cur, n = [], 0
        for p in params:
            cur.append(p)
            n += p.numel()
            if n >= cap:
                groups.append(cur)
                cur, n = [], 0
        if cur:
            groups.append(cur)
        # The last bucket becomes ready only when backward has finished (it holds the stem), so its all-reduce is the one
        # collective nothing can hide: keep it small by splitting the earliest layers (<= tail_cap_mb) off into their own
        # bucket; the rest of that bucket then overlaps with the backward of those layers.
        tail_cap = int(tail_cap_mb * 1024 * 1024 / 4)
        if groups and tail_cap > 0 and sum(p.numel() for p in groups[-1]) > tail_cap and len(groups[-1]) > 1:
            last, tail, n = groups[-1], [], 0
            while len(last) > 1 and n + last[-1].numel() <= tail_cap:
                n += last[-1].numel()
                tail.insert(0, last.pop())
            if tail:
                groups.append(tail)
        self.buckets: List[_Bucket] = [_Bucket(g) for g in groups]
        self._index = {}
        self._handles = []
        for bi, b in enumerate(self.buckets):
            for p in b.params:
                self._index[p] = bi
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._comm_stream = None
        self.timing = False         # bench.py: HIP events around the waits of finish() -> exposed_ms()
        self._exposed = []
        self.reset()

    def describe(self) -> dict:
        """What the exchange looks like from this rank: filled from the communicator, not from the launcher's environment."""
        on = dist.is_available() and dist.is_initialized()
        return dict(ranks=dist.get_world_size(self.group) if on else 1, backend=dist.get_backend(self.group) if on else None,
                    buckets=len(self.buckets), payload_mb=round(sum(b.numel for b in self.buckets) * 4 / 2 ** 20, 2),
                    bucket_mb=[round(b.numel * 4 / 2 ** 20, 2) for b in self.buckets])

    def exposed_ms(self):
        """Mean time per step the compute stream spent blocked in finish() waiting for the collectives (None: not timed / CPU)."""
        if not self._exposed:
            return None
        if isinstance(self._exposed[0], float):             # CPU tensors (gloo tests): host wall clock around the waits
            return sum(self._exposed) / len(self._exposed)
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def reset(self):
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None

    def remove(self):
        for h in self._handles:
            h.remove()

    # autograd calls this right after p.grad has been written for this backward pass
    def _on_grad(self, p: torch.nn.Parameter):
        b = self.buckets[self._index[p]]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b: _Bucket):
        b.ensure()
        on_gpu = b.flat.is_cuda
        if on_gpu:
            from .functional import join_side_stream
            join_side_stream(b.flat.device)         # weight gradients are produced on the wgrad side stream
            if self._comm_stream is None:
                self._comm_stream = _comm_stream(b.flat.device)
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self._comm_stream.wait_event(ready)
            ctx = torch.cuda.stream(self._comm_stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx, torch.no_grad():
            # a parameter without a gradient this step contributes zeros to the sum; it is remembered so that finish()
            # leaves its .grad at None (no rank used it: the optimizer must skip it, as torch's DDP + SGD do)
            b.no_grad = [p.grad is None for p in b.params]
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b.params]
            torch._foreach_copy_(b.views, grads)                      # pack (copy plumbing, no arithmetic)
            if getattr(b, '_flag_pattern', None) != b.no_grad:        # (device copy of the pattern: rebuilt only when it changes)
                b._flag_pattern = list(b.no_grad)
                b._flag_src = torch.tensor([0.0 if s else 1.0 for s in b.no_grad], dtype=b.flat.dtype).to(b.flat.device)
            b.flags.copy_(b._flag_src)
            # The gradients were allocated on the compute stream and are read here on the communication stream.  They stay
            # referenced (p.grad, and this list) until finish() has made the compute stream wait for this bucket; record_stream()
            # instead would keep every one of these blocks out of the allocator's pool for a step (functional.SIDE_LAG says what
            # that costs: 115 GB reserved for 24 GB of tensors).
            b.held = grads
            if self.world > 1 or (_FORCE and dist.is_initialized()):
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if on_gpu:
                b.event = torch.cuda.Event()
                b.event.record(self._comm_stream)

    def finish(self):
        """Wait for all buckets; afterwards ``p.grad`` aliases the reduced (summed) bucket slices.  A parameter that received no
        gradient on ANY rank keeps ``.grad = None``; one that only this rank did not use gets the other ranks' sum, as under torch's
        DistributedDataParallel (the flags travel in the bucket, so every rank decides alike)."""
        for b in self.buckets:
            if b.pending != 0:
                # parameters that received no gradient this step (unused / frozen late): reduce what we have
                self._launch(b)
        timed = bool(self.timing and self.buckets and self.buckets[0].flat is not None)
        on_gpu = timed and self.buckets[0].flat.is_cuda
        if on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        elif timed:
            import time
            t0 = time.perf_counter()
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
            if b.event is not None:
                torch.cuda.current_stream().wait_event(b.event)
        if on_gpu:
            e1.record()
            self._exposed.append((e0, e1))
        elif timed:
            self._exposed.append((time.perf_counter() - t0) * 1e3)
        for b in self.buckets:
            b.held = None       # the compute stream is ordered behind the pack copies now
            used_elsewhere = None
            if any(b.no_grad) and self.world > 1:
                used_elsewhere = (b.flags > 0).tolist()     # rare path (a parameter unused on this rank): one small read-back
            for i, (p, v, skipped) in enumerate(zip(b.params, b.views, b.no_grad or [False] * len(b.params))):
                p.grad = None if (skipped and not (used_elsewhere and used_elsewhere[i])) else v
        self.reset()
