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

    def ensure(self):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(self.numel, dtype=p0.dtype, device=p0.device)
            self.views, off = [], 0
            for p in self.params:
                # a view with the parameter's own (dense, possibly channels_last) strides
                v = torch.as_strided(self.flat, p.shape, p.stride(), storage_offset=off)
                self.views.append(v)
                off += p.numel()


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
        self.reset()

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
                self._comm_stream = torch.cuda.Stream(device=b.flat.device)
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
            if on_gpu:
                for g in grads:
                    g.record_stream(self._comm_stream)
            if self.world > 1 or (_FORCE and dist.is_initialized()):
                b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if on_gpu:
                b.event = torch.cuda.Event()
                b.event.record(self._comm_stream)

    def finish(self):
        """Wait for all buckets; afterwards ``p.grad`` aliases the reduced (summed) bucket slices."""
        for b in self.buckets:
            if b.pending != 0:
                # parameters that received no gradient this step (unused / frozen late): reduce what we have
                self._launch(b)
            if b.work is not None:
                b.work.wait()
            if b.event is not None:
                torch.cuda.current_stream().wait_event(b.event)
            for p, v, skipped in zip(b.params, b.views, getattr(b, 'no_grad', None) or [False] * len(b.params)):
                p.grad = None if skipped else v
        self.reset()
