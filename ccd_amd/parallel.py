"""Data parallelism for the arena-backed student: one process per GPU, gradients all-reduced (mean) over RCCL/xGMI
in contiguous arena buckets that are launched WHILE the backward pass is still running (replaces
DistributedDataParallel(student, find_unused_parameters=True), train.py:106, collective C2 of SURVEY.md 2.3).

The HIP backward writes gradients straight into the arena, so "gradient ready" is signalled by the engine
(module.grad_ready_hook(prefix)) instead of autograd hooks: the DINO head finishes first (its 16.8 M-element last
layer is the largest bucket), then the transformer blocks in reverse order, then patch-embed; everything else
(segmentation head, norm layers touched last) goes in finish().  Works on any backend: 'nccl' (= RCCL) on GPUs,
'gloo' in the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn


class GradReducer:
    def __init__(self, arena, process_group=None, bucket_elems=8 * 1024 * 1024, at_world1=False, reserve_window=0):
        """at_world1: issue the collectives even on a single rank (they are identities there) - the 1-GPU smoke of the
        N > 1 path (bench.py BENCH_FORCE_DIST=1, tests/test_model_gpu.py)."""
        self.arena, self.pg, self.bucket_elems = arena, process_group, bucket_elems
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = dist.is_initialized() and (self.world > 1 or at_world1)
        self._pending, self._works, self._done = [], [], []
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.reserve_window = reserve_window   # kernel launches behind a bucket launch that leave `cu_reserve` CUs to the collective

    def _launch(self, lo, hi):
        if not self.active or hi <= lo:
            return
        buf = self.arena.grad[lo:hi]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self._works.append((dist.all_reduce(buf, op=op, group=self.pg, async_op=True), lo, hi))
        if self.reserve_window > 0 and buf.device.type == "cuda":
            # the kernels launched next run beside this collective: they leave `cu_reserve` CUs free - `reserve_window` launches
            # per bucket-size worth of gradients (the 16.8 M-element head range is one 67-MB all-reduce: three windows)
            from . import ops
            scale = max(1, -(-(hi - lo) // self.bucket_elems))
            ops.policy_set("cu_reserve_left", self.reserve_window * scale)

    def mark_ready(self, prefix):
        """All gradients of parameters named `prefix`* are final: reduce the range (merged into buckets)."""
        lo, hi = self.arena.range_of(prefix)
        self._done.append((lo, hi))
        self._pending.append((lo, hi))
        size = sum(h - l for l, h in self._pending)
        if size >= self.bucket_elems:
            self._flush()

    def _flush(self):
        # merge adjacent ranges so one collective covers as much contiguous memory as possible
        for lo, hi in _merge(self._pending):
            self._launch(lo, hi)
        self._pending = []

    def finish(self):
        """Reduce whatever has not been announced yet, then wait for every collective."""
        self._flush()
        covered = _merge(self._done)
        pos = 0
        for lo, hi in covered + [(self.arena.total, self.arena.total)]:
            if lo > pos:
                self._launch(pos, lo)
            pos = max(pos, hi)
        for work, lo, hi in self._works:
            work.wait()
            if not self._avg and self.world > 1:
                self.arena.grad[lo:hi].div_(self.world)
        self._works, self._done = [], []
        if self.reserve_window > 0 and self.arena.device.type == "cuda":
            # every collective has been waited for: what is left of the last bucket's window must not reach into the optimizer and the
            # next iteration's forward pass (measured on the N > 1 path of one rank: its first block halves ran 20 % longer)
            from . import ops
            ops.policy_set("cu_reserve_left", 0)


def _merge(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1] = (out[-1][0], max(out[-1][1], hi))
        else:
            out.append((lo, hi))
    return out


class DataParallel(nn.Module):
    """Drop-in for DistributedDataParallel around an arena-backed ccd_amd model: `.module`, `module.`-prefixed
    state-dict keys (the finetune script expects them, train_finetune.py:193-200), rank-0 parameter broadcast at
    construction, overlapped gradient averaging."""

    def __init__(self, module, device_ids=None, find_unused_parameters=False, process_group=None,
                 bucket_elems=8 * 1024 * 1024, reduce_at_world1=False, cu_reserve=8, reserve_window=3):
        """cu_reserve: compute units the persistent GEMM grids leave free so that the RCCL kernels of a bucket's all-reduce find
        a slot next to them (policy key `cu_reserve`; 0 = take every CU) - for the `reserve_window` kernel launches that follow
        the bucket's launch (3 launches ~ 0.75 ms of the backward pass; a 32-MB ring all-reduce over 8 GPUs moves 56 MB per GPU over
        xGMI, ~0.3 ms), not for the whole step: 248 instead of 256 workgroups cost the row-owner kernels a whole extra round of tiles
        (measured on the N > 1 path of one rank: fused MLP + 20 %, LayerNorm-backward product + 18 %), which is why a row-owner launch
        that the reserve would push into an extra round takes every CU anyway (csrc/abi_impl.h: ccd_grid_cus) and the window is reset
        once every collective has been waited for.  reserve_window = -1: every launch."""
        super().__init__()
        self.module = module
        self.reducer = None
        if dist.is_available() and dist.is_initialized():
            arena = module.ensure_arena()
            dist.broadcast(arena.flat, src=0, group=process_group)
            for b in module.buffers():
                if b.is_floating_point() or b.dtype in (torch.int64, torch.int32):
                    dist.broadcast(b, src=0, group=process_group)
            arena.refresh_mirrors()
            world = dist.get_world_size(process_group)
            if any(p.requires_grad for p in module.parameters()) and (world > 1 or reduce_at_world1):
                # gradient buckets travel on their OWN process group (= their own RCCL stream): the blocking SyncBatchNorm
                # exchanges of the segmentation head's backward pass must not queue behind a 22 M-element bucket
                ranks = list(range(dist.get_world_size())) if process_group is None else dist.get_process_group_ranks(process_group)
                grad_pg = dist.new_group(ranks=ranks, backend=dist.get_backend(process_group))
                on_gpu = arena.device.type == "cuda" and cu_reserve
                self.reducer = GradReducer(arena, grad_pg, bucket_elems, at_world1=reduce_at_world1,
                                           reserve_window=int(reserve_window) if on_gpu else 0)
                if on_gpu:
                    from . import ops
                    ops.policy_set("cu_reserve", int(cu_reserve))
                    ops.policy_set("cu_reserve_window", int(reserve_window))
                    ops.policy_set("cu_reserve_left", 0)
                hook = self.reducer.mark_ready
                for m in module.modules():
                    if hasattr(m, "grad_ready_hook"):
                        m.grad_ready_hook = hook

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def finish_gradient_sync(self):
        if self.reducer is not None:
            self.reducer.finish()
