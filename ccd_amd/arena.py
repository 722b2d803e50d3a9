"""Flat parameter arena: every parameter of a network lives in ONE fp32 buffer (master weights), with a parallel
fp32 gradient buffer, a bf16 mirror of the whole buffer (what the MFMA GEMMs read) and bf16 TRANSPOSED copies of the
2-D weights whose backward needs them (dX = dY . W runs as an NT GEMM against W^T).

nn.Parameter objects keep their reference names (state-dict compatible, Dino/*: SURVEY.md 8b) but their .data /
.grad are views into the arena, so that
  * the fused clip+AdamW / EMA kernels and the gradient all-reduce work on contiguous ranges,
  * HIP kernels write gradients straight into their slots (no autograd accumulation pass).
Host-side bookkeeping only; all data movement is done by kernels from ccd_amd.ops.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import torch

from . import ops

ALIGN = 64          # elements; keeps every tensor 256-B aligned in fp32 and 128-B aligned in bf16
CHUNK = 1024        # optimizer chunk (elements)


class Segment:
    __slots__ = ("name", "offset", "numel", "shape", "t_offset", "index")

    def __init__(self, name, offset, numel, shape, index):
        self.name, self.offset, self.numel, self.shape, self.index = name, offset, numel, tuple(shape), index
        self.t_offset = None


class ParamArena:
    def __init__(self, named_params, device, with_grad=True, transposed=()):
        """named_params: iterable of (name, nn.Parameter) in model order.  `transposed`: names needing a W^T mirror."""
        self.device = torch.device(device)
        self.segments: "OrderedDict[str, Segment]" = OrderedDict()
        self.params = OrderedDict()
        off = 0
        for i, (name, p) in enumerate(named_params):
            self.segments[name] = Segment(name, off, p.numel(), p.shape, i)
            self.params[name] = p
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device) if with_grad else None
        self.mirror = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        t_off = 0
        for name in transposed:
            seg = self.segments[name]
            assert len(seg.shape) == 2, name
            seg.t_offset = t_off
            t_off += (seg.numel + ALIGN - 1) // ALIGN * ALIGN
        self.mirror_t = torch.zeros(max(t_off, 1), dtype=torch.bfloat16, device=self.device)
        # adopt the parameters: copy current values in, then re-point .data (and .grad) at the arena
        with torch.no_grad():
            for name, p in self.params.items():
                seg = self.segments[name]
                view = self.flat[seg.offset:seg.offset + seg.numel].view(seg.shape)
                view.copy_(p.data.to(self.device, torch.float32))
                p.data = view
                if with_grad and p.requires_grad:
                    p.grad = self.grad[seg.offset:seg.offset + seg.numel].view(seg.shape)
        self._mirror_descs = None
        self._build_transpose_descs()
        self._opt_tables = None
        self.skip_substrings = set()       # tensors whose update is cancelled for the coming optimizer step
        self.stale = False
        self.refresh_mirrors()

    # ------------------------------------------------------------------------------------------ views
    def w(self, name):
        s = self.segments[name]
        return self.flat[s.offset:s.offset + s.numel].view(s.shape)

    def g(self, name):
        s = self.segments[name]
        return self.grad[s.offset:s.offset + s.numel].view(s.shape)

    def wb(self, name):
        s = self.segments[name]
        return self.mirror[s.offset:s.offset + s.numel].view(s.shape)

    def wbt(self, name):
        s = self.segments[name]
        return self.mirror_t[s.t_offset:s.t_offset + s.numel].view(s.shape[1], s.shape[0])

    def span(self, first, rows, which="wb"):
        """[rows, cols] view over CONSECUTIVE 2-D tensors of equal width starting at `first` (e.g. linear_q/k/v stacked
        into one [3*512, 512] operand).  which: 'w' fp32 weights, 'g' fp32 gradients, 'wb' bf16 mirror."""
        names = list(self.segments)
        i = names.index(first)
        cols = self.segments[first].shape[1]
        off, have = self.segments[first].offset, 0
        while have < rows:
            s = self.segments[names[i]]
            assert len(s.shape) == 2 and s.shape[1] == cols and s.offset == off + have * cols, (first, names[i])
            have += s.shape[0]
            i += 1
        assert have == rows, (first, rows, have)
        buf = {"w": self.flat, "g": self.grad, "wb": self.mirror}[which]
        return buf[off:off + rows * cols].view(rows, cols)

    def range_of(self, prefix):
        """[lo, hi) element range covered by the parameters whose name starts with `prefix` (contiguous by construction)."""
        segs = [s for n, s in self.segments.items() if n.startswith(prefix)]
        lo = min(s.offset for s in segs)
        hi = max(s.offset + (s.numel + ALIGN - 1) // ALIGN * ALIGN for s in segs)
        return lo, hi

    # ---------------------------------------------------------------------------------------- mirrors
    def _build_transpose_descs(self):
        blob, tiles, n = b"", 0, 0
        for s in self.segments.values():
            if s.t_offset is None:
                continue
            rows, cols = s.shape
            src = self.flat.data_ptr() + 4 * s.offset
            dst_t = self.mirror_t.data_ptr() + 2 * s.t_offset
            blob += struct.pack("PPPiiii", src, 0, dst_t, rows, cols, tiles, 0)
            tiles += ((rows + 63) // 64) * ((cols + 63) // 64)            # (64 x 64 tiles: include/ccd_hip.h, ccd_mirror_desc)
            n += 1
        if n:
            host = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
            self._mirror_descs = (host.to(self.device), n, tiles)

    def refresh_transposes(self):
        if self._mirror_descs is not None:
            ops.mirror_bf16(*self._mirror_descs)

    def refresh_mirrors(self):
        """Full refresh (after load_state_dict / external edits). AdamW and EMA keep `mirror` current themselves."""
        ops.cast_bf16(self.flat, self.mirror)
        self.refresh_transposes()
        self.stale = False

    def zero_grad(self):
        if self.grad is not None:
            self.grad.zero_()
            # a backward pass that starts from zeroed gradients may read a slot it has just completed as "this pass's sum"
            # (engine.backbone_backward: colsum(gb) = proj.bias's gradient); cleared by the first backward pass that runs
            self.grad_fresh = True

    # -------------------------------------------------------------------------------- optimizer tables
    def opt_tables(self):
        """(chunk_seg int32, chunk_begin int64, chunk_len int32) on the device; one entry per <= 1024-element chunk."""
        if self._opt_tables is None:
            cs, cb, cl = [], [], []
            for s in self.segments.values():
                for c in range(0, s.numel, CHUNK):
                    cs.append(s.index)
                    cb.append(s.offset + c)
                    cl.append(min(CHUNK, s.numel - c))
            dev = self.device
            self._opt_tables = (torch.tensor(cs, dtype=torch.int32, device=dev),
                                torch.tensor(cb, dtype=torch.int64, device=dev),
                                torch.tensor(cl, dtype=torch.int32, device=dev))
        return self._opt_tables
