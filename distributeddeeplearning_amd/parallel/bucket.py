"""Static gradient-bucket planning and flat gradient arenas.

This replaces Horovod's tensor-fusion buffer and per-cycle negotiation (SURVEY §2.2 N1/N2,
§2.5 X1/X8).  Horovod 0.13 re-negotiates the set of ready tensors every 5 ms and memcpy's them
into a 64 MB fusion buffer before each ``ncclAllReduce`` [EXT].  Data-parallel training
produces the same gradients in the same order every step, so here the plan is computed once:

* parameters are laid out in *reverse registration order* (≈ the order backward produces
  their gradients: fc first, stem last — SURVEY §2.7 "Gradient bytes by stage");
* each dtype gets one contiguous **arena**; a bucket is a contiguous slice of the arena, so
  the all-reduce runs in place on the slice — no pack, no unpack;
* ``param.grad`` is a *view* into the arena (zero-copy; wgrad kernels write straight into the
  communication buffer);
* the first bucket is kept small so communication starts early in backward and the last
  bucket (the stem, tiny) leaves only a latency-bound tail exposed.

Sizing for MI355X/xGMI: RCCL rings are bound by one ≈153 GB/s link, so a 32 MiB
bucket costs ≈0.4 ms on 8 GPUs — large enough to amortise launch latency, small enough that
4–6 buckets pipeline behind the remaining backward.  Segment offsets are aligned to 256 B and
bucket ends to 8 shards × 256 B so that reduce-scatter style kernels can split a bucket evenly
across 8 peers with 16-byte vector accesses.

The planner itself is pure Python over ``(name, numel, dtype)`` records and has a native twin
(``ddl_bucket_plan`` in ``csrc/engine/engine.cpp``); ``tests/test_native_engine.py`` checks that
both produce the same plan for ResNet-50.

``HVD_COMM_DTYPE=fp32`` (``grad_dtype``) keeps the gradient arenas in fp32 whatever the parameter
dtype: the split-K weight-gradient reduce writes fp32 directly, the collective reduces fp32 on the
wire and multi-pass accumulation (``backward_passes_per_step > 1``) happens in fp32.  Offsets are
shared between the parameter, gradient, momentum and master arenas of one parameter dtype.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, Sequence

import torch

__all__ = ["Segment", "Bucket", "ArenaPlan", "plan_arenas", "GradArena", "SEG_ALIGN_BYTES",
           "BUCKET_ALIGN_BYTES", "arena_options"]

SEG_ALIGN_BYTES = 256
BUCKET_ALIGN_BYTES = 8 * 256


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Segment:
    name: str
    offset: int  # elements from arena start
    numel: int
    shape: tuple
    index: int  # position in the caller's parameter list


@dataclass
class Bucket:
    index: int
    start: int  # elements from arena start
    end: int  # exclusive, padded
    segments: list = field(default_factory=list)

    @property
    def numel(self) -> int:
        return self.end - self.start


@dataclass
class ArenaPlan:
    dtype: torch.dtype
    numel: int
    segments: list
    buckets: list


def plan_arenas(entries: Sequence[tuple], first_bucket_bytes: int, bucket_cap_bytes: int,
                reverse: bool = True, tail_bucket_bytes: int = 0, num_buckets: int = 0,
                wire_esize: dict | None = None) -> dict:
    """Plan one arena per dtype.

    ``entries`` is a sequence of ``(name, shape, dtype)`` in registration (forward) order.
    Returns ``{dtype: ArenaPlan}``.  A tensor is never split across buckets; a tensor larger
    than the cap gets a bucket of its own.

    ``tail_bucket_bytes`` > 0 closes the bucket before the last ``tail_bucket_bytes`` of the
    arena: those tensors (the first layers) only become ready at the very end of backward, and
    whatever shares their bucket cannot be exchanged before — with a small tail bucket the large
    one in front of it overlaps with the stem's backward instead of waiting for it.

    ``num_buckets`` > 0 (``HVD_NUM_BUCKETS``) replaces the byte caps: every arena is cut into that
    many buckets of about equal size.  ``wire_esize`` maps a dtype to the element size the byte
    caps refer to when the gradients travel in another dtype than the parameters.
    """
    by_dtype: dict = {}
    order = list(enumerate(entries))
    if reverse:
        order.reverse()
    for idx, (name, shape, dtype) in order:
        by_dtype.setdefault(dtype, []).append((idx, name, tuple(shape)))
    plans = {}
    for dtype, items in by_dtype.items():
        esize = torch.empty((), dtype=dtype).element_size()
        seg_align = max(SEG_ALIGN_BYTES // esize, 1)
        bkt_align = max(BUCKET_ALIGN_BYTES // esize, 1)
        esize = (wire_esize or {}).get(dtype, esize)   # caps are in bytes on the wire
        segments, buckets = [], []
        cursor = 0
        cur = Bucket(0, 0, 0)
        tail_start = None
        sizes = []
        for _, _, shape in items:
            sz = esize
            for d in shape:
                sz *= int(d)
            sizes.append(sz)
        first_cap, later_cap = first_bucket_bytes, bucket_cap_bytes
        if num_buckets > 0:
            first_cap = later_cap = max((sum(sizes) + num_buckets - 1) // num_buckets, 1)
            tail_bucket_bytes = 0
        if tail_bucket_bytes > 0 and sum(sizes) > 4 * tail_bucket_bytes:
            acc = 0
            for pos in range(len(items) - 1, 0, -1):
                if acc > 0 and acc + sizes[pos] > tail_bucket_bytes:
                    break
                acc += sizes[pos]
                tail_start = pos
        for pos, (idx, name, shape) in enumerate(items):
            n = 1
            for d in shape:
                n *= int(d)
            cap = first_cap if not buckets else later_cap
            used_bytes = (cursor - cur.start) * esize
            if cur.segments and (used_bytes + n * esize > cap or pos == tail_start):
                cursor = _round_up(cursor, bkt_align)
                cur.end = cursor
                buckets.append(cur)
                cur = Bucket(len(buckets), cursor, cursor)
            seg = Segment(name, cursor, n, shape, idx)
            segments.append(seg)
            cur.segments.append(seg)
            cursor = _round_up(cursor + n, seg_align)
        if cur.segments:
            cursor = _round_up(cursor, bkt_align)
            cur.end = cursor
            buckets.append(cur)
        plans[dtype] = ArenaPlan(dtype, cursor, segments, buckets)
    return plans


def arena_options(cfg) -> dict:
    """GradArena keyword arguments from an ``EngineConfig`` (HVD_FIRST_BUCKET_BYTES,
    HVD_FUSION_THRESHOLD, HVD_TAIL_BUCKET_BYTES, HVD_NUM_BUCKETS, HVD_COMM_DTYPE)."""
    comm = (cfg.comm_dtype or "auto").lower()
    if comm not in ("auto", "bf16", "fp32", "float32", "bfloat16"):
        raise ValueError("unknown HVD_COMM_DTYPE %r (expected auto|bf16|fp32)" % cfg.comm_dtype)
    return {"first_bucket_bytes": cfg.first_bucket_bytes, "bucket_cap_bytes": cfg.fusion_threshold,
            "tail_bucket_bytes": cfg.tail_bucket_bytes, "num_buckets": cfg.num_buckets,
            "grad_dtype": torch.float32 if comm in ("fp32", "float32") else None}


def _hip_memset(t) -> bool:
    try:
        from ..ops import kernels

        if not kernels.enabled():
            return False
        kernels.memset_async(t, 0)
        return True
    except Exception:  # pragma: no cover - ops layer unavailable
        return False


def _shaped_view(flat: torch.Tensor, seg: Segment, like: torch.Tensor) -> torch.Tensor:
    """View of ``flat[seg]`` with ``like``'s logical shape and dense stride order (so a
    channels-last conv weight keeps KRSC storage inside the arena)."""
    chunk = flat[seg.offset: seg.offset + seg.numel]
    if like.dim() == 4 and like.is_contiguous(memory_format=torch.channels_last) \
            and not like.is_contiguous():
        o, i, r, s = like.shape
        return chunk.view(o, r, s, i).permute(0, 3, 1, 2)
    return chunk.view(like.shape)


class GradArena:
    """Flat gradient storage for a list of parameters, with ``param.grad`` views.

    Optionally also relocates the parameters themselves into a flat arena with the same
    offsets (``flatten_params=True``) so that a fused optimizer can update a whole bucket with
    one kernel launch over ``params[a:b], grads[a:b], momentum[a:b]``.
    """

    def __init__(self, named_params: Iterable, first_bucket_bytes: int, bucket_cap_bytes: int,
                 flatten_params: bool = False, grad_dtype=None, tail_bucket_bytes: int = 0,
                 num_buckets: int = 0):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters")
        names = [n for n, _ in named]
        if len(set(names)) != len(names):
            dup = sorted({n for n in names if names.count(n) > 1})
            raise ValueError("parameter names are not unique: %s" % ", ".join(dup[:5]))
        devices = {p.device for _, p in named}
        if len(devices) != 1:
            raise ValueError("all parameters must live on one device, got %s" % devices)
        self.device = devices.pop()
        self.named = named
        # arenas are keyed by PARAMETER dtype (parameter, gradient, momentum and master arenas of
        # one key share their offsets); the gradient arena's own dtype may differ (grad_dtype)
        self.grad_dtype = {p.dtype: (grad_dtype or p.dtype) for _, p in named}
        wire = {dt: torch.empty((), dtype=g).element_size() for dt, g in self.grad_dtype.items()}
        self.plans = plan_arenas([(n, p.shape, p.dtype) for n, p in named],
                                 first_bucket_bytes, bucket_cap_bytes,
                                 tail_bucket_bytes=tail_bucket_bytes, num_buckets=num_buckets,
                                 wire_esize=wire)
        self.grads = {dt: torch.zeros(pl.numel, dtype=self.grad_dtype[dt], device=self.device)
                      for dt, pl in self.plans.items()}
        self.params = {}
        self.param_of_segment = {}
        self.bucket_of_param = {}
        self.views = {}
        for dt, plan in self.plans.items():
            if flatten_params:
                self.params[dt] = torch.zeros(plan.numel, dtype=dt, device=self.device)
            for b in plan.buckets:
                for seg in b.segments:
                    p = named[seg.index][1]
                    if flatten_params:
                        pv = _shaped_view(self.params[dt], seg, p)
                        with torch.no_grad():
                            pv.copy_(p)
                        p.data = pv
                    gv = _shaped_view(self.grads[dt], seg, p)
                    if gv.dtype != p.dtype and hasattr(p, "grad_dtype"):
                        p.grad_dtype = gv.dtype   # torch >= 2.9 checks .grad against it (HVD_COMM_DTYPE=fp32)
                    p.grad = gv
                    self.views[id(p)] = gv
                    self.param_of_segment[(dt, seg.offset)] = p
                    self.bucket_of_param[id(p)] = (dt, b.index)

    def all_buckets(self):
        """[(dtype, Bucket)] in planned launch order (largest-dtype arenas interleaved by
        bucket index so early buckets of every dtype go first)."""
        out = []
        for dt, plan in self.plans.items():
            out.extend((dt, b) for b in plan.buckets)
        out.sort(key=lambda t: t[1].index)
        return out

    def bucket_tensor(self, dtype, bucket: Bucket) -> torch.Tensor:
        return self.grads[dtype][bucket.start: bucket.end]

    def zero_(self):
        """K16: one fill per arena (the reference's ``optimizer.zero_grad()`` :212 issues 161)."""
        for g in self.grads.values():
            if g.is_cuda and _hip_memset(g):
                continue
            g.zero_()

    def ensure_view(self, p) -> torch.Tensor:
        """Re-attach ``p.grad`` to its arena slot if user code replaced or dropped it."""
        view = self.views[id(p)]
        if p.grad is None:
            p.grad = view
        elif p.grad.data_ptr() != view.data_ptr():
            with torch.no_grad():
                view.copy_(p.grad)
            p.grad = view
        return view

    def describe(self) -> str:
        lines = []
        for dt, plan in self.plans.items():
            for b in plan.buckets:
                esz = self.grads[dt].element_size()
                wire = "" if self.grads[dt].dtype == dt else \
                    " as %s" % str(self.grads[dt].dtype).replace("torch.", "")
                lines.append("%s bucket %d: %d tensors, %.2f MiB%s" % (
                    str(dt).replace("torch.", ""), b.index, len(b.segments),
                    b.numel * esz / 2 ** 20, wire))
        return "\n".join(lines)
