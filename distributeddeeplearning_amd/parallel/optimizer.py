"""``DistributedOptimizer``: gradient exchange overlapped with backward.

Parity target: ``hvd.DistributedOptimizer(optimizer, named_parameters=...)`` at
``HorovodPytorch/src/imagenet_pytorch_horovod.py:336-338``: the
returned object *is-a* the wrapped optimizer class, gradients are allreduce-averaged
asynchronously as backward produces them, ``step()`` waits for the exchange first, parameter
names are collective keys and duplicates are rejected.

What differs from Horovod's design (MI355X-first):

* one collective per **static bucket** (a contiguous arena slice), not per tensor and not per
  5 ms negotiation cycle; buckets launch from ``register_post_accumulate_grad_hook`` when their
  last gradient lands, in reverse-registration order, onto RCCL's stream;
* ``param.grad`` are views into the arena: nothing is packed or unpacked;
* averaging is fused into the reduction (``ncclAvg``);
* ``step()`` orders the compute stream after the collectives with event waits — the host
  never blocks.  With ``optim.FusedSGD`` the update runs bucket by bucket: the stream waits for
  bucket k, updates bucket k's slice of the arenas, then waits for bucket k+1 — the update of
  the early buckets overlaps the tail of the exchange (``FusedLARS`` needs every norm first and
  updates after the last bucket);
* per step the exposed communication time (how long the compute stream actually sat in those
  waits, from HIP events) and the device time of every bucket's collective (from the engine)
  are available through ``comm_metrics()`` and land in ``metrics.jsonl``;
* the first backward records the actual hook order; a mismatch with the plan is reported in
  debug mode (``HVD_DEBUG=1``) together with a cross-rank plan hash check.
"""
from __future__ import annotations

import hashlib
import os

import torch

from ..ops import tape as _tape
from ..utils import trace as _trace
from . import backend as _b
from . import timeline as _tl
from .bucket import GradArena, arena_options
from .compression import Compression

__all__ = ["DistributedOptimizer"]


class _NullContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _grad_exchange_context():
    """Fused HIP blocks may have written some gradients of the bucket on their weight-gradient
    stream: launch the collective where it is ordered after both streams."""
    try:
        from ..ops import hip_ops as _hip_ops

        return _hip_ops.grad_exchange_context()
    except Exception:  # pragma: no cover - ops layer unavailable
        return _NullContext()


class _WeakListener:
    """Readiness callback registered with ``ops.hip_ops``; dies with its optimizer."""

    def __init__(self, opt):
        import weakref

        self.ref = weakref.ref(opt)

    def __call__(self, p):
        opt = self.ref()
        if opt is None:
            return False  # tells hip_ops to drop this listener
        opt._ddl_on_direct_grad(p)
        return True


class _CommMetrics:
    """Exposed communication time per step from HIP events around the stream waits of
    ``synchronize()``; read one step late so that nothing ever synchronises with the device."""

    def __init__(self):
        self.pairs = []        # events recorded this step
        self.prev = []         # last step's pairs (complete by the time the next step waits)
        self.pool = []
        self.last_exposed_ms = None
        self.total_exposed_ms = 0.0
        self.steps = 0

    def _event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def before_wait(self):
        if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return None
        e = self._event()
        e.record()
        return e

    def after_wait(self, e0):
        if e0 is None:
            return
        e1 = self._event()
        e1.record()
        self.pairs.append((e0, e1))

    def end_step(self):
        # harvest the PREVIOUS step's events (long finished), then rotate
        if self.prev and all(b.query() for _, b in self.prev):
            ms = sum(a.elapsed_time(b) for a, b in self.prev)
            self.last_exposed_ms = ms
            self.total_exposed_ms += ms
            self.steps += 1
            for a, b in self.prev:
                self.pool += [a, b]
            self.prev = []
        if not self.prev:
            self.prev, self.pairs = self.pairs, []
        else:  # still running (host far ahead): drop this step's sample rather than wait
            for a, b in self.pairs:
                self.pool += [a, b]
            self.pairs = []


class _BucketState:
    __slots__ = ("dtype", "bucket", "tensor", "pending", "expected", "work", "launched",
                 "needs_scale", "wire", "ctx", "early_done")

    def __init__(self, dtype, bucket, tensor, passes=1):
        self.dtype = dtype
        self.bucket = bucket
        self.tensor = tensor
        self.expected = len(bucket.segments) * passes
        self.pending = self.expected
        self.work = None
        self.launched = False
        self.needs_scale = False
        self.wire = None
        self.ctx = None
        self.early_done = False      # updated during backward (DistributedOptimizer.enable_early_update)


class _DistributedMixin:
    """Methods grafted onto a dynamic subclass of the user's optimizer class."""

    # -- construction ---------------------------------------------------------------------
    def _ddl_setup(self, named_parameters, compression, backward_passes_per_step, op,
                   arena=None):
        cfg = _b.engine_config()
        if named_parameters is None:
            named_parameters = [("param.%d.%d" % (gi, pi), p)
                                for gi, g in enumerate(self.param_groups)
                                for pi, p in enumerate(g["params"])]
        named_parameters = list(named_parameters)
        in_optimizer = {id(p) for g in self.param_groups for p in g["params"]}
        named = [(n, p) for n, p in named_parameters if id(p) in in_optimizer]
        missing = in_optimizer - {id(p) for _, p in named}
        if missing:
            raise ValueError("named_parameters does not cover %d optimizer parameter(s)"
                             % len(missing))
        self._ddl_compression = compression
        self._ddl_passes = max(int(backward_passes_per_step), 1)
        self._ddl_op = op
        self._ddl_step_count = 0
        self._ddl_order_seen = []
        self._ddl_order_checked = False
        self._ddl_direct_seen = set()
        self._ddl_early = False          # enable_early_update
        self._ddl_in_direct = False
        self._ddl_fault = _parse_fault(cfg.fault_inject)
        self._ddl_corrupt = _parse_fault(cfg.fault_corrupt)
        # replica check: after each of the first N steps (every step in debug mode) all ranks compare
        # checksums of the reduced gradients and of the updated parameters
        self._ddl_check_steps = (1 << 62) if cfg.debug else max(int(cfg.replica_check_steps), 0)
        if arena is None:
            arena = getattr(self, "arena", None)  # fused optimizers bring their own arena
        if arena is None:
            arena = GradArena(named, **arena_options(cfg))
        self._ddl_arena = arena
        self._ddl_buckets = [_BucketState(dt, b, arena.bucket_tensor(dt, b), self._ddl_passes)
                             for dt, b in arena.all_buckets()]
        self._ddl_bucket_index = {(s.dtype, s.bucket.index): s for s in self._ddl_buckets}
        self._ddl_hooks = []
        self._ddl_names = {id(p): n for n, p in named}
        for n, p in named:
            if p.requires_grad:
                self._ddl_hooks.append(p.register_post_accumulate_grad_hook(self._ddl_on_grad))
        # fused HIP blocks write gradients straight into the arena and report readiness here
        # instead of through autograd's accumulate hook.  The listener holds this optimizer
        # weakly: a dropped or re-wrapped optimizer must not keep receiving notifications (it
        # would re-point param.grad at its stale arena and launch unmatched collectives).
        try:
            from ..ops import hip_ops as _hip_ops

            self._ddl_listener = _WeakListener(self)
            _hip_ops.claim_grad_listener(self._ddl_listener, self._ddl_names)
            _hip_ops.set_direct_grads(True)
        except Exception:  # pragma: no cover - ops layer unavailable
            self._ddl_listener = None
        # HVD_ALLREDUCE_ALGO=twoshot|oneshot|auto: the arenas become peer-mapped regions of the
        # direct xGMI all-reduce (collective call: same order on every rank)
        c = _b.comm()
        if c is not None and hasattr(c, "register_region"):
            for dt in sorted(arena.grads, key=str):
                c.register_region(arena.grads[dt])
        self._ddl_metrics = _CommMetrics() if (c is not None and cfg.comm_metrics) else None
        if cfg.timeline and not _tl.enabled():
            _tl.start(cfg.timeline, _b.rank())
        if cfg.debug and _b.size() > 1:
            self._ddl_check_plan_hash()

    def _ddl_check_plan_hash(self):
        text = self._ddl_arena.describe() + "|" + "|".join(sorted(self._ddl_names.values()))
        digest = int.from_bytes(hashlib.sha1(text.encode()).digest()[:6], "big")
        probe = torch.tensor([digest, -digest], dtype=torch.int64, device=_b.device())
        _b.comm().all_reduce(probe, "max", name="plan_hash").wait()
        hi, neg_lo = probe.tolist()
        if hi != digest or -neg_lo != digest:
            raise RuntimeError("gradient bucket plans differ across ranks")

    # -- backward-time path ---------------------------------------------------------------
    def _ddl_arrive(self, p):
        arena = self._ddl_arena
        arena.ensure_view(p)
        if not self._ddl_order_checked:
            self._ddl_order_seen.append(self._ddl_names[id(p)])
        st = self._ddl_bucket_index[arena.bucket_of_param[id(p)]]
        st.pending -= 1
        if st.pending == 0 and not st.launched:
            self._ddl_launch(st)

    def _ddl_on_direct_grad(self, p):
        """Readiness of a gradient that a fused HIP block wrote straight into the arena.  autograd
        runs the parameter's post-accumulate hook as well when the block's node returns - even
        though the node hands back None for it (torch >= 2.x) - so the hook path skips parameters
        served here; counting both launched every bucket after HALF of its gradients."""
        if id(p) in self._ddl_names:
            self._ddl_direct_seen.add(id(p))
            self._ddl_in_direct = True
            try:
                self._ddl_arrive(p)
            finally:
                self._ddl_in_direct = False

    # -- one process: update a bucket as soon as its gradients are complete ----------------------------
    def enable_early_update(self, on=True):
        """ONE process (no communicator), fused HIP path, one backward pass per step, and nothing between
        ``backward()`` and ``step()`` that reads or rewrites gradients (clipping!): a bucket whose
        gradients are complete is updated right away, on the weight-gradient stream next to the rest of
        backward, instead of in ``step()`` behind it - the reference's update waits for the whole backward
        pass (``HorovodPytorch/src/imagenet_pytorch_horovod.py:216-218``).  Same kernels, same arguments,
        same values: backward reads the transposed weight copies (refreshed after ``step()``), the
        forward filters of a complete bucket have no reader left, and the stream is ordered behind the
        main stream's gradient writes.  Returns whether the mode is active."""
        base = super(self.__class__, self)
        ok = bool(on) and _b.size() == 1 and _b.comm() is None and self._ddl_passes == 1 and \
            getattr(base, "step_bucket", None) is not None and _b.engine_config().per_bucket_update
        if ok:
            try:
                from ..ops import kernels as _kernels

                ok = self._ddl_arena.device.type == "cuda" and _kernels.enabled()
            except Exception:  # pragma: no cover - ops layer unavailable
                ok = False
        self._ddl_early = ok
        return ok

    def _ddl_early_update(self, st):
        from ..ops import hip_ops as _hip_ops, kernels as _kernels

        base = super(self.__class__, self)
        raw = _hip_ops._side_stream(st.tensor.device).cuda_stream
        # behind the main stream's gradient writes queued so far (BatchNorm finalize kernels, the stem);
        # the weight gradients themselves were reduced on this very stream
        _kernels.stream_wait_stream(raw, _kernels._raw_stream(), _hip_ops._SIDE["event"])
        base.begin_step()
        with _kernels.launch_on(raw):
            base.step_bucket(st.dtype, st.bucket.start, st.bucket.end)
        _hip_ops._side_mark_used()      # the end-of-backward join orders the main stream behind the update
        st.early_done = True

    def _ddl_on_grad(self, p):
        if id(p) in self._ddl_direct_seen:
            return
        self._ddl_arrive(p)

    def _ddl_launch(self, st):
        st.launched = True
        if _b.size() == 1 and _b.comm() is None:
            if self._ddl_early and self._ddl_in_direct and not self._ddl_checking():
                self._ddl_early_update(st)
            return
        # the hand-off to the communication engine cannot live on a launch tape: it runs as a live
        # action, again and in place at every replay of a recorded step (ops/tape.py)
        _tape.live(lambda: self._ddl_launch_now(st))

    def _ddl_launch_now(self, st):
        name = "grad.%s.bucket%d" % (str(st.dtype).replace("torch.", ""), st.bucket.index)
        _tl.begin(name, "ALLREDUCE")
        _trace.mark("launch " + name)
        buf = st.tensor
        with _grad_exchange_context():
            wire, ctx = self._ddl_compression.compress(buf)
            if _b.backend_name() == "gloo" and wire.is_cuda:
                wire = wire.cpu()
            st.wire, st.ctx = wire, ctx
            if _b.supports_avg():
                st.work = _b.comm().all_reduce(wire, "average", name)
                st.needs_scale = False
            else:
                st.work = _b.comm().all_reduce(wire, "sum", name)
                st.needs_scale = True

    def _ddl_finish(self, st):
        if st.work is None:
            return
        _tape.live(lambda: self._ddl_finish_now(st))

    def _ddl_finish_now(self, st):
        if st.work is None:
            return
        deadline = _b.engine_config().collective_timeout_s
        m = self._ddl_metrics
        e0 = m.before_wait() if (m is not None and st.tensor.is_cuda) else None
        # device collectives are stream-ordered (the host does not block); host ones wait here
        if not st.work.wait(deadline):
            raise RuntimeError("gradient allreduce timed out after %.0f s" % deadline)
        if e0 is not None:
            m.after_wait(e0)
        wire = st.wire
        if st.needs_scale:
            wire.div_(_b.size())
        out = self._ddl_compression.decompress(wire, st.ctx)
        if out.data_ptr() != st.tensor.data_ptr():
            st.tensor.copy_(out, non_blocking=True)
        st.work = st.wire = st.ctx = None
        _tl.end("grad.%s.bucket%d" % (str(st.dtype).replace("torch.", ""), st.bucket.index))

    def _ddl_prepare_sync(self):
        if self._ddl_fault is not None:
            frank, fstep = self._ddl_fault
            if _b.rank() == frank and self._ddl_step_count == fstep:
                os._exit(17)  # HVD_FAULT_INJECT: die abruptly mid-step
        for st in self._ddl_buckets:
            if not st.launched:
                self._ddl_launch(st)
        try:
            from ..ops import hip_ops as _hip_ops

            _hip_ops._join_side_stream()
        except Exception:  # pragma: no cover - ops layer unavailable
            pass

    def synchronize(self):
        """Launch any bucket whose gradients did not all arrive (unused parameters) and order
        the compute stream after every outstanding exchange."""
        self._ddl_prepare_sync()
        for st in self._ddl_buckets:
            self._ddl_finish(st)
        if self._ddl_metrics is not None:
            _tape.live(self._ddl_metrics.end_step)
        self._ddl_verify_order_once()

    def comm_metrics(self):
        """{"comm_exposed_ms": last measured step, "comm_exposed_ms_mean", "bucket_ms": {name:
        device ms of that bucket's last collective}} — None entries until measured."""
        out = {"comm_exposed_ms": None, "comm_exposed_ms_mean": None, "bucket_ms": {}}
        m = self._ddl_metrics
        if m is not None and m.steps:
            out["comm_exposed_ms"] = round(m.last_exposed_ms, 4)
            out["comm_exposed_ms_mean"] = round(m.total_exposed_ms / m.steps, 4)
        c = _b.comm()
        if c is not None and hasattr(c, "stats"):
            out["bucket_ms"] = {k: round(v["last_ms"], 4) for k, v in c.stats().items()
                                if k.startswith("grad.")}
        return out

    def _ddl_verify_order_once(self):
        """First step only: did the buckets COMPLETE in the order they were planned?  (Within a bucket
        the arrival order of its gradients is irrelevant: gamma/beta/weight of one layer arrive in
        another order than they were registered in, which says nothing about overlap.)"""
        if self._ddl_order_checked:
            return
        self._ddl_order_checked = True
        where = {}
        for st in self._ddl_buckets:
            for seg in st.bucket.segments:
                where[seg.name] = (str(st.dtype), st.bucket.index, len(st.bucket.segments))
        left, completed = {}, {}
        for n in self._ddl_order_seen:
            if n not in where:
                continue
            dt, bi, cnt = where[n]
            left[(dt, bi)] = left.get((dt, bi), cnt) - 1
            if left[(dt, bi)] == 0:
                completed.setdefault(dt, []).append(bi)
        inversions = sum(1 for seq in completed.values() for a, b in zip(seq, seq[1:]) if b < a)
        nb = sum(len(v) for v in completed.values())
        self._ddl_order_inversions = inversions
        self._ddl_buckets_completed = nb
        if _b.engine_config().debug and inversions > 0:
            import warnings

            warnings.warn("gradient buckets completed out of plan order in %d of %d places (%s); overlap "
                          "with backward will be reduced" % (inversions, nb, completed))
        self._ddl_order_seen = []

    # -- replica check ----------------------------------------------------------------------
    @staticmethod
    def _ddl_checksum(t):
        """[sum of words, position-weighted sum of words] of a tensor's bit pattern, wrapping int64:
        equal tensors give equal pairs, a flipped, missing or moved word changes them."""
        flat = t.detach().reshape(-1)
        words = flat.view(torch.int16 if flat.element_size() == 2 else torch.int32).to(torch.int64)
        idx = torch.arange(words.numel(), device=words.device, dtype=torch.int64) % 65521 + 1
        return torch.stack([words.sum(), (words * idx).sum()])

    def _ddl_replica_check(self, stage):
        """Collective.  ``stage`` = "gradients" (after the exchange, before the update) or
        "parameters" (after the update).  An xGMI coherence bug, a collective that skipped a
        shard or a rank that missed a broadcast shows up HERE, on step 1, as an exception naming the
        arena and the ranks - not hours later as a diverging loss."""
        if _b.trivial() or _b.size() == 1:
            return
        arena = self._ddl_arena
        names, sums = [], []
        for dt in sorted(arena.plans, key=str):
            key = str(dt).replace("torch.", "")
            if stage == "gradients":
                names.append("grad." + key)
                sums.append(self._ddl_checksum(arena.grads[dt]))
            else:
                src = None
                masters = getattr(self, "master", None)
                if isinstance(masters, dict) and dt in masters:
                    src = masters[dt]
                elif getattr(arena, "params", None) and dt in arena.params:
                    src = arena.params[dt]
                if src is None:   # parameters are not flattened: checksum them one by one
                    parts = [self._ddl_checksum(arena.named[seg.index][1]) for seg in arena.plans[dt].segments]
                    names.append("param." + key)
                    sums.append(torch.stack(parts).sum(0))
                else:
                    names.append("param." + key)
                    sums.append(self._ddl_checksum(src))
        mine = torch.stack(sums).reshape(-1).contiguous()
        if _b.backend_name() == "gloo" and mine.is_cuda:
            mine = mine.cpu()
        world = _b.size()
        out = torch.zeros((world, mine.numel()), dtype=torch.int64, device=mine.device)
        w = _b.comm().all_gather_into(out, mine, "replica_check.%s" % stage)
        if not w.wait(_b.engine_config().collective_timeout_s):
            raise RuntimeError("replica check timed out")
        rows = out.cpu().view(world, len(names), 2)
        for k, name in enumerate(names):
            ref = rows[0, k]
            differing = [r for r in range(world) if not torch.equal(rows[r, k], ref)]
            if differing:
                raise RuntimeError(
                    "replicas differ after step %d: %s of %s on rank(s) %s do not match rank 0 "
                    "(engine %s, all-reduce %s) - the gradient exchange delivered different data to "
                    "different ranks" % (self._ddl_step_count + (0 if stage == "gradients" else 0), stage, name,
                                         differing, _b.engine_name(),
                                         getattr(_b.comm(), "algo", "?")))

    def _ddl_checking(self):
        return self._ddl_step_count < self._ddl_check_steps and not _b.trivial() and _b.size() > 1

    def _ddl_maybe_corrupt(self):
        if self._ddl_corrupt is not None:
            crank, cstep = self._ddl_corrupt
            if _b.rank() == crank and self._ddl_step_count == cstep:
                for dt in sorted(self._ddl_arena.grads, key=str):
                    g = self._ddl_arena.grads[dt].view(-1)
                    g[g.numel() // 2] += 1.0   # HVD_FAULT_CORRUPT: one wrong word on one rank
                    break

    def _ddl_reset_buckets(self):
        self._ddl_direct_seen.clear()
        for st in self._ddl_buckets:
            st.pending = st.expected
            st.launched = False

    # -- optimizer surface ----------------------------------------------------------------
    def step(self, closure=None):
        base = super(self.__class__, self)
        per_bucket = getattr(base, "step_bucket", None)
        checking = self._ddl_checking()
        if per_bucket is None or closure is not None or not _b.engine_config().per_bucket_update or checking:
            if any(st.early_done for st in self._ddl_buckets):
                raise RuntimeError("early bucket updates already ran in this step: the whole-arena step would apply "
                                   "them twice (closure / replica check with enable_early_update)")
            self.synchronize()
            self._ddl_reset_buckets()
            self._ddl_maybe_corrupt()
            if checking:
                self._ddl_replica_check("gradients")
            out = base.step(closure)
            if checking:
                self._ddl_replica_check("parameters")
            self._ddl_step_count += 1
            return out
        # fused optimizer: wait for bucket k, update bucket k, wait for bucket k+1 ... - the
        # update of the early buckets runs while the tail of the exchange is still in flight
        self._ddl_prepare_sync()
        base.begin_step()
        for st in self._ddl_buckets:
            with _trace.range("wait+update %s.bucket%d" % (str(st.dtype).replace("torch.", ""), st.bucket.index)):
                self._ddl_finish(st)
                if st.early_done:       # updated during backward (enable_early_update)
                    st.early_done = False
                    continue
                per_bucket(st.dtype, st.bucket.start, st.bucket.end)
        if self._ddl_metrics is not None:
            _tape.live(self._ddl_metrics.end_step)
        self._ddl_verify_order_once()
        self._ddl_reset_buckets()
        self._ddl_step_count += 1
        return base.end_step()

    def after_replayed_step(self):
        self._ddl_step_count += 1
        done = getattr(super(self.__class__, self), "after_replayed_step", None)
        if done is not None:
            done()

    def zero_grad(self, set_to_none=False):
        """Zero the arenas (one memset per dtype).  Gradients stay views into the arena, so
        ``set_to_none`` is ignored by design."""
        if any(st.work is not None for st in self._ddl_buckets):
            raise AssertionError("zero_grad() called between backward and step(): an "
                                 "allreduce is still in flight")
        self._ddl_arena.zero_()

    def describe_buckets(self) -> str:
        return self._ddl_arena.describe()


def _timedelta(seconds):
    import datetime

    return datetime.timedelta(seconds=seconds)


def _parse_fault(spec):
    if not spec:
        return None
    r, s = spec.split(":")
    return int(r), int(s)


def DistributedOptimizer(optimizer, named_parameters=None, compression=Compression.none,
                         backward_passes_per_step=1, op=None):
    """Wrap ``optimizer`` so that gradients are averaged across ranks during backward.

    The instance's class is replaced by a dynamic subclass of its own class mixed with the
    exchange logic, so ``isinstance(opt, type_of_original)`` keeps holding and optimizer state
    and hyper-parameters are untouched.
    """
    if not _b.is_initialized():
        _b.init()
    base = optimizer.__class__
    if getattr(base, "_ddl_wrapped", False):
        raise ValueError("optimizer is already a DistributedOptimizer")
    body = {k: v for k, v in _DistributedMixin.__dict__.items()
            if k not in ("__dict__", "__weakref__")}
    body["_ddl_wrapped"] = True
    cls = type(base.__name__, (base,), body)
    optimizer.__class__ = cls
    optimizer._ddl_setup(named_parameters, compression, backward_passes_per_step, op)
    return optimizer
