"""Arena-based fused SGD-momentum and LARS optimizers (kernels K14/K15/K16 of SURVEY §2.6).

The reference uses ``optim.SGD(model.parameters(), lr=0.001*size, momentum=0.9)``
(``imagenet_pytorch_horovod.py:333``): a Python loop issuing ~4 tiny device launches for each
of ResNet-50's 161 parameter tensors (~650 launches per step).  The Keras
flavour adds L2 weight decay 5e-5 on kernels (``imagenet_keras_horovod.py:104-115``) and
BASELINE.json config 4 adds LARS for 8×1024 large-batch training.

MI355X design: parameters, gradients, momentum and fp32 master weights live in flat per-dtype
arenas with identical offsets (``parallel.bucket.GradArena`` with ``flatten_params=True``), so
an update is **one launch per hyper-parameter run** — two for ResNet-50 (bf16 conv/fc weights
with weight decay, fp32 BatchNorm/bias without) — and the gradient arena doubles as the
all-reduce buffer.  bf16 parameters are updated from fp32 masters (bf16 has 8 mantissa bits;
``lr·g`` would vanish against the weight otherwise) and the bf16 copy is rewritten in the
same pass.  ``zero_grad`` is one fill launch per arena.

Update rule (identical to ``torch.optim.SGD``): ``g' = g + wd·w``; ``v = μ·v + g'``;
``w = w − lr·(g' + μ·v if nesterov else v)``.
LARS (You et al. 2017) for tensors with ndim > 1: ``λ = η·‖w‖ / (‖g‖ + wd·‖w‖ + ε)``;
``v = μ·v + lr·λ·(g + wd·w)``; ``w = w − v``; 1-D tensors (BN, bias) use plain momentum SGD.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from ..parallel.bucket import GradArena, arena_options
from ..utils.config import EngineConfig

__all__ = ["FusedSGD", "FusedLARS", "default_weight_decay_filter"]


def default_weight_decay_filter(name: str, param: torch.Tensor) -> bool:
    """Weight decay applies to conv / dense kernels only — the layers that carry a
    ``kernel_regularizer`` in the Keras flavour (imagenet_keras_horovod.py:104-109)."""
    return param.dim() > 1


def _dtype_key(dt) -> str:
    return str(dt).replace("torch.", "")


class _ArenaOptimizer(Optimizer):
    def __init__(self, named_parameters, defaults, weight_decay_filter, engine_config=None):
        named = [(n, p) for n, p in named_parameters if p.requires_grad]
        if named and not isinstance(named[0], tuple):
            raise TypeError("pass model.named_parameters(), not model.parameters()")
        super().__init__([p for _, p in named], defaults)
        if len(self.param_groups) != 1:
            raise ValueError("arena optimizers use a single parameter group")
        cfg = engine_config or EngineConfig.from_env()
        self.arena = GradArena(named, flatten_params=True, **arena_options(cfg))
        flt = weight_decay_filter or default_weight_decay_filter
        self._names = {id(p): n for n, p in named}
        self.momentum_buf = {}
        self.master = {}
        for dt, plan in self.arena.plans.items():
            dev = self.arena.device
            self.momentum_buf[dt] = torch.zeros(plan.numel, dtype=torch.float32, device=dev)
            if dt != torch.float32:
                self.master[dt] = self.arena.params[dt].to(torch.float32)
        # runs: maximal [start, end) ranges of consecutive segments with equal decay/adapt flags
        self.runs = []
        for dt, plan in self.arena.plans.items():
            cur = None
            for seg in plan.segments:
                p = named[seg.index][1]
                decay = bool(flt(seg.name, p))
                adapt = p.dim() > 1
                end = seg.offset + seg.numel
                if cur is not None and cur["decay"] == decay and cur["adapt"] == adapt:
                    cur["end"] = end
                    cur["segments"].append(seg)
                else:
                    cur = {"dtype": dt, "start": seg.offset, "end": end, "decay": decay,
                           "adapt": adapt, "segments": [seg]}
                    self.runs.append(cur)
        self._step = 0
        # device-side learning-rate multiplier (see freeze_lr): None = lr is passed by value
        self.lr_scale = None
        self._frozen_lr = None
        try:  # gradients are arena views: fused HIP blocks may accumulate into them directly
            from ..ops import hip_ops as _hip_ops

            _hip_ops.set_direct_grads(True)
        except Exception:  # pragma: no cover
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _weights(self, dt):
        """fp32 view of the weights being optimised (masters for low-precision arenas)."""
        return self.master[dt] if dt in self.master else self.arena.params[dt]

    def _publish(self, dt, start, end):
        if dt in self.master:
            self.arena.params[dt][start:end].copy_(self.master[dt][start:end])

    def zero_grad(self, set_to_none=False):
        self.arena.zero_()

    # -- learning rate under a captured HIP graph -------------------------------------------
    def freeze_lr(self):
        """From now on the kernels receive the CURRENT lr by value times a 1-element device
        tensor (``lr_scale``).  A captured graph (training/graph.py) then follows schedules:
        ``sync_lr()`` — outside the graph — rewrites that float when ``param_groups[0]["lr"]``
        moved.  HIP kernels only; the CPU reference path keeps reading ``lr`` directly."""
        if self.lr_scale is None:
            lr = float(self.param_groups[0]["lr"])
            # the by-value factor is exactly 1.0 and the device scalar carries the absolute learning
            # rate: 1.0f * fp32(lr) is the very float the by-value path passes, so a frozen step is
            # bit-identical to an unfrozen one (a ratio lr / lr0 is not: fp32(lr0) * fp32(lr / lr0)
            # rounds differently, tests/test_gpu_tape.py)
            self._frozen_lr = 1.0
            self._scale_written = lr
            self.lr_scale = torch.full((1,), self._scale_written, dtype=torch.float32,
                                       device=self.arena.device)
        return self.lr_scale

    def sync_lr(self):
        if self.lr_scale is None:
            return
        lr = float(self.param_groups[0]["lr"])
        scale = lr / self._frozen_lr
        if scale != self._scale_written:
            self.lr_scale.fill_(scale)
            self._scale_written = scale

    def _kernel_lr(self, lr):
        """(lr by value, device multiplier) for the HIP kernels."""
        if self.lr_scale is None:
            return lr, None
        return self._frozen_lr, self.lr_scale

    @staticmethod
    def _notify_weights_changed(hip):
        if hip:  # parameters were rewritten through raw pointers: drop derived weight copies
            from ..ops import hip_ops

            hip_ops.weights_changed()

    def after_replayed_step(self):
        """Host-side bookkeeping of a step that ran from the launch tape (training/tape.py): the
        device work was replayed, the Python ``step()`` was not."""
        self._step += 1

    # -- checkpoint surface ----------------------------------------------------------------
    def state_dict(self):
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        arenas = {}
        for dt in self.arena.plans:
            entry = {"momentum": self.momentum_buf[dt]}
            if dt in self.master:
                entry["master"] = self.master[dt]
            arenas[_dtype_key(dt)] = entry
        layout = {_dtype_key(dt): [[s.name, s.offset, s.numel] for s in plan.segments]
                  for dt, plan in self.arena.plans.items()}
        return {"param_groups": groups, "arenas": arenas, "layout": layout, "step": self._step}

    def load_state_dict(self, state):
        for g, saved in zip(self.param_groups, state["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
        self._step = int(state.get("step", 0))
        for dt, plan in self.arena.plans.items():
            key = _dtype_key(dt)
            if key not in state["arenas"]:
                raise KeyError("optimizer state has no arena for %s" % key)
            saved_layout = state.get("layout", {}).get(key)
            mine = [[s.name, s.offset, s.numel] for s in plan.segments]
            if saved_layout is not None and [list(x) for x in saved_layout] != mine:
                raise ValueError("optimizer arena layout differs from the checkpoint's (%s)" % key)
            entry = state["arenas"][key]
            self.momentum_buf[dt].copy_(entry["momentum"])
            if dt in self.master:
                if "master" in entry:
                    self.master[dt].copy_(entry["master"])
                else:
                    self.master[dt].copy_(self.arena.params[dt])
                self._publish(dt, 0, plan.numel)

    def master_state_dict(self, model):
        """{state_dict key: fp32 master tensor} for parameters held in low precision."""
        out = {}
        for dt, plan in self.arena.plans.items():
            if dt not in self.master:
                continue
            for seg in plan.segments:
                p = self.arena.named[seg.index][1]
                flat = self.master[dt][seg.offset: seg.offset + seg.numel]
                if p.dim() == 4 and not p.is_contiguous():
                    o, i, r, s = p.shape
                    flat = flat.view(o, r, s, i).permute(0, 3, 1, 2).contiguous()
                out[seg.name] = flat.reshape(p.shape)
        return out

    def load_master_from_state(self, model_state, model=None):
        """After a checkpoint load: masters ← the checkpoint's fp32 model entries."""
        for dt, plan in self.arena.plans.items():
            if dt not in self.master:
                continue
            for seg in plan.segments:
                if seg.name not in model_state:
                    continue
                p = self.arena.named[seg.index][1]
                src = model_state[seg.name].to(torch.float32).reshape(p.shape)
                if p.dim() == 4 and not p.is_contiguous():
                    src = src.permute(0, 2, 3, 1).contiguous()
                self.master[dt][seg.offset: seg.offset + seg.numel].copy_(src.reshape(-1))
            self._publish(dt, 0, plan.numel)

    def sync_master_from_root(self, root_rank=0):
        from ..parallel import functions as _f

        _f.coalesced_broadcast_(list(self.master.values()) + list(self.momentum_buf.values()),
                                root_rank, name="optimizer_arenas")
        for dt, plan in self.arena.plans.items():
            self._publish(dt, 0, plan.numel)

    def refresh_master_from_params(self):
        """Call after parameters were overwritten from outside (e.g. broadcast_parameters)."""
        for dt in self.master:
            self.master[dt].copy_(self.arena.params[dt])


class FusedSGD(_ArenaOptimizer):
    def __init__(self, named_parameters, lr=1e-3, momentum=0.9, weight_decay=0.0,
                 nesterov=False, weight_decay_filter=None, engine_config=None):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("negative hyper-parameter")
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov)
        super().__init__(named_parameters, defaults, weight_decay_filter, engine_config)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.begin_step()
        for run in self.runs:
            self._update(run["dtype"], run["start"], run["end"], run["decay"])
        self.end_step()
        return loss

    # -- bucket-wise stepping (DistributedOptimizer: update bucket k right after the stream has
    #    waited for bucket k's all-reduce, while later buckets are still in flight) ------------
    def begin_step(self):
        from ..ops import kernels

        self._hip = self.arena.device.type == "cuda" and kernels.enabled()

    @torch.no_grad()
    def step_bucket(self, dtype, start, end):
        """Update the slice [start, end) of ``dtype``'s arenas (any range: it is intersected with
        the hyper-parameter runs)."""
        for run in self.runs:
            if run["dtype"] != dtype:
                continue
            a, b = max(start, run["start"]), min(end, run["end"])
            if a < b:
                self._update(dtype, a, b, run["decay"])

    def end_step(self):
        self._step += 1
        self._notify_weights_changed(self._hip)
        return None

    def _update(self, dt, a, b, decay):
        g0 = self.param_groups[0]
        lr, mu, wd, nesterov = g0["lr"], g0["momentum"], g0["weight_decay"], g0["nesterov"]
        run_wd = wd if decay else 0.0
        grads = self.arena.grads[dt][a:b]
        w = self._weights(dt)[a:b]
        v = self.momentum_buf[dt][a:b]
        if self._hip:
            from ..ops import kernels

            lowp = self.arena.params[dt][a:b] if dt in self.master else None
            klr, kscale = self._kernel_lr(lr)
            kernels.fused_sgd_(w, grads, v, lowp, klr, mu, run_wd, nesterov, kscale)
            return
        g = grads.to(torch.float32)
        if run_wd:
            g = g.add(w, alpha=run_wd)
        v.mul_(mu).add_(g)
        w.add_(g.add(v, alpha=mu) if nesterov else v, alpha=-lr)
        self._publish(dt, a, b)


class FusedLARS(_ArenaOptimizer):
    def __init__(self, named_parameters, lr=1e-3, momentum=0.9, weight_decay=0.0,
                 trust_coefficient=0.001, eps=1e-8, weight_decay_filter=None,
                 engine_config=None):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay,
                        trust_coefficient=trust_coefficient, eps=eps)
        super().__init__(named_parameters, defaults, weight_decay_filter, engine_config)
        # per-segment table for the segmented-norm kernel, one per adaptive run
        for run in self.runs:
            if run["adapt"]:
                offs = [s.offset - run["start"] for s in run["segments"]] + \
                       [run["end"] - run["start"]]
                run["seg_offsets"] = torch.tensor(offs, dtype=torch.int64,
                                                  device=self.arena.device)
                ends = [s.offset + s.numel - run["start"] for s in run["segments"]]
                run["seg_ends"] = torch.tensor(ends, dtype=torch.int64, device=self.arena.device)

    def trust_ratios(self, run, lr_unused=None):
        """λ per segment of an adaptive run (fp32 tensor)."""
        g0 = self.param_groups[0]
        wd = g0["weight_decay"] if run["decay"] else 0.0
        dt, a = run["dtype"], run["start"]
        w = self._weights(dt)
        g = self.arena.grads[dt]
        from ..ops import kernels

        if self.arena.device.type == "cuda" and kernels.enabled():
            return kernels.lars_trust_ratios(w[a:run["end"]], g[a:run["end"]], run["seg_offsets"], run["seg_ends"],
                                             g0["trust_coefficient"], wd, g0["eps"])
        else:
            wn = torch.stack([w[s.offset: s.offset + s.numel].norm() for s in run["segments"]])
            gn = torch.stack([g[s.offset: s.offset + s.numel].float().norm()
                              for s in run["segments"]])
        ratio = g0["trust_coefficient"] * wn / (gn + wd * wn + g0["eps"])
        one = torch.ones_like(ratio)
        return torch.where((wn > 0) & (gn > 0), ratio, one)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g0 = self.param_groups[0]
        lr, mu, wd = g0["lr"], g0["momentum"], g0["weight_decay"]
        from ..ops import kernels

        hip = self.arena.device.type == "cuda" and kernels.enabled()
        for run in self.runs:
            dt, a, b = run["dtype"], run["start"], run["end"]
            run_wd = wd if run["decay"] else 0.0
            w = self._weights(dt)
            v = self.momentum_buf[dt]
            grads = self.arena.grads[dt]
            if not run["adapt"]:
                if hip:
                    lowp = self.arena.params[dt][a:b] if dt in self.master else None
                    klr, kscale = self._kernel_lr(lr)
                    kernels.fused_sgd_(w[a:b], grads[a:b], v[a:b], lowp, klr, mu, run_wd, False, kscale)
                    continue
                g = grads[a:b].to(torch.float32)
                if run_wd:
                    g = g.add(w[a:b], alpha=run_wd)
                v[a:b].mul_(mu).add_(g)
                w[a:b].add_(v[a:b], alpha=-lr)
                self._publish(dt, a, b)
                continue
            ratios = self.trust_ratios(run)
            if hip:
                lowp = self.arena.params[dt][a:b] if dt in self.master else None
                klr, kscale = self._kernel_lr(lr)
                kernels.fused_lars_(w[a:b], grads[a:b], v[a:b], lowp, ratios,
                                    run["seg_offsets"], run["seg_ends"], klr, mu, run_wd, kscale)
                continue
            for k, s in enumerate(run["segments"]):
                sl = slice(s.offset, s.offset + s.numel)
                g = grads[sl].to(torch.float32)
                if run_wd:
                    g = g.add(w[sl], alpha=run_wd)
                v[sl].mul_(mu).add_(g * (lr * ratios[k]))
                w[sl].sub_(v[sl])
            self._publish(dt, a, b)
        self._step += 1
        self._notify_weights_changed(hip)
        return loss
