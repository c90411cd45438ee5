"""PyTorch ``state_dict`` checkpoints with rank-0 save and broadcast resume.

The reference's PyTorch flavour never writes a checkpoint; its Keras
flavour designs the flow but hard-codes ``resume_from_epoch = 0``
(``imagenet_keras_horovod.py:287-291,310-324``).  This module makes that flow functional for
PyTorch (north-star requirement):

    save:    rank 0 only → ``checkpoint-{epoch}.pt`` written atomically (tmp + rename)
    resume:  rank 0 finds the latest epoch → ``broadcast`` of the epoch scalar → rank 0 loads
             → ``broadcast_parameters`` + ``broadcast_optimizer_state`` equalise every rank

Model entries use torchvision key names and OIHW logical weight shapes; bf16 compute models
export their fp32 master weights when the optimizer holds them, so a checkpoint loads into a
stock fp32 torchvision ResNet.  Files are loaded with ``weights_only=True``.
"""
from __future__ import annotations

import os
import re

import torch

from ..parallel import backend as _b
from ..parallel import collectives as _c
from ..parallel import functions as _f

__all__ = ["save", "load", "latest_epoch", "resume", "checkpoint_path", "model_dir", "restore_rng"]

_PATTERN = re.compile(r"^checkpoint-(\d+)\.pt$")


def model_dir(output_dir=None, temp_dir=None):
    """Rank 0 writes to the output directory; other ranks get a scratch directory
    (reference ``_get_model_dir`` imagenet_keras_horovod.py:181-191)."""
    if not _b.is_initialized() or _b.rank() == 0:
        return output_dir
    return temp_dir


def checkpoint_path(directory, epoch) -> str:
    return os.path.join(directory, "checkpoint-%d.pt" % epoch)


def _export_model_state(model, optimizer):
    state = {k: v.detach() for k, v in model.state_dict().items()}
    masters = getattr(optimizer, "master_state_dict", None)
    if callable(masters):
        by_name = masters(model)
        for k, v in by_name.items():
            if k in state:
                state[k] = v.detach().reshape(state[k].shape)
    return {k: v.to("cpu").contiguous() for k, v in state.items()}


def save(directory, epoch, model, optimizer=None, scheduler=None, extra=None, keep_last=None):
    """Write ``checkpoint-{epoch}.pt`` (call on rank 0 only; other ranks return None)."""
    if _b.is_initialized() and _b.rank() != 0:
        return None
    os.makedirs(directory, exist_ok=True)
    payload = {
        "epoch": int(epoch),
        "model": _export_model_state(model, optimizer),
        "rng": _rng_state(),
        "extra": extra or {},
    }
    if optimizer is not None:
        payload["optimizer"] = optimizer.state_dict()
    if scheduler is not None and hasattr(scheduler, "state_dict"):
        payload["scheduler"] = scheduler.state_dict()
    path = checkpoint_path(directory, epoch)
    tmp = path + ".tmp.%d" % os.getpid()
    torch.save(payload, tmp)
    os.replace(tmp, path)
    if keep_last:
        epochs = sorted(_list_epochs(directory))
        for old in epochs[:-keep_last]:
            try:
                os.remove(checkpoint_path(directory, old))
            except OSError:
                pass
    return path


def _rng_state():
    """Generator states a resumed run needs to continue the SAME random sequence (data order of
    the synthetic loader, dropout-free model: the loaders draw from torch's generators)."""
    state = {"torch": torch.get_rng_state()}
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        state["cuda"] = torch.cuda.get_rng_state()
    return state


def restore_rng(state):
    if not state:
        return
    if "torch" in state:
        torch.set_rng_state(state["torch"].to(torch.uint8).cpu())
    if "cuda" in state and torch.cuda.is_available():
        torch.cuda.set_rng_state(state["cuda"].to(torch.uint8).cpu())


def _list_epochs(directory):
    if not directory or not os.path.isdir(directory):
        return []
    out = []
    for f in os.listdir(directory):
        m = _PATTERN.match(f)
        if m:
            out.append(int(m.group(1)))
    return out


def latest_epoch(directory) -> int:
    """Highest checkpointed epoch in ``directory`` (0 when there is none)."""
    epochs = _list_epochs(directory)
    return max(epochs) if epochs else 0


def load(path, model=None, optimizer=None, scheduler=None, map_location="cpu", rng=True):
    """Restore ``model`` / ``optimizer`` / ``scheduler`` (each optional) from ``path`` and, with
    ``rng=True``, the random generators the checkpoint recorded.  Safe loader only."""
    payload = torch.load(path, map_location=map_location, weights_only=True)
    if model is not None:
        own = model.state_dict()
        with torch.no_grad():
            for k, v in payload["model"].items():
                if k not in own:
                    raise KeyError("checkpoint entry %r is not in the model" % k)
                own[k].copy_(v.reshape(own[k].shape).to(own[k].dtype))
        missing = set(own) - set(payload["model"])
        if missing:
            raise KeyError("model entries missing from checkpoint: %s" % sorted(missing)[:5])
    if optimizer is not None and "optimizer" in payload:
        optimizer.load_state_dict(payload["optimizer"])
    if optimizer is not None and model is not None:
        # an optimizer with fp32 master weights (bf16 compute) takes them from the checkpoint's fp32
        # model entries: also when the file carries no optimizer state (fine-tuning from weights
        # only), and bit-exactly when it does (the entries ARE the saved masters)
        masters = getattr(optimizer, "load_master_from_state", None)
        if callable(masters):
            masters(payload["model"], model)
    if scheduler is not None and "scheduler" in payload:
        scheduler.load_state_dict(payload["scheduler"])
    if model is not None:
        try:  # parameters were rewritten: derived copies (transposed filters of the HIP path) are stale
            from ..ops import hip_ops

            hip_ops.weights_changed()
        except Exception:  # pragma: no cover - ops layer unavailable
            pass
    if rng:
        restore_rng(payload.get("rng"))
    return payload


def resume(directory, model, optimizer=None, scheduler=None, root_rank=0) -> int:
    """Restore the latest checkpoint on every rank; returns the epoch to continue from."""
    if not _b.is_initialized():
        _b.init()
    is_root = _b.rank() == root_rank
    epoch = latest_epoch(directory) if (is_root and directory) else 0
    epoch = int(_c.broadcast_object(epoch, root_rank, name="resume_from_epoch"))
    if epoch == 0:
        return 0
    if is_root:
        load(checkpoint_path(directory, epoch), model, optimizer, scheduler)
    _f.broadcast_parameters(model.state_dict(), root_rank)
    if optimizer is not None:
        _f.broadcast_optimizer_state(optimizer, root_rank)
        sync = getattr(optimizer, "sync_master_from_root", None)
        if callable(sync):
            sync(root_rank)
    return epoch
