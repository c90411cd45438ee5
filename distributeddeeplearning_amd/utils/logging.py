"""Rank- and epoch-aware logger.

Parity target: ``HorovodAdapter`` / ``_get_logger`` in the reference
(``HorovodPytorch/src/imagenet_pytorch_horovod.py:70-95``, duplicated in the Keras and TF
scripts).  Same record format ``"%(levelname)s:%(name)s:%(gpurank)d: %(epoch)s %(message)s"``,
same ``[Epoch n]`` tag, stdout handler, cached singleton, and every rank logs (no rank-0
filter) so that logs are diff-able against the reference's.

New here: ``MetricsWriter`` appends machine-readable ``metrics.jsonl`` records.
"""
from __future__ import annotations

import json
import logging
import os
import sys
import time

__all__ = ["RankAdapter", "get_logger", "MetricsWriter", "LOG_FORMAT"]

LOG_FORMAT = "%(levelname)s:%(name)s:%(gpurank)d: %(epoch)s %(message)s"


def _current_rank() -> int:
    """Rank if the distributed layer is initialised, else 0 (reference ``_get_rank`` :59-66)."""
    try:
        from ..parallel import backend

        return backend.rank() if backend.is_initialized() else int(os.environ.get("RANK", 0))
    except Exception:  # pragma: no cover - defensive: logging must never raise
        return 0


class RankAdapter(logging.LoggerAdapter):
    """Injects ``gpurank`` and ``epoch`` into every record."""

    def __init__(self, logger, extra=None):
        self._epoch_tag = ""
        super().__init__(logger, extra or {})

    def set_epoch(self, epoch):
        self._epoch_tag = "[Epoch {}]".format(epoch)

    def process(self, msg, kwargs):
        extra = dict(kwargs.get("extra") or {})
        extra.setdefault("gpurank", _current_rank())
        extra.setdefault("epoch", self._epoch_tag)
        kwargs["extra"] = extra
        return msg, kwargs


_CACHE = {}


def get_logger(name="__main__", level=None, stream=None) -> RankAdapter:
    """Cached adapter writing the reference format to stdout."""
    if name in _CACHE:
        return _CACHE[name]
    base = logging.getLogger(name)
    if level is None:
        level = getattr(logging, os.environ.get("LOG_LEVEL", "INFO").upper(), logging.INFO)
    base.setLevel(level)
    base.propagate = False
    if not any(getattr(h, "_ddl_handler", False) for h in base.handlers):
        handler = logging.StreamHandler(stream or sys.stdout)
        handler.setFormatter(logging.Formatter(LOG_FORMAT))
        handler._ddl_handler = True
        base.addHandler(handler)
    adapter = RankAdapter(base)
    _CACHE[name] = adapter
    return adapter


class MetricsWriter:
    """Append-only JSON-lines metrics sink (one file per rank unless ``rank0_only``)."""

    def __init__(self, path, rank=None, rank0_only=True):
        self.rank = _current_rank() if rank is None else rank
        self.enabled = bool(path) and (not rank0_only or self.rank == 0)
        self._fh = None
        if self.enabled:
            os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
            self._fh = open(path, "a", buffering=1)

    def write(self, **record):
        if self._fh is None:
            return
        record.setdefault("ts", time.time())
        record.setdefault("rank", self.rank)
        self._fh.write(json.dumps(record, sort_keys=True) + "\n")

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
