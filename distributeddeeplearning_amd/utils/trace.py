"""roctx ranges for rocprofv3 timelines: ``with trace.range("fwd"): ...``.

Uses ``librocprofiler-sdk-roctx`` when present (ROCm 7), otherwise the ranges are no-ops.
Enable with ``DDL_ROCTX=1``; disabled by default so the hot loop pays nothing.
"""
from __future__ import annotations

import contextlib
import ctypes
import os

__all__ = ["range", "mark", "enabled"]

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    if os.environ.get("DDL_ROCTX", "0") not in ("1", "true", "True"):
        return None
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            _lib = ctypes.CDLL(name)
            break
        except OSError:
            continue
    return _lib


def enabled() -> bool:
    return _load() is not None


class _Range:
    __slots__ = ("lib", "name")

    def __init__(self, lib, name):
        self.lib, self.name = lib, name

    def __enter__(self):
        self.lib.roctxRangePushA(self.name)
        return self

    def __exit__(self, *exc):
        self.lib.roctxRangePop()
        return False


_NULL = contextlib.nullcontext()


def range(name: str):  # noqa: A001 - mirrors roctxRange
    """Context manager: a roctx range around the body (a shared no-op object when disabled, so the
    training loops call it unconditionally)."""
    lib = _load()
    if lib is None:
        return _NULL
    return _Range(lib, name.encode())


def mark(name: str) -> None:
    lib = _load()
    if lib is not None:
        lib.roctxMarkA(name.encode())
