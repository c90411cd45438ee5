"""Loader for the in-tree native libraries.

Two shared objects are built in-tree by ``build.py`` (driven by ``__graft_entry__.build``):

* ``_ddl_kernels.so`` — the hand-written gfx950 HIP kernels (``csrc/kernels/*.hip``), compiled
  by ``hipcc --offload-arch=gfx950`` and exposed as plain ``extern "C"`` launchers that take
  device pointers and a ``hipStream_t``.  They are called through ``ctypes`` with
  ``tensor.data_ptr()`` and the current PyTorch HIP stream handle: no torch headers, no
  hipify step, no ABI coupling to the torch build.
* ``_ddl_engine.so`` — the host-side C++ runtime (bucket planner, TCP rendezvous store,
  timeline writer, watchdog; ``csrc/engine/*.cpp``).

On a machine with a GPU the kernel library is mandatory: ``kernels_lib()`` raises if it is
missing rather than letting a silent eager fallback masquerade as the native path.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
_lock = threading.Lock()
_cache: dict[str, object] = {}

# DDL_KERNELS_SO: A/B a differently built kernel library (e.g. `hipcc -DDDL_EPI_BATCH=8 ...`) in one GPU session
KERNELS_SO = os.environ.get("DDL_KERNELS_SO") or os.path.join(_PKG, "_ddl_kernels.so")
ENGINE_SO = os.path.join(_PKG, "_ddl_engine.so")


class NativeLibraryMissing(RuntimeError):
    pass


def _load(path):
    with _lock:
        if path in _cache:
            return _cache[path]
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL) if os.path.isfile(path) else None
        _cache[path] = lib
        return lib


def kernels_available() -> bool:
    return os.path.isfile(KERNELS_SO)


def kernels_lib():
    """The HIP kernel library; raises NativeLibraryMissing when it has not been built."""
    import torch  # noqa: F401  (libamdhip64 must already be mapped by torch before dlopen)

    lib = _load(KERNELS_SO)
    if lib is None:
        raise NativeLibraryMissing(
            "%s not found: run `python build.py` (or __graft_entry__.build()) first" % KERNELS_SO)
    return lib


def engine_lib():
    """The host runtime library, or None when it has not been built."""
    return _load(ENGINE_SO)
