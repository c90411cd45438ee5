"""Chrome-trace timeline of collective activity (Horovod-timeline compatible idea).

The reference never enables ``HOROVOD_TIMELINE`` but Horovod's core ships it
[EXT]; here ``HVD_TIMELINE=/path/trace.json`` (or ``HOROVOD_TIMELINE``) records one complete
("X") event per named collective / gradient bucket with enqueue → completion-wait times, plus
instant markers, loadable in chrome://tracing or Perfetto.  Each rank writes its own file
(``.rankN`` suffix for N>0).  When the native engine library is present the records are
written by its C++ writer thread (``csrc/engine/timeline.cpp``); otherwise a pure-Python
writer with the same format is used.
"""
from __future__ import annotations

import atexit
import json
import os
import threading
import time

__all__ = ["start", "stop", "begin", "end", "instant", "enabled"]

_lock = threading.Lock()
_fh = None
_open_events: dict[str, tuple[float, str]] = {}
_first = True
_pid = 0
_native = None


def enabled() -> bool:
    return _fh is not None or _native is not None


def _now_us() -> float:
    return time.monotonic() * 1e6


def start(path: str, rank: int = 0) -> None:
    """Begin recording to ``path`` (idempotent)."""
    global _fh, _first, _pid, _native
    with _lock:
        if enabled():
            return
        if rank:
            path = "%s.rank%d" % (path, rank)
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        _pid = rank
        try:
            from ..ops import native

            lib = native.engine_lib()
            if lib is not None and lib.ddl_timeline_start(path.encode(), rank) == 0:
                _native = lib
        except Exception:
            _native = None
        if _native is None:
            _fh = open(path, "w")
            _fh.write("[\n")
            _first = True
    atexit.register(stop)


def _emit(ev: dict) -> None:
    global _first
    if _fh is None:
        return
    _fh.write(("" if _first else ",\n") + json.dumps(ev))
    _first = False


def begin(name: str, category: str = "ALLREDUCE") -> None:
    if not enabled():
        return
    with _lock:
        _open_events[name] = (_now_us(), category)


def end(name: str, args: dict | None = None) -> None:
    if not enabled():
        return
    with _lock:
        t0_cat = _open_events.pop(name, None)
        if t0_cat is None:
            return
        t0, cat = t0_cat
        dur = _now_us() - t0
        if _native is not None:
            _native.ddl_timeline_event(name.encode(), cat.encode(), t0, dur)
        else:
            _emit({"name": name, "cat": cat, "ph": "X", "ts": t0, "dur": dur,
                   "pid": _pid, "tid": 0, "args": args or {}})


def instant(name: str, category: str = "MARK") -> None:
    if not enabled():
        return
    with _lock:
        if _native is not None:
            _native.ddl_timeline_event(name.encode(), category.encode(), _now_us(), -1.0)
        else:
            _emit({"name": name, "cat": category, "ph": "i", "ts": _now_us(), "pid": _pid,
                   "tid": 0, "s": "p"})


def stop() -> None:
    global _fh, _native
    with _lock:
        if _native is not None:
            _native.ddl_timeline_stop()
            _native = None
        if _fh is not None:
            _fh.write("\n]\n")
            _fh.close()
            _fh = None
        _open_events.clear()
