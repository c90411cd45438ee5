"""Python face of the launch tape (``csrc/kernels/tape.h``): record the device work of a training
step once, replay it from native code (see :mod:`distributeddeeplearning_amd.training.tape`).

Two kinds of step content exist:

* launches of the in-tree kernels, stream-ordering calls and memsets - recorded by the library
  itself while a recording is open;
* *live actions* - Python that has to run again at every replay because it talks to something the
  tape cannot hold (the communication engine's collectives, their completion waits, metric
  events).  Code wraps them in :func:`live`: during a recording the action runs with the tape
  paused, the tape is cut there, and the replay loop calls the same closure between the two
  segments.  Outside a recording ``live(fn)`` is just ``fn()``.
"""
from __future__ import annotations

import threading

from . import kernels as K

__all__ = ["recording", "live", "Recording", "replay", "free", "TapeError"]


class TapeError(RuntimeError):
    pass


_state = {"open": None}
_lock = threading.RLock()


def recording() -> bool:
    return _state["open"] is not None


def live(fn):
    """Run ``fn()`` now; when a recording is open, also make it part of every replay."""
    rec = _state["open"]
    if rec is None:
        return fn()
    with _lock:
        lib = K.lib()
        lib.ddl_tape_pause()
        try:
            out = fn()
        finally:
            lib.ddl_tape_resume()
        nseg = lib.ddl_tape_mark()          # the action sits between segment nseg-1 and segment nseg
        rec.actions.append((nseg, fn))
        return out


class Recording:
    """``with Recording() as rec: <one training step>`` -> ``rec.tape`` (id), ``rec.actions``."""

    def __init__(self):
        self.actions = []
        self.tape = None
        self.segments = 0
        self.calls = 0

    def __enter__(self):
        if _state["open"] is not None:
            raise TapeError("a tape recording is already open")
        if K.lib().ddl_tape_begin() != 0:
            raise TapeError("the kernel library refused to open a recording")
        _state["open"] = self
        return self

    def __exit__(self, exc_type, exc, tb):
        _state["open"] = None
        lib = K.lib()
        if exc_type is not None:
            lib.ddl_tape_abort()
            return False
        self.tape = lib.ddl_tape_end()
        if self.tape < 0:
            raise TapeError("no recording to close")
        self.segments = lib.ddl_tape_segments(self.tape)
        lib.ddl_tape_calls.restype = __import__("ctypes").c_longlong
        self.calls = int(lib.ddl_tape_calls(self.tape))
        return False


def replay(tape, actions, segments):
    """Re-issue ``tape``: segment, live action, segment, ... in recorded order."""
    lib = K.lib()
    lo = 0
    for at, fn in actions:
        if at > lo:
            K.check(lib.ddl_tape_replay(tape, lo, at), "tape replay")
            lo = at
        fn()
    if segments > lo:
        K.check(lib.ddl_tape_replay(tape, lo, segments), "tape replay")


def free(tape):
    if tape is not None and tape >= 0:
        K.lib().ddl_tape_free(tape)
