"""Wall-clock ``Timer`` and ``timer`` decorator.

API parity target: reference ``common/timer.py:7-105`` (``Timer.start/stop/elapsed``, context
manager that emits ``prefix + fmt.format(elapsed)``, ``__str__`` with 3 decimals, pluggable
clock and scale ``factor``; ``timer`` decorator usable bare or with arguments).  The reference
checks ``collections.Callable`` (removed in Python 3.10); this
implementation uses ``callable``.

MI355X addition: ``DeviceTimer`` brackets a region with HIP events on the current stream so the
benchmark harness can report device time next to wall time.
"""
from __future__ import annotations

import functools
import logging
from timeit import default_timer

__all__ = ["Timer", "timer", "DeviceTimer"]


class Timer:
    """Wall-clock stopwatch.

    Keyword arguments
    -----------------
    timer:   zero-argument clock returning seconds (default ``timeit.default_timer``)
    factor:  multiplier applied to the elapsed value (1000 gives milliseconds)
    output:  ``True`` -> print on context exit; a callable -> call it with the message
    fmt:     ``str.format`` template receiving the elapsed value
    prefix:  text placed (plus a space) in front of the message.  Giving only a prefix implies
             ``output=True``.
    """

    def __init__(self, timer=default_timer, factor=1, output=None,
                 fmt="took {:.3f} seconds", prefix=""):
        self._clock = timer
        self._factor = factor
        self._output = output
        self._fmt = fmt
        self._prefix = prefix
        self._start = None
        self._end = None

    def __call__(self):
        """Current reading of the underlying clock."""
        return self._clock()

    def start(self):
        self._start = self()
        self._end = None

    def stop(self):
        self._end = self()

    @property
    def elapsed(self):
        """Live value while running, frozen value once stopped."""
        if self._start is None:
            raise RuntimeError("Timer.elapsed read before Timer.start()")
        upto = self() if self._end is None else self._end
        return (upto - self._start) * self._factor

    def __enter__(self):
        self.start()
        return self

    def __exit__(self, exc_type, exc_value, exc_traceback):
        self.stop()
        sink = self._output
        if sink is True or (sink is None and self._prefix):
            sink = print
        if callable(sink):
            sink(" ".join([self._prefix, self._fmt.format(self.elapsed)]))
        return False

    def __str__(self):
        return "%.3f" % self.elapsed


def timer(*func_or_none, logger=None, level=logging.INFO,
          fmt="function %(function_name)s execution time: %(execution_time).3f",
          **timer_kwargs):
    """Decorator logging a function's execution time.

    ``@timer`` and ``@timer(logger=log, level=logging.DEBUG, factor=1000)`` both work.  With a
    logger the record carries ``function_name``/``execution_time`` in ``extra``; without one
    the line is printed.
    """

    def decorate(fn):
        @functools.wraps(fn)
        def wrapped(*args, **kwargs):
            with Timer(**timer_kwargs) as t:
                result = fn(*args, **kwargs)
            ctx = {"function_name": fn.__name__, "execution_time": t.elapsed}
            if logger is not None:
                logger.log(level, fmt % ctx, extra=ctx)
            else:
                print(fmt % ctx)
            return result

        return wrapped

    if len(func_or_none) == 1 and callable(func_or_none[0]):
        return decorate(func_or_none[0])
    return decorate


class DeviceTimer:
    """HIP-event stopwatch on the current stream (milliseconds of device time).

    ``elapsed_ms`` synchronises on the end event only, so it can be read after the timed region
    without a device-wide sync inside it.
    """

    def __init__(self):
        import torch

        self._torch = torch
        self._a = torch.cuda.Event(enable_timing=True)
        self._b = torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        self._a.record()
        return self

    def __exit__(self, *exc):
        self._b.record()
        return False

    @property
    def elapsed_ms(self):
        self._b.synchronize()
        return self._a.elapsed_time(self._b)
