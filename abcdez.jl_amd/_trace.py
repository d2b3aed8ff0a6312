"""roctx ranges around the phases of a generation (the reference has no tracing; the plan is rocprofv3
markers) -- prologue, group of sweeps, own-range sweep, each collective, replay, resampling -- so that a sharded generation on
real xGMI can be attributed phase by phase against DESIGN.md section 7's tables:

    rocprofv3 --kernel-trace --marker-trace -- python3 bench.py ...

Zero cost unless a profiler is attached: the ranges are no-ops (one attribute test) unless ABZ_ROCTX=1 is set or this process
runs under rocprofv3 (its tool library is preloaded); the library is loaded lazily and only then.
"""
from __future__ import annotations

import ctypes as C
import os

_push = _pop = None
_ENABLED = None


def _want() -> bool:
    if os.environ.get("ABZ_ROCTX") is not None:
        return os.environ["ABZ_ROCTX"] not in ("", "0")
    return "rocprofiler-sdk-tool" in os.environ.get("LD_PRELOAD", "") or "ROCPROF_OUTPUT_PATH" in os.environ \
        or any(k.startswith("ROCPROF_") for k in os.environ)


def enabled() -> bool:
    global _ENABLED, _push, _pop
    if _ENABLED is None:
        _ENABLED = False
        if _want():
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = C.CDLL(name)
                    lib.roctxRangePushA.argtypes = [C.c_char_p]
                    lib.roctxRangePushA.restype = C.c_int
                    lib.roctxRangePop.restype = C.c_int
                    _push, _pop = lib.roctxRangePushA, lib.roctxRangePop
                    _ENABLED = True
                    break
                except (OSError, AttributeError):
                    continue
    return _ENABLED


class _Range:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        _push(self.name)
        return self

    def __exit__(self, *exc):
        _pop()
        return False


class _Null:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _Null()


def rng(name: str):
    """context manager: a roctx range named `name` (b"abcdez:" prefix) when tracing is on, a shared no-op otherwise"""
    if _ENABLED is False:
        return _NULL
    if not enabled():
        return _NULL
    return _Range(b"abcdez:" + name.encode())
