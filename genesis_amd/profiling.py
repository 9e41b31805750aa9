"""Live per-kernel profiling through the C ABI (gx_profile_*): HIP events bracket every kernel launch on
its launch stream; each launch carries its algorithmic flops / bytes."""
import ctypes

from . import _lib


def enable(on=True, ctx=None):
    """Profiling records belong to a library context (gx_ctx_*): ctx = a TrainStep's context id (ts._ctx) to profile its
    iterations, None = the calling thread's current context."""
    if ctx is None:
        _lib.call('gx_profile_enable', int(bool(on)))
        return
    prev = _lib.current_ctx()
    _lib.make_current(ctx)
    try:
        _lib.call('gx_profile_enable', int(bool(on)))
    finally:
        _lib.make_current(prev)


def collect(ctx=None):
    """-> list of dict(name, ms, launches, flops, bytes) for kernels launched (in context `ctx`) since the last collect."""
    if ctx is not None:
        prev = _lib.current_ctx()
        _lib.make_current(ctx)
        try:
            return collect()
        finally:
            _lib.make_current(prev)
    lib = _lib.load()
    n = lib.gx_profile_num_kernels()
    arr = lambda: (ctypes.c_double * n)()  # noqa: E731
    ms, cnt, fl, by = arr(), arr(), arr(), arr()
    _lib.call('gx_profile_collect', ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p),
              ctypes.cast(fl, ctypes.c_void_p), ctypes.cast(by, ctypes.c_void_p))
    rows = []
    for k in range(n):
        if cnt[k] > 0:
            rows.append(dict(name=lib.gx_profile_kernel_name(k).decode(), ms=ms[k], launches=int(cnt[k]),
                             flops=fl[k], bytes=by[k]))
    return rows
