"""ctypes binding of the C-ABI library (include/sarssl_hip.h).

The HIP library is the product: there is no CPU/PyTorch fallback.  Importing this module never
touches the GPU; calling any kernel without the built library (or on CPU tensors) raises.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  -- must be imported (and its HIP runtime loaded) BEFORE libsarssl_hip.so so both share one runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SARSSL_HIP_LIB") or os.path.join(_HERE, "csrc", "libsarssl_hip.so")   # override: A/B runs of two builds
_lib = None


class SarsslHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SarsslHipError(
                "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the product path)" % LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.sarssl_last_error.restype = ctypes.c_char_p
        _lib.sarssl_create.restype = ctypes.c_void_p
    return _lib


# ---- contexts (include/sarssl_hip.h): the library has no global mutable state - what a caller configures (gradient-convolution
# workgroup count, clock probe, attached step state, zeroed arena) lives in a sarssl_ctx, and kernels run under the context that is
# current on the calling THREAD.  This module keeps one context per (process, device) and makes it current on every thread that issues
# calls for that device (the training thread, autograd's backward thread, loader threads); `use_ctx` switches to another one (tests).
_ctxs = {}                       # device index -> sarssl_ctx* (int)
_ctx_lock = threading.Lock()
_tls = threading.local()         # .key = (device, ctx) the thread last made current


def ctx(device=None):
    """The process's context of `device` (default: torch's current device), created on first use."""
    if device is None:
        device = torch.cuda.current_device()
    c = _ctxs.get(device)
    if c is None:
        with _ctx_lock:
            c = _ctxs.get(device)
            if c is None:
                c = lib().sarssl_create(ctypes.c_int(device))
                if not c:
                    raise SarsslHipError("sarssl_create(%d): %s" % (device, lib().sarssl_last_error().decode()))
                _ctxs[device] = c
    return c


def _make_current():
    dev = torch.cuda.current_device()
    c = getattr(_tls, "override", None) or ctx(dev)
    if getattr(_tls, "key", None) != (dev, c):
        lib().sarssl_make_current(ctypes.c_void_p(c))
        _tls.key = (dev, c)
    return c


class use_ctx:
    """``with use_ctx(c):`` - calls issued by this thread inside run under context ``c`` (a handle from ``new_ctx``)."""

    def __init__(self, c):
        self.c = c

    def __enter__(self):
        self.prev = getattr(_tls, "override", None)
        _tls.override = self.c
        return self.c

    def __exit__(self, *exc):
        _tls.override = self.prev
        return False


def new_ctx(device=None):
    c = lib().sarssl_create(ctypes.c_int(torch.cuda.current_device() if device is None else device))
    if not c:
        raise SarsslHipError("sarssl_create: %s" % lib().sarssl_last_error().decode())
    return c


def destroy_ctx(c):
    if getattr(_tls, "key", (None, None))[1] == c:
        _tls.key = None
    lib().sarssl_destroy(ctypes.c_void_p(c))


def check(rc, what=""):
    if rc != 0:
        raise SarsslHipError("%s failed (rc=%d): %s" % (what, rc, lib().sarssl_last_error().decode()))


ncalls = 0          # C-ABI calls made so far (launch accounting: tools/host_time.py, graph.py's empty-segment check)


call_timer = None   # hip.profile_start(all_calls=True): context-manager factory(name) bracketing every C call with events


_HAS_GPU = None


def call(name, *args):
    global ncalls, _HAS_GPU
    ncalls += 1
    fn = getattr(lib(), name)
    if _HAS_GPU is None:
        _HAS_GPU = torch.cuda.is_available()
    if _HAS_GPU:
        _make_current()                 # (a dictionary lookup when nothing changed: the thread already runs under its device's context)
    if call_timer is not None:
        with call_timer(name):
            rc = fn(*args)
    else:
        rc = fn(*args)
    check(rc, name)


c_void_p, c_int, c_long, c_float, c_ulonglong, c_double = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_ulonglong, ctypes.c_double)
