"""ctypes binding of the C-ABI library (include/sarssl_hip.h).

The HIP library is the product: there is no CPU/PyTorch fallback.  Importing this module never
touches the GPU; calling any kernel without the built library (or on CPU tensors) raises.
"""
import ctypes
import os

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
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise SarsslHipError("%s failed (rc=%d): %s" % (what, rc, lib().sarssl_last_error().decode()))


ncalls = 0          # C-ABI calls made so far (launch accounting: tools/host_time.py, graph.py's empty-segment check)


call_timer = None   # hip.profile_start(all_calls=True): context-manager factory(name) bracketing every C call with events


def call(name, *args):
    global ncalls
    ncalls += 1
    fn = getattr(lib(), name)
    if call_timer is not None:
        with call_timer(name):
            rc = fn(*args)
    else:
        rc = fn(*args)
    check(rc, name)


c_void_p, c_int, c_long, c_float, c_ulonglong, c_double = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_ulonglong, ctypes.c_double)
