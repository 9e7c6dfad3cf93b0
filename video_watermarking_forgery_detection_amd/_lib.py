"""ctypes loader for libwm_hip.so (the C ABI declared in include/wm_hip.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError
is raised.  Nothing under oracle/ is ever imported from here.

Two builds of the same sources live in lib/: libwm_hip.so (release: no knobs, no environment variables, no state between
calls) is what every product path loads; libwm_hip_dbg.so (-DWM_DEBUG: the wm_debug_* A/B switches) is loaded only on
request, by tools/ and by the tests that compare a fused kernel with its unfused form (use_debug_library()).
"""
import contextlib
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libwm_hip.so")
# tools/ only (same-box A/B of two BUILDS, tools/ab_libs.sh): WM_LIB_VARIANT=<name> loads tools/micro/ab/libwm_hip_<name>.so in place of the
# release file -- the release file is never overwritten, so an interrupted A/B cannot leave a variant installed.  loaded_path() says which
_VARIANT = os.environ.get("WM_LIB_VARIANT")
if _VARIANT:
    LIB_PATH = os.path.join(os.path.dirname(_PKG), "tools", "micro", "ab", f"libwm_hip_{_VARIANT}.so")
DEBUG_LIB_PATH = os.path.join(_PKG, "lib", "libwm_hip_dbg.so")
_lib = None
_release = None
_debug = None


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -m video_watermarking_forgery_detection_amd.build` "
            "(there is no CPU / PyTorch fallback for the HIP path)")
    h = ctypes.CDLL(path)
    h.wm_last_error_string.restype = ctypes.c_char_p
    return h


def lib():
    """the library every op goes through: the release build unless a use_debug_library() block is open"""
    global _lib, _release
    if _lib is None:
        if _release is None:
            _release = _load(LIB_PATH)
        _lib = _release
    return _lib


def loaded_path():
    """path of the library lib() loads (bench.py reports its hash next to the kernel sources' hash)"""
    return LIB_PATH


def debug_lib():
    global _debug
    if _debug is None:
        _debug = _load(DEBUG_LIB_PATH)
    return _debug


@contextlib.contextmanager
def use_debug_library():
    """inside the block every op runs on the -DWM_DEBUG build, whose wm_debug_* switches the caller may flip; yields its handle"""
    global _lib
    prev = lib()
    _lib = debug_lib()
    try:
        yield _lib
    finally:
        _lib = prev


def check(rc, name):
    if rc != 0:
        msg = lib().wm_last_error_string().decode(errors="replace")
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


def ptr(t):
    """raw device pointer of a tensor (or NULL for None)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_of(t=None):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
