"""Thin Python bindings over the C ABI (include/wm_hip.h): argument checking,
output allocation by torch (device memory + streams are torch's job), raw
pointers into libwm_hip.so.  Every function here launches HIP kernels; none has
a CPU or PyTorch fallback.
"""
import ctypes

import torch

from . import _lib

c_int = ctypes.c_int
c_float = ctypes.c_float
c_size_t = ctypes.c_size_t
c_double = ctypes.c_double

WM_F32, WM_BF16 = 0, 1
JPEG_ROUND, JPEG_SS, JPEG_MASK = 0, 1, 2


def dtype_id(t):
    if t.dtype == torch.float32:
        return WM_F32
    if t.dtype == torch.bfloat16:
        return WM_BF16
    raise TypeError(f"unsupported activation dtype {t.dtype} (float32 or bfloat16)")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("wm ops run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _host_floats(vals):
    arr = (ctypes.c_float * len(vals))(*[float(v) for v in vals])
    return arr


# ----------------------------------------------------------------------------- block JPEG
def jpeg_fwd(x, mode, tables, subsample=0):
    """x [B,3,H,W] f32 cuda; tables: 128 python floats (lum, chroma) or None for mask."""
    _need_cuda(x)
    assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty_like(x)
    B, _, H, W = x.shape
    tb = _host_floats(tables) if tables is not None else None
    rc = _lib.lib().wm_jpeg_fwd(_p(x), _p(y), c_int(B), c_int(H), c_int(W), c_int(mode), tb, c_int(subsample), _stream())
    _lib.check(rc, "wm_jpeg_fwd")
    return y


def jpeg_bwd(x, gy, mode, tables, subsample=0):
    _need_cuda(gy)
    gy = gy.contiguous()
    gx = torch.empty_like(gy)
    B, _, H, W = gy.shape
    tb = _host_floats(tables) if tables is not None else None
    xx = x.contiguous() if x is not None else None
    rc = _lib.lib().wm_jpeg_bwd(_p(xx), _p(gy), _p(gx), c_int(B), c_int(H), c_int(W), c_int(mode), tb, c_int(subsample), _stream())
    _lib.check(rc, "wm_jpeg_bwd")
    return gx
