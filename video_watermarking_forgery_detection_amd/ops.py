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

WM_F32, WM_BF16, WM_F16 = 0, 1, 2
_DT = {torch.float32: WM_F32, torch.bfloat16: WM_BF16, torch.float16: WM_F16}


def dt_id(dtype):
    """torch dtype -> WM_F32 / WM_BF16 / WM_F16"""
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"unsupported activation dtype {dtype} (float32, bfloat16 or float16)") from None
JPEG_ROUND, JPEG_SS, JPEG_MASK = 0, 1, 2


def dtype_id(t):
    return dt_id(t.dtype)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("wm ops run on the GPU only (tensor is on %s); there is no CPU fallback" % t.device)


def _wrote(*ts):
    """a kernel of this library wrote these tensors in place through their raw pointers: bump torch's version counter of each, so that
    every check built on `_version` (engine.cbr_backward's "is this still the tensor my consumer produced", PackPlan's stale-pack test,
    autograd's saved-tensor check) sees the write as it sees torch's own in-place ops.  The counter is shared by a tensor and the views
    torch made of it (a flat buffer and its slices); a parameter whose `.data` was pointed at such a view keeps a counter of its own."""
    for t in ts:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _host_floats(vals):
    arr = (ctypes.c_float * len(vals))(*[float(v) for v in vals])
    return arr


# ----------------------------------------------------------------------------- block JPEG
def jpeg_fwd(x, mode, tables, subsample=0, act16_dtype=None):
    """x [B,3,H,W] f32 cuda; tables: 128 python floats (lum, chroma) or None for mask.  act16_dtype: -> (y, the same image as the
    [B,H,W,16] zero-padded NHWC tensor of that dtype: engine.image_to_act's result without its launch)"""
    _need_cuda(x)
    assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty_like(x)
    B, _, H, W = x.shape
    a16 = torch.empty(B, H, W, 16, device=x.device, dtype=act16_dtype) if act16_dtype is not None else None
    tb = _host_floats(tables) if tables is not None else None
    rc = _timed("jpeg_fwd", None, lambda: _lib.lib().wm_jpeg_fwd_act(_p(x), _p(y), _p(a16), c_int(dt_id(act16_dtype) if a16 is not None else WM_F32),
                                                                     c_int(B), c_int(H), c_int(W), c_int(mode), tb, c_int(subsample), _stream()))
    _lib.check(rc, "wm_jpeg_fwd")
    return (y, a16) if act16_dtype is not None else y


def jpeg_bwd(x, gy, mode, tables, subsample=0):
    _need_cuda(gy)
    gy = gy.contiguous()
    gx = torch.empty_like(gy)
    B, _, H, W = gy.shape
    tb = _host_floats(tables) if tables is not None else None
    xx = x.contiguous() if x is not None else None
    rc = _timed("jpeg_bwd", None, lambda: _lib.lib().wm_jpeg_bwd(_p(xx), _p(gy), _p(gx), c_int(B), c_int(H), c_int(W), c_int(mode), tb,
                                                                 c_int(subsample), _stream()))
    _lib.check(rc, "wm_jpeg_bwd")
    return gx


# ----------------------------------------------------------------------------- layout
def _host_ints(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def nchw_to_nhwc(x, out, C_off=0, zero_tail=0):
    """x [B,C,H,W] f32 -> out[B,H,W,ld] (bf16|f32) channels [C_off, C_off+C), zero tail after."""
    _need_cuda(x, out)
    x = x.contiguous()
    B, C, H, W = x.shape
    rc = _lib.lib().wm_nchw_to_nhwc(_p(x), _p(out), c_int(B), c_int(C), c_int(H), c_int(W), c_int(out.shape[-1]),
                                    c_int(C_off), c_int(zero_tail), c_int(dtype_id(out)), _stream())
    _lib.check(rc, "wm_nchw_to_nhwc")
    return out


def u8_hwc_to_planes(x, scale=1.0 / 255.0):
    """decoded frames uint8 [N,H,W,C] (cuda) -> float32 planes [N,C,H,W] * scale"""
    _need_cuda(x)
    assert x.dtype == torch.uint8 and x.dim() == 4 and x.shape[-1] <= 4
    x = x.contiguous()
    N, H, W, C = x.shape
    y = torch.empty(N, C, H, W, device=x.device, dtype=torch.float32)
    rc = _lib.lib().wm_u8_hwc_to_planes(_p(x), _p(y), c_int(N), c_int(H), c_int(W), c_int(C), c_float(scale), _stream())
    _lib.check(rc, "wm_u8_hwc_to_planes")
    return y


def nhwc_to_nchw(x, C, C_off=0):
    _need_cuda(x)
    B, H, W, ld = x.shape
    y = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    rc = _lib.lib().wm_nhwc_to_nchw(_p(x), _p(y), c_int(B), c_int(C), c_int(H), c_int(W), c_int(ld), c_int(C_off),
                                    c_int(dtype_id(x)), _stream())
    _lib.check(rc, "wm_nhwc_to_nchw")
    return y


def broadcast_to_nhwc(v, out, C_off):
    _need_cuda(v, out)
    v = v.contiguous().float()
    B, H, W, ld = out.shape
    rc = _lib.lib().wm_broadcast_to_nhwc(_p(v), _p(out), c_int(B), c_int(v.shape[1]), c_int(H), c_int(W), c_int(ld),
                                         c_int(C_off), c_int(dtype_id(out)), _stream())
    _lib.check(rc, "wm_broadcast_to_nhwc")
    return out


def concat_tail(msg, img, out, C_off):
    """out[..., C_off:] = [msg | img | 0]  (encoder concat tail, one vectorised pass)"""
    _need_cuda(msg, img, out)
    B, H, W, ld = out.shape
    msg = msg.contiguous().float(); img = img.contiguous().float()
    rc = _lib.lib().wm_concat_tail(_p(msg), _p(img), _p(out), c_int(B), c_int(msg.shape[1]), c_int(H), c_int(W), c_int(ld),
                                   c_int(C_off), c_int(ld - C_off), c_int(dtype_id(out)), _stream())
    _lib.check(rc, "wm_concat_tail")
    return out


def concat_full(x, scale, shift, msg, img, out, C):
    """out = [relu(scale*x+shift)[:C] | msg | img | 0]  (the encoder concat row, one pass, full-line writes)"""
    _need_cuda(x, msg, img, out)
    B, H, W, ld = out.shape
    msg = msg.contiguous().float(); img = img.contiguous().float()
    rc = _lib.lib().wm_concat_full(_p(x), c_int(x.shape[-1]), _p(scale), _p(shift), _p(msg), _p(img), _p(out), c_int(B), c_int(C),
                                   c_int(msg.shape[1]), c_int(H), c_int(W), c_int(ld), c_int(dtype_id(out)), _stream())
    _lib.check(rc, "wm_concat_full")
    return out


def bnrelu_copy(x, scale, shift, out, C_off, C):
    B, H, W, ldx = x.shape
    rc = _lib.lib().wm_bnrelu_copy(_p(x), c_int(ldx), _p(scale), _p(shift), _p(out), c_int(out.shape[-1]), c_int(C_off),
                                   c_size_t(B * H * W), c_int(C), c_int(dtype_id(x)), _stream())
    _lib.check(rc, "wm_bnrelu_copy")
    return out


def pack_w3x3(w, CoutP, CinP, dtype, perm=None, transpose=False):
    """w [Cout,Cin,3,3] f32 -> [9,CoutP,CinP] (or [9,CinP,CoutP] transposed/flipped for dgrad)."""
    _need_cuda(w)
    Cout, Cin = w.shape[0], w.shape[1]
    shape = (9, CinP, CoutP) if transpose else (9, CoutP, CinP)
    wp = torch.empty(shape, device=w.device, dtype=dtype)
    pa = _host_ints(perm) if perm is not None else None
    rc = _lib.lib().wm_pack_w3x3(_p(w), _p(wp), c_int(Cout), c_int(Cin), c_int(CoutP), c_int(CinP), pa,
                                 c_int(1 if transpose else 0), c_int(dtype_id(wp)), _stream())
    _lib.check(rc, "wm_pack_w3x3")
    return wp


class PackPlan:
    """All packed conv weights of one network, refreshed by ONE launch (wm_pack_w3x3_batch).

    Requests are registered lazily (`get` packs one-off with wm_pack_w3x3 while the plan is not valid and remembers the
    request); `refresh()` (re)packs every registered request from the current parameter values and marks the plan valid;
    `invalidate()` must be called whenever the parameters may have changed (optimiser step, state_dict load)."""

    def __init__(self):
        self.req = {}        # key -> (w, CoutP, CinP, dtype, perm tuple | None, transpose)
        self.packed = {}     # key -> packed tensor (stable storage, rewritten by refresh)
        self.jobs = None
        self.njobs = 0
        self.valid = False
        self.max_elems = 0
        self._perm_dev = {}
        self.ver = {}        # key -> the weight tensor's version counter at the last refresh.  What this catches: an in-place write -- torch's or,
                             # through ops._wrote, this library's -- to the very tensor OBJECT handed to get() (or a torch view sharing its
                             # counter).  What it cannot catch: writes through another alias of the storage (`p.data` makes a fresh counter
                             # every time; a parameter pointed at a slice of a flat buffer does not share the buffer's): the owner of
                             # the parameters calls invalidate() / refresh() for those (FlatModule: optimiser step, load_state_dict,
                             # broadcast_parameters)

    @staticmethod
    def key(w, CoutP, CinP, dtype, perm, transpose):
        return (w.data_ptr(), tuple(w.shape), CoutP, CinP, dtype, None if perm is None else tuple(int(v) for v in perm), bool(transpose))

    def get(self, w, CoutP, CinP, dtype, perm=None, transpose=False):
        k = self.key(w, CoutP, CinP, dtype, perm, transpose)
        if self.valid and k in self.packed and self.jobs is not None and self.ver.get(k) == w._version:
            return self.packed[k]
        if k not in self.req:
            self.req[k] = (w, CoutP, CinP, dtype, perm, transpose)
            self.jobs = None           # the job table has to be rebuilt
        return pack_w3x3(w, CoutP, CinP, dtype, perm=perm, transpose=transpose)

    def invalidate(self):
        self.valid = False

    def _build(self):
        import struct
        recs = []
        self.max_elems = 0
        dts = set()
        for k, (w, CoutP, CinP, dtype, perm, transpose) in self.req.items():
            if k not in self.packed:
                shape = (9, CinP, CoutP) if transpose else (9, CoutP, CinP)
                self.packed[k] = torch.empty(shape, device=w.device, dtype=dtype)
            pd = 0
            if perm is not None:
                pk = tuple(int(v) for v in perm)
                if pk not in self._perm_dev:
                    self._perm_dev[pk] = torch.tensor(pk, dtype=torch.int32, device=w.device)
                pd = self._perm_dev[pk].data_ptr()
            recs.append(struct.pack("<QQQiiiiii", w.data_ptr(), self.packed[k].data_ptr(), pd, w.shape[0], w.shape[1],
                                    CoutP, CinP, 1 if transpose else 0, 0))
            self.max_elems = max(self.max_elems, 9 * CoutP * CinP)
            dts.add(dtype)
        assert len(dts) == 1, "one compute dtype per plan"
        self.dtype = dts.pop()
        dev = next(iter(self.req.values()))[0].device
        raw = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8)
        self.jobs = raw.to(dev)
        self.njobs = len(recs)

    def refresh(self):
        if not self.req:
            return
        if self.jobs is None:
            self._build()
        rc = _lib.lib().wm_pack_w3x3_batch(_p(self.jobs), c_int(self.njobs), c_size_t(self.max_elems),
                                           c_int(dt_id(self.dtype)), _stream())
        _lib.check(rc, "wm_pack_w3x3_batch")
        for k, req in self.req.items():
            self.ver[k] = req[0]._version
        self.valid = True


# ----------------------------------------------------------------------------- conv / bn
def conv3x3_nparts(B, H, W, Cin, CoutP, dtype):
    return _lib.lib().wm_conv3x3_nparts(c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(CoutP), c_int(dt_id(dtype)))


class _WmBnBwdFin(ctypes.Structure):   # include/wm_hip.h: WmBnBwdFin
    _fields_ = [("partials", ctypes.c_void_p), ("nparts", ctypes.c_int), ("C", ctypes.c_int), ("CP", ctypes.c_int), ("count", ctypes.c_double),
                ("gamma", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("invstd", ctypes.c_void_p), ("dgamma", ctypes.c_void_p),
                ("dbeta", ctypes.c_void_p), ("accumulate", ctypes.c_int), ("coef", ctypes.c_void_p)]


def fin_rider_enabled():
    return bool(_lib.lib().wm_fin_rider_enabled())


def _fin_rider(fin):
    """fin: None or dict(partials [n,2,CP], y_shape, stats [4,CP], C, gamma, dgamma, dbeta, accumulate) -- the BatchNorm-backward
    finalisation (bn_bwd_coef_raw) of ANOTHER layer, carried by this launch's slab reduction.  -> (ctypes struct or None, coef or None)"""
    if fin is None:
        return None, None
    part, stats = fin["partials"], fin["stats"]
    B, H, W, CP = fin["y_shape"]
    assert part.shape[0] <= 256 and part.shape[2] == CP and part.is_contiguous()
    coef = torch.empty(3, CP, device=part.device, dtype=torch.float32)
    st = _WmBnBwdFin(partials=part.data_ptr(), nparts=part.shape[0], C=fin["C"], CP=CP, count=float(B * H * W), gamma=fin["gamma"].data_ptr(),
                     mean=stats[2].data_ptr(), invstd=stats[3].data_ptr(), dgamma=fin["dgamma"].data_ptr(), dbeta=fin["dbeta"].data_ptr(),
                     accumulate=1 if fin["accumulate"] else 0, coef=coef.data_ptr())
    return st, coef


def _sweep(reverse):
    """sweep_reverse argument of the conv / dgrad / wgrad entry points: walk the tiles backwards (start where the producer of the
    input stopped -- its last tiles are still in the Infinity Cache)"""
    return c_int(1 if reverse else 0)


def conv3x3_fwd(x, wp, bias, in_scale, in_shift, want_stats, Cin=None, reverse=False):
    """x [B,H,W,ld]; wp [9,CoutP,Cin]; returns y [B,H,W,CoutP] and stat partials (or None)."""
    _need_cuda(x, wp)
    B, H, W, ldx = x.shape
    CoutP, CinW = wp.shape[1], wp.shape[2]
    Cin = CinW if Cin is None else Cin
    assert Cin == CinW and Cin <= ldx
    y = torch.empty(B, H, W, CoutP, device=x.device, dtype=x.dtype)
    st = torch.empty(conv3x3_nparts(B, H, W, Cin, CoutP, x.dtype), 2, CoutP, device=x.device, dtype=torch.float32) if want_stats else None
    info = {"B": B, "H": H, "W": W, "Cin": Cin, "CoutP": CoutP, "xform": in_scale is not None, "dtype": x.dtype}
    rc = _timed("conv3x3_fwd", info, lambda: _lib.lib().wm_conv3x3_fwd(
        _p(x), c_int(ldx), _p(wp), _p(bias), c_int(0 if bias is None else bias.numel()), _p(in_scale), _p(in_shift),
        _p(y), c_int(CoutP), _p(st), c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(CoutP), c_int(dtype_id(x)), _sweep(reverse), _stream()))
    _lib.check(rc, "wm_conv3x3_fwd")
    return y, st


def conv3x3_fwd_addin(x, wp, in_scale, in_shift, addend, reverse=False):
    """dense 64 -> 64 forward conv (16-bit dtypes) whose epilogue adds `addend` [B,H,W,64] before the BatchNorm statistics:
    y = conv3x3(relu(in_scale*x + in_shift), wp) + addend.  Returns (y, stat partials)."""
    _need_cuda(x, wp, addend)
    B, H, W, C = x.shape
    assert C == 64 and tuple(wp.shape) == (9, 64, 64) and addend.shape == x.shape and addend.dtype == x.dtype and x.is_contiguous() and addend.is_contiguous()
    y = torch.empty_like(x)
    st = torch.empty(conv3x3_nparts(B, H, W, 64, 64, x.dtype), 2, 64, device=x.device, dtype=torch.float32)
    info = {"B": B, "H": H, "W": W, "Cin": 64, "CoutP": 64, "xform": True, "dtype": x.dtype, "addin": True}
    rc = _timed("conv3x3_fwd", info, lambda: _lib.lib().wm_conv3x3_fwd_addin(_p(x), _p(wp), _p(in_scale), _p(in_shift), _p(addend), _p(y), _p(st), c_int(B),
                                                                             c_int(H), c_int(W), c_int(dtype_id(x)), _sweep(reverse), _stream()))
    _lib.check(rc, "wm_conv3x3_fwd_addin")
    return y, st


def concat_side_fwd(img, w, bias, msg, dtype, c_msg, L, c_img):
    """P [B,H,W,64] = conv3(image channels of w) + bias + the message term (hidden_models/encoder.py:34-41 without the concat):
    img [B,3,H,W] f32, w [64,Cin,3,3] f32, msg [B,L] f32."""
    _need_cuda(img, w, msg)
    B, _, H, W = img.shape
    Cin = w.shape[1]
    assert w.shape[0] == 64 and img.shape[1] == 3 and img.dtype == torch.float32 and img.is_contiguous() and w.is_contiguous()
    msg = msg.contiguous().float()
    P = torch.empty(B, H, W, 64, device=img.device, dtype=dtype)
    wside = torch.empty(64 * 32, device=img.device, dtype=dtype)
    mbias = torch.empty(B, 9, 64, device=img.device, dtype=torch.float32)
    rc = _lib.lib().wm_concat_side_fwd(_p(img), _p(w), _p(bias), _p(msg), _p(wside), _p(mbias), _p(P), c_int(B), c_int(H), c_int(W), c_int(Cin),
                                       c_int(c_msg), c_int(L), c_int(c_img), c_int(dt_id(dtype)), _stream())
    _lib.check(rc, "wm_concat_side_fwd")
    return P


def concat_side_msg_wgrad(dy, msg, dw, accumulate, c_msg, L):
    """dw[:, c_msg + l, tap] (+)= sum_b msg[b,l] * (sum of dy[b] over the pixels for which the tap lies inside the image)"""
    B, H, W, C = dy.shape
    assert C == 64 and dy.is_contiguous() and dw.is_contiguous() and dw.shape[0] == 64
    msg = msg.contiguous().float()
    partial = torch.empty(B, _lib.lib().wm_concat_side_partial_rows(), 64, device=dy.device, dtype=torch.float32)
    S = torch.empty(B, 9, 64, device=dy.device, dtype=torch.float32)
    rc = _lib.lib().wm_concat_side_msg_wgrad(_p(dy), _p(msg), _p(partial), _p(S), _p(dw), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W),
                                             c_int(dw.shape[1]), c_int(c_msg), c_int(L), c_int(dtype_id(dy)), _stream())
    _lib.check(rc, "wm_concat_side_msg_wgrad")


def bn_finalize(partials, C, CP, count, gamma, beta, running_mean, running_var, momentum, eps):
    dev = partials.device
    out = torch.empty(4, CP, device=dev, dtype=torch.float32)  # scale, shift, mean, invstd
    rc = _lib.lib().wm_bn_finalize(_p(partials), c_int(partials.shape[0]), c_int(C), c_int(CP), c_double(count), _p(gamma),
                                   _p(beta), _p(running_mean), _p(running_var), c_float(momentum), c_float(eps),
                                   _p(out[0]), _p(out[1]), _p(out[2]), _p(out[3]), _stream())
    _lib.check(rc, "wm_bn_finalize")
    return out


def bn_bwd_coef(g, gvec, y, stats, C, gamma, dgamma, dbeta, accumulate):
    """first half of the ReLU+BN backward: the reduce pass + finalisation.  Writes dgamma / dbeta, returns coef [3,CP]
    (gamma*invstd, mean(gz), mean(gz*xhat)) for the apply pass."""
    B, H, W, CP = y.shape
    hw = H * W
    L = _lib.lib()
    nparts = L.wm_bn_bwd_nparts(c_size_t(B * hw))
    dev = y.device
    part = torch.empty(nparts, 2, CP, device=dev, dtype=torch.float32)
    ldg = c_int(0 if g is None else g.shape[-1])
    rc = L.wm_bn_bwd_reduce(_p(g), ldg, _p(gvec), _p(y), c_int(CP), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]),
                            _p(part), c_int(B), c_size_t(hw), c_int(CP), c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_bn_bwd_reduce")
    coef = torch.empty(3, CP, device=dev, dtype=torch.float32)
    rc = L.wm_bn_bwd_finalize(_p(part), c_int(nparts), c_int(C), c_int(CP), c_double(B * hw), _p(gamma), _p(stats[3]),
                              _p(dgamma), _p(dbeta), c_int(1 if accumulate else 0), _p(coef), _stream())
    _lib.check(rc, "wm_bn_bwd_finalize")
    return coef


def bn_bwd_coef_raw(partials, y, stats, C, gamma, dgamma, dbeta, accumulate):
    """bn_bwd_coef from partial rows [n,2,CP] = sum(gz), sum(gz*y) already reduced by the dgrad that produced g
    (conv3x3_dgrad_bwdstats): no pass over (g, y)."""
    B, H, W, CP = y.shape
    coef = torch.empty(3, CP, device=y.device, dtype=torch.float32)
    rc = _lib.lib().wm_bn_bwd_finalize_raw(_p(partials), c_int(partials.shape[0]), c_int(C), c_int(CP), c_double(B * H * W), _p(gamma),
                                           _p(stats[2]), _p(stats[3]), _p(dgamma), _p(dbeta), c_int(1 if accumulate else 0), _p(coef),
                                           _stream())
    _lib.check(rc, "wm_bn_bwd_finalize_raw")
    return coef


def bn_bwd(g, gvec, y, stats, C, gamma, dgamma, dbeta, accumulate, dbias, coef=None):
    """ReLU+BN backward.  g [B,H,W,ld] or None with gvec [B,CP]; y raw conv output [B,H,W,CP];
    stats = [4,CP] (scale, shift, mean, invstd).  Returns dy [B,H,W,CP]; writes dgamma/dbeta/dbias.
    coef: the result of bn_bwd_coef / bn_bwd_coef_raw when the reduce pass already ran (dgamma / dbeta written there)."""
    B, H, W, CP = y.shape
    hw = H * W
    L = _lib.lib()
    dev = y.device
    if coef is None:
        coef = bn_bwd_coef(g, gvec, y, stats, C, gamma, dgamma, dbeta, accumulate)
    nparts = L.wm_bn_bwd_nparts(c_size_t(B * hw))
    ldg = c_int(0 if g is None else g.shape[-1])
    dy = torch.empty_like(y)
    bpart = torch.empty(nparts, CP, device=dev, dtype=torch.float32) if dbias is not None else None
    rc = L.wm_bn_bwd_apply(_p(g), ldg, _p(gvec), _p(y), c_int(CP), _p(stats[0]), _p(stats[1]), _p(stats[2]), _p(stats[3]),
                           _p(coef), _p(dy), c_int(CP), _p(bpart), c_int(B), c_size_t(hw), c_int(CP), c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_bn_bwd_apply")
    if dbias is not None:
        colsum(bpart, dbias.numel(), CP, dbias, accumulate)
    return dy


def conv3x3_wgrad_bnfused_supported(CinX, CoutY, dtype):
    return bool(_lib.lib().wm_conv3x3_wgrad_bnfused_supported(c_int(CinX), c_int(CoutY), c_int(dt_id(dtype))))


def conv3x3_wgrad_bnfused(x, g, y, stats, coef, dw, accumulate):
    """weight gradient of an image-fed first layer with the BatchNorm-backward apply pass fused: dy is formed from
    (g, y, stats [4,CP] contiguous, coef [3,CP]) while the tile is staged and never written to memory."""
    B, H, W, ldx = x.shape
    CoutY = y.shape[-1]
    L = _lib.lib()
    L.wm_conv3x3_wgrad_ws_bytes.restype = c_size_t
    nbytes = L.wm_conv3x3_wgrad_ws_bytes(c_int(B), c_int(H), c_int(W), c_int(ldx), c_int(CoutY))
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    Cout, Cin = dw.shape[0], dw.shape[1]
    assert dw.is_contiguous() and stats.is_contiguous() and coef.is_contiguous()
    rc = L.wm_conv3x3_wgrad_bnfused(_p(x), c_int(ldx), c_int(ldx), _p(g), c_int(g.shape[-1]), _p(y), c_int(CoutY), c_int(CoutY),
                                    _p(stats), _p(coef), _p(ws), _p(dw), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W),
                                    c_int(Cin), c_int(Cout), c_int(dtype_id(x)), _stream())
    _lib.check(rc, "wm_conv3x3_wgrad_bnfused")


def conv3x3_gvfused_supported(CinX, CoutY, dtype):
    return bool(_lib.lib().wm_conv3x3_gvfused_supported(c_int(CinX), c_int(CoutY), c_int(dt_id(dtype))))


def conv3x3_wgrad_gvfused(x, in_scale, in_shift, gvec, y, stats, coef, dw, accumulate, fin=None):
    """weight gradient of a globally pooled ConvBNRelu with the BatchNorm-backward apply pass fused: dy is formed from
    (gvec [B,CP], y, stats [4,CP] contiguous, coef [3,CP]) while the tile is staged."""
    B, H, W, ldx = x.shape
    CoutY = y.shape[-1]
    L = _lib.lib()
    L.wm_conv3x3_wgrad_ws_bytes.restype = c_size_t
    nbytes = L.wm_conv3x3_wgrad_ws_bytes(c_int(B), c_int(H), c_int(W), c_int(ldx), c_int(CoutY))
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    Cout, Cin = dw.shape[0], dw.shape[1]
    assert dw.is_contiguous() and stats.is_contiguous() and coef.is_contiguous() and gvec.is_contiguous() and gvec.shape[-1] == CoutY
    fst, fcoef = _fin_rider(fin)
    rc = L.wm_conv3x3_wgrad_gvfused_fin(_p(x), c_int(ldx), c_int(ldx), _p(in_scale), _p(in_shift), _p(gvec), _p(y), c_int(CoutY), c_int(CoutY),
                                        _p(stats), _p(coef), _p(ws), _p(dw), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W),
                                        c_int(Cin), c_int(Cout), c_int(dtype_id(x)), ctypes.byref(fst) if fst is not None else None, _stream())
    _lib.check(rc, "wm_conv3x3_wgrad_gvfused")
    return fcoef


def conv3x3_dgrad_gvfused(y, wpt, gvec, stats, coef):
    """input gradient of the same layer: conv3x3 of the on-the-fly dy with the transposed packed filter wpt [9,CinP,CoutY]"""
    B, H, W, CoutY = y.shape
    CinP = wpt.shape[1]
    assert wpt.shape[2] == CoutY and stats.is_contiguous() and coef.is_contiguous() and gvec.is_contiguous()
    dx = torch.empty(B, H, W, CinP, device=y.device, dtype=y.dtype)
    rc = _lib.lib().wm_conv3x3_dgrad_gvfused(_p(y), c_int(CoutY), c_int(CoutY), _p(wpt), _p(gvec), _p(stats), _p(coef), _p(dx), c_int(B),
                                             c_int(H), c_int(W), c_int(CinP), c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_conv3x3_dgrad_gvfused")
    return dx


def conv3x3_dgrad_bwdstats_supported(CoutY, CinP, dtype):
    return bool(_lib.lib().wm_conv3x3_dgrad_bwdstats_supported(c_int(CoutY), c_int(CinP), c_int(dt_id(dtype))))


def conv3x3_dgrad_bwdstats(src, wpt, ry, r_scale, r_shift, gvec=None, stats=None, coef=None, reverse=False):
    """input gradient dx = conv3x3(dy, wpt) whose epilogue also reduces the BatchNorm-backward sums of the layer that feeds
    this one (raw output ry, constants r_scale / r_shift).  src = dy, or (with gvec, stats, coef) this layer's raw output
    with the apply pass fused.  Returns (dx, partials [n,2,64])."""
    B, H, W, lds = src.shape
    CinP, CoutY = wpt.shape[1], wpt.shape[2]
    assert ry.shape == (B, H, W, CinP) and ry.is_contiguous() and CoutY <= lds
    assert (gvec is None) == (stats is None) == (coef is None)
    dx = torch.empty(B, H, W, CinP, device=src.device, dtype=src.dtype)
    part = torch.empty(conv3x3_nparts(B, H, W, CoutY, CinP, src.dtype), 2, CinP, device=src.device, dtype=torch.float32)
    rc = _lib.lib().wm_conv3x3_dgrad_bwdstats(_p(src), c_int(lds), c_int(CoutY), _p(wpt), _p(gvec), _p(stats), _p(coef), _p(ry), _p(r_scale),
                                              _p(r_shift), _p(dx), _p(part), c_int(B), c_int(H), c_int(W), c_int(CinP),
                                              c_int(dtype_id(src)), _sweep(reverse), _stream())
    _lib.check(rc, "wm_conv3x3_dgrad_bwdstats")
    dx._wm_masked = True   # the kernel writes gz = g * [z > 0] of the feeding layer (what conv3x3_bwd_fused's premasked staging expects)
    return dx, part


def conv3x3_dgrad_applyfused_supported(CoutY, CinP, dtype):
    return bool(_lib.lib().wm_conv3x3_dgrad_applyfused_supported(c_int(CoutY), c_int(CinP), c_int(dt_id(dtype))))


def conv3x3_dgrad_applyfused(g, y, stats, coef, wpt, ry=None, r_scale=None, r_shift=None, reverse=False, want_dy=True):
    """64 -> 64 layer, gradient g a dense tensor: the BatchNorm-backward apply pass inside the input-gradient kernel.
    Returns (dy, dx, partials or None): dy for the weight gradient, dx = conv3x3(dy, wpt), partials = the feeding layer's
    BatchNorm-backward sums when its raw output ry (+ r_scale, r_shift) is given."""
    B, H, W, C = y.shape
    CinP = wpt.shape[1]
    assert C == 64 and g.shape == y.shape and g.is_contiguous() and y.is_contiguous() and tuple(wpt.shape) == (9, CinP, 64) and CinP in (64, 32)
    assert stats.is_contiguous() and coef.is_contiguous() and (ry is None or (CinP == 64 and ry.shape == y.shape and ry.is_contiguous()))
    dy = torch.empty_like(y) if want_dy else None     # want_dy False: the input gradient alone (no weight gradient will read dy)
    dx = torch.empty(B, H, W, CinP, device=y.device, dtype=y.dtype)
    part = torch.empty(conv3x3_nparts(B, H, W, 64, 64, y.dtype), 2, 64, device=y.device, dtype=torch.float32) if ry is not None else None
    info = {"B": B, "H": H, "W": W, "feed": ry is not None, "dtype": y.dtype}
    rc = _timed("conv3x3_dgrad_applyfused", info, lambda: _lib.lib().wm_conv3x3_dgrad_applyfused(
        _p(g), _p(y), _p(stats), _p(coef), _p(wpt), _p(dy), _p(dx), _p(ry), _p(r_scale), _p(r_shift), _p(part), c_int(B), c_int(H), c_int(W),
        c_int(CinP), c_int(dtype_id(y)), _sweep(reverse), _stream()))
    _lib.check(rc, "wm_conv3x3_dgrad_applyfused")
    if ry is not None:
        dx._wm_masked = True   # as conv3x3_dgrad_bwdstats
    return dy, dx, part


def conv3x3_bwd_fused_supported(dtype, shape=None):
    """the one-kernel backward exists for this dtype (and, given y's shape [B,H,W,64], the tensor fits its 32-bit offsets)"""
    if shape is None:
        return bool(_lib.lib().wm_conv3x3_bwd_fused_supported(c_int(dt_id(dtype))))
    return bool(_lib.lib().wm_conv3x3_bwd_fused_supported_shape(c_int(shape[0]), c_int(shape[1]), c_int(shape[2]), c_int(dt_id(dtype))))


def conv3x3_bwd_fused_gvec_max_batch():
    return int(_lib.lib().wm_conv3x3_bwd_fused_gvec_max_batch())


def conv3x3_bwd_fused(g, y, stats, coef, wpt, xr, in_scale, in_shift, dw, accumulate, reverse=False, fin=None, premasked=False, gvec=None):
    """The whole backward of a 64 -> 64 body layer fed by another ConvBNRelu in one pass (csrc/bwd_ws.hip): returns
    (dx -- multiplied by the feeding layer's ReLU mask --, partials [nwg,2,64] = the feeding layer's BatchNorm-backward sums, the
    rider's coef or None); dw is written in place by the slab reduction that follows the kernel.  gvec [B,64] f32 instead of g: the
    layer's output was globally pooled (one gradient row per sample)."""
    B, H, W, C = y.shape
    if (g is None) == (gvec is None):
        raise RuntimeError("conv3x3_bwd_fused: exactly one of g (a tensor) and gvec (one row per sample) is given")
    assert C == 64 and xr.shape == y.shape and y.is_contiguous() and xr.is_contiguous()
    assert (g.shape == y.shape and g.is_contiguous()) if g is not None else (tuple(gvec.shape) == (B, 64) and gvec.is_contiguous() and gvec.dtype == torch.float32)
    assert tuple(wpt.shape) == (9, 64, 64) and stats.is_contiguous() and coef.is_contiguous() and dw.is_contiguous()
    L = _lib.lib()
    nwg = L.wm_conv3x3_bwd_fused_nwg(c_int(B), c_int(H), c_int(W))
    dx = torch.empty_like(y)
    part = torch.empty(nwg, 2, 64, device=y.device, dtype=torch.float32)
    fst, fcoef = _fin_rider(fin(part) if callable(fin) else fin)     # fin: a rider dict, or a function of the partial rows this call produces
    ws = torch.empty(nwg * 9 * 64 * 64, device=y.device, dtype=torch.float32)
    Cout, Cin = dw.shape[0], dw.shape[1]
    info = {"B": B, "H": H, "W": W, "dtype": y.dtype, "gvec": gvec is not None}
    rc = _timed("conv3x3_bwd_fused", info, lambda: L.wm_conv3x3_bwd_fused(
        _p(g), _p(gvec), _p(y), _p(stats), _p(coef), _p(wpt), _p(xr), _p(in_scale), _p(in_shift), _p(dx), _p(part), _p(ws), c_int(B), c_int(H),
        c_int(W), c_int(dtype_id(y)), c_int(1 if premasked else 0), _sweep(reverse), _stream()))
    _lib.check(rc, "wm_conv3x3_bwd_fused")
    rc = L.wm_conv3x3_bwd_fused_reduce(_p(ws), _p(dw), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(Cout),
                                       ctypes.byref(fst) if fst is not None else None, _stream())
    _lib.check(rc, "wm_conv3x3_bwd_fused_reduce")
    dx._wm_masked = True
    return dx, part, fcoef


def conv3x3_bwd_fused16_supported(shape, dtype):
    """the one-kernel backward of an image-fed first layer exists for y's shape [B,H,W,64] (whole 8x16 tiles) and dtype"""
    return bool(_lib.lib().wm_conv3x3_bwd_fused16_supported(c_int(shape[0]), c_int(shape[1]), c_int(shape[2]), c_int(dt_id(dtype))))


def conv3x3_bwd_fused16(g, y, stats, coef, wpt, x, dw, accumulate, reverse=False, premasked=False):
    """The whole backward of an image-fed first ConvBNRelu in one pass (csrc/bwd_ws16.hip): returns dx [B,H,W,16] (gradient wrt the 16-channel
    image tensor x); dw [Cout,Cin<=16,3,3] is written in place by the slab reduction that follows the kernel."""
    B, H, W, C = y.shape
    assert C == 64 and g.shape == y.shape and g.is_contiguous() and y.is_contiguous() and tuple(x.shape) == (B, H, W, 16) and x.is_contiguous()
    assert tuple(wpt.shape) == (9, 16, 64) and stats.is_contiguous() and coef.is_contiguous() and dw.is_contiguous() and dw.shape[0] <= 64 and dw.shape[1] <= 16
    L = _lib.lib()
    nwg = L.wm_conv3x3_bwd_fused16_nwg(c_int(B), c_int(H), c_int(W))
    dx = torch.empty(B, H, W, 16, device=y.device, dtype=y.dtype)
    ws = torch.empty(nwg * 9 * 16 * 64, device=y.device, dtype=torch.float32)
    info = {"B": B, "H": H, "W": W, "dtype": y.dtype}
    rc = _timed("conv3x3_bwd_fused16", info, lambda: L.wm_conv3x3_bwd_fused16(
        _p(g), _p(y), _p(stats), _p(coef), _p(wpt), _p(x), _p(dx), _p(ws), _p(dw), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W),
        c_int(dw.shape[1]), c_int(dw.shape[0]), c_int(dtype_id(y)), c_int(1 if premasked else 0), _sweep(reverse), _stream()))
    _lib.check(rc, "wm_conv3x3_bwd_fused16")
    return dx


def linear_head_fwd(pooled, w, bias, I):
    """pooled [B,ldp] f32 (first I columns used), w [O,I], bias [O] -> [B,O]"""
    _need_cuda(pooled, w)
    B, ldp = pooled.shape
    O = w.shape[0]
    out = torch.empty(B, O, device=pooled.device, dtype=torch.float32)
    rc = _lib.lib().wm_linear_head_fwd(_p(pooled), c_int(ldp), _p(w), _p(bias), _p(out), c_int(B), c_int(I), c_int(O), _stream())
    _lib.check(rc, "wm_linear_head_fwd")
    return out


def linear_head_bwd(pooled, w, g_out, dw, db, accumulate, CP, inv_hw):
    """writes dw [O,I], db [O]; returns gvec [B,CP] = (g_out @ w) * inv_hw (zero padded)"""
    _need_cuda(pooled, w, g_out)
    B, ldp = pooled.shape
    O, I = w.shape
    g = g_out.contiguous().float()
    gvec = torch.empty(B, CP, device=pooled.device, dtype=torch.float32)
    rc = _lib.lib().wm_linear_head_bwd(_p(pooled), c_int(ldp), _p(w), _p(g), _p(dw), _p(db), c_int(1 if accumulate else 0), _p(gvec),
                                       c_int(CP), c_float(inv_hw), c_int(B), c_int(I), c_int(O), _stream())
    _lib.check(rc, "wm_linear_head_bwd")
    return gvec


def pooled_head_supported(B, CP, I, O):
    return bool(_lib.lib().wm_pooled_head_supported(c_int(B), c_int(CP), c_int(I), c_int(O)))


def pooled_head(out3, I, w, bias, kind, target, messages, gscale, gscale_dev, dw, db, accumulate, inv_hw, C, count, gamma, stats, dgamma, dbeta):
    """linear_head_fwd + the loss (kind 0: bce_logits vs the constant `target`; kind 1: message_loss vs messages [B,O]) + linear_head_bwd +
    bn_bwd_coef_pooled as ONE launch.  out3 [3,B,CP] = bnrelu_avgpool_stats' buffer (pooled, N+, S+); stats [4,CP] of the pooled
    ConvBNRelu.  -> (logits [B,O], loss [1] or [2], gvec [B,CP], coef [3,CP]); dw, db, dgamma, dbeta written in place."""
    _, B, CP = out3.shape
    O = w.shape[0]
    assert out3.is_contiguous() and out3.dtype == torch.float32 and tuple(w.shape) == (O, I) and w.is_contiguous() and dw.is_contiguous()
    dev = out3.device
    logits = torch.empty(B, O, device=dev, dtype=torch.float32)
    loss = torch.empty(2 if kind == 1 else 1, device=dev, dtype=torch.float32)
    gvec = torch.empty(B, CP, device=dev, dtype=torch.float32)
    coef = torch.empty(3, CP, device=dev, dtype=torch.float32)
    msg = messages.contiguous().float() if messages is not None else None
    rc = _lib.lib().wm_pooled_head(_p(out3), c_int(B), c_int(CP), c_int(I), c_int(O), _p(w), _p(bias), c_int(kind), c_float(float(target)), _p(msg),
                                   c_float(float(gscale)), _p(gscale_dev), _p(logits), _p(loss), _p(dw), _p(db), c_int(1 if accumulate else 0),
                                   _p(gvec), c_float(inv_hw), c_int(C), c_double(float(count)), _p(gamma), _p(stats[2]), _p(stats[3]),
                                   _p(dgamma), _p(dbeta), _p(coef), _stream())
    _lib.check(rc, "wm_pooled_head")
    return logits, loss, gvec, coef


def bce_logits(logits, target, gscale=1.0, want_grad=True, gscale_dev=None):
    """BCEWithLogitsLoss(mean) against a constant label: (loss [1] tensor, gscale * d loss / d logits or None).
    gscale_dev (here and in the other loss ops): optional device scalar multiplied into gscale (the AMP loss scale)."""
    _need_cuda(logits)
    x = logits.contiguous().float()
    loss = torch.empty(1, device=x.device, dtype=torch.float32)
    grad = torch.empty_like(x) if want_grad else None
    rc = _lib.lib().wm_bce_logits(_p(x), c_float(float(target)), c_int(x.numel()), c_float(float(gscale)), _p(gscale_dev), _p(loss), _p(grad),
                                  _stream())
    _lib.check(rc, "wm_bce_logits")
    return loss, grad


def message_loss(decoded, messages, gscale, want_grad=True, gscale_dev=None):
    """(out [2] = [mean squared error, bitwise error], gscale * (decoded - messages) or None)."""
    _need_cuda(decoded, messages)
    d = decoded.contiguous().float(); m = messages.contiguous().float()
    out = torch.empty(2, device=d.device, dtype=torch.float32)
    grad = torch.empty_like(d) if want_grad else None
    rc = _lib.lib().wm_message_loss(_p(d), _p(m), c_int(d.numel()), c_float(float(gscale)), _p(gscale_dev), _p(out), _p(grad), _stream())
    _lib.check(rc, "wm_message_loss")
    return out, grad


def hidden_metrics(enc_partials, n_img, msg2, adv, d_cover, d_enc, w_adv, w_enc, w_dec):
    """[7] f32: loss, encoder mse, decoder mse, bitwise error, adversarial bce, D(cover) bce, D(encoded) bce (one launch)"""
    out = torch.empty(7, device=enc_partials.device, dtype=torch.float32)
    rc = _lib.lib().wm_hidden_metrics(_p(enc_partials), c_int(enc_partials.numel()), c_double(float(n_img)), _p(msg2), _p(adv), _p(d_cover),
                                      _p(d_enc), c_float(w_adv), c_float(w_enc), c_float(w_dec), _p(out), _stream())
    _lib.check(rc, "wm_hidden_metrics")
    return out


def colsum(partials, C, ldp, out, accumulate):
    rc = _lib.lib().wm_colsum_finalize(_p(partials), c_int(partials.shape[0]), c_int(C), c_int(ldp), _p(out),
                                       c_int(1 if accumulate else 0), _stream())
    _lib.check(rc, "wm_colsum_finalize")


def conv3x3_wgrad(x, CinX, in_scale, in_shift, dy, dw, accumulate, perm_dev=None, reverse=False, fin=None):
    """dw [Cout,Cin,3,3] f32 view (written in place).  fin: see _fin_rider; then the rider's coef [3,CP] is returned."""
    fst, fcoef = _fin_rider(fin)
    B, H, W, ldx = x.shape
    CoutY = dy.shape[-1]
    L = _lib.lib()
    L.wm_conv3x3_wgrad_ws_bytes.restype = c_size_t
    nbytes = L.wm_conv3x3_wgrad_ws_bytes(c_int(B), c_int(H), c_int(W), c_int(CinX), c_int(CoutY))
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
    Cout, Cin = dw.shape[0], dw.shape[1]
    assert dw.is_contiguous()
    rc = L.wm_conv3x3_wgrad_fin(_p(x), c_int(ldx), c_int(CinX), _p(in_scale), _p(in_shift), _p(dy), c_int(CoutY), c_int(CoutY),
                                _p(ws), _p(dw), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W), c_int(Cin),
                                c_int(Cout), _p(perm_dev), c_int(dtype_id(x)), ctypes.byref(fst) if fst is not None else None, _sweep(reverse),
                                _stream())
    _lib.check(rc, "wm_conv3x3_wgrad")
    return fcoef


def conv3x3_fwd_elu_supported(Cin, CoutP, dtype):
    return bool(_lib.lib().wm_conv3x3_fwd_elu_supported(c_int(Cin), c_int(CoutP), c_int(dt_id(dtype))))


def conv3x3_fwd_elu(x, wp, bias):
    """out [B,H,W,64] = elu(conv3x3(x, wp) + bias) in one launch (16-bit dtypes; the pre-activation is not stored)"""
    _need_cuda(x, wp)
    B, H, W, ldx = x.shape
    Cin = wp.shape[2]
    assert wp.shape[1] == 64 and Cin <= ldx
    out = torch.empty(B, H, W, 64, device=x.device, dtype=x.dtype)
    rc = _lib.lib().wm_conv3x3_fwd_elu(_p(x), c_int(ldx), _p(wp), _p(bias), c_int(0 if bias is None else bias.numel()), _p(out), c_int(B), c_int(H),
                                       c_int(W), c_int(Cin), c_int(dtype_id(x)), c_int(0), _stream())
    _lib.check(rc, "wm_conv3x3_fwd_elu")
    return out


def conv3x3_dgrad_elufused_supported(CinP, dtype):
    return bool(_lib.lib().wm_conv3x3_dgrad_elufused_supported(c_int(CinP), c_int(dt_id(dtype))))


def conv3x3_dgrad_elufused(g, out, wpt, want_gz=True, dx_stride=None):
    """backward of a conv + ELU layer, input-gradient half: gz = g * (out > 0 ? 1 : out + 1) formed while staging; returns
    (dx [B,H,W,CinP], gz [B,H,W,64] or None, bias_partials f32 [nparts,64]).  dx_stride = 16: wpt packed to 32 rows (upper 16 zero),
    dx [B,H,W,16]"""
    _need_cuda(g, out, wpt)
    B, H, W, C = g.shape
    CinP = wpt.shape[1] if dx_stride is None else dx_stride
    assert C == 64 and out.shape == g.shape and out.dtype == g.dtype and g.is_contiguous() and out.is_contiguous() and wpt.shape[2] == 64
    assert CinP == wpt.shape[1] or (CinP == 16 and wpt.shape[1] == 32)
    L = _lib.lib()
    dx = torch.empty(B, H, W, CinP, device=g.device, dtype=g.dtype)
    gz = torch.empty_like(g) if want_gz else None
    part = torch.empty(L.wm_conv3x3_dgrad_elufused_nparts(c_int(B), c_int(H), c_int(W)), 64, device=g.device, dtype=torch.float32)
    rc = L.wm_conv3x3_dgrad_elufused(_p(g), _p(out), _p(wpt), _p(dx), _p(gz), _p(part), c_int(B), c_int(H), c_int(W), c_int(CinP),
                                     c_int(dtype_id(g)), c_int(0), _stream())
    _lib.check(rc, "wm_conv3x3_dgrad_elufused")
    return dx, gz, part


def conv3x3_wgrad_bias(x, gz, dw, accumulate, bias_partials, db, db_accumulate):
    """dw [Cout,Cin,3,3] (+)= the weight gradient from (x, gz); db [Cout] (+)= the column sums of bias_partials, in the same reduction launch"""
    B, H, W, ldx = x.shape
    CoutY = gz.shape[-1]
    L = _lib.lib()
    L.wm_conv3x3_wgrad_ws_bytes.restype = c_size_t
    ws = torch.empty(L.wm_conv3x3_wgrad_ws_bytes(c_int(B), c_int(H), c_int(W), c_int(ldx), c_int(CoutY)) // 4, device=x.device, dtype=torch.float32)
    Cout, Cin = dw.shape[0], dw.shape[1]
    assert dw.is_contiguous() and db.is_contiguous() and db.numel() == Cout and bias_partials.is_contiguous() and bias_partials.shape[1] == CoutY
    rc = L.wm_conv3x3_wgrad_bias(_p(x), c_int(ldx), c_int(ldx), _p(gz), c_int(CoutY), c_int(CoutY), _p(ws), _p(dw), c_int(1 if accumulate else 0),
                                 c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(Cout), c_int(dtype_id(x)), _p(bias_partials),
                                 c_int(bias_partials.shape[0]), _p(db), c_int(1 if db_accumulate else 0), _stream())
    _lib.check(rc, "wm_conv3x3_wgrad_bias")


# ----------------------------------------------------------------------------- heads
def pool_stats_enabled():
    return bool(_lib.lib().wm_pool_stats_enabled())


def bnrelu_avgpool_stats(y, scale, shift):
    """global average pool of relu(scale*y+shift) plus, per (sample, channel), the active-pixel count N+ and the sum S+ of y over
    the active pixels.  Returns (pooled [B,CP], (N+ [B,CP], S+ [B,CP]))."""
    B, H, W, CP = y.shape
    L = _lib.lib()
    S = L.wm_avgpool_slices(c_size_t(H * W))
    ws = torch.empty(B * S * 3 * CP, device=y.device, dtype=torch.float32)
    out3 = torch.empty(3, B, CP, device=y.device, dtype=torch.float32)
    rc = L.wm_bnrelu_avgpool_stats(_p(y), c_int(CP), _p(scale), _p(shift), _p(out3), _p(ws), c_int(B), c_size_t(H * W), c_int(CP),
                                   c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_bnrelu_avgpool_stats")
    return out3[0], (out3[1], out3[2])


def pooled_bwd_rows(gvec, pool_stats):
    """partial rows [B,2,CP] = (gvec*N+, gvec*S+) for bn_bwd_coef_raw: the pooled layer's BatchNorm-backward sums without a pass over y"""
    B, CP = gvec.shape
    npos, ysum = pool_stats
    assert gvec.is_contiguous() and npos.is_contiguous() and ysum.is_contiguous() and npos.shape == gvec.shape
    rows = torch.empty(B, 2, CP, device=gvec.device, dtype=torch.float32)
    rc = _lib.lib().wm_pooled_bn_bwd_rows(_p(gvec), _p(npos), _p(ysum), c_int(B), c_int(CP), _p(rows), _stream())
    _lib.check(rc, "wm_pooled_bn_bwd_rows")
    return rows


def bn_bwd_coef_pooled(gvec, pool_stats, y, stats, C, gamma, dgamma, dbeta, accumulate):
    """bn_bwd_coef of a globally pooled layer from the forward pool's (N+, S+): one small launch, no pass over y"""
    B, H, W, CP = y.shape
    npos, ysum = pool_stats
    assert gvec.is_contiguous() and npos.is_contiguous() and ysum.is_contiguous() and tuple(gvec.shape) == (B, CP) == tuple(npos.shape)
    coef = torch.empty(3, CP, device=y.device, dtype=torch.float32)
    rc = _lib.lib().wm_bn_bwd_finalize_pooled(_p(gvec), _p(npos), _p(ysum), c_int(B), c_int(C), c_int(CP), c_double(B * H * W), _p(gamma),
                                              _p(stats[2]), _p(stats[3]), _p(dgamma), _p(dbeta), c_int(1 if accumulate else 0), _p(coef),
                                              _stream())
    _lib.check(rc, "wm_bn_bwd_finalize_pooled")
    return coef


def bnrelu_avgpool(y, scale, shift):
    B, H, W, CP = y.shape
    L = _lib.lib()
    S = L.wm_avgpool_slices(c_size_t(H * W))
    ws = torch.empty(B * S * CP, device=y.device, dtype=torch.float32)
    out = torch.empty(B, CP, device=y.device, dtype=torch.float32)
    rc = L.wm_bnrelu_avgpool(_p(y), c_int(CP), _p(scale), _p(shift), _p(out), _p(ws), c_int(B), c_size_t(H * W), c_int(CP),
                             c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_bnrelu_avgpool")
    return out


def conv1x1_head_fwd(y, scale, shift, w, bias, act=0, want_act16=False):
    """y [B,H,W,Cin]; w [Cout,Cin(,1,1)] f32; -> out [B,Cout,H,W] f32; want_act16: -> (out, the same image as the [B,H,W,16] zero-padded
    NHWC tensor of y's dtype that the image-fed first layers read: engine.image_to_act's result without its launch)"""
    B, H, W, Cin = y.shape
    Cout = w.shape[0]
    out = torch.empty(B, Cout, H, W, device=y.device, dtype=torch.float32)
    a16 = torch.empty(B, H, W, 16, device=y.device, dtype=y.dtype) if want_act16 else None
    rc = _lib.lib().wm_conv1x1_head_fwd_act(_p(y), c_int(Cin), _p(scale), _p(shift), _p(w), _p(bias), _p(out), _p(a16), c_int(B),
                                            c_size_t(H * W), c_int(Cin), c_int(Cout), c_int(act), c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_conv1x1_head_fwd")
    return (out, a16) if want_act16 else out


def conv1x1_head_bwd(y, scale, shift, w, gout, dw, dbias, accumulate, want_bn_partials=False):
    """-> g (gradient wrt the ReLU output feeding the head); with want_bn_partials: (g, rows [n,2,Cin] of sum(gz), sum(gz*y) of the
    ConvBNRelu that produced y, for bn_bwd_coef_raw: that layer then needs no reduce pass over (g, y))"""
    B, H, W, Cin = y.shape
    Cout = w.shape[0]
    L = _lib.lib()
    nparts = L.wm_conv1x1_head_nparts(c_size_t(B * H * W))
    part = torch.empty(nparts, Cout * (Cin + 1), device=y.device, dtype=torch.float32)
    bnp = torch.empty(nparts, 2, Cin, device=y.device, dtype=torch.float32) if want_bn_partials and scale is not None else None
    g = torch.empty_like(y)
    gout = gout.contiguous()
    rc = L.wm_conv1x1_head_bwd(_p(y), c_int(Cin), _p(scale), _p(shift), _p(w), _p(gout), _p(g), c_int(Cin), _p(part), _p(bnp), c_int(B),
                               c_size_t(H * W), c_int(Cin), c_int(Cout), c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_conv1x1_head_bwd")
    ldp = Cout * (Cin + 1)
    colsum(part, Cout * Cin, ldp, dw, accumulate)
    # the first call folded the rows in place to <= 64 (wm_colsum_finalize treats partials as scratch)
    colsum(part[:(nparts if nparts <= 256 else 64), Cout * Cin:], Cout, ldp, dbias, accumulate)
    if want_bn_partials:
        return g, bnp
    return g


# ----------------------------------------------------------------------------- losses / optimiser
def mse_fwd_bwd(a, b, gscale, want_grad=True, gscale_dev=None):
    """returns (sum of squared differences partials [nparts], grad = gscale*(a-b))"""
    a = a.contiguous(); b = b.contiguous()
    n = a.numel()
    nparts = max(1, min(1024, (n + 4095) // 4096))
    part = torch.empty(nparts, device=a.device, dtype=torch.float32)
    grad = torch.empty_like(a) if want_grad else None
    rc = _lib.lib().wm_mse_fwd_bwd(_p(a), _p(b), _p(grad), c_float(gscale), _p(gscale_dev), _p(part), c_int(nparts), c_size_t(n), _stream())
    _lib.check(rc, "wm_mse_fwd_bwd")
    return part, grad


def image_grad_mse(g, a, b, gscale, C=3, C_off=0, gscale_dev=None):
    """(out [B,C,H,W] f32 = nhwc_to_nchw(g)[:, :C] + gscale * (a - b), partials of sum (a - b)^2): the discriminator's input gradient, the
    image-fidelity term's gradient and its loss in one pass (nhwc_to_nchw + mse_fwd_bwd + axpy_ before)"""
    _need_cuda(g, a, b)
    B, H, W, ld = g.shape
    assert a.shape == (B, C, H, W) and b.shape == a.shape and a.dtype == torch.float32 and b.dtype == torch.float32 and g.is_contiguous()
    a = a.contiguous(); b = b.contiguous()
    n = a.numel()
    nparts = max(1, min(1024, (n + 4095) // 4096))
    out = torch.empty_like(a)
    part = torch.empty(nparts, device=a.device, dtype=torch.float32)
    rc = _lib.lib().wm_image_grad_mse(_p(g), c_int(ld), c_int(C_off), _p(a), _p(b), _p(out), c_float(gscale), _p(gscale_dev), _p(part), c_int(nparts),
                                      c_int(B), c_int(C), c_int(H), c_int(W), c_int(dtype_id(g)), _stream())
    _lib.check(rc, "wm_image_grad_mse")
    return out, part


def axpy_(a, b, s=1.0):
    assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    rc = _lib.lib().wm_axpy(_p(a), _p(b), c_float(s), c_size_t(a.numel()), _stream())
    _lib.check(rc, "wm_axpy")
    _wrote(a)
    return a


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, decoupled=False, grad_scale=1.0):
    rc = _lib.lib().wm_adam_step(_p(p), _p(g), _p(m), _p(v), c_size_t(p.numel()), c_float(lr), c_float(beta1), c_float(beta2),
                                 c_float(eps), c_float(weight_decay), c_int(1 if decoupled else 0), c_int(step),
                                 c_float(grad_scale), _stream())
    _lib.check(rc, "wm_adam_step")
    _wrote(p, m, v)


def adam_hyper(lr, beta1, beta2, step):
    """(lr / (1 - beta1^step), sqrt(1 - beta2^step)) as wm_adam_step derives them (host arithmetic, no launch)"""
    out = (c_float * 2)()
    rc = _lib.lib().wm_adam_hyper(c_float(lr), c_float(beta1), c_float(beta2), c_int(step), out)
    _lib.check(rc, "wm_adam_hyper")
    return float(out[0]), float(out[1])


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, weight_decay, hyper_dev, decoupled=False, grad_scale=1.0):
    """adam_step with the step-count-dependent constants read from the device tensor hyper_dev [2] f32 (adam_hyper's pair): what a
    captured step launches -- the caller refreshes hyper_dev before each replay"""
    assert hyper_dev.is_cuda and hyper_dev.dtype == torch.float32 and hyper_dev.numel() >= 2 and hyper_dev.is_contiguous()
    rc = _lib.lib().wm_adam_step_dev(_p(p), _p(g), _p(m), _p(v), c_size_t(p.numel()), c_float(lr), c_float(beta1), c_float(beta2),
                                     c_float(eps), c_float(weight_decay), c_int(1 if decoupled else 0), _p(hyper_dev),
                                     c_float(grad_scale), _stream())
    _lib.check(rc, "wm_adam_step_dev")
    _wrote(p, m, v)


class AmpState:
    """torch.cuda.amp.GradScaler (IRNcrop_model.py:143,407-416) with its state on the device: see include/wm_hip.h (wm_amp_*).
    `scale` is the device scalar the loss kernels multiply their gradient seeds by."""
    N = 16

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        st = torch.zeros(self.N, dtype=torch.float32)
        st[0], st[2], st[3], st[4] = init_scale, growth_factor, backoff_factor, growth_interval
        self.state = st.to(device)
        self.scale = self.state[0:1]
        self.nopt = 0

    def slot(self):
        """index of a new optimiser under this scaler (<= 4)"""
        k = self.nopt
        if k >= 4:
            raise ValueError("at most four optimisers per scaler")
        self.nopt += 1
        return k

    def found_inf(self, k, parts):
        """found_inf[k] from the wm_sumsq rows of optimiser k's gradient buffers"""
        arr = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
        ns = (ctypes.c_int * len(parts))(*[p.numel() for p in parts])
        rc = _lib.lib().wm_amp_found_inf(arr, ns, c_int(len(parts)), _p(self.state), c_int(k), _stream())
        _lib.check(rc, "wm_amp_found_inf")

    def update(self):
        rc = _lib.lib().wm_amp_update(_p(self.state), c_int(max(1, self.nopt)), _stream())
        _lib.check(rc, "wm_amp_update")

    def get_scale(self):
        return float(self.state[0].item())

    def step_count(self, k):
        return int(self.state[12 + k].item())

    def set_step_count(self, k, n):
        """restore optimiser k's step count (the t of Adam's bias correction under the scaler lives on the device, not on the host)"""
        self.state[12 + k] = float(n)

    def state_dict(self):
        """torch.amp.GradScaler.state_dict()'s keys (scale, growth_factor, backoff_factor, growth_interval, _growth_tracker)"""
        st = self.state.detach().cpu()
        return {"scale": float(st[0]), "growth_factor": float(st[2]), "backoff_factor": float(st[3]), "growth_interval": int(st[4]),
                "_growth_tracker": int(st[1])}

    def load_state_dict(self, sd):
        st = self.state.detach().cpu()
        st[0], st[1], st[2], st[3], st[4] = float(sd["scale"]), float(sd.get("_growth_tracker", 0)), float(sd["growth_factor"]), \
            float(sd["backoff_factor"]), float(sd["growth_interval"])
        st[8:12] = 0.0
        self.state.copy_(st)


def adam_step_amp(p, g, m, v, lr, beta1, beta2, eps, weight_decay, amp, k, decoupled=False, grad_scale=1.0):
    rc = _lib.lib().wm_adam_step_amp(_p(p), _p(g), _p(m), _p(v), c_size_t(p.numel()), c_float(lr), c_float(beta1), c_float(beta2), c_float(eps),
                                     c_float(weight_decay), c_int(1 if decoupled else 0), c_float(grad_scale), _p(amp.state), c_int(k), _stream())
    _lib.check(rc, "wm_adam_step_amp")
    _wrote(p, m, v)


def sumsq(x):
    n = x.numel()
    nparts = max(1, min(1024, (n + 4095) // 4096))
    part = torch.empty(nparts, device=x.device, dtype=torch.float32)
    rc = _lib.lib().wm_sumsq(_p(x), c_size_t(n), _p(part), c_int(nparts), _stream())
    _lib.check(rc, "wm_sumsq")
    return part


# ----------------------------------------------------------------------------- tamper-localisation branch
def clamp_quant(x):
    """round(255*clamp(x,0,1))/255: clamp_with_grad + Quantization (IRNcrop_model.py:320-322,344-345,372-373); backward = identity"""
    _need_cuda(x)
    x = x.contiguous().float()
    y = torch.empty_like(x)
    rc = _lib.lib().wm_clamp_quant_fwd(_p(x), _p(y), c_size_t(x.numel()), _stream())
    _lib.check(rc, "wm_clamp_quant_fwd")
    return y


def splice_fwd(enc, real=None, prev=None, mask=None, want_fwd=False):
    """q = clamp_quant(enc); tampered = q*(1-mask) + prev*mask (IRNcrop_model.py:344-348); with `real`, also the partial sums of
    the squared difference of the int-truncated images for the PSNR.  Returns (fwd_q or None, tampered or None, psnr partials or None)."""
    _need_cuda(enc)
    enc = enc.contiguous()
    B, C, H, W = enc.shape
    L = _lib.lib()
    fwd = torch.empty_like(enc) if want_fwd else None
    tam = torch.empty_like(enc) if prev is not None else None
    part = None
    if real is not None:
        part = torch.empty(L.wm_splice_nparts(c_size_t(enc.numel())), device=enc.device, dtype=torch.float64)
        real = real.contiguous()
    if prev is not None:
        prev = prev.contiguous(); mask = mask.contiguous()
        assert prev.shape == enc.shape and tuple(mask.shape) == (B, 1, H, W) and mask.dtype == torch.float32
    rc = L.wm_splice_fwd(_p(enc), _p(real), _p(prev), _p(mask), _p(fwd), _p(tam), _p(part), c_int(B), c_int(C), c_size_t(H * W), _stream())
    _lib.check(rc, "wm_splice_fwd")
    return fwd, tam, part


def psnr_gate(partials, n, threshold=33.0, w_below=1.0, w_above=0.8):
    """[2] f32 device tensor: (PSNR, forward-loss weight) -- IRNcrop_model.py:379-388, no host sync"""
    out = torch.empty(2, device=partials.device, dtype=torch.float32)
    rc = _lib.lib().wm_psnr_gate(_p(partials), c_int(partials.numel()), c_double(float(n)), c_float(threshold), c_float(w_below),
                                 c_float(w_above), _p(out), _stream())
    _lib.check(rc, "wm_psnr_gate")
    return out


def mse_fwd_bwd_gated(a, b, gscale, gate, gscale_dev=None):
    """mse_fwd_bwd with the gradient scale multiplied by the device scalar gate[0]"""
    a = a.contiguous(); b = b.contiguous()
    n = a.numel()
    nparts = max(1, min(1024, (n + 4095) // 4096))
    part = torch.empty(nparts, device=a.device, dtype=torch.float32)
    grad = torch.empty_like(a)
    rc = _lib.lib().wm_mse_fwd_bwd_gated(_p(a), _p(b), _p(grad), c_float(gscale), _p(gate), _p(gscale_dev), _p(part), c_int(nparts), c_size_t(n),
                                         _stream())
    _lib.check(rc, "wm_mse_fwd_bwd_gated")
    return part, grad


def bce_logits_target(p, target, gscale=1.0, want_grad=True, chain_sigmoid=False, gscale_dev=None):
    """BCEWithLogitsLoss(mean)(p, target) for tensors: (loss [1] device tensor, gscale * d loss / d p or None).
    chain_sigmoid: p is a sigmoid output s(z) and the gradient returned is wrt z."""
    _need_cuda(p, target)
    p = p.contiguous().float(); target = target.contiguous().float()
    assert p.numel() == target.numel()
    n = p.numel()
    nparts = max(1, min(1024, (n + 4095) // 4096))
    part = torch.empty(nparts, device=p.device, dtype=torch.float32)
    loss = torch.empty(1, device=p.device, dtype=torch.float32)
    grad = torch.empty_like(p) if want_grad else None
    rc = _lib.lib().wm_bce_logits_target(_p(p), _p(target), c_size_t(n), c_float(gscale), _p(gscale_dev), _p(part), c_int(nparts), _p(loss), _p(grad),
                                         c_int(1 if chain_sigmoid else 0), _stream())
    _lib.check(rc, "wm_bce_logits_target")
    return loss, grad


def masked_axpy_(a, g, mask):
    """a += g * (1 - mask), mask [B,1,H,W] broadcast over channels"""
    B, C, H, W = a.shape
    assert a.is_contiguous() and g.is_contiguous() and g.shape == a.shape and mask.is_contiguous() and tuple(mask.shape) == (B, 1, H, W)
    rc = _lib.lib().wm_masked_axpy(_p(a), _p(g), _p(mask), c_int(B), c_int(C), c_size_t(H * W), _stream())
    _lib.check(rc, "wm_masked_axpy")
    _wrote(a)
    return a


def mask_threshold(p, threshold=0.5):
    """uint8 tamper mask: p > threshold"""
    _need_cuda(p)
    p = p.contiguous().float()
    out = torch.empty(p.shape, device=p.device, dtype=torch.uint8)
    rc = _lib.lib().wm_mask_threshold(_p(p), c_float(threshold), _p(out), c_size_t(p.numel()), _stream())
    _lib.check(rc, "wm_mask_threshold")
    return out


def clip_grad_norm_(flats, max_norm, parts=None):
    """nn.utils.clip_grad_norm_ over the parameters of SEVERAL flat gradient buffers taken together (IRNcrop_model.py:410-412:
    netG.parameters() is one group), without a host sync.  Returns the [2] device tensor (clip coefficient, total norm)."""
    assert 1 <= len(flats) <= 4
    parts = parts if parts is not None else [sumsq(f) for f in flats]
    arr = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    ns = (ctypes.c_int * len(parts))(*[p.numel() for p in parts])
    out = torch.empty(2, device=flats[0].device, dtype=torch.float32)
    L = _lib.lib()
    rc = L.wm_clip_coef(arr, ns, c_int(len(parts)), c_float(float(max_norm)), _p(out), _stream())
    _lib.check(rc, "wm_clip_coef")
    for f in flats:
        rc = L.wm_scale_dev(_p(f), c_size_t(f.numel()), _p(out), _stream())
        _lib.check(rc, "wm_scale_dev")
    return out


# ----------------------------------------------------------------------------- per-kernel timing hook
# bench.py brackets launches of one named kernel family with events on the launch stream
# (torch's current stream == the hipStream_t handed to the C ABI).
_TIMER = None


class KernelTimer:
    def __init__(self, match):
        self.match = match  # callable(name, info) -> bool
        self.pairs = []     # (name, start event, end event)

    def elapsed_ms(self, name=None):
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for n, a, b in self.pairs if name is None or n == name]


def set_kernel_timer(timer):
    global _TIMER
    _TIMER = timer


def kernel_timer_installed():
    return _TIMER is not None


def _timed(name, info, launch):
    t = _TIMER
    if t is None or not t.match(name, info):
        return launch()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    r = launch()
    b.record()
    t.pairs.append((name, a, b))
    return r


# ----------------------------------------------------------------------------- stencil / resample attacks
BILINEAR, BICUBIC = 0, 1


def _planes(x):
    _need_cuda(x)
    assert x.dim() == 4 and x.dtype == torch.float32
    x = x.contiguous()
    B, C, H, W = x.shape
    return x, B * C, H, W


def stencil3(x, w9):
    x, N, H, W = _planes(x)
    y = torch.empty_like(x)
    rc = _lib.lib().wm_stencil3_fwd(_p(x), _p(y), c_int(N), c_int(H), c_int(W), _host_floats(w9), _stream())
    _lib.check(rc, "wm_stencil3_fwd")
    return y


def median_fwd(x, k, want_idx=True):
    x, N, H, W = _planes(x)
    y = torch.empty_like(x)
    idx = torch.empty(x.shape, device=x.device, dtype=torch.int8) if want_idx else None
    rc = _lib.lib().wm_median_fwd(_p(x), _p(y), _p(idx), c_int(N), c_int(H), c_int(W), c_int(k), _stream())
    _lib.check(rc, "wm_median_fwd")
    return y, idx


def median_bwd(gy, idx, k):
    gy, N, H, W = _planes(gy)
    gx = torch.empty_like(gy)
    rc = _lib.lib().wm_median_bwd(_p(gy), _p(idx), _p(gx), c_int(N), c_int(H), c_int(W), c_int(k), _stream())
    _lib.check(rc, "wm_median_bwd")
    return gx


def resample_fwd(x, rect, out_hw, kind, clamp01=False):
    """rect = (h0, hs, w0, ws) sub-rectangle of x resampled to out_hw."""
    x, N, H, W = _planes(x)
    h0, hs, w0, ws = rect
    OH, OW = out_hw
    y = torch.empty(x.shape[0], x.shape[1], OH, OW, device=x.device, dtype=torch.float32)
    rc = _lib.lib().wm_resample_fwd(_p(x), _p(y), c_int(N), c_int(H), c_int(W), c_int(h0), c_int(hs), c_int(w0), c_int(ws),
                                    c_int(OH), c_int(OW), c_int(kind), c_int(1 if clamp01 else 0), _stream())
    _lib.check(rc, "wm_resample_fwd")
    return y


def resample_bwd(gy, y_clamped, in_hw, rect, kind, separable=True):
    """the transpose of resample_fwd: separable (default) = two passes through a workspace [N,OH,W] (wm_resample_bwd_sep, 3x faster at the
    Resize attack's ratios); separable=False = the one-kernel gather form (wm_resample_bwd)"""
    gy, N, OH, OW = _planes(gy)
    H, W = in_hw
    h0, hs, w0, ws = rect
    gx = torch.empty(gy.shape[0], gy.shape[1], H, W, device=gy.device, dtype=torch.float32)
    yc = y_clamped.contiguous() if y_clamped is not None else None
    if separable:
        tmp = torch.empty(N, OH, W, device=gy.device, dtype=torch.float32)
        rc = _lib.lib().wm_resample_bwd_sep(_p(gy), _p(yc), _p(gx), _p(tmp), c_int(N), c_int(H), c_int(W), c_int(h0), c_int(hs), c_int(w0),
                                            c_int(ws), c_int(OH), c_int(OW), c_int(kind), _stream())
        _lib.check(rc, "wm_resample_bwd_sep")
        return gx
    rc = _lib.lib().wm_resample_bwd(_p(gy), _p(yc), _p(gx), c_int(N), c_int(H), c_int(W), c_int(h0), c_int(hs), c_int(w0),
                                    c_int(ws), c_int(OH), c_int(OW), c_int(kind), _stream())
    _lib.check(rc, "wm_resample_bwd")
    return gx


def quant(x):
    _need_cuda(x)
    x = x.contiguous().float()
    y = torch.empty_like(x)
    rc = _lib.lib().wm_quant_fwd(_p(x), _p(y), c_size_t(x.numel()), _stream())
    _lib.check(rc, "wm_quant_fwd")
    return y


# ----------------------------------------------------------------------------- UNet pieces
def bnrelu_maxpool2(y, scale, shift, C, act_out=None, act_c0=0):
    """y [B,H,W,ld] raw conv output -> pooled activated [B,H/2,W/2,C]; optionally writes the activated
    full-resolution map into act_out[..., act_c0:act_c0+C] (the skip half of a concat buffer)."""
    B, H, W, ld = y.shape
    pooled = torch.empty(B, H // 2, W // 2, C, device=y.device, dtype=y.dtype)
    rc = _lib.lib().wm_bnrelu_maxpool2(_p(y), c_int(ld), _p(scale), _p(shift), _p(pooled), c_int(C), _p(act_out),
                                       c_int(0 if act_out is None else act_out.shape[-1]), c_int(act_c0), c_int(B), c_int(H),
                                       c_int(W), c_int(C), c_int(dtype_id(y)), _stream())
    _lib.check(rc, "wm_bnrelu_maxpool2")
    return pooled


def maxpool2_bwd(y, scale, shift, gpooled, g_skip, g_skip_c0, C):
    """gradient wrt the activated full-resolution map: skip gradient (slice of a concat gradient) + pooled path."""
    B, H, W, ld = y.shape
    g = torch.empty(B, H, W, C, device=y.device, dtype=y.dtype)
    gs_ptr = None
    ldgs = 0
    if g_skip is not None:
        ldgs = g_skip.shape[-1]
        gs_ptr = ctypes.c_void_p(g_skip.data_ptr() + g_skip_c0 * g_skip.element_size())
    rc = _lib.lib().wm_maxpool2_bwd(_p(y), c_int(ld), _p(scale), _p(shift), _p(gpooled), c_int(gpooled.shape[-1]), gs_ptr,
                                    c_int(ldgs), _p(g), c_int(C), c_int(B), c_int(H), c_int(W), c_int(C), c_int(dtype_id(y)),
                                    _stream())
    _lib.check(rc, "wm_maxpool2_bwd")
    return g


def upconv2x2_mfma_supported(Cin, Cout, dtype):
    return bool(_lib.lib().wm_upconv2x2_mfma_supported(c_int(Cin), c_int(Cout), c_int(dt_id(dtype))))


def upconv2x2_pack(w, dtype=torch.bfloat16):
    """w [Cin,Cout,2,2] f32 -> (wf [4*Cout, Cin] rows (ij, co), wb [Cin, 4*Cout] columns (ij, co)) in the 16-bit activation dtype: the MFMA operands"""
    Cin, Cout = w.shape[0], w.shape[1]
    wf = torch.empty(4 * Cout, Cin, device=w.device, dtype=dtype)
    wb = torch.empty(Cin, 4 * Cout, device=w.device, dtype=dtype)
    rc = _lib.lib().wm_upconv2x2_pack(_p(w), _p(wf), _p(wb), c_int(Cin), c_int(Cout), c_int(dt_id(dtype)), _stream())
    _lib.check(rc, "wm_upconv2x2_pack")
    return wf, wb


def upconv2x2_fwd(x, scale, shift, w, bias, out, c0):
    """x [B,H,W,Cin] (raw + pending BN/ReLU) ; w [Cin,Cout,2,2] f32 -> writes out[B,2H,2W,ld] channels [c0,c0+Cout)."""
    B, H, W, ldx = x.shape
    Cin, Cout = w.shape[0], w.shape[1]
    if upconv2x2_mfma_supported(Cin, Cout, x.dtype):
        wf, _ = upconv2x2_pack(w, x.dtype)
        rc = _lib.lib().wm_upconv2x2_fwd_mfma(_p(x), c_int(ldx), _p(scale), _p(shift), _p(wf), _p(bias), _p(out), c_int(out.shape[-1]),
                                              c_int(c0), c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(Cout), c_int(dtype_id(x)), _stream())
        _lib.check(rc, "wm_upconv2x2_fwd_mfma")
        return out
    rc = _lib.lib().wm_upconv2x2_fwd(_p(x), c_int(ldx), _p(scale), _p(shift), _p(w), _p(bias), _p(out), c_int(out.shape[-1]),
                                     c_int(c0), c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(Cout), c_int(dtype_id(x)), _stream())
    _lib.check(rc, "wm_upconv2x2_fwd")
    return out


def upconv2x2_bwd(x, scale, shift, w, gy, c0, dw, dbias, accumulate):
    """returns gx [B,H,W,Cin]; writes dw [Cin,Cout,2,2], dbias [Cout]."""
    B, H, W, ldx = x.shape
    Cin, Cout = w.shape[0], w.shape[1]
    L = _lib.lib()
    if upconv2x2_mfma_supported(Cin, Cout, x.dtype):
        _, wb = upconv2x2_pack(w, x.dtype)
        gx = torch.empty(B, H, W, Cin, device=x.device, dtype=x.dtype)
        rc = L.wm_upconv2x2_dgrad_mfma(_p(gy), c_int(gy.shape[-1]), c_int(c0), _p(wb), _p(gx), c_int(Cin), c_int(B), c_int(H), c_int(W),
                                       c_int(Cin), c_int(Cout), c_int(dtype_id(x)), _stream())
        _lib.check(rc, "wm_upconv2x2_dgrad_mfma")
        ns = L.wm_upconv2x2_wgrad_nsplit(c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(Cout))
        part = torch.empty(ns, Cin, 4 * Cout, device=x.device, dtype=torch.float32)
        bpart = torch.empty(ns, 4 * Cout, device=x.device, dtype=torch.float32)
        rc = L.wm_upconv2x2_wgrad_mfma(_p(x), c_int(ldx), _p(scale), _p(shift), _p(gy), c_int(gy.shape[-1]), c_int(c0), _p(part), _p(bpart),
                                       _p(dw), _p(dbias), c_int(1 if accumulate else 0), c_int(B), c_int(H), c_int(W), c_int(Cin),
                                       c_int(Cout), c_int(dtype_id(x)), _stream())
        _lib.check(rc, "wm_upconv2x2_wgrad_mfma")
        return gx
    chunks = L.wm_upconv2x2_dw_chunks(c_int(B), c_int(H), c_int(W))
    N = 4 * Cout
    part = torch.empty(chunks, (Cin + 1) * N, device=x.device, dtype=torch.float32)
    w_t = w.permute(2, 3, 1, 0).reshape(N, Cin).contiguous()
    gx = torch.empty(B, H, W, Cin, device=x.device, dtype=x.dtype)
    rc = L.wm_upconv2x2_bwd(_p(x), c_int(ldx), _p(scale), _p(shift), _p(w_t), _p(gy), c_int(gy.shape[-1]), c_int(c0), _p(gx),
                            c_int(Cin), _p(part), c_int(B), c_int(H), c_int(W), c_int(Cin), c_int(Cout), c_int(dtype_id(x)), _stream())
    _lib.check(rc, "wm_upconv2x2_bwd")
    red = torch.empty((Cin + 1) * N, device=x.device, dtype=torch.float32)
    colsum(part, (Cin + 1) * N, (Cin + 1) * N, red, False)
    gw = red[:Cin * N].view(Cin, Cout, 2, 2)
    gb = red[Cin * N:].view(Cout, 4).sum(1)
    if accumulate:
        dw += gw
        dbias += gb
    else:
        dw.copy_(gw)
        dbias.copy_(gb)
    return gx


# ----------------------------------------------------------------------------- DiffJPEG
ROUND, ROUND_ONLY_AT_0, DIFF_ROUND = 0, 1, 2


def diffjpeg_fwd(x, rounding, factor):
    _need_cuda(x)
    assert x.dim() == 4 and x.shape[1] == 3 and x.dtype == torch.float32
    x = x.contiguous()
    y = torch.empty_like(x)
    B, _, H, W = x.shape
    rc = _lib.lib().wm_diffjpeg_fwd(_p(x), _p(y), c_int(B), c_int(H), c_int(W), c_int(rounding), c_float(factor), _stream())
    _lib.check(rc, "wm_diffjpeg_fwd")
    return y


def diffjpeg_bwd(x, gy, rounding, factor):
    _need_cuda(x, gy)
    x = x.contiguous(); gy = gy.contiguous()
    gx = torch.empty_like(gy)
    B, _, H, W = x.shape
    rc = _lib.lib().wm_diffjpeg_bwd(_p(x), _p(gy), _p(gx), c_int(B), c_int(H), c_int(W), c_int(rounding), c_float(factor), _stream())
    _lib.check(rc, "wm_diffjpeg_bwd")
    return gx


# ----------------------------------------------------------------------------- general layer family (include/wm_hip.h)
ACT_KINDS = {"relu": 0, "lrelu": 1, "gelu": 2, "elu": 3, "sigmoid": 4, "tanh": 5}
ACT_BWD_KINDS = dict(ACT_KINDS, elu_out=6)   # backward only: the ELU derivative from the layer's output (conv3x3_fwd_elu keeps no pre-activation)


def cpad(c):
    """channel stride of an NHWC activation with c real channels"""
    return (int(c) + 15) // 16 * 16


def _nhwc(x):
    _need_cuda(x)
    if x.dim() != 4 or not x.is_contiguous() or x.shape[3] % 16:
        raise ValueError(f"expected a contiguous NHWC tensor with a channel stride that is a multiple of 16, got {tuple(x.shape)}")
    return x


def gconv_pack(w, rows, cols, transpose, dtype):
    """w [Cout,Cin,KH,KW] f32 -> [KH*KW][rows][cols] of dtype (rows = Cout / cols = Cin, or swapped with transpose)"""
    _need_cuda(w)
    assert w.dim() == 4 and w.dtype == torch.float32
    w = w.contiguous()
    Cout, Cin, KH, KW = w.shape
    wp = torch.empty(KH * KW, rows, cols, device=w.device, dtype=dtype)
    rc = _lib.lib().wm_gconv_pack(_p(w), _p(wp), c_int(Cout), c_int(Cin), c_int(KH), c_int(KW), c_int(rows), c_int(cols), c_int(1 if transpose else 0),
                                  c_int(dt_id(dtype)), _stream())
    _lib.check(rc, "wm_gconv_pack")
    return wp


def gconv_fwd(x, wp, bias, out_hw, KH, KW, stride, pad, dgrad=False):
    """x [B,IH,IW,KC]; wp [taps][NC][KC] (gconv_pack); bias f32 [NC] or None; -> [B,OH,OW,NC]"""
    x = _nhwc(x)
    B, IH, IW, KC = x.shape
    taps, NC, KC2 = wp.shape
    if taps != KH * KW or KC2 != KC or wp.dtype != x.dtype:
        raise ValueError(f"packed filter {tuple(wp.shape)} {wp.dtype} does not fit the input {tuple(x.shape)} {x.dtype} / {KH}x{KW}")
    if bias is not None and (bias.numel() != NC or bias.dtype != torch.float32):
        raise ValueError("bias must be f32 with one entry per (padded) output channel")
    OH, OW = out_hw
    out = torch.empty(B, OH, OW, NC, device=x.device, dtype=x.dtype)
    rc = _lib.lib().wm_gconv_fwd(_p(x), _p(wp), _p(bias), _p(out), c_int(B), c_int(IH), c_int(IW), c_int(KC), c_int(OH), c_int(OW), c_int(NC),
                                 c_int(KH), c_int(KW), c_int(stride), c_int(pad), c_int(1 if dgrad else 0), c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_gconv_fwd")
    return out


def gconv_wgrad(dout, x, Cout, Cin, KH, KW, stride, pad, want_bias=True, dw_acc=None, db_acc=None):
    """dw [Cout,Cin,KH,KW] f32, dbias [Cout] f32 (or None) of the conv x [B,IH,IW,KC] -> dout [B,OH,OW,NC].
    dw_acc (and db_acc when a bias gradient is wanted): contiguous f32 tensors of those shapes the results are ADDED to instead"""
    dout, x = _nhwc(dout), _nhwc(x)
    B, OH, OW, NC = dout.shape
    _, IH, IW, KC = x.shape
    if x.shape[0] != B or x.dtype != dout.dtype:
        raise ValueError("gconv_wgrad: operands disagree")
    L = _lib.lib()
    L.wm_gconv_wgrad_scratch_floats.restype = c_size_t
    partial = torch.empty(L.wm_gconv_wgrad_scratch_floats(c_int(B), c_int(OH), c_int(OW), c_int(KC), c_int(NC), c_int(KH), c_int(KW)), device=x.device,
                          dtype=torch.float32)
    acc = dw_acc is not None
    if acc and (not dw_acc.is_contiguous() or dw_acc.numel() != Cout * Cin * KH * KW or dw_acc.dtype != torch.float32 or
                (want_bias and (db_acc is None or not db_acc.is_contiguous() or db_acc.numel() != Cout or db_acc.dtype != torch.float32))):
        raise ValueError("gconv_wgrad: accumulation targets do not match the gradients")
    dw = dw_acc if acc else torch.empty(Cout, Cin, KH, KW, device=x.device, dtype=torch.float32)
    db = (db_acc if acc else torch.empty(Cout, device=x.device, dtype=torch.float32)) if want_bias else None
    rc = _lib.lib().wm_gconv_wgrad(_p(dout), _p(x), _p(partial), _p(dw), _p(db), c_int(1 if acc else 0), c_int(B), c_int(IH), c_int(IW), c_int(KC), c_int(OH),
                                   c_int(OW), c_int(NC), c_int(KH), c_int(KW), c_int(stride), c_int(pad), c_int(Cout), c_int(Cin),
                                   c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_gconv_wgrad")
    return dw, db


def gcolsum(x, creal, out_acc=None):
    """column sums [creal] f32 of x [.., C]; out_acc: a contiguous f32 [creal] tensor they are ADDED to instead"""
    x = _nhwc(x)
    C = x.shape[3]
    if out_acc is not None and (not out_acc.is_contiguous() or out_acc.numel() != creal or out_acc.dtype != torch.float32):
        raise ValueError("gcolsum: accumulation target does not match")
    out = out_acc if out_acc is not None else torch.empty(creal, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    L.wm_gcolsum_scratch_floats.restype = c_size_t
    scratch = torch.empty(L.wm_gcolsum_scratch_floats(c_size_t(x.numel() // C), c_int(C)), device=x.device, dtype=torch.float32)
    rc = L.wm_gcolsum(_p(x), c_size_t(x.numel() // C), c_int(C), _p(out), c_int(creal), c_int(0 if out_acc is None else 1), _p(scratch), c_int(dt_id(x.dtype)),
                      _stream())
    _lib.check(rc, "wm_gcolsum")
    return out


def unary_fwd(x, kind):
    _need_cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    rc = _lib.lib().wm_unary_fwd(_p(x), _p(y), c_size_t(x.numel()), c_int(ACT_KINDS[kind]), c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_unary_fwd")
    return y


def unary_bwd(x, gy, kind):
    _need_cuda(x, gy)
    gy = gy.contiguous()
    gx = torch.empty_like(x)
    rc = _lib.lib().wm_unary_bwd(_p(x), _p(gy), _p(gx), c_size_t(x.numel()), c_int(ACT_KINDS[kind]), c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_unary_bwd")
    return gx


def unary_bwd_colsum(x, gy, kind, creal, db_acc=None):
    """(gx, db): gx = gy * act'(x) and db [creal] = the column sums of gx (the bias gradient of the convolution under the activation)
    in one pass over the data (csrc/gelem.hip); NHWC, channel stride a multiple of 16.  db_acc: added to instead (contiguous f32 [creal])"""
    _need_cuda(x, gy)
    x, gy = _nhwc(x), gy.contiguous()
    C = x.shape[3]
    npix = x.numel() // C
    gx = torch.empty_like(x)
    if db_acc is not None and (not db_acc.is_contiguous() or db_acc.numel() != creal or db_acc.dtype != torch.float32):
        raise ValueError("unary_bwd_colsum: accumulation target does not match")
    db = db_acc if db_acc is not None else torch.empty(creal, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    L.wm_unary_bwd_colsum_scratch_floats.restype = c_size_t
    part = torch.empty(L.wm_unary_bwd_colsum_scratch_floats(c_size_t(npix), c_int(C)), device=x.device, dtype=torch.float32)
    rc = L.wm_unary_bwd_colsum(_p(x), _p(gy), _p(gx), c_size_t(npix), c_int(C), c_int(ACT_BWD_KINDS[kind]), _p(part), _p(db), c_int(creal),
                               c_int(0 if db_acc is None else 1),
                               c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_unary_bwd_colsum")
    return gx, db


def add_scaled(a, b, alpha=1.0):
    _need_cuda(a, b)
    if a.shape != b.shape or a.dtype != b.dtype:
        raise ValueError("add_scaled: operands disagree")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    rc = _lib.lib().wm_add_scaled(_p(a), _p(b), _p(out), c_size_t(a.numel()), c_float(alpha), c_int(dt_id(a.dtype)), _stream())
    _lib.check(rc, "wm_add_scaled")
    return out


def qfatt_fwd(x, res, gamma, beta):
    """x + gamma[b,c] * res + beta[b,c]; gamma / beta f32 [B, >= C]"""
    x, res = _nhwc(x), _nhwc(res)
    B, H, W, C = x.shape
    gamma, beta = gamma.contiguous(), beta.contiguous()
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.shape == beta.shape and gamma.shape[0] == B
    out = torch.empty_like(x)
    rc = _lib.lib().wm_qfatt_fwd(_p(x), _p(res), _p(gamma), _p(beta), _p(out), c_int(B), c_size_t(H * W), c_int(C), c_int(gamma.shape[1]),
                                 c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_qfatt_fwd")
    return out


def qfatt_bwd(g, res, gamma):
    g, res = _nhwc(g.contiguous()), _nhwc(res)
    B, H, W, C = g.shape
    gamma = gamma.contiguous()
    gres = torch.empty_like(g)
    gg = torch.zeros_like(gamma)
    gb = torch.zeros_like(gamma)
    L = _lib.lib()
    L.wm_qfatt_bwd_scratch_floats.restype = c_size_t
    scratch = torch.empty(L.wm_qfatt_bwd_scratch_floats(c_int(B), c_size_t(H * W), c_int(gamma.shape[1])), device=g.device, dtype=torch.float32)
    rc = L.wm_qfatt_bwd(_p(g), _p(res), _p(gamma), _p(gres), _p(gg), _p(gb), _p(scratch), c_int(B), c_size_t(H * W), c_int(C), c_int(gamma.shape[1]),
                        c_int(dt_id(g.dtype)), _stream())
    _lib.check(rc, "wm_qfatt_bwd")
    return gres, gg, gb


def gpool_fwd(x):
    x = _nhwc(x)
    B, H, W, C = x.shape
    out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    rc = _lib.lib().wm_gpool_fwd(_p(x), _p(out), c_int(B), c_size_t(H * W), c_int(C), c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_gpool_fwd")
    return out


def gpool_bwd(g, shape, dtype):
    _need_cuda(g)
    B, H, W, C = shape
    g = g.contiguous().float()
    gx = torch.empty(B, H, W, C, device=g.device, dtype=dtype)
    rc = _lib.lib().wm_gpool_bwd(_p(g), _p(gx), c_int(B), c_size_t(H * W), c_int(C), c_int(dt_id(dtype)), _stream())
    _lib.check(rc, "wm_gpool_bwd")
    return gx


PAD_SYMMETRIC, PAD_REPLICATE = 0, 1


def pad_nchw_to_nhwc(x, pads, mode, dtype):
    """x [B,C,H,W] f32 -> [B,H+top+bottom,W+left+right,cpad(C)] of dtype; pads = (left, right, top, bottom)"""
    _need_cuda(x)
    x = x.contiguous().float()
    B, C, H, W = x.shape
    l, r, t, b = pads
    out = torch.empty(B, H + t + b, W + l + r, cpad(C), device=x.device, dtype=dtype)
    rc = _lib.lib().wm_pad_nchw_to_nhwc(_p(x), _p(out), c_int(B), c_int(C), c_int(H), c_int(W), c_int(l), c_int(r), c_int(t), c_int(b), c_int(mode),
                                        c_int(out.shape[3]), c_int(dt_id(dtype)), _stream())
    _lib.check(rc, "wm_pad_nchw_to_nhwc")
    return out


def pad_nchw_to_nhwc_bwd(gp, shape, pads, mode):
    gp = _nhwc(gp.contiguous())
    B, C, H, W = shape
    l, r, t, b = pads
    gx = torch.empty(B, C, H, W, device=gp.device, dtype=torch.float32)
    rc = _lib.lib().wm_pad_nchw_to_nhwc_bwd(_p(gp), _p(gx), c_int(B), c_int(C), c_int(H), c_int(W), c_int(l), c_int(r), c_int(t), c_int(b), c_int(mode),
                                            c_int(gp.shape[3]), c_int(dt_id(gp.dtype)), _stream())
    _lib.check(rc, "wm_pad_nchw_to_nhwc_bwd")
    return gx


def gunpack_nchw(x, C, H, W):
    x = _nhwc(x)
    B, PH, PW, CP = x.shape
    out = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    rc = _lib.lib().wm_gunpack_nchw(_p(x), _p(out), c_int(B), c_int(C), c_int(H), c_int(W), c_int(PH), c_int(PW), c_int(CP), c_int(dt_id(x.dtype)),
                                    _stream())
    _lib.check(rc, "wm_gunpack_nchw")
    return out


def gunpack_nchw_bwd(g, shape, dtype):
    _need_cuda(g)
    g = g.contiguous().float()
    B, C, H, W = g.shape
    _, PH, PW, CP = shape
    gx = torch.empty(B, PH, PW, CP, device=g.device, dtype=dtype)
    rc = _lib.lib().wm_gunpack_nchw_bwd(_p(g), _p(gx), c_int(B), c_int(C), c_int(H), c_int(W), c_int(PH), c_int(PW), c_int(CP), c_int(dt_id(dtype)),
                                        _stream())
    _lib.check(rc, "wm_gunpack_nchw_bwd")
    return gx


def spectral_norm_fwd(w, u, v, do_iter):
    """w [Cout, ...] f32; u [Cout], v [N] updated in place when do_iter; -> (w / sigma, sigma [1])"""
    _need_cuda(w, u, v)
    w = w.contiguous()
    M, N = w.shape[0], w.numel() // w.shape[0]
    assert u.numel() == M and v.numel() == N and u.is_contiguous() and v.is_contiguous() and u.dtype == torch.float32 and v.dtype == torch.float32
    sigma = torch.empty(1, device=w.device, dtype=torch.float32)
    wsn = torch.empty_like(w)
    L = _lib.lib()
    L.wm_spectral_norm_scratch_floats.restype = c_size_t
    scratch = torch.empty(L.wm_spectral_norm_scratch_floats(c_int(M), c_int(N)), device=w.device, dtype=torch.float32)
    rc = L.wm_spectral_norm_fwd(_p(w), _p(u), _p(v), _p(sigma), _p(wsn), _p(scratch), c_int(M), c_int(N), c_int(1 if do_iter else 0), _stream())
    _lib.check(rc, "wm_spectral_norm_fwd")
    return wsn, sigma


def spectral_norm_bwd(g, wsn, u, v, sigma):
    _need_cuda(g)
    g = g.contiguous()
    M, N = g.shape[0], g.numel() // g.shape[0]
    partial = torch.empty(256, device=g.device, dtype=torch.float32)
    gw = torch.empty_like(g)
    rc = _lib.lib().wm_spectral_norm_bwd(_p(g), _p(wsn), _p(u), _p(v), _p(sigma), _p(partial), _p(gw), c_int(M), c_int(N), c_int(0), _stream())
    _lib.check(rc, "wm_spectral_norm_bwd")
    return gw


def bayar_constrain_(w):
    """the Bayar constraint on w [Co,Ci,5,5] in place (conditional_jpeg_generator.py:814-817)"""
    _need_cuda(w)
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape[-2:] == (5, 5)
    rc = _lib.lib().wm_bayar_constrain(_p(w), c_int(w.shape[0] * w.shape[1]), _stream())
    _lib.check(rc, "wm_bayar_constrain")
    return w


# ----------------------------------------------------------------------------- invertible embedder pieces
def haar(x, C, fac, up, by_wavelet=False):
    """up False: [B,2H,2W,cpad(C)] -> [B,H,W,cpad(4C)] (analysis); up True: [B,H,W,cpad(4C)] -> [B,2H,2W,cpad(C)] (synthesis).
    by_wavelet: the 4C channels in wavelet-major order k*C + c (HaarDownsampling(order_by_wavelet=True)) instead of 4*c + k"""
    x = _nhwc(x)
    B, XH, XW, CPin = x.shape
    if up:
        if CPin < 4 * C:
            raise ValueError(f"haar synthesis of {C} channels needs {4 * C} input channels, stride is {CPin}")
        H, W = XH, XW
        out = torch.empty(B, 2 * H, 2 * W, cpad(C), device=x.device, dtype=x.dtype)
    else:
        if XH % 2 or XW % 2 or CPin < C:
            raise ValueError(f"haar analysis needs even height / width and {C} channels, got {tuple(x.shape)}")
        H, W = XH // 2, XW // 2
        out = torch.empty(B, H, W, cpad(4 * C), device=x.device, dtype=x.dtype)
    rc = _lib.lib().wm_haar(_p(x), _p(out), c_int(B), c_int(H), c_int(W), c_int(C), c_int(CPin), c_int(out.shape[3]), c_float(fac),
                            c_int((1 if up else 0) | (2 if by_wavelet else 0)), c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_haar")
    return out


def chan_copy_(dst, doff, src, soff, n):
    """dst[..., doff:doff+n] = src[..., soff:soff+n] (NHWC, same pixel grid)"""
    dst, src = _nhwc(dst), _nhwc(src)
    if dst.shape[:3] != src.shape[:3] or dst.dtype != src.dtype:
        raise ValueError("chan_copy_: pixel grids / dtypes disagree")
    npix = dst.shape[0] * dst.shape[1] * dst.shape[2]
    rc = _lib.lib().wm_chan_copy(_p(src), _p(dst), c_size_t(npix), c_int(src.shape[3]), c_int(soff), c_int(dst.shape[3]), c_int(doff), c_int(n),
                                 c_int(dt_id(dst.dtype)), _stream())
    _lib.check(rc, "wm_chan_copy")
    return dst


def chan_place(dstride, a, aoff, adst, na, b=None, boff=0, bdst=0, nb=0):
    """a new NHWC tensor [.., dstride], written whole by one launch: a[..., aoff:aoff+na] at channel adst, b[..., boff:boff+nb] (optional)
    at bdst, zero elsewhere"""
    a = _nhwc(a)
    if b is not None:
        b = _nhwc(b)
        if b.shape[:3] != a.shape[:3] or b.dtype != a.dtype:
            raise ValueError("chan_place: pixel grids / dtypes disagree")
    if dstride % 16:
        raise ValueError("chan_place: the channel stride must be a multiple of 16")
    out = torch.empty(*a.shape[:3], dstride, device=a.device, dtype=a.dtype)
    npix = a.shape[0] * a.shape[1] * a.shape[2]
    rc = _lib.lib().wm_chan_place(_p(a), c_int(a.shape[3]), c_int(aoff), c_int(adst), c_int(na), _p(b), c_int(b.shape[3] if b is not None else 0), c_int(boff),
                                  c_int(bdst), c_int(nb), _p(out), c_int(dstride), c_size_t(npix), c_int(dt_id(a.dtype)), _stream())
    _lib.check(rc, "wm_chan_place")
    return out


def coupling_fwd(x, s, t, clamp, eps, rev):
    _need_cuda(x, s, t)
    if not (x.shape == s.shape == t.shape and x.dtype == s.dtype == t.dtype):
        raise ValueError("coupling_fwd: operands disagree")
    x, s, t = x.contiguous(), s.contiguous(), t.contiguous()
    y = torch.empty_like(x)
    rc = _lib.lib().wm_coupling_fwd(_p(x), _p(s), _p(t), _p(y), c_size_t(x.numel()), c_float(clamp), c_float(eps), c_int(1 if rev else 0),
                                    c_int(dt_id(x.dtype)), _stream())
    _lib.check(rc, "wm_coupling_fwd")
    return y


def coupling_bwd(g, v, s, clamp, eps, rev):
    _need_cuda(g, v, s)
    g = g.contiguous()
    gx, gs, gt = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
    rc = _lib.lib().wm_coupling_bwd(_p(g), _p(v), _p(s), _p(gx), _p(gs), _p(gt), c_size_t(g.numel()), c_float(clamp), c_float(eps),
                                    c_int(1 if rev else 0), c_int(dt_id(g.dtype)), _stream())
    _lib.check(rc, "wm_coupling_bwd")
    return gx, gs, gt


# ----------------------------------------------------------------------------- losses of the literal IRNrhi step
def _nparts(n):
    return max(1, min(1024, (n + 4095) // 4096))


def smooth_l1(a, b, beta=1.0, want_grad=True):
    """nn.SmoothL1Loss()(a, b) -> (loss [1] device scalar, d loss / d a or None)"""
    _need_cuda(a, b)
    a, b = a.contiguous().float(), b.contiguous().float()
    if a.shape != b.shape:
        raise ValueError("smooth_l1: shapes disagree")
    n = a.numel()
    part = torch.empty(_nparts(n), device=a.device, dtype=torch.float32)
    loss = torch.empty(1, device=a.device, dtype=torch.float32)
    grad = torch.empty_like(a) if want_grad else None
    rc = _lib.lib().wm_smooth_l1(_p(a), _p(b), c_size_t(n), c_float(beta), _p(part), c_int(part.numel()), _p(loss), _p(grad), _stream())
    _lib.check(rc, "wm_smooth_l1")
    return loss, grad


def bce_prob(p, target, want_grad=True):
    """nn.BCELoss()(p, full_like(p, target)) -> (loss [1], d loss / d p or None)"""
    _need_cuda(p)
    p = p.contiguous().float()
    n = p.numel()
    part = torch.empty(_nparts(n), device=p.device, dtype=torch.float32)
    loss = torch.empty(1, device=p.device, dtype=torch.float32)
    grad = torch.empty_like(p) if want_grad else None
    rc = _lib.lib().wm_bce_prob(_p(p), c_float(target), c_size_t(n), _p(part), c_int(part.numel()), _p(loss), _p(grad), _stream())
    _lib.check(rc, "wm_bce_prob")
    return loss, grad


def cross_entropy(logits, labels, want_grad=True):
    """nn.CrossEntropyLoss()(logits [B,K] f32, labels [B] int64) -> (loss [1], d loss / d logits or None)"""
    _need_cuda(logits, labels)
    logits = logits.contiguous().float()
    labels = labels.contiguous()
    if logits.dim() != 2 or labels.dtype != torch.int64 or labels.numel() != logits.shape[0]:
        raise ValueError("cross_entropy: logits [B,K] f32 and labels [B] int64 expected")
    B, K = logits.shape
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    grad = torch.empty_like(logits) if want_grad else None
    rc = _lib.lib().wm_cross_entropy(_p(logits), _p(labels), c_int(B), c_int(K), c_int(K), _p(loss), _p(grad), _stream())
    _lib.check(rc, "wm_cross_entropy")
    return loss, grad


def clamp01_fwd(x):
    _need_cuda(x)
    x = x.contiguous().float()
    y = torch.empty_like(x)
    rc = _lib.lib().wm_clamp01_fwd(_p(x), _p(y), c_size_t(x.numel()), _stream())
    _lib.check(rc, "wm_clamp01_fwd")
    return y


def clamp01_bwd(x, g):
    _need_cuda(x, g)
    g = g.contiguous().float()
    gx = torch.empty_like(g)
    rc = _lib.lib().wm_clamp01_bwd(_p(x), _p(g), _p(gx), c_size_t(g.numel()), _stream())
    _lib.check(rc, "wm_clamp01_bwd")
    return gx


def psnr255(a, b):
    """PSNR of int(255 a) against int(255 b) (metrics.py:30-46 on postprocess()ed images) -> [1] device scalar (0 when equal)"""
    _need_cuda(a, b)
    a, b = a.contiguous().float(), b.contiguous().float()
    n = a.numel()
    part = torch.empty(_nparts(n), device=a.device, dtype=torch.float64)
    rc = _lib.lib().wm_psnr255_partials(_p(a), _p(b), c_size_t(n), _p(part), c_int(part.numel()), _stream())
    _lib.check(rc, "wm_psnr255_partials")
    return psnr_gate(part, n)[0:1]


def scale_dev_(x, scale_dev):
    """x *= scale_dev[0] (a device scalar)"""
    _need_cuda(x, scale_dev)
    assert x.is_contiguous() and x.dtype == torch.float32 and scale_dev.dtype == torch.float32
    rc = _lib.lib().wm_scale_dev(_p(x), c_size_t(x.numel()), _p(scale_dev), _stream())
    _lib.check(rc, "wm_scale_dev")
    _wrote(x)
    return x
