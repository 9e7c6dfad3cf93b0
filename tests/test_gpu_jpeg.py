"""GPU parity: fused block-JPEG HIP kernels vs the oracle (oracle/jpeg_ref.py) and the
golden vectors generated from the reference (tests/golden/jpeg.npz)."""
import numpy as np
import pytest
import torch

import detgen
from oracle import jpeg_ref

pytestmark = pytest.mark.gpu

MODES = {"Jpeg": ("round", 0), "JpegSS": ("ss", 1), "JpegMask": ("mask", 2)}


def tables_for(Q):
    lum, chroma = jpeg_ref.quant_tables(jpeg_ref.scale_factor(Q))
    return lum.flatten().tolist() + chroma.flatten().tolist()


def test_jpeg_golden(golden):
    from video_watermarking_forgery_detection_amd import ops
    g = golden("jpeg")
    keys = sorted({k.split("/")[0] for k in g.files if k.startswith("Jpeg")})
    nflip = 0
    for key in keys:
        kind, Q, size, sub = key.split("_")
        mname, mode = MODES[kind]
        Q = int(Q[1:]); sub = int(sub[1:])
        H, W = map(int, size.split("x"))
        seed = int(g[key + "/seed"])
        x = detgen.uniform((2, 3, H, W), seed).cuda()
        gy = detgen.normal((2, 3, H, W), seed + 5000).cuda()
        y = ops.jpeg_fwd(x, mode, tables_for(Q), sub)
        gx = ops.jpeg_bwd(x, gy, mode, tables_for(Q), sub)
        ref = g[key + "/y"]
        err = np.abs(y.cpu().numpy() - ref)
        if mname == "round":
            bad = err > 2e-4
            nflip += int(bad.any())
            assert bad.mean() < 0.02, (key, err.max())
        else:
            assert err.max() < 1e-4, (key, err.max())
        np.testing.assert_allclose(gx.cpu().numpy(), g[key + "/gx"], rtol=1e-3, atol=3e-4, err_msg=key)
    assert nflip <= 3


@pytest.mark.parametrize("mname", ["round", "ss", "mask"])
@pytest.mark.parametrize("shape", [(3, 3, 128, 128), (2, 3, 100, 200), (1, 3, 8, 8), (2, 3, 61, 75)])
def test_jpeg_vs_oracle(mname, shape):
    from video_watermarking_forgery_detection_amd import ops
    mode = {"round": 0, "ss": 1, "mask": 2}[mname]
    for Q, sub in ((50, 0), (90, 2), (30, 0)):
        x = detgen.uniform(shape, 11 + Q).requires_grad_(True)
        gy = detgen.normal(shape, 12 + Q)
        yr = jpeg_ref.jpeg_layer(x, Q, mname, sub)
        (yr * gy).sum().backward()
        y = ops.jpeg_fwd(x.detach().cuda(), mode, tables_for(Q), sub)
        gx = ops.jpeg_bwd(x.detach().cuda(), gy.cuda(), mode, tables_for(Q), sub)
        err = (y.cpu() - yr.detach()).abs()
        if mname == "round":
            assert (err > 2e-4).float().mean() < 0.01
        else:
            assert err.max() < 1e-4
        torch.testing.assert_close(gx.cpu(), x.grad, rtol=1e-3, atol=3e-4)


def test_jpeg_full_size_properties():
    """BASELINE size (B=16, 256x256): size-independent properties.
    mask mode is linear and idempotent (a projection); ss mode: blocks are independent."""
    from video_watermarking_forgery_detection_amd import ops
    x = detgen.uniform((16, 3, 256, 256), 5).cuda()
    z = detgen.uniform((16, 3, 256, 256), 6).cuda()
    m = lambda t: ops.jpeg_fwd(t, 2, None, 0)
    y = m(x)
    torch.testing.assert_close(m(y), y, rtol=0, atol=3e-4)  # idempotent up to the inexact yuv<->rgb inverse pair (jpeg.py:147-163)
    torch.testing.assert_close(m(x + z), y + m(z), rtol=0, atol=2e-5)            # linear
    # block independence: changing one 8x8 block changes only that block
    tb = tables_for(50)
    y0 = ops.jpeg_fwd(x, 1, tb, 0)
    x2 = x.clone(); x2[3, :, 64:72, 128:136] += 0.25
    y1 = ops.jpeg_fwd(x2, 1, tb, 0)
    d = (y1 - y0).abs()
    d[3, :, 64:72, 128:136] = 0
    assert d.max().item() == 0.0
    # adjoint test of the backward: <J v, w> == <v, J^T w> for the linear mask operator
    v, w = z, detgen.normal((16, 3, 256, 256), 7).cuda()
    lhs = (m(v) * w).sum().item()
    rhs = (v * ops.jpeg_bwd(None, w, 2, None, 0)).sum().item()
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def test_jpeg_errors():
    from video_watermarking_forgery_detection_amd import ops
    x = torch.zeros(1, 3, 8, 8, device="cuda")
    with pytest.raises(RuntimeError):
        ops.jpeg_fwd(x, 7, [1.0] * 128, 0)
    with pytest.raises(RuntimeError):
        ops.jpeg_fwd(x, 0, [1.0] * 128, 1)
    with pytest.raises(RuntimeError):
        ops.jpeg_fwd(torch.zeros(1, 3, 8, 8), 0, [1.0] * 128, 0)
