"""GPU: SURVEY 8f row 2 -- the invertible watermark embedder (models/invertible_net.py Inveritible_Decolorization_PAMI) on the HIP
layer family (csrc/inn.hip for the Haar transforms / coupling affine / channel moves, csrc/gconv.hip for the subnets' convolutions).

  * per kernel: Haar analysis / synthesis against the oracle's butterflies (and each as the other's adjoint), the coupling affine
    and its three gradients in both directions against autograd, channel slice / cat;
  * the network, f32: forward, rev=True (recovered, out_middle), input and every parameter gradient against tests/golden/f2.npz,
    which the REFERENCE's classes generated (make_golden.py gen_f2), for ResBlock and DenseBlock subnets;
  * properties at a full-size clip frame (256x256): rev(forward(x)) == x to fp32 round-off, and the zero-initialised embedder is
    the identity, in f32; bf16 against the oracle at the 16-bit bound.
"""
import numpy as np
import pytest
import torch

import detgen
from oracle import f2_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def _mods():
    from video_watermarking_forgery_detection_amd.models.invertible_net import Inveritible_Decolorization_PAMI, ResBlock, DenseBlock
    return Inveritible_Decolorization_PAMI, ResBlock, DenseBlock


def test_haar_coupling_and_channel_kernels():
    from video_watermarking_forgery_detection_amd import glayers as G
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 1e-2)):
        for C in (3, 4, 16, 20):
            x = detgen.normal((2, C, 12, 20), C)
            xd = x.to(DEV).requires_grad_(True)
            lo = G.haar_down(G.to_nhwc(xd, dtype), C, 0.5)
            assert lo.shape == (2, 6, 10, G.cpad(4 * C))
            y = G.to_nchw(lo, 4 * C)
            assert rel(y, f2_ref.haar_analysis(x.to(dtype).float(), 0.5)) < tol
            if 4 * C < lo.shape[3]:
                assert float(lo.detach()[..., 4 * C:].abs().max()) == 0.0
            gy = detgen.normal(tuple(y.shape), 7)
            (y * gy.to(DEV)).sum().backward()
            assert rel(xd.grad, f2_ref.haar_synthesis(gy.to(dtype).float(), 0.5)) < tol       # the adjoint
            back = G.to_nchw(G.haar_up(lo.detach(), C, 0.5), C)
            assert rel(back, x) < (2e-6 if dtype == torch.float32 else 2e-2)                     # 0.5 / 0.5: orthonormal
    # coupling affine, both directions, against autograd of the definition
    for rev in (False, True):
        x, s, t = (detgen.normal((2, 5, 6, 16), i, std=1.5).requires_grad_(True) for i in (1, 2, 3))
        e = torch.exp(1.0 * (torch.sigmoid(s) * 2 - 1)) + 1e-4
        ref = (x - t) / e if rev else e * x + t
        g = detgen.normal((2, 5, 6, 16), 4)
        (ref * g).sum().backward()
        xd, sd_, td = (v.detach().to(DEV).requires_grad_(True) for v in (x, s, t))
        out = G.coupling(xd, sd_, td, 1.0, 1e-4, rev)
        (out * g.to(DEV)).sum().backward()
        assert rel(out, ref) < 2e-6
        assert rel(xd.grad, x.grad) < 5e-6 and rel(sd_.grad, s.grad) < 5e-6 and rel(td.grad, t.grad) < 5e-6
    # channel slice / cat
    a = detgen.normal((2, 3, 4, 32), 9).to(DEV).requires_grad_(True)
    s1, s2 = G.chan_slice(a, 0, 10), G.chan_slice(a, 10, 13)
    assert s1.shape[3] == 16 and torch.equal(s1[..., :10], a.detach()[..., :10]) and float(s1.detach()[..., 10:].abs().max()) == 0
    cat = G.chan_cat(s1, 10, s2, 13)
    assert cat.shape[3] == 32 and torch.equal(cat[..., :23], a.detach()[..., :23]) and float(cat.detach()[..., 23:].abs().max()) == 0
    cat.sum().backward()
    assert float(a.grad[..., :23].min()) == 1.0 and float(a.grad[..., 23:].abs().max()) == 0.0


def _check_param_grads(g, key, net, tol, stride=97):
    params = dict(net.named_parameters())
    n = 0
    for k in [f[len(key) + 3:] for f in g.files if f.startswith(key + "/g/")]:
        got = params[k].grad
        assert got is not None, k
        ref_norm = float(g[f"{key}/gnorm/{k}"])
        assert abs(got.norm().item() - ref_norm) <= tol * ref_norm + 1e-6, (k, got.norm().item(), ref_norm)
        d = np.abs(detgen.subsample(got.cpu(), stride).numpy() - g[f"{key}/g/{k}"]).max()
        assert d <= tol * max(np.abs(g[f"{key}/g/{k}"]).max(), ref_norm / max(got.numel(), 1) ** 0.5) + 1e-7, (k, d)
        n += 1
    return n


def test_haar_order_by_wavelet_against_reference_fixture(golden):
    """HaarDownsampling(order_by_wavelet=True, rebalance=0.7) (invertible_net.py:207-233): forward, rev and their gradients against
    tests/golden/f12x.npz; in 16 bit against the plain order permuted by the reference's index list"""
    from video_watermarking_forgery_detection_amd import glayers as G
    from video_watermarking_forgery_detection_amd.models.invertible_net import HaarDownsampling
    g = golden("f12x")
    net = HaarDownsampling([[3, 8, 12]], order_by_wavelet=True, rebalance=0.7).to(DEV)
    x = detgen.uniform((2, 3, 8, 12), 9700).to(DEV).requires_grad_(True)
    lo = net(G.to_nhwc(x, torch.float32))
    y = G.to_nchw(lo, 12)
    (y * detgen.normal(tuple(y.shape), 9701).to(DEV)).sum().backward()
    assert rel(y, g["haarw/y"]) < 1e-6 and rel(x.grad, g["haarw/gx"]) < 1e-6
    assert float(lo.detach()[..., 12:].abs().max()) == 0.0
    z = detgen.uniform((2, 12, 4, 6), 9702).to(DEV).requires_grad_(True)
    r = G.to_nchw(net(G.to_nhwc(z, torch.float32), rev=True), 3)
    (r * detgen.normal(tuple(r.shape), 9703).to(DEV)).sum().backward()
    assert rel(r, g["haarw/rev"]) < 1e-6 and rel(z.grad, g["haarw/rev_gx"]) < 1e-6
    for dtype in (torch.bfloat16, torch.float16):
        for C in (4, 20):
            perm = [i + 4 * j for i in range(4) for j in range(C)]
            xx = G.to_nhwc(detgen.normal((2, C, 12, 20), C).to(DEV), dtype)
            plain = G.to_nchw(G.haar_down(xx, C, 0.5), 4 * C)
            assert torch.equal(G.to_nchw(G.haar_down(xx, C, 0.5, True), 4 * C), plain[:, perm])


def test_embedder_against_reference_fixture(golden):
    PAMI, ResBlock, DenseBlock = _mods()
    g = golden("f2")
    net = detgen.fill_f2(PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock)).to(DEV).train()
    x = detgen.uniform((2, 4, 32, 32), 9500).to(DEV).requires_grad_(True)
    y = net(x)
    (y * detgen.normal(tuple(y.shape), 9501).to(DEV)).sum().backward()
    assert rel(y, g["pami/y"]) < 1e-4
    assert rel(x.grad, g["pami/gx"]) < 1e-3
    assert _check_param_grads(g, "pami", net, 2e-3) > 150
    net.zero_grad(set_to_none=True)
    z = detgen.uniform((2, 4, 32, 32), 9502).to(DEV).requires_grad_(True)
    r, mid = net(z, rev=True)
    assert tuple(mid.shape) == (2, 256, 4, 4)
    ((r * detgen.normal(tuple(r.shape), 9503).to(DEV)).sum() + 0.1 * (mid * detgen.normal(tuple(mid.shape), 9504).to(DEV)).sum()).backward()
    assert rel(r, g["pami_rev/y"]) < 1e-4 and rel(mid, g["pami_rev/mid"]) < 1e-4
    assert rel(z.grad, g["pami_rev/gx"]) < 1e-3
    assert _check_param_grads(g, "pami_rev", net, 2e-3) > 150
    with torch.no_grad():
        back, _ = net(net(x), rev=True)
    assert float((back - x.detach()).abs().max()) < 2e-5            # the reference's own round trip: tests/golden f2 pami/roundtrip_err = 1.7e-6

    net = detgen.fill_f2(PAMI(dims_in=[[3, 16, 16]], down_num=2, block_num=[1, 1], subnet_constructor=DenseBlock)).to(DEV).train()
    x = detgen.uniform((2, 3, 16, 16), 9600).to(DEV).requires_grad_(True)
    y = net(x)
    (y * detgen.normal(tuple(y.shape), 9601).to(DEV)).sum().backward()
    assert rel(y, g["dense/y"]) < 1e-4
    assert rel(x.grad, g["dense/gx"]) < 1e-3
    assert _check_param_grads(g, "dense", net, 2e-3) > 50


def test_embedder_properties_at_full_frame_size():
    PAMI, ResBlock, _ = _mods()
    # freshly constructed: every subnet's last conv is zero (invertible_net.py:354), so s = t = 0 and each coupling multiplies by
    # e(0) = 1 + 1e-4: the embedder is the identity up to that factor per coupling
    net = PAMI(dims_in=[[4, 256, 256]], block_num=[1, 1, 1], subnet_constructor=ResBlock).to(DEV)
    x = torch.rand(2, 4, 256, 256, device=DEV)
    with torch.no_grad():
        y = net(x)
    assert rel(y, x) < 2e-3
    detgen.fill_f2(net)
    with torch.no_grad():
        y = net(x)
        back, mid = net(y, rev=True)
    assert tuple(mid.shape) == (2, 256, 32, 32)
    assert torch.isfinite(y).all() and float((back - x).abs().max()) < 1e-4
    # bf16 against the oracle
    net16 = detgen.fill_f2(PAMI(dims_in=[[4, 64, 64]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=torch.bfloat16)).to(DEV)
    sd = f2_ref.params({k: v.cpu() for k, v in net16.state_dict().items()})
    x = detgen.uniform((2, 4, 64, 64), 3)
    with torch.no_grad():
        assert rel(net16(x.to(DEV)), f2_ref.pami(sd, x)) < 5e-2
    with pytest.raises(ValueError):
        net(torch.zeros(1, 4, 100, 100, device=DEV))
    with pytest.raises(RuntimeError, match="GPU only"):
        net(torch.zeros(1, 4, 64, 64))


def test_captured_step_replays_the_eager_step():
    """glayers.CapturedStep: forward + reverse + backward of the embedder captured into a hipGraph; a replay on new input (copied into
    the static tensor) leaves the same outputs and the same flat gradient as the eager step on that input."""
    from video_watermarking_forgery_detection_amd import glayers as G
    PAMI, ResBlock, _ = _mods()
    net = detgen.fill_f2(PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=torch.bfloat16)).to(DEV)
    opt = G.FlatAdamW(net, lr=1e-4)
    xs = torch.zeros(2, 4, 32, 32, device=DEV)

    def fwd_bwd():
        y = net(xs)
        back, mid = net(y, rev=True)
        loss = ((y - xs) ** 2).mean() + (back ** 2).mean() + (mid ** 2).mean()
        opt.zero_grad()
        loss.backward()
        return y, loss

    x1, x2 = detgen.uniform((2, 4, 32, 32), 5).to(DEV), detgen.uniform((2, 4, 32, 32), 6).to(DEV)
    xs.copy_(x1)
    step = G.CapturedStep(fwd_bwd)
    xs.copy_(x2)
    y_s, loss_s = step.replay()
    torch.cuda.synchronize()
    y2, g2, l2 = y_s.clone(), opt.grad.clone(), loss_s.clone()
    assert float(g2.abs().max()) > 0
    xs.copy_(x1)
    step.replay()                              # overwrites the static outputs and the flat gradient
    torch.cuda.synchronize()
    assert not torch.equal(opt.grad, g2)
    xs.copy_(x2)
    y_e, loss_e = fwd_bwd()                    # the eager step on x2
    torch.cuda.synchronize()
    assert torch.equal(y_e, y2) and torch.equal(opt.grad, g2) and torch.equal(loss_e.detach(), l2.detach())


@pytest.mark.parametrize("subnet", ["res", "dense"])
def test_coupling_subnets_on_two_streams_are_the_one_stream_step(subnet):
    """invertible_net.PARALLEL_SUBNETS: the s / t subnets of every coupling on two streams (forward forked by _pair, backward by autograd's
    per-node streams) -- outputs, the reverse pass and the whole flat gradient bit for bit the one-stream step's, enqueued and replayed from
    a hipGraph, over several steps with the optimiser in between (a block re-issued under a reader on the other stream would show here)."""
    from video_watermarking_forgery_detection_amd import glayers as G
    from video_watermarking_forgery_detection_amd.models import invertible_net as inn
    PAMI, ResBlock, DenseBlock = _mods()

    def run(par, graph):
        inn.PARALLEL_SUBNETS = par
        try:
            torch.manual_seed(1)
            net = detgen.fill_f2(PAMI(dims_in=[[4, 64, 64]], block_num=[2, 1, 1], subnet_constructor=ResBlock if subnet == "res" else DenseBlock,
                                      dtype=torch.bfloat16)).to(DEV)
            opt = G.FlatAdamW(net, lr=1e-3)
            xs = torch.zeros(3, 4, 64, 64, device=DEV)
            outs = {}

            def fwd_bwd():
                y = net(xs)
                back, mid = net(y, rev=True)
                loss = ((y - xs) ** 2).mean() + (back ** 2).mean() + (mid ** 2).mean()
                opt.zero_grad()
                loss.backward()
                return y, back, loss

            xs.copy_(detgen.uniform((3, 4, 64, 64), 40).to(DEV))
            step = G.CapturedStep(fwd_bwd) if graph else None
            got = []
            for i in range(4):
                xs.copy_(detgen.uniform((3, 4, 64, 64), 41 + i).to(DEV))
                y, back, loss = step.replay() if graph else fwd_bwd()
                torch.cuda.synchronize()
                got.append((y.detach().clone(), back.detach().clone(), loss.detach().clone(), opt.grad.clone()))
                opt.step()
            torch.cuda.synchronize()
            return got, [p.detach().clone() for p in net.parameters()]
        finally:
            inn.PARALLEL_SUBNETS = True

    base, pbase = run(False, False)
    assert float(base[0][3].abs().max()) > 0
    for par, graph in ((True, False), (True, True), (False, True)):
        got, params = run(par, graph)
        for i, (a, b) in enumerate(zip(base, got)):
            for name, u, v in zip(("y", "back", "loss", "grad"), a, b):
                assert torch.equal(u, v), (par, graph, i, name, float((u.float() - v.float()).abs().max()))
        assert all(torch.equal(u, v) for u, v in zip(pbase, params)), (par, graph)


def test_captured_step_refuses_an_fn_that_leaks_its_autograd_graph():
    """the fn shape that ended round 2's capture in a segmentation fault inside hipStreamEndCapture (gpurun_out/inn3.log): a first eager
    call on the default stream, an fn that rebinds outer names to tensors requiring grad (so every call's autograd graph lives until the
    next call has built its own), and a parameter whose gradient goes through autograd's AccumulateGrad node (then: every conv weight;
    now: the plain `gain` -- round 3 reproduced the crash with exactly this fn on round 2's CapturedStep, tools/dbg_capture.py 1 0 0 1 1 1).
    CapturedStep must answer with a Python error BEFORE capturing -- never crash --, and the same fn handing out detached tensors must
    capture and replay."""
    from video_watermarking_forgery_detection_amd import glayers as G
    PAMI, ResBlock, _ = _mods()
    net = PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=torch.bfloat16).to(DEV)
    opt = G.FlatAdamW(net, lr=1e-4)
    gain = torch.nn.Parameter(torch.ones(4, 1, 1, device=DEV))
    xs = detgen.uniform((2, 4, 32, 32), 7).to(DEV)
    out = {}

    def make(detach):
        def fwd_bwd():
            y = net(xs)
            back, _ = net(y, rev=True)
            loss = ((y - xs) ** 2).mean() + (back ** 2).mean() + ((y * gain) ** 2).mean()
            opt.zero_grad()
            if gain.grad is not None:
                gain.grad.zero_()
            loss.backward()
            out["y"], out["loss"] = (y.detach(), loss.detach()) if detach else (y, loss)
            return loss.detach()
        return fwd_bwd

    leaky = make(False)
    leaky()                                    # the eager call on the default stream that binds the outer names first
    with pytest.raises(RuntimeError, match="keeps the autograd graph"):
        G.CapturedStep(leaky)
    out.clear()
    clean = make(True)
    clean()
    step = G.CapturedStep(clean)
    step.replay()
    torch.cuda.synchronize()
    g_replay, gg_replay, l_replay = opt.grad.clone(), gain.grad.clone(), step.result.clone()
    assert float(gg_replay.abs().max()) > 0
    l_eager = clean()
    torch.cuda.synchronize()
    assert torch.equal(opt.grad, g_replay) and torch.equal(gain.grad, gg_replay) and torch.equal(l_eager, l_replay) and torch.equal(out["loss"], l_replay)


def test_pack_plan_trains_the_same_parameters():
    """glayers.set_pack_plan: persistent packed 3x3 weights re-packed by one launch after FlatAdamW.step give bit-identical training to
    packing per call; an in-place torch write to a weight is noticed (autograd version) and not served from the stale pack."""
    from video_watermarking_forgery_detection_amd import glayers as G, ops
    PAMI, ResBlock, _ = _mods()
    x = detgen.uniform((2, 4, 32, 32), 9).to(DEV)
    finals = []
    for use_plan in (False, True):
        net = detgen.fill_f2(PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=torch.bfloat16)).to(DEV)
        opt = G.FlatAdamW(net, lr=1e-3)
        plan = ops.PackPlan() if use_plan else None
        G.set_pack_plan(plan)
        try:
            for step in range(4):
                y = net(x)
                back, mid = net(y, rev=True)
                loss = ((y - x) ** 2).mean() + (back ** 2).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
                if step == 1:                  # an in-place write the plan did not make: the next forward must see it
                    with torch.no_grad():
                        net.operations_down[1].s1.conv2[0].weight.mul_(0.5)
            if use_plan:
                assert plan.valid and len(plan.packed) > 20
        finally:
            G.set_pack_plan(None)
        finals.append(opt.flat.clone())
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gradients_added_into_the_flat_buffer_match_autograd(dtype):
    """with a FlatAdamW the conv backward kernels add weight / bias gradients straight into the optimiser's flat gradient (and return
    None to autograd); without one autograd accumulates the returned tensors -- same sums (the embedder uses every weight twice)"""
    from video_watermarking_forgery_detection_amd import glayers as G
    PAMI, ResBlock, _ = _mods()
    x = detgen.uniform((2, 4, 32, 32), 11).to(DEV)

    def run(net):
        y = net(x)
        back, mid = net(y, rev=True)
        return ((y - x) ** 2).mean() + (back ** 2).mean() + (mid ** 2).mean()

    plain = detgen.fill_f2(PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=dtype)).to(DEV)
    run(plain).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in plain.parameters() if p.requires_grad])      # (the Haar filters are frozen parameters)
    flat = detgen.fill_f2(PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=dtype)).to(DEV)
    opt = G.FlatAdamW(flat, lr=1e-3)
    for _ in range(2):                      # twice: zero_grad really restarts the sums
        opt.zero_grad()
        run(flat).backward()
        assert rel(opt.grad, ref) < 1e-6
    del opt, flat                           # a collected optimiser must not leave live accumulation targets behind
    import gc
    gc.collect()
    again = detgen.fill_f2(PAMI(dims_in=[[4, 32, 32]], block_num=[1, 1, 1], subnet_constructor=ResBlock, dtype=dtype)).to(DEV)
    run(again).backward()
    assert rel(torch.cat([p.grad.reshape(-1) for p in again.parameters() if p.requires_grad]), ref) < 1e-6
