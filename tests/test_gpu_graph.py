"""Round 4: the HiDDeN-order training step replayed from a hipGraph (Hidden.enable_graph, hidden_models/hidden.py::_StepGraph) against the same
step enqueued eagerly -- bit for bit: every logged scalar of every step, the outputs, and after the last step every parameter, BatchNorm
buffer and Adam moment.  The reference runs one Python step per batch per rank (/root/reference/train.py:99-109, hidden.py:54-118); the
graph holds the very launches of that step, with fresh inputs copied into its static tensors and Adam's step-count-dependent constants
refreshed in device memory before each replay."""
import pytest
import torch

import detgen

pytestmark = pytest.mark.gpu


def _make(size, noise, dtype, keep_dead=True):
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    h = Hidden(HiDDenConfiguration(H=size, W=size), torch.device("cuda"), noise, None, compute_dtype=dtype, keep_dead_discriminator_grads=keep_dead)
    for m in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
        detgen.fill_module(m)
    return h


def _state(h):
    out = {}
    for k, m in (("E", h.encoder_decoder.encoder), ("Dec", h.encoder_decoder.decoder), ("D", h.discriminator)):
        for n, t in m.state_dict().items():
            out[f"{k}.{n}"] = t.detach().clone()
        out[f"{k}.grad"] = m.flat_grads.detach().clone()
    for k, o in (("optD", h.optimizer_discrim), ("optED", h.optimizer_enc_dec)):
        for i, (m, v) in enumerate(zip(o._m, o._v)):
            out[f"{k}.m{i}"], out[f"{k}.v{i}"] = m.detach().clone(), v.detach().clone()
        out[f"{k}.steps"] = torch.tensor(o.step_count)
    return out


@pytest.mark.parametrize("case", [("Jpeg50", torch.bfloat16, 64, 4, False), ("JpegSS70", torch.bfloat16, 64, 4, True), ("JpegMask50", torch.float32, 32, 2, True),
                                  ("Resize", torch.float16, 64, 4, True)])
def test_captured_step_equals_eager_bit_for_bit(case):
    from video_watermarking_forgery_detection_amd import noise_layers as NL, ops
    name, dt, S, B, keep = case

    def noise():
        if name == "Resize":
            class Fixed:   # Resize with its ratio pinned: the random draw is a host decision and would be baked into the graph
                capturable = True

                def __init__(self):
                    self.l = NL.Resize()
                def fwd(self, x):
                    return self.l.fwd(x, resize_ratio=0.7)
                def bwd(self, c, g):
                    return self.l.bwd(c, g)
            return Fixed()
        kind = "".join(c for c in name if not c.isdigit())
        return getattr(NL, kind)(int(name[len(kind):]))

    def amp():
        return ops.AmpState(torch.device("cuda")) if dt == torch.float16 else None

    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration

    def make():
        h = Hidden(HiDDenConfiguration(H=S, W=S), torch.device("cuda"), noise(), None, compute_dtype=dt, keep_dead_discriminator_grads=keep, amp=amp())
        for m in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
            detgen.fill_module(m)
        return h

    eager, graph = make(), make().enable_graph()
    NSTEP = 6       # graph: 2 eager warm-up steps, capture + replay at the 3rd, replays after
    for i in range(NSTEP):
        images = detgen.uniform((B, 3, S, S), 7000 + i).cuda()          # a fresh tensor every step: the graph's static inputs are refreshed by copy
        messages = detgen.bits((B, 30), 7100 + i).cuda()
        le, (ee, ne, de) = eager.train_on_batch([images, messages])
        lg, (eg, ng, dg) = graph.train_on_batch([images, messages])
        assert list(le.keys()) == list(lg.keys())
        for k in le:
            assert le[k] == lg[k], (i, k, le[k], lg[k])                  # floats from the same f32 device values: equality, not closeness
        assert torch.equal(ee, eg) and torch.equal(ne, ng) and torch.equal(de, dg), i
    g = next(iter(graph._graphs.values()))
    assert g.graph is not None and g.calls == NSTEP                      # the graph path really ran (3 eager-equivalent calls + replays)
    se, sg = _state(eager), _state(graph)
    assert se.keys() == sg.keys()
    for k in se:
        assert torch.equal(se[k], sg[k]), k
    # a call the graph cannot serve (a clip callable) falls back to the eager path and keeps the two models in step
    images = detgen.uniform((B, 3, S, S), 7050).cuda(); messages = detgen.bits((B, 30), 7150).cuda()
    seen = []
    le, _ = eager.train_on_batch([images, messages], clip=lambda flats: seen.append(len(flats)))
    lg, _ = graph.train_on_batch([images, messages], clip=lambda flats: seen.append(len(flats)))
    assert all(le[k] == lg[k] for k in le) and seen == [1, 2, 1, 2]
    # ... and the next plain call replays again, with the step counts the eager call advanced
    le, _ = eager.train_on_batch([images, messages])
    lg, _ = graph.train_on_batch([images, messages])
    assert all(le[k] == lg[k] for k in le)
    for k, v in _state(eager).items():
        assert torch.equal(v, _state(graph)[k]), k


def test_graph_mode_refuses_nothing_silently():
    """an attack layer without an explicit fwd / bwd (autograd inside the step) is served eagerly, not captured"""
    from video_watermarking_forgery_detection_amd import noise_layers as NL

    class Plain(torch.nn.Module):
        def forward(self, pair):
            return [pair[0] * 0.5 + 0.25, pair[1]]

    h = _make(32, Plain(), torch.float32).enable_graph()
    images = detgen.uniform((2, 3, 32, 32), 1).cuda(); messages = detgen.bits((2, 30), 2).cuda()
    for _ in range(4):
        h.train_on_batch([images, messages])
    assert h._graphs == {}
    # ... and so is a layer that draws its arguments on the host (Resize's ratio, Crop's rectangle): a capture would freeze one draw
    for layer in (NL.Resize(), NL.Crop(), NL.Combined([NL.Jpeg(50), NL.JpegSS(50)])):
        h = _make(32, layer, torch.float32).enable_graph()
        for _ in range(4):
            h.train_on_batch([images, messages])
        assert h._graphs == {}, type(layer).__name__


@pytest.mark.parametrize("case", [("Jpeg50", torch.bfloat16, 256, 16), ("Identity", torch.bfloat16, 128, 8), ("JpegSS50", torch.float32, 64, 4),
                                  ("Crop", torch.bfloat16, 64, 4)])
def test_two_chain_step_equals_one_stream_bit_for_bit(case):
    """Hidden.two_streams: the discriminator's passes and encoder -> attack -> decoder as two chains on two streams (hidden.py:54-118 has no
    dependency between them until the encoder's backward) -- the same launches, so the same bits, eagerly and replayed from a hipGraph; at the
    benchmark's full size too, where a missing cross-stream dependency would show (the kernels run for hundreds of microseconds)."""
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    name, dt, S, B = case

    def noise():
        if name == "Identity":
            return NL.Identity()
        if name == "Crop":
            class Fixed:
                capturable = True

                def __init__(self):
                    self.l = NL.Crop()
                def fwd(self, x):
                    H, W = x.shape[2], x.shape[3]
                    return self.l.fwd(x, apex=(H // 8, H // 8 + int(0.75 * H), W // 8, W // 8 + int(0.75 * W)))
                def bwd(self, c, g):
                    return self.l.bwd(c, g)
            return Fixed()
        kind = "".join(c for c in name if not c.isdigit())
        return getattr(NL, kind)(int(name[len(kind):]))

    one = _make(S, noise(), dt, keep_dead=False)
    two = _make(S, noise(), dt, keep_dead=False); two.two_streams = True
    twog = _make(S, noise(), dt, keep_dead=False).enable_graph(); twog.two_streams = True
    assert two.skip_zero_attack_gradient             # (default: Jpeg50's zero gradient is not computed -- on both paths -- and on two streams the encoder's backward joins chain A)
    for i in range(5):
        images = detgen.uniform((B, 3, S, S), 7300 + i).cuda()
        messages = detgen.bits((B, 30), 7400 + i).cuda()
        l1, o1 = one.train_on_batch([images, messages])
        for h in (two, twog):
            l2, o2 = h.train_on_batch([images, messages])
            for k in l1:
                assert l1[k] == l2[k], (i, k, l1[k], l2[k])
            assert all(torch.equal(a, b) for a, b in zip(o1, o2)), i
    s1 = _state(one)
    for h in (two, twog):
        s2 = _state(h)
        for k in s1:
            assert torch.equal(s1[k], s2[k]), k          # parameters, BatchNorm buffers, EVERY .grad buffer, Adam moments and step counts
    assert next(iter(twog._graphs.values())).graph is not None


def test_zero_attack_gradient_shortcut_changes_nothing_but_a_summation_order():
    """Hidden.skip_zero_attack_gradient on the one-stream path: Jpeg(Q)'s torch.round passes back zeros (reference noise_layers/jpeg.py:226-240),
    so the decoder's gradient wrt its input, the attack's backward and the addition of its zeros are not launched.  Every loss and output is
    bit-identical; so is every parameter, .grad and optimiser state EXCEPT the decoder's first-layer weight gradient, which a different
    kernel now sums (the weight-gradient-only kernel instead of the one-pass input + weight gradient kernel): the same sum in another order,
    equal to f32 round-off (asserted at 1e-5 of the tensor's scale).  JpegSS (a real gradient) is never short-cut."""
    from video_watermarking_forgery_detection_amd import noise_layers as NL, ops
    S, B = 64, 4
    for layer, zero in ((NL.Jpeg(50), True), (NL.JpegSS(50), False), (NL.Combined([NL.JpegSS(50), NL.Jpeg(70)]), True)):
        full = _make(S, layer, torch.bfloat16); full.skip_zero_attack_gradient = False
        short = _make(S, layer, torch.bfloat16)
        full.noise_id = short.noise_id = 1
        calls = {"n": 0}
        orig = ops.jpeg_bwd

        def counting(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        images = detgen.uniform((B, 3, S, S), 7500).cuda(); messages = detgen.bits((B, 30), 7600).cuda()
        lf, of = full.train_on_batch([images, messages])
        ops.jpeg_bwd = counting
        try:
            ls, os_ = short.train_on_batch([images, messages])
        finally:
            ops.jpeg_bwd = orig
        assert all(lf[k] == ls[k] for k in lf) and all(torch.equal(a, b) for a, b in zip(of, os_))
        assert (calls["n"] == 0) == zero, (type(layer).__name__, calls)
        sf, ss = _state(full), _state(short)
        for k in sf:
            if zero and (k.startswith("Dec.layers.0.layers.0.weight") or k in ("Dec.grad", "optED.m1", "optED.v1")):
                # (one Adam step moves a weight by lr * m / (sqrt(v) + eps): a gradient element near zero may take a visibly different step
                # from a round-off-sized difference, so the parameter and the moments are compared through the gradient's scale)
                d = (sf[k].float() - ss[k].float()).abs().max().item()
                assert d <= 1e-5 * max(sf[k].float().abs().max().item(), 1e-3 if "weight" in k else 0.0) + (2.1e-3 if "weight" in k else 0.0), (k, d)
            else:
                assert torch.equal(sf[k], ss[k]), k


def test_graph_inputs_same_tensor_skips_the_copy_but_sees_in_place_changes():
    """a loop over one resident batch hands train_on_batch the same tensor objects every step: the graph's static inputs are then not
    re-copied -- unless the tensor was modified in place since (torch's version counter)"""
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    eager, graph = _make(64, NL.JpegSS(50), torch.bfloat16), _make(64, NL.JpegSS(50), torch.bfloat16).enable_graph()
    images = detgen.uniform((4, 3, 64, 64), 7900).cuda(); messages = detgen.bits((4, 30), 7901).cuda()
    for i in range(7):
        if i == 5:
            images.mul_(0.5).add_(0.1)          # in place: same object, new contents
            messages.copy_(1.0 - messages)
        le, oe = eager.train_on_batch([images, messages])
        lg, og = graph.train_on_batch([images, messages])
        assert all(le[k] == lg[k] for k in le) and all(torch.equal(a, b) for a, b in zip(oe, og)), i
    for k, v in _state(eager).items():
        assert torch.equal(v, _state(graph)[k]), k


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_model_surface_attack_cycle_replayed_equals_eager(tmp_path, dtype):
    """configuration C3 through feed_data / optimize_parameters (BASELINE.json configs[2]; the reference's loop: train.py:99-109 ->
    IRNrhi_model.optimize_parameters): with train.graph (the default) every layer of the 7-attack cycle gets its own captured step --
    Resize and Crop run with the cycle's fixed arguments -- and 5 rounds of the cycle (2 eager calls per layer, the capture, 2 replays)
    leave every logged scalar, the attack's name, every parameter, buffer and Adam moment exactly as the eagerly enqueued model's.
    f16: the device-side GradScaler's state too."""
    from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
    from video_watermarking_forgery_detection_amd.options.options import dict_to_nonedict
    attacks = ["Jpeg50", "JpegSS70", "JpegMask90", "GaussianBlur", "MiddleBlur3", "Resize", "Crop"]

    def make(graph):
        opt = dict_to_nonedict({"gpu_ids": [0], "dist": False, "is_train": True, "datasets": {"train": {"GT_size": 64, "batch_size": 4}},
                                "train": {"compute_dtype": dtype, "attacks": attacks, "lr_G": 1e-3, "manual_seed": 10, "save_interval": 3000,
                                          "localizer": False, "graph": graph},
                                "path": {"models": str(tmp_path / "models"), "training_state": str(tmp_path / "state")}})
        m = IRNrhiModel(opt)
        for net in (m.netG.encoder, m.netG.decoder, m.discriminator):
            detgen.fill_module(net)
        return m

    eager, graph = make(False), make(True)
    assert eager.hidden._graphs is None and graph.hidden._graphs is not None
    nsteps = 2 + 7 * 5
    for step in range(1, nsteps + 1):
        x = detgen.uniform((4, 3, 64, 64), 100 + step)
        msg = (detgen.uniform((4, 30), 500 + step) > 0.5).float().cuda()
        out = []
        for m in (eager, graph):
            m.feed_data(x)
            m.messages = msg
            logs, _ = m.optimize_parameters(step, None)
            out.append((logs, m.attack.name))
        assert out[0][1] == out[1][1], step
        assert out[0][0] == out[1][0], (step, out[0][1], out[0][0], out[1][0])
    g = graph.hidden._graphs
    assert len(g) == 7 and all(v.graph is not None for v in g.values()), {k[-1]: v.calls for k, v in g.items()}
    a, b = _state(eager.hidden), _state(graph.hidden)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    if eager.amp is not None:
        assert torch.equal(eager.amp.state, graph.amp.state)


def test_failed_capture_falls_back_to_the_eager_step_with_a_warning():
    """a capture that raises leaves the model untouched (nothing executes while a stream captures): the step and the following ones are enqueued
    eagerly, a warning says so once, results equal the eager model's bit for bit"""
    import warnings
    from video_watermarking_forgery_detection_amd import noise_layers as NL

    class Flaky(NL.Identity):
        def fwd(self, x):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("no capture through this layer today")
            return super().fwd(x)

    eager, graph = _make(32, NL.Identity(), torch.bfloat16), _make(32, Flaky(), torch.bfloat16).enable_graph()
    x = detgen.uniform((2, 3, 32, 32), 3)
    msg = (detgen.uniform((2, 30), 4) > 0.5).float().cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for _ in range(5):
            le, _ = eager.train_on_batch([x, msg])
            lg, _ = graph.train_on_batch([x, msg])
            assert dict(le) == dict(lg)
    assert sum("capture of the training step failed" in str(i.message) for i in w) == 1
    (g,) = graph._graphs.values()
    assert g.failed and g.graph is None
    a, b = _state(eager), _state(graph)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("two", [False, True])
def test_replayed_steps_without_a_host_read_in_between(two):
    """A loop that does not look at the losses every step (bench.py) lets the host run ahead of the GPU: step k + 1's inputs and Adam constants
    are staged while step k's replay is still queued.  Twelve such steps must leave the model exactly where the eagerly enqueued, per-step
    synchronised run leaves it (round 4: a reused pinned staging buffer handed step k the constants of step k + 1)."""
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    eager, graph = _make(128, NL.Identity(), torch.bfloat16), _make(128, NL.Identity(), torch.bfloat16).enable_graph()
    graph.two_streams = two
    batches = [(detgen.uniform((16, 3, 128, 128), 900 + i).cuda(), (detgen.uniform((16, 30), 950 + i) > 0.5).float().cuda()) for i in range(12)]
    for x, m in batches:
        le, _ = eager.train_on_batch([x, m])
        dict(le)                                   # the eager run reads every step
        torch.cuda.synchronize()
    kept = [graph.train_on_batch([x, m])[0] for x, m in batches]     # nothing is read until all twelve are enqueued
    torch.cuda.synchronize()
    assert dict(kept[-1]) == dict(le)
    a, b = _state(eager), _state(graph)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_jpeg_kernels_beside_the_sixteen_channel_backward_kernels():
    """The two-chain schedule runs the attack (chain B) beside the discriminator's passes (chain A), whose image-fed first layer uses
    bwd_ws16 / the 16-channel weight gradient: the only kernels with transposing LDS reads that were small enough to share a CU with the
    JPEG kernels -- whose results came out wrong (a quarter-wave of one register: two pixels of an 8x8 block) in 22-30 of 30 launches
    while such a workgroup was resident beside them (round 4; cause not understood, nothing written out of bounds).  The two kernels now
    request LDS up to 137,472 B (csrc/wgrad_ws.hip WM_LDS_PAD16), so that a JPEG workgroup no longer fits on their CU.  Fixed operands, the
    solo result is the reference; forward and backward, all three quantisation modes + DiffJPEG."""
    from video_watermarking_forgery_detection_amd import ops
    from video_watermarking_forgery_detection_amd import noise_layers as NL
    B, H, W, C, dt = 16, 256, 256, 64, torch.bfloat16
    x = detgen.uniform((B, 3, H, W), 1).cuda()
    gy = detgen.normal((B, 3, H, W), 2).cuda()
    g = detgen.normal((B, H, W, C), 3).cuda().to(dt); y = detgen.normal((B, H, W, C), 4).cuda().to(dt)
    stats = detgen.uniform((4, C), 5).cuda() + 0.5
    coef = detgen.uniform((3, C), 6).cuda() * 0.01; coef[0] += 1.0
    x16 = detgen.normal((B, H, W, 16), 7).cuda().to(dt); x16[..., 3:] = 0
    w16 = detgen.normal((C, 3, 3, 3), 8, std=0.05).cuda(); dw16 = torch.zeros(C, 3, 3, 3, device="cuda")
    wpt16 = ops.pack_w3x3(w16, C, 16, dt, transpose=True)
    dwf = torch.zeros(C, 16, 3, 3, device="cuda")

    def neighbours():
        ops.conv3x3_bwd_fused16(g, y, stats, coef, wpt16, x16, dw16, False)
        ops.conv3x3_wgrad(x16, 16, None, None, g, dwf, False)
        ops.conv3x3_bwd_fused16(g, y, stats, coef, wpt16, x16, dw16, False)

    probes = {}
    for name, L in (("Jpeg", NL.Jpeg(50)), ("JpegSS", NL.JpegSS(50)), ("JpegMask", NL.JpegMask(50))):
        probes[name + " fwd"] = (lambda L=L: ops.jpeg_fwd(x, L._mode, L._tables, 0))
        probes[name + " bwd"] = (lambda L=L: ops.jpeg_bwd(x, gy, L._mode, L._tables, 0))
    probes["DiffJPEG fwd"] = lambda: ops.diffjpeg_fwd(x, 1, 1.0)
    probes["DiffJPEG bwd"] = lambda: ops.diffjpeg_bwd(x, gy, 1, 1.0)
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    for name, fn in probes.items():
        solo = fn()
        torch.cuda.synchronize()
        for it in range(8):
            sA.wait_stream(torch.cuda.current_stream()); sB.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sA):
                neighbours()
            with torch.cuda.stream(sB):
                out = fn()
            torch.cuda.synchronize()
            assert torch.equal(out, solo), (name, it, int((out != solo).sum()))


def test_two_chain_training_under_a_differentiable_jpeg_is_the_one_stream_run():
    """the schedule-level form of the test above (tools/train_sanity_modes.py found it): JpegSS at the benchmark's size, six steps, the
    two-chain order three times over -- every run must end exactly where the one-stream run ends (round 4: one in two did not)."""
    from video_watermarking_forgery_detection_amd import noise_layers as NL

    def run(two):
        h = _make(256, NL.JpegSS(50), torch.bfloat16)
        h.two_streams = two
        for i in range(6):
            h.train_on_batch([detgen.uniform((16, 3, 256, 256), 40 + i).cuda(), (detgen.uniform((16, 30), 60 + i) > 0.5).float().cuda()])
            torch.cuda.synchronize()
        return _state(h)

    ref = run(False)
    for r in range(3):
        got = run(True)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (r, k)
