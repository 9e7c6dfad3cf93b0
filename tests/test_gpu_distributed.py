"""GPU, two data-parallel replicas of the training step in ONE process (two host threads, one card): the bucket protocol of
Hidden.train_on_batch / IRNrhiModel._localise -- discriminator bucket under the decoder's forward, decoder bucket under the attack +
encoder backward, the encoder in two reverse-order buckets, the UNet in four, 1/world folded into the optimiser kernel -- driven
through a stand-in for distributed.GradSync that sums the two replicas' buckets at a thread barrier.  (No child processes: a
pytest process that has initialised the GPU must not fork+exec on this pool.  The torch.distributed calls themselves are covered
by tests/test_cpu_distributed.py -- gloo, two ranks -- and by test_gpu_configs.test_grad_sync_one_rank_is_identity -- RCCL.)

Checked: both replicas end bit-identical, every bucket of every network was exchanged exactly once per step with matching
sizes on both sides, and the discriminator after its first update equals one hand-written Adam step on the MEAN of the two
shards' gradients."""
import threading

import pytest
import torch

import detgen

pytestmark = pytest.mark.gpu


class PairSync:
    """distributed.GradSync's interface for replica `k` of 2 living in one process"""
    world = 2
    scale = 0.5
    active = True

    def __init__(self, shared, k):
        self.sh, self.k, self.n = shared, k, 0
        self.log = []

    def start(self, flat):
        idx = self.n
        self.n += 1
        self.log.append(flat.numel())
        return idx, flat

    def finish(self, handle):
        if handle is None:
            return
        idx, flat = handle
        sh = self.sh
        torch.cuda.synchronize()
        sh["buf"][self.k] = (idx, flat)
        sh["barrier"].wait()
        if self.k == 0:
            (i0, a), (i1, b) = sh["buf"][0], sh["buf"][1]
            assert i0 == i1 and a.numel() == b.numel(), "the replicas exchanged different buckets"
            s = a + b
            a.copy_(s)
            b.copy_(s)
            torch.cuda.synchronize()
        sh["barrier"].wait()

    def finish_all(self, handles):
        for h in handles:
            self.finish(h)

    def average_(self, flat):
        flat.mul_(self.scale)
        return flat

    def __call__(self, flat):
        self.finish(self.start(flat))
        return self.average_(flat)


def _run_pair(make, step):
    shared = {"barrier": threading.Barrier(2), "buf": [None, None]}
    reps, errs = [None, None], []

    def body(k):
        try:
            torch.cuda.set_device(0)
            reps[k] = make(k, PairSync(shared, k))
            shared["barrier"].wait()
            step(k, reps[k])
            torch.cuda.synchronize()
        except BaseException as e:   # noqa: BLE001  (a failing replica must not leave the other one at the barrier)
            errs.append(e)
            shared["barrier"].abort()

    ts = [threading.Thread(target=body, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    if errs:
        raise errs[0]
    return reps


def _hidden(sync, dtype=torch.float32):
    from video_watermarking_forgery_detection_amd.hidden_models import Hidden
    from video_watermarking_forgery_detection_amd.noise_layers import JpegSS
    from video_watermarking_forgery_detection_amd.options import HiDDenConfiguration
    h = Hidden(HiDDenConfiguration(H=32, W=32), torch.device("cuda", 0), JpegSS(50), None, compute_dtype=dtype, grad_sync=sync)
    for m in (h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator):
        detgen.fill_module(m)
    return h


def _nets(h):
    return [h.encoder_decoder.encoder, h.encoder_decoder.decoder, h.discriminator]


def test_two_replicas_train_on_batch():
    shards = [(detgen.uniform((2, 3, 32, 32), 500 + r), detgen.bits((2, 30), 600 + r)) for r in range(2)]
    p0 = torch.cat([n.flat_params for n in _nets(_hidden(None))]).clone()
    reps = _run_pair(lambda k, sync: _hidden(sync), lambda k, h: h.train_on_batch(list(shards[k])))
    a, b = reps
    fa, fb = (torch.cat([n.flat_params for n in _nets(h)]) for h in (a, b))
    assert torch.equal(fa, fb) and not torch.equal(fa, p0)
    # buckets per step: discriminator, decoder, encoder [after_concat + final], encoder [conv_layers]
    enc, dec, dis = _nets(a)
    cut = enc.body_param_count()
    assert a.grad_sync.log == b.grad_sync.log == [dis.flat_grads.numel(), dec.flat_grads.numel(), enc.flat_grads.numel() - cut, cut]
    # the discriminator after its update == one Adam step (zero moments, betas 0.9 / 0.999) on the mean of the shards' gradients
    gD = []
    for r in range(2):
        h = _hidden(None)
        seen = {}
        h.train_on_batch(list(shards[r]), clip=lambda flats: seen.setdefault(len(seen), [f.clone() for f in flats]))
        gD.append(seen[0][0])
    n_ed = enc.flat_params.numel() + dec.flat_params.numel()
    g = 0.5 * (gD[0] + gD[1])
    m, v = 0.1 * g, 0.001 * g * g
    upd = p0[n_ed:] - (1e-3 / (1 - 0.9)) * m / (v.sqrt() / (1 - 0.999) ** 0.5 + 1e-8)
    torch.testing.assert_close(fa[n_ed:], upd, rtol=1e-5, atol=2e-7)


def test_two_replicas_with_localiser_and_clipping(tmp_path):
    """the model surface with the UNet head: its 31 MB of gradients leave in four reverse-order buckets from inside the backward;
    with gradient clipping the averaged gradients are what gets clipped"""
    from video_watermarking_forgery_detection_amd.models.IRNrhi_model import IRNrhiModel
    from video_watermarking_forgery_detection_amd.options.options import dict_to_nonedict

    def make(k, sync):
        opt = dict_to_nonedict({"gpu_ids": [0], "dist": False, "is_train": True, "datasets": {"train": {"GT_size": 32, "batch_size": 4}},
                                "train": {"compute_dtype": "f32", "attacks": ["JpegSS70"], "lr_G": 1e-3, "localizer": True,
                                          "gradient_clipping": 1.0, "save_interval": 3000},
                                "path": {"models": str(tmp_path / f"m{k}")}})
        m = IRNrhiModel(opt)
        for net in (m.netG.encoder, m.netG.decoder, m.discriminator, m.localizer):
            detgen.fill_module(net)
        m.grad_sync = sync
        m.hidden.grad_sync = sync
        return m

    def step(k, m):
        imgs = detgen.uniform((2, 3, 32, 32), 700 + k)
        mask = torch.zeros(2, 1, 32, 32)
        mask[:, :, 4:20, 8:24] = 1.0
        prev = detgen.uniform((2, 3, 32, 32), 800 + k).cuda()
        m.previous_images = prev
        m.previous_previous_images = prev
        m.feed_data({"GT": imgs, "mask": mask, "messages": detgen.bits((2, 30), 900 + k)})
        logs, _ = m.optimize_parameters(1, None)
        assert all(v == v for _, v in logs if isinstance(v, float))

    a, b = _run_pair(make, step)
    for na, nb in zip((a.netG.encoder, a.netG.decoder, a.discriminator, a.localizer), (b.netG.encoder, b.netG.decoder, b.discriminator, b.localizer)):
        assert torch.equal(na.flat_params, nb.flat_params)
    sizes = [hi - lo for lo, hi in a.localizer.grad_buckets()]
    assert sum(sizes) == a.localizer.flat_grads.numel()
    log = a.grad_sync.log
    assert log == b.grad_sync.log and all(s in log for s in sizes) and len(log) == 4 + 4
