"""CPU: the rows either side of the hot path: the DAVIS clip loader + sampler + stroke masks (data/), the Progbar that
consumes `logs`, image sheets, and the TensorBoard scalar files -- the last against a file the reference's own TensorBoard run wrote."""
import io
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_davis(root, clips, T=5, size=(48, 64), big=()):
    rng = np.random.RandomState(0)
    for c in clips:
        os.makedirs(os.path.join(root, "JPEGImages", "480p", c))
        os.makedirs(os.path.join(root, "Annotations", "480p", c))
        for t in range(T):
            Image.fromarray(rng.randint(0, 256, size=size + (3,)).astype(np.uint8)).save(os.path.join(root, "JPEGImages", "480p", c, "%05d.png" % t))
            m = np.zeros(size, dtype=np.uint8)
            if c in big:
                m[:, : size[1] // 2] = 255           # half the frame: rate 0.5 > 0.2 -> rejected
            else:
                m[8:16, 8:20] = 38                    # DAVIS palette index > 0 -> 1 after the > 0 binarisation
            Image.fromarray(m).save(os.path.join(root, "Annotations", "480p", c, "%05d.png" % t))


def test_davis_clip_loader(tmp_path):
    from video_watermarking_forgery_detection_amd.data import DVDataset, DistIterSampler, create_dataloader
    from video_watermarking_forgery_detection_amd.options.options import dict_to_nonedict
    root = str(tmp_path / "DAVIS")
    _fake_davis(root, ["bear", "cat", "dog"], big=("dog",))
    ds = DVDataset(root_path=root, image_size=32, clip_length=4)
    assert len(ds) == 3
    np.random.seed(0)
    seen = set()
    for _ in range(12):
        v, m = ds[0]
        assert v.shape == (3, 4, 32, 32) and m.shape == (1, 4, 32, 32) and v.dtype == torch.float32
        assert 0.0 <= float(v.min()) and float(v.max()) <= 1.0 and 0.05 < float(v.std())          # frames are NOT binarised
        assert set(np.unique(m.numpy())) <= {0.0, 1.0} and 0 < float(m.mean()) < 0.2
        seen.add(float(v.sum()))
    assert len(seen) == 2 and ds.list[ds.skip_list[0]] == "dog"      # the high-rate clip went on the skip list, the other two are served
    # resize = bilinear with half-pixel centres (cv2.INTER_LINEAR): the known-answer of a 2x2 -> 4x4 ramp
    from video_watermarking_forgery_detection_amd.data.Dataloader import _resize
    r = _resize(torch.tensor([[[0.0, 1.0], [2.0, 3.0]]]), 4)[0]
    np.testing.assert_allclose(r[0].numpy(), [0.0, 0.25, 0.75, 1.0], atol=1e-6)
    np.testing.assert_allclose(r[:, 0].numpy(), [0.0, 0.5, 1.5, 2.0], atol=1e-6)
    # loader: batch of clips with the shapes feed_data takes
    opt = dict_to_nonedict({"phase": "train", "dist": False, "gpu_ids": [0]})
    dl = create_dataloader(ds, dict_to_nonedict({"batch_size": 2, "n_workers": 0}), opt, None)
    v, m = next(iter(dl))
    assert v.shape == (2, 3, 4, 32, 32) and m.shape == (2, 1, 4, 32, 32)
    # sampler: ranks partition the enlarged index list
    s0, s1 = DistIterSampler(ds, num_replicas=2, rank=0, ratio=4), DistIterSampler(ds, num_replicas=2, rank=1, ratio=4)
    i0, i1 = list(s0), list(s1)
    assert len(i0) == len(i1) == len(s0) == 6 and all(0 <= i < 3 for i in i0 + i1)
    # a broken clip raises the reference's IOError
    os.remove(os.path.join(root, "Annotations", "480p", "bear", "00000.png"))
    open(os.path.join(root, "Annotations", "480p", "bear", "00000.png"), "w").write("not an image")
    ds2 = DVDataset(root_path=root, image_size=32, clip_length=4)
    ds2.list = ["bear"]
    with pytest.raises(IOError, match="Load bear Error"):
        ds2[0]


def test_stroke_masks():
    from video_watermarking_forgery_detection_amd.data import generate_stroke_mask
    np.random.seed(1)
    rates = []
    for _ in range(20):
        m, rate = generate_stroke_mask([64, 64])
        assert m.shape == (64, 64) and m.dtype == torch.float32 and set(np.unique(m.numpy())) <= {0.0, 1.0}
        assert abs(rate - float(m.mean())) < 1e-6
        rates.append(rate)
    assert 0.0 < min(rates) and max(rates) < 0.95 and np.mean(rates) > 0.1      # coverage drawn from U(0, 0.5) as a lower bound
    np.random.seed(1)
    m2, _ = generate_stroke_mask([64, 64])
    np.random.seed(1)
    m3, _ = generate_stroke_mask([64, 64])
    assert torch.equal(m2, m3)                                                    # driven by the numpy RNG stream only


def test_progbar_consumes_logs():
    from video_watermarking_forgery_detection_amd.utils import Progbar
    buf, old = io.StringIO(), sys.stdout
    sys.stdout = buf
    try:
        bar = Progbar(40, stateful_metrics=["lr", "Kind"], interval=0.0)
        bar._dynamic_display = False
        bar.add(16, values=[("loss", 0.5), ("lr", 1e-3), ("Kind", "Jpeg50")])
        vals = bar.add(16, values=[("loss", 0.25), ("lr", 5e-4), ("Kind", "Resize")])
        bar.update(40, values=[("loss", 0.125)])
    finally:
        sys.stdout = old
    text = buf.getvalue()
    assert "16/40 [=========>...............]" in text and "ETA:" in text
    assert "32/40 [===================>.....]" in text
    assert "40/40 [=========================]" in text and "/step" in text
    assert "loss: 0.375000" in text and "lr: 0.0005" in text and "Kind: Resize" in text      # running mean vs stateful
    assert "loss: 0.325000" in text                                                          # (16*.5 + 16*.25 + 8*.125) / 40
    assert set(vals) == {"loss", "lr", "Kind"}


def test_stitch_images_and_postprocess(tmp_path):
    from video_watermarking_forgery_detection_amd.utils import imsave, postprocess, stitch_images
    x = torch.rand(4, 3, 8, 8)
    p = postprocess(x)
    assert p.shape == (4, 8, 8, 3) and p.dtype == torch.int32 and int(p.max()) <= 255
    assert torch.equal(p, (x * 255.0).permute(0, 2, 3, 1).int())
    mask = postprocess(torch.rand(4, 1, 8, 8))
    sheet = stitch_images(p, postprocess(x.flip(0)), mask, img_per_row=2)
    assert sheet.size == (8 * 2 * 3 + 5, 8 * 2)
    a = np.asarray(sheet)
    np.testing.assert_array_equal(a[0:8, 0:8], p[0].numpy().astype(np.uint8))
    np.testing.assert_array_equal(a[8:16, 8 * 3 + 5:8 * 3 + 5 + 8], p[3].numpy().astype(np.uint8))
    np.testing.assert_array_equal(a[0:8, 16:24, 0], mask[0].numpy().astype(np.uint8)[:, :, 0])
    imsave(p[0], str(tmp_path / "d" / "x.png"))
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / "d" / "x.png"))), p[0].numpy().astype(np.uint8))


def test_tensorboard_scalar_files(tmp_path):
    from video_watermarking_forgery_detection_amd.utils.tb_writer import SummaryWriter, crc32c, masked_crc, read_events
    assert crc32c(b"123456789") == 0xE3069283                                  # the CRC-32C check value
    # a file the reference's TensorBoard run wrote (tests/golden/make_golden.py tfevents): framing, masked crc and protobuf layout
    ev = read_events(os.path.join(ROOT, "tests", "golden", "tfevents_head.bin"))
    assert len(ev) == 10 and ev[0]["file_version"] == "brain.Event:2" and ev[0]["scalars"] == []
    assert ev[1]["step"] == 4 and ev[1]["scalars"][0][0] == "PSNR Forward" and abs(ev[1]["scalars"][0][1] - 31.539167404174805) < 1e-6
    assert {t for e in ev for t, _ in e["scalars"]} <= {"PSNR Forward", "PSNR Backward", "PSNR Watermark"}
    # what we write reads back, and is byte-compatible in structure with the reference's records
    w = SummaryWriter(str(tmp_path / "runs" / "RHI3"))
    w.add_scalar("PSNR Forward", 31.5, global_step=4, walltime=1653714422.68)
    w.add_scalar("BCEWithLogitsLoss", torch.tensor(0.693), global_step=4)
    w.close()
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    mine = read_events(w.path)
    assert [e["file_version"] for e in mine] == ["brain.Event:2", None, None]
    assert mine[1]["step"] == 4 and mine[1]["scalars"] == [("PSNR Forward", 31.5)] and mine[1]["wall_time"] == 1653714422.68
    assert mine[2]["scalars"][0][0] == "BCEWithLogitsLoss" and abs(mine[2]["scalars"][0][1] - 0.693) < 1e-6
    raw_ref = open(os.path.join(ROOT, "tests", "golden", "tfevents_head.bin"), "rb").read()
    raw_mine = open(w.path, "rb").read()
    # the second record of both files is Event{wall_time, step=4, summary{value{tag "PSNR Forward", simple_value}}}: same length, same
    # bytes apart from the wall time (8 bytes), the value (4 bytes) and the payload crc
    import struct
    def rec(raw, k):
        i = 0
        for _ in range(k):
            i += 16 + struct.unpack("<Q", raw[i:i + 8])[0]
        n = struct.unpack("<Q", raw[i:i + 8])[0]
        return raw[i:i + 16 + n]
    a, b = rec(raw_ref, 1), rec(raw_mine, 1)
    assert len(a) == len(b) and a[:12] == b[:12]
    pa, pb = a[12:-4], b[12:-4]
    assert pa[0] == pb[0] == 0x09 and pa[9:-4] == pb[9:-4]                 # everything between the wall time and the float value
    with pytest.raises(ValueError):
        bad = bytearray(raw_mine); bad[30] ^= 1
        open(str(tmp_path / "bad"), "wb").write(bytes(bad))
        read_events(str(tmp_path / "bad"))


def test_deferred_logs_and_attack_cycle_host_logic():
    """host-side pieces of the round-4 model surface (no GPU): DeferredLogs reads once, on the first look, and behaves as the reference's list of
    (name, value) pairs (train.py:109 feeds it to Progbar.add); the attack cycle names the layer a step runs and keys its captured step by it"""
    from video_watermarking_forgery_detection_amd.models.IRNrhi_model import DeferredLogs, _AttackCycle
    calls = []

    def read():
        calls.append(1)
        return [("loss", 1.5), ("lr", 1e-3)]
    d = DeferredLogs(read)
    assert calls == []
    assert len(d) == 2 and calls == [1]
    assert list(d) == [("loss", 1.5), ("lr", 1e-3)] and d[0] == ("loss", 1.5) and d == [("loss", 1.5), ("lr", 1e-3)] and dict(d)["lr"] == 1e-3
    assert calls == [1] and "loss" in repr(d)
    assert not DeferredLogs(lambda: [])          # an empty step (the first two calls of a run) is falsy, as the reference's []

    class L:
        def __init__(self, name):
            self.name = name

        def fwd(self, x):
            return x, None

        def bwd(self, c, g):
            return g
    cyc = _AttackCycle([L("a"), L("b"), L("c")])
    keys = []
    for k in range(7):
        cyc.k = k
        keys.append(cyc.capture_key())
        assert cyc.name == "abc"[k % 3]
        y, ctx = cyc.fwd("img")
        assert y == "img" and ctx[0].name == cyc.name
    assert keys[0] == keys[3] == keys[6] and len(set(keys)) == 3 and all(k[0] == "cycle" for k in keys)
