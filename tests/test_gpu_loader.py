"""row f3 on the device: the clip loader's resize through the HIP kernels (wm_u8_hwc_to_planes + wm_resample_fwd bilinear) against the host
path (torch CPU bilinear, align_corners=False -- the float arithmetic of cv2.INTER_LINEAR; /root/reference/data/Dataloader.py:33-34,50-52)
on a synthetic DAVIS-layout clip.  Not pinned: cv2's uint8 fixed-point resize (see data/Dataloader.py's docstring)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_cpu_host_side import _fake_davis

pytestmark = pytest.mark.gpu


def test_u8_frames_to_planes_and_bilinear_resize_match_the_host_path():
    from video_watermarking_forgery_detection_amd import ops
    from video_watermarking_forgery_detection_amd.data.Dataloader import resize_clip_device
    rs = np.random.RandomState(5)
    frames = rs.randint(0, 256, size=(5, 48, 70, 3), dtype=np.uint8)          # T, H, W, C: not a multiple of anything
    planes = ops.u8_hwc_to_planes(torch.from_numpy(frames).cuda())
    ref = torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255.0
    assert planes.shape == (5, 3, 48, 70) and (planes.cpu() - ref).abs().max().item() <= 1.2e-7     # x * (1/255) vs x / 255: one f32 ulp
    for size in (32, 64, 37):
        got = resize_clip_device(frames, size).cpu()                            # [3, T, S, S]
        want = F.interpolate(ref, size=(size, size), mode="bilinear", align_corners=False).permute(1, 0, 2, 3)
        assert got.shape == (3, 5, size, size)
        assert (got - want).abs().max().item() < 1e-6
    gray = rs.randint(0, 256, size=(2, 40, 40, 1), dtype=np.uint8)
    assert resize_clip_device(gray, 16).shape == (1, 2, 16, 16)


def test_dataset_items_device_resize_equal_host_resize(tmp_path):
    from video_watermarking_forgery_detection_amd.data import DVDataset
    root = str(tmp_path / "DAVIS")
    _fake_davis(root, ["bear", "cat", "dog"], big=("dog",))
    host = DVDataset(root_path=root, image_size=32, clip_length=4)
    dev = DVDataset(root_path=root, image_size=32, clip_length=4, resize_on="device")
    for seed in range(6):
        np.random.seed(seed); vh, mh = host[0]
        np.random.seed(seed); vd, md = dev[0]
        assert vd.is_cuda and md.is_cuda and vd.shape == vh.shape and md.shape == mh.shape and vd.dtype == torch.float32
        assert (vd.cpu() - vh).abs().max().item() < 1e-6
        assert torch.equal(md.cpu(), mh)                                         # the {0,1} masks: identical
    assert sorted(host.skip_list) == sorted(dev.skip_list) and len(dev.skip_list) == 1   # the same clip exceeded the mask rate on both paths
    with pytest.raises(ValueError):
        DVDataset(root_path=root, resize_on="gpu")
