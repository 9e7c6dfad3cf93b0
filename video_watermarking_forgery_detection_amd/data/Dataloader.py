"""DVDataset -- the DAVIS clip reader either side of the hot path; mirror of the reference's data/Dataloader.py:22-99.

Layout read: `<root>/JPEGImages/480p/<clip>/NNNNN.jpg` frames and `<root>/Annotations/480p/<clip>/NNNNN.png` object masks.  An item is
(`Video_GT [3,T,S,S]` float in [0,1], `Mask_GT [1,T,S,S]` in {0,1}) -- the shapes `train.py` hands to `feed_data` -- with every frame
resized DIRECTLY to S x S (no crop, :33-34) by bilinear interpolation with half-pixel centres and no antialiasing (cv2.INTER_LINEAR).

The reference file cannot run as written (SURVEY box: `read_mask` is never called and `read_img`'s single return value is unpacked into
two at :88, the frames are binarised by a line copied from the mask path at :38, a bare `import util`): what is implemented is its evident
intent -- frames as floats, masks binarised at > 0, a clip accepted when its mean mask rate is below 0.2 and put on a skip list
otherwise (:77-95), `IOError("Load <clip> Error")` on a failed read (:89-90).  cv2 is absent from the image, so decoding goes through PIL.

Two resize paths, same arithmetic (bilinear, half-pixel centres, no antialiasing):
  * resize_on="host" (default; works in DataLoader worker processes): torch's CPU bilinear kernel per frame;
  * resize_on="device": the decoded uint8 frames of a clip go to the GPU as they are (854x480x3 bytes per frame instead of 3 float planes)
    and are converted and resized there by the HIP kernels (ops.u8_hwc_to_planes -> ops.resample_fwd, the same ATen-exact bilinear kernel
    the Crop attack runs on); items come back as CUDA tensors, so this path is for num_workers = 0 / the training process itself.
What stays unpinned against the reference: cv2.resize on uint8 input uses a fixed-point (11-bit coefficient) path whose result can
differ from the float arithmetic by one grey level (1/255) before the reference's `.float()`; cv2 is not in the image, the reference holds
no fixture for its loader, and the reference's read_img cannot run as written."""
import os

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data as data
from PIL import Image


def _frame_key(name):
    stem = os.path.splitext(name)[0]
    digits = "".join(ch for ch in stem if ch.isdigit())
    return int(digits) if digits else 0


def _resize(t, size):
    """[C,H,W] float -> [C,size,size]; cv2.INTER_LINEAR = bilinear, half-pixel centres, no antialias"""
    return F.interpolate(t.unsqueeze(0), size=(size, size), mode="bilinear", align_corners=False).squeeze(0)


def _decode_clip(clip, root_path, sub, mode, clip_length):
    """the clip's frames decoded on the host: uint8 [T,H,W,C] (C = 3 for "RGB", 1 for "L") when every frame has the same size (DAVIS
    480p does), else a list of [H_i,W_i,C] arrays -- the host path resizes frame by frame and never needed equal sizes"""
    d = os.path.join(root_path, sub, clip)
    names = sorted(os.listdir(d), key=_frame_key)
    if clip_length:
        names = names[:clip_length]
    frames = [np.asarray(Image.open(os.path.join(d, n)).convert(mode), dtype=np.uint8) for n in names]
    frames = [f if f.ndim == 3 else f[..., None] for f in frames]
    if len({f.shape for f in frames}) > 1:
        return frames
    return np.stack(frames)


def resize_clip_device(frames_u8, size, device="cuda"):
    """uint8 [T,H,W,C] (numpy or tensor) -> float32 [C,T,size,size] in [0,1] on the GPU: one conversion launch and one bilinear launch for
    the whole clip (Dataloader.py:27-35 per frame on the host).  A list of frames of unequal sizes is resized frame by frame."""
    from .. import ops
    if isinstance(frames_u8, (list, tuple)):
        return torch.cat([resize_clip_device(f[None], size, device) for f in frames_u8], dim=1)
    t = torch.as_tensor(frames_u8).to(device, non_blocking=True)
    planes = ops.u8_hwc_to_planes(t, 1.0 / 255.0)                                   # [T,C,H,W]
    T, C, H, W = planes.shape
    out = ops.resample_fwd(planes, (0, H, 0, W), (size, size), 0).view(T, C, size, size)    # kind 0 = bilinear
    return out.permute(1, 0, 2, 3).contiguous()


def read_img_device(clip, root_path, videopath, GT_size, clip_length=None, device="cuda"):
    """read_img with the resize on the GPU: [3,T,S,S] float32 cuda"""
    return resize_clip_device(_decode_clip(clip, root_path, videopath, "RGB", clip_length), GT_size, device)


def read_mask_device(clip, root_path, maskpath, GT_size, clip_length=None, device="cuda"):
    """read_mask with the resize on the GPU: ([1,T,S,S] in {0,1} cuda, mean mask rate as a float -- one host sync per clip, the skip-list
    decision of DVDataset.__getitem__ needs the number)"""
    m = resize_clip_device(_decode_clip(clip, root_path, maskpath, "L", clip_length), GT_size, device)
    m = (m > 0).float()
    return m, float(m.mean())


def read_img(clip, root_path, videopath, GT_size, clip_length=None):
    """[3,T,S,S] float32 RGB in [0,1] (Dataloader.py:22-41 without the stray binarisation of :38)"""
    d = os.path.join(root_path, videopath, clip)
    names = sorted(os.listdir(d), key=_frame_key)
    if clip_length:
        names = names[:clip_length]
    out = torch.zeros(3, len(names), GT_size, GT_size)
    for i, n in enumerate(names):
        im = np.asarray(Image.open(os.path.join(d, n)).convert("RGB"), dtype=np.float32) / 255.0
        out[:, i] = _resize(torch.from_numpy(im).permute(2, 0, 1), GT_size)
    return out


def read_mask(clip, root_path, maskpath, GT_size, clip_length=None):
    """([1,T,S,S] in {0,1}, mean mask rate) -- Dataloader.py:43-57 + the > 0 binarisation"""
    d = os.path.join(root_path, maskpath, clip)
    names = sorted(os.listdir(d), key=_frame_key)
    if clip_length:
        names = names[:clip_length]
    out = torch.zeros(1, len(names), GT_size, GT_size)
    rate = []
    for i, n in enumerate(names):
        im = np.asarray(Image.open(os.path.join(d, n)).convert("L"), dtype=np.float32) / 255.0
        m = _resize(torch.from_numpy(im).unsqueeze(0), GT_size)
        m = torch.where(m > 0, torch.ones_like(m), torch.zeros_like(m))
        out[:, i] = m
        rate.append(float(m.mean()))
    return out, sum(rate) / max(1, len(rate))


class DVDataset(data.Dataset):
    def __init__(self, root_path='/home/groupshare/DAVIS/', image_size=256, is_train=True, clip_length=None, max_mask_rate=0.2, resize_on="host"):
        super(DVDataset, self).__init__()
        if resize_on not in ("host", "device"):
            raise ValueError("resize_on must be 'host' or 'device'")
        self.resize_on = resize_on          # "device": items are CUDA tensors (HIP resize); use with num_workers = 0
        self.is_train = is_train
        self.root_path = root_path
        self.image_size = image_size
        self.clip_length = clip_length      # frames per item (the clips of one batch must agree); None = the whole clip
        self.max_mask_rate = max_mask_rate
        self.videopath = 'JPEGImages/480p'
        self.maskpath = 'Annotations/480p'
        self.list = sorted(os.listdir(os.path.join(root_path, 'JPEGImages', '480p')))
        self.skip_list = []

    def __getitem__(self, index):
        # a clip is valid if the rate of its mask is below max_mask_rate; otherwise resample (Dataloader.py:77-95: the index handed
        # in is ignored, a random clip is drawn with the numpy RNG)
        while True:
            if len(self.skip_list) >= len(self.list):
                raise IOError("no clip with a mask rate below {}".format(self.max_mask_rate))
            index = np.random.randint(0, len(self.list))
            if index in self.skip_list:
                continue
            clip = self.list[index]
            try:
                rd_img, rd_mask = (read_img_device, read_mask_device) if self.resize_on == "device" else (read_img, read_mask)
                Video_GT = rd_img(clip, self.root_path, self.videopath, self.image_size, self.clip_length)
                Mask_GT, rate = rd_mask(clip, self.root_path, self.maskpath, self.image_size, self.clip_length)
            except (OSError, ValueError) as e:
                # the reference's message (Dataloader.py:86-88) for what it is about: a frame that cannot be read or decoded.  Anything else --
                # a HIP / runtime error of the device resize, out of memory -- is not a file error and propagates as what it is
                raise IOError("Load {} Error".format(clip)) from e
            if rate < self.max_mask_rate:
                return Video_GT, Mask_GT
            self.skip_list.append(index)

    def __len__(self):
        return len(self.list)
