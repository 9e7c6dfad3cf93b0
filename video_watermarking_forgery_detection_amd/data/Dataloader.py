"""DVDataset -- the DAVIS clip reader either side of the hot path; mirror of the reference's data/Dataloader.py:22-99.

Layout read: `<root>/JPEGImages/480p/<clip>/NNNNN.jpg` frames and `<root>/Annotations/480p/<clip>/NNNNN.png` object masks.  An item is
(`Video_GT [3,T,S,S]` float in [0,1], `Mask_GT [1,T,S,S]` in {0,1}) -- the shapes `train.py` hands to `feed_data` -- with every frame
resized DIRECTLY to S x S (no crop, :33-34) by bilinear interpolation with half-pixel centres and no antialiasing (cv2.INTER_LINEAR).

The reference file cannot run as written (SURVEY box: `read_mask` is never called and `read_img`'s single return value is unpacked into
two at :88, the frames are binarised by a line copied from the mask path at :38, a bare `import util`): what is implemented is its evident
intent -- frames as floats, masks binarised at > 0, a clip accepted when its mean mask rate is below 0.2 and put on a skip list
otherwise (:77-95), `IOError("Load <clip> Error")` on a failed read (:89-90).  cv2 is absent from the image, so decoding goes through PIL
and the resize through torch's CPU bilinear kernel (the same arithmetic as cv2's float path; cv2's uint8 fixed-point path may differ by
one grey level: parity unpinned for that, like every cv2 / kornia dependency of the reference)."""
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
    def __init__(self, root_path='/home/groupshare/DAVIS/', image_size=256, is_train=True, clip_length=None, max_mask_rate=0.2):
        super(DVDataset, self).__init__()
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
                Video_GT = read_img(clip, self.root_path, self.videopath, self.image_size, self.clip_length)
                Mask_GT, rate = read_mask(clip, self.root_path, self.maskpath, self.image_size, self.clip_length)
            except Exception:
                raise IOError("Load {} Error".format(clip))
            if rate < self.max_mask_rate:
                return Video_GT, Mask_GT
            self.skip_list.append(index)

    def __len__(self):
        return len(self.list)
