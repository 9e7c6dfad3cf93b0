"""Image dumps of the training loop -- `stitch_images` / `imsave` / `postprocess` of the reference (utils/__init__.py:68-85,95-97;
models/IRNcrop_model.py:660-664; used at IRNcrop_model.py:421-437 every 500 steps): a sheet with one row per sample and one column per
tensor (input | watermarked | 10 x |difference| | attacked | predicted mask | ground-truth mask)."""
import os

import numpy as np
import torch
from PIL import Image


def postprocess(img):
    """[B,C,H,W] in [0,1] -> [B,H,W,C] int in [0,255] (IRNcrop_model.py:660-664)"""
    img = img * 255.0
    img = img.permute(0, 2, 3, 1)
    return img.int()


def _to_pil(t):
    a = np.asarray(t.detach().cpu()).astype(np.uint8).squeeze()
    return Image.fromarray(a)


def stitch_images(inputs, *outputs, img_per_row=2):
    """inputs / outputs: sequences of [H,W,C] (or [H,W,1]) tensors in [0,255]; returns the PIL sheet (utils/__init__.py:68-85)"""
    gap = 5
    columns = len(outputs) + 1
    width, height = inputs[0][:, :, 0].shape
    rows = int(len(inputs) / img_per_row)
    sheet = Image.new('RGB', (width * img_per_row * columns + gap * (img_per_row - 1), height * rows))
    groups = [inputs, *outputs]
    for ix in range(len(inputs)):
        col, row = ix % img_per_row, ix // img_per_row
        xoffset = col * width * columns + col * gap
        for cat, group in enumerate(groups):
            sheet.paste(_to_pil(group[ix]), (xoffset + cat * width, row * height))
    return sheet


def imsave(img, path):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    _to_pil(img).save(path)
