"""Free-form stroke masks -- mirror of the reference's models/IRNrhi_model.py:1147-1193 (`generate_stroke_mask`, `np_free_form_mask`):
the same numpy RNG draws in the same order (lower bound of the covered fraction, then per stroke: vertex count, start point, and per
vertex: angle, length, brush width), strokes accumulated until the covered fraction reaches the bound.

The reference rasterises with cv2.line / cv2.circle, which this image does not have; the strokes are rasterised here as the set of pixels
within brushWidth/2 of the segment (a thick line with round ends) plus the one-pixel circle outlines of radius brushWidth//2 the
reference draws at the joints.  Pixel-exact agreement with cv2's Bresenham-style thick lines is not claimed (parity unpinned: cv2 absent);
what is kept is the sampling procedure, value range {0,1} and the coverage statistics."""
import numpy as np
import torch


def _line(mask, p0, p1, width):
    h, w = mask.shape
    (x0, y0), (x1, y1) = p0, p1
    r = max(width / 2.0, 0.5)
    lo_y, hi_y = int(max(0, min(y0, y1) - r - 1)), int(min(h - 1, max(y0, y1) + r + 1))
    lo_x, hi_x = int(max(0, min(x0, x1) - r - 1)), int(min(w - 1, max(x0, x1) + r + 1))
    if hi_y < lo_y or hi_x < lo_x:
        return
    yy, xx = np.mgrid[lo_y:hi_y + 1, lo_x:hi_x + 1].astype(np.float32)
    dx, dy = float(x1 - x0), float(y1 - y0)
    den = dx * dx + dy * dy
    t = np.clip(((xx - x0) * dx + (yy - y0) * dy) / den, 0.0, 1.0) if den > 0 else np.zeros_like(xx)
    d2 = (xx - (x0 + t * dx)) ** 2 + (yy - (y0 + t * dy)) ** 2
    mask[lo_y:hi_y + 1, lo_x:hi_x + 1][d2 <= r * r] = 1.0


def _circle_outline(mask, c, radius, value):
    h, w = mask.shape
    x0, y0 = c
    lo_y, hi_y = int(max(0, y0 - radius - 1)), int(min(h - 1, y0 + radius + 1))
    lo_x, hi_x = int(max(0, x0 - radius - 1)), int(min(w - 1, x0 + radius + 1))
    if hi_y < lo_y or hi_x < lo_x:
        return
    yy, xx = np.mgrid[lo_y:hi_y + 1, lo_x:hi_x + 1].astype(np.float32)
    d = np.sqrt((xx - x0) ** 2 + (yy - y0) ** 2)
    mask[lo_y:hi_y + 1, lo_x:hi_x + 1][np.abs(d - radius) <= 0.5] = value


def np_free_form_mask(mask_re, maxVertex, maxLength, maxBrushWidth, maxAngle, h, w):
    """IRNrhi_model.py:1173-1193 (cv2 points are (x, y); the reference passes (startY, startX) as such, kept)"""
    mask = np.zeros_like(mask_re)
    numVertex = np.random.randint(1, maxVertex + 1)
    startY = np.random.randint(h)
    startX = np.random.randint(w)
    brushWidth = 0
    for i in range(numVertex):
        angle = np.random.randint(maxAngle + 1)
        angle = angle / 360.0 * 2 * np.pi
        if i % 2 == 0:
            angle = 2 * np.pi - angle
        length = np.random.randint(8, maxLength + 1)
        brushWidth = np.random.randint(8, maxBrushWidth + 1) // 2 * 2
        nextY = startY + length * np.cos(angle)
        nextX = startX + length * np.sin(angle)
        nextY = int(np.maximum(np.minimum(nextY, h - 1), 0))
        nextX = int(np.maximum(np.minimum(nextX, w - 1), 0))
        _line(mask, (startY, startX), (nextY, nextX), brushWidth)
        _circle_outline(mask, (startY, startX), brushWidth // 2, 2)
        startY, startX = nextY, nextX
    _circle_outline(mask, (startY, startX), brushWidth // 2, 2)
    return mask


def generate_stroke_mask(im_size, parts=5, parts_square=2, maxVertex=4, maxLength=64, maxBrushWidth=32, maxAngle=360, percent_range=(0.0, 0.5)):
    """IRNrhi_model.py:1147-1171 -> (mask tensor [H,W] in {0,1}, covered fraction)"""
    maxLength = int(im_size[0] / 5)
    maxBrushWidth = int(im_size[0] / 5)
    mask = np.zeros((im_size[0], im_size[1]), dtype=np.float32)
    lower_bound_percent = percent_range[0] + (percent_range[1] - percent_range[0]) * np.random.rand()
    while True:
        mask = mask + np_free_form_mask(mask, maxVertex, maxLength, maxBrushWidth, maxAngle, im_size[0], im_size[1])
        mask = np.minimum(mask, 1.0)
        percent = np.mean(mask)
        if percent >= lower_bound_percent:
            break
    mask = np.maximum(mask, 0.0)
    return torch.from_numpy(mask).contiguous(), float(np.mean(mask))
