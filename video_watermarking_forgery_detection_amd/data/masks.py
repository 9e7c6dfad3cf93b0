"""Free-form stroke masks -- mirror of the reference's models/IRNrhi_model.py:1147-1193 (`generate_stroke_mask`, `np_free_form_mask`):
the same numpy RNG draws in the same order (lower bound of the covered fraction, then per stroke: vertex count, start point, and per
vertex: angle, length, brush width), strokes accumulated until the covered fraction reaches the bound.

The reference rasterises with cv2.line / cv2.circle, which this image does not have; the strokes are rasterised here as the set of pixels
within brushWidth/2 of the segment (a thick line with round ends) plus the one-pixel circle outlines of radius brushWidth//2 the
reference draws at the joints.  Pixel-exact agreement with cv2's Bresenham-style thick lines is not claimed (parity unpinned: cv2 absent);
what is kept is the sampling procedure, value range {0,1} and the coverage statistics.

Contract restated (not the reference's text): `generate_stroke_mask(im_size)` draws, from numpy's GLOBAL RNG and in this order,
  1. one uniform u -> the target coverage  lo + (hi - lo) * u  of `percent_range`;
  2. then polylines until the covered fraction of the image reaches the target, each polyline drawing
     a. its number of segments  ~ randint(1, maxVertex + 1),
     b. its first joint (two randint: first axis h, second axis w),
     c. per segment: a turn in whole degrees ~ randint(maxAngle + 1) (mirrored on even segments), a length ~ randint(8, S + 1) and an even
        brush width = randint(8, S + 1) rounded down to even, with S = im_size[0] // 5; the next joint is the clamped end point.
The joints are handed to the rasteriser as the reference hands them to cv2 (first coordinate as cv2's x)."""
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


def _polyline(h, w, max_segments, max_step, max_turn_deg):
    """one polyline of the contract above: yields (joint, next joint, even brush width) per segment, drawing from np.random in the
    contract's order; the joints are (first-axis, second-axis) integer pairs clamped to the image"""
    n = np.random.randint(1, max_segments + 1)
    joint = (np.random.randint(h), np.random.randint(w))
    for seg in range(n):
        turn = np.deg2rad(float(np.random.randint(max_turn_deg + 1)))
        if seg % 2 == 0:
            turn = 2.0 * np.pi - turn
        step = np.random.randint(8, max_step + 1)
        brush = (np.random.randint(8, max_step + 1) // 2) * 2
        nxt = (int(min(max(joint[0] + step * np.cos(turn), 0), h - 1)), int(min(max(joint[1] + step * np.sin(turn), 0), w - 1)))
        yield joint, nxt, brush
        joint = nxt


def _draw_polyline(h, w, max_segments, max_step, max_turn_deg):
    """a fresh [h,w] canvas with one polyline: thick segments of value 1, joint outlines (radius brush // 2) of value 2 -- the
    outlines saturate to 1 with everything else when the caller clips the accumulated canvas"""
    canvas = np.zeros((h, w), dtype=np.float32)
    end, brush = None, 0
    for a, b, brush in _polyline(h, w, max_segments, max_step, max_turn_deg):
        _line(canvas, a, b, brush)
        _circle_outline(canvas, a, brush // 2, 2)
        end = b
    if end is not None:
        _circle_outline(canvas, end, brush // 2, 2)
    return canvas


def generate_stroke_mask(im_size, maxVertex=4, maxAngle=360, percent_range=(0.0, 0.5), **unused):
    """-> (mask tensor [H,W] float32 in {0,1}, covered fraction).  Reference: models/IRNrhi_model.py:1147-1193 (the keyword names it
    passes are accepted; `parts`, `parts_square`, `maxLength`, `maxBrushWidth` never had an effect there -- the two sizes are
    overwritten by im_size[0] / 5 -- and are swallowed by **unused)."""
    h, w = int(im_size[0]), int(im_size[1])
    reach = int(h / 5)
    lo, hi = percent_range
    target = lo + (hi - lo) * np.random.rand()
    covered = np.zeros((h, w), dtype=np.float32)
    while True:
        np.minimum(covered + _draw_polyline(h, w, maxVertex, reach, maxAngle), 1.0, out=covered)
        if covered.mean() >= target:
            break
    return torch.from_numpy(covered).contiguous(), float(covered.mean())
