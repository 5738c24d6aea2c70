"""CPU oracle for next-row N1 (image preprocessing): an integer-exact numpy restatement of what the reference's
`CLIPImageProcessor` call does (models/visualcla/modeling_utils.py:130,150-152): PIL bicubic resize of the shortest edge,
centre crop, 1/255 rescale, CLIP mean/std normalisation.

TEST INFRASTRUCTURE ONLY.  Pinned in tests/test_preprocess_oracle.py against Pillow's own `Image.resize(BICUBIC)` and
against `transformers.CLIPImageProcessor` (both present in this image and on the GPU box): bit-exact.

PIL's resampler (third-party, Pillow `src/libImaging/Resample.c`) is a separable convolution with fixed-point coefficients:
  scale = in/out, filterscale = max(scale, 1), support = 2 * filterscale (bicubic, a = -0.5);
  per output index: centre = (i + .5) * scale, taps [int(centre - support + .5), int(centre + support + .5)) clipped to the
  image, weights bicubic((x - centre + .5) / filterscale) normalised to sum 1, then rounded to 22-bit fixed point;
  each pass accumulates from 1 << 21, shifts by 22 and clips to uint8 (horizontal pass first, 8-bit intermediate).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> List[Tuple[int, List[int]]]:
    """[(first tap, fixed-point weights)] per output index."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support, ss = 2.0 * fs, 1.0 / fs
    out = []
    for i in range(out_size):
        center = (i + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        ks = [_bicubic((x - center + 0.5) * ss) for x in range(lo, hi)]
        ww = sum(ks)
        ki = []
        for k in ks:
            k = k / ww
            ki.append(int(math.floor(0.5 + k * (1 << PRECISION_BITS))) if k >= 0 else -int(math.floor(0.5 - k * (1 << PRECISION_BITS))))
        out.append((lo, ki))
    return out


def coeff_tables(in_size: int, out_size: int, first: int, count: int):
    """dense int32 tables for output indices [first, first+count): (lo [count], n [count], k [count, kmax])"""
    cs = resample_coeffs(in_size, out_size)[first:first + count]
    kmax = max(len(k) for _, k in cs)
    lo = np.array([c[0] for c in cs], np.int32)
    n = np.array([len(c[1]) for c in cs], np.int32)
    k = np.zeros((count, kmax), np.int32)
    for i, (_, ks) in enumerate(cs):
        k[i, :len(ks)] = ks
    return lo, n, k


def resized_shape(h: int, w: int, shortest_edge: int) -> Tuple[int, int]:
    """transformers get_resize_output_image_size(default_to_square=False): short side -> S, long side -> int(S*long/short)"""
    short, long_ = (h, w) if h <= w else (w, h)
    new_short, new_long = shortest_edge, int(shortest_edge * long_ / short)
    return (new_short, new_long) if h <= w else (new_long, new_short)


def resize_bicubic_u8(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    h, w, c = img.shape
    if (oh, ow) == (h, w):
        return img.copy()
    tmp = np.zeros((h, ow, c), np.uint8)
    for x, (lo, ki) in enumerate(resample_coeffs(w, ow)):
        acc = np.full((h, c), 1 << (PRECISION_BITS - 1), np.int64)
        for t, k in enumerate(ki):
            acc += img[:, lo + t, :].astype(np.int64) * k
        tmp[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((oh, ow, c), np.uint8)
    for y, (lo, ki) in enumerate(resample_coeffs(h, oh)):
        acc = np.full((ow, c), 1 << (PRECISION_BITS - 1), np.int64)
        for t, k in enumerate(ki):
            acc += tmp[lo + t, :, :].astype(np.int64) * k
        out[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def clip_preprocess(img: np.ndarray, size: int = 224, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """uint8 HWC RGB -> float32 [3, size, size]"""
    h, w, _ = img.shape
    oh, ow = resized_shape(h, w, size)
    r = resize_bicubic_u8(img, oh, ow)
    top, left = (oh - size) // 2, (ow - size) // 2
    # hf:image_transforms.py rescale(): float64 multiply, then the downcast; normalize(): float32 (x - mean) / std
    crop = (r[top:top + size, left:left + size].astype(np.float64) * (1 / 255)).astype(np.float32)
    x = (crop - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))
