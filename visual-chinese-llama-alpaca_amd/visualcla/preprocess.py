"""GPU image preprocessing (next row N1): a CLIPImageProcessor-shaped object whose resize / crop / rescale / normalise run
in libvisualcla_hip.so, bit-exact with Pillow's bicubic resampler.  Opt-in: `attach_runtime(..., gpu_preprocess=True)` or
`model.image_processor = GpuClipImageProcessor.from_hf(hf_processor, device)`.

Host work that remains: decoding the file and `convert("RGB")` (PIL), and building two small coefficient tables per distinct
(H, W) (cached)."""
from __future__ import annotations

import ctypes as C
import math
from types import SimpleNamespace
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib

_PREC = 22


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _tables(in_size: int, out_size: int, first: int, count: int):
    """Pillow's precompute_coeffs for output indices [first, first + count): first tap, tap count, 22-bit weights."""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support, ss = 2.0 * fs, 1.0 / fs
    rows = []
    for i in range(first, first + count):
        center = (i + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        ks = [_bicubic((x - center + 0.5) * ss) for x in range(lo, hi)]
        ww = sum(ks)
        ki = [(int(math.floor(0.5 + k / ww * (1 << _PREC))) if k / ww >= 0 else -int(math.floor(0.5 - k / ww * (1 << _PREC)))) for k in ks]
        rows.append((lo, ki))
    kmax = max(len(k) for _, k in rows)
    lo = np.array([r[0] for r in rows], np.int32)
    n = np.array([len(r[1]) for r in rows], np.int32)
    k = np.zeros((count, kmax), np.int32)
    for i, (_, ks) in enumerate(rows):
        k[i, :len(ks)] = ks
    return lo, n, k


def plan_tables(H: int, W: int, S: int):
    """Coefficient tables of the S cropped output columns (horizontal pass over W) and rows (vertical pass over H) for
    CLIPImageProcessor's shortest-edge resize to S followed by an S x S centre crop."""
    short, long_ = (H, W) if H <= W else (W, H)
    new_long = int(S * long_ / short)
    oh, ow = (S, new_long) if H <= W else (new_long, S)
    top, left = (oh - S) // 2, (ow - S) // 2
    return _tables(W, ow, left, S), _tables(H, oh, top, S)


class GpuClipImageProcessor:
    model_input_names = ["pixel_values"]

    def __init__(self, size: int = 224, image_mean=(0.48145466, 0.4578275, 0.40821073),
                 image_std=(0.26862954, 0.26130258, 0.27577711), rescale_factor: float = 1 / 255, device="cuda:0",
                 dtype: torch.dtype = torch.float32):
        _lib.require_device()
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean, self.image_std, self.rescale_factor = list(image_mean), list(image_std), float(rescale_factor)
        self.device, self.dtype = torch.device(device), dtype
        self._cache: Dict[Tuple[int, int], tuple] = {}

    @classmethod
    def from_hf(cls, proc, device="cuda:0", dtype=torch.float32):
        """Mirror a transformers CLIPImageProcessor (its size / crop_size are dicts in 4.x, SizeDict objects in 5.x)."""
        def get(d, key):
            v = d.get(key) if isinstance(d, dict) else getattr(d, key, None)
            return v
        size = get(proc.size, "shortest_edge")
        if size is None:
            raise ValueError("GpuClipImageProcessor needs size['shortest_edge'] (the CLIP preprocessor layout)")
        if get(proc.crop_size, "height") != size or get(proc.crop_size, "width") != size:
            raise ValueError("GpuClipImageProcessor expects crop_size == shortest_edge (the CLIP default)")
        if int(getattr(proc, "resample", 3)) != 3:
            raise ValueError("GpuClipImageProcessor implements bicubic resampling only")
        return cls(size, proc.image_mean, proc.image_std, proc.rescale_factor, device, dtype)

    def _plan(self, H: int, W: int):
        key = (H, W)
        if key not in self._cache:
            h_tab, v_tab = plan_tables(H, W, self.size["shortest_edge"])
            dev = [torch.from_numpy(np.ascontiguousarray(t)).to(self.device) for t in (*h_tab, *v_tab)]
            self._cache[key] = (dev, h_tab[2].shape[1], v_tab[2].shape[1])
        return self._cache[key]

    @staticmethod
    def _as_uint8_hwc(image) -> torch.Tensor:
        if hasattr(image, "convert"):
            image = np.array(image.convert("RGB"))   # a writable copy: torch.as_tensor warns on PIL's read-only buffer
        img = torch.as_tensor(image)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("expected a uint8 HWC RGB image")
        return img

    def preprocess_batch(self, images: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """images: uint8 [N, H, W, 3] (host -- ideally pinned -- or device) of ONE size -> [N, 3, S, S]: one host-to-device copy,
        one launch pair (vcla_image_preprocess_batch), no per-image allocation."""
        lib = _lib.load()
        if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
            raise ValueError("expected a uint8 [N, H, W, 3] batch")
        N, H, W = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
        S = self.size["shortest_edge"]
        if out is None:
            out = torch.empty(N, 3, S, S, dtype=self.dtype, device=self.device)
        dev_imgs = images.to(self.device, non_blocking=True).contiguous()
        (h_lo, h_n, h_k, v_lo, v_n, v_k), hk, vk = self._plan(H, W)
        nbytes = N * H * S * 3
        # intermediate of the horizontal pass: one buffer PER STREAM (two calls on different streams must not share it), grown, never
        # shrunk -- and never freed while a launch may still read it: an outgrown buffer stays referenced until this stream has
        # passed the event recorded behind its last use (record_stream), so there is no per-call allocation and no cross-stream race
        tmps = self.__dict__.setdefault("_tmp_by_stream", {})
        skey = torch.cuda.current_stream(self.device).cuda_stream
        tmp = tmps.get(skey)
        if tmp is None or tmp.numel() < nbytes:
            tmp = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            tmp.record_stream(torch.cuda.current_stream(self.device))
            tmps[skey] = tmp
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        with torch.cuda.device(self.device):
            _lib.check(lib.vcla_image_preprocess_batch(dev_imgs.data_ptr(), N, H, W, tmp.data_ptr(), S, h_lo.data_ptr(), h_n.data_ptr(),
                                                       h_k.data_ptr(), hk, v_lo.data_ptr(), v_n.data_ptr(), v_k.data_ptr(), vk,
                                                       self.rescale_factor, mean, std, out.data_ptr(), _lib.dtype_code(out.dtype),
                                                       _lib.stream_ptr()))
        return out

    def preprocess_into(self, image, out: torch.Tensor) -> None:
        """image: PIL.Image or uint8 HWC array/tensor; out: [3, S, S] slice of the batch tensor (device, contiguous)."""
        self.preprocess_batch(self._as_uint8_hwc(image)[None], out=out[None])

    def __call__(self, images, return_tensors="pt", **kwargs):
        """CLIPImageProcessor's call: images of the same size are grouped into one staging buffer and one launch pair per group
        (a batch of same-sized frames / thumbnails is ONE group); output order follows the input order."""
        imgs: List = list(images) if isinstance(images, (list, tuple)) else [images]
        S = self.size["shortest_edge"]
        out = torch.empty(len(imgs), 3, S, S, dtype=self.dtype, device=self.device)
        arrs = [self._as_uint8_hwc(im) for im in imgs]
        groups: Dict[Tuple[int, int], List[int]] = {}
        for i, a in enumerate(arrs):
            groups.setdefault((int(a.shape[0]), int(a.shape[1])), []).append(i)
        for (H, W), idx in groups.items():
            if len(idx) == len(imgs):                       # one size: preprocess straight into the output tensor
                self.preprocess_batch(torch.stack(arrs) if len(idx) > 1 else arrs[0][None], out=out)
            else:
                res = self.preprocess_batch(torch.stack([arrs[i] for i in idx]))
                out[torch.tensor(idx, device=self.device)] = res
        return SimpleNamespace(pixel_values=out)
