"""GPU evaluation transform: the reference's ``Resize((r, r), BICUBIC) + ToTensor + Normalize`` (reference
dataset/transformer_dataset.py:38-42) on raw RGB uint8 batches, bit-exact with Pillow's resample, through
``xmh_image_preprocess_u8``.  The host only builds the per-size coefficient tables (a few hundred doubles, cached)."""
from __future__ import annotations

import ctypes
from typing import Dict, Tuple

import numpy as np
import torch

from .._lib import check, current_stream, lib, ptr

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)          # reference dataset/transformer_dataset.py:41
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_PRECISION_BITS = 32 - 8 - 2


def resample_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's bicubic coefficient tables for resizing ``in_size`` -> ``out_size`` pixels (whole-image box):
    bounds int32 [out, 2] = (first input index, tap count), kk int32 [out, ksize] = taps with 22 fractional bits.
    All arithmetic in float64 in Pillow's own order (the tap sum runs left to right)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)              # C (int) truncation; operands are positive or clamped
    xmin = np.where(center - support + 0.5 < 0, 0, xmin)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    x = (taps + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale)
    ax = np.abs(x)
    a = -0.5
    w = np.where(ax < 1.0, ((a + 2.0) * ax - (a + 3.0)) * ax * ax + 1, np.where(ax < 2.0, (((ax - 5) * ax + 8) * ax - 4) * a, 0.0))
    live = taps < xmax[:, None]
    w = np.where(live, w, 0.0)
    ww = np.zeros(out_size, np.float64)
    for t in range(ksize):                                                        # left-to-right like the C loop
        ww = ww + w[:, t]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    q = np.where(w < 0, -0.5 + w * (1 << _PRECISION_BITS), 0.5 + w * (1 << _PRECISION_BITS))
    kk = np.where(live, np.trunc(q), 0.0).astype(np.int64).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk


class GpuEvalTransform:
    """``transform(images_u8)``: uint8 RGB ``[B, H, W, 3]`` (or ``[H, W, 3]``), CPU or CUDA -> float32 ``[B, 3, r, r]`` on the GPU,
    equal to stacking the reference's eval transform over the batch."""

    def __init__(self, resolution: int = 224, mean=CLIP_MEAN, std=CLIP_STD):
        self.resolution = int(resolution)
        self._mean = (ctypes.c_float * 3)(*mean)
        self._std = (ctypes.c_float * 3)(*std)
        self._tables: Dict[tuple, tuple] = {}

    def _table(self, in_size: int, device) -> tuple:
        key = (in_size, str(device))
        t = self._tables.get(key)
        if t is None:
            b, k = resample_tables(in_size, self.resolution)
            t = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), int(k.shape[1]))
            self._tables[key] = t
        return t

    def _run(self, images: torch.Tensor, want_u8: bool, want_f32: bool):
        if not torch.cuda.is_available():
            raise RuntimeError("GpuEvalTransform needs an MI355X; there is no CPU fallback")
        if images.dtype != torch.uint8:
            raise TypeError("GpuEvalTransform takes raw uint8 RGB images, got %s" % images.dtype)
        if images.dim() == 3:
            images = images.unsqueeze(0)
        if images.dim() != 4 or images.shape[-1] != 3:
            raise ValueError("expected [B, H, W, 3] uint8, got %s" % (tuple(images.shape),))
        if not images.is_cuda:
            images = images.cuda(non_blocking=True)
        images = images.contiguous()
        B, H, W, _ = images.shape
        r, dev = self.resolution, images.device
        bw = kw = bh = kh = None
        ksw = ksh = 0
        tmp = None
        if W != r:
            bw, kw, ksw = self._table(W, dev)
            tmp = torch.empty(B, H, r, 3, dtype=torch.uint8, device=dev)
        if H != r:
            bh, kh, ksh = self._table(H, dev)
        out_u8 = torch.empty(B, r, r, 3, dtype=torch.uint8, device=dev) if want_u8 else None
        out_f = torch.empty(B, 3, r, r, dtype=torch.float32, device=dev) if want_f32 else None
        check(lib.xmh_image_preprocess_u8(ptr(images), B, H, W, r, r, ptr(bw), ptr(kw), ksw, ptr(bh), ptr(kh), ksh,
                                          ctypes.cast(self._mean, ctypes.c_void_p), ctypes.cast(self._std, ctypes.c_void_p),
                                          ptr(tmp), ptr(out_u8), ptr(out_f), current_stream()), "xmh_image_preprocess_u8")
        return out_u8, out_f

    def __call__(self, images) -> torch.Tensor:
        """a stacked uint8 batch, or a list of [H, W, 3] uint8 photos of any sizes (one launch pair per size group)."""
        if isinstance(images, (list, tuple)):
            if not images:
                raise ValueError("empty image list")
            dev = images[0].device if images[0].is_cuda else torch.device("cuda", torch.cuda.current_device())
            out = torch.empty(len(images), 3, self.resolution, self.resolution, dtype=torch.float32, device=dev)
            groups: Dict[tuple, list] = {}
            for i, im in enumerate(images):
                groups.setdefault(tuple(im.shape), []).append(i)
            for shape, idx in groups.items():
                batch = torch.stack([images[i] if images[i].is_cuda else images[i].to(dev, non_blocking=True) for i in idx])
                res = self._run(batch, False, True)[1]
                out[torch.tensor(idx, device=out.device)] = res
            return out
        return self._run(images, False, True)[1]

    def resize_u8(self, images: torch.Tensor) -> torch.Tensor:
        """only the Pillow-exact resize: uint8 [B, r, r, 3]."""
        return self._run(images, True, False)[0]
