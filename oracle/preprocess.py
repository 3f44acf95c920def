"""Oracle for the evaluation image transform.  TEST INFRASTRUCTURE ONLY.

The reference's eval transform is torchvision's
    Compose([Resize((224, 224), interpolation=Image.BICUBIC), ToTensor(), Normalize(CLIP mean, CLIP std)])
(reference dataset/transformer_dataset.py:38-42) applied to a PIL RGB image.  The arithmetic lives in third-party
dependencies that are not under /root/reference:
  * Pillow (requirements.txt:3, unpinned; 12.2.0 in the build container) -- ``Image.resize`` =
    ``ImagingResample`` (src/libImaging/Resample.c): two passes (horizontal, then vertical) of an 8-bit fixed-point
    convolution with per-output-pixel bicubic coefficients (a = -0.5, support 2 * max(scale, 1)), coefficients rounded to
    22 fractional bits, accumulator started at 2^21, result ``>> 22`` and clipped to [0, 255] after EACH pass;
  * torchvision 0.10 (requirements.txt:14) -- ``ToTensor`` = HWC uint8 -> CHW float32 ``/ 255``; ``Normalize`` =
    ``(x - mean) / std`` in float32.
This file restates that published algorithm; it is pinned by tests/golden/preprocess.npz, whose expected outputs were
produced by Pillow itself in the build container (oracle/make_golden_preprocess.py).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)          # reference dataset/transformer_dataset.py:41
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x):
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


def resample_coeffs(in_size: int, out_size: int):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the box (0, in_size): bounds [out, 2] = (first input
    index, count), kk [out, ksize] int32 fixed-point weights (zero beyond count)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = _bicubic((np.arange(xmax, dtype=np.float64) + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:                                   # sequential double sum, like the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = np.trunc(q).astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, xmax = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(xmax):
            acc += img[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """[H, W, 3] uint8 -> [out_h, out_w, 3] uint8 == PIL ``Image.resize((out_w, out_h), Image.BICUBIC)``."""
    H, W, _ = img.shape
    out = img
    if W != out_w:
        out = _pass(out, *resample_coeffs(W, out_w), axis=1)
    if H != out_h:
        out = _pass(out, *resample_coeffs(H, out_h), axis=0)
    return out


def to_tensor_normalize(img_u8: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """ToTensor + Normalize: [H, W, 3] uint8 -> [3, H, W] float32, every step in float32 like torch does it."""
    x = img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return ((x - m) / s).astype(np.float32)


def eval_transform(img_u8: np.ndarray, resolution: int = 224) -> np.ndarray:
    """the whole eval transform of reference dataset/transformer_dataset.py:38-42 on one RGB uint8 image."""
    return to_tensor_normalize(resize_bicubic_u8(img_u8, resolution, resolution))
