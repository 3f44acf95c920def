"""Generate tests/golden/preprocess.npz with Pillow's own ``Image.resize(..., BICUBIC)`` (the arithmetic behind the
reference's eval transform, dataset/transformer_dataset.py:38-42) and torch's float32 ToTensor/Normalize formula.
TEST INFRASTRUCTURE ONLY; runs in the build container.   python oracle/make_golden_preprocess.py"""
import os

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def pattern(H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    smooth = np.stack([128 + 100 * np.sin(xx / 13.0) * np.cos(yy / 7.0), (xx * 3 + yy * 5) % 256, 255 * (((xx // 16) + (yy // 16)) % 2)], -1)
    noise = rng.integers(0, 256, (H, W, 3))
    mix = np.where(rng.random((H, W, 1)) < 0.5, smooth, noise)
    return mix.clip(0, 255).astype(np.uint8)


def main():
    out = {"pillow_version": np.array(Image.__version__ if hasattr(Image, "__version__") else "?")}
    for n, (H, W) in enumerate([(96, 128), (224, 224), (60, 45), (150, 301), (224, 100)]):
        img = pattern(H, W, 100 + n)
        res = np.asarray(Image.fromarray(img, mode="RGB").resize((224, 224), Image.BICUBIC))
        t = torch.from_numpy(res.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)      # torchvision F.to_tensor
        t = t.sub(torch.as_tensor(MEAN, dtype=torch.float32)[:, None, None]).div(torch.as_tensor(STD, dtype=torch.float32)[:, None, None])
        out["img%d" % n] = img
        out["resized%d" % n] = res
        if n == 0:                                     # the float stage is elementwise: one case pins it
            out["tensor%d" % n] = t.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "preprocess.npz"), **out)
    print("wrote preprocess.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
