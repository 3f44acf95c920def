import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/clip-based-cross-modal-hash_amd"]
import torch
from xmh import ops, towers, retrieval as R
from xmh.models.dcmht import DCMHT
from xmh.utils.config import Config
from xmh.models import weights as W
model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64).cuda().eval()
for B in (100, 200, 400):
    image = W.synth_images(5, 100).cuda().repeat(B // 100, 1, 1, 1)
    def t(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
    for mode in ("f32", "f16"):
        ops.set_precision(mode)
        whole = t(lambda: model.encode_image(image))
        h = B // 2
        a, b = image[:h].contiguous(), image[h:].contiguous()
        split = t(lambda: towers.run_both(lambda: model.encode_image(a), lambda: model.encode_image(b)))
        print("B=%d %s: one forward %.3f ms (%.0f img/s)   two halves on two streams %.3f ms (%.0f img/s)" % (B, mode, whole * 1e3, B / whole, split * 1e3, B / split))
