#!/usr/bin/env python3
"""Experiment: one image forward of batch B against the same batch as S slices on S HIP streams (kernels of different slices overlap:
the latency-bound attention / LayerNorm launches of one slice under the GEMMs of another).  python tools/exp_split_streams.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh.models.dcmht import DCMHT
from xmh.utils.config import Config
from xmh.models import weights as W

model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=64).cuda().eval()
streams = [torch.cuda.Stream() for _ in range(4)]


def split_forward(image, S):
    cur = torch.cuda.current_stream()
    parts = image.chunk(S)
    outs = []
    for st, p in zip(streams, parts):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(model.encode_image(p))
    for st in streams[:S]:
        cur.wait_stream(st)
    return torch.cat(outs)


for B in (100, 200, 400):
    image = W.synth_images(5, 100).cuda().repeat(B // 100, 1, 1, 1)
    ref = model.encode_image(image)
    for S in (1, 2, 4):
        fn = (lambda: model.encode_image(image)) if S == 1 else (lambda: split_forward(image, S))
        out = fn()
        assert torch.equal(out, ref), (B, S)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("B=%d slices=%d  %.3f ms  %.0f images/s" % (B, S, dt * 1e3, B / dt))
