#!/usr/bin/env python3
"""attention kernel timing (run on the GPU box): python tools/bench_attention.py  [XMH_ATTENTION_VALU=1 for the VALU kernel]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops
SHAPES = ((100, 50, 12, False), (100, 32, 8, True), (100, 64, 12, False))
for B, L, H, causal in (SHAPES[:1] if os.environ.get("XMH_ATT_ONE") else SHAPES):
    qkv = torch.randn(B, L, 3 * 64 * H, device="cuda")
    for _ in range(3):
        ops.attention(qkv, H, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.attention(qkv, H, causal=causal)
    e1.record()
    torch.cuda.synchronize()
    print("B=%d L=%d H=%d causal=%s  %.1f us" % (B, L, H, causal, e0.elapsed_time(e1) / 50 * 1e3))
