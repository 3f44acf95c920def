#!/usr/bin/env python3
"""CLIP ViT-B/32 encode throughput on one MI355X (SURVEY 8d metric iii): images/s and captions/s of the HIP encoder +
DCMHT 64-bit head on device-resident synthetic batches (B=100, random-init weights of the reference architecture),
in parity mode ("f32": hi/lo-split fp32 activations x fp16-exact weights on the fp16 MFMA, fp32-grade error), exact mode
("f32x": fp32 MFMA) and fast mode ("f16": fp16 operands, fp32 accumulate), with the MFMA roofline of the GEMM kernels.

    python bench_encode.py [--batch 100 --steps 10]
Called by bench.py (``measure()``)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

# Flops the GEMM launches really execute per item (the TFLOP/s figures divide these by the GEMM kernel time).  SURVEY 2.2 counts
# 8.86 GFLOP per image and 2.46 per caption for the reference, which runs every token through every layer; since round 4 the last
# block's row-wise half (out_proj, c_fc, c_proj) runs on the one row per sequence the tower returns (cls / EOS):
#   image: 8.86 - 49/50 * 2 * 50 * (768^2 + 2 * 768 * 3072) = 8.86 - 0.52;   caption (per kept row fraction f of 32 rows):
#   12 blocks of 2 * 32 f * (512 * 1536 + 512^2 + 2 * 512 * 2048) minus the same tail, + the projection.
FLOP_IMAGE = 8.86e9 - 49 * 2 * (768 * 768 + 2 * 768 * 3072)
FLOP_IMAGE_REFERENCE = 8.86e9


def _flop_text(frac):
    per_row_block = 2.0 * (512 * 1536 + 512 * 512 + 2 * 512 * 2048)
    tail = 2.0 * (512 * 512 + 2 * 512 * 2048)
    rows = 32.0 * frac
    return 12 * rows * per_row_block - (rows - 1) * tail + 2.0 * 512 * 512


FLOP_TEXT = _flop_text(1.0)
# TFLOP/s dense MFMA peaks, MI355X_MICROARCH.md.  The split kernel spends two fp16 MFMAs per product: its ceiling in
# useful flops is half the fp16 peak.
PEAK = {"f32": 2500.0 / 2, "f32x": 157.3, "f16": 2500.0}
SLOTS = {"f32": ("gemm_s16", "gemm_f32"), "f32x": ("gemm_f32",), "f16": ("gemm_f16", "gemm_f32")}


def measure(batch=100, steps=10, warmup=5, K=64, modes=("f32", "f32x", "f16"), extras=True):
    from xmh import _lib, ops
    from xmh import retrieval as R
    from xmh.models.dcmht import DCMHT
    from xmh.utils.config import Config
    model = DCMHT.from_config(Config({"clip_path": "synthetic:1814"}), output_dim=K).cuda().eval()
    from xmh.models import weights as W
    image = W.synth_images(5, batch).cuda()
    ids, _ = W.synth_text(5, batch)
    ids = ids.cuda()
    out = {}
    # round 4: with only the EOS embedding wanted the text tower runs on the rows up to each caption's EOS (xmh_text_forward_packed,
    # bit-identical output); the synthetic captions of SURVEY 8d are 4-30 tokens + SOS + EOS of 32.  Rates are per caption either way;
    # the GEMM TFLOP/s count the rows that really ran, and the padded tower's rate is reported beside it.
    import xmh.models.clip as CM
    frac = float((ids.argmax(1) + 1).sum()) / ids.numel()
    packed = CM.TEXT_PACKING and frac < 0.9
    out["captions_rows_run_fraction"] = frac if packed else 1.0
    for mode in modes:
        ops.set_precision(mode)
        try:
            for what, fn, flop in (("images", lambda: R.pack_pair_argmax(model.encode_image(image)), FLOP_IMAGE),
                                   ("captions", lambda: R.pack_pair_argmax(model.encode_text(ids)), FLOP_TEXT)):
                for _ in range(warmup):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()                           # throughput: no profiling hooks, launches run ahead of the GPU
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                host = 0.0                                         # host time to enqueue one forward (GPU idle when it starts)
                for _ in range(3):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    fn()
                    host += (time.perf_counter() - t1) / 3
                out["%s_host_enqueue_ms_%s" % (what, mode)] = host * 1e3
                gemm_total = float("inf")                              # seconds of GEMM kernels per forward
                for _ in range(2):                                 # separate passes: HIP events around every GEMM launch; the smaller of two
                    _lib.prof_enable(True)                         # (a box stalled for tens of ms inside one pass now and then: 211 TF beside 38 k images/s)
                    for _ in range(steps):
                        fn()
                    torch.cuda.synchronize()
                    total = 0.0
                    for slot in SLOTS[mode]:
                        gemm_ms, launches = _lib.prof_read(slot)
                        total += gemm_ms * 1e-3 * launches / steps
                    _lib.prof_enable(False)
                    gemm_total = min(gemm_total, total)
                out["%s_per_s_%s" % (what, mode)] = batch / dt
                out["%s_ms_per_batch_%s" % (what, mode)] = dt * 1e3
                out["%s_gemm_tflops_%s" % (what, mode)] = (_flop_text(frac if packed else 1.0) if what == "captions" else flop) * batch / gemm_total / 1e12
                out["%s_gemm_share_%s" % (what, mode)] = gemm_total / dt
        finally:
            ops.set_precision("f32")
    if packed and "f32" in modes:                              # the padded text tower (what rounds 1-3 measured), parity mode
        CM.TEXT_PACKING = False
        try:
            fn = lambda: R.pack_pair_argmax(model.encode_text(ids))     # noqa: E731
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            out["captions_per_s_f32_padded_tower"] = batch * steps / (time.perf_counter() - t0)
        finally:
            CM.TEXT_PACKING = True
    if not extras:                                             # multi-GPU leg of bench.py: throughput of the chosen modes only
        return out
    # the eval transform on raw photo bytes (COCO-like 375 x 500 RGB uint8, device-resident)
    from xmh.dataset.preprocess import GpuEvalTransform
    tf = GpuEvalTransform(224)
    raw = torch.randint(0, 256, (batch, 375, 500, 3), dtype=torch.uint8, device="cuda")
    for _ in range(warmup):
        tf(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tf(raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pp_bytes = batch * (375 * 500 * 3 + 2 * 375 * 224 * 3 + 224 * 224 * 12)          # read in, write+read the horizontal pass, write floats
    out["preprocess"] = {"images_per_s": batch / dt, "ms_per_batch": dt * 1e3, "achieved_GBps": pp_bytes / dt / 1e9,
                         "workload": "Resize((224,224), BICUBIC) + ToTensor + Normalize of %d RGB uint8 images 375x500, Pillow-exact" % batch}
    ach = out["images_gemm_tflops_f32"]
    # the evaluation loop fuses run.encode_fuse (default 4) loader batches into one forward (BaseTrainer.encode_streams): the
    # same towers at batch 400, where the GEMM grids fill the chip
    fused = {}
    big_img, big_ids = image.repeat(4, 1, 1, 1), ids.repeat(4, 1)
    for mode in ("f32", "f16"):
        ops.set_precision(mode)
        try:
            for what, fn in (("images", lambda: R.pack_pair_argmax(model.encode_image(big_img))), ("captions", lambda: R.pack_pair_argmax(model.encode_text(big_ids)))):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    fn()
                torch.cuda.synchronize()
                fused["%s_per_s_%s" % (what, mode)] = 4 * batch * steps / (time.perf_counter() - t0)
                if what == "images":                                # the GEMM launches of the fused forward: HIP events per launch, as above
                    _lib.prof_enable(True)
                    for _ in range(steps):
                        fn()
                    torch.cuda.synchronize()
                    gemm_total = 0.0
                    for slot in SLOTS[mode]:
                        gemm_ms, launches = _lib.prof_read(slot)
                        gemm_total += gemm_ms * 1e-3 * launches / steps
                    _lib.prof_enable(False)
                    fused["images_gemm_tflops_%s" % mode] = FLOP_IMAGE * 4 * batch / gemm_total / 1e12
        finally:
            ops.set_precision("f32")
    fused["workload"] = "the same towers at batch %d (4 fused loader batches, what valid() runs)" % (4 * batch)
    out["fused_batches"] = fused
    # image + caption pairs with the text tower on a second HIP stream under the image tower (xmh/towers.py: what the runners'
    # generate_hash does) against the two towers back to back
    from xmh import towers
    pairs = {}
    for tag, im, tx, n in (("b%d" % batch, image, ids, batch), ("b%d" % (4 * batch), big_img, big_ids, 4 * batch)):
        def both():
            a, b = towers.run_both(lambda: model.encode_image(im), lambda: model.encode_text(tx))
            R.pack_pair_argmax(a)
            R.pack_pair_argmax(b)
        for env in ("1", "0"):
            os.environ["XMH_TOWER_STREAMS"] = env
            for _ in range(3):
                both()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                both()
            torch.cuda.synchronize()
            pairs["pairs_per_s_%s_%s" % (tag, "two_streams" if env == "1" else "one_stream")] = n * steps / (time.perf_counter() - t0)
    os.environ.pop("XMH_TOWER_STREAMS", None)
    pairs["workload"] = "image + caption pairs through both towers and the DCMHT head, parity mode"
    out["both_towers"] = pairs
    out["roofline"] = {"kernel": "k_gemm_g16 with two activation planes (parity mode: all GEMM launches of one image forward, HIP events per launch)", "bound": "mfma",
                       "achieved": ach, "peak": PEAK["f32"], "unit": "TFLOP/s", "frac": ach / PEAK["f32"], "traffic": None,
                       "note": "useful flops; parity mode issues two fp16 MFMAs per product (hi and lo activation planes), so its ceiling is half the 2.5 PFLOP/s fp16 peak",
                       "exact_mode": {"kernel": "k_gemm_nt_f32 (v_mfma_f32_32x32x2_f32)", "achieved": out["images_gemm_tflops_f32x"],
                                      "peak": PEAK["f32x"], "frac": out["images_gemm_tflops_f32x"] / PEAK["f32x"]},
                       "fast_mode": {"kernel": "k_gemm_g16, one plane per operand", "achieved": out["images_gemm_tflops_f16"], "peak": PEAK["f16"],
                                     "frac": out["images_gemm_tflops_f16"] / PEAK["f16"]}}
    # the same at the batch valid() actually runs (run.encode_fuse = 4 loader batches per forward): 400 x 50 tokens = 20 000 rows, the
    # GEMM grids fill the chip and the tile quantisation of the 5000-row grids (720 tiles on 512 slots) is gone
    out["roofline"]["fused_batch_400"] = {
        "parity_mode": {"achieved": fused["images_gemm_tflops_f32"], "peak": PEAK["f32"], "frac": fused["images_gemm_tflops_f32"] / PEAK["f32"]},
        "fast_mode": {"achieved": fused["images_gemm_tflops_f16"], "peak": PEAK["f16"], "frac": fused["images_gemm_tflops_f16"] / PEAK["f16"]},
        "workload": fused["workload"]}
    # state the fractions against the 2.5 PFLOP/s dense fp16 peak on USEFUL flops for both modes (parity issues twice its
    # useful flops, so the "parity ceiling" of 1.25 PF above is the kinder number); summary.encode* carries these
    out["gemm_frac_of_fp16_peak_f32"] = ach / PEAK["f16"]
    out["gemm_frac_of_fp16_peak_f16"] = out["images_gemm_tflops_f16"] / PEAK["f16"]
    fused["gemm_frac_of_fp16_peak_f32"] = fused["images_gemm_tflops_f32"] / PEAK["f16"]
    fused["gemm_frac_of_fp16_peak_f16"] = fused["images_gemm_tflops_f16"] / PEAK["f16"]
    out["config"] = {"workload": "CLIP ViT-B/32 + DCMHT %d-bit head, batch %d, 224x224 / 32 tokens, random-init weights" % (K, batch)}
    return out


def measure_mith(batch=100, steps=10, warmup=5, K=64):
    """BASELINE configs[2]'s method: MITH (CLIP in return_patches mode + the concept/token hash head), parity mode, batch 100."""
    import xmh.models  # noqa: F401
    from xmh.common.register import registry
    from xmh.models import weights as W
    from xmh.utils.config import Config
    model = registry.get_model_class("MITH").from_config(Config({"clip_path": "synthetic:1814"}), output_dim=K, train_num=1000).cuda().eval()
    image = W.synth_images(5, batch).cuda()
    ids, _ = W.synth_text(5, batch)
    ids = ids.cuda()
    kpm = ids == 0
    out = {"config": {"workload": "CLIP ViT-B/32 (return_patches) + MITH %d-bit head, batch %d, parity mode, random-init weights" % (K, batch)}}
    with torch.no_grad():
        for what, fn in (("images", lambda: model.encode_image(image)), ("captions", lambda: model.encode_text(ids, kpm))):
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[what + "_per_s"] = batch / dt
            out[what + "_ms_per_batch"] = dt * 1e3
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    print(json.dumps(measure(a.batch, a.steps)))
