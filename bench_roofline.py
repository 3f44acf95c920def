"""Roofline bookkeeping of the benchmark lines: per-opcode VALU issue cost from the committed micro-benchmark output
(profiles/r02_ubench_valu.txt, produced on the MI355X by tools/ubench_valu.hip), the instruction mix of the two scan passes
(ISA counts, profiles/r02_scan_isa.txt, produced by tools/isa_count.py from the very build that runs), HBM traffic from the
committed rocprofv3 PMC summaries, and the HIP-event kernel timings taken live.

Every `frac` in the bench line can be recomputed from files under profiles/ plus the live timings in the line itself:
  VALU lane-ops/s achieved = pairs x instructions-per-pair / launch time
  peak (guide)             = 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 7.86e13 lane-ops/s (MI355X_MICROARCH.md: a wave64 VALU
                             instruction issues over 2 cycles on a SIMD-32)
  peak (measured mix)      = 256 x 4 x 64 lanes x 2.4 GHz / (mean measured cycles per instruction of THIS kernel's mix): on gfx950
                             only v_xor/v_and/v_or/v_add/v_fmac/v_mov reach ~2.5 cycles; v_bcnt, v_and_or, v_lshl_add/or,
                             v_min/max, v_alignbyte, v_bfe, v_cvt, v_mul_u24 take ~4.2 and v_rcp_f32 8.2 (ubench, 8 waves/SIMD)
"""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
CLOCK_HZ = 2.4e9
SIMDS = 256 * 4
VALU_PEAK_GUIDE = SIMDS * 32 * CLOCK_HZ / 1e9          # G lane-ops/s at 2 cycles per wave64 instruction = 78 643

# opcode class -> name of the line in the ubench output that measures it
_UBENCH_LINE = {"full": "v_xor_b32", "half": "v_bcnt_u32_b32", "trans": "v_rcp_f32"}
# which class an opcode of the scan loops belongs to (measured individually: see the file)
_CLASS = {"v_xor_b32": "full", "v_and_b32": "full", "v_or_b32": "full", "v_add_u32": "full", "v_fmac_f32": "full", "v_mov_b32": "full",
          "v_fma_f32": "full", "v_sub_u32": "full", "v_add_f32": "full", "v_cndmask_b32": "full",
          "v_rcp_f32": "trans"}


def ubench_cycles(wps=8):
    """{class: cycles per wave64 instruction} from the newest committed ubench output (None if absent)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_ubench_valu.txt")))
    if not files:
        return None, None
    out = {}
    for line in open(files[-1]):
        m = re.match(r"(\S.*?)\s+wps=(\d+)\s+[\d.]+ ms\s+([\d.]+) cyc/inst", line)
        if m and int(m.group(2)) == wps:
            for cls, name in _UBENCH_LINE.items():
                if m.group(1).strip() == name:
                    out[cls] = float(m.group(3))
    return (out if len(out) == 3 else None), os.path.relpath(files[-1], ROOT)


def isa_mix(kernel_key):
    """{opcode: count per 64 pairs} of a kernel's steady-state loop from profiles/r*_scan_isa.json (tools/isa_count.py)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_scan_isa.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    return d.get(kernel_key), os.path.relpath(files[-1], ROOT)


def mix_cycles(mix, cyc):
    """mean measured issue cycles per VALU instruction of a mix, and instructions per 64 pairs"""
    n = sum(mix.values())
    c = sum(cnt * cyc["trans" if op.startswith(("v_rcp", "v_rsq", "v_exp", "v_log", "v_sqrt")) else _CLASS.get(op, "half")] for op, cnt in mix.items())
    return c / n, n


def _norm(name):
    return re.sub(r"\s+", "", name)


def _newest_summary():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_summary.json")))
    for f in reversed(files):
        try:
            return json.load(open(f)), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def pmc_row(kernel_instance):
    """The PMC entry of EXACTLY this kernel instance (e.g. "k_scan_hist_r2<2, 4, 4, true>", from xmh_scan_describe) in the
    newest committed rocprofv3 summary, or (None, source).  Round 2 matched by prefix and kept the last hit, which handed the
    64-bit headline the 128-bit kernel's counters; an exact name cannot do that, and two rows normalising to one name raise."""
    d, src = _newest_summary()
    if not d:
        return None, None
    hits = [(n, e) for n, e in d.get("pmc", {}).items() if _norm(n) == _norm(kernel_instance)]
    if len(hits) > 1:
        raise RuntimeError("bench_roofline: %d PMC rows match %r in %s" % (len(hits), kernel_instance, src))
    return (hits[0][1] if hits else None), src


def pmc_traffic(kernel_instance):
    """HBM bytes per launch of a kernel instance (FETCH_SIZE / WRITE_SIZE passes with the gfx950 corrections of MI355X_MICROARCH.md,
    tools/summarize_profile.py); None if the newest profile has no row for it."""
    e, src = pmc_row(kernel_instance)
    if not e or "hbm_bytes_per_launch" not in e:
        return None
    h = e["hbm_bytes_per_launch"]
    return {"bytes": h["total"], "fetch_raw": h["fetch_raw"], "write_raw": h["write_raw"], "fetch_correction": h["fetch_correction"],
            "kernel": kernel_instance, "source": src}


def pmc_counter(kernel_instance, counter):
    e, src = pmc_row(kernel_instance)
    if not e or counter not in e.get("per_launch", {}):
        return None
    return e["per_launch"][counter], src


def scan_kernels(Q, Rn, K, C, tern):
    """(pass-1 kernel instance, [pass-2 kernel instances]) this shape launches: xmh_scan_describe"""
    import ctypes
    from xmh import _lib
    buf = ctypes.create_string_buffer(1024)
    _lib.check(_lib.lib.xmh_scan_describe(Q, Rn, K, C, int(tern), buf, 1024), "xmh_scan_describe")
    parts = dict(x.split("=", 1) for x in buf.value.decode().split(";"))
    return parts["pass1"], parts["pass2"].split("|")


def _pass_times(scan, steps):
    import torch
    from xmh import _lib
    _lib.prof_enable(True)
    for _ in range(steps):
        scan.histograms(False)
        scan.ap_sums(None)
    torch.cuda.synchronize()
    t_hist, n_hist = _lib.prof_read("scan_hist")
    # pass 2 exists in two counter widths (packed 32-bit / 64-bit); the device-side gate lets exactly one of them do the work
    t64, n64 = _lib.prof_read("scan_ap")
    t32, n32 = _lib.prof_read("scan_ap32")
    _lib.prof_enable(False)
    packed = t32 > t64
    t_ap, n_ap = (t32, n32) if packed else (t64, n64)
    return t_hist * 1e-3, n_hist, t_ap * 1e-3, n_ap, packed


def scan_roofline(scan, Q, Rn, K, C, steps=20, step_s=None):
    """`roofline` object of the headline step; the dominant kernel = the longer of the two passes.

    Q=5000 queries share every gallery byte, so the step cannot be HBM-bound: the bound is VALU lane-ops.  Two numbers,
    labelled as what they are:
      * `frac` / `algorithmic`: SURVEY 8(d)'s per-pair work -- K/32 XOR + K/32 popcount-accumulate + Lw AND/OR = 2W + Lw lane-ops --
        x pairs per launch / the kernel's HIP-event time / the guide's VALU peak (256 CU x 4 SIMD x 32 lanes x 2.4 GHz).  The
        kernels do that work on the i8 MFMA and with LDS atomics, but the figure stays the reference formulation's.
      * `issue_utilisation`: the kernel's OWN dynamic instruction count (SQ_INSTS_VALU of exactly this kernel instance in the
        committed profile) / the same peak -- how busy the VALU issue port is, not an algorithmic fraction.
    `traffic` = the PMC HBM bytes of that same instance (its own pair cache and bucket tables, not re-reads of the gallery)."""
    from xmh import _lib
    t_hist, n_hist, t_ap, n_ap, packed = _pass_times(scan, steps)
    W, Lw = (K + 31) // 32, (C + 31) // 32
    tern = scan.qz is not None
    alg_bytes = Rn * 4 * (W + Lw) + Q * 4 * (W + Lw) + Q * 12          # gallery once + queries + ap_sum/cap out
    alg_ops = (2 * W + Lw) if not tern else (4 * W + Lw)             # ternary: the zero planes double the code-word work
    pl = scan.plan
    table_bytes = (pl.nchunk + 1) * pl.nbuckets * pl.qpad * 8 + pl.nchunk * pl.qpad * 4
    cache_bytes = int(_lib.lib.xmh_scan_pair_cache_bytes(Q, Rn, K, int(tern)))
    pairs = Q * Rn
    k1, k2s = scan_kernels(Q, Rn, K, C, tern)
    k2 = k2s[0] if (packed or len(k2s) == 1) else k2s[-1]              # both counter widths are launched beyond 64 bits: the one that ran
    cyc, cyc_src = ubench_cycles()

    def isa_key(kernel):
        m = re.match(r"(k_scan_\w+)<(.*)>", kernel)
        return (m.group(1) + "<" + re.sub(r"\s+", "", m.group(2)) + ">") if m else kernel

    def pass_entry(kernel, t):
        e = {"kernel": kernel, "avg_launch_ms": t * 1e3,
             "algorithmic": {"lane_ops_per_pair": alg_ops, "achieved": pairs * alg_ops / t / 1e9, "unit": "G lane-ops/s", "peak": VALU_PEAK_GUIDE,
                             "frac": pairs * alg_ops / t / 1e9 / VALU_PEAK_GUIDE}}
        mix, src = isa_mix(isa_key(kernel))
        if mix:
            e["static_hot_path_valu_per_64_pairs"] = sum(mix.values())
            e["isa_source"] = src
            e["mix"] = mix
            if cyc:
                mean_c, _ = mix_cycles(mix, cyc)
                e["mean_measured_cycles_per_valu_instruction"] = mean_c
                e["ubench_source"] = cyc_src
        dyn = pmc_counter(kernel, "SQ_INSTS_VALU")
        if dyn is not None and (Q, Rn, K, C) == (5000, 117218, 64, 80):      # the committed profile is of the default shape
            ops = dyn[0] / (pairs / 64.0)
            e["issue_utilisation"] = {"dynamic_valu_and_mfma_per_64_pairs": ops, "frac_of_guide_peak": pairs * ops / t / 1e9 / VALU_PEAK_GUIDE,
                                      "source": dyn[1] + ": SQ_INSTS_VALU per launch of this kernel instance / (pairs / 64); counts the MFMAs too",
                                      "what": "issue-slot utilisation by the kernel's own instruction count, NOT an algorithmic fraction"}
            for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES",
                      "SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS"):
                v = pmc_counter(kernel, c)
                if v is not None:
                    e["issue_utilisation"][c] = v[0]
        return e
    v1, v2 = pass_entry(k1, t_hist), pass_entry(k2, t_ap)
    if k1.startswith(("k_scan_hist_r2", "k_scan_hist_b")):
        nml = 1 if Lw <= 2 else 2
        nmc = 1 if K <= 64 else (2 if K <= 128 else 4)
        chains = (2 * nmc if (k1.startswith("k_scan_hist_r2") and cache_bytes) else nmc) + nml       # address chain (+ the 2 * distance chain of the cache byte) + label tiles
        v1["mfma"] = {"instruction": "v_mfma_i32_16x16x64_i8", "per_64_pairs": chains / 4.0, "cycles_each": 16,
                      "matrix_pipe_frac": pairs / 64.0 * (chains / 4.0) * 16 / (t_hist * CLOCK_HZ * SIMDS)}
    dom_is_hist = t_hist > t_ap
    t_dom, n_dom, vd, kd = (t_hist, n_hist, v1, k1) if dom_is_hist else (t_ap, n_ap, v2, k2)
    traffic = pmc_traffic(kd)
    alg = dict(vd["algorithmic"])
    alg["frac_dominant"] = alg.pop("frac")
    if step_s:
        alg["frac_step"] = pairs * alg_ops / step_s / 1e9 / VALU_PEAK_GUIDE
        alg["step_ms"] = step_s * 1e3
    return {
        "kernel": "%s (pass %d of the fused mAP scan), HIP events around the launch, %d launches" % (kd, 1 if dom_is_hist else 2, n_dom),
        "bound": "valu", "achieved": alg["achieved"], "peak": VALU_PEAK_GUIDE, "unit": "G lane-ops/s", "frac": alg["frac_dominant"],
        "algorithmic": alg, "issue_utilisation": vd.get("issue_utilisation"),
        "frac_step": alg.get("frac_step"),
        "traffic": (traffic or {}).get("bytes"), "traffic_detail": traffic,
        "hbm": {"bound": "hbm", "algorithmic_bytes": alg_bytes, "achieved": alg_bytes / t_dom / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS,
                "note": "reported because the contract names it: Q=%d queries share every gallery byte, the launch cannot be HBM-bound" % Q},
        "workspace_table_bytes": table_bytes, "pair_cache_bytes": cache_bytes, "avg_launch_ms": t_dom * 1e3, "ternary": tern,
        "pass1_avg_launch_ms": t_hist * 1e3, "pass2_avg_launch_ms": t_ap * 1e3, "pass1": v1, "pass2": v2,
        "note": "frac = algorithmic lane-ops (2W + Lw per pair, SURVEY 8d) x pairs / launch time / the guide's VALU peak; "
                "issue_utilisation is the kernel's own instruction count against the same peak.  PMC traffic above the algorithmic bytes is "
                "the scheme's own data (bucket tables + the pair cache: one byte per pair written by pass 1, read by pass 2), not re-reads "
                "of the gallery",
    }


def synth_gpu(Q, R, K, C, seed, p=0.04):
    """SURVEY 8d synthetic inputs (label-correlated +-1 codes, multi-hot labels with >= 1 label per row) generated ON the GPU --
    the extra legs go up to 1.25 M x 256 bit, which the CPU generator of the headline would spend seconds on.  Returns packed
    (q, ql, r, rl).  tests/test_gpu_retrieval.py builds the same tensors from the same seed to check the legs against the oracle."""
    import torch
    from xmh import retrieval as R_
    g = torch.Generator(device="cuda").manual_seed(seed)
    Wm = torch.randn(C, K, generator=g, device="cuda")

    def side(n):
        codes, labs = [], []
        for lo in range(0, n, 262144):                               # bounded temporaries
            m = min(262144, n - lo)
            L = torch.rand(m, C, generator=g, device="cuda") < p
            L[torch.arange(m, device="cuda"), torch.randint(0, C, (m,), generator=g, device="cuda")] = True
            B = (L.float() @ Wm + 0.8 * torch.randn(m, K, generator=g, device="cuda")).sign()
            B[B == 0] = 1
            codes.append(R_.pack_sign(B).bits)
            labs.append(R_.pack_labels(L.to(torch.uint8)))
        return R_.PackedCodes(torch.cat(codes), None, K), torch.cat(labs)
    q, ql = side(Q)
    r, rl = side(R)
    return q, ql, r, rl


def extra_scan_leg(what, Q, Rn, K, C, p_label, seed, steps=30, return_scan=False):
    """one more driver-visible scan shape: whole-step time, both pass times and the mAP (checked against the oracle on a query
    subsample by tests/test_gpu_retrieval.py::test_bench_legs_full_shapes_match_the_oracle, which calls this very function)"""
    import time
    import torch
    from xmh import _lib
    from xmh import retrieval as R
    q, ql, r, rl = synth_gpu(Q, Rn, K, C, seed, p_label)
    scan = R.RankingScan(q, ql, r, rl, C)
    for _ in range(min(10, steps)):
        scan.histograms(False)
        m = scan.map_all(None)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        scan.histograms(False)
        m = scan.map_all(None)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t_hist, _, t_ap, _, packed = _pass_times(scan, min(10, steps))
    k1, k2s = scan_kernels(Q, Rn, K, C, False)
    W, Lw = (K + 31) // 32, (C + 31) // 32
    out = {"workload": "%s: Q=%d x R=%d, %d bit, C=%d, mAP@all" % (what, Q, Rn, K, C), "ms_per_step": dt * 1e3,
           "pairs_per_s": Q * Rn / dt, "mAP": float(m.item()), "pass1_ms": t_hist * 1e3, "pass2_ms": t_ap * 1e3,
           "pass1_kernel": k1, "pass2_kernel": k2s[0] if (packed or len(k2s) == 1) else k2s[-1],
           "pair_cache_bytes": int(_lib.lib.xmh_scan_pair_cache_bytes(Q, Rn, K, 0)),
           "algorithmic_frac_step": Q * Rn * (2 * W + Lw) / dt / 1e9 / VALU_PEAK_GUIDE, "seed": seed, "p_label": p_label}
    return (out, scan) if return_scan else out


# the extra N=1 legs of bench.py: BASELINE configs[0] (the reference's default 16-bit codes, MIRFlickr-shaped), the same length at the
# COCO shape, configs[3] (DSPH COCO 128 bit) and one GPU's shard of configs[4] (10 M x 256 bit / 8) through the mAP scan
EXTRA_LEGS = {
    "configs0_dcmht_16bit_mirflickr": dict(what="configs[0] DCMHT MIRFlickr-shaped 16-bit", Q=5000, Rn=20015, K=16, C=24, p_label=0.10, seed=1816, steps=30),
    "k16_coco_shape": dict(what="16-bit codes at the configs[1] COCO shape", Q=5000, Rn=117218, K=16, C=80, p_label=0.04, seed=1817, steps=30),
    "configs3_dsph_128bit": dict(what="configs[3] DSPH COCO-shaped 128-bit", Q=5000, Rn=117218, K=128, C=80, p_label=0.04, seed=3814, steps=30),
    "configs4_shard_scan_256bit": dict(what="configs[4] one GPU's shard (10 M / 8) through the mAP scan (12.7 GB pair cache)", Q=5000, Rn=1250000, K=256, C=80,
                                       p_label=0.04, seed=4814, steps=4),
    # SURVEY 8d shape (5): "plus the unsharded 10 M on one GPU" -- a 100 GB pair cache, under the 128 GB cap since round 4 (uncached: 90 ms)
    "configs4_unsharded_scan_256bit": dict(what="configs[4] UNSHARDED: the whole 10 M x 256-bit gallery through the mAP scan on one GPU (100 GB pair "
                                                "cache in the 288 GB of HBM)", Q=5000, Rn=10000000, K=256, C=80, p_label=0.04, seed=4815, steps=2),
}
