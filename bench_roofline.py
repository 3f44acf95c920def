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


def pmc_traffic(kernel_prefix, kernel_suffix=""):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary under profiles/ (collected by
    tools/profile_round.sh with separate FETCH_SIZE / WRITE_SIZE passes and the gfx950 corrections of
    MI355X_MICROARCH.md); None if no profile has been committed for it."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for name, e in d.get("pmc", {}).items():
            if name.startswith(kernel_prefix) and kernel_suffix in name and "hbm_bytes_per_launch" in e:
                best = {"bytes": e["hbm_bytes_per_launch"]["total"], "fetch_raw": e["hbm_bytes_per_launch"]["fetch_raw"],
                        "write_raw": e["hbm_bytes_per_launch"]["write_raw"], "fetch_correction": e["hbm_bytes_per_launch"]["fetch_correction"],
                        "source": os.path.relpath(f, ROOT)}
    return best


def pmc_counter(kernel_prefix, kernel_suffix, counter):
    """per-launch value of a PMC counter of a kernel from the newest committed rocprofv3 summary, with its source file"""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for name, e in d.get("pmc", {}).items():
            if name.startswith(kernel_prefix) and kernel_suffix in name and counter in e.get("per_launch", {}):
                best = (e["per_launch"][counter], os.path.relpath(f, ROOT))
    return best


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


def scan_roofline(scan, Q, Rn, K, C, steps=20):
    """`roofline` object of the headline step: the dominant kernel = the longer of the two passes.  Both passes share every gallery
    byte between Q=5000 queries, so HBM is idle by construction; what they saturate is VALU issue -- `bound` says so,
    `frac` is against the guide's 2-cycle issue peak, `valu.frac_of_measured_mix` against what this instruction mix can reach on
    gfx950 (per-opcode cycles from the committed ubench), and the HBM numbers the contract names stay as side fields."""
    from xmh import _lib
    t_hist, n_hist, t_ap, n_ap, packed = _pass_times(scan, steps)
    W, Lw = (K + 31) // 32, (C + 31) // 32
    alg_bytes = Rn * 4 * (W + Lw) + Q * 4 * (W + Lw) + Q * 12          # gallery once + queries + ap_sum/cap out
    pl = scan.plan
    table_bytes = (pl.nchunk + 1) * pl.nbuckets * pl.qpad * 8 + pl.nchunk * pl.qpad * 4
    cache_bytes = int(_lib.lib.xmh_scan_pair_cache_bytes(Q, Rn, K, 0))
    cached = cache_bytes > 0
    pairs = Q * Rn
    cyc, cyc_src = ubench_cycles()
    tern = scan.qz is not None

    def pass_entry(key, fallback_ops, t, pmc_name):
        mix, src = isa_mix(key)
        static_ops = sum(mix.values()) if mix else fallback_ops
        ops = static_ops
        dyn = pmc_counter(*pmc_name, "SQ_INSTS_VALU")
        e = {}
        if dyn is not None and (Q, Rn, K, C) == (5000, 117218, 64, 80):      # the committed profile is of the default shape
            ops = dyn[0] / (pairs / 64.0)
            e.update({"dynamic_valu_per_64_pairs": ops, "dynamic_source": dyn[1] + " (SQ_INSTS_VALU per launch / (pairs / 64))"})
        e.update({"static_hot_path_valu_per_64_pairs": static_ops})
        e.update({"lane_ops_per_pair": ops, "isa_source": src, "achieved": pairs * ops / t / 1e9, "unit": "G lane-ops/s",
             "peak_guide_2cyc": VALU_PEAK_GUIDE, "frac_of_guide_peak": pairs * ops / t / 1e9 / VALU_PEAK_GUIDE})
        if mix and cyc:
            mean_c, _ = mix_cycles(mix, cyc)
            peak_mix = SIMDS * 64 * CLOCK_HZ / mean_c / 1e9
            e.update({"mean_measured_cycles_per_instruction": mean_c, "ubench_source": cyc_src, "peak_measured_mix": peak_mix,
                      "frac_of_measured_mix": pairs * ops / t / 1e9 / peak_mix, "mix": mix})
        return e
    ops_eval = 2 * W + Lw + 1
    # binary codes of 33..64 bits: pass 1 is k_scan_hist_m (pairs evaluated by the i8 MFMA, xmh_scan.hip mfma_shape)
    mfma_p1 = (not tern) and 32 < K <= 64 and Lw <= 4 and os.environ.get("XMH_SCAN_MFMA", "1") != "0"
    key1 = ("histm_L%d_%s" % (Lw, "cache" if cached else "plain")) if mfma_p1 else "hist_W%d_L%d_%s" % (W, Lw, "cache" if cached else "plain")
    key2 = "ap_W%d_L%d_%s_%s" % (W, Lw, "cache" if cached else "plain", "p32" if packed else "u64")
    ap_suffix = ", true, false, 1," if packed else ", false, false, 1,"
    p1_kernel = "k_scan_hist_m<" if mfma_p1 else "k_scan_hist_s<"
    v1 = pass_entry(key1, (1 + (5 if cached else 0)) if mfma_p1 else ops_eval + 2 + (2 if cached else 0), t_hist, (p1_kernel, ""))
    if mfma_p1:
        nm = 1 + (1 if Lw <= 2 else 2)                            # MFMAs per 16 items x 16 queries: one code tile + the label tiles
        mf = pmc_counter(p1_kernel, "", "SQ_VALU_MFMA_BUSY_CYCLES")
        v1["mfma"] = {"instruction": "v_mfma_i32_16x16x64_i8", "per_64_pairs": nm / 4.0, "cycles_each": 16,
                      "matrix_pipe_frac": pairs / 64.0 * (nm / 4.0) * 16 / (t_hist * CLOCK_HZ * SIMDS),
                      "note": "SQ_INSTS_VALU counts the MFMAs too; the static VALU count does not"}
        if mf:
            v1["mfma"]["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] = mf[0]
    v2 = pass_entry(key2, (2 if cached else ops_eval) + 1 + (0 if packed else 1) + 5, t_ap, ("k_scan_ap_s<", ap_suffix))
    dom_is_hist = t_hist > t_ap
    t_dom, n_dom, vd = (t_hist, n_hist, v1) if dom_is_hist else (t_ap, n_ap, v2)
    traffic = pmc_traffic(p1_kernel, "") if dom_is_hist else pmc_traffic("k_scan_ap_s<", ap_suffix)
    if dom_is_hist and mfma_p1:
        name = "k_scan_hist_m (pass 1 of the fused mAP scan: Hamming distance and label overlap on the i8 MFMA, bucket histogram by LDS atomics%s)" % (" + pair cache" if cached else "")
    elif dom_is_hist:
        name = "k_scan_hist_s (pass 1 of the fused mAP scan: pair evaluation + bucket histogram%s)" % (" + pair cache" if cached else "")
    else:
        name = "k_scan_ap_s, %s counters (pass 2 of the fused mAP scan)" % ("packed 32-bit" if packed else "64-bit")
    return {
        "kernel": "%s, HIP events around the launch, %d launches" % (name, n_dom),
        "bound": "valu", "achieved": vd["achieved"], "peak": VALU_PEAK_GUIDE, "unit": "G lane-ops/s", "frac": vd["frac_of_guide_peak"],
        "frac_of_measured_mix": vd.get("frac_of_measured_mix"),
        "traffic": (traffic or {}).get("bytes"), "traffic_detail": traffic,
        "hbm": {"bound": "hbm", "algorithmic_bytes": alg_bytes, "achieved": alg_bytes / t_dom / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS,
                "note": "reported because the contract names it: Q=%d queries share every gallery byte, the launch cannot be HBM-bound" % Q},
        "workspace_table_bytes": table_bytes, "pair_cache_bytes": cache_bytes, "avg_launch_ms": t_dom * 1e3, "ternary": tern,
        "pass1_avg_launch_ms": t_hist * 1e3, "pass2_avg_launch_ms": t_ap * 1e3, "pass1_valu": v1, "pass2_valu": v2,
        "note": "frac = achieved / the guide's VALU issue peak (2 cycles per wave64 instruction).  frac_of_measured_mix prices the same "
                "instruction stream at the per-opcode cycles measured on this chip (profiles/*_ubench_valu.txt): most opcodes of these "
                "loops (v_bcnt, v_and_or, v_lshl_or, v_min, v_alignbyte, v_bfe, v_cvt, v_mul_u24) issue at half rate, v_rcp at an "
                "eighth.  PMC traffic above the algorithmic bytes is the scheme's own data (bucket tables + the pair cache: one byte "
                "per pair written by pass 1, read by pass 2), not re-reads of the gallery",
    }


def extra_scan_leg(synth, Q, Rn, K, C, p_label, steps=30):
    """one more driver-visible scan shape (configs[3]: DSPH COCO 128-bit): whole-step time and mAP"""
    import time
    import torch
    from xmh import retrieval as R
    qB, qL, rB, rL = synth(Q, Rn, K, C, seed=3814, p=p_label)
    scan = R.RankingScan(R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda()), C)
    for _ in range(10):
        scan.histograms(False)
        m = scan.map_all(None)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        scan.histograms(False)
        m = scan.map_all(None)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    t_hist, _, t_ap, _, packed = _pass_times(scan, 10)
    return {"workload": "configs[3] DSPH COCO-shaped %d-bit: Q=%d x R=%d, C=%d, mAP@all" % (K, Q, Rn, C), "ms_per_step": dt * 1e3,
            "pairs_per_s": Q * Rn / dt, "mAP": float(m.item()), "pass1_ms": t_hist * 1e3, "pass2_ms": t_ap * 1e3,
            "pass2_counters": "packed 32-bit" if packed else "64-bit"}
