"""The ONE line bench.py prints on stdout, kept short.

Round 4's line was 22 KB and the driver could not parse it out of its stdout tail (BENCH_r04.json: parsed null).  The contract
line is now assembled here from the full result object: contract keys, a trimmed `roofline`, a trimmed `cpu_baseline` and a
numbers-only `summary`; everything else goes to gpurun_out/bench_detail.json (and stderr).  tests/test_bench_line.py pushes a
fake full-size object through compact_line() and asserts strict JSON under MAX_LINE_BYTES.  No torch import here.
"""
import json
import math
import os

MAX_LINE_BYTES = 4096

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")
_ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_step", "traffic", "avg_launch_ms", "kernels_per_step",
                  "algorithmic_bytes_per_step", "traffic_per_step")
_CPU_KEYS = ("value", "unit", "cores", "physical_cores", "kind", "sample", "abs_delta_gpu_vs_stable", "abs_delta_gpu_vs_default")


def _num(x, sig=6):
    """floats rounded to `sig` significant digits; non-finite floats become None so the line is STRICT JSON"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {str(k): _num(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, sig) for v in x]
    return str(x)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def _summary(out):
    """numbers only, short keys: the other legs of the run"""
    s = {}
    for name, key in (("topk_q1", "roofline_hbm_regime"), ("topk_q8", "roofline_hbm_regime_q8"), ("topk_q64", "roofline_hbm_regime_q64"),
                      ("topk_q1_64bit", "roofline_hbm_regime_64bit"), ("topk_q1_32bit", "roofline_hbm_regime_32bit")):
        if key in out:
            s[name] = _pick(out[key], ("frac", "achieved", "whole_call_GBps", "whole_call_ms", "error"))
            if isinstance(out[key].get("mfma"), dict):              # several queries per pass: the i8 matrix-core rate is the bound (useful / issued, of 3944 TOPS)
                s[name]["mfma_frac"] = out[key]["mfma"]["frac"]
                s[name]["mfma_frac_issued"] = out[key]["mfma"]["frac_issued"]
    if isinstance(out.get("topk_ternary"), dict):
        s["topk_q1_ternary"] = _pick(out["topk_ternary"], ("whole_call_ms", "whole_call_GBps", "filter_GBps", "error"))
    if isinstance(out.get("float_route"), dict):
        s["float_map_q500"] = _pick(out["float_route"], ("ms_per_call", "pairs_per_s", "error"))
    cd = out.get("topk_infinity_cache_defeated", {})
    for qn in ("Q1", "Q8", "Q64"):
        legs = cd.get(qn) if isinstance(cd, dict) else None
        rot = [v for k, v in legs.items() if k.startswith("rotating")] if isinstance(legs, dict) else []
        if rot:
            s["topk_%s_cold" % qn.lower()] = _pick(rot[0], ("filter_frac_of_8TBps", "filter_GBps", "whole_call_GBps", "whole_call_frac_of_8TBps"))
    if isinstance(out.get("valid_e2e"), dict):
        s["valid_e2e"] = _pick(out["valid_e2e"], ("valid_seconds", "encode_seconds", "retrieve_seconds", "error"))
    e = out.get("encode")
    if isinstance(e, dict):
        s["encode"] = _pick(e, ("images_per_s_f32", "captions_per_s_f32", "images_per_s_f16", "images_gemm_tflops_f32", "images_gemm_tflops_f16",
                                "gemm_frac_of_fp16_peak_f32", "gemm_frac_of_fp16_peak_f16", "error"))
        if isinstance(e.get("fused_batches"), dict):
            s["encode_b400"] = _pick(e["fused_batches"], ("images_per_s_f32", "images_per_s_f16", "images_gemm_tflops_f32", "images_gemm_tflops_f16",
                                                          "gemm_frac_of_fp16_peak_f32", "gemm_frac_of_fp16_peak_f16"))
    if isinstance(out.get("encode_mith_b400"), dict):
        s["encode_mith_b400"] = _pick(out["encode_mith_b400"], ("images_per_s", "captions_per_s", "error"))
    if isinstance(out.get("encode_mith"), dict):
        s["encode_mith"] = _pick(out["encode_mith"], ("images_per_s", "captions_per_s", "images_per_s_f32", "captions_per_s_f32", "error"))
    if isinstance(out.get("cpu_baseline_encode"), dict):
        s["cpu_encode"] = _pick(out["cpu_baseline_encode"], ("images_per_s", "captions_per_s", "cores", "error"))
    for short, key in (("cfg0_k16", "configs0_dcmht_16bit_mirflickr"), ("k16_coco", "k16_coco_shape"), ("cfg3_k128", "configs3_dsph_128bit"),
                       ("cfg4_shard_k256", "configs4_shard_scan_256bit"), ("cfg4_10M_k256", "configs4_unsharded_scan_256bit"),
                       ("topk_q5000_10M", "topk_q5000_10M_256bit")):
        if isinstance(out.get(key), dict):
            s[short] = _pick(out[key], ("ms_per_step", "whole_call_ms", "pairs_per_s", "pairs_per_s_whole_call", "traffic_per_step", "error"))
    if isinstance(out.get("materialised_outputs"), dict):
        s["materialised_outputs"] = _pick(out["materialised_outputs"], ("hamming_dist_GBps", "label_sim_GBps", "torch_fill_GBps"))
    if isinstance(out.get("boundary_inclusive"), dict):
        s["boundary_inclusive"] = _pick(out["boundary_inclusive"], ("ms_per_call", "pairs_per_s"))
    if isinstance(out.get("strong_scaling"), dict):
        ss = out["strong_scaling"]
        s["strong"] = {"error": ss["error"]} if "error" in ss else {
            "cfg2_map": _pick(ss.get("configs2_nuswide_map", {}), ("pairs_per_s", "ms_per_step", "mAP")),
            "cfg4_topk": {q: _pick(v, ("pairs_per_s", "ms_per_call", "gallery_GBps"))
                          for q, v in ss.get("configs4_topk_10M_256bit", {}).get("legs", {}).items()}}
    return s


def compact(out):
    """the contract object: every contract key, mAP, trimmed roofline / cpu_baseline, summary"""
    line = {k: out[k] for k in CONTRACT_KEYS if k in out}
    if isinstance(line.get("config"), dict):
        cfg = dict(line["config"])
        cfg.pop("collectives_in_step", None) if len(json.dumps(cfg)) > 600 else None
        line["config"] = cfg
    for k in ("mAP", "settle_steps", "rccl_ranks", "gpu_over_cpu", "dry_run", "ranks_in_group", "backend", "exchange_ms", "build"):
        if k in out:
            line[k] = out[k]
    if isinstance(out.get("roofline"), dict):
        r = _pick(out["roofline"], _ROOFLINE_KEYS)
        alg = out["roofline"].get("algorithmic")
        if "frac_step" not in r and isinstance(alg, dict) and "frac_step" in alg:
            r["frac_step"] = alg["frac_step"]
        hbm = out["roofline"].get("hbm")
        if isinstance(hbm, dict) and "frac" in hbm:                     # the same launch against the HBM roof (Q queries share every byte)
            r["hbm_frac"] = hbm["frac"]
        r["kernel"] = _short(r.get("kernel"), 96)
        line["roofline"] = r
    if isinstance(out.get("cpu_baseline"), dict):
        c = _pick(out["cpu_baseline"], _CPU_KEYS)
        c["sample"] = _short(c.get("sample"), 160)
        line["cpu_baseline"] = c
    line["summary"] = _summary(out)
    line["detail"] = "gpurun_out/bench_detail.json"
    return _num(line)


def compact_line(out):
    """one strict-JSON line under MAX_LINE_BYTES; drops summary entries from the back rather than exceed it"""
    obj = compact(out)
    text = json.dumps(obj, separators=(",", ":"), allow_nan=False)
    while len(text.encode()) > MAX_LINE_BYTES and obj.get("summary"):
        obj["summary"].pop(next(reversed(obj["summary"])))
        text = json.dumps(obj, separators=(",", ":"), allow_nan=False)
    if len(text.encode()) > MAX_LINE_BYTES:
        raise ValueError("bench line is %d bytes without its summary" % len(text.encode()))
    return text


def write_detail(out, root):
    """the full object, for people: gpurun_out/bench_detail.json (merged back by gpurun) -- best effort"""
    try:
        d = os.path.join(root, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_detail.json")
        with open(path, "w") as f:
            json.dump(_num(out, 9), f, indent=1)
        return path
    except OSError:
        return None
