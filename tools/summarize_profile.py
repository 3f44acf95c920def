#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into small, committable summaries: <out>/summary_<tag>.md and .json"""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


res = {"tag": tag, "kernel_stats": [], "pmc": {}}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        res["kernel_stats"].append({"name": short(r["Name"]), "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                    "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"]), "min_ns": float(r["MinNs"]),
                                    "max_ns": float(r["MaxNs"])})
# steady-state launch time: rocprofv3's own average runs over EVERY launch of the process, the first (cold clocks, cold caches) ones
# included, and sat 4-8 % above the HIP-event times of the timed steps; the mean over the LAST 20 launches of a kernel is what the
# bench line's avg_launch_ms must agree with
steady = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
        per[short(r["Kernel_Name"])][grid].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for k, grids in per.items():
        # one kernel instance may serve several shapes (k_scan_ap_c<false>: every code length up to 64 bits): the steady figure is of
        # the grid size launched most often, i.e. the headline shape
        grid, v = max(grids.items(), key=lambda kv: len(kv[1]))
        v.sort()
        last = [d for _, d in v[-20:]]
        steady[k] = {"launches": sum(len(x) for x in grids.values()), "steady_avg_ns": sum(last) / len(last), "steady_over": len(last), "steady_grid": grid,
                     "grid_sizes_seen": len(grids)}
for r in res["kernel_stats"]:
    if r["name"] in steady:
        r.update(steady[r["name"]])
res["kernel_stats"].sort(key=lambda r: -r["total_ns"])
res["kernel_stats"] = res["kernel_stats"][:30]
for sub in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_sq3"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        # One kernel instance may serve several SHAPES in one process (k_topk_filter_mfma<8, 4>: the Q = 64 leg, one pass over the gallery, and
        # the Q = 5000 leg, 79 passes): a per-launch average over all of them is nobody's number -- round 4's profile read "1.47 GB per
        # launch, 4.6 x the gallery" out of exactly that mix (profiles/r05_topk_q64_pmc.txt: 320.5 MB at Q = 64 alone).  Rows are therefore
        # averaged per (kernel, grid size); the headline per_launch is the grid launched most often, the others are listed beside it.
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        disp = collections.defaultdict(lambda: collections.defaultdict(set))
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or ""
            agg[k][grid][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][grid].add(r["Dispatch_Id"])
        for k, grids in agg.items():
            e = res["pmc"].setdefault(k, {"per_launch": {}})
            main = max(grids, key=lambda g: len(disp[k][g]))
            n = max(1, len(disp[k][main]))
            e["launches_seen"] = n
            e["grid"] = main
            for c, val in grids[main].items():
                e["per_launch"][c] = val / n
            for g in grids:
                if g != main:
                    og = e.setdefault("other_grids", {}).setdefault(g, {"launches_seen": len(disp[k][g])})
                    for c, val in grids[g].items():
                        og[c] = val / max(1, len(disp[k][g]))
# HBM traffic per launch as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half
# the bytes of a wide (16 B/lane) coalesced streaming read -> doubled for the kernels whose reads are of that kind.
WIDE = ("k_topk_filter", "k_gemm_nt", "k_gemm_g16", "k_topk_stream", "k_im2col_patch", "k_scan_hist_m", "k_scan_ap_c")   # coalesced streaming reads of 8-16 bytes per lane: LDS-DMA of the operand images, the pair cache, the gallery stream (k_topk_filter_mfma's 8-byte loads are halved by FETCH_SIZE too: 160 MB raw against TCC_MISS_sum x 128 B = 327 MB)
for k, e in res["pmc"].items():
    pl = e["per_launch"]
    if "FETCH_SIZE" in pl or "WRITE_SIZE" in pl:
        f = pl.get("FETCH_SIZE", 0.0) * 1024.0
        w = pl.get("WRITE_SIZE", 0.0) * 1024.0
        corr = 2.0 if any(k.startswith(x) for x in WIDE) or (k.startswith("k_scan_ap_s") and k.endswith(", true>")) else 1.0      # cached pass 2 streams the cache with 16-byte loads
        e["hbm_bytes_per_launch"] = {"fetch_raw": f, "write_raw": w, "fetch_correction": corr, "total": f * corr + w}
json.dump(res, open(os.path.join(out, "summary_%s.json" % tag), "w"), indent=1)
with open(os.path.join(out, "summary_%s.md" % tag), "w") as md:
    md.write("# rocprofv3 summary %s\n\ncommand: `python bench.py --steps 20 --warmup 3 --no-cpu-baseline` under `rocprofv3 --kernel-trace --stats` and separate `--pmc` passes\n\n" % tag)
    md.write("## kernel stats (top by total time; steady = mean of the last 20 launches)\n\n| kernel | calls | avg us | steady us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in res["kernel_stats"]:
        md.write("| %s | %d | %.2f | %.2f | %.2f | %.2f | %.2f |\n" % (r["name"], r["calls"], r["avg_ns"] / 1e3, r.get("steady_avg_ns", float("nan")) / 1e3,
                                                                     r["min_ns"] / 1e3, r["max_ns"] / 1e3, r["pct"]))
    md.write("\n## PMC per launch (averaged over the launches seen)\n\n")
    for k, e in sorted(res["pmc"].items()):
        if not any(s in k for s in ("k_scan", "k_topk", "k_gemm", "k_attention", "k_layernorm", "k_pack")):
            continue
        md.write("### %s (%d launches)\n\n" % (k, e["launches_seen"]))
        for c, v in sorted(e["per_launch"].items()):
            md.write("- %s = %.4g\n" % (c, v))
        if "hbm_bytes_per_launch" in e:
            h = e["hbm_bytes_per_launch"]
            md.write("- **HBM bytes/launch** = %.4g (FETCH_SIZE*1024*%g + WRITE_SIZE*1024)\n" % (h["total"], h["fetch_correction"]))
        md.write("\n")
print(open(os.path.join(out, "summary_%s.md" % tag)).read()[:6000])
