#!/usr/bin/env python3
"""Headline benchmark: Hamming query x gallery pairs/s of the fused mAP@all retrieval pass
(BASELINE.json metric, configs[1] = DCMHT COCO-shaped 64-bit: Q 5000 x R 117218, 80 classes).

One "step" = one calc_map_k-equivalent pass over packed codes already resident in HBM:
  pass 1 (bucket histograms) -> [N>1: RCCL all-gather of histograms] -> pass 2 (ranks + AP sums)
  -> [N>1: all-reduce of the per-query sums] -> mean.
N GPUs: every rank holds its own R-row gallery shard (weak scaling, contiguous global index ranges);
queries are replicated after the one-off all-gather of packed query codes.

    python bench.py [--gpus N --steps K --warmup W]        (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel
(the longer of the two passes: k_scan_hist_s with the pair cache, else k_scan_ap_s), a second roofline for the HBM-bound top-k regime (configs[4] shape, one GPU's share) and the
CPU baseline (oracle port of the reference's calc_map_k) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the host driver of this pool only supports dmabuf IPC: RCCL / CUDA-tensor sharing across processes needs this before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_GLOPS = 39321.6         # VALU issue: 256 CU x 4 SIMD x 64 lanes / 4 cycles x 2.4 GHz (tools/ubench_valu.hip: 4.0-4.4 cycles per wave64 op for this mix)


def synth(Q, R, K, C, seed, p=0.04, device="cuda"):
    """SURVEY 8d synthetic inputs: label-correlated +-1 codes, multi-hot labels with >= 1 label per row."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    Wm = torch.randn(C, K, generator=g)

    def side(n):
        L = torch.rand(n, C, generator=g) < p
        L[torch.arange(n), torch.randint(0, C, (n,), generator=g)] = True
        B = (L.float() @ Wm + 0.8 * torch.randn(n, K, generator=g)).sign()
        B[B == 0] = 1
        return B, L.to(torch.int64)
    qB, qL = side(Q)
    rB, rL = side(R)
    return qB, qL, rB, rL


def cpu_baseline(qB, qL, rB, rL, budget_s=20.0):
    """The oracle's step-by-step port of the reference calc_map_k (float GEMM + int64 label matmul + full
    sort + per-query loop) on a bounded query subsample, on this host's cores.  A 32-query probe sizes the
    sample so the whole leg takes about `budget_s` seconds."""
    from oracle import retrieval as orc
    threads = min(32, os.cpu_count() or 1)       # the int64 label matmul stops scaling (and thrashes) beyond that
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    orc.map_k(qB[:32].clone(), rB, qL[:32].clone(), rL, None, stable=True)
    probe = time.perf_counter() - t0
    qsub = int(max(32, min(qB.shape[0], 32 * budget_s / max(probe, 1e-3))))
    t0 = time.perf_counter()
    m = orc.map_k(qB[:qsub].clone(), rB, qL[:qsub].clone(), rL, None, stable=True)
    dt = time.perf_counter() - t0
    return {"value": qsub * rB.shape[0] / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": "first %d of %d queries x full %d-item gallery, oracle.retrieval.map_k (torch CPU, %d threads), %.1f s"
                      % (qsub, qB.shape[0], rB.shape[0], threads, dt), "map": float(m)}


def pmc_traffic(kernel_prefix, kernel_suffix=""):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary under profiles/ (collected by
    tools/profile_round.sh with separate FETCH_SIZE / WRITE_SIZE passes and the gfx950 corrections of
    MI355X_MICROARCH.md); None if no profile has been committed for it."""
    import glob
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed steps run before the --warmup steps: clocks and the Infinity Cache need ~25 steps (15 ms) to reach "
                         "steady state (measured 0.62 ms/step over the first 5 steps against 0.52 in steady state)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--Q", type=int, default=5000)
    ap.add_argument("--R", type=int, default=117218, help="gallery rows PER GPU")
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--C", type=int, default=80)
    ap.add_argument("--p-label", type=float, default=0.04, help="per-class label probability of the synthetic multi-hot labels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-regime", action="store_true")
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="use the multi-GPU exchange path even with one rank (RCCL smoke test)")
    ap.add_argument("--query-blocks", type=int, default=1,
                    help="sharded path: query blocks whose histogram gathers are pipelined (default 1: on one GPU every extra "
                         "block costs 0.18 ms per step, more than the gather it would hide)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    # stdout must carry exactly ONE JSON line: route everything else (RCCL banners printed from C, warnings) to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 through torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product path has no CPU fallback")
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_sharded
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm

    from xmh import retrieval as R
    from xmh import sharded

    Q, Rn, K, C = args.Q, args.R, args.K, args.C
    # every rank synthesises the same queries and its own shard (global rows [rank*R, (rank+1)*R))
    qB, qL, _, _ = synth(Q, 8, K, C, seed=1814, p=args.p_label)
    _, _, rB, rL = synth(8, Rn, K, C, seed=1814 + 1 + rank, p=args.p_label)
    q = R.pack_sign(qB.cuda())
    ql = R.pack_labels(qL.cuda())
    r = R.pack_sign(rB.cuda())
    rl = R.pack_labels(rL.cuda())
    ops = sharded.HipShardOps(q, ql, r, rl, C)
    scan = ops.scan
    nqb = max(1, args.query_blocks)
    piped = sharded.QueryBlocks.split(q, ql, r, rl, C, nqb) if use_dist and nqb > 1 else None

    def step():
        if not use_dist:
            scan.histograms(False)
            return scan.map_all(None)[0]
        return sharded.map_k_sharded(piped if piped is not None else ops, None)[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.settle) + args.warmup):
        m = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    map_value = float(m.item())

    # per-kernel timing: the library brackets each scan kernel with HIP events on the launch stream
    from xmh import _lib
    _lib.prof_enable(True)
    for _ in range(max(args.steps, 10)):
        scan.histograms(False)
        scan.ap_sums(None)
    torch.cuda.synchronize()
    t_hist, n_hist = _lib.prof_read("scan_hist")
    t_hist *= 1e-3
    # pass 2 exists in two device-gated variants (packed 32-bit / 64-bit counters); exactly one of them does the work
    t64, n64 = _lib.prof_read("scan_ap")
    t32, n32 = _lib.prof_read("scan_ap32")
    packed = t32 > t64
    ap_kernel = "k_scan_ap_s, packed 32-bit counters" if packed else "k_scan_ap_s, 64-bit counters"
    # template arguments <W, Lw, TERN, CAPPED, S, P32, MASKED, NW, CACHE>: the profile summary holds both gated launches
    ap_traffic = pmc_traffic("k_scan_ap_s<", ", 4, true, false, 1," if packed else ", 4, false, false, 1,")
    t_ap, n_ap = (t32, n32) if t32 > t64 else (t64, n64)
    t_ap *= 1e-3
    _lib.prof_enable(False)

    W, Lw = (K + 31) // 32, (C + 31) // 32
    alg_bytes = Rn * 4 * (W + Lw) + Q * 4 * (W + Lw) + Q * 12          # gallery once + queries + ap_sum/cap out
    pl = scan.plan
    # the two-pass scheme's own tables that pass 2 reads: below[chunk][bucket][q] (8 B) + dpre[bucket][q] (8 B), written once
    # by the tiny table kernels -- this, not re-reading of inputs, is what the PMC traffic above the algorithmic bytes is
    table_bytes = (pl.nchunk + 1) * pl.nbuckets * pl.qpad * 8 + pl.nchunk * pl.qpad * 4
    # pair cache (xmh_scan_pair_cache_bytes): pass 1 writes a byte per pair, pass 2 reads it instead of evaluating the pair again
    cache_bytes = int(_lib.lib.xmh_scan_pair_cache_bytes(Q, Rn, K, 0))
    cached = cache_bytes > 0
    # VALU instructions per wave-item (ISA count).  Pair evaluation: xor+bcnt per code word, and + and_or per further label word, min.
    ops_eval = 2 * W + Lw + 1
    ops_pair_hist = ops_eval + 2 + (2 if cached else 0)                 # + counter address, add operand (+ cache byte, append)
    # pass 2: (cached: two field extracts instead of the evaluation) + address (+ hi-word mov of the 64-bit variant),
    # credit = 2 cvt + rcp + mul24 + fmac
    ops_pair_ap = (2 if cached else ops_eval) + 1 + (0 if packed else 1) + 5
    hist_kernel = "k_scan_hist_s (pass 1 of the fused mAP scan: pair evaluation + bucket histogram%s)" % (" + pair cache" if cached else "")
    # the roofline object describes the DOMINANT kernel of the step: whichever pass takes longer
    dom_is_hist = t_hist > t_ap
    t_dom, n_dom = (t_hist, n_hist) if dom_is_hist else (t_ap, n_ap)
    ops_dom = ops_pair_hist if dom_is_hist else ops_pair_ap
    dom_traffic = pmc_traffic("k_scan_hist_s<", "") if dom_is_hist else ap_traffic
    roofline = {
        "kernel": "%s, HIP events around the launch, %d launches" % (hist_kernel if dom_is_hist else ap_kernel + " (pass 2 of the fused mAP scan)", n_dom),
        "bound": "hbm", "achieved": alg_bytes / t_dom / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_bytes / t_dom / 1e9 / HBM_PEAK_GBS, "traffic": (dom_traffic or {}).get("bytes"),
        "traffic_detail": dom_traffic,
        "algorithmic_bytes": alg_bytes, "workspace_table_bytes": table_bytes, "pair_cache_bytes": cache_bytes, "avg_launch_ms": t_dom * 1e3,
        "valu": {"lane_ops_per_pair": ops_dom, "achieved": Q * Rn * ops_dom / t_dom / 1e9,
                 "peak": VALU_PEAK_GLOPS, "unit": "G lane-ops/s", "frac": Q * Rn * ops_dom / t_dom / 1e9 / VALU_PEAK_GLOPS},
        "note": "Q=5000 queries share every gallery byte: this launch is VALU-bound (SURVEY H5), HBM fraction is "
                "reported as the contract asks; the HBM-bound regime is in roofline_hbm_regime.  PMC traffic includes the "
                "scheme's own tables and the pair cache (one byte per pair, written by pass 1, read by pass 2)",
        "pass1_avg_launch_ms": t_hist * 1e3, "pass2_avg_launch_ms": t_ap * 1e3,
        "pass1_valu": {"lane_ops_per_pair": ops_pair_hist, "frac": Q * Rn * ops_pair_hist / t_hist / 1e9 / VALU_PEAK_GLOPS},
        "pass2_valu": {"lane_ops_per_pair": ops_pair_ap, "frac": Q * Rn * ops_pair_ap / t_ap / 1e9 / VALU_PEAK_GLOPS},
    }
    if ap_traffic is not None:
        # the cached pass 2 streams the pair cache with 16-byte loads per lane, which FETCH_SIZE counts at half on gfx950
        # (MI355X_MICROARCH.md, HBM section); its 8-byte table reads are counted in full
        corr = cache_bytes / 2 if cached else 0
        roofline["pass2_traffic"] = {"bytes": ap_traffic["bytes"] + corr, "fetch_raw": ap_traffic["fetch_raw"], "write_raw": ap_traffic["write_raw"],
                                     "wide_read_correction_bytes": corr, "source": ap_traffic["source"]}

    out = {
        "metric": "Hamming query x gallery pairs/sec (fused mAP@all pass, DCMHT COCO-shaped 64-bit)",
        "value": Q * Rn * world * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle_steps": max(0, args.settle), "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "configs[1] DCMHT COCO 64-bit retrieval: Q=%d queries x R=%d gallery items per GPU "
                               "(x%d GPUs, contiguous shards), K=%d bits, C=%d classes, mAP@all" % (Q, Rn, world, K, C),
                   "Q": Q, "R_per_gpu": Rn, "K": K, "C": C, "parallelism": "gallery-shard x%d" % world,
                   "query_blocks": nqb if use_dist else 1},
        "mAP": map_value, "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_hbm_regime:
        try:
            import bench_topk
            out["roofline_hbm_regime"] = bench_topk.measure(Q=1)
            out["roofline_hbm_regime_q8"] = bench_topk.measure(Q=8)
        except Exception as exc:                                           # keep the headline line alive
            out["roofline_hbm_regime"] = {"error": repr(exc)}
    if rank == 0 and world == 1:
        # boundary-inclusive rate: the drop-in calc_map_k handed HOST fp32 [N,K] codes and int64 labels like the reference's
        # callers do (H2D over PCIe + pack + both passes + D2H of the scalar); reported next to `value`, never as `value`
        from xmh.common import calc_utils as cu
        cu.calc_map_k(qB, rB, qL, rL)
        t0 = time.perf_counter()
        for _ in range(3):
            m_host = cu.calc_map_k(qB, rB, qL, rL)
        t_host = (time.perf_counter() - t0) / 3
        out["boundary_inclusive"] = {"what": "xmh.common.calc_utils.calc_map_k on host fp32 codes / int64 labels (PCIe H2D + pack + scan + D2H)",
                                     "ms_per_call": t_host * 1e3, "pairs_per_s": Q * Rn / t_host, "mAP": float(m_host)}
    if rank == 0 and world == 1 and not use_dist and not args.no_encode:
        try:
            import bench_encode
            out["encode"] = bench_encode.measure()
        except Exception as exc:
            out["encode"] = {"error": repr(exc)}
    if use_dist and not args.no_encode:
        # encode is data-parallel (every rank encodes its own shard of images / captions, no collective on the path): each rank
        # measures its own rate at the same time as the others, the job rate is the sum
        try:
            import bench_encode
            e = bench_encode.measure(warmup=5, modes=("f32",), extras=False)
            mine = [e["images_per_s_f32"], e["captions_per_s_f32"]]
        except Exception as exc:                                            # every rank still joins the collective below
            mine = [float("nan"), float("nan")]
            print("encode leg failed on rank %d: %r" % (rank, exc), file=sys.stderr)
        t = torch.tensor(mine, dtype=torch.float64, device="cuda")
        lo = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        out["encode"] = {"images_per_s_f32": float(t[0]), "captions_per_s_f32": float(t[1]), "slowest_rank_images_per_s": float(lo[0]),
                         "slowest_rank_captions_per_s": float(lo[1]), "n_gpus": world,
                         "config": {"workload": "CLIP ViT-B/32 + DCMHT 64-bit head, batch 100 per GPU, parity mode; one replica per GPU, "
                                                "rates summed over ranks (measured concurrently)"}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        _, _, rB0, rL0 = rB, rL, rB, rL
        out["cpu_baseline"] = cpu_baseline(qB, qL, rB0, rL0, args.cpu_seconds)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
