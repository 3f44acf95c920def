#!/usr/bin/env python3
"""Headline benchmark: Hamming query x gallery pairs/s of the fused mAP@all retrieval pass
(BASELINE.json metric, configs[1] = DCMHT COCO-shaped 64-bit: Q 5000 x R 117218, 80 classes).

One "step" = one calc_map_k-equivalent pass over packed codes already resident in HBM:
  [N>1: RCCL all-gather of the packed query codes / labels each rank "encoded"] -> pass 1 (bucket histograms)
  -> [N>1: RCCL all-gather of histograms] -> pass 2 (ranks + AP sums) -> [N>1: all-reduce of the per-query sums] -> mean.
N GPUs (default, "weak"): every rank holds its own R-row gallery shard (contiguous global index ranges), the query set is
split over the ranks and all-gathered inside the timed step, like runners/base.py does after encoding.
After the headline the N>1 run also measures the fixed-gallery ("strong") shapes of BASELINE configs[2] (NUS-WIDE-shaped
mAP scan, Q 5000 x R 188 000 in total) and configs[4] (10 M x 256-bit top-k, shard top-k -> gather -> host merge) and
reports them under "strong_scaling".

    python bench.py [--gpus N --steps K --warmup W]

N>1 may be launched either way: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE
in the environment), or plain `python bench.py --gpus N`, which re-executes itself under torch.distributed.run
(rendezvous on 127.0.0.1, a free port) -- the reference's own entry script spawns its ranks the same way (main.py:38-51).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel of the step, a second
roofline for the HBM-bound top-k regime (configs[4] shape, one GPU's share), the encoder leg and the CPU baselines (oracle
ports of the reference's calc_map_k and CLIP forward) timed on this host.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the host driver of this pool only supports dmabuf IPC: RCCL / CUDA-tensor sharing across processes needs this before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench_emit  # noqa: E402
import bench_roofline as RL  # noqa: E402


def synth(Q, R, K, C, seed, p=0.04):
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


def build_info():
    """which libxmh.so this process loaded and whether it is a build of the sources beside it: the library carries the sha256 of
    csrc/*.hip, csrc/*.h and include/xmh.h it was compiled from (xmh_build_id, the Makefile's SRCID), recomputed here from the files
    (the .so travels prebuilt in the snapshot; tests/test_boundary_cpu.py rebuilds it from scratch on the CPU side)"""
    import glob
    import hashlib
    from xmh import _lib
    pkg = os.path.join(ROOT, "clip-based-cross-modal-hash_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip"))) + sorted(glob.glob(os.path.join(pkg, "csrc", "*.h")) + [os.path.join(pkg, "..", "include", "xmh.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    so = _lib.LIB_PATH
    bid = _lib.lib.xmh_build_id().decode()
    return {"lib": os.path.relpath(so, ROOT), "lib_bytes": os.path.getsize(so), "build_id": bid, "lib_is_a_build_of_these_sources": bid == h.hexdigest()[:16],
            "sources": len(files), "xmh_version": int(_lib.lib.xmh_version())}


def host_description():
    """CPU model string, logical CPUs and physical cores of this host ('core count and CPU model printed')."""
    model, phys, sockets = "unknown", None, set()
    try:
        cores = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
                sockets.add(pid)
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    cores.add((pid, cid))
                pid = cid = None
        phys = len(cores) or None
    except OSError:
        pass
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "physical_cores": phys, "sockets": len(sockets) or None}


def cpu_baseline(qB, qL, rB, rL, budget_s=20.0, gpu_map_fn=None):
    """The oracle's step-by-step port of the reference calc_map_k (float GEMM + int64 label matmul + full sort + per-query
    loop) on a bounded query subsample, on this host's cores.  A 32-query probe sizes the sample so the timed leg takes
    about `budget_s` seconds.  The same sample is then ranked once more with the reference's DEFAULT (unstable) sort, and by
    the GPU path, so the line carries the real tie-order delta at this scale ('5e-4' was a guess)."""
    from oracle import retrieval as orc
    host = host_description()
    # SURVEY 8d asks for all physical cores; the int64 label matmul of the port does not scale that far on every host, so both are
    # probed (64 queries each) and the faster thread count runs the timed leg; the sweep is reported in the line
    cands = sorted({min(32, os.cpu_count() or 1), min(host["physical_cores"] or 32, os.cpu_count() or 1)})
    sweep = {}
    for th in cands:
        torch.set_num_threads(th)
        orc.map_k(qB[:8].clone(), rB, qL[:8].clone(), rL, None, stable=True)          # thread pool start-up
        t0 = time.perf_counter()
        orc.map_k(qB[:64].clone(), rB, qL[:64].clone(), rL, None, stable=True)
        sweep[th] = 64 * rB.shape[0] / (time.perf_counter() - t0)
    threads = max(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    probe = 32 * rB.shape[0] / sweep[threads]
    qsub = int(max(32, min(qB.shape[0], 32 * budget_s / max(probe, 1e-3))))
    t0 = time.perf_counter()
    m = orc.map_k(qB[:qsub].clone(), rB, qL[:qsub].clone(), rL, None, stable=True)
    dt = time.perf_counter() - t0
    m_default = orc.map_k(qB[:qsub].clone(), rB, qL[:qsub].clone(), rL, None, stable=False)
    out = {"value": qsub * rB.shape[0] / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
           "thread_sweep_pairs_per_s": {str(k): v for k, v in sweep.items()},
           "sample": "first %d of %d queries x full %d-item gallery, oracle.retrieval.map_k (torch CPU, %d threads), %.1f s"
                     % (qsub, qB.shape[0], rB.shape[0], threads, dt),
           "cpu_model": host["cpu_model"], "physical_cores": host["physical_cores"], "logical_cpus": host["logical_cpus"],
           "map_stable": float(m), "map_default": float(m_default), "abs_delta_default_vs_stable": abs(float(m) - float(m_default))}
    if gpu_map_fn is not None:
        g = float(gpu_map_fn(qsub))
        out["map_gpu_same_sample"] = g
        out["abs_delta_gpu_vs_stable"] = abs(g - float(m))
        out["abs_delta_gpu_vs_default"] = abs(g - float(m_default))
    return out


def cpu_encode_baseline(batch=100):
    """the fp32 PyTorch encode on this host's cores -- the oracle's restatement of the reference CLIP
    ViT-B/32 forward (oracle/encode.py) on one batch of 100 images and one of 100 captions, synthetic weights."""
    from oracle import encode as enc
    from xmh.models import weights as W
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    sd = W.synth_clip_state_dict(1814) if hasattr(W, "synth_clip_state_dict") else None
    if sd is None:
        return {"error": "no synthetic state-dict generator"}
    image = W.synth_images(5, batch)
    ids, _ = W.synth_text(5, batch)
    res = {}
    with torch.no_grad():
        for what, fn in (("images", lambda: enc.clip_image(sd, image)), ("captions", lambda: enc.clip_text(sd, ids))):
            fn()                                                        # first call pays allocator / thread-pool start-up
            t0 = time.perf_counter()
            fn()
            res[what + "_per_s"] = batch / (time.perf_counter() - t0)
    res.update({"cores": threads, "kind": "port", "cpu_model": host_description()["cpu_model"],
                "sample": "one batch of %d images and one of %d captions through oracle.encode.clip_image / clip_text (torch CPU fp32, %d threads)"
                          % (batch, batch, threads)})
    return res


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args_list, n):
    """plain `python bench.py --gpus N`: re-execute under torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + args_list
    env = dict(os.environ)
    env["XMH_BENCH_CHILD"] = "1"
    return subprocess.call(cmd, env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--settle", type=int, default=30,
                    help="untimed steps run before the --warmup steps: clocks and the Infinity Cache need ~25 steps (15 ms) to reach "
                         "steady state (measured 0.62 ms/step over the first 5 steps against 0.52 in steady state)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--Q", type=int, default=5000)
    ap.add_argument("--R", type=int, default=117218, help="gallery rows PER GPU (weak scaling)")
    ap.add_argument("--K", type=int, default=64)
    ap.add_argument("--C", type=int, default=80)
    ap.add_argument("--p-label", type=float, default=0.04, help="per-class label probability of the synthetic multi-hot labels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-regime", action="store_true")
    ap.add_argument("--no-encode", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="N>1: skip the fixed-gallery legs (configs[2], configs[4])")
    ap.add_argument("--no-extra-configs", action="store_true", help="N=1: skip the configs[3] (K=128) and MITH encode legs")
    ap.add_argument("--force-sharded", action="store_true", help="use the multi-GPU exchange path even with one rank (RCCL smoke test)")
    ap.add_argument("--force-strong", action="store_true", help="with --force-sharded: also run the N>1 fixed-gallery legs on the one rank")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST HOOK, never a measurement: every rank uses cuda:0 and the group is gloo (RCCL refuses two ranks on one device), "
                         "so the N>1 code path -- shard ops on device tensors, the collectives on workspace views, the fixed-gallery legs -- "
                         "runs with world > 1 on a one-GPU box; the line says so in `launcher`")
    ap.add_argument("--query-blocks", type=int, default=1,
                    help="sharded path: query blocks whose histogram gathers are pipelined (default 1: on one GPU every extra "
                         "block costs 0.18 ms per step, more than the gather it would hide)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher + rendezvous + the step's collectives on zero tensors over gloo, no GPU work (CPU test of the N>1 path)")
    return ap.parse_args(argv)


def dry_run(args, world, rank):
    """the N>1 choreography of one step on CPU tensors over gloo: query all-gather, histogram all-gather, all-reduce"""
    from xmh import sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ranks = dist.get_world_size() if world > 1 else 1
    assert ranks == args.gpus, (ranks, args.gpus)
    Q, nb, W = 64, 65, 2
    qb = sharded.shard_bounds(Q, world)
    mine = torch.full((qb[rank + 1] - qb[rank], W), rank + 1, dtype=torch.int32)
    t0 = time.perf_counter()
    if world > 1:
        full = sharded.all_gather_rows(mine, [qb[r + 1] - qb[r] for r in range(world)])
        g = sharded._gather_hist_pair(torch.zeros(Q, nb, dtype=torch.int32), torch.ones(Q, nb, dtype=torch.int32))
        ap = torch.ones(Q, dtype=torch.float64)
        dist.all_reduce(ap)
        assert full.shape == (Q, W) and g.shape == (world, 2, Q, nb) and float(ap[0]) == world
        assert all(int(full[qb[r]][0]) == r + 1 for r in range(world))
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return {"dry_run": True, "n_gpus": args.gpus, "ranks_in_group": ranks, "backend": "gloo" if world > 1 else None, "exchange_ms": dt * 1e3}


class ShardedQueries:
    """This rank's slice of the packed query CODES plus the persistent full-size buffer the scan objects point at: gather() is
    the all-gather runners/base.py performs after encoding (packed codes <= 40 KB in all).  The query LABELS are not gathered:
    every rank holds the whole label matrices, as in the reference (runners/base.py keeps query_labels / retrieval_labels complete
    on each rank and xmh/runners/base.py packs them locally)."""

    def __init__(self, q, ql, world, rank):
        from xmh import sharded
        self.sharded = sharded
        b = sharded.shard_bounds(q.n, world)
        self.counts = [b[r + 1] - b[r] for r in range(world)]
        self.q_loc = q.bits[b[rank]:b[rank + 1]].clone()
        self.q_full, self.ql_full = q, ql                     # the gathered rows are written into these (same values)

    def gather(self):
        if len(set(self.counts)) == 1 and hasattr(dist, "all_gather_into_tensor") and self.q_loc.is_cuda:
            # equal slices (Q % world == 0): the two collectives write the full buffers in place, no pad / cat / copy kernels
            # (the ragged form below costs ~10 small launches = 85 us of a 0.56 ms step)
            dist.all_gather_into_tensor(self.q_full.bits, self.q_loc)
            return
        self.q_full.bits.copy_(self.sharded.all_gather_rows(self.q_loc, self.counts))


def timed(step, barrier, steps, use_dist):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        m = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, m


def strong_legs(args, world, rank, barrier):
    """Fixed-gallery shapes sharded with shard_bounds.  configs[2]: NUS-WIDE-shaped mAP scan, Q 5000 x
    R 188 000 IN TOTAL, C 21, K 64; configs[4]: exact top-100 over 10 M x 256 bit IN TOTAL for Q in {1, 8, 64}: shard
    top-k -> all-gather of the lists -> host merge (north_star)."""
    from xmh import retrieval as R
    from xmh import sharded
    out = {}
    # ---- configs[2] ----
    Q, Rt, K, C = 5000, 188000, 64, 21
    b = sharded.shard_bounds(Rt, world)
    qB, qL, _, _ = synth(Q, 8, K, C, seed=2814, p=0.1)
    n_loc = b[rank + 1] - b[rank]
    _, _, rB, rL = synth(8, n_loc, K, C, seed=2815 + rank, p=0.1)
    q, ql = R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda())
    r, rl = R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda())
    sq = ShardedQueries(q, ql, world, rank)
    ops = sharded.HipShardOps(q, ql, r, rl, C)

    def step():
        sq.gather()
        return sharded.map_k_sharded(ops, None, map_only=True)[0]
    for _ in range(10):
        step()
    steps = max(20, args.steps // 4)
    dt, m = timed(step, barrier, steps, True)
    out["configs2_nuswide_map"] = {"workload": "MITH NUS-WIDE-shaped 64-bit mAP scan: Q=%d x R=%d IN TOTAL over %d contiguous shards, C=%d" % (Q, Rt, world, C),
                                   "scaling": "strong", "pairs_per_s": Q * Rt * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
                                   "mAP": float(m.item()), "rows_this_rank": n_loc}
    del ops, sq, q, ql, r, rl
    # ---- configs[4] ----
    Rt, K, k = 10_000_000, 256, 100
    W = K // 32
    b = sharded.shard_bounds(Rt, world)
    n_loc = b[rank + 1] - b[rank]
    g = torch.Generator(device="cuda").manual_seed(4814 + rank)
    rb = torch.randint(-2**31, 2**31 - 1, (n_loc, W), dtype=torch.int32, device="cuda", generator=g)
    g.manual_seed(4813)
    qall = torch.randint(-2**31, 2**31 - 1, (64, W), dtype=torch.int32, device="cuda", generator=g)
    r = R.PackedCodes(rb, None, K)
    legs = {}
    for Qn in (1, 8, 64):
        qq = R.PackedCodes(qall[:Qn].contiguous(), None, K)

        def tstep():
            return sharded.topk_sharded(qq, r, k, b[rank])
        for _ in range(3):
            tstep()
        steps = 20
        dt, res = timed(tstep, barrier, steps, True)
        legs["Q%d" % Qn] = {"pairs_per_s": Qn * Rt * steps / dt, "ms_per_call": dt / steps * 1e3,
                            "gallery_GBps": Rt * W * 4 * steps / dt / 1e9, "first_hit": [int(res[0][0, 0]), int(res[1][0, 0])],
                            # digest of the merged (distance, index) lists: a test re-derives it from ONE top-k over all shards
                            "lists_crc32": zlib.crc32(res[0].numpy().tobytes() + res[1].numpy().tobytes())}
    out["configs4_topk_10M_256bit"] = {"workload": "exact top-%d over R=%d x %d-bit IN TOTAL (%d shards of ~%d rows): shard top-k, "
                                                   "all-gather of [Q,k] lists, host merge" % (k, Rt, K, world, n_loc),
                                       "scaling": "strong", "legs": legs}
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        if os.environ.get("XMH_BENCH_CHILD"):
            raise SystemExit("bench.py: launched as a child without RANK/WORLD_SIZE")
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # stdout must carry exactly ONE JSON line: route everything else (RCCL banners printed from C, warnings) to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        """stdout gets ONE compact contract line (bench_emit: <= 4 KB, strict JSON); the full object goes to
        gpurun_out/bench_detail.json and stderr"""
        sys.stdout.flush()
        if rank == 0:
            path = bench_emit.write_detail(obj, ROOT)
            print("bench detail (%s): %s" % (path, json.dumps(bench_emit._num(obj, 9))), file=sys.stderr)
            sys.stderr.flush()
            os.write(json_fd, (bench_emit.compact_line(obj) + "\n").encode())
        os.close(json_fd)

    if args.dry_run:
        emit(dry_run(args, world, rank))
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the product path has no CPU fallback")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_sharded
    ranks_in_group = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        if args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local),   # "nccl" is RCCL on ROCm
                                    timeout=datetime.timedelta(minutes=5))      # every collective here is sub-second: fail fast, do not hang
        ranks_in_group = dist.get_world_size()
        if ranks_in_group != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (ranks_in_group, args.gpus))

    from xmh import retrieval as R
    from xmh import sharded

    Q, Rn, K, C = args.Q, args.R, args.K, args.C
    # every rank synthesises the same queries (it keeps its own slice) and its own shard (global rows [rank*R, (rank+1)*R))
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
    sq = ShardedQueries(q, ql, world, rank) if use_dist else None

    def step():
        if not use_dist:
            scan.histograms(False)
            return scan.map_all(None)[0]
        sq.gather()                                            # packed query codes + labels over RCCL, inside the timed step
        return sharded.map_k_sharded(piped if piped is not None else ops, None, map_only=True)[0]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.settle) + args.warmup):
        m = step()
    dt, m = timed(step, barrier, args.steps, use_dist)
    map_value = float(m.item())

    # what the timed step really exchanged: the one-shot form runs map_k_sharded(map_only=True), whose default is the
    # all-to-all by query slice whenever the padded query count divides by the world size; query blocks pipeline the [Q] form
    if not use_dist:
        collectives = []
    elif piped is not None:
        collectives = ["all_gather packed query codes", "%d x async all_gather [2,Q/%d,K+1] histograms" % (nqb, nqb), "all_reduce [Q] f64"]
    elif scan.plan.qpad % world == 0:
        collectives = ["all_gather packed query codes", "all_to_all totals-table query slices [world,K+1,qpad/world,2] u32",
                       "all_to_all offset rows [world,K+2,qpad/world,2] u32", "all_reduce [1] f64"]
    else:
        collectives = ["all_gather packed query codes", "all_gather totals tables [K+1,qpad,2] u32", "all_reduce [1] f64"]
    roofline = RL.scan_roofline(scan, Q, Rn, K, C, steps=max(args.steps, 10), step_s=None if use_dist else dt / args.steps)

    out = {
        "metric": "Hamming query x gallery pairs/sec (fused mAP@all pass, DCMHT COCO-shaped 64-bit)",
        "value": Q * Rn * world * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "settle_steps": max(0, args.settle), "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "configs[1] DCMHT COCO 64-bit retrieval: Q=%d queries x R=%d gallery items per GPU "
                               "(x%d GPUs, contiguous shards), K=%d bits, C=%d classes, mAP@all" % (Q, Rn, world, K, C),
                   "Q": Q, "R_per_gpu": Rn, "K": K, "C": C, "parallelism": "gallery-shard x%d" % world,
                   "query_blocks": nqb if use_dist else 1,
                   "collectives_in_step": collectives},
        "rccl_ranks": ranks_in_group, "launcher": ("torch.distributed.run" if world > 1 else "single process") + (" (--share-gpu: all ranks on cuda:0 over gloo, a code-path check and not a measurement)" if args.share_gpu else ""),
        "mAP": map_value, "roofline": roofline, "build": build_info(),
    }

    if rank == 0 and world == 1 and not args.no_hbm_regime:
        import bench_topk
        # one try per leg: a failure (e.g. no room for 80 M rows) must not overwrite a leg already measured
        for key, kw in (("roofline_hbm_regime", dict(Q=1)), ("roofline_hbm_regime_q8", dict(Q=8)), ("roofline_hbm_regime_q64", dict(Q=64)),
                        # the same 320 MB as the reference's own code lengths: 40 M x 64 bit and 80 M x 32 bit (k_topk_filter_short), robust path timing left out
                        ("roofline_hbm_regime_64bit", dict(R=40_000_000, K=64, Q=1, robust=False)),
                        ("roofline_hbm_regime_32bit", dict(R=80_000_000, K=32, Q=1, robust=False))):
            try:
                out[key] = bench_topk.measure(**kw)
            except Exception as exc:                                       # keep the headline line alive
                out[key] = {"error": repr(exc)}
            torch.cuda.empty_cache()
        try:
            out["topk_structured_codes"] = bench_topk.measure_structured()
        except Exception as exc:
            out["topk_structured_codes"] = {"error": repr(exc)}
        try:
            out["topk_ternary"] = bench_topk.measure_ternary()
        except Exception as exc:
            out["topk_ternary"] = {"error": repr(exc)}
        torch.cuda.empty_cache()
        for key, fn in (("topk_q5000_10M_256bit", bench_topk.measure_many_queries), ("topk_infinity_cache_defeated", bench_topk.measure_cache_defeat)):
            try:
                out[key] = fn()
            except Exception as exc:
                out[key] = {"error": repr(exc)}
            torch.cuda.empty_cache()
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
    if rank == 0 and world == 1 and not use_dist and not args.no_extra_configs:
        # round 6: calc_map_k on un-quantised float "codes" (UMoED-style tanh outputs, reference runners/UMoED/runner.py:162-186) -- the
        # reference's own route, fp32 GEMM + one stable sort per query (xmh_gemm_f32_sort_map), Q 500 x R at the COCO shape
        try:
            from xmh.common import calc_utils as cu2
            import xmh.dense as dense2
            dense2._warned_float = True
            gq = torch.Generator().manual_seed(5)
            fq, fr = torch.tanh(torch.randn(500, K, generator=gq)).cuda(), torch.tanh(torch.randn(Rn, K, generator=gq)).cuda()
            fql, frl = qL[:500].cuda(), rL.cuda()
            cu2.calc_map_k(fq, fr, fql, frl)
            torch.cuda.synchronize()
            tfs = []
            for _ in range(7):                                  # median: one call in a few dozen pays a hipMalloc of the 1.2 GB sort workspace
                t0 = time.perf_counter()
                mf = cu2.calc_map_k(fq, fr, fql, frl)
                tfs.append(time.perf_counter() - t0)
            tf = sorted(tfs)[len(tfs) // 2]
            out["float_route"] = {"workload": "calc_map_k on tanh float codes, Q=500 x R=%d x %d: fp32 GEMM + segmented radix sort per query + AP pass" % (Rn, K),
                                  "ms_per_call": tf * 1e3, "ms_per_call_max": max(tfs) * 1e3, "timing": "median of 7 calls",
                                  "pairs_per_s": 500 * Rn / tf, "mAP": float(mf)}
            del fq, fr
            cu2.release_scan_workspace()
        except Exception as exc:
            out["float_route"] = {"error": repr(exc)}
        torch.cuda.empty_cache()
        # SURVEY 8 rows a-1 / a-5 as matrices (calc_hammingDist / calc_label_sim, 2000 x R floats): write-bound, against torch's fill_
        try:
            qs, qls = R.PackedCodes(q.bits[:2000].contiguous(), None, K), ql[:2000].contiguous()
            rs = R.PackedCodes(r.bits, None, K)

            def _t(fn):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 5 * 1e-3
            nbytes = 2000 * Rn * 4
            scratch = torch.empty(2000, Rn, dtype=torch.float32, device="cuda")
            out["materialised_outputs"] = {"workload": "calc_hammingDist / calc_label_sim as float32 [2000, %d] matrices (%d-bit codes, %d classes)" % (Rn, K, C),
                                           "hamming_dist_GBps": nbytes / _t(lambda: R.hamming_dist(qs, rs)) / 1e9,
                                           "label_sim_GBps": nbytes / _t(lambda: R.label_sim(qls, rl, C)) / 1e9,
                                           "torch_fill_GBps": nbytes / _t(lambda: scratch.fill_(1.0)) / 1e9}
            del scratch
        except Exception as exc:
            out["materialised_outputs"] = {"error": repr(exc)}
        torch.cuda.empty_cache()
        for name, leg in RL.EXTRA_LEGS.items():
            try:
                out[name] = RL.extra_scan_leg(**leg)
            except Exception as exc:
                out[name] = {"error": repr(exc)}
            torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not use_dist and not args.no_encode:
        try:
            import bench_encode
            out["encode"] = bench_encode.measure()
            if not args.no_extra_configs:
                out["encode_mith"] = bench_encode.measure_mith()
                out["encode_mith_b400"] = bench_encode.measure_mith(batch=400, steps=5)      # the runner's fused evaluation batches (encode_fuse = 4)
        except Exception as exc:
            out["encode"] = {"error": repr(exc)}
    if rank == 0 and world == 1 and not use_dist and not args.no_encode and not args.no_extra_configs:
        # the thing the reference runs: BaseTrainer.valid() at the configs[1] shape, end to end, with the encode / retrieve split
        try:
            import bench_valid
            del ops, scan
            torch.cuda.empty_cache()
            out["valid_e2e"] = bench_valid.measure(Q=Q, Rn=Rn, K=K, C=C)
        except Exception as exc:
            out["valid_e2e"] = {"error": repr(exc)}
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
    if (world > 1 or (args.force_sharded and args.force_strong)) and not args.no_strong:
        del ops, scan, piped, sq
        torch.cuda.empty_cache()
        try:
            out["strong_scaling"] = strong_legs(args, world, rank, barrier)
        except Exception as exc:
            # a failure on ONE rank leaves the others inside a collective until the process-group timeout (5 minutes, set above);
            # nothing after this point needs the group again, so the failing rank just records it and every rank falls through to
            # destroy_process_group
            out["strong_scaling"] = {"error": repr(exc)}
            print("strong-scaling legs failed on rank %d: %r" % (rank, exc), file=sys.stderr)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def gpu_map(qsub):
            return R.map_k_packed(R.pack_sign(qB[:qsub].cuda()), r, R.pack_labels(qL[:qsub].cuda()), rl, C, None).item()
        out["cpu_baseline"] = cpu_baseline(qB, qL, rB, rL, args.cpu_seconds, gpu_map)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        if not args.no_encode:
            try:
                out["cpu_baseline_encode"] = cpu_encode_baseline()
            except Exception as exc:
                out["cpu_baseline_encode"] = {"error": repr(exc)}
        v, ce = out.get("valid_e2e", {}), out.get("cpu_baseline_encode", {})
        if "valid_seconds" in v and "images_per_s" in ce:
            # the same valid() on this host's cores, from the two CPU legs of this run (encode of Q + R pairs + 4 calc_map_k)
            n = Q + Rn
            v["cpu_estimate_seconds"] = {"encode": n / ce["images_per_s"] + n / ce["captions_per_s"], "retrieve": 4 * Q * Rn / out["cpu_baseline"]["value"],
                                         "how": "(Q + R) / cpu_baseline_encode rates + 4 Q R / cpu_baseline pairs/s, %d threads" % out["cpu_baseline"]["cores"]}
    if use_dist:
        dist.destroy_process_group()
    emit(out)


if __name__ == "__main__":
    main()
