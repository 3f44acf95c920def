"""N>1 exchange logic (xmh/sharded.py) under a real world_size-2 process group on CPU (gloo): contiguous gallery
shards, all-gather of per-shard bucket histograms, rank offsets, all-reduce of AP sums, ragged row gather, top-k
merge.  Per-shard compute is injected from the C oracle (the HIP ops need a GPU); what is under test is the
collective choreography, which is identical on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleShardOps:
    """same interface as xmh.sharded.HipShardOps, backed by oracle/c_oracle.py"""

    def __init__(self, qb, ql, rb, rl, nb):
        self.a = (qb, ql, rb, rl)
        self.nb = nb

    def histograms(self):
        from oracle import c_oracle as co
        ha, hr = co.hist(*self.a, self.nb)
        return torch.from_numpy(ha.astype(np.int32)), torch.from_numpy(hr.astype(np.int32))

    def ap_sums(self, k, base_all, base_rel, nrel_total):
        from oracle import c_oracle as co
        s, cap = co.ap(*self.a, self.nb, k, base_all.numpy().view(np.uint32), base_rel.numpy().view(np.uint32),
                       nrel_total.numpy().view(np.uint32))
        return torch.from_numpy(s), torch.from_numpy(cap)

    def totals(self):
        """the shard's totals table in the layout the HIP workspace keeps it: [nbuckets, qpad (= Q here), 2] {all, relevant}"""
        ha, hr = self.histograms()
        return torch.stack([ha.t(), hr.t()], dim=2).contiguous()

    def map_partial(self, k, totals_gathered, rank):
        """what HipShardOps.map_partial does in one library call: offsets from the gathered tables, pass 2, this shard's share"""
        from xmh import sharded
        ha = totals_gathered[..., 0].transpose(1, 2).contiguous()     # [world, Q, nb]
        hr = totals_gathered[..., 1].transpose(1, 2).contiguous()
        base_a, base_r, nrel = sharded.rank_offsets(ha, hr, rank)
        ap, cap = self.ap_sums(k, base_a, base_r, nrel)
        return (ap / cap.to(torch.float64)).sum().reshape(1) / ap.shape[0]


    def slice_offsets(self, totals_slices):
        from xmh import sharded
        return sharded.slice_offsets_reference(totals_slices)

    def map_partial_offsets(self, k, offsets):
        """what xmh_hamming_map_sharded_offsets does: rows by slice owner -> [Q, nb] offsets, pass 2, this shard's share"""
        world, nb1, S, _ = offsets.shape
        rows = offsets.permute(1, 0, 2, 3).reshape(nb1, world * S, 2)            # [nb + 1, qpad, 2]
        Q = self.a[0].shape[0]
        base_a = rows[:-1, :Q, 0].t().contiguous()
        base_r = rows[:-1, :Q, 1].t().contiguous()
        nrel = rows[-1, :Q, 0].contiguous()
        ap, cap = self.ap_sums(k, base_a, base_r, nrel)
        return (ap / cap.to(torch.float64)).sum().reshape(1) / ap.shape[0]


class PaddedOracleShardOps(OracleShardOps):
    """the totals table padded to a multiple of 4 queries, as the HIP plan pads to 64 / 128: the all-to-all exchange needs
    qpad % world == 0"""

    def totals(self):
        t = super().totals()
        qpad = (t.shape[1] + 3) // 4 * 4
        out = torch.zeros(t.shape[0], qpad, 2, dtype=t.dtype)
        out[:, : t.shape[1]] = t
        return out


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "clip-based-cross-modal-hash_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        from oracle import c_oracle as co
        from xmh import sharded
        rng = np.random.default_rng(123)                      # identical data on every rank
        Q, R, K, C = 37, 1501, 64, 40
        qb = rng.integers(0, 2**32, size=(Q, 2), dtype=np.uint32)
        rb = rng.integers(0, 2**32, size=(R, 2), dtype=np.uint32)[rng.integers(0, 300, size=R)]     # many ties
        ql = rng.integers(0, 2**32, size=(Q, 2), dtype=np.uint32) & rng.integers(0, 2**32, size=(Q, 2), dtype=np.uint32)
        rl = rng.integers(0, 2**32, size=(R, 2), dtype=np.uint32) & rng.integers(0, 2**32, size=(R, 2), dtype=np.uint32)
        ql[:, 1] &= 0xFF
        rl[:, 1] &= 0xFF
        ql[:, 0] |= 1
        rl[::5, 0] |= 1
        b = sharded.shard_bounds(R, world)
        lo, hi = b[rank], b[rank + 1]
        # ragged gather of the query rows each rank "encoded"
        qbnd = sharded.shard_bounds(Q, world)
        mine = torch.from_numpy(qb[qbnd[rank]:qbnd[rank + 1]].view(np.int32))
        full = sharded.all_gather_rows(mine, [qbnd[r + 1] - qbnd[r] for r in range(world)])
        assert np.array_equal(full.numpy().view(np.uint32), qb)
        for k in (None, 5):
            ops = OracleShardOps(qb, ql, rb[lo:hi], rl[lo:hi], K + 1)
            m, ap, cap = sharded.map_k_sharded(ops, k)
            want_s, want_cap = co.ap(qb, ql, rb, rl, K + 1, k)
            assert np.array_equal(cap.numpy(), want_cap)
            assert np.allclose(ap.numpy(), want_s, rtol=1e-12)
            assert abs(float(m) - float(np.mean(want_s / want_cap))) < 1e-12
            m1, ap1, cap1 = sharded.map_k_sharded(ops, k, map_only=True, exchange="gather")      # shares of the mean, one scalar all-reduce
            assert ap1 is None and cap1 is None and abs(float(m1) - float(m)) < 1e-12
            # the all-to-all exchange by query slice against the all-gather it replaces (Q = 37 padded to 40: slices of 20 queries)
            pops = PaddedOracleShardOps(qb, ql, rb[lo:hi], rl[lo:hi], K + 1)
            m2, _, _ = sharded.map_k_sharded(pops, k, map_only=True, exchange="alltoall")
            assert abs(float(m2) - float(m1)) < 1e-12
            m3, _, _ = sharded.map_k_sharded(pops, k, map_only=True)                          # "auto" picks it: 40 % 2 == 0
            assert float(m3) == float(m2)
            with pytest.raises(ValueError):
                sharded.map_k_sharded(ops, k, map_only=True, exchange="alltoall")             # 37 queries do not split over 2 ranks
            with pytest.raises(ValueError):
                sharded.map_k_sharded(ops, k, map_only=True, exchange="all-to-all")           # unknown spellings are not "auto"
        # ranks whose tables differ in shape (a per-process XMH_SCAN_* switch changes the query padding) would enter different
        # collectives: the shape agreement check turns the hang into an error on every rank
        class Mispadded(PaddedOracleShardOps):
            def totals(self):
                t = super().totals()
                return t if rank == 0 else torch.cat([t, torch.zeros(t.shape[0], 4, 2, dtype=t.dtype)], dim=1)
        with pytest.raises(RuntimeError, match="ranks disagree"):
            sharded.map_k_sharded(Mispadded(qb, ql, rb[lo:hi], rl[lo:hi], K + 1), None, map_only=True)
        # the same over query blocks (asynchronous gathers, one per block): identical results
        for nblk in (2, 3):
            qbl = sharded.shard_bounds(Q, nblk)
            for k in (None, 5):
                blocks = sharded.QueryBlocks([OracleShardOps(qb[a:c], ql[a:c], rb[lo:hi], rl[lo:hi], K + 1) for a, c in zip(qbl[:-1], qbl[1:])])
                m, ap, cap = sharded.map_k_sharded(blocks, k)
                want_s, want_cap = co.ap(qb, ql, rb, rl, K + 1, k)
                assert np.array_equal(cap.numpy(), want_cap) and np.allclose(ap.numpy(), want_s, rtol=1e-12)
                assert abs(float(m) - float(np.mean(want_s / want_cap))) < 1e-12
        # top-k: per-shard exact lists gathered and merged on the host
        kk = 20
        d, i = co.topk(qb, rb[lo:hi], K + 1, kk, base_index=lo)
        gd = [torch.empty(Q, kk, dtype=torch.int32) for _ in range(world)]
        gi = [torch.empty(Q, kk, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(gd, torch.from_numpy(d.astype(np.int32)))
        dist.all_gather(gi, torch.from_numpy(i))
        md, mi = sharded.merge_topk(torch.stack(gd), torch.stack(gi), kk)
        wd, wi = co.topk(qb, rb, K + 1, kk)
        assert np.array_equal(mi.numpy(), wi) and np.array_equal(md.numpy().astype(np.uint16), wd)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_size_2_sharded_map_and_topk(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


def _worker_flags_topk(rank, world, port, tmp):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for p in (root, os.path.join(root, "clip-based-cross-modal-hash_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        from oracle import c_oracle as co
        from xmh import sharded
        # only rank 1's shard saw an exact 0 (flag bit0), only rank 0 saw an un-quantised value (bit1): after the
        # reduction every rank holds the OR, so every rank keeps / drops its zero planes alike (same bucket count everywhere)
        flags = torch.tensor([2 if rank == 0 else 1], dtype=torch.int32)
        sharded.reduce_flags(flags)
        assert int(flags.item()) == 3
        flags = torch.zeros(1, dtype=torch.int32)
        sharded.reduce_flags(flags)
        assert int(flags.item()) == 0
        # gather to the writer only
        b = sharded.shard_bounds(11, world)
        mine = torch.arange(b[rank], b[rank + 1], dtype=torch.int32).reshape(-1, 1).repeat(1, 3)
        full = sharded.gather_rows_to(mine, [b[r + 1] - b[r] for r in range(world)], dst=0)
        if rank == 0:
            assert full.shape == (11, 3) and torch.equal(full[:, 0], torch.arange(11, dtype=torch.int32))
        else:
            assert full is None
        # topk_sharded: shard lists -> one all-gather -> host merge == global top-k; second round: rank 1 owns no rows at all
        rng = np.random.default_rng(7)
        Q, R, K, kk = 9, 700, 64, 25
        qb = rng.integers(0, 2**32, size=(Q, 2), dtype=np.uint32)
        rb = rng.integers(0, 2**32, size=(R, 2), dtype=np.uint32)[rng.integers(0, 90, size=R)]

        def fn(q, r, k, base):
            d, i = co.topk(q, r, K + 1, k, base_index=base)
            return torch.from_numpy(d.view(np.int16).copy()), torch.from_numpy(i)
        wd, wi = co.topk(qb, rb, K + 1, kk)
        for bounds in (sharded.shard_bounds(R, world), [0, R, R]):
            lo, hi = bounds[rank], bounds[rank + 1]
            md, mi = sharded.topk_sharded(qb, rb[lo:hi], kk, lo, topk_fn=fn)
            assert np.array_equal(mi.numpy(), wi) and np.array_equal(md.numpy().astype(np.uint16), wd)
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_size_2_flags_writer_gather_and_topk_sharded(tmp_path):
    world = 2
    mp.spawn(_worker_flags_topk, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(world))


@pytest.mark.timeout(300)
def test_bench_self_launches_n_ranks_dry_run():
    """`python bench.py --gpus 2` (no RANK/WORLD_SIZE in the environment, the way the round driver calls it) re-executes
    itself under torch.distributed.run; --dry-run walks the launcher, the rendezvous and one step's collectives over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["ranks_in_group"] == 2
    # N=1 takes no launcher
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["ranks_in_group"] == 1


@pytest.mark.timeout(600)
def test_bench_self_launches_eight_ranks_dry_run():
    """the driver's N = 8 command line with --dry-run: launcher, rendezvous of eight ranks and one step's collectives over gloo"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run"], env=env, capture_output=True, text=True, timeout=580)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 8 and d["ranks_in_group"] == 8


def test_topk_merge_host_equals_the_torch_merge():
    """xmh_topk_merge_host (k-way merge in C on the gathered records) against sharded.merge_topk (argsort of 64-bit keys), bit for
    bit: ties across shards, lists shorter than k (index -1 padding), shards without rows, fewer rows than k in all."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "clip-based-cross-modal-hash_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from xmh import sharded
    from xmh._lib import lib
    rng = np.random.default_rng(3)
    for world, Q, k, rows in ((8, 5, 100, 300), (3, 17, 7, 4), (2, 1, 1, 1), (4, 9, 20, 0), (5, 3, 16, 2)):
        rec = int(lib.xmh_topk_record_bytes(Q, k))
        assert rec >= Q * k * 6 and rec % 4 == 0
        gathered = torch.zeros(world, rec, dtype=torch.uint8)
        ds, is_ = [], []
        base = 0
        for w in range(world):
            n = 0 if (w == 1 and world > 2) else rows + w                    # one shard owns no rows at all
            d = np.full((Q, k), 0xFFFF, dtype=np.uint16)
            i = np.full((Q, k), -1, dtype=np.int32)
            for q in range(Q):
                m = min(n, k)
                dd = np.sort(rng.integers(0, 6, size=n))[:m]                 # few distinct distances: many ties
                ii = np.concatenate([np.sort(rng.choice(n, size=int((dd == v).sum()), replace=False)) for v in np.unique(dd)]) if m else np.zeros(0, np.int64)
                d[q, :m], i[q, :m] = dd, base + ii
            base += n
            gathered[w, : Q * k * 4] = torch.from_numpy(i.view(np.uint8).reshape(-1).copy())
            gathered[w, Q * k * 4: Q * k * 6] = torch.from_numpy(d.view(np.uint8).reshape(-1).copy())
            ds.append(torch.from_numpy(d.view(np.int16).copy()))
            is_.append(torch.from_numpy(i))
        got_d, got_i = sharded.merge_topk_records(gathered, world, Q, k)
        want_d, want_i = sharded.merge_topk(torch.stack(ds), torch.stack(is_), k)
        assert torch.equal(got_i, want_i), (world, Q, k)
        live = want_i >= 0
        assert torch.equal(got_d[live], want_d[live]) and bool((got_d[~live] == 0xFFFF).all())
