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
