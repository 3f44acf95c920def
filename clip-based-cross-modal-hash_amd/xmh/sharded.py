"""Gallery-sharded retrieval across the GPUs of one node.

Replaces the reference's eval gather -- a dense all_reduce(SUM) of the zero-initialised [N,K] fp32 code
buffers that leaves every rank with the whole gallery (runners/base.py:259-264) -- with:

  * rank r keeps the packed codes + label masks of a CONTIGUOUS gallery index range (so the canonical
    (distance, index) order is shard-major and in-bucket offsets are prefix sums over lower ranks);
  * packed query codes/labels are all-gathered (<= 210 KB in total);
  * mAP: per-shard bucket histograms [Q, K+1] x2 are all-gathered, every rank derives its rank offsets,
    runs pass 2 on its shard and the [Q] partial sums are all-reduced (RCCL over xGMI; gloo in CPU tests);
  * top-k: per-shard exact top-k lists are gathered and merged on the host.

The collectives are the only thing this module does itself; per-shard compute is delegated to a
``ShardOps`` object -- the HIP ops in production, injectable so the exchange logic can be exercised with
world_size-2 gloo tests on CPU.
"""
from __future__ import annotations

from typing import Optional, Sequence

import threading

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int) -> list:
    """contiguous, balanced: first n % world shards get one extra row."""
    base, extra = divmod(n, world)
    out = [0]
    for r in range(world):
        out.append(out[-1] + base + (1 if r < extra else 0))
    return out


class HipShardOps:
    """Per-shard compute through libxmh.so (the product path).  A rank may own no gallery rows at all (fewer rows than
    ranks): it then contributes zero histograms and zero partial sums, and derives the caps like every other rank."""

    def __init__(self, q, qlab, r, rlab, C):
        from . import retrieval as R
        self._R = R
        self.q = R.widened(q)
        self.empty = r.n == 0
        self.scan = None if self.empty else R.RankingScan(q, qlab, r, rlab, C)
        self._ternary = (q.zero is not None) or (r.zero is not None)

    def histograms(self):
        if self.empty:
            nb = self._R.scan_plan(self.q.n, 1, self.q.K, self._ternary).nbuckets
            pair = torch.zeros(2, self.q.n, nb, dtype=torch.int32, device=self.q.bits.device)
            return pair[0], pair[1]
        return self.scan.histograms(True)

    def offsets(self, hist_gathered, rank):
        return self._R.shard_offsets(hist_gathered, rank)        # one kernel instead of a dozen tensor ops

    def ap_sums(self, k, base_all, base_rel, nrel_total):
        if self.empty:
            cap = nrel_total if k is None else torch.clamp(nrel_total, max=int(k))
            return torch.zeros(self.q.n, dtype=torch.float64, device=nrel_total.device), cap.to(torch.int32)
        return self.scan.ap_sums(k, base_all, base_rel, nrel_total)

    def finalize(self, ap, cap):
        return self._R.map_finalize(ap, cap)

    def totals(self):
        """pass 1 on the local shard; returns its totals table [nbuckets, qpad, 2] int32 (a view of the scan workspace) -- what
        the fused sharded evaluation all-gathers instead of exported [Q, nbuckets] histograms"""
        if self.empty:
            p = self._R.scan_plan(self.q.n, 1, self.q.K, self._ternary)
            return torch.zeros(p.nbuckets, p.qpad, 2, dtype=torch.int32, device=self.q.bits.device)
        self.scan.histograms(False)
        return self.scan.totals()

    def map_partial(self, k, totals_gathered, rank):
        """this shard's share of the mAP (offsets + pass 2 + reduction in one library call; the shares add up over the shards)"""
        if self.empty:
            nrel = totals_gathered[:, :, : self.q.n, 1].sum(dim=(0, 1))
            cap = nrel if k is None else torch.clamp(nrel, max=int(k))
            z = torch.zeros(self.q.n, dtype=torch.float64, device=totals_gathered.device)
            return (z / cap.to(torch.float64)).sum().reshape(1) / self.q.n      # cap == 0 -> NaN, like every other rank
        return self.scan.map_sharded(k, totals_gathered, rank)[0]


    def slice_offsets(self, totals_slices):
        """all-to-all exchange, slice owner's part (one kernel): [world, nb, S, 2] -> [world, nb + 1, S, 2]"""
        return self._R.slice_offsets(totals_slices)

    def map_partial_offsets(self, k, offsets):
        """this shard's share of the mAP from its offset rows [world (slice owner), nb + 1, S, 2]"""
        if self.empty:
            nrel = offsets[:, -1, :, 0].reshape(-1)[: self.q.n]
            cap = nrel if k is None else torch.clamp(nrel, max=int(k))
            z = torch.zeros(self.q.n, dtype=torch.float64, device=offsets.device)
            return (z / cap.to(torch.float64)).sum().reshape(1) / self.q.n      # cap == 0 -> NaN, like every other rank
        return self.scan.map_sharded_offsets(k, offsets)[0]


def slice_offsets_reference(totals_slices: torch.Tensor) -> torch.Tensor:
    """torch statement of xmh_shard_slice_offsets (test doubles; the HIP kernel is checked against it on the GPU)"""
    t = totals_slices.to(torch.int64)                                # [world, nb, S, 2]
    tot = t.sum(0)                                                   # [nb, S, 2]
    lower = torch.cumsum(tot, 0) - tot                               # lower buckets on any shard
    below = torch.cumsum(t, 0) - t                                   # same bucket on lower shards
    rows = lower.unsqueeze(0) + below                                # [world, nb, S, 2]
    last = torch.stack([tot[..., 1].sum(0), tot[..., 0].sum(0)], dim=-1)          # {relevant, all} over every shard and bucket
    last = last.unsqueeze(0).unsqueeze(0).expand(t.shape[0], 1, -1, -1)
    return torch.cat([rows, last], dim=1).to(torch.int32).contiguous()


def _range(name: str):
    """roctx range around a collective (xmh_range_push / _pop; a no-op unless _lib.prof_enable(2)): the exchange steps show up between
    the pass-1 and pass-2 ranges of a `rocprofv3 --marker-trace` timeline"""
    from ._lib import prof_range
    return prof_range(name)


def all_gather_rows(t: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """all_gather of row-ragged tensors (rank i contributes counts[i] rows) -> concatenated rows."""
    world = dist.get_world_size(group)
    m = max(counts)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    with _range("collective: all_gather rows (packed codes)"):
        dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


def rank_offsets(hist_all: torch.Tensor, hist_rel: torch.Tensor, rank: int):
    """[world, Q, nb] per-shard bucket counts -> (base_all, base_rel, nrel_total) for ``rank``:
    base[q, d] = (# items in buckets < d on ANY shard) + (# items in bucket d on shards < rank)."""
    ha, hr = hist_all.to(torch.int64), hist_rel.to(torch.int64)
    tot_a, tot_r = ha.sum(0), hr.sum(0)
    lower_a = torch.cumsum(tot_a, 1) - tot_a
    lower_r = torch.cumsum(tot_r, 1) - tot_r
    base_a = lower_a + ha[:rank].sum(0)
    base_r = lower_r + hr[:rank].sum(0)
    nrel = tot_r.sum(1)
    i32 = torch.int32
    return base_a.to(i32).contiguous(), base_r.to(i32).contiguous(), nrel.to(i32).contiguous()


def _all_gather_stacked(out: torch.Tensor, inp: torch.Tensor, group=None, async_op: bool = False):
    """all_gather_into_tensor with ``out`` = [world, *inp.shape], handed over as the concatenation along dim 0 (the same memory): RCCL takes
    either form, gloo -- which carries device tensors in the one-GPU two-rank tests -- only the concatenated one."""
    flat = out.view((out.shape[0] * inp.shape[0],) + tuple(inp.shape[1:]))
    return dist.all_gather_into_tensor(flat, inp, group=group, async_op=async_op)


def _gather_hist_pair(ha: torch.Tensor, hr: torch.Tensor, group=None, async_op: bool = False):
    """both histogram planes of every shard in ONE collective -> [world, 2, Q, nb] (async_op: (tensor, work handle))."""
    world = dist.get_world_size(group)
    base = getattr(ha, "_base", None)
    if base is not None and base.dim() == 3 and base.shape[0] == 2 and base.is_contiguous() and hr._base is base:
        pair = base                                              # RankingScan.histograms hands out views of one buffer
    else:
        pair = torch.stack([ha, hr]).contiguous()
    out = torch.empty((world,) + tuple(pair.shape), dtype=pair.dtype, device=pair.device)
    if hasattr(dist, "all_gather_into_tensor") and pair.is_cuda:
        work = _all_gather_stacked(out, pair, group=group, async_op=async_op)
    else:                                                        # gloo (CPU tests) has no all_gather_into_tensor
        work = dist.all_gather(list(out.unbind(0)), pair, group=group, async_op=async_op)
    return (out, work) if async_op else out


class QueryBlocks:
    """The query set cut into contiguous blocks, one shard-ops object per block (same gallery shard).  map_k_sharded then
    pipelines them: the histogram gather of block b travels while pass 1 of block b+1 (and pass 2 of block b-1) runs, so
    only the first gather's head and the last one's tail stay on the critical path."""

    def __init__(self, blocks: Sequence):
        if not blocks:
            raise ValueError("QueryBlocks needs at least one block")
        self.blocks = list(blocks)

    @staticmethod
    def split(q, qlab, r, rlab, C, nblocks: int) -> "QueryBlocks":
        """HipShardOps per query block; q / qlab are PackedCodes / packed labels of ALL queries."""
        n = q.n
        bounds = shard_bounds(n, max(1, min(int(nblocks), n)))
        return QueryBlocks([HipShardOps(q.rows(lo, hi), qlab[lo:hi], r, rlab, C) for lo, hi in zip(bounds[:-1], bounds[1:])])


def map_k_sharded(ops, k: Optional[int] = None, group=None, map_only: bool = False, exchange: str = "auto"):
    """mAP over a gallery sharded across ``group``.  ``ops`` wraps this rank's shard (HipShardOps or a test
    double) and already holds the FULL (all-gathered) query set.  Returns (map float64 tensor [1], ap_sum,
    cap) -- identical on every rank.  Per call: pass 1, ONE all-gather of the [2, Q, nb] histograms (2.6 MB/rank at
    Q=5000, K=64), one offsets kernel, pass 2, one all-reduce of [Q] f64, one finalize kernel.
    ``map_only``: the caller wants the mean alone -> (map, None, None): the shards' totals tables are gathered as pass 1 left
    them in the workspace (no export pass), every rank folds its own share of the mean into pass 2's reduction
    (ops.map_partial: offsets, pass 2 and reduction are three launches of one library call) and ONE 8-byte all-reduce adds the
    shares; no [Q] all-reduce, no finalize launch.  ``exchange`` (map_only): "alltoall" = two all-to-alls by query slice (see
    below), "gather" = the all-gather of whole tables it replaces (kept as the checked reference), "auto" = all-to-all whenever
    the padded query count divides by the world size.  ``map_only`` is ignored for QueryBlocks (they pipeline the [Q] form)."""
    if exchange not in ("auto", "alltoall", "gather"):
        raise ValueError("map_k_sharded: exchange must be 'auto', 'alltoall' or 'gather', not %r" % (exchange,))
    rank = dist.get_rank(group)
    if isinstance(ops, QueryBlocks):
        m, ap, cap = _map_k_blocks(ops.blocks, k, rank, group)
        return (m, None, None) if map_only else (m, ap, cap)     # the blocks pipeline the [Q] form; map_only only trims the result
    if map_only and hasattr(ops, "totals"):
        t = ops.totals()                                         # pass 1; the shard's totals table where pass 1 left it
        world = dist.get_world_size(group)
        nb, qpad = t.shape[0], t.shape[1]
        # which collective follows is decided from the table shape, and the padded query count depends on per-process
        # environment switches (XMH_SCAN_MFMA, XMH_SCAN_CACHE_MB, XMH_SCAN_AP_R2): ranks that disagree would enter different
        # collectives and hang until the group's timeout.  One 16-byte MIN/MAX all-reduce turns that into an error.
        _require_same_shape(nb, qpad, hasattr(ops, "slice_offsets"), t.device, group)
        if exchange != "gather" and hasattr(ops, "slice_offsets") and qpad % world == 0:
            # all-to-all by query slice: rank j resolves the offsets of qpad / world queries for every shard.  Per rank 2 x the table
            # travels (2 x 2.6 MB at Q 5000, K 64) instead of world x (21 MB at 8 ranks), and the offsets are computed once, not
            # world times.  The column slices are not contiguous in the table: one strided copy each way.
            S = qpad // world
            send = t.view(nb, world, S, 2).permute(1, 0, 2, 3).contiguous()      # [world (slice owner), nb, S, 2]
            recv = torch.empty_like(send)
            with _range("collective: all_to_all totals-table query slices"):
                dist.all_to_all_single(recv, send, group=group)  # [world (shard), nb, S, 2]: every shard's columns of MY slice
            offs = ops.slice_offsets(recv)                       # [world (shard), nb + 1, S, 2]
            back = torch.empty_like(offs)
            with _range("collective: all_to_all offset rows"):
                dist.all_to_all_single(back, offs, group=group)  # [world (slice owner), nb + 1, S, 2]: MY rows for every slice
            m = ops.map_partial_offsets(k, back)                 # scatter, pass 2, this shard's share of the mean
            with _range("collective: all_reduce mAP [1] f64"):
                dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
            return m, None, None
        if exchange == "alltoall":
            raise ValueError("map_k_sharded: the all-to-all exchange needs qpad %% world == 0 (qpad=%d, world=%d)" % (qpad, world))
        g = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        with _range("collective: all_gather totals tables"):
            if hasattr(dist, "all_gather_into_tensor") and t.is_cuda:
                _all_gather_stacked(g, t, group=group)            # [world, nb, qpad, 2]
            else:
                dist.all_gather(list(g.unbind(0)), t, group=group)
        m = ops.map_partial(k, g, rank)                          # offsets, pass 2, this shard's share of the mean
        with _range("collective: all_reduce mAP [1] f64"):
            dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)    # [1] f64
        return m, None, None
    ha, hr = ops.histograms()                                    # pass 1 on the local shard
    g = _gather_hist_pair(ha, hr, group)                         # [world, 2, Q, nb]
    if hasattr(ops, "offsets"):
        base_a, base_r, nrel = ops.offsets(g, rank)
    else:
        base_a, base_r, nrel = rank_offsets(g[:, 0], g[:, 1], rank)
    ap, cap = ops.ap_sums(k, base_a, base_r, nrel)               # pass 2 on the local shard
    dist.all_reduce(ap, op=dist.ReduceOp.SUM, group=group)       # [Q] f64
    if hasattr(ops, "finalize"):
        m = ops.finalize(ap, cap)
    else:
        m = (ap / cap.to(torch.float64)).mean().reshape(1)       # cap == 0 -> NaN like the reference
    return m, ap, cap


_shape_seen = {}      # process group -> the table shape the ranks last agreed on


def _require_same_shape(nb: int, qpad: int, has_slices: bool, device, group) -> None:
    """checked when a group is first used and whenever the local shape changes (in a consistent job every rank changes at the same
    call), so the steady state pays nothing for it"""
    key, shape = id(group), (nb, qpad, has_slices)
    if _shape_seen.get(key) == shape:
        return
    v = torch.tensor([nb, qpad, int(has_slices)], dtype=torch.int64, device=device)
    both = torch.stack([v, -v])                                  # MAX of (v, -v) = (max, -min)
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    hi, lo = both[0], -both[1]
    if not torch.equal(hi, lo):
        raise RuntimeError("map_k_sharded: ranks disagree on the totals table (nbuckets, padded queries, slice kernel): min %s max %s "
                           "-- the XMH_SCAN_* environment must be the same on every rank" % (lo.tolist(), hi.tolist()))
    _shape_seen[key] = shape


def _offsets(ops, g, rank):
    if hasattr(ops, "offsets"):
        return ops.offsets(g, rank)
    return rank_offsets(g[:, 0], g[:, 1], rank)


def _map_k_blocks(blocks, k, rank, group):
    """map_k_sharded over query blocks.  Every rank issues the same collectives in the same order (one gather per block, in
    block order, then one all-reduce).  The gathers are asynchronous: torch runs them on the process group's own stream,
    ordered after the kernels enqueued so far, and ``wait()`` orders the current stream after them -- no host blocking on
    RCCL."""
    inflight = []
    for b in blocks:
        ha, hr = b.histograms()                                  # pass 1 of block b; its gather overlaps pass 1 of b+1
        inflight.append(_gather_hist_pair(ha, hr, group, async_op=True))
    aps, caps = [], []
    for b, (g, work) in zip(blocks, inflight):
        work.wait()
        base_a, base_r, nrel = _offsets(b, g, rank)
        ap, cap = b.ap_sums(k, base_a, base_r, nrel)             # pass 2 of block b; overlaps the gather of b+1
        aps.append(ap)
        caps.append(cap)
    ap, cap = torch.cat(aps), torch.cat(caps)
    dist.all_reduce(ap, op=dist.ReduceOp.SUM, group=group)       # [Q] f64
    fin = blocks[0]
    if hasattr(fin, "finalize"):
        m = fin.finalize(ap, cap)
    else:
        m = (ap / cap.to(torch.float64)).mean().reshape(1)
    return m, ap, cap


def merge_topk(dists: torch.Tensor, idxs: torch.Tensor, k: int):
    """Host merge of per-shard exact top-k lists (north_star: 'partial top-k lists merged on the host').
    dists/idxs: [world, Q, k] (any device; int16 distances are the uint16 bit patterns xmh_hamming_topk writes).
    Order key = (distance, global index); unused slots carry idx -1 and are pushed to the end."""
    d = dists.to("cpu").to(torch.int64)
    if dists.dtype == torch.int16:
        d = d & 0xFFFF
    d = d.permute(1, 0, 2).reshape(dists.shape[1], -1)
    i = idxs.to("cpu").to(torch.int64).permute(1, 0, 2).reshape(idxs.shape[1], -1)
    key = torch.where(i < 0, torch.full_like(d, 1 << 62), (d << 32) | i)
    order = torch.argsort(key, dim=1)[:, :k]
    return torch.gather(d, 1, order).to(torch.int32), torch.gather(i, 1, order).to(torch.int32)


def reduce_flags(flags: torch.Tensor, group=None) -> torch.Tensor:
    """bitwise OR of the quantiser's value flags (int32 [1]: bit0 = an exact 0 was seen, bit1 = a value outside
    {-1,0,+1}) over all ranks, in place.  Every rank must rank its shard in the SAME mode: a zero seen on one rank only
    would otherwise leave that rank with 2K+1 half-unit buckets and the others with K+1, and the histogram all-gather
    with mismatched shapes.  RCCL has no bitwise reductions: the bits travel as separate MAX lanes."""
    lanes = torch.stack([(flags.reshape(-1)[0] >> b) & 1 for b in range(2)]).to(torch.int32)
    dist.all_reduce(lanes, op=dist.ReduceOp.MAX, group=group)
    flags.reshape(-1)[0] = lanes[0] | (lanes[1] << 1)
    return flags


def gather_rows_to(t: torch.Tensor, counts: Sequence[int], dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """like all_gather_rows, but only ``dst`` receives the concatenation (others get None): the .mat writer needs the
    whole retrieval set on one rank only."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    m = max(counts)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p[:c] for p, c in zip(parts, counts)])


_pinned = {}          # record bytes -> pinned host staging buffer for the gathered lists (reused over calls)
# The prepared top-k workspace of the last (Q, R, K, k, device, stream) topk_sharded ran on THIS thread: a query loop over one gallery
# shard calls with the same shape every time.  One entry per thread (a workspace is for one stream at a time), dropped when a call through
# it fails (its control words may be dirty) and by release_topk_workspace() -- a 10 M x 256-bit shape holds its device buffer otherwise
# (ADVICE r4).
_topk_local = threading.local()


def release_topk_workspace() -> None:
    """free the prepared top-k workspace (and the pinned staging buffers) topk_sharded keeps between calls"""
    _topk_local.__dict__.pop("entry", None)
    _pinned.clear()


def _prepared_topk_workspace(shape, make):
    entry = _topk_local.__dict__.get("entry")
    if entry is None or entry[0] != shape:
        entry = _topk_local.entry = (shape, make())
    return entry[1]


def merge_topk_records(gathered: torch.Tensor, world: int, nq: int, k: int):
    """xmh_topk_merge_host over `world` gathered records ([Q][k] i32 indices then [Q][k] u16 distances each, uint8 tensor on any
    device): ONE device-to-host copy into pinned memory, then a k-way merge in C.  Returns (dist, idx) int32 [Q, k] CPU tensors."""
    from ._lib import check, lib, ptr
    rec = int(lib.xmh_topk_record_bytes(nq, k))
    if gathered.dtype != torch.uint8 or gathered.numel() != world * rec:
        raise ValueError("merge_topk_records: expected %d x %d bytes of records" % (world, rec))
    if gathered.is_cuda:
        host = _pinned.get(world * rec)
        if host is None:
            host = _pinned[world * rec] = torch.empty(world * rec, dtype=torch.uint8).pin_memory()
        host.copy_(gathered.reshape(-1), non_blocking=True)
        torch.cuda.current_stream(gathered.device).synchronize()
    else:
        host = gathered.reshape(-1).contiguous()
    d = torch.empty(nq, k, dtype=torch.int32)
    i = torch.empty(nq, k, dtype=torch.int32)
    check(lib.xmh_topk_merge_host(ptr(host), world, nq, k, ptr(d), ptr(i)), "xmh_topk_merge_host")
    return d, i


def topk_sharded(q, r_shard, k: int, base_index: int, group=None, topk_fn=None, ternary: Optional[bool] = None):
    """north_star retrieval mode over a sharded gallery: exact top-k of every query on this rank's shard (global indices
    = base_index + row) written straight into this rank's record, ONE all-gather of the records ([Q, k] indices + distances,
    6 bytes per entry), one pinned device-to-host copy, k-way merge on the host (xmh_topk_merge_host).  Returns
    (dist int32 [Q,k], idx int32 [Q,k]) CPU tensors, identical on every rank; unused slots (fewer than k rows in all)
    carry distance 0xFFFF and idx -1.  ``topk_fn(q, r_shard, k, base_index) -> (dist, idx)`` defaults to the HIP op; a rank
    without gallery rows contributes empty lists.
    Ternary code sets (sign_() left an exact 0 somewhere, reference runners/base.py:407-410) are ranked in half units (K - q.r) and
    EVERY rank must use that unit: ``ternary=None`` agrees on it with one 4-byte all-reduce (MAX of "my query set or my shard has a
    zero plane"), ``True`` / ``False`` is the caller's word (e.g. from reduce_flags over the quantiser's value flags) and costs nothing."""
    from ._lib import lib
    world = dist.get_world_size(group)
    n_rows = r_shard.n if hasattr(r_shard, "n") else len(r_shard)
    nq = q.n if hasattr(q, "n") else len(q)
    hip = topk_fn is None
    dev = q.bits.device if hip else torch.device("cpu")
    rec = int(lib.xmh_topk_record_bytes(nq, k))
    out = torch.empty(world, rec, dtype=torch.uint8, device=dev)            # the all-gather lands here; this rank's record is written in place
    mine = out[dist.get_rank(group)]
    i_view = mine[: nq * k * 4].view(torch.int32).view(nq, k)
    d_view = mine[nq * k * 4: nq * k * 6].view(torch.int16).view(nq, k)
    if hip and ternary is None:
        ternary = getattr(q, "zero", None) is not None or getattr(r_shard, "zero", None) is not None
        if world > 1:
            t = torch.tensor([1 if ternary else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            ternary = bool(int(t.item()))
    if n_rows == 0:
        mine.fill_(0xFF)                                                     # distance 0xFFFF, index -1 = unused slots
    elif hip:
        from . import retrieval as R
        # a query loop over one gallery shard calls with the same shape every time: keep the prepared workspace of the last shape
        # (cleared once, left clean by every call) instead of allocating and clearing a scratch one per call
        # (code sets with zero planes -- sign_() left an exact 0 -- rank in half units: every rank must agree on that, see reduce_flags)
        key = R.topk_ws_key(q, r_shard, k, ternary)
        shape = key + (dev, torch.cuda.current_stream(dev).cuda_stream)
        ws = _prepared_topk_workspace(shape, lambda: R.TopkWorkspace(key[0], key[1], key[2], key[3], dev, ternary=key[4]))
        try:
            R.hamming_topk(q, r_shard, k, base_index, workspace=ws, out=(d_view, i_view), ternary=ternary)
        except Exception:
            _topk_local.__dict__.pop("entry", None)                          # its control words may be dirty: the next call prepares a fresh one
            raise
    else:
        d, i = topk_fn(q, r_shard, k, base_index)
        d_view.copy_(d.to(torch.int16) if d.dtype != torch.int16 else d)
        i_view.copy_(i.to(torch.int32))
    with _range("collective: all_gather top-k records"):
        if hasattr(dist, "all_gather_into_tensor") and out.is_cuda:
            # in place (NCCL / RCCL semantics: the input may be the rank's own slot of the output): rank r's record is already at out[r]
            assert mine.data_ptr() == out.data_ptr() + dist.get_rank(group) * rec and mine.is_contiguous()
            _all_gather_stacked(out, mine, group=group)
        else:
            dist.all_gather(list(out.unbind(0)), mine.clone(), group=group)
    with _range("top-k: D2H + host merge"):
        return merge_topk_records(out, world, nq, k)
