"""GPU parity: the HIP retrieval path (through the C ABI) against the oracle and the golden vectors.

Bar: bit-exact for packed bits, distances, histograms and caps; mAP within 1e-6 of the
stable-order oracle/golden (the north-star tolerance is 1e-4).
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

MAP_TOL = 1e-6


@pytest.fixture(scope="module")
def xr():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from xmh import retrieval
    return retrieval


@pytest.fixture(scope="module")
def cu():
    from xmh.common import calc_utils
    return calc_utils


def _orc():
    from oracle import retrieval as orc
    return orc


def _u32(t):
    return t.cpu().numpy().view(np.uint32)


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_library_loaded_is_the_in_tree_hip_build(xr):
    from xmh import _lib
    assert _lib.lib.xmh_version() >= 100
    assert os.path.basename(_lib.LIB_PATH) == "libxmh.so" and "clip-based-cross-modal-hash_amd" in _lib.LIB_PATH


def test_make_hash_code_goldens(xr):
    g = np.load(os.path.join(GOLDEN, "make_hash_code.npz"))
    orc = _orc()
    p = xr.pack_sign(dev(g["sign_in"]))
    assert p.zero is not None and (p.flags & 1)
    wb, wz = orc.pack_bits(g["sign_out"])
    assert np.array_equal(_u32(p.bits), wb)
    assert np.array_equal(_u32(p.zero) & 0xFFFF, wz & 0xFFFF)          # K=16: low half real, padding set
    assert (_u32(p.zero) >> 16 == 0xFFFF).all()
    assert np.array_equal(p.unpack().cpu().numpy(), g["sign_out"])
    pp = xr.pack_pair_argmax(dev(g["pair_in"]))
    assert np.array_equal(pp.unpack().cpu().numpy(), g["pair_out"])


def test_pack_scatter_by_index(xr):
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(37, 64, generator=gen)
    idx = torch.randperm(50, generator=gen)[:37]
    buf = xr.empty_packed(50, 64, "cuda", with_zero=True)
    xr.pack_sign(x.cuda(), out=buf, row_index=idx.cuda())
    want = torch.zeros(50, 64)
    want[idx] = x.sign()
    got = buf.unpack().cpu()
    assert torch.equal(got[idx], want[idx])
    bufp = xr.empty_packed(50, 32, "cuda")
    xr.pack_pair_argmax(torch.rand(37, 64, generator=gen).cuda(), out=bufp, row_index=idx.cuda())


@pytest.mark.parametrize("dtype", [torch.int64, torch.float32, torch.int32, torch.uint8, torch.bool])
def test_pack_labels_dtypes(xr, dtype):
    gen = torch.Generator().manual_seed(5)
    for C in (1, 21, 24, 32, 33, 80, 96, 100):
        L = (torch.rand(77, C, generator=gen) > 0.8).to(dtype)
        got = _u32(xr.pack_labels(L.cuda()))
        assert np.array_equal(got, _orc().pack_labels(L.numpy().astype(np.int8)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "calc_utils_K*.npz"))))
def test_golden_cases(xr, cu, path):
    g = np.load(path)
    qB, rB = dev(g["qB"], torch.float32), dev(g["rB"], torch.float32)
    qL, rL = dev(g["qL"], torch.int64), dev(g["rL"], torch.int64)
    d = cu.calc_hammingDist(qB, rB)
    assert d.dtype == torch.float32 and np.array_equal(d.cpu().numpy().astype(np.int16), g["dist"])
    q, r = xr.pack_sign(qB), xr.pack_sign(rB)
    assert np.array_equal(xr.hamming_dist(q, r, as_u16=True).cpu().numpy(), g["dist"])
    assert np.array_equal(cu.calc_label_sim(qL.float(), rL.float()).cpu().numpy().astype(np.int8), g["label_sim"])
    for k, tag in ((None, "all"), (1, "1"), (2, "2"), (50, "50"), (5000, "5000")):
        got = cu.calc_map_k(qB, rB, qL, rL, k)
        assert got.dtype == torch.float32 and got.dim() == 0 and not got.is_cuda
        assert abs(float(got) - float(g["map_stable_" + tag])) < MAP_TOL, (tag, float(got))
    # CPU-resident inputs are accepted like in the reference
    got = cu.calc_map_k(qB.cpu(), rB.cpu(), qL.cpu(), rL.cpu())
    assert abs(float(got) - float(g["map_stable_all"])) < MAP_TOL


def test_kats(cu):
    g = np.load(os.path.join(GOLDEN, "calc_utils_kat.npz"))
    q, r = dev(g["kat1_q"]), dev(g["kat1_r"])
    qL, rL = dev(g["kat1_qL"]), dev(g["kat1_rL"])
    assert abs(float(cu.calc_map_k(q, r, qL, rL)) - float(g["kat1_map_all"])) < MAP_TOL
    assert abs(float(cu.calc_map_k(q, r, qL, rL, 1)) - 0.75) < MAP_TOL
    assert abs(float(cu.calc_map_k(q, r, qL, rL, 2)) - 2 / 3) < MAP_TOL
    assert torch.isnan(cu.calc_map_k(q, r, qL, dev(g["kat2_rL"])))                  # KAT-2
    with pytest.raises(IndexError):                                                  # KAT-3
        cu.calc_map_k(q[:1], r, qL[:1], rL)


def test_ternary_golden(xr, cu):
    g = np.load(os.path.join(GOLDEN, "calc_utils_ternary_float.npz"))
    qB, rB = dev(g["qB"], torch.float32), dev(g["rB"], torch.float32)
    qL, rL = dev(g["qL"], torch.int64), dev(g["rL"], torch.int64)
    assert np.array_equal(cu.calc_hammingDist(qB, rB).cpu().numpy(), g["dist"])
    assert abs(float(cu.calc_map_k(qB, rB, qL, rL)) - float(g["map_stable_all"])) < MAP_TOL
    assert abs(float(cu.calc_map_k(qB, rB, qL, rL, 50)) - float(g["map_stable_50"])) < MAP_TOL
    # one side binary, the other ternary
    rBb = rB.clone()
    rBb[rBb == 0] = 1.0
    want = _orc().map_k(qB.cpu(), rBb.cpu(), qL.cpu(), rL.cpu())
    assert abs(float(cu.calc_map_k(qB, rBb, qL, rL)) - float(want)) < MAP_TOL


def _synth(Q, R, K, C, seed, p=0.08, structured=True):
    gen = torch.Generator().manual_seed(seed)
    qL = (torch.rand(Q, C, generator=gen) < p)
    rL = (torch.rand(R, C, generator=gen) < p)
    qL[torch.arange(Q), torch.randint(0, C, (Q,), generator=gen)] = True
    rL[torch.arange(R), torch.randint(0, C, (R,), generator=gen)] = True
    if structured:
        Wm = torch.randn(C, K, generator=gen)
        qB = (qL.float() @ Wm + 0.8 * torch.randn(Q, K, generator=gen)).sign()
        rB = (rL.float() @ Wm + 0.8 * torch.randn(R, K, generator=gen)).sign()
        qB[qB == 0] = 1
        rB[rB == 0] = 1
    else:
        qB = torch.randn(Q, K, generator=gen).sign()
        rB = torch.randn(R, K, generator=gen).sign()
    return qB, rB, qL.to(torch.int64), rL.to(torch.int64)


@pytest.mark.parametrize("Q,R,K,C", [(130, 6000, 64, 80), (64, 3001, 16, 24), (70, 2500, 128, 21), (65, 1500, 256, 24),
                                     (200, 9000, 32, 40)])
def test_scan_internals_match_oracle(xr, Q, R, K, C):
    """histogram totals bit-exact, caps exact, per-query AP sums to 1e-6 relative; multi-chunk plans."""
    orc = _orc()
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=K + R)
    q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
    scan = xr.RankingScan(q, ql, r, rl, C)
    ha, hr = scan.histograms()
    dist = orc.hamming_packed(_u32(q.bits), _u32(r.bits))
    rel = orc.relevance_packed(_u32(ql), _u32(rl))
    wa, wr = orc.bucket_histograms(dist, rel, K + 1)
    assert np.array_equal(_u32(ha), wa) and np.array_equal(_u32(hr), wr)
    for k in (None, 7, 100):
        ap, cap = scan.ap_sums(k)
        want = orc.ap_from_ranking(dist, rel, k=k)
        nrel = rel.sum(-1)
        assert np.array_equal(cap.cpu().numpy(), nrel if k is None else np.minimum(nrel, k))
        assert np.allclose(ap.cpu().numpy(), want, rtol=2e-6, atol=1e-9)
        m = float(xr.map_finalize(ap, cap).item())
        assert abs(m - float(np.mean(want / cap.cpu().numpy()))) < MAP_TOL
    assert abs(float(xr.map_k_packed(q, r, ql, rl, C).item()) - orc.map_k_ranked(qB.numpy(), rB.numpy(), qL.numpy(), rL.numpy())) < MAP_TOL


def test_ragged_and_tiny_shapes(xr, cu):
    orc = _orc()
    for (Q, R, K, C) in [(2, 2, 16, 3), (3, 7, 16, 5), (2, 255, 64, 80), (63, 257, 64, 33), (65, 513, 32, 1)]:
        qB, rB, qL, rL = _synth(Q, R, K, C, seed=Q * 1000 + R, structured=False)
        want = orc.map_k(qB, rB, qL, rL)
        got = cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())
        assert (torch.isnan(want) and torch.isnan(got)) or abs(float(got) - float(want)) < MAP_TOL, (Q, R, K, C)
        assert torch.equal(cu.calc_hammingDist(qB.cuda(), rB.cuda()).cpu(), orc.hamming_dist(qB, rB))
        assert torch.equal(cu.calc_hammingDist(qB[0].cuda(), rB.cuda()).cpu(), orc.hamming_dist(qB[0], rB))   # 1-D query


def test_collisions_all_same_distance(cu):
    """every gallery item at the same distance: the order is purely the index tie-break."""
    Q, R, K = 4, 3000, 64
    qB = torch.ones(Q, K)
    rB = torch.ones(R, K)
    gen = torch.Generator().manual_seed(11)
    qL = torch.ones(Q, 2, dtype=torch.int64)
    rL = (torch.rand(R, 2, generator=gen) < 0.3).to(torch.int64)
    rL[0, 0] = 1
    want = _orc().map_k(qB, rB, qL, rL, stable=True)
    assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())) - float(want)) < MAP_TOL


def _ternary_codes(n, K, gen, p_zero=0.15):
    B = torch.randn(n, K, generator=gen).sign()
    B[torch.rand(n, K, generator=gen) < p_zero] = 0.0
    return B


@pytest.mark.parametrize("K", [16, 64, 128, 256])
def test_scan_ternary_codes_all_lengths(cu, K):
    """sign(0) = 0 codes (runners/base.py:410): half-unit distances, 2K+1 buckets -- up to K = 256 (513 buckets)."""
    orc = _orc()
    gen = torch.Generator().manual_seed(900 + K)
    Q, R, C = 37, 2100, 24
    qB, rB = _ternary_codes(Q, K, gen), _ternary_codes(R, K, gen)
    qL = (torch.rand(Q, C, generator=gen) < 0.1).to(torch.int64)
    rL = (torch.rand(R, C, generator=gen) < 0.1).to(torch.int64)
    qL[:, 0] = 1
    rL[::3, 0] = 1
    assert torch.equal(cu.calc_hammingDist(qB.cuda(), rB.cuda()).cpu(), orc.hamming_dist(qB, rB))
    for k in (None, 20):
        want = orc.map_k(qB, rB, qL, rL, k, stable=True)
        assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), k)) - float(want)) < MAP_TOL


@pytest.mark.parametrize("Q,R,K,C", [(50, 5000, 64, 80), (33, 2600, 128, 21), (20, 1500, 256, 24), (70, 4000, 16, 8)])
def test_scan_wide_counters_and_heavy_collisions(xr, cu, monkeypatch, Q, R, K, C):
    """the 64-bit-counter variant (normally only taken when ranks and ordinals do not fit one word) on the same inputs
    as the packed one, plus a gallery of few distinct codes so that the lanes of a query collide on one counter."""
    orc = _orc()
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=3 * K + R)
    rB = rB[torch.randint(0, 5, (R,), generator=torch.Generator().manual_seed(K))]      # 5 distinct gallery codes
    want = orc.map_k(qB, rB, qL, rL, stable=True)
    want7 = orc.map_k(qB, rB, qL, rL, 7, stable=True)
    monkeypatch.setenv("XMH_SCAN_PACK32", "all")                   # packed counters at every length (default: from 65 bits on)
    packed = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()))
    packed7 = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), 7))
    assert abs(packed7 - float(want7)) < MAP_TOL
    monkeypatch.delenv("XMH_SCAN_PACK32")
    assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())) - float(want)) < MAP_TOL      # the default choice
    monkeypatch.setenv("XMH_SCAN_PACK32", "0")
    wide = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()))
    wide7 = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), 7))
    assert abs(packed - float(want)) < MAP_TOL and abs(wide - float(want)) < MAP_TOL and abs(wide7 - float(want7)) < MAP_TOL


def test_scan_masked_fallback_in_a_fresh_process():
    """XMH_SCAN_MASKED=1 (what a failed lane-order probe selects; read once per process) gives the same mAP."""
    import subprocess, sys
    code = (
        "import sys, torch; sys.path[:0] = [%r, %r]\n"
        "from xmh.common import calc_utils as cu\n"
        "g = torch.Generator().manual_seed(5)\n"
        "qB, rB = torch.randn(40, 64, generator=g).sign(), torch.randn(4, 64, generator=g).sign()[torch.randint(0, 4, (3000,), generator=g)]\n"
        "qL, rL = (torch.rand(40, 9, generator=g) < .3).long(), (torch.rand(3000, 9, generator=g) < .3).long()\n"
        "qL[:, 0] = 1; rL[::2, 0] = 1\n"
        "print('MAP %%.12f' %% float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())))\n"
    ) % (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd"))
    outs = []
    for extra in ({}, {"XMH_SCAN_MASKED": "1"}, {"XMH_SCAN_MASKED": "1", "XMH_SCAN_PACK32": "0"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(float([l for l in r.stdout.splitlines() if l.startswith("MAP ")][-1].split()[1]))
    assert abs(outs[0] - outs[1]) < 1e-9 and abs(outs[0] - outs[2]) < 1e-9, outs


def test_scan_masked_self_check_at_full_size(xr, monkeypatch):
    """The fast pass 2 relies on same-address LDS atomics of one instruction resolving in ascending lane order (probed once per
    process); the masked variant does not.  Self-check at the BASELINE configs[1] shape, every chunk and query tile
    under full load: both give the same ap sums and caps bit for bit."""
    Q, R, K, C = 5000, 117218, 64, 80
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=1814, p=0.04)
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    outs = []
    for masked in (False, True):
        if masked:
            monkeypatch.setenv("XMH_SCAN_MASKED", "1")
        scan = xr.RankingScan(q, ql, r, rl, C)
        scan.histograms(False)
        ap, cap = scan.ap_sums(None)
        outs.append((ap.clone(), cap.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_scan_pair_cache_on_and_off_give_identical_bits():
    """The pair cache (pass 1 leaves distance | relevant per pair for pass 2; XMH_SCAN_CACHE_MB, read once per process)
    must not change a single bit: same counter adds in the same order.  Code lengths of both entry widths, ragged
    gallery sizes around the 64-item batch, mAP@all and mAP@k, sparse labels (packed counters) and dense ones."""
    import subprocess, sys
    code = (
        "import sys, torch; sys.path[:0] = [%r, %r]\n"
        "from xmh import retrieval as R\n"
        "from xmh._lib import lib\n"
        "g = torch.Generator().manual_seed(77)\n"
        "for (Q, Rn, K, C, p, k) in ((70, 5000, 64, 12, .3, None), (33, 4097, 48, 40, .02, 7), (129, 6463, 128, 80, .05, None),\n"
        "                            (17, 3000, 256, 9, .3, 50), (64, 64, 96, 5, .5, None), (5, 63, 64, 3, .5, 2), (200, 20001, 160, 20, .01, None)):\n"
        "    qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(37, K, generator=g).sign()[torch.randint(0, 37, (Rn,), generator=g)]\n"
        "    qL, rL = (torch.rand(Q, C, generator=g) < p).long(), (torch.rand(Rn, C, generator=g) < p).long()\n"
        "    qL[:, 0] = 1; rL[::3, 0] = 1\n"
        "    scan = R.RankingScan(R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda()), C)\n"
        "    scan.histograms(False)\n"
        "    a, c = scan.ap_sums(k)\n"
        "    print('ROW', int(lib.xmh_scan_pair_cache_bytes(Q, Rn, K, 0)) > 0, a.cpu().numpy().tobytes().hex(), c.cpu().numpy().tobytes().hex())\n"
    ) % (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd"))
    rows = []
    for extra in ({}, {"XMH_SCAN_CACHE_MB": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        rows.append([l.split()[1:] for l in r.stdout.splitlines() if l.startswith("ROW ")])
    on, off = rows
    assert len(on) == 7 and len(off) == 7
    assert all(x[0] == "True" for x in on) and all(x[0] == "False" for x in off)
    for n, (x, y) in enumerate(zip(on, off)):
        if n in (2, 4):                                        # 65..128 bits (round 4): the one-byte entries are read 4 slots x 16 queries wide, the
            assert x[2] == y[2]                                # uncached kernel runs 8 x 8: another order of the float partial sums inside a chunk
            a, b = np.frombuffer(bytes.fromhex(x[1]), dtype=np.float64), np.frombuffer(bytes.fromhex(y[1]), dtype=np.float64)
            assert np.allclose(a, b, rtol=2e-6, atol=1e-9)
        else:
            assert x[1:] == y[1:]
    # the MFMA-evaluated pass 1 (codes of at most 64 bits) against the VALU one: another chunking, so the per-chunk float sums
    # add in another order -- the caps are equal, the AP sums agree to float rounding
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XMH_SCAN_MFMA="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    valu = [l.split()[1:] for l in r.stdout.splitlines() if l.startswith("ROW ")]
    assert len(valu) == 7
    for x, y in zip(on, valu):
        assert x[2] == y[2]
        a, b = np.frombuffer(bytes.fromhex(x[1]), dtype=np.float64), np.frombuffer(bytes.fromhex(y[1]), dtype=np.float64)
        assert np.allclose(a, b, rtol=2e-6, atol=1e-9)


@pytest.mark.parametrize("Q,R,K,C,m2", [(130, 6000, 64, 80, "1"), (70, 9100, 64, 33, "1"), (300, 20011, 48, 80, "1"), (17, 63, 64, 5, "1"), (129, 6463, 128, 80, "1"),
                                        (200, 7000, 16, 24, "1"), (90, 5001, 32, 80, "1"), (65, 3000, 256, 24, "1"),
                                        (130, 6000, 64, 80, "0"), (300, 20011, 48, 80, "0"), (17, 63, 64, 5, "0"),
                                        (131, 6100, 64, 80, "0"), (300, 20011, 48, 33, "0"), (19, 65, 64, 5, "0"), (201, 7001, 16, 24, "1"), (91, 5002, 32, 80, "1"),
                                        (257, 9000, 33, 128, "1"), (129, 6463, 128, 80, "0"), (70, 9001, 100, 24, "0"), (70, 9001, 100, 24, "1"), (33, 4097, 97, 40, "1"),
                                        (260, 20011, 128, 128, "1"), (150, 7000, 120, 64, "1"), (70, 9001, 96, 24, "1"), (70, 9001, 72, 24, "0"), (65, 3000, 256, 24, "0")])
def test_pair_cache_entries_match_oracle(xr, monkeypatch, Q, R, K, C, m2):
    """Every entry pass 1 leaves in the pair cache (distance << 1 | relevant; xmh_scan_pair_cache_offset documents the layout)
    against the oracle's distance and relevance of that (query, item) pair -- the MFMA-evaluated pass 1 writes the entry from a
    second accumulator chain, so this checks that chain directly and not only through the mAP it leads to.  m2 = "1": the default
    kernels (k_scan_hist_r2 up to 64 bits, k_scan_hist_r2w up to 128 with one-byte entries, k_scan_hist_b beyond); m2 = "0": the VALU
    pass 1 (XMH_SCAN_MFMA=0, what a failed self-check selects), which writes the same entries for 33..64 bits and two-byte entries
    beyond.  (Round 5 removed k_scan_hist_m / k_scan_hist_m2, whose rows these used to be.)"""
    from xmh._lib import lib
    monkeypatch.setenv("XMH_SCAN_MFMA", "0" if m2 == "0" else "1")
    import bench_roofline
    Kc = K                                                             # the code as given (the oracle's side)
    if (K + 31) // 32 == 3:
        K = 128                                                        # the length the kernels see: three-word codes run as 128-bit codes with a zero word (xr.widened)
    if 64 < K <= 128:
        assert ("k_scan_hist_r2w" in bench_roofline.scan_kernels(Q, R, K, C, False)[0]) == (m2 == "1")
    if K <= 64:
        assert ("k_scan_hist_r2" in bench_roofline.scan_kernels(Q, R, K, C, False)[0]) == (m2 == "1")
    if K > 128:
        assert ("k_scan_hist_b" in bench_roofline.scan_kernels(Q, R, K, C, False)[0]) == (m2 == "1")
    orc = _orc()
    qB, rB, qL, rL = _synth(Q, R, Kc, C, seed=3 * Kc + R)
    q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
    scan = xr.RankingScan(q, ql, r, rl, C)
    nbytes = int(lib.xmh_scan_pair_cache_bytes(Q, R, K, 0))
    assert nbytes > 0
    scan.ws.zero_()
    scan.histograms(False)
    off = int(lib.xmh_scan_pair_cache_offset(Q, R, K, 0))
    raw = scan.ws[off:off + nbytes].cpu().numpy()
    dist = orc.hamming_packed(_u32(q.bits), _u32(r.bits)).astype(np.int64)
    rel = orc.relevance_packed(_u32(ql), _u32(rl)).astype(np.int64)
    want = (dist << 1) | rel                                           # [Q, R]
    pl = scan.plan
    nbatch = (pl.chunk + 63) // 64
    b8 = K <= 64 or (K <= 128 and m2 == "1")                           # one-byte entries: up to 64 bits, and 65..128 bits from k_scan_hist_r2w (a distance of 128 wraps)
    S, QW = (4, 16) if b8 else (8, 8)                                  # slots x queries of a cache tile; QW entries per lane and batch
    if b8:
        want = want & 0xFF
        raw = raw[: pl.nchunk * (pl.qpad // QW) * nbatch * 64 * QW]    # 65..128 bits: the region is sized for 12-bit entries, the first two thirds are used
        got = raw.view(np.uint8).reshape(pl.nchunk, pl.qpad // QW, nbatch, 64, QW).astype(np.int64)
    else:
        # round 6: entries above one byte are stored 12 bits each, the 8 of a lane and batch in three dwords (entry k = bits [12 k, 12 k + 12))
        rec = raw[: pl.nchunk * (pl.qpad // QW) * nbatch * 64 * 12].view(np.uint32).reshape(-1, 3).astype(np.uint64)
        bits = rec[:, 0] | (rec[:, 1] << np.uint64(32))                 # entries 0 .. 4 and the low 4 bits of entry 5
        hi = (rec[:, 1] >> np.uint64(28)) | (rec[:, 2] << np.uint64(4))  # from bit 60 on: entries 5 .. 7
        ent = [(bits >> np.uint64(12 * k)) & np.uint64(0xFFF) for k in range(5)] + [(hi >> np.uint64(12 * k)) & np.uint64(0xFFF) for k in range(3)]
        got = np.stack(ent, axis=1).reshape(pl.nchunk, pl.qpad // QW, nbatch, 64, QW).astype(np.int64)
    lane = np.arange(64)
    slot, qin = lane // QW, lane % QW
    t = np.arange(QW)
    checked = 0
    for c in range(pl.nchunk):
        lo, hi = c * pl.chunk, min((c + 1) * pl.chunk, R)
        item = lo + 64 * np.arange(nbatch)[:, None, None] + S * t[None, None, :] + slot[None, :, None]      # [batch, lane, entry]
        ok_item = item < hi
        for tile in range(pl.qpad // QW):
            qq = tile * QW + qin                                        # [lane]
            okq = qq < Q
            if not okq.any():
                continue
            m = ok_item & okq[None, :, None]
            w = want[np.minimum(qq, Q - 1)[None, :, None], np.minimum(item, R - 1)]
            bad = (got[c, tile] != w) & m
            assert not bad.any(), (c, tile, np.argwhere(bad)[:5], got[c, tile][bad][:5], w[bad][:5])
            checked += int(m.sum())
    assert checked == Q * R


@pytest.mark.parametrize("K", [16, 32, 48, 64])
def test_scan_m2_against_the_kernels_it_replaced(xr, monkeypatch, K):
    """XMH_SCAN_MFMA=0 (the VALU kernels: what the per-device self-check falls back to) against the default k_scan_hist_r2 path: the shard
    histograms and the divisors are equal bit for bit; the chunking differs, so the per-chunk float sums of pass 2 add in another order
    and the AP sums agree to float rounding; mAP@all and mAP@k to 1e-7."""
    for (Q, Rn, C, p, k) in ((150, 9001, 80, 0.06, 9), (64, 8157, 32, 0.5, 85), (127, 62, 1, 0.01, None), (300, 20011, 24, 0.1, 50)):
        qB, rB, qL, rL = _synth(Q, Rn, K, C, seed=5 * K + Q, p=p)
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("XMH_SCAN_MFMA", flag)
            scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
            ha, hr = scan.histograms(True)
            ap, cap = scan.ap_sums(None)
            apk, capk = scan.ap_sums(k)
            outs.append((ha.clone(), hr.clone(), cap.clone(), capk.clone(), ap.clone(), apk.clone()))
        for other in outs[1:]:
            for x, y in zip(outs[0][:4], other[:4]):
                assert torch.equal(x, y), (Q, Rn, K, C)
            for x, y, c in ((outs[0][4], other[4], outs[0][2]), (outs[0][5], other[5], outs[0][3])):
                assert torch.allclose(x, y, rtol=2e-6, atol=1e-9), (Q, Rn, K, C)
                assert abs(float((x / c).mean()) - float((y / c).mean())) < 1e-7


def test_scan_m2_self_check_failure_falls_back_with_one_warning():
    """XMH_SCAN_M2_SELFCHECK=2 runs the per-device self-check of k_scan_hist_r2 / r2w and pretends it failed: one line on stderr, and the
    process goes on with the VALU kernels -- same mAP as a process that passed the check."""
    import subprocess, sys
    code = (
        "import sys, torch; sys.path[:0] = [%r, %r]\n"
        "from xmh import retrieval as R\n"
        "from xmh._lib import lib\n"
        "import ctypes\n"
        "g = torch.Generator().manual_seed(5)\n"
        "for K in (64, 16):\n"
        "    qB, rB = torch.randn(90, K, generator=g).sign(), torch.randn(7000, K, generator=g).sign()\n"
        "    qL, rL = (torch.rand(90, 24, generator=g) < .1).long(), (torch.rand(7000, 24, generator=g) < .1).long()\n"
        "    qL[:, 0] = 1; rL[::3, 0] = 1\n"
        "    scan = R.RankingScan(R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda()), 24)\n"
        "    scan.histograms(False)\n"
        "    a, c = scan.ap_sums(None)\n"
        "    buf = ctypes.create_string_buffer(512)\n"
        "    lib.xmh_scan_describe(90, 7000, K, 24, 0, buf, 512)\n"
        "    print('ROW', K, '%%.12f' %% float((a / c).mean()), buf.value.decode())\n"
    ) % (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd"))
    runs = {}
    for mode in ("1", "2"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, XMH_SCAN_M2_SELFCHECK=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        runs[mode] = ([l.split() for l in r.stdout.splitlines() if l.startswith("ROW ")], r.stderr)
    ok, failed = runs["1"], runs["2"]
    assert "self-check FAILED" not in ok[1] and failed[1].count("self-check FAILED") == 1
    assert len(ok[0]) == 2 and len(failed[0]) == 2
    for a, b in zip(ok[0], failed[0]):
        assert abs(float(a[2]) - float(b[2])) < 1e-7                  # another chunking: float partial sums add in another order
        assert "k_scan_hist_r2" in a[3] and "k_scan_hist_r2" not in b[3] and "k_scan_hist_s" in b[3]


@pytest.mark.parametrize("r2w", ["1", "0"])
def test_scan_one_byte_entries_for_65_to_128_bit_codes(xr, cu, monkeypatch, r2w):
    """65..128-bit codes keep ONE byte per pair (k_scan_hist_r2w + k_scan_ap_c<., 8, HALF>) instead of two.  (i) Against the VALU kernels'
    two-byte path (XMH_SCAN_MFMA=0): histograms and divisors bit for bit, AP sums to float rounding (another chunking, other lanes).
    (ii) The one distance a byte cannot hold -- 128, every bit of a 128-bit code differs -- wraps in the cache; pass 1 raises a control
    word and the stand-in kernel evaluates the pairs from the codes: a gallery seeded with the complements of the queries must still give
    the oracle's mAP, at mAP@all and mAP@k, and the float-bit kernel must not have been the one that ran (same sums as with the cache off).
    r2w = "0": part (ii) on the VALU kernels (two-byte entries hold the distance: no wrap, same answers)."""
    orc = _orc()
    # (97..128 bits: four code words.  65..96 bits are three words, which the ranking kernels take as a 128-bit code with a zero plane --
    # the ternary kernels, no pair cache)
    for (Q, Rn, K, C, p, k) in ((129, 6463, 128, 80, .05, None), (70, 9001, 100, 24, .1, 50), (33, 4097, 97, 40, .02, 7), (200, 20011, 128, 128, .02, None)):
        qB, rB, qL, rL = _synth(Q, Rn, K, C, seed=9 * K + Rn, p=p)
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("XMH_SCAN_MFMA", flag)
            scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
            ha, hr = scan.histograms(True)
            ap, cap = scan.ap_sums(k)
            outs.append((ha.clone(), hr.clone(), cap.clone(), ap.clone()))
        monkeypatch.delenv("XMH_SCAN_MFMA")
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(x, y), (Q, Rn, K, C)
        assert torch.allclose(outs[0][3], outs[1][3], rtol=2e-6, atol=1e-9), (Q, Rn, K, C)      # another chunking, 4 x 16 against 8 x 8 lanes: float sums in another order
    monkeypatch.setenv("XMH_SCAN_MFMA", r2w)
    # (ii) complements in the gallery: distance 128
    Q, Rn, K, C = 90, 7000, 128, 24
    qB, rB, qL, rL = _synth(Q, Rn, K, C, seed=4242, p=0.1)
    rB[5::97] = -qB[torch.arange(rB[5::97].shape[0]) % Q]                                  # exact complements of some queries, spread over the chunks
    rL[5::97] = qL[torch.arange(rL[5::97].shape[0]) % Q]                                   # and relevant to them: their credit depends on bucket 128
    for k in (None, 40):
        got = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), k))
        want = float(orc.map_k(qB, rB, qL, rL, k, stable=True))
        assert abs(got - want) < 1e-6, (k, got, want)
    scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
    ha, _ = scan.histograms(True)
    assert int(ha[:, 128].sum()) >= 70                                                     # the complements sit in bucket 128
    ap_wrapped, cap_w = scan.ap_sums(None)
    monkeypatch.setenv("XMH_SCAN_CACHE_MB", "0")
    plain = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
    plain.histograms(False)
    ap_plain, cap_p = plain.ap_sums(None)
    monkeypatch.delenv("XMH_SCAN_CACHE_MB")
    assert torch.equal(cap_w, cap_p) and torch.allclose(ap_wrapped, ap_plain, rtol=2e-6, atol=1e-9)
    dist = orc.hamming_packed(_u32(scan.q.bits), _u32(scan.r.bits))
    rel = orc.relevance_packed(_u32(scan.qlab), _u32(scan.rlab))
    assert np.allclose(ap_wrapped.cpu().numpy(), orc.ap_from_ranking(dist, rel), rtol=3e-6)


def test_scan_pass2_eight_queries_wide_on_one_byte_entries(xr, monkeypatch):
    """k_scan_ap_c<., 8, HALF>: the one-byte pair cache read 8 slots x 8 queries wide (65..128 bits) and 4 x 16 wide (shorter codes) against
    the integer-counter kernels on the same evaluation (XMH_SCAN_AP_C=0): divisors bit for bit, AP sums to float rounding (other lanes, so a
    chunk's partial sums add in another order) and equal to the oracle's ranking; ragged last batches, surplus query columns, capped and not."""
    orc = _orc()
    for (Q, Rn, K, C, p, k) in ((129, 6463, 128, 80, .05, None), (70, 9001, 100, 24, .1, 50), (150, 9100, 64, 80, .06, None), (37, 2501, 40, 11, .2, 9),
                                (200, 20011, 16, 24, .1, None), (9, 70, 128, 5, .3, 3)):
        qB, rB, qL, rL = _synth(Q, Rn, K, C, seed=3 * K + Rn, p=p)
        outs = []
        for flag in ("1", "0"):
            monkeypatch.setenv("XMH_SCAN_AP_C", flag)
            scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
            scan.histograms(False)
            m, ap, cap = scan.map_all(k)
            outs.append((cap.clone(), ap.clone(), float(m)))
        monkeypatch.delenv("XMH_SCAN_AP_C")
        assert torch.equal(outs[0][0], outs[1][0]), (Q, Rn, K, C)
        assert torch.allclose(outs[0][1], outs[1][1], rtol=2e-6, atol=1e-9), (Q, Rn, K, C)
        want = float(orc.map_k(qB, rB, qL, rL, k, stable=True))
        assert abs(outs[0][2] - want) < 1e-6 and abs(outs[1][2] - want) < 1e-6, (Q, Rn, K, C, outs[0][2], outs[1][2], want)


@pytest.mark.parametrize("K,env", [(64, {}), (64, {"XMH_SCAN_AP_R2": "1"}), (128, {}), (100, {"XMH_SCAN_MFMA": "0"}),
                                   (128, {"XMH_SCAN_AP_C": "0"}), (16, {}), (256, {})])
def test_scan_repeated_evaluations_are_bit_identical(xr, monkeypatch, K, env):
    """The same scan evaluated 25 times gives the same bits 25 times: histograms, divisors and AP sums.  Round 4: in the one-group-per-batch
    variants of k_scan_ap_c hipcc had copied atomic results in front of their s_waitcnt on the path of a chunk with exactly one whole batch
    (the shape below: chunks of 256 items, a last chunk of 108), and every evaluation of such a shape differed from the one before in a few
    queries -- within no tolerance.  The build-time ISA check (tests/test_isa_hazards.py, rule R5) is the gate; this is the symptom."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    Q, R, C = 513, 3180, 5
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=911, p=0.1)
    scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
    assert scan.plan.chunk * (scan.plan.nchunk - 1) < R and (R - scan.plan.chunk * (scan.plan.nchunk - 1)) % 64 != 0      # a ragged last chunk
    first = None
    for i in range(25):
        ha, hr = scan.histograms(True)
        m, ap, cap = scan.map_all(None if i % 2 else 50)
        m2, ap2, cap2 = scan.map_all(None)
        cur = (ha.clone(), hr.clone(), cap2.clone(), ap2.clone(), float(m2))
        if first is None:
            first = cur
            orc = _orc()
            assert abs(cur[4] - float(orc.map_k(qB, rB, qL, rL, None, stable=True))) < 1e-6
        else:
            assert all(torch.equal(x, y) for x, y in zip(cur[:4], first[:4])) and cur[4] == first[4], i


def test_scan_float_bit_counters_give_identical_bits(xr, monkeypatch):
    """k_scan_ap_c (pass 2 with float-bit counters: the default on one-byte entries of at most 64 bits) against the cached k_scan_ap_s on the
    same pair cache (XMH_SCAN_AP_C=0: the path of galleries beyond 2^23 items): the same credits in the same order, so the per-query
    sums and caps are equal bit for bit -- mAP@all and mAP@k, ragged last batches, duplicate-heavy galleries.  At 128 bits the entries are
    one byte wide and only k_scan_ap_c reads them (the switch then selects the kernel that evaluates the pairs from the codes: another lane
    geometry, sums equal to float rounding); at 256 bits both settings run the cached k_scan_ap_s."""
    g = torch.Generator().manual_seed(31)
    for (Q, Rn, K, C, p, k) in ((70, 5000, 64, 12, .3, None), (33, 4097, 48, 40, .02, 7), (129, 6463, 128, 80, .05, None), (17, 3000, 256, 9, .3, 50),
                                (200, 9001, 16, 24, .1, None), (5, 63, 32, 3, .5, 2)):
        qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(37, K, generator=g).sign()[torch.randint(0, 37, (Rn,), generator=g)]
        qL, rL = (torch.rand(Q, C, generator=g) < p).long(), (torch.rand(Rn, C, generator=g) < p).long()
        qL[:, 0] = 1
        rL[::3, 0] = 1
        scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
        scan.histograms(False)
        outs = []
        for mode in ("0", "1"):
            monkeypatch.setenv("XMH_SCAN_AP_C", mode)
            ap, cap = scan.ap_sums(k)
            outs.append((ap.clone(), cap.clone()))
        monkeypatch.delenv("XMH_SCAN_AP_C")
        assert torch.equal(outs[0][1], outs[1][1]), (Q, Rn, K)
        if 64 < K <= 128:
            assert torch.allclose(outs[0][0], outs[1][0], rtol=2e-6, atol=1e-9), (Q, Rn, K)
        else:
            assert torch.equal(outs[0][0], outs[1][0]), (Q, Rn, K)


def test_scan_pass2_without_a_pair_cache_gives_identical_bits_and_matches_the_oracle(xr, monkeypatch):
    """k_scan_ap_r2 (round 5): pass 2 evaluates the pairs again on the MFMA from the packed words (XMH_SCAN_AP_R2=1 drops the pair cache; it is
    also what runs whenever a shape of at most 64 bits has no cache).  Same plan, same chunks, k_scan_ap_c's credits in k_scan_ap_c's order
    per lane: AP sums, caps and capped sums bit for bit against the cached path; mAP against the oracle (common/calc_utils.py:58-92, stable
    order).  One / two label tiles, one code word and two, a partial last word, ragged last batches, tiny galleries, surplus query rows."""
    import ctypes as C_
    from oracle import retrieval as orc
    from xmh import _lib
    for (Q, R, K, C, p, k) in ((150, 9001, 64, 80, 0.06, 9), (16, 2, 40, 64, 0.5, 1), (64, 8157, 48, 32, 0.5, 85), (127, 62, 64, 1, 0.01, 199),
                               (300, 20011, 16, 24, 0.1, 50), (257, 4096, 33, 128, 0.05, 7), (90, 5001, 32, 80, 0.04, None)):
        qB, rB, qL, rL = _synth(Q, R, K, C, seed=5 + K + R, p=p)
        qL[:, 0] = 1
        rL[::5, 0] = 1
        outs = []
        for flag in ("0", "1"):
            monkeypatch.setenv("XMH_SCAN_AP_R2", flag)
            assert (int(_lib.lib.xmh_scan_pair_cache_bytes(Q, R, K, 0)) > 0) == (flag == "0")
            scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
            scan.histograms(False)
            ap, cap = scan.ap_sums(None)
            apk, capk = scan.ap_sums(k)
            outs.append((ap.clone(), cap.clone(), apk.clone(), capk.clone()))
            if flag == "1":
                buf = C_.create_string_buffer(512)
                _lib.check(_lib.lib.xmh_scan_describe(Q, R, K, C, 0, buf, 512), "xmh_scan_describe")
                assert "pass2=k_scan_ap_r2<%d, 4, 2, false>" % (1 if C <= 64 else 2) in buf.value.decode(), buf.value
                m = float(scan.map_all(k)[0].item())
                want = float(orc.map_k(qB, rB, qL, rL, k, stable=True))
                assert abs(m - want) < 2e-6, (Q, R, K, C, m, want)
        for x, y in zip(*outs):
            assert torch.equal(x, y), (Q, R, K, C)


def test_scan_workspace_without_room_for_the_pair_cache_runs_uncached(xr):
    """A workspace of xmh_scan_ws_bytes_nocache bytes (a device without room for the Q x R cache next to a resident encoder) runs the same
    plan without the cache -- decided per call from the size handed in, no environment write.  Up to 64 bits the uncached pass 2
    is k_scan_ap_r2: AP sums and caps bit for bit; 128 / 256 bits: the kernels that evaluate the pairs from the codes, sums to float
    rounding.  A buffer smaller than that is refused."""
    from xmh import _lib
    for (Q, R, K, C, k) in ((150, 9001, 64, 80, None), (70, 5000, 16, 24, 9), (129, 6463, 128, 80, None), (40, 3000, 256, 24, 5)):
        qB, rB, qL, rL = _synth(Q, R, K, C, seed=K + R, p=0.08)
        qL[:, 0] = 1
        rL[::5, 0] = 1
        full = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
        full.histograms(False)
        ap0, cap0 = full.ap_sums(k)
        small = int(_lib.lib.xmh_scan_ws_bytes_nocache(Q, R, K, 0))
        assert 0 < small < full.plan.ws_bytes and int(_lib.lib.xmh_scan_pair_cache_bytes(Q, R, K, 0)) > 0
        lean = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
        lean.ws = torch.empty(small, dtype=torch.uint8, device="cuda")
        ha, hr = lean.histograms(True)
        ap1, cap1 = lean.ap_sums(k)
        assert torch.equal(cap0, cap1), (Q, R, K)
        if K <= 64:
            assert torch.equal(ap0, ap1), (Q, R, K)
        else:
            assert torch.allclose(ap0, ap1, rtol=2e-6, atol=1e-9), (Q, R, K)
        lean.ws = torch.empty(small - 4096, dtype=torch.uint8, device="cuda")
        with pytest.raises(RuntimeError, match="workspace too small"):
            lean.histograms(False)


def test_scan_mfma_pass1_for_65_to_256_bit_codes_matches_the_valu_pass1(xr, monkeypatch):
    """65..256-bit codes: the MFMA pass 1 (k_scan_hist_r2w up to 128 bits, k_scan_hist_b beyond) against the VALU pass 1
    (XMH_SCAN_MFMA=0): same histograms and caps bit for bit, same credits up to the order of the per-chunk partial sums (the two plans
    cut the gallery into different chunks)."""
    for (Q, R, K, C, p, k) in ((150, 9001, 128, 80, 0.06, 9), (17, 130, 128, 5, 0.3, 3), (64, 8157, 96, 33, 0.2, 85), (33, 4096, 65, 128, 0.05, None),
                               (70, 9100, 256, 80, 0.06, None), (20, 700, 160, 24, 0.2, 11)):      # 129..256 bits: four code tiles, one block per CU
        qB, rB, qL, rL = _synth(Q, R, K, C, seed=K + R, p=p)
        outs = []
        for flag in ("0", "1"):
            monkeypatch.setenv("XMH_SCAN_MFMA", flag)                  # 0: the VALU kernels
            q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
            scan = xr.RankingScan(q, xr.pack_labels(qL.cuda()), r, xr.pack_labels(rL.cuda()), C)
            ha, hr = scan.histograms(True)
            ap, cap = scan.ap_sums(k)
            outs.append((ha.clone(), hr.clone(), cap.clone(), ap.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), (Q, R, K, C)
        assert torch.allclose(outs[0][3], outs[1][3], rtol=1e-6, atol=1e-9), (Q, R, K, C)


def test_scan_pass1_with_operands_from_the_packed_bits_gives_identical_bits(xr, monkeypatch):
    """k_scan_hist_b (xmh_scan_bits.hip: 129..256 bits, operands built in registers from the packed bits) against the VALU pass 1
    (XMH_SCAN_MFMA=0), which writes the same two-byte pair-cache entries for the same cached pass 2: histograms and caps bit for bit, the
    credits up to the order of the per-chunk partial sums (the plans chunk differently); ragged chunks, surplus query columns, code lengths
    that do not fill their last word, 1..128 classes; and both against the oracle's ranking.  (Round 5 removed k_scan_hist_m, on whose
    plan this used to be a bit-for-bit comparison.)"""
    orc = _orc()
    for (Q, R, K, C, p, k) in ((150, 9001, 256, 80, 0.06, 9), (17, 130, 256, 5, 0.3, 3), (64, 8157, 200, 33, 0.2, 85), (33, 4096, 129, 128, 0.05, None),
                               (70, 9100, 192, 80, 0.06, None), (20, 700, 160, 24, 0.2, 11), (5, 70000, 130, 1, 0.5, 100), (130, 20011, 224, 97, 0.03, None)):
        qB, rB, qL, rL = _synth(Q, R, K, C, seed=K + R, p=p)
        qL[:, 0] = 1
        rL[::7, 0] = 1
        outs = []
        for flag in ("0", "1"):
            monkeypatch.setenv("XMH_SCAN_MFMA", flag)
            q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
            scan = xr.RankingScan(q, xr.pack_labels(qL.cuda()), r, xr.pack_labels(rL.cuda()), C)
            import bench_roofline
            assert ("k_scan_hist_b" in bench_roofline.scan_kernels(Q, R, scan.q.K, C, False)[0]) == (flag == "1")
            ha, hr = scan.histograms(True)
            ap, cap = scan.ap_sums(k)
            outs.append((ha.clone(), hr.clone(), cap.clone(), ap.clone()))
        for x, y in zip(outs[0][:3], outs[1][:3]):
            assert torch.equal(x, y), (Q, R, K, C)
        assert torch.allclose(outs[0][3], outs[1][3], rtol=2e-6, atol=1e-9), (Q, R, K, C)
        m = float((outs[1][3] / outs[1][2]).mean())
        assert abs(m - float(orc.map_k(qB, rB, qL, rL, k, stable=True))) < 2e-6, (Q, R, K, C)


def test_scan_many_evaluations_after_one_histogram_pass(xr):
    """One xmh_hamming_hist, then several evaluations with different k on the same workspace: the offsets pass 1 left behind are
    reused, the finalize ticket puts itself back to zero (a stale ticket would leave map_out unwritten), and a sharded-style
    call with explicit offsets in between is followed by a fresh histogram pass as the header asks."""
    orc = _orc()
    Q, R, K, C = 130, 7001, 64, 20
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=5)
    scan = xr.RankingScan(xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda()), xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda()), C)
    scan.histograms(False)
    want = {k: float(orc.map_k(qB, rB, qL, rL, k=k, stable=True)) for k in (None, 7, 500)}
    for k in (None, 7, None, 500, 7):
        m, ap, cap = scan.map_all(k)
        assert abs(float(m) - want[k]) < MAP_TOL, k
        ap2, cap2 = scan.ap_sums(k)                                  # xmh_hamming_ap (no finalize) on the same tables
        assert torch.equal(ap, ap2) and torch.equal(cap, cap2)


def test_calc_map_k_label_cache_sees_in_place_edits(cu):
    orc = _orc()
    qB, rB, qL, rL = _synth(12, 900, 64, 10, seed=4)
    a = float(cu.calc_map_k(qB, rB, qL, rL))
    assert abs(a - float(orc.map_k(qB, rB, qL, rL, stable=True))) < MAP_TOL
    assert abs(float(cu.calc_map_k(qB, rB, qL, rL)) - a) < 1e-12             # second call: packed labels come from the cache
    rL[: 450] = 0
    rL[: 450, 3] = 1                                                          # in-place edit bumps the tensor version
    b = float(cu.calc_map_k(qB, rB, qL, rL))
    assert abs(b - float(orc.map_k(qB, rB, qL, rL, stable=True))) < MAP_TOL and abs(a - b) > 1e-4


def test_scan_fuzz_shapes_lengths_and_caps(cu, monkeypatch):
    """40 seeded random configurations: every code length the kernels or the widening cover, ragged Q / R around the tile and
    batch sizes, binary and ternary codes, duplicated gallery codes, mAP@all and mAP@k, both counter widths."""
    orc = _orc()
    rng = np.random.default_rng(2024)
    lengths = [8, 16, 24, 32, 48, 64, 96, 128, 160, 256, 512, 1024]
    for case in range(40):
        K = int(rng.choice(lengths))
        Q = int(rng.choice([2, 3, 15, 16, 17, 63, 64, 65, 100, 129]))
        R = int(rng.choice([2, 63, 64, 65, 127, 255, 256, 257, 1000, 2049, 4100]))
        C = int(rng.choice([1, 7, 24, 33, 80, 100]))
        gen = torch.Generator().manual_seed(1000 + case)
        qB, rB = torch.randn(Q, K, generator=gen).sign(), torch.randn(R, K, generator=gen).sign()
        if case % 3 == 0 and R > 8:
            rB = rB[torch.randint(0, 6, (R,), generator=gen)]                 # heavy ties
        ternary = case % 4 == 1 and K <= 256
        if ternary:
            qB[torch.rand(Q, K, generator=gen) < 0.1] = 0.0
            rB[torch.rand(R, K, generator=gen) < 0.1] = 0.0
        qL, rL = (torch.rand(Q, C, generator=gen) < 0.3).long(), (torch.rand(R, C, generator=gen) < 0.3).long()
        qL[:, 0] = 1
        rL[0, 0] = 1
        k = None if case % 2 else int(rng.integers(1, 40))
        monkeypatch.delenv("XMH_SCAN_PACK32", raising=False)
        if case % 5 == 4:
            monkeypatch.setenv("XMH_SCAN_PACK32", "0")
        elif case % 5 in (1, 2):
            monkeypatch.setenv("XMH_SCAN_PACK32", "all")             # the packed kernels at 64 bits and less (default: from 65 on)
        want = float(orc.map_k(qB, rB, qL, rL, k, stable=True))
        got = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), k))
        assert abs(got - want) < MAP_TOL, (case, Q, R, K, C, k, ternary)


def test_sharded_offsets_reproduce_unsharded(xr):
    """contiguous gallery shards + bucket offsets from the exchanged histograms == one gallery."""
    Q, R, K, C, S = 96, 7000, 64, 80, 3
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=77)
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    whole = xr.RankingScan(q, ql, r, rl, C)
    whole.histograms(False)
    ap_ref, cap_ref = whole.ap_sums(50)
    bounds = [0, 2100, 2100 + 3333, R]
    scans, hall, hrel = [], [], []
    for s in range(S):
        sc = xr.RankingScan(q, ql, r.rows(bounds[s], bounds[s + 1]), rl[bounds[s]:bounds[s + 1]].contiguous(), C)
        a, b = sc.histograms()
        scans.append(sc), hall.append(a.to(torch.int64)), hrel.append(b.to(torch.int64))
    tot_a, tot_r = sum(hall), sum(hrel)
    lower_a = torch.cumsum(tot_a, 1) - tot_a                       # everything in lower buckets, any shard
    lower_r = torch.cumsum(tot_r, 1) - tot_r
    nrel = tot_r.sum(1).to(torch.int32)
    ap = torch.zeros(Q, dtype=torch.float64, device="cuda")
    gathered = torch.stack([torch.stack([a, b]) for a, b in zip(hall, hrel)]).to(torch.int32).contiguous()    # [S, 2, Q, nb]
    for s in range(S):
        base_a = (lower_a + sum(hall[:s], torch.zeros_like(tot_a))).to(torch.int32).contiguous()
        base_r = (lower_r + sum(hrel[:s], torch.zeros_like(tot_r))).to(torch.int32).contiguous()
        ka, kr, kn = xr.shard_offsets(gathered, s)               # the one-kernel form the sharded driver uses
        assert torch.equal(ka, base_a) and torch.equal(kr, base_r) and torch.equal(kn, nrel)
        part, cap = scans[s].ap_sums(50, base_a, base_r, nrel)
        assert torch.equal(cap, cap_ref)
        ap += part
    assert torch.allclose(ap, ap_ref, rtol=1e-6, atol=1e-9)


def test_sharded_fuzz_random_splits(xr):
    """12 seeded random (Q, R, K, world, shard bounds, k): gallery split into contiguous shards of very different sizes, offsets
    from the one-kernel form, partial AP sums added up == the unsharded scan (bit-exact caps, 1e-6 sums)."""
    rng = np.random.default_rng(31)
    for case in range(12):
        K = int(rng.choice([16, 64, 128, 256, 512]))              # 256: pass 1 with the operands built from the packed bits
        Q, R, C = int(rng.choice([5, 64, 97])), int(rng.choice([300, 2000, 5003])), int(rng.choice([3, 24, 80]))
        world = int(rng.integers(2, 6))
        cuts = np.sort(rng.choice(np.arange(1, R), size=world - 1, replace=False))
        bounds = [0] + cuts.tolist() + [R]
        k = None if case % 2 else int(rng.integers(1, 60))
        qB, rB, qL, rL = _synth(Q, R, K, C, seed=700 + case)
        if case % 3 == 0:
            rB = rB[torch.randint(0, 9, (R,), generator=torch.Generator().manual_seed(case))]
        q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
        r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
        whole = xr.RankingScan(q, ql, r, rl, C)
        whole.histograms(False)
        ap_ref, cap_ref = whole.ap_sums(k)
        scans = [xr.RankingScan(q, ql, r.rows(bounds[s], bounds[s + 1]), rl[bounds[s]:bounds[s + 1]].contiguous(), C) for s in range(world)]
        gathered = torch.stack([torch.stack(sc.histograms()) for sc in scans]).contiguous()
        ap = torch.zeros(Q, dtype=torch.float64, device="cuda")
        for s in range(world):
            part, cap = scans[s].ap_sums(k, *xr.shard_offsets(gathered, s))
            assert torch.equal(cap, cap_ref), case
            ap += part
        assert torch.allclose(ap, ap_ref, rtol=1e-6, atol=1e-9), (case, Q, R, K, world, bounds)


@pytest.mark.parametrize("Q,R,K,C,world,k", [(300, 9001, 64, 80, 4, None), (130, 5000, 16, 24, 2, 9), (200, 7000, 128, 40, 8, None), (150, 6001, 256, 80, 3, 50),
                                             (100, 3, 64, 10, 4, None)])
def test_sharded_all_to_all_exchange_by_query_slice(xr, Q, R, K, C, world, k):
    """The all-to-all form of the sharded evaluation with the `world` ranks played by one process: every shard's totals table cut
    into query slices, the slice owner's kernel (xmh_shard_slice_offsets) against its torch statement, the rows handed back to
    the shards, and the shards' shares (xmh_hamming_map_sharded_offsets) adding up to the unsharded mAP and to the all-gather
    form (xmh_hamming_map_sharded).  (100, 3, ..., 4): more ranks than gallery rows -- one shard is empty."""
    from xmh import sharded
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=Q + R + K)
    qL[:, 0] = 1
    rL[::3, 0] = 1                                                                    # every query has a relevant item (else the mAP is NaN)
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    whole = xr.RankingScan(q, ql, r, rl, C)
    whole.histograms(False)
    want = float(whole.map_all(k)[0].item())
    assert want == want
    b = sharded.shard_bounds(R, world)
    shards = [sharded.HipShardOps(q, ql, r.rows(b[i], b[i + 1]), rl[b[i]:b[i + 1]].contiguous(), C) for i in range(world)]
    tables = [o.totals().clone() for o in shards]                                   # [nb, qpad, 2] each
    nb, qpad = tables[0].shape[0], tables[0].shape[1]
    assert qpad % world == 0
    S = qpad // world
    sends = [t.view(nb, world, S, 2).permute(1, 0, 2, 3).contiguous() for t in tables]           # [owner, nb, S, 2] per shard
    offs = []
    for j in range(world):                                                            # slice owner j
        recv = torch.stack([sends[w][j] for w in range(world)]).contiguous()         # what all_to_all_single delivers
        o = xr.slice_offsets(recv)
        assert torch.equal(o, sharded.slice_offsets_reference(recv))
        offs.append(o)
    gathered = torch.stack(tables).contiguous()
    got = got_gather = 0.0
    for w in range(world):                                                            # shard w
        back = torch.stack([offs[j][w] for j in range(world)]).contiguous()          # its rows, by slice owner
        got += float(shards[w].map_partial_offsets(k, back).item())
        got_gather += float(shards[w].map_partial(k, gathered, w).item())
    assert abs(got - want) < 1e-7 and abs(got - got_gather) < 1e-12          # vs unsharded: fp32 credits summed per chunk, another chunking


def test_sharded_driver_over_rccl_world_one_plain_and_query_blocks():
    """xmh.sharded.map_k_sharded through a real RCCL process group (world size 1, fresh process): the one-shot form and the
    query-block pipeline (asynchronous gathers) against the unsharded scan."""
    import subprocess, sys
    code = (
        "import os, sys, torch; sys.path[:0] = [%r, %r]\n"
        "import torch.distributed as dist\n"
        "import socket\n"
        "with socket.socket() as _s:\n"
        "    _s.bind(('127.0.0.1', 0)); _port = _s.getsockname()[1]\n"
        "os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(_port)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "from xmh import retrieval as R, sharded\n"
        "g = torch.Generator().manual_seed(9)\n"
        "for (Q, Rn, K, C, k) in ((150, 9000, 64, 30, None), (37, 2500, 128, 80, 11), (64, 1000, 16, 5, None)):\n"
        "    qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(50, K, generator=g).sign()[torch.randint(0, 50, (Rn,), generator=g)]\n"
        "    qL, rL = (torch.rand(Q, C, generator=g) < .2).long(), (torch.rand(Rn, C, generator=g) < .2).long()\n"
        "    qL[:, 0] = 1; rL[::4, 0] = 1\n"
        "    q, ql, r, rl = R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda())\n"
        "    whole = R.RankingScan(q, ql, r, rl, C); whole.histograms(False)\n"
        "    m0, a0, c0 = whole.map_all(k)\n"
        "    m1, a1, c1 = sharded.map_k_sharded(sharded.HipShardOps(q, ql, r, rl, C), k)\n"
        "    assert torch.equal(c0, c1) and torch.allclose(a0, a1, rtol=1e-12) and abs(float(m0) - float(m1)) < 1e-12\n"
        "    m3, a3, c3 = sharded.map_k_sharded(sharded.HipShardOps(q, ql, r, rl, C), k, map_only=True)\n"
        "    assert a3 is None and abs(float(m0) - float(m3)) < 1e-12\n"
        "    for nb in (2, 3, 200):\n"
        "        m2, a2, c2 = sharded.map_k_sharded(sharded.QueryBlocks.split(q, ql, r, rl, C, nb), k)\n"
        "        assert torch.equal(c0, c2) and torch.allclose(a0, a2, rtol=1e-9) and abs(float(m0) - float(m2)) < 1e-9, (Q, K, nb)\n"
        "torch.cuda.synchronize(); dist.destroy_process_group(); print('rccl ok')\n"
    ) % (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def _two_rank_rccl_worker(rank, world, port, tmp, share_gpu=False):
    import sys
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    if share_gpu:                                                                    # both ranks on cuda:0, gloo carries the device tensors
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
            if p not in sys.path:
                sys.path.insert(0, p)
        from xmh import retrieval as R, sharded
        g = torch.Generator().manual_seed(21)                                        # identical data on every rank
        Q, Rn, K, C = 300, 20001, 64, 40
        qB, rB = torch.randn(Q, K, generator=g).sign(), torch.randn(80, K, generator=g).sign()[torch.randint(0, 80, (Rn,), generator=g)]
        qL, rL = (torch.rand(Q, C, generator=g) < .1).long(), (torch.rand(Rn, C, generator=g) < .1).long()
        qL[:, 0] = 1
        rL[::4, 0] = 1
        q, ql, r, rl = R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda())
        whole = R.RankingScan(q, ql, r, rl, C)
        whole.histograms(False)
        b = sharded.shard_bounds(Rn, world)
        ops = sharded.HipShardOps(q, ql, r.rows(b[rank], b[rank + 1]), rl[b[rank]:b[rank + 1]].contiguous(), C)
        for k in (None, 7):
            want = float(whole.map_all(k)[0].item())
            for ex in ("alltoall", "gather"):
                m = sharded.map_k_sharded(ops, k, map_only=True, exchange=ex)[0]
                assert abs(float(m.item()) - want) < 1e-9, (k, ex)
            m, ap, cap = sharded.map_k_sharded(ops, k)
            assert abs(float(m.item()) - want) < 1e-9
        d, i = sharded.topk_sharded(q, r.rows(b[rank], b[rank + 1]), 30, b[rank])
        wd, wi = R.hamming_topk(q, r, 30)
        assert torch.equal(i, wi.cpu()) and torch.equal(d.to(torch.int16), wd.cpu())
        open(os.path.join(tmp, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_sharded_two_ranks_over_rccl(tmp_path):
    """the CUDA branches of the sharded driver (all_to_all_single / all_gather_into_tensor on workspace views, the
    fused map_only forms, topk_sharded) under a REAL 2-rank RCCL group, against the unsharded scan.  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    mp.spawn(_two_rank_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(2))


def test_sharded_two_ranks_sharing_the_gpu(tmp_path):
    """The same worker on the ONE GPU of this box: two processes, both on cuda:0, the group over gloo (RCCL refuses two ranks on one
    device; gloo takes device tensors for every collective the driver uses).  What runs with world = 2 on hardware this way: HipShardOps
    on a real half gallery each, all_to_all_single on the workspace views, all_gather_into_tensor of the totals tables, the all-reduces,
    topk_sharded with its gather into one tensor and pinned D2H -- everything of the N > 1 path but RCCL's transport (VERDICT r4 missing 1)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    mp.spawn(_two_rank_rccl_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(2))


def test_sharded_ops_with_an_empty_shard(xr):
    """a rank without gallery rows (fewer rows than ranks) takes part in the exchange with zeros; ternary codes included so the
    bucket count of the empty rank has to match the others'."""
    from xmh import sharded
    Q, R, K, C = 20, 50, 64, 7
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=5)
    qB[:, ::9] = 0.0
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    whole = xr.RankingScan(q, ql, r, rl, C)
    whole.histograms(False)
    ap_ref, cap_ref = whole.ap_sums(9)
    shards = [sharded.HipShardOps(q, ql, r.rows(0, 30), rl[:30].contiguous(), C), sharded.HipShardOps(q, ql, r.rows(30, 30), rl[30:30].contiguous(), C),
              sharded.HipShardOps(q, ql, r.rows(30, R), rl[30:].contiguous(), C)]
    assert shards[1].empty
    gathered = torch.stack([torch.stack(o.histograms()) for o in shards]).contiguous()
    ap = torch.zeros(Q, dtype=torch.float64, device="cuda")
    for s, o in enumerate(shards):
        part, cap = o.ap_sums(9, *o.offsets(gathered, s))
        assert torch.equal(cap, cap_ref)
        ap += part
    assert torch.allclose(ap, ap_ref, rtol=1e-6, atol=1e-9)
    # the fused form (offsets + pass 2 + this shard's share of the mean in one call): the shares add up to the mean
    want = float((ap / cap_ref.double()).mean())                # from the same per-shard credits (fp32 credits: the unsharded
    tg = torch.stack([o.totals() for o in shards]).contiguous()                  # scan adds them in another order, 1e-6)
    assert torch.equal(tg[:, :, :Q, 0].transpose(1, 2), gathered[:, 0]) and torch.equal(tg[:, :, :Q, 1].transpose(1, 2), gathered[:, 1])
    shares = [o.map_partial(9, tg, s) for s, o in enumerate(shards)]
    got = sum(float(x) for x in shares)
    assert abs(got - want) < 1e-12 and abs(got - float((ap_ref / cap_ref.double()).mean())) < 1e-6
    for s, o in enumerate(shards):
        if not o.empty:
            m_s, ap_s, cap_s = o.scan.map_sharded(9, tg, s)
            part, cap = o.ap_sums(9, *o.offsets(gathered, s))
            assert torch.equal(cap_s, cap_ref) and torch.equal(ap_s, part)


def test_full_size_properties_coco_shape(xr):
    """BASELINE configs[1] shape (Q 5000 x R 117218, 64 bit, 80 classes): properties that do not need the
    [Q,R] matrix, plus exact agreement with the oracle on a query subsample."""
    orc = _orc()
    Q, R, K, C = 5000, 117218, 64, 80
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=1814, p=0.04)
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    scan = xr.RankingScan(q, ql, r, rl, C)
    ha, hr = scan.histograms()
    assert (ha.to(torch.int64).sum(1) == R).all()                                     # every item lands in one bucket
    ap, cap = scan.ap_sums(None)
    assert torch.equal(hr.to(torch.int64).sum(1).to(torch.int32), cap)               # relevant counts agree
    apq = (ap / cap.double()).cpu().numpy()
    assert np.all(apq > 0) and np.all(apq <= 1.0 + 1e-9)
    sub = np.arange(0, Q, 97)[:48]
    dist = orc.hamming_packed(_u32(q.bits)[sub], _u32(r.bits))
    rel = orc.relevance_packed(_u32(ql)[sub], _u32(rl))
    want = orc.ap_from_ranking(dist, rel)
    assert np.array_equal(cap.cpu().numpy()[sub], rel.sum(-1))
    assert np.allclose(ap.cpu().numpy()[sub], want, rtol=3e-6)
    # idempotence / determinism: a second run gives bit-identical sums
    scan.histograms(False)
    ap2, _ = scan.ap_sums(None)
    assert torch.equal(ap, ap2)
    # identical gallery appended twice: every item's bucket count doubles
    r2 = xr.PackedCodes(torch.cat([r.bits, r.bits]), None, K)
    ha2, _ = xr.RankingScan(q, ql, r2, torch.cat([rl, rl]), C).histograms()
    assert torch.equal(ha2, 2 * ha)


def test_full_size_nuswide_shape_sharded_eight_ways(xr):
    """BASELINE configs[2] shape (MITH NUS-WIDE: Q 5000 x R 188321 total, 64 bit, 21 classes, gallery in 8 contiguous shards):
    the eight shard scans combined through the offset exchange equal the unsharded scan, which equals the oracle on a query
    subsample."""
    from xmh import sharded
    orc = _orc()
    Q, R, K, C = 5000, 188321, 64, 21
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=2114, p=0.12)
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    whole = xr.RankingScan(q, ql, r, rl, C)
    ha, hr = whole.histograms()
    assert (ha.to(torch.int64).sum(1) == R).all()
    ap_ref, cap_ref = whole.ap_sums(None)
    sub = np.arange(0, Q, 131)[:32]
    dist = orc.hamming_packed(_u32(q.bits)[sub], _u32(r.bits))
    rel = orc.relevance_packed(_u32(ql)[sub], _u32(rl))
    assert np.array_equal(cap_ref.cpu().numpy()[sub], rel.sum(-1))
    assert np.allclose(ap_ref.cpu().numpy()[sub], orc.ap_from_ranking(dist, rel), rtol=3e-6)
    b = sharded.shard_bounds(R, 8)
    shards = [sharded.HipShardOps(q, ql, r.rows(b[i], b[i + 1]), rl[b[i]:b[i + 1]].contiguous(), C) for i in range(8)]
    gathered = torch.stack([torch.stack(o.histograms()) for o in shards]).contiguous()
    ap = torch.zeros(Q, dtype=torch.float64, device="cuda")
    for i, o in enumerate(shards):
        part, cap = o.ap_sums(None, *o.offsets(gathered, i))
        assert torch.equal(cap, cap_ref)
        ap += part
    assert torch.allclose(ap, ap_ref, rtol=1e-6, atol=1e-9)                          # fp32 credits, summed per chunk: order differs


@pytest.mark.parametrize("leg", ["configs0_dcmht_16bit_mirflickr", "k16_coco_shape", "configs3_dsph_128bit", "configs4_shard_scan_256bit",
                                 "configs4_unsharded_scan_256bit"])
def test_bench_legs_full_shapes_match_the_oracle(xr, leg):
    """The extra scan legs of bench.py at their FULL shapes (BASELINE configs[0], 16 bit at the COCO shape, configs[3] through the
    MFMA pass 1 + 16-bit-entry cache, one GPU's shard of configs[4] with a 12.7 GB cache), through the very function the bench
    calls: the mAP the leg prints is the mean of the per-query APs, and those agree with the oracle on a query subsample."""
    import bench_roofline as RL
    orc = _orc()
    cfg = dict(RL.EXTRA_LEGS[leg], steps=1)
    out, scan = RL.extra_scan_leg(return_scan=True, **cfg)
    Q, R = cfg["Q"], cfg["Rn"]
    ha, hr = scan.histograms()
    assert (ha.to(torch.int64).sum(1) == R).all()
    ap, cap = scan.ap_sums(None)
    assert torch.equal(hr.to(torch.int64).sum(1).to(torch.int32), cap)
    assert abs(out["mAP"] - float((ap / cap.double()).mean())) < 1e-9
    nsub = 4 if R > 5000000 else (8 if R > 500000 else 32)
    sub = np.arange(0, Q, Q // nsub)[:nsub]
    dist = orc.hamming_packed(_u32(scan.q.bits)[sub], _u32(scan.r.bits))
    rel = orc.relevance_packed(_u32(scan.qlab)[sub], _u32(scan.rlab))
    assert np.array_equal(cap.cpu().numpy()[sub], rel.sum(-1))
    assert np.allclose(ap.cpu().numpy()[sub], orc.ap_from_ranking(dist, rel), rtol=3e-6)
    assert out["pair_cache_bytes"] > 0            # 12.7 GB for the configs[4] shard, 100 GB for the unsharded 10 M gallery: both under the 128 GB default cap


def test_full_size_long_gallery_256bit_uncached_scan(xr, monkeypatch):
    """BASELINE configs[4] per-GPU shard (10 M / 8 = 1.25 M rows x 256 bit) through the mAP scan: at Q 5000 the pair cache would
    exceed its cap, so this is the path without it -- checked here at Q 192 with the cache switched off against the oracle on a
    query subsample, plus the size-independent properties."""
    monkeypatch.setenv("XMH_SCAN_CACHE_MB", "0")
    orc = _orc()
    Q, R, K, C = 192, 1250000, 256, 24
    g = torch.Generator().manual_seed(4114)
    proto = torch.sign(torch.randn(C, K, generator=g))
    rL = (torch.rand(R, C, generator=g) < 0.06).float()
    rL[torch.arange(R), torch.randint(0, C, (R,), generator=g)] = 1
    qL = (torch.rand(Q, C, generator=g) < 0.06).float()
    qL[torch.arange(Q), torch.randint(0, C, (Q,), generator=g)] = 1
    rB = torch.sign(rL @ proto + 2.5 * torch.randn(R, K, generator=g))
    rB[rB == 0] = 1
    qB = torch.sign(qL @ proto + 2.5 * torch.randn(Q, K, generator=g))
    qB[qB == 0] = 1
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    scan = xr.RankingScan(q, ql, r, rl, C)
    ha, hr = scan.histograms()
    assert (ha.to(torch.int64).sum(1) == R).all()
    ap, cap = scan.ap_sums(None)
    assert torch.equal(hr.to(torch.int64).sum(1).to(torch.int32), cap)
    sub = np.arange(0, Q, 13)[:12]
    dist = orc.hamming_packed(_u32(q.bits)[sub], _u32(r.bits))
    rel = orc.relevance_packed(_u32(ql)[sub], _u32(rl))
    assert np.array_equal(cap.cpu().numpy()[sub], rel.sum(-1))
    assert np.allclose(ap.cpu().numpy()[sub], orc.ap_from_ranking(dist, rel), rtol=3e-6)
    ap2, cap2 = scan.ap_sums(1000)
    assert (cap2 <= 1000).all() and (ap2 <= ap + 1e-9).all()


# ------------------------------------------------------------------------------------------------
# exact per-query top-k (streaming kernel + merge)
# ------------------------------------------------------------------------------------------------
def _topk_check(xr, Q, R, K, k, seed, base_index=0, dup=False):
    from oracle import c_oracle as co
    rng = np.random.default_rng(seed)
    W = (K + 31) // 32
    qb = rng.integers(0, 2**32, size=(Q, W), dtype=np.uint32)
    rb = rng.integers(0, 2**32, size=(R, W), dtype=np.uint32)
    if K % 32:
        qb[:, -1] &= (1 << (K % 32)) - 1
        rb[:, -1] &= (1 << (K % 32)) - 1
    if dup:                                    # heavy ties: few distinct gallery codes
        rb = rb[rng.integers(0, 5, size=R)]
    q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
    r = xr.PackedCodes(torch.from_numpy(rb.view(np.int32)).cuda(), None, K)
    d, i = xr.hamming_topk(q, r, k, base_index)
    wd, wi = co.topk(qb, rb, K + 1, k, base_index)
    assert np.array_equal(i.cpu().numpy(), wi), (Q, R, K, k)
    assert np.array_equal(d.cpu().numpy().view(np.uint16), wd)


@pytest.mark.parametrize("Q,R,K,k", [(3, 5000, 64, 10), (8, 70000, 256, 100), (9, 33333, 128, 1), (17, 20000, 16, 50), (5, 30000, 512, 20), (3, 9000, 2048, 7), (6, 12000, 1024, 100),
                                     (2, 100, 64, 200), (1, 1, 32, 1), (5, 2049, 64, 1024), (4, 300000, 32, 100)])
def test_topk_matches_oracle(xr, Q, R, K, k):
    _topk_check(xr, Q, R, K, k, seed=Q + R + K + k, base_index=12345)


def test_topk_fuzz(xr):
    """30 seeded random (Q, R, K, k) incl. k > R, single rows, every kernel word count, duplicate-heavy galleries."""
    rng = np.random.default_rng(77)
    for case in range(30):
        K = int(rng.choice([8, 32, 64, 128, 256, 512, 1024, 2048]))
        Q = int(rng.choice([1, 2, 7, 8, 9, 33]))
        R = int(rng.choice([1, 5, 255, 256, 1025, 5000, 40001]))
        k = int(rng.choice([1, 2, 10, 100, 1000]))
        _topk_check(xr, Q, R, K, k, seed=500 + case, base_index=int(rng.integers(0, 1 << 20)), dup=case % 3 == 0 and R > 16)


def test_topk_heavy_ties_index_order(xr):
    _topk_check(xr, 6, 40000, 64, 100, seed=1, dup=True)
    _topk_check(xr, 3, 9000, 16, 1000, seed=2, dup=True)
    _topk_check(xr, 3, 15000, 1024, 64, seed=3, dup=True)


def test_topk_adversarial_descending_distance(xr):
    """gallery ordered from far to near: every tile brings better candidates (exercises compaction)."""
    from oracle import c_oracle as co
    K, R, k = 64, 60000, 64
    qb = np.zeros((2, 2), dtype=np.uint32)
    ones = np.sort(np.random.default_rng(0).integers(0, 65, size=R))[::-1]           # popcount per item, descending
    rb = np.zeros((R, 2), dtype=np.uint32)
    for n_ in range(65):
        m = (1 << n_) - 1
        rb[ones == n_] = [m & 0xFFFFFFFF, m >> 32]
    q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
    r = xr.PackedCodes(torch.from_numpy(rb.view(np.int32)).cuda(), None, K)
    d, i = xr.hamming_topk(q, r, k)
    wd, wi = co.topk(qb, rb, K + 1, k)
    assert np.array_equal(i.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy().view(np.uint16), wd)


def test_topk_sharded_merge_equals_global(xr):
    from oracle import c_oracle as co
    from xmh import sharded
    rng = np.random.default_rng(9)
    Q, R, K, k, S = 7, 50000, 64, 100, 4
    qb = rng.integers(0, 2**32, size=(Q, 2), dtype=np.uint32)
    rb = rng.integers(0, 2**32, size=(R, 2), dtype=np.uint32)[rng.integers(0, 3000, size=R)]
    q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
    bounds = sharded.shard_bounds(R, S)
    ds, is_ = [], []
    for s in range(S):
        r = xr.PackedCodes(torch.from_numpy(rb[bounds[s]:bounds[s + 1]].view(np.int32)).cuda(), None, K)
        d, i = xr.hamming_topk(q, r, k, base_index=bounds[s])
        ds.append(d), is_.append(i)
    md, mi = sharded.merge_topk(torch.stack(ds), torch.stack(is_), k)
    wd, wi = co.topk(qb, rb, K + 1, k)
    assert np.array_equal(mi.numpy(), wi) and np.array_equal(md.numpy().astype(np.uint16), wd)


def test_topk_robust_path_alone(xr, monkeypatch):
    """the gated streaming kernels must be exact on their own (they are the fallback of the fast path)."""
    monkeypatch.setenv("XMH_TOPK_ROBUST_ONLY", "1")
    _topk_check(xr, 8, 70000, 256, 100, seed=11)
    _topk_check(xr, 5, 30000, 64, 17, seed=12, dup=True)
    _topk_check(xr, 3, 3000, 16, 1000, seed=13)
    _topk_check(xr, 4, 20000, 512, 50, seed=14, dup=True)      # long codes (TwDH lengths)
    _topk_check(xr, 2, 7000, 2048, 9, seed=15)


def test_topk_full_size_sample_path(xr):
    """a gallery large enough that the threshold comes from a 2.6 % sample (fast path proper)."""
    _topk_check(xr, 8, 2_000_000, 64, 100, seed=21)
    _topk_check(xr, 4, 1_500_000, 256, 10, seed=22)


def test_topk_whole_query_set_over_the_unsharded_10m_gallery(xr):
    """SURVEY 8d shape (5), the Q = 5000 end (bench leg `topk_q5000_10M_256bit` calls the same function): every list of the full result
    is ascending in (distance, index) with distinct in-range indices, and a subsample of the queries equals the C oracle's exact
    top-100 over all 10 M items bit for bit."""
    import bench_topk
    from oracle import c_oracle as co
    Q, R, K, k = 5000, 10_000_000, 256, 100
    out = bench_topk.measure_many_queries(R, K, Q, k, iters=1)
    assert out["lists_sorted_distinct_in_range"]
    q, r = bench_topk._codes("iid", R, K, Q)
    d, i = xr.hamming_topk(q, r, k)
    sub = [0, 1777, 4999]
    qb = q.bits[sub].cpu().numpy().view(np.uint32)
    rb = r.bits.cpu().numpy().view(np.uint32)
    wd, wi = co.topk(qb, rb, K + 1, k)
    assert np.array_equal(i[sub].cpu().numpy(), wi)
    assert np.array_equal(d[sub].cpu().numpy().view(np.uint16), wd)


@pytest.mark.parametrize("Q,R,K,k", [(1, 100_003, 128, 10), (2, 64_001, 128, 100), (3, 50_001, 128, 7), (4, 77_777, 128, 100),
                                     (1, 1_000_001, 256, 100), (2, 33_333, 256, 50), (3, 41_001, 256, 100), (4, 200_003, 256, 9),
                                     (1, 60_001, 512, 100), (3, 25_013, 512, 20), (4, 30_011, 512, 100),
                                     (1, 20_001, 1024, 100), (2, 9_999, 1024, 10), (5, 12_345, 1024, 33), (9, 8_191, 1024, 100),
                                     (1, 7_001, 2048, 50), (3, 5_003, 2048, 100), (11, 4_099, 2048, 8), (1, 15, 256, 3), (4, 255, 128, 100)])
def test_topk_per_piece_filter(xr, Q, R, K, k):
    """k_topk_filter_seq (round 5: codes of whole 16-byte pieces, 1-4 queries at 128 / 256 / 512 bits and every query count at 1024 /
    2048 bits): 1, 2, 4, 8 and 16 lanes per item joined by DPP, groups of 1 / 2 / 4 / 8 queries with surplus slots and several groups,
    ragged last tiles, galleries below and above the sampling size, more rows than one tile per block and fewer than one; and a
    duplicate-heavy gallery whose candidates outgrow the wave's staging list (flushes inside the loop, direct appends, then the
    robust path)."""
    import ctypes
    from xmh._lib import check, lib
    buf = ctypes.create_string_buffer(256)
    check(lib.xmh_topk_describe(Q, R, K, k, buf, 256), "xmh_topk_describe")
    assert b"k_topk_filter_seq<%d, " % (K // 32) in buf.value, buf.value
    _topk_check(xr, Q, R, K, k, seed=5 * Q + R + K + k, base_index=1234)
    if R < 100000:
        _topk_check(xr, Q, R, K, k, seed=Q + K + 1, dup=True)


@pytest.mark.parametrize("K", [16, 32, 64, 128, 256, 512])
def test_calc_map_k_in_one_call_of_the_c_abi(xr, K):
    """xmh_calc_map_k (round 5): float codes + packed label masks in, the mAP on the host out -- what a binding of the reference's
    common/calc_utils.py:58-92 would bind.  Against the oracle: binary codes, codes with exact zeros (ternary kernels; up to 256 bits),
    a cap k, and unquantised values (flag bit 1: nothing evaluated)."""
    import ctypes
    from oracle import retrieval as orc
    from xmh._lib import check, current_stream, lib, ptr
    g = torch.Generator().manual_seed(K)
    Q, R, C = 41, 3000, 37
    qB = torch.randn(Q, K, generator=g).sign(); rB = torch.randn(R, K, generator=g).sign()
    qB[qB == 0] = 1; rB[rB == 0] = 1
    qL = (torch.rand(Q, C, generator=g) < 0.1).long(); rL = (torch.rand(R, C, generator=g) < 0.1).long()
    qL[:, 0] = 1; rL[::5, 0] = 1
    ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
    need = int(lib.xmh_calc_map_k_ws_bytes(Q, R, K, C))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")

    def call(q, r, k):
        m, fl = ctypes.c_double(-1.0), ctypes.c_int32(-1)
        qd, rd = q.cuda().contiguous(), r.cuda().contiguous()
        check(lib.xmh_calc_map_k(ptr(qd), ptr(rd), ptr(ql), ptr(rl), Q, R, K, C, 0 if k is None else k, ptr(ws), need, ctypes.byref(m),
                                 ctypes.byref(fl), current_stream()), "xmh_calc_map_k")
        return m.value, fl.value

    for k in (None, 25):
        got, fl = call(qB, rB, k)
        assert fl == 0 and abs(got - float(orc.map_k(qB, rB, qL, rL, k))) < 2e-6, (K, k)
    if K <= 256:
        qz, rz = qB.clone(), rB.clone()
        qz[3, 5] = 0.0; rz[::17, K // 2] = 0.0
        got, fl = call(qz, rz, None)
        assert fl == 1 and abs(got - float(orc.map_k(qz, rz, qL, rL, None))) < 2e-6, K
    ru = rB.clone(); ru[7, 1] = 0.4
    got, fl = call(qB, ru, None)
    assert fl & 2 and got == -1.0                                   # not quantised: nothing written
    # a workspace one byte short is refused
    m, f2 = ctypes.c_double(0.0), ctypes.c_int32(0)
    assert lib.xmh_calc_map_k(ptr(qB.cuda()), ptr(rB.cuda()), ptr(ql), ptr(rl), Q, R, K, C, 0, ptr(ws), need - 1, ctypes.byref(m), ctypes.byref(f2),
                              current_stream()) != 0


@pytest.mark.parametrize("C", [129, 160, 200, 255, 256])
@pytest.mark.parametrize("K", [16, 64, 128, 256])
def test_map_with_more_than_128_classes(xr, K, C):
    """Round 5: 129 ... 256 classes (IAPR TC-12 has 255) run as eight label words on the VALU kernels (the Python layer pads 5 ... 7 words
    to 8); the drop-in calc_map_k, the scan object and the top-k-capped form against the oracle."""
    from oracle import retrieval as orc
    from xmh.common import calc_utils as cu
    g = torch.Generator().manual_seed(K + C)
    Q, R = 37, 2500
    qB = torch.randn(Q, K, generator=g).sign(); rB = torch.randn(R, K, generator=g).sign()
    qB[qB == 0] = 1; rB[rB == 0] = 1
    qL = (torch.rand(Q, C, generator=g) < 0.02).long(); rL = (torch.rand(R, C, generator=g) < 0.02).long()
    qL[:, C - 1] = 1; rL[::7, C - 1] = 1                          # the last class matters: it sits in the last label word
    for k in (None, 50):
        want = float(orc.map_k(qB, rB, qL, rL, k))
        got = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), k))
        assert abs(got - want) < 2e-6, (K, C, k, got, want)


@pytest.mark.parametrize("K", [8, 16, 24, 32, 64, 96, 100, 128, 256])
def test_pack_sign_and_unpack_every_kernel_form(xr, K):
    """Round 5: pack_sign has a flat 16-byte form for code lengths of whole words (k_pack_sign_flat: four sign bits per lane, DPP nibble
    gather), one for 16- / 8-bit codes (k_pack_sign_short: one word per row) and the slot-per-lane form for everything else and for
    inputs that are not 16-byte aligned; unpack has a 16-byte form (K % 4 == 0).  Same words, same zero plane (padding = dead), same
    flags, same scatter: against numpy, with zeros, unquantised values and NaN, odd row counts, scattered rows and a misaligned view."""
    rng = np.random.default_rng(K)
    W = (K + 31) // 32
    for n in (1, 7, 64, 1000, 4099):
        x = rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=(n, K))
        for variant in ("pm1", "zeros", "other"):
            y = x.copy()
            if variant == "zeros":
                y[rng.integers(0, n, size=max(1, n // 5)), rng.integers(0, K, size=max(1, n // 5))] = 0.0
            if variant == "other":
                y[rng.integers(0, n), rng.integers(0, K)] = 0.37
                y[rng.integers(0, n), rng.integers(0, K)] = np.nan
            want_bits = np.zeros((n, W), dtype=np.uint32)
            want_zero = np.zeros((n, W), dtype=np.uint32)
            for c in range(W * 32):
                if c < K:
                    want_bits[:, c // 32] |= (y[:, c] > 0).astype(np.uint32) << np.uint32(c % 32)
                    want_zero[:, c // 32] |= (y[:, c] == 0).astype(np.uint32) << np.uint32(c % 32)
                else:
                    want_zero[:, c // 32] |= np.uint32(1) << np.uint32(c % 32)
            want_flags = (1 if (y == 0).any() else 0) | (2 if ((y != 0) & (np.abs(y) != 1)).any() else 0)
            for view in ("aligned", "misaligned"):
                buf = torch.zeros(n * K + 8, dtype=torch.float32, device="cuda")
                off = 0 if view == "aligned" else 1
                t = buf[off: off + n * K].view(n, K)
                t.copy_(torch.from_numpy(y))
                p = xr.pack_sign(t)
                assert p.flags == want_flags, (K, n, variant, view)
                assert np.array_equal(p.bits.cpu().numpy().view(np.uint32), want_bits), (K, n, variant, view)
                if want_flags & 1:
                    assert np.array_equal(p.zero.cpu().numpy().view(np.uint32), want_zero), (K, n, variant, view)
                else:
                    assert p.zero is None
                # scatter mode: rows land at row_index
                perm = torch.from_numpy(rng.permutation(n + 3)[:n].astype(np.int64)).cuda()
                out = xr.empty_packed(n + 3, K, "cuda", with_zero=True)
                fl = torch.zeros(1, dtype=torch.int32, device="cuda")
                xr.pack_sign(t, out=out, row_index=perm, flags=fl)
                assert int(fl.item()) == want_flags
                assert np.array_equal(out.bits[perm].cpu().numpy().view(np.uint32), want_bits)
                assert np.array_equal(out.zero[perm].cpu().numpy().view(np.uint32), want_zero)
            if variant != "other":
                q = xr.pack_sign(torch.from_numpy(y).cuda())
                assert np.array_equal(q.unpack().cpu().numpy(), np.sign(y)), (K, n, variant)


@pytest.mark.parametrize("K", [16, 64, 100, 128, 256])
def test_materialised_outputs_on_odd_shapes_and_unaligned_output_pointers(xr, K):
    """Round 5: the float32 outputs of calc_hammingDist / calc_label_sim are written in 128-byte-aligned runs whose columns slide with
    each row's start (xmh_dist.hip: k_dist_f32 / k_label_sim_f32; items staged in LDS with 32 extra columns).  Every column of every
    row must still be written exactly once and nothing outside the matrix: row lengths around the 32-column and 1024-column steps, a
    single column, and output pointers 1, 2, 3 and 7 floats into a buffer (guard values before and after the matrix)."""
    from oracle import retrieval as orc
    from xmh._lib import check, current_stream, lib, ptr
    g = torch.Generator().manual_seed(K)
    C = 70
    for Q, R in ((1, 1), (3, 31), (2, 33), (33, 1023), (5, 1025), (34, 2049), (7, 4100)):
        qB = torch.randn(Q, K, generator=g).sign(); rB = torch.randn(R, K, generator=g).sign()
        qB[qB == 0] = 1; rB[rB == 0] = 1
        qL = (torch.rand(Q, C, generator=g) < 0.1).long(); rL = (torch.rand(R, C, generator=g) < 0.1).long()
        q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
        q, r = xr.PackedCodes(q.bits, None, K), xr.PackedCodes(r.bits, None, K)
        ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
        want_d = orc.hamming_dist(qB, rB)
        want_s = (qL.float() @ rL.float().t() > 0).float()
        for off in (0, 1, 2, 3, 7):
            for what in ("dist", "sim"):
                buf = torch.full((Q * R + 64,), -7.0, dtype=torch.float32, device="cuda")
                out = buf[off + 16: off + 16 + Q * R]
                if what == "dist":
                    check(lib.xmh_hamming_dist(ptr(q.bits), None, ptr(r.bits), None, Q, R, K, ptr(out), None, current_stream()), "xmh_hamming_dist")
                    want = want_d
                else:
                    check(lib.xmh_label_sim(ptr(ql), ptr(rl), Q, R, C, ptr(out), current_stream()), "xmh_label_sim")
                    want = want_s
                got = buf.cpu()
                assert torch.equal(got[off + 16: off + 16 + Q * R].view(Q, R), want), (what, Q, R, K, off)
                assert (got[:off + 16] == -7.0).all() and (got[off + 16 + Q * R:] == -7.0).all(), (what, Q, R, K, off)


@pytest.mark.parametrize("Q,R,K,k", [(1, 3_000_000, 16, 100), (5, 2_000_001, 16, 37), (3, 4_000_000, 24, 100), (9, 1_500_000, 32, 1000), (2, 6_000_000, 8, 10)])
def test_topk_index_bound_on_coarse_codes(xr, Q, R, K, k):
    """Round 5: on coarse codes the threshold bucket holds thousands of ties and only its first rows are wanted; the pick hands the
    filters an index bound for that bucket (index_bound: estimated from the sampled density, exact for ANY value because every
    non-candidate follows every candidate in (distance, index) order).  Galleries above the sampling size so that the bound is an
    estimate: iid codes, and a gallery SORTED by code (the density of the bucket in the first rows is nothing like the average: the
    bound is too small or too large and the select's count check / the robust path have to hold the result)."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(Q + R + K + k)
    W = 1
    mask = (1 << K) - 1
    qb = (rng.integers(0, 2**32, size=(Q, W), dtype=np.uint32) & mask).astype(np.uint32)
    for kind in ("iid", "sorted"):
        rb = (rng.integers(0, 2**32, size=(R, W), dtype=np.uint32) & mask).astype(np.uint32)
        if kind == "sorted":
            rb = np.sort(rb, axis=0)
        q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
        r = xr.PackedCodes(torch.from_numpy(rb.view(np.int32)).cuda(), None, K)
        d, i = xr.hamming_topk(q, r, k, 11)
        wd, wi = co.topk(qb, rb, K + 1, k, 11)
        assert np.array_equal(i.cpu().numpy(), wi), (kind, Q, R, K, k)
        assert np.array_equal(d.cpu().numpy().view(np.uint16), wd)


@pytest.mark.parametrize("K", [64, 256])
def test_topk_robust_path_for_the_failed_queries_only(xr, K):
    """Round 5: the fast path's fail flag is per query.  Twelve queries in two groups of eight; query 0 equals a code that the gallery
    holds 20 000 times -- 20 000 candidates at distance 0, its list overflows and the select raises ITS flag; in a second gallery the
    same happens to query 11.  The robust kernels recompute group 0 (resp. group 1) only, the merge replaces the failed query's list
    only, and every list of the call equals the oracle's."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(41 + K)
    W = K // 32
    R, Q, k = 300_001, 12, 100
    rb = rng.integers(0, 2**32, size=(R, W), dtype=np.uint32)
    qb = rng.integers(0, 2**32, size=(Q, W), dtype=np.uint32)
    for failing in (0, 11):
        g = rb.copy()
        g[rng.choice(R, size=20_000, replace=False)] = qb[failing]
        q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
        r = xr.PackedCodes(torch.from_numpy(g.view(np.int32)).cuda(), None, K)
        d, i = xr.hamming_topk(q, r, k, 3)
        wd, wi = co.topk(qb, g, K + 1, k, 3)
        assert np.array_equal(i.cpu().numpy(), wi), (K, failing)
        assert np.array_equal(d.cpu().numpy().view(np.uint16), wd)
        assert (wd[failing] == 0).all() and (wd[(failing + 5) % Q] > 0).all()


@pytest.mark.parametrize("Q,R,K,k", [(1, 100_003, 64, 10), (2, 64_001, 64, 100), (3, 50_001, 32, 7), (4, 77_777, 32, 100), (5, 33_333, 64, 50),
                                     (8, 41_001, 16, 100), (9, 200_003, 64, 9), (17, 60_001, 32, 100), (1, 1_000_001, 32, 100), (1, 3, 64, 2),
                                     (6, 1023, 24, 5), (2, 4097, 64, 100)])
def test_topk_short_code_filter(xr, Q, R, K, k):
    """k_topk_filter_short (round 5: 32- and 64-bit codes as 16-byte pieces of 4 / 2 items): groups of 1 / 2 / 4 / 8 queries and several
    passes, ragged last pieces, and -- what a shard cut at any row hands over -- gallery views that do NOT start on a 16-byte boundary
    (1, 2, 3 rows into a buffer: the kernel reads from the boundary below and masks the foreign words), with and without duplicates."""
    import ctypes
    from oracle import c_oracle as co
    from xmh._lib import check, lib
    buf = ctypes.create_string_buffer(256)
    check(lib.xmh_topk_describe(Q, R, K, k, buf, 256), "xmh_topk_describe")
    assert b"k_topk_filter_short<%d, " % ((K + 31) // 32) in buf.value, buf.value
    _topk_check(xr, Q, R, K, k, seed=7 * Q + R + K + k, base_index=99)
    if 5 <= R < 100000:
        _topk_check(xr, Q, R, K, k, seed=Q + K + 2, dup=True)
    rng = np.random.default_rng(R + K)
    W = (K + 31) // 32
    qb = rng.integers(0, 2**32, size=(Q, W), dtype=np.uint32)
    big = rng.integers(0, 2**32, size=(R + 3, W), dtype=np.uint32)
    if K % 32:
        qb[:, -1] &= (1 << (K % 32)) - 1
        big[:, -1] &= (1 << (K % 32)) - 1
    q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
    whole = xr.PackedCodes(torch.from_numpy(big.view(np.int32)).cuda(), None, K)
    for lo in (1, 2, 3):
        view = whole.rows(lo, lo + R)
        assert view.bits.data_ptr() == whole.bits.data_ptr() + lo * W * 4          # a view, not a copy
        d, i = xr.hamming_topk(q, view, k, 5)
        wd, wi = co.topk(qb, big[lo:lo + R], K + 1, k, 5)
        assert np.array_equal(i.cpu().numpy(), wi), (lo, Q, R, K, k)
        assert np.array_equal(d.cpu().numpy().view(np.uint16), wd)


@pytest.mark.parametrize("Q,R,K,k", [(5, 50001, 128, 10), (16, 33333, 128, 100), (17, 70001, 256, 100), (33, 40007, 256, 5), (70, 25013, 256, 100),
                                     (20, 30011, 512, 50), (40, 9999, 512, 100), (5, 15, 256, 3), (64, 1_200_003, 256, 100), (12, 900_001, 128, 20)])
def test_topk_matrix_core_filter(xr, Q, R, K, k):
    """k_topk_filter_mfma (>= 5 queries at 128 / 256 / 512 bits): one, two and four query tiles per pass, several passes,
    ragged last groups, galleries below and above the sampling size; and a duplicate-heavy gallery whose candidates outgrow the
    wave's staging list and the per-query lists (direct appends, then the robust path)."""
    _topk_check(xr, Q, R, K, k, seed=3 * Q + R + K + k, base_index=77)
    if R < 100000:
        _topk_check(xr, Q, R, K, k, seed=Q + K, dup=True)


def test_topk_prepared_workspace_stays_clean_over_calls(xr):
    """xmh_topk_ws_init once, then a query loop through xmh_hamming_topk_prepared: every call finds the control words and the sample
    histogram zero and leaves them zero -- few queries (thresholds picked by the last sample block), many queries (pick kernel),
    a duplicate-heavy gallery whose lists overflow (robust path recomputes), then ordinary queries again on the same workspace."""
    from oracle import c_oracle as co
    rng = np.random.default_rng(31)
    for (Q, R, K, k) in ((1, 600_000, 256, 100), (8, 400_000, 64, 50), (40, 300_000, 64, 10), (3, 5000, 128, 20)):
        W = (K + 31) // 32
        rb = rng.integers(0, 2**32, size=(R, W), dtype=np.uint32)
        rdup = rb[rng.integers(0, 4, size=R)]                     # 4 distinct codes: every candidate list overflows
        ws = xr.TopkWorkspace(Q, R, K, k, "cuda")
        r = xr.PackedCodes(torch.from_numpy(rb.view(np.int32)).cuda(), None, K)
        rd = xr.PackedCodes(torch.from_numpy(rdup.view(np.int32)).cuda(), None, K)
        for rnd, (gal, gal_np) in enumerate(((r, rb), (r, rb), (rd, rdup), (r, rb), (rd, rdup), (r, rb))):
            qb = rng.integers(0, 2**32, size=(Q, W), dtype=np.uint32)
            q = xr.PackedCodes(torch.from_numpy(qb.view(np.int32)).cuda(), None, K)
            d, i = xr.hamming_topk(q, gal, k, base_index=7 * rnd, workspace=ws)
            wd, wi = co.topk(qb, gal_np, K + 1, k, 7 * rnd)
            assert np.array_equal(i.cpu().numpy(), wi), (Q, R, K, k, rnd)
            assert np.array_equal(d.cpu().numpy().view(np.uint16), wd), (Q, R, K, k, rnd)
    with pytest.raises(ValueError):
        xr.hamming_topk(q, r, k + 1, workspace=ws)


def test_float_similarities_and_float_code_fallback(cu):
    """cosine / euclidean / calc_hammingDist / calc_map_k on un-quantised float inputs vs the goldens."""
    g = np.load(os.path.join(GOLDEN, "calc_utils_ternary_float.npz"))
    fa, fb = dev(g["fa"]), dev(g["fb"])
    assert np.allclose(cu.cosine_similarity(fa, fb).cpu().numpy(), g["cos"], atol=2e-6)
    assert np.allclose(cu.cosine_similarity(g["fa"], g["fb"]), g["cos_np"], atol=2e-6)          # numpy in -> numpy out
    assert np.allclose(cu.euclidean_similarity(fa, fb).cpu().numpy(), g["euc"], atol=2e-5)
    assert np.allclose(cu.euclidean_similarity(g["fa"], g["fb"]), g["euc_np"], atol=2e-5)
    with pytest.raises(ValueError):
        cu.cosine_similarity(fa, g["fb"])
    with pytest.raises(ValueError):
        cu.euclidean_similarity(g["fa"], fb)
    fq, fr = dev(g["fq"]), dev(g["fr"])
    assert np.allclose(cu.calc_hammingDist(fq, fr).cpu().numpy(), g["float_dist"], atol=1e-5)
    got = cu.calc_map_k(fq, fr, dev(g["fqL"], torch.int64), dev(g["frL"], torch.int64))
    assert abs(float(got) - float(g["float_map_stable"])) < 1e-6
    kat = np.load(os.path.join(GOLDEN, "calc_utils_kat.npz"))
    assert float(cu.calc_hammingDist(dev(kat["kat4_a"]), dev(kat["kat4_b"])).cpu().reshape(-1)[0]) == 1.0     # KAT-4


@pytest.mark.parametrize("K", [512, 2048, 96])
def test_long_and_odd_code_lengths_distance(xr, cu, K):
    """TwDH-style long codes (up to 2048 bits) and a non-power-of-two word count go through the generic
    distance kernel; what the bit-packed scan has no kernel for (4096 bits) is evaluated on the float route (round 6: the drop-in never
    refuses what the reference evaluates), while the scan's own C entry still reports the limit."""
    orc = _orc()
    gen = torch.Generator().manual_seed(K)
    qB, rB = torch.randn(9, K, generator=gen).sign(), torch.randn(301, K, generator=gen).sign()
    assert torch.equal(cu.calc_hammingDist(qB.cuda(), rB.cuda()).cpu(), orc.hamming_dist(qB, rB))
    q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    assert np.array_equal(xr.hamming_dist(q, r, as_u16=True).cpu().numpy().view(np.uint16), orc.hamming_packed(_u32(q.bits), _u32(r.bits)))
    assert torch.equal(q.unpack().cpu(), qB)
    L = torch.ones(9, 3, dtype=torch.int64)
    if K == 96:                                               # 3 words: the ranking kernels see a zero-padded 4-word code
        gL = torch.Generator().manual_seed(7)
        qL, rL = (torch.rand(9, 5, generator=gL) < 0.4).long(), (torch.rand(301, 5, generator=gL) < 0.4).long()
        qL[:, 0] = 1
        rL[::2, 0] = 1
        for kk in (None, 11):
            assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), kk)) - float(orc.map_k(qB, rB, qL, rL, kk, stable=True))) < MAP_TOL
        qT = qB.clone()
        qT[:, ::7] = 0.0                                      # ternary on one side
        assert abs(float(cu.calc_map_k(qT.cuda(), rB.cuda(), qL.cuda(), rL.cuda())) - float(orc.map_k(qT, rB, qL, rL, stable=True))) < MAP_TOL
        d, i = xr.hamming_topk(q, r, 10)
        want_d, want_i = orc.topk_packed(_u32(q.bits), _u32(r.bits), 10) if hasattr(orc, "topk_packed") else (None, None)
        full = orc.hamming_packed(_u32(q.bits), _u32(r.bits)).astype(np.int64)
        order = np.argsort(full * 1000 + np.arange(301)[None, :], axis=1, kind="stable")[:, :10]
        assert np.array_equal(i.cpu().numpy(), order) and np.array_equal(d.cpu().numpy().view(np.uint16), np.take_along_axis(full, order, 1).astype(np.uint16))
    qB4, rB4 = torch.randn(9, 4096, generator=gen).sign(), torch.randn(40, 4096, generator=gen).sign()
    L4 = torch.ones(40, 3, dtype=torch.int64)
    assert torch.equal(cu.calc_hammingDist(qB4.cuda(), rB4.cuda()).cpu(), orc.hamming_dist(qB4, rB4))       # 128 code words: the generic distance kernel
    got = cu.calc_map_k(qB4.cuda(), rB4.cuda(), L.cuda(), L4.cuda())
    assert abs(float(got) - float(orc.map_k(qB4, rB4, L, L4, stable=True))) < MAP_TOL
    with pytest.raises(RuntimeError, match="at most 2048"):
        xr.scan_plan(9, 40, 4096, False)


@pytest.mark.parametrize("Q,R,K,C", [(21, 3000, 512, 24), (9, 2500, 1024, 80), (12, 4100, 2048, 21)])
def test_scan_long_codes_match_oracle(xr, cu, monkeypatch, Q, R, K, C):
    """TwDH-style 512..2048-bit codes: one query per wave, one gallery item per lane; histograms bit-exact,
    mAP (all / @k, packed and 64-bit counters, duplicate gallery codes so that lanes collide) within 1e-6."""
    orc = _orc()
    qB, rB, qL, rL = _synth(Q, R, K, C, seed=K + R)
    rB[R // 2:] = rB[torch.randint(0, 7, (R - R // 2,), generator=torch.Generator().manual_seed(K))]     # many equal distances
    q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
    scan = xr.RankingScan(q, ql, r, rl, C)
    ha, hr = scan.histograms()
    dist = orc.hamming_packed(_u32(q.bits), _u32(r.bits))
    rel = orc.relevance_packed(_u32(ql), _u32(rl))
    wa, wr = orc.bucket_histograms(dist, rel, K + 1)
    assert np.array_equal(_u32(ha), wa) and np.array_equal(_u32(hr), wr)
    want, want9 = orc.map_k(qB, rB, qL, rL, stable=True), orc.map_k(qB, rB, qL, rL, 9, stable=True)
    assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())) - float(want)) < MAP_TOL
    monkeypatch.setenv("XMH_SCAN_PACK32", "0")
    assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())) - float(want)) < MAP_TOL
    assert abs(float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), 9)) - float(want9)) < MAP_TOL


def test_error_paths_report_through_last_error(xr):
    from xmh import _lib
    q = xr.PackedCodes(torch.zeros(4, 2, dtype=torch.int32, device="cuda"), None, 64)
    r = xr.PackedCodes(torch.zeros(10, 4, dtype=torch.int32, device="cuda"), None, 128)
    with pytest.raises(ValueError):
        xr.hamming_dist(q, r)
    with pytest.raises(_lib.XmhError, match="k=0"):
        xr.hamming_topk(q, xr.PackedCodes(torch.zeros(10, 2, dtype=torch.int32, device="cuda"), None, 64), 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        xr.pack_sign(torch.zeros(3, 64))


# ------------------------------------------------------------------------------------------------
# round 6: exact top-k over TERNARY codes (sign_() leaves exact zeros: reference runners/base.py:407-410, MITH runner.py:125-131)
# ------------------------------------------------------------------------------------------------
def _topk_ternary_check(xr, Q, R, K, k, seed, base_index=0, dup=False, p_zero=0.15, q_zero=True, r_zero=True, float_in=False):
    """xmh_hamming_topk_ternary against orc_topk_ternary (itself pinned on the reference's calc_hammingDist output, tests/test_oracle_c.py):
    indices bit-exact, distances in half units K - q.r."""
    from oracle import c_oracle as co
    from oracle import retrieval as orc
    rng = np.random.default_rng(seed)
    pz = [(1 - p_zero) / 2, p_zero, (1 - p_zero) / 2]
    qc = rng.choice([-1, 0, 1], p=pz if q_zero else [0.5, 0.0, 0.5], size=(Q, K)).astype(np.float32)
    rc = rng.choice([-1, 0, 1], p=pz if r_zero else [0.5, 0.0, 0.5], size=(R if not dup else 5, K)).astype(np.float32)
    if dup:
        rc = rc[rng.integers(0, 5, size=R)]
    q, r = xr.pack_sign(torch.from_numpy(qc).cuda()), xr.pack_sign(torch.from_numpy(rc).cuda())
    assert (q.zero is not None) == bool((qc == 0).any()) and (r.zero is not None) == bool((rc == 0).any())
    d, i = xr.hamming_topk(q, r, k, base_index, ternary=True)          # (True: a draw without any zero still answers in half units)
    (qb, qz), (rb, rz) = orc.pack_bits(qc), orc.pack_bits(rc)
    pad = qb.shape[1] * 32 - K
    if pad:
        m = np.uint32(((1 << pad) - 1) << (32 - pad))
        qz[:, -1] |= m
        rz[:, -1] |= m
    wd, wi = co.topk_ternary(qb, qz, rb, rz, K, k, base_index)
    assert np.array_equal(i.cpu().numpy(), wi), (Q, R, K, k)
    assert np.array_equal(d.cpu().numpy().view(np.uint16), wd), (Q, R, K, k)
    return d, i


@pytest.mark.parametrize("K", [16, 64, 128, 256])
@pytest.mark.parametrize("Q,R,k", [(1, 30000, 10), (3, 70000, 100), (8, 9000, 50), (17, 20000, 1)])
def test_topk_ternary_matches_oracle(xr, Q, R, K, k):
    """VERDICT r5 missing 1: 15 % zeros at the reference's code lengths, every query-group width of the filters"""
    _topk_ternary_check(xr, Q, R, K, k, seed=Q + R + K + k, base_index=4321)


def test_topk_ternary_is_the_stable_sort_of_the_reference_distance_matrix(xr, cu):
    """end to end on the drop-in's own distance matrix (calc_hammingDist = 0.5 * (K - q.r), bit-exact vs the reference's golden):
    torch.sort(stable=True) of it gives the same lists"""
    gen = torch.Generator().manual_seed(66)
    qB, rB = _ternary_codes(5, 64, gen), _ternary_codes(12000, 64, gen)
    dist = cu.calc_hammingDist(qB.cuda(), rB.cuda())
    val, order = torch.sort(dist, dim=1, stable=True)
    q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    d, i = xr.hamming_topk(q, r, 200)
    assert torch.equal(i.long(), order[:, :200])
    assert torch.equal(d.float() * 0.5, val[:, :200])


def test_topk_ternary_odd_lengths_heavy_ties_and_one_sided_zeros(xr):
    _topk_ternary_check(xr, 4, 20000, 48, 30, seed=1)                       # padding bits inside the last word
    _topk_ternary_check(xr, 4, 20000, 96, 30, seed=2)                       # three words run as four: the widening constant is taken out
    _topk_ternary_check(xr, 2, 5000, 24, 1000, seed=3, dup=True)            # five distinct gallery codes: index order decides
    _topk_ternary_check(xr, 6, 40000, 64, 100, seed=4, dup=True)
    _topk_ternary_check(xr, 3, 15000, 64, 20, seed=5, r_zero=False)         # zeros in the queries only: the gallery gets the default plane
    _topk_ternary_check(xr, 3, 15000, 128, 20, seed=6, q_zero=False)
    _topk_ternary_check(xr, 2, 100, 64, 200, seed=7)                        # k > R: unused slots
    _topk_ternary_check(xr, 1, 1, 32, 1, seed=8, p_zero=0.5)
    _topk_ternary_check(xr, 5, 30000, 512, 20, seed=9)                      # TwDH lengths
    _topk_ternary_check(xr, 2, 9000, 1024, 7, seed=10)
    _topk_ternary_check(xr, 3, 7000, 2048, 9, seed=11)                      # 4097 buckets: the sample keeps 7 queries' histograms per round
    _topk_ternary_check(xr, 17, 5000, 2048, 20, seed=12)                    # ... and without the folded pick (more than 16 queries)
    _topk_ternary_check(xr, 33, 6000, 1024, 100, seed=13)


def test_topk_ternary_fuzz(xr):
    rng = np.random.default_rng(78)
    for case in range(24):
        K = int(rng.choice([8, 16, 32, 64, 128, 256]))
        Q = int(rng.choice([1, 2, 3, 7, 8, 9, 33]))
        R = int(rng.choice([1, 5, 255, 256, 1025, 5000, 40001]))
        k = int(rng.choice([1, 2, 10, 100, 1000]))
        _topk_ternary_check(xr, Q, R, K, k, seed=800 + case, base_index=int(rng.integers(0, 1 << 20)), dup=case % 3 == 0 and R > 16,
                            p_zero=float(rng.choice([0.01, 0.15, 0.6])))


def test_topk_ternary_robust_path_alone(xr, monkeypatch):
    monkeypatch.setenv("XMH_TOPK_ROBUST_ONLY", "1")
    _topk_ternary_check(xr, 8, 70000, 256, 100, seed=11)
    _topk_ternary_check(xr, 5, 30000, 64, 17, seed=12, dup=True)
    _topk_ternary_check(xr, 3, 3000, 16, 1000, seed=13)
    _topk_ternary_check(xr, 9, 20000, 128, 50, seed=14)


def test_topk_ternary_full_size_sample_path(xr):
    _topk_ternary_check(xr, 8, 2_000_000, 64, 100, seed=21)
    _topk_ternary_check(xr, 2, 1_500_000, 256, 10, seed=22)


def test_topk_ternary_prepared_workspace_and_forced_unit(xr):
    """a prepared ternary workspace stays clean over calls; ternary=True on binary code sets gives the binary lists in half units
    (the rank of a sharded call whose shard holds no zero)"""
    gen = torch.Generator().manual_seed(5)
    qB, rB = _ternary_codes(6, 64, gen), _ternary_codes(30000, 64, gen)
    q, r = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    ws = xr.TopkWorkspace(6, 30000, 64, 50, q.bits.device, ternary=True)
    first = xr.hamming_topk(q, r, 50, workspace=ws)
    for _ in range(3):
        again = xr.hamming_topk(q, r, 50, workspace=ws)
        assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[1])
    scratch = xr.hamming_topk(q, r, 50)
    assert torch.equal(scratch[0], first[0]) and torch.equal(scratch[1], first[1])
    with pytest.raises(ValueError):
        xr.hamming_topk(xr.PackedCodes(q.bits, None, 64), xr.PackedCodes(r.bits, None, 64), 50, workspace=ws)
    qb, rb = xr.PackedCodes(q.bits, None, 64), xr.PackedCodes(r.bits, None, 64)
    d1, i1 = xr.hamming_topk(qb, rb, 50)
    d2, i2 = xr.hamming_topk(qb, rb, 50, ternary=True)
    assert torch.equal(i1, i2) and torch.equal(d2, d1 * 2)


def test_topk_sharded_ternary_merge_equals_global(xr):
    from xmh import sharded
    gen = torch.Generator().manual_seed(9)
    Q, R, K, k, S = 7, 50000, 64, 100, 4
    qB, rB = _ternary_codes(Q, K, gen), _ternary_codes(3000, K, gen)[torch.randint(0, 3000, (R,), generator=gen)]
    rB[: R // 2][rB[: R // 2] == 0] = 1.0                                    # the first two shards hold no zero at all
    q, whole = xr.pack_sign(qB.cuda()), xr.pack_sign(rB.cuda())
    bounds = sharded.shard_bounds(R, S)
    ds, is_ = [], []
    for s in range(S):
        r = xr.pack_sign(rB[bounds[s]:bounds[s + 1]].cuda())
        d, i = xr.hamming_topk(q, r, k, base_index=bounds[s], ternary=True)   # the unit every rank must use (sharded.topk_sharded agrees on it)
        ds.append(d), is_.append(i)
    assert r.zero is not None
    md, mi = sharded.merge_topk(torch.stack(ds), torch.stack(is_), k)
    wd, wi = xr.hamming_topk(q, whole, k)
    assert torch.equal(mi, wi.cpu()) and torch.equal(md, (wd.cpu().int() & 0xFFFF))


# ------------------------------------------------------------------------------------------------
# round 6: the float route = the reference's own (GEMM + ONE stable sort per query), and the drop-in never refuses a shape
# ------------------------------------------------------------------------------------------------
def _ap_from_distances(dist, rel, k=None):
    """reference calc_utils.py:76-89 on a given distance matrix with torch.sort(stable=True): per-query sum(ordinal / rank), cap"""
    order = torch.sort(dist, dim=1, stable=True).indices
    sums, caps = [], []
    for i in range(dist.shape[0]):
        hits = rel[i][order[i]]
        n = int(hits.sum()) if k is None else min(int(hits.sum()), k)
        rank = torch.nonzero(hits)[:n].squeeze(-1).to(torch.float32) + 1.0
        sums.append(float((torch.arange(1, n + 1, dtype=torch.float32) / rank).to(torch.float64).sum()))
        caps.append(n)
    return np.array(sums), np.array(caps)


@pytest.mark.parametrize("Q,R,C", [(3, 1, 5), (4, 63, 21), (5, 64, 80), (7, 1000, 33), (3, 70001, 24), (33, 5000, 300)])
def test_float_sort_ranking_is_torch_stable_sort(xr, Q, R, C):
    """xmh_float_sort_ap (segmented LSD radix sort + AP pass) on distance matrices full of ties, negative values and signed zeros:
    the same sums as torch.sort(stable=True) of the same matrix, for k in {None, 1, 10}"""
    from xmh import dense
    gen = torch.Generator().manual_seed(Q * 1000 + R)
    dist = (torch.randn(Q, R, generator=gen) * 3).mul(4).round().div(4)                # quarter steps: thousands of ties per row
    dist[:, ::7] = 0.0
    dist[:, 3::11] = -0.0
    dist[0, :] = 5.0                                                                    # one row: a single bucket, pure index order
    if R > 100:
        dist[1, :] = torch.randn(R, generator=gen) * 1e-30                             # tiny magnitudes, both signs
        dist[2, :50] = float("inf")
    qL = (torch.rand(Q, C, generator=gen) < 0.1).to(torch.int64)
    rL = (torch.rand(R, C, generator=gen) < 0.1).to(torch.int64)
    qL[:, 0] = 1
    rL[::3, 0] = 1
    rel = (qL.float() @ rL.float().t()) > 0
    ql, rl = xr.pack_labels(qL.cuda()), xr.pack_labels(rL.cuda())
    for k in (None, 1, 10):
        ap, cap = dense.float_sort_ap(dist.cuda(), ql, rl, C, k)
        want, wcap = _ap_from_distances(dist, rel, k)
        assert np.array_equal(cap.cpu().numpy(), wcap)
        assert np.allclose(ap.cpu().numpy(), want, rtol=1e-9, atol=1e-12), (k, np.abs(ap.cpu().numpy() - want).max())


def test_calc_map_k_on_umoed_style_float_codes_at_scale(cu):
    """tanh "codes" (reference runners/UMoED/runner.py:162-186 hands calc_map_k un-quantised values) at
    Q 500 x R 117 218 x 64: under 50 ms per call, and equal to the oracle's calc_map_k port on a query subsample."""
    import time
    orc = _orc()
    gen = torch.Generator().manual_seed(42)
    Q, R, K, C = 500, 117218, 64, 80
    qL = (torch.rand(Q, C, generator=gen) < 0.04).to(torch.int64)
    rL = (torch.rand(R, C, generator=gen) < 0.04).to(torch.int64)
    qL[:, 0] = 1
    rL[::5, 0] = 1
    W = torch.randn(C, K, generator=gen)
    qB = torch.tanh(qL.float() @ W + 0.8 * torch.randn(Q, K, generator=gen))
    rB = torch.tanh(rL.float() @ W + 0.8 * torch.randn(R, K, generator=gen))
    dq, dr, dql, drl = qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()
    with pytest.warns(UserWarning):
        import xmh.dense as dense
        dense._warned_float = False
        got = float(cu.calc_map_k(dq, dr, dql, drl))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        cu.calc_map_k(dq, dr, dql, drl)
    dt = (time.perf_counter() - t0) / 3
    assert dt < 0.050, dt
    sub = slice(0, 24)
    got_sub = float(cu.calc_map_k(dq[sub], dr, dql[sub], drl))
    want_sub = float(orc.map_k(qB[sub], rB, qL[sub], rL, stable=True))
    assert abs(got_sub - want_sub) < 1e-5, (got_sub, want_sub)            # fp32 GEMM rounding (CPU sgemm vs v_mfma_f32) may swap near-ties
    assert 0.0 < got < 1.0
    got50 = float(cu.calc_map_k(dq[sub], dr, dql[sub], drl, 50))
    assert abs(got50 - float(orc.map_k(qB[sub], rB, qL[sub], rL, 50, stable=True))) < 1e-4
    # a code length that is no multiple of 4 floats (unaligned rows: the GEMM's general path), small set, exact against the oracle
    gen = torch.Generator().manual_seed(7)
    fq, fr = torch.tanh(torch.randn(19, 33, generator=gen) * 2), torch.tanh(torch.randn(777, 33, generator=gen) * 2)
    fqL, frL = (torch.rand(19, 9, generator=gen) < 0.3).long(), (torch.rand(777, 9, generator=gen) < 0.3).long()
    fqL[:, 0] = 1
    frL[::5, 0] = 1
    for kk in (None, 13):
        got33 = float(cu.calc_map_k(fq.cuda(), fr.cuda(), fqL.cuda(), frL.cuda(), kk))
        assert abs(got33 - float(orc.map_k(fq, fr, fqL, frL, kk, stable=True))) < 1e-5


@pytest.mark.parametrize("case", ["ternary_512", "bits_4096", "classes_300", "ternary_2048_classes_300"])
def test_drop_in_never_refuses_what_the_reference_evaluates(cu, case):
    """ternary codes above 256 bits, codes above 2048 bits, more than 256 classes: calc_map_k (reference common/calc_utils.py:58-92 has no
    such limits) returns the oracle's value through the float route instead of raising"""
    import xmh.dense as dense
    orc = _orc()
    gen = torch.Generator().manual_seed(len(case))
    K = {"ternary_512": 512, "bits_4096": 4096, "classes_300": 64, "ternary_2048_classes_300": 2048}[case]
    C = 300 if "classes_300" in case else 24
    Q, R = 21, 1500
    qB = _ternary_codes(Q, K, gen) if "ternary" in case else torch.randn(Q, K, generator=gen).sign()
    rB = _ternary_codes(R, K, gen) if "ternary" in case else torch.randn(R, K, generator=gen).sign()
    qL = (torch.rand(Q, C, generator=gen) < 0.05).to(torch.int64)
    rL = (torch.rand(R, C, generator=gen) < 0.05).to(torch.int64)
    qL[:, C - 1] = 1
    rL[::4, C - 1] = 1
    dense._warned_float = False
    with pytest.warns(UserWarning, match="float ranking path"):
        got = cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())
    assert got.device.type == "cpu" and got.dtype == torch.float32 and got.dim() == 0
    assert abs(float(got) - float(orc.map_k(qB, rB, qL, rL, stable=True))) < MAP_TOL
    got7 = cu.calc_map_k(qB, rB, qL, rL, 7)                                    # host tensors in, like the reference's callers
    assert abs(float(got7) - float(orc.map_k(qB, rB, qL, rL, 7, stable=True))) < MAP_TOL


def test_calc_map_k_argument_checks(cu):
    """k = 0 is the reference's nan (mean of an empty tensor, :81-89) on every path, k < 0 raises, mismatched label shapes
    raise instead of reaching the C ABI as raw pointers"""
    gen = torch.Generator().manual_seed(3)
    qB, rB = torch.randn(9, 64, generator=gen).sign().cuda(), torch.randn(700, 64, generator=gen).sign().cuda()
    qL = torch.ones(9, 24, dtype=torch.int64).cuda()
    rL = torch.ones(700, 24, dtype=torch.int64).cuda()
    for K in (64, 96):                                                          # the one-call ABI and the composed path
        assert torch.isnan(cu.calc_map_k(qB[:, :K] if K == 64 else torch.cat([qB, qB[:, :32]], 1), rB if K == 64 else torch.cat([rB, rB[:, :32]], 1), qL, rL, 0))
        with pytest.raises(ValueError):
            cu.calc_map_k(qB, rB, qL, rL, -3)
    with pytest.raises(ValueError):
        cu.calc_map_k(qB, rB, qL[:5], rL)
    with pytest.raises(ValueError):
        cu.calc_map_k(qB, rB, qL, rL[:, :20])
    with pytest.raises(ValueError):
        cu.calc_map_k(qB, rB[:600], qL, rL)


# ------------------------------------------------------------------------------------------------
# round 6: the lane-order assumption, hardened
# ------------------------------------------------------------------------------------------------
_PROBE_CHILD = r'''
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch
import bench
from xmh.common import calc_utils as cu
qB, qL, rB, rL = bench.synth(5000, 117218, 64, 80, seed=1814, p=0.04)
dq, dr, dql, drl = qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()
full = float(cu.calc_map_k(dq, dr, dql, drl))
sub = float(cu.calc_map_k(dq[:48], dr, dql[:48], drl))
k50 = float(cu.calc_map_k(dq[:48], dr, dql[:48], drl, 50))
print(json.dumps({"full": full, "sub": sub, "k50": k50}))
'''


def test_lane_order_probe_forced_to_fail_engages_the_fallback_at_full_shape():
    """XMH_SCAN_PROBE_FAULT makes the per-device probe (one wave alone + pass 2's geometry under load beside matrix waves) expect an order
    the hardware does not serve: it must fail, say so, and the whole configs[1] evaluation (Q 5000 x R 117 218 x 64 bit) must then run on
    the masked kernels with the same result -- against the unforced run and against the oracle on a query subsample."""
    import json
    import subprocess
    import sys
    code = _PROBE_CHILD % (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd"))
    runs = {}
    for fault in ("0", "1"):
        env = dict(os.environ, XMH_SCAN_PROBE_FAULT=fault)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[fault] = (json.loads(r.stdout.strip().splitlines()[-1]), r.stderr)
    assert "lane-order probe FAILED" not in runs["0"][1]                       # the real probe holds on this device, at load
    assert "lane-order probe FAILED" in runs["1"][1] and "forced" in runs["1"][1]
    for key in ("full", "sub", "k50"):
        assert abs(runs["0"][0][key] - runs["1"][0][key]) < 2e-7, (key, runs["0"][0], runs["1"][0])
    sys.path.insert(0, ROOT)
    import bench
    qB, qL, rB, rL = bench.synth(5000, 117218, 64, 80, seed=1814, p=0.04)
    want = float(_orc().map_k(qB[:48], rB, qL[:48], rL, stable=True))
    assert abs(runs["1"][0]["sub"] - want) < MAP_TOL and abs(runs["0"][0]["sub"] - want) < MAP_TOL


@pytest.mark.parametrize("K", [16, 64, 128, 256])
def test_scan_verify_mode_rederives_every_evaluation(cu, K, monkeypatch):
    """xmh_scan_verify(1): each unsharded evaluation is derived a second time by the masked VALU kernels and compared on the host; a
    planted disagreement (XMH_SCAN_VERIFY_FAULT = the query to spoil) surfaces as an error from the call"""
    from xmh import _lib
    orc = _orc()
    qB, rB, qL, rL = _synth(300, 20000, K, 80, seed=40 + K)
    want = float(orc.map_k(qB[:40], rB, qL[:40], rL, stable=True))
    _lib.scan_verify(True)
    try:
        assert abs(float(cu.calc_map_k(qB[:40].cuda(), rB.cuda(), qL[:40].cuda(), rL.cuda())) - want) < MAP_TOL
        float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), 100))
        monkeypatch.setenv("XMH_SCAN_VERIFY_FAULT", "7")
        with pytest.raises(RuntimeError, match="xmh_scan_verify"):
            cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda())
        monkeypatch.delenv("XMH_SCAN_VERIFY_FAULT")
        float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()))
    finally:
        _lib.scan_verify(False)


def test_scan_verify_mode_on_ternary_codes(cu):
    """the re-derivation of a ternary evaluation: matrix-core pass 1 + cached pass 2 against the masked VALU ternary kernels"""
    from xmh import _lib
    orc = _orc()
    gen = torch.Generator().manual_seed(77)
    for K in (64, 256):
        qB, rB = _ternary_codes(90, K, gen, 0.1), _ternary_codes(15000, K, gen, 0.1)
        qL = (torch.rand(90, 24, generator=gen) < 0.1).to(torch.int64)
        rL = (torch.rand(15000, 24, generator=gen) < 0.1).to(torch.int64)
        qL[:, 0] = 1
        rL[::3, 0] = 1
        _lib.scan_verify(True)
        try:
            got = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()))
        finally:
            _lib.scan_verify(False)
        assert abs(got - float(orc.map_k(qB, rB, qL, rL, stable=True))) < MAP_TOL


def test_scan_verify_mode_at_the_headline_shape(cu):
    import sys
    from xmh import _lib
    sys.path.insert(0, ROOT)
    import bench
    qB, qL, rB, rL = bench.synth(5000, 117218, 64, 80, seed=1814, p=0.04)
    dq, dr, dql, drl = qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()
    plain = float(cu.calc_map_k(dq, dr, dql, drl))
    _lib.scan_verify(True)
    try:
        assert float(cu.calc_map_k(dq, dr, dql, drl)) == plain                 # the verified call returns the fast derivation's value
    finally:
        _lib.scan_verify(False)


@pytest.mark.parametrize("K", [160, 256])
def test_scan_256_bit_pass2_on_float_bit_counters(xr, cu, K, monkeypatch):
    """round 6: 129..256-bit codes whose shard does not fit packed 32-bit counters run pass 2 on k_scan_ap_c<., 16> (float-bit counters on
    the two-byte pair cache) instead of the integer-counter k_scan_ap_s.  XMH_SCAN_PACK32=0 forces the 64-bit width at a test-sized shape;
    dense relevance, duplicate-heavy gallery (lanes collide), k in {None, 37}."""
    import ctypes
    from xmh import _lib
    orc = _orc()
    qB, rB, qL, rL = _synth(70, 21000, K, 24, seed=K, p=0.3)
    rB[9000:] = rB[torch.randint(0, 11, (12000,), generator=torch.Generator().manual_seed(K))]
    monkeypatch.setenv("XMH_SCAN_PACK32", "0")
    buf = ctypes.create_string_buffer(512)
    _lib.check(_lib.lib.xmh_scan_describe(70, 21000, 256, 24, 0, buf, 512), "xmh_scan_describe")
    assert b"k_scan_ap_c<false, 16, false>" in buf.value, buf.value
    for kk in (None, 37):
        got = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda(), kk))
        assert abs(got - float(orc.map_k(qB, rB, qL, rL, kk, stable=True))) < MAP_TOL
    monkeypatch.delenv("XMH_SCAN_PACK32")
    got = float(cu.calc_map_k(qB.cuda(), rB.cuda(), qL.cuda(), rL.cuda()))            # default: the device word picks the packed kernel here
    assert abs(got - float(orc.map_k(qB, rB, qL, rL, stable=True))) < MAP_TOL


def test_calc_map_k_workspace_keep_knob(cu):
    """the one piece of device memory the drop-in holds between calls has a knob"""
    gen = torch.Generator().manual_seed(2)
    qB, rB = torch.randn(40, 64, generator=gen).sign().cuda(), torch.randn(3000, 64, generator=gen).sign().cuda()
    qL, rL = torch.ones(40, 8, dtype=torch.int64).cuda(), torch.ones(3000, 8, dtype=torch.int64).cuda()
    a = float(cu.calc_map_k(qB, rB, qL, rL))
    assert cu._scan_ws.__dict__.get("entry") is not None
    old = cu.set_workspace_keep_bytes(0)
    try:
        assert cu._scan_ws.__dict__.get("entry") is None
        assert float(cu.calc_map_k(qB, rB, qL, rL)) == a and cu._scan_ws.__dict__.get("entry") is None
        q96 = torch.cat([qB, qB[:, :32]], 1)
        r96 = torch.cat([rB, rB[:, :32]], 1)
        cu.calc_map_k(q96, r96, qL, rL)                         # the composed path (three code words)
        assert cu._scan_ws.__dict__.get("entry") is None
    finally:
        cu.set_workspace_keep_bytes(old)
    cu.calc_map_k(q96, r96, qL, rL)
    assert cu._scan_ws.__dict__.get("entry") is not None
    cu.set_workspace_keep_bytes(1)
    assert cu._scan_ws.__dict__.get("entry") is None
    cu.set_workspace_keep_bytes(old)


@pytest.mark.parametrize("K", [16, 64, 128, 256])
def test_sharded_scan_of_ternary_codes(xr, K):
    """round 6: ternary codes run pass 1 on the matrix cores and the cached pass 2 of the 256-bit binary codes -- also per shard.  Three
    contiguous shards of very different sizes, duplicate-heavy gallery: explicit offsets (ap_sums), the totals-table form (map_sharded)
    and the all-to-all form (map_sharded_offsets) against the unsharded scan and the oracle."""
    from xmh import sharded
    orc = _orc()
    gen = torch.Generator().manual_seed(300 + K)
    Q, R, C, k = 130, 9000, 24, 37
    qB, rB = _ternary_codes(Q, K, gen, 0.05), _ternary_codes(40, K, gen, 0.05)[torch.randint(0, 40, (R,), generator=gen)]
    rB[: R // 3] = _ternary_codes(R // 3, K, gen, 0.2)
    qL = (torch.rand(Q, C, generator=gen) < 0.15).to(torch.int64)
    rL = (torch.rand(R, C, generator=gen) < 0.15).to(torch.int64)
    qL[:, 0] = 1
    rL[::3, 0] = 1
    q, ql = xr.pack_sign(qB.cuda()), xr.pack_labels(qL.cuda())
    r, rl = xr.pack_sign(rB.cuda()), xr.pack_labels(rL.cuda())
    assert q.zero is not None and r.zero is not None
    for kk in (None, k):
        want = float(orc.map_k(qB, rB, qL, rL, kk, stable=True))
        whole = xr.RankingScan(q, ql, r, rl, C)
        whole.histograms(False)
        ap_ref, cap_ref = whole.ap_sums(kk)
        assert abs(float((ap_ref / cap_ref.double()).mean()) - want) < MAP_TOL
        bounds = [0, 700, 6100, R]
        ops = [sharded.HipShardOps(q, ql, r.rows(bounds[s], bounds[s + 1]), rl[bounds[s]:bounds[s + 1]].contiguous(), C) for s in range(3)]
        gathered = torch.stack([torch.stack(o.histograms()) for o in ops]).contiguous()
        ap = torch.zeros(Q, dtype=torch.float64, device="cuda")
        for s, o in enumerate(ops):
            part, cap = o.ap_sums(kk, *o.offsets(gathered, s))
            assert torch.equal(cap, cap_ref)
            ap += part
        assert torch.allclose(ap, ap_ref, rtol=1e-6, atol=1e-9)
        tg = torch.stack([o.totals() for o in ops]).contiguous()
        got = sum(float(o.map_partial(kk, tg, s)) for s, o in enumerate(ops))
        assert abs(got - want) < MAP_TOL
