"""GPU parity of the encoder: HIP kernels (through the C ABI) against torch-CPU fp32 references of the same op,
the encoder oracle, and the goldens generated from the reference's own modules.

Tolerances (fp32 "parity mode", v_mfma_f32_32x32x2_f32): elementwise ops 1e-5 relative; whole 12-layer towers
1e-4 relative to the reference golden (different but equally valid fp32 summation orders); fp16 "fast mode" is
checked for the error level SURVEY H4 predicts, not for parity."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from xmh import ops as o
    o.set_precision("f32")
    return o


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def g_(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("M,N,K", [(5000, 768, 768), (128, 128, 16), (37, 70, 50), (1, 1, 1), (200, 2304, 768), (130, 129, 3072), (64, 512, 3)])
@pytest.mark.parametrize("mode", ["f32", "f32x"])
def test_gemm_f32_matches_torch(ops, M, N, K, mode):
    prec = ops._NAMES[mode]            # parity mode (split on the fp16 MFMA where the shape allows) and exact fp32 MFMA
    A, W = torch.randn(M, K, generator=g_(1)), torch.randn(N, K, generator=g_(2)) * 0.1
    bias, res = torch.randn(N, generator=g_(3)), torch.randn(M, N, generator=g_(4))
    want = A.double() @ W.double().t()
    got = ops.gemm_nt(A.cuda(), W.cuda(), precision=prec)
    assert rel(got, want) < 5e-6
    for act, fn in ((ops.ACT_QUICKGELU, lambda x: x * torch.sigmoid(1.702 * x)), (ops.ACT_GELU_ERF, F.gelu), (ops.ACT_TANH, torch.tanh),
                    (ops.ACT_RELU, torch.relu)):
        got = ops.gemm_nt(A.cuda(), W.cuda(), bias.cuda(), residual=res.cuda(), act=act, precision=prec)
        assert rel(got, fn((want + bias.double()).float()).double() + res.double()) < 1e-5
    # strided views (leading dimension > K) and in-place residual
    Abig = torch.randn(M, K + 8, generator=g_(5)).cuda()
    out = res.clone().cuda()
    ops.gemm_nt(Abig[:, :K], W.cuda(), residual=out, out=out, precision=prec)
    assert rel(out, Abig[:, :K].cpu().double() @ W.double().t() + res.double()) < 5e-6   # fp32 chain over K<=3072


def test_gemm_f32_transpose_detecting(ops):
    """identity A against an asymmetric W (guide rule 16): C must equal W^T, not W."""
    n = 96
    Wt = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 128.0      # 14 significant bits: exact in every precision mode but f16
    for prec in (None, ops.PREC_F32X):
        got = ops.gemm_nt(torch.eye(n).cuda(), Wt.cuda(), precision=prec)
        assert torch.equal(got.cpu(), Wt.t())


@pytest.mark.parametrize("M,N,K", [(5000, 768, 768), (200, 2304, 768), (77, 130, 3072), (1, 64, 32), (129, 512, 96), (20001, 2304, 768)])   # last: 128x256 tiles
def test_gemm_split16_parity_mode_on_fp16_exact_weights(ops, M, N, K):
    """parity mode routes fp16-exact weights (every CLIP weight) through the hi/lo split kernel: fp32-grade error over a wide
    dynamic range of activations, bias/activation/residual epilogues, strided A."""
    from xmh import _lib
    gen = g_(7)
    A = torch.randn(M, K, generator=gen) * torch.exp(torch.randn(M, K, generator=gen) * 2.0)          # ~1e-4 .. 1e4, inside the fp16 range
    W = (torch.randn(N, K, generator=g_(8)) * 0.05).half().float()
    bias, res = torch.randn(N, generator=g_(3)), torch.randn(M, N, generator=g_(4))
    want = A.double() @ W.double().t()
    _lib.prof_enable(True)
    got = ops.gemm_nt(A.cuda(), W.cuda())
    torch.cuda.synchronize()
    assert _lib.prof_read("gemm_s16")[1] >= 1                 # the split kernel ran
    _lib.prof_enable(False)
    assert rel(got, want) < 2e-6
    exact = ops.gemm_nt(A.cuda(), W.cuda(), precision=ops.PREC_F32X)
    assert rel(exact, want) < 5e-6 and rel(got, want) < 2 * rel(exact, want) + 1e-7      # not worse than the fp32 MFMA
    A1 = torch.randn(M, K, generator=g_(9))                   # O(1) pre-activations: the epilogue check is well conditioned
    got = ops.gemm_nt(A1.cuda(), W.cuda(), bias.cuda(), residual=res.cuda(), act=ops.ACT_TANH)
    assert rel(got, torch.tanh(A1.double() @ W.double().t() + bias.double()) + res.double()) < 2e-6
    Abig = torch.randn(M, K + 8, generator=g_(5)).cuda()
    assert rel(ops.gemm_nt(Abig[:, :K], W.cuda()), Abig[:, :K].cpu().double() @ W.double().t()) < 2e-6
    huge = A.clone()
    huge[0, 0] = 3.0e5                                         # outside the domain: saturates, never inf - inf
    assert torch.isfinite(ops.gemm_nt(huge.cuda(), W.cuda())).all()


@pytest.mark.parametrize("M,N,K", [(5000, 768, 768), (300, 256, 768), (77, 130, 3072), (1, 64, 32)])
def test_gemm_parity_mode_splits_inexact_weights_too(ops, M, N, K):
    """weights that are NOT fp16-exact (fine-tuned in fp32: hash heads, a trained backbone) are split hi/lo as well: three
    fp16 MFMAs per product, still fp32-grade."""
    from xmh import _lib
    A = (torch.randn(M, K, generator=g_(1)) * torch.exp(torch.randn(M, K, generator=g_(2)))).cuda()
    W = (torch.randn(N, K, generator=g_(3)) * 0.1).cuda()                                                         # full fp32 mantissas
    bias = torch.randn(N, generator=g_(4)).cuda()
    want = A.cpu().double() @ W.cpu().double().t() + bias.cpu().double()
    _lib.prof_enable(True)
    got = ops.gemm_nt(A, W, bias)
    torch.cuda.synchronize()
    assert _lib.prof_read("gemm_s16")[1] >= 1 and _lib.prof_read("gemm_f32")[1] == 0       # not the exact-MFMA kernel
    _lib.prof_enable(False)
    exact = ops.gemm_nt(A, W, bias, precision=ops.PREC_F32X)
    assert rel(got, want) < 2e-6 and rel(got, want) < 2 * rel(exact, want) + 1e-7
    Wc = W.clone()
    a = ops.gemm_nt(A, Wc, bias)
    Wc += 1e-3                                                 # in-place update: the cached fp16 parts must not be reused
    b = ops.gemm_nt(A, Wc, bias)
    assert rel(b, A.cpu().double() @ Wc.cpu().double().t() + bias.cpu().double()) < 2e-6 and not torch.equal(a, b)


def test_gemm_fuzz_shapes_modes_epilogues(ops):
    """30 seeded random (M, N, K, mode, weight kind, epilogue): ragged tiles, K with and without the 32-alignment the MFMA fp16
    paths need (the wrapper must fall back), fp16-exact and full-mantissa weights, strided A, in-place residual."""
    rng = np.random.default_rng(5)
    acts = {ops.ACT_NONE: lambda x: x, ops.ACT_QUICKGELU: lambda x: x * torch.sigmoid(1.702 * x), ops.ACT_GELU_ERF: F.gelu,
            ops.ACT_TANH: torch.tanh, ops.ACT_RELU: torch.relu}
    for case in range(30):
        M, N = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        K = int(rng.choice([32, 64, 96, 160, 768, 1024])) if case % 3 else int(rng.integers(1, 300))
        mode = ["f32", "f32x", "f16"][case % 3 if case % 7 else 0]
        gen = g_(100 + case)
        A = torch.randn(M, K + 8, generator=gen)[:, :K] if case % 4 == 0 else torch.randn(M, K, generator=gen)
        W = torch.randn(N, K, generator=gen) * 0.1
        if case % 2:
            W = W.half().float()
        bias = torch.randn(N, generator=gen) if case % 5 else None
        res = torch.randn(M, N, generator=gen) if case % 3 == 1 else None
        act = list(acts)[case % 5]
        pre = A.double() @ W.double().t() + (0 if bias is None else bias.double())
        want = acts[act](pre.float()).double() + (0 if res is None else res.double())
        out = None if res is None else res.clone().cuda()
        got = ops.gemm_nt(A.cuda(), W.cuda(), None if bias is None else bias.cuda(), residual=out, act=act, out=out, precision=ops._NAMES[mode])
        tol = 3e-3 if mode == "f16" else 1e-5
        assert rel(got, want) < tol, (case, M, N, K, mode)


def test_gemm_f16_fast_mode_error_level(ops):
    A, W = torch.randn(700, 768, generator=g_(1)), (torch.randn(512, 768, generator=g_(2)) * 0.05).half().float()
    want = A.double() @ W.double().t()
    got = ops.gemm_nt(A.cuda(), W.cuda(), precision=ops.PREC_F16)
    assert 1e-6 < rel(got, want) < 3e-3                      # activations rounded to fp16, fp32 accumulate
    got_h = ops.gemm_nt(A.half().float().cuda(), W.cuda(), precision=ops.PREC_F16)
    assert rel(got_h, A.half().double() @ W.double().t()) < 2e-6      # fp16-exact inputs: only accumulation order differs


def test_layernorm_and_rowwise_ops(ops):
    x = torch.randn(333, 768, generator=g_(1)) * 3 + 0.5
    w, b = torch.randn(768, generator=g_(2)), torch.randn(768, generator=g_(3))
    assert rel(ops.layernorm(x.cuda(), w.cuda(), b.cuda()), F.layer_norm(x, (768,), w, b, 1e-5)) < 2e-6
    x5 = torch.randn(7, 512, generator=g_(4))
    assert rel(ops.layernorm(x5.cuda(), w[:512].cuda(), b[:512].cuda()), F.layer_norm(x5, (512,), w[:512], b[:512], 1e-5)) < 2e-6
    mean, var = torch.randn(512, generator=g_(5)), torch.rand(512, generator=g_(6)) + 0.5
    want = F.batch_norm(x5, mean, var, w[:512], b[:512], False, 0.0, 1e-5)
    assert rel(ops.affine_cols(x5.cuda(), mean.cuda(), var.cuda(), w[:512].cuda(), b[:512].cuda()), want) < 2e-6
    r = torch.relu(torch.randn(9, 128, generator=g_(7)))
    r[0, 0:2] = 0.0                                                             # exact tie -> (0.5, 0.5)
    got = ops.pair_softmax(r.cuda()).cpu()
    assert rel(got, torch.softmax(r.view(9, -1, 2), -1).view(9, -1)) < 1e-6 and got[0, 0] == 0.5 and got[0, 1] == 0.5


@pytest.mark.parametrize("B,L,H,causal,masked", [(3, 50, 12, False, False), (4, 32, 8, True, False), (4, 32, 8, True, True), (2, 64, 8, False, False),
                                                  (2, 100, 8, False, False), (1, 1, 8, True, False)])
def test_attention_matches_torch(ops, B, L, H, causal, masked):
    D = 64 * H
    qkv = torch.randn(B, L, 3 * D, generator=g_(L))
    kpm = None
    if masked:
        kpm = torch.zeros(B, L, dtype=torch.bool)
        for b in range(B):
            kpm[b, 5 + 3 * b:] = True
    q, k, v = [t.view(B, L, H, 64).transpose(1, 2) for t in qkv.chunk(3, -1)]
    s = (q / 8.0) @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, L, D)
    got = ops.attention(qkv.cuda(), H, causal=causal, key_padding_mask=None if kpm is None else kpm.cuda())
    assert rel(got, want) < 3e-6


def test_attention_fuzz_lengths_and_masks(ops):
    """25 seeded random (B, L, H, causal, padding mask): both kernels (MFMA for L <= 64, VALU above), L off the 32-row blocks."""
    rng = np.random.default_rng(9)
    for case in range(25):
        B, H = int(rng.integers(1, 5)), int(rng.choice([1, 8, 12]))
        L = int(rng.choice([1, 2, 31, 32, 33, 50, 63, 64, 65, 77, 128]))
        causal = bool(case % 2)
        D = 64 * H
        qkv = torch.randn(B, L, 3 * D, generator=g_(300 + case)) * float(rng.choice([0.3, 1.0, 3.0]))
        kpm = None
        if case % 3 == 0 and L > 2:
            kpm = torch.zeros(B, L, dtype=torch.bool)
            for b in range(B):
                kpm[b, int(rng.integers(1, L)):] = True               # key 0 always stays: no fully masked row
        q, k, v = [t.view(B, L, H, 64).transpose(1, 2) for t in qkv.chunk(3, -1)]
        sc = (q / 8.0) @ k.transpose(-1, -2)
        if causal:
            sc = sc + torch.full((L, L), float("-inf")).triu_(1)
        if kpm is not None:
            sc = sc.masked_fill(kpm[:, None, None, :], float("-inf"))
        want = (torch.softmax(sc.double(), -1) @ v.double()).transpose(1, 2).reshape(B, L, D)
        got = ops.attention(qkv.cuda(), H, causal=causal, key_padding_mask=None if kpm is None else kpm.cuda())
        assert rel(got, want) < 5e-6, (case, B, L, H, causal)


def test_patch_and_embedding_kernels(ops):
    img = torch.randn(3, 3, 224, 224, generator=g_(1))
    w = torch.randn(40, 3, 32, 32, generator=g_(2)) * 0.02
    cols = ops.im2col_patch(img.cuda(), 32)
    want = F.conv2d(img, w, stride=32).reshape(3, 40, 49).permute(0, 2, 1).reshape(147, 40)
    assert rel(ops.gemm_nt(cols, w.reshape(40, -1).cuda()), want) < 3e-6
    D = 768
    patches, cls, pos = torch.randn(3 * 49, D, generator=g_(3)), torch.randn(D, generator=g_(4)), torch.randn(50, D, generator=g_(5))
    gam, bet = torch.randn(D, generator=g_(6)), torch.randn(D, generator=g_(7))
    x = torch.cat([cls.expand(3, 1, D), patches.view(3, 49, D)], 1) + pos
    assert rel(ops.vit_assemble(patches.cuda(), cls.cuda(), pos.cuda(), gam.cuda(), bet.cuda(), 3, 49), F.layer_norm(x, (D,), gam, bet, 1e-5)) < 3e-6
    from xmh.models import weights as W
    ids, _ = W.synth_text(5, 6)
    tok, tpos = torch.randn(49408, 64, generator=g_(8)), torch.randn(77, 64, generator=g_(9))
    xe, eos = ops.text_embed(ids.cuda(), tok.cuda(), tpos.cuda())
    assert torch.equal(xe.cpu(), tok[ids] + tpos[:32]) and torch.equal(eos.cpu().long(), ids.argmax(-1))
    assert torch.equal(ops.gather_rows(xe, group=32, idx=eos).cpu(), xe.cpu()[torch.arange(6), ids.argmax(-1)])
    assert torch.equal(ops.gather_rows(xe, group=32, offset=0).cpu(), xe.cpu()[:, 0])


@pytest.fixture(scope="module")
def clip_models(ops):
    from xmh.models import weights as W
    from xmh.models.clip import build_model
    g = np.load(os.path.join(GOLDEN, "encode_clip_b2.npz"))
    seed = int(g["seed"])
    return g, W, build_model(W.synth_clip_state_dict(seed)).cuda(), build_model(W.synth_clip_state_dict(seed), return_patches=True).cuda()


def test_clip_towers_match_reference_goldens(ops, clip_models):
    g, W, m, m_rp = clip_models
    seed = int(g["seed"])
    image, (ids, pad) = W.synth_images(seed, 2).cuda(), W.synth_text(seed, 2)
    assert rel(m.encode_image(image), torch.from_numpy(g["img_cls"])) < 1e-4
    assert rel(m.encode_text(ids.cuda()), torch.from_numpy(g["txt_eos"])) < 1e-4
    cls, tok, _ = m_rp.encode_image(image)
    assert rel(cls, torch.from_numpy(g["img_cls_rp"])) < 1e-4 and rel(tok, torch.from_numpy(g["img_tokens_rp"])) < 1e-4
    eos, ttok, _, nm = m_rp.encode_text(ids.cuda(), key_padding_mask=pad.cuda())
    assert rel(eos, torch.from_numpy(g["txt_eos_rp"])) < 1e-4
    assert np.array_equal(nm.cpu().numpy(), g["txt_mask_rp"])
    keep = torch.from_numpy(~g["txt_mask_rp"].T)
    assert rel(ttok.cpu()[keep], torch.from_numpy(g["txt_tokens_rp"])[keep]) < 1e-4


def test_whole_tower_entry_points_equal_the_primitive_chain(ops, clip_models):
    """xmh_vit_b32_forward / xmh_text_forward / xmh_clip_blocks_forward enqueue the same kernels in the same order as the
    per-primitive chain driven from Python: results must be identical bit for bit, in every precision and both token modes."""
    import xmh.models.clip as C
    g, W, m, m_rp = clip_models
    image, (ids, pad) = W.synth_images(3, 5).cuda(), W.synth_text(3, 5)
    ids, pad = ids.cuda(), pad.cuda()

    def run_all():
        outs = [m.encode_image(image), m.encode_text(ids), m.encode_text(ids, key_padding_mask=pad)]
        cls, tok, _ = m_rp.encode_image(image)
        eos, ttok, _, nm = m_rp.encode_text(ids, key_padding_mask=pad)
        eos2, ttok2, _, nm2 = m_rp.encode_text(ids)
        assert nm2 is None
        return outs + [cls, tok, eos, ttok, nm, eos2, ttok2]

    before = ops.get_precision()
    try:
        for prec in ("f32", "f32x", "f16"):
            ops.set_precision(prec)
            assert C.NATIVE_FORWARD
            native = [t.clone() for t in run_all()]
            C.NATIVE_FORWARD = False
            try:
                chain = run_all()
            finally:
                C.NATIVE_FORWARD = True
            for a, b in zip(native, chain):
                assert a.shape == b.shape and torch.equal(a, b), prec
    finally:
        ops.set_precision(before)


def test_text_tower_without_its_padding_rows_is_bit_identical(ops, clip_models):
    """xmh_text_forward_packed (round 4): the text tower on the rows up to each caption's EOS only.  Under the causal mask nothing behind
    EOS reaches the EOS row, and every kept row runs through the same kernels, so the embeddings equal the padded call's bit for bit --
    in all three precisions, for captions of every length from 1 token (a degenerate all-padding row: argmax = 0) to the full 32, and
    for a batch large enough that the packed GEMMs pick other tiles than the padded ones."""
    import xmh.models.clip as C
    g, W, m, _ = clip_models
    gen = torch.Generator().manual_seed(77)
    B, L = 203, 32
    ids = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        n = [0, 1, 30, 29][b] if b < 4 else int(torch.randint(2, 30, (1,), generator=gen))       # tokens between SOS and EOS
        if b == 0:
            continue                                                                             # all zeros: length 1
        ids[b, 0] = 49406
        ids[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=gen)
        ids[b, 1 + n] = 49407
    ids = ids.cuda()
    before = ops.get_precision()
    assert C.TEXT_PACKING
    try:
        for prec in ("f32", "f32x", "f16"):
            ops.set_precision(prec)
            packed = m.encode_text(ids).clone()
            C.TEXT_PACKING = False
            try:
                padded = m.encode_text(ids)
            finally:
                C.TEXT_PACKING = True
            assert torch.equal(packed, padded), prec
            small = m.encode_text(ids[:5].contiguous())                                          # another batch size, other tiles: same rows
            assert torch.equal(small, padded[:5]), prec
    finally:
        ops.set_precision(before)
    # the entry point's argument checks
    from xmh._lib import lib
    assert lib.xmh_text_forward_packed(None, None, None, 0, 4, 32, 0, None, None, 0, None) != 0


def test_packed_text_tower_with_key_padding_mask_and_token_outputs(ops, clip_models):
    """xmh_text_forward_packed_dev (round 5): caption lengths counted on the device, key_padding_mask applied to the keys, token outputs in
    the reference's padded layout (MITH's call: models/MITH/MITH.py:59-66 -> models/CLIP/model.py:373-396 with return_patches).  Against the
    padded xmh_text_forward: the EOS embeddings and every KEPT token row bit for bit; the dropped rows -- all hidden by the mask, behind
    the last visible position -- zero.  Masks: the dataset's (ids == 0), one that hides a token in front of EOS, one that leaves a padding
    position visible, none of which the packing may get wrong; and the EOS-only output with a mask."""
    import xmh.models.clip as C
    g, W, m, m_rp = clip_models
    gen = torch.Generator().manual_seed(78)
    B, L = 67, 32
    ids = torch.zeros(B, L, dtype=torch.int64)
    for b in range(B):
        n = [1, 30, 29, 2][b] if b < 4 else int(torch.randint(2, 30, (1,), generator=gen))
        ids[b, 0] = 49406
        ids[b, 1:1 + n] = torch.randint(1, 49405, (n,), generator=gen)
        ids[b, 1 + n] = 49407
    kpm = ids == 0
    kpm[5, 2] = True                                              # a hidden token in front of EOS
    eos = ids.argmax(dim=1)
    kpm[6, min(L - 1, int(eos[6]) + 3)] = False                   # a visible position behind EOS: the kept rows reach that far
    kpm[7, :] = False                                             # no padding hidden at all: every row is kept
    ids, kpm = ids.cuda(), kpm.cuda()
    last_visible = torch.where(~kpm, torch.arange(L, device="cuda").expand(B, L), torch.full((B, L), -1, device="cuda")).max(dim=1).values
    length = torch.maximum(ids.argmax(dim=1), last_visible) + 1   # rows kept per caption
    kept = (torch.arange(L, device="cuda")[None, :] < length[:, None]).T      # [L, B]: the token outputs are LND
    assert int(length[6]) == min(L, int(eos[6]) + 4) and int(length[7]) == L and int(length[0]) == 3
    before = ops.get_precision()
    assert C.TEXT_PACKING
    try:
        for prec in ("f32", "f16"):
            ops.set_precision(prec)
            e1, t1, _, m1 = m_rp.encode_text(ids, key_padding_mask=kpm, masked_rows="zero")
            e1, t1 = e1.clone(), t1.clone()
            e0, t0, _, m0 = m_rp.encode_text(ids, key_padding_mask=kpm)                  # "exact": the padded call
            assert torch.equal(m0, m1) and torch.equal(e0, e1), prec
            assert torch.equal(t1[kept], t0[kept]), prec
            assert not bool(t1[~kept].any()) and bool(t0[~kept].any()), prec
            assert not bool((~m1.T)[~kept].any())                 # every dropped row is one the returned mask hides
            # without token outputs: the EOS embedding under a mask, packed against padded
            p1 = m.encode_text(ids, key_padding_mask=kpm).clone()
            C.TEXT_PACKING = False
            try:
                p0 = m.encode_text(ids, key_padding_mask=kpm)
            finally:
                C.TEXT_PACKING = True
            assert torch.equal(p0, p1), prec
    finally:
        ops.set_precision(before)
    from xmh._lib import lib
    assert lib.xmh_text_forward_packed_dev(None, None, None, 4, 32, 0, None, None, None, 0, None) != 0


def _vit_front(m, image):
    """the image tower up to the block stack, through the primitives (VisionTransformer.run's chain)"""
    from xmh import ops as o
    v = m.visual
    width, n_patches = v.conv1.weight.shape[0], v.positional_embedding.shape[0] - 1
    patches = o.gemm_nt(o.im2col_patch(image, v.patch_size), v.conv1.weight.reshape(width, -1))
    return o.vit_assemble(patches, v.class_embedding, v.positional_embedding, v.ln_pre.weight, v.ln_pre.bias, image.shape[0], n_patches)


def test_saved_activation_forward_matches_reference_hooks_oracle_and_the_plain_forward(ops, clip_models):
    """xmh_clip_blocks_forward_saved: the per-layer records against (1) forward hooks inside the reference's own
    ResidualAttentionBlocks (tests/golden/encode_saved_b2.npz), (2) the oracle's restatement for every field of every layer,
    qkv and the pre-out_proj attention output included, (3) the plain forward, whose output it must reproduce bit for bit in
    every precision."""
    from oracle import encode as enc
    from test_oracle_encode import saved_subset
    g, W, m, _ = clip_models
    gs = np.load(os.path.join(GOLDEN, "encode_saved_b2.npz"))
    seed = int(gs["seed"])
    image = W.synth_images(seed, 2)
    T = m.visual.transformer
    before = ops.get_precision()
    try:
        for prec, tol in (("f32", 1e-4), ("f32x", 1e-4), ("f16", None)):
            ops.set_precision(prec)
            x0 = _vit_front(m, image.cuda())
            plain = T.run(x0.clone())
            y, saved = T.run_saved(x0.clone())
            assert torch.equal(y, plain), prec
            assert len(saved) == 12 and saved[0]["qkv"].shape == (2, 50, 2304) and saved[0]["fc_act"].shape == (2, 50, 3072)
            assert torch.equal(saved[0]["x_in"], x0)
            for i in range(11):                                    # the residual stream hops from record to record
                assert saved[i + 1]["x_in"].data_ptr() == saved[i]["fc_act"].data_ptr() + saved[i]["fc_act"].numel() * 4
            if tol is None:
                continue
            lnd = [{k: v.permute(1, 0, 2).cpu() for k, v in rec.items()} for rec in saved]            # reference layout [L, B, n]
            got = saved_subset(lnd, lambda li: lnd[li + 1]["x_in"] if li + 1 < 12 else y.permute(1, 0, 2).cpu(), gs)
            for name, v in got.items():
                assert rel(torch.from_numpy(v), torch.from_numpy(gs[name])) < tol, (prec, name)
            sd = enc.fp16_round_like_reference(W.synth_clip_state_dict(seed))
            with torch.no_grad():
                _, want = enc.blocks_saved(enc.vit_front(sd, image).permute(1, 0, 2), sd, "visual.transformer.", 12, 12, None)
            for li in (0, 6, 11):
                for name in want[li]:
                    assert rel(lnd[li][name], want[li][name]) < tol, (prec, li, name)
    finally:
        ops.set_precision(before)


def test_saved_activation_forward_text_tower_masks_and_bad_arguments(ops, clip_models):
    import xmh.models.clip as C
    from oracle import encode as enc
    from xmh._lib import XmhError, lib
    g, W, m, _ = clip_models
    seed = int(g["seed"])
    ids, pad = W.synth_text(seed, 3)
    sd = enc.fp16_round_like_reference(W.synth_clip_state_dict(seed))
    x0 = (sd["token_embedding.weight"][ids] + sd["positional_embedding"][:ids.shape[1]]).float()
    T = m.transformer
    plain = T.run(x0.cuda().clone(), causal=True, key_padding_mask=pad.cuda())
    y, saved = T.run_saved(x0.cuda().clone(), causal=True, key_padding_mask=pad.cuda())
    assert torch.equal(y, plain)
    L = ids.shape[1]
    mask = torch.full((L, L), float("-inf")).triu_(1)[None].repeat(3, 1, 1).masked_fill(pad[:, None, :].bool(), float("-inf"))
    with torch.no_grad():
        _, want = enc.blocks_saved(x0.permute(1, 0, 2), sd, "transformer.", 12, 8, mask)
    keep = ~pad                                                    # padded query rows are unspecified (SURVEY: masked rows)
    for li in (0, 11):
        for name in want[li]:
            assert rel(saved[li][name].cpu()[keep], want[li][name].permute(1, 0, 2)[keep]) < 1e-4, (li, name)
    assert lib.xmh_clip_saved_bytes(2, 50, 768, 12) == 12 * 16 * 2 * 50 * 768 * 4 and lib.xmh_clip_saved_bytes(0, 50, 768, 12) == 0
    x = torch.zeros(1, 4, 64, device="cuda")
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    blk = (C._lib.ClipBlock * 1)()
    with pytest.raises(XmhError):                                  # saved buffer too small
        C.check(lib.xmh_clip_blocks_forward_saved(blk, 1, 64, 1, C.ptr(x), 1, 4, 0, None, 0, C.ptr(ws), ws.numel(), C.ptr(x), 16,
                                                  C.current_stream()), "saved")
    with pytest.raises(XmhError):                                  # no saved buffer
        C.check(lib.xmh_clip_blocks_forward_saved(blk, 1, 64, 1, C.ptr(x), 1, 4, 0, None, 0, C.ptr(ws), ws.numel(), None, 1 << 30,
                                                  C.current_stream()), "saved")
    with pytest.raises(ValueError):
        T.run_saved(torch.zeros(2, 4, 512))


def test_whole_tower_entry_points_follow_weight_updates_and_reject_bad_input(ops, clip_models):
    import copy
    import xmh.models.clip as C
    from xmh._lib import XmhError, lib
    g, W, m, _ = clip_models
    image = W.synth_images(4, 2).cuda()
    m2 = copy.deepcopy(m)                                          # descriptors live outside the module: deep copies work
    a = m2.encode_image(image)
    with torch.no_grad():
        m2.visual.ln_post.weight.mul_(2.0)                         # in-place update -> descriptor is rebuilt
    b = m2.encode_image(image)
    C.NATIVE_FORWARD = False
    try:
        want = m2.encode_image(image)
    finally:
        C.NATIVE_FORWARD = True
    assert torch.equal(b, want) and not torch.equal(a, b)
    with pytest.raises(ValueError):
        m.encode_image(image[:, :, :100, :100])
    assert lib.xmh_clip_workspace_bytes(0, 50, 768, 3072, 0, 0) == 0
    x = torch.zeros(1, 4, 64, device="cuda")
    blk = (C._lib.ClipBlock * 1)()
    with pytest.raises(XmhError):                                  # precision out of range
        C.check(lib.xmh_clip_blocks_forward(blk, 1, 64, 1, C.ptr(x), 1, 4, 0, None, 7, C.ptr(x), 16, C.current_stream()), "blocks")
    with pytest.raises(XmhError):                                  # workspace too small
        C.check(lib.xmh_clip_blocks_forward(blk, 1, 64, 1, C.ptr(x), 1, 4, 0, None, 0, C.ptr(x), 16, C.current_stream()), "blocks")


def test_head_entry_points_equal_the_primitive_chain_and_pack(ops):
    """xmh_head_dcmht / xmh_head_dsph against the per-primitive chain (bit for bit), and their packed outputs against
    xmh_pack_pair_argmax / xmh_pack_sign on the float outputs, including the row scatter."""
    import ctypes
    import xmh.models.clip as C
    from xmh import retrieval as xr
    from xmh._lib import check, current_stream, lib, ptr
    from xmh.models import heads
    g = torch.Generator().manual_seed(11)
    emb = torch.randn(37, 512, generator=g).cuda()
    before = ops.get_precision()
    try:
        for prec in ("f32", "f32x", "f16"):
            ops.set_precision(prec)
            for K in (16, 64, 512):
                torch.manual_seed(K)
                layer = heads.DCMHTHashLayer(512, K).cuda().eval()
                with torch.no_grad():
                    layer.img_hash.norm.running_mean.normal_()
                    layer.img_hash.norm.running_var.uniform_(0.5, 2.0)
                dsph = heads.DSPHHashLayer(512, K).cuda().eval()
                for mod in (layer.img_hash, layer.txt_hash, dsph.img_hash):
                    native = mod(emb)
                    C.NATIVE_FORWARD = False
                    try:
                        chain = mod(emb)
                    finally:
                        C.NATIVE_FORWARD = True
                    assert torch.equal(native, chain), (prec, K, type(mod).__name__)
                # packed outputs, scattered to permuted rows
                perm = torch.randperm(37, generator=g).cuda()
                nbytes = lib.xmh_head_workspace_bytes(37, 512, ops._precision)
                ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
                desc, precision = C._cached_desc(layer.txt_hash, layer.txt_hash._desc,
                                                 params=list(layer.txt_hash.parameters()) + list(layer.txt_hash.buffers()), slot="dcmht")
                got = xr.empty_packed(37, K, emb.device)
                check(lib.xmh_head_dcmht(ctypes.byref(desc), ptr(emb), 37, precision, None, ptr(got.bits), ptr(perm), ptr(ws), nbytes,
                                         current_stream()), "xmh_head_dcmht")
                want = xr.pack_pair_argmax(layer.txt_hash(emb), row_index=perm, out=xr.empty_packed(37, K, emb.device))
                assert torch.equal(got.bits, want.bits)
                fdesc, precision = C._cached_desc(dsph.txt_hash, lambda p, keep: C._linear_desc(dsph.txt_hash.fc.weight, dsph.txt_hash.fc.bias, p, keep), slot="dsph")
                got = xr.empty_packed(37, K, emb.device, with_zero=True)
                flags = torch.zeros(1, dtype=torch.int32, device="cuda")
                check(lib.xmh_head_dsph(ctypes.byref(fdesc), ptr(emb), 37, precision, None, ptr(got.bits), ptr(got.zero), ptr(flags), ptr(perm),
                                        ptr(ws), nbytes, current_stream()), "xmh_head_dsph")
                want = xr.empty_packed(37, K, emb.device, with_zero=True)
                xr.pack_sign(dsph.txt_hash(emb), out=want, row_index=perm, flags=torch.zeros(1, dtype=torch.int32, device="cuda"))
                assert torch.equal(got.bits, want.bits) and torch.equal(got.zero, want.zero)
    finally:
        ops.set_precision(before)


@pytest.mark.parametrize("mode,M,N,K", [("f16", 4096, 3072, 768),       # 256 x 256 tiles (8 waves)
                                         ("f16", 5000, 2304, 768),       # 192 x 128
                                         ("f16", 5000, 768, 3072),       # 128 x 128, BK 64
                                         ("f16", 300, 512, 96),          # BK 32 (K % 64 != 0), 64-row tiles
                                         ("f32", 5000, 2304, 768),       # 128 x 192, two activation planes
                                         ("f32", 5000, 768, 3072),       # 128 x 128, BK 64, one block per CU
                                         ("f32", 5000, 3072, 768),       # 128 x 256 (8 waves)
                                         ("f32", 3200, 512, 2048)])      # 64-row tiles
def test_gemm_tile_shapes_share_one_k_order(ops, mode, M, N, K):
    """Every tile shape of the fp16-MFMA GEMM (k_gemm_g16) walks k in the same order: the first rows of a large product -- which
    picks the large-grid tile of its mode -- equal the same rows computed alone (a small grid, another tile shape) BIT FOR BIT,
    with bias + QuickGELU + residual in the epilogue; and the large product matches float64."""
    gen = g_(M + N + K)
    A = torch.randn(M, K, generator=gen)
    W = (torch.randn(N, K, generator=gen) * 0.05).half().float()
    bias, res = torch.randn(N, generator=gen), torch.randn(M, N, generator=gen)
    prec = ops._NAMES[mode]
    Ad, Wd, bd, rd = A.cuda(), W.cuda(), bias.cuda(), res.cuda()
    big = ops.gemm_nt(Ad, Wd, bd, residual=rd, act=ops.ACT_QUICKGELU, precision=prec)
    rows = 130
    small = ops.gemm_nt(Ad[:rows].contiguous(), Wd, bd, residual=rd[:rows].contiguous(), act=ops.ACT_QUICKGELU, precision=prec)
    assert torch.equal(big[:rows], small)
    Aref = A.half().double() if mode == "f16" else A.double()
    pre = (Aref[:512] @ W.double().t() + bias.double()).float()
    want = (pre * torch.sigmoid(1.702 * pre)).double() + res[:512].double()
    assert rel(big[:512], want) < (2e-5 if mode == "f16" else 3e-6)


def test_clip_state_dict_keys_are_the_reference_contract(clip_models):
    _, W, m, _ = clip_models
    want = set(W.synth_clip_state_dict(1).keys())
    assert set(m.state_dict().keys()) == want                     # SURVEY 8c weight-file contract


def test_clip_batch_100_matches_oracle_on_a_subsample(ops, clip_models):
    """BASELINE configs[1] batch size (100): per-sample results must not depend on the batch around them."""
    from oracle import encode as enc
    g, W, m, _ = clip_models
    seed = int(g["seed"])
    image, (ids, _) = W.synth_images(7, 100), W.synth_text(7, 100)
    sd = enc.fp16_round_like_reference(W.synth_clip_state_dict(seed))
    pick = [0, 41, 99]
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        want_i, want_t = enc.clip_image(sd, image[pick]), enc.clip_text(sd, ids[pick])
    got_i, got_t = m.encode_image(image.cuda()), m.encode_text(ids.cuda())
    assert rel(got_i[pick], want_i) < 1e-4 and rel(got_t[pick], want_t) < 1e-4
    assert torch.equal(got_i[:2], m.encode_image(image[:2].cuda()))          # bitwise batch invariance


def _load_head(head, params):
    sd = head.state_dict()
    for k, v in params.items():
        assert k in sd, k
        sd[k] = v
    head.load_state_dict(sd)
    return head.cuda().eval()


def test_heads_match_reference_goldens(ops):
    from test_oracle_encode import dcmht_params, dsph_params
    from xmh import retrieval as xr
    from xmh.models import heads, weights as W
    g = np.load(os.path.join(GOLDEN, "encode_heads.npz"))
    seed = int(g["seed"])
    emb = torch.from_numpy(g["emb"]).cuda()
    for K in (16, 64):
        layer = heads.DCMHTHashLayer(512, K)
        _load_head(layer.img_hash, dcmht_params(W, seed, K, "img"))
        _load_head(layer.txt_hash, dcmht_params(W, seed, K, "txt"))
        for mod, fn in (("img", layer.encode_img), ("txt", layer.encode_txt)):
            out = fn(emb)
            assert (out.cpu() - torch.from_numpy(g["dcmht%d_%s" % (K, mod)])).abs().max() < 3e-6
            code = xr.pack_pair_argmax(out).unpack().cpu().numpy()
            assert (code != g["dcmht%d_%s_code" % (K, mod)]).mean() < 0.002
    layer = heads.DSPHHashLayer(512, 128)
    _load_head(layer.img_hash, dsph_params(W, seed, 128, "img"))
    _load_head(layer.txt_hash, dsph_params(W, seed, 128, "txt"))
    for mod, fn in (("img", layer.encode_img), ("txt", layer.encode_txt)):
        out = fn(emb)
        assert (out.cpu() - torch.from_numpy(g["dsph128_%s" % mod])).abs().max() < 3e-6
        code = xr.pack_sign(out).unpack().cpu().numpy()
        assert (code != g["dsph128_%s_code" % mod]).mean() < 0.002


def test_twdh_heads_match_reference_golden(ops):
    """the long (512-bit) DCMHT hash layer and the [2*long, 2*short] transforms of models/TwDH/TwDH.py:66-85."""
    from test_oracle_encode import dcmht_params
    from xmh import retrieval as xr
    from xmh.models import heads, weights as W
    g = np.load(os.path.join(GOLDEN, "encode_twdh.npz"))
    seed = int(g["seed"])
    emb = torch.from_numpy(g["emb"]).cuda()
    layer = heads.DCMHTHashLayer(512, 512)
    _load_head(layer.img_hash, dcmht_params(W, seed, 512, "img", family="twdh"))
    _load_head(layer.txt_hash, dcmht_params(W, seed, 512, "txt", family="twdh"))
    for mod, fn in (("img", layer.encode_img), ("txt", layer.encode_txt)):
        long_hash = fn(emb)
        assert (long_hash.cpu() - torch.from_numpy(g["long_%s" % mod])).abs().max() < 3e-6
        assert (xr.pack_pair_argmax(long_hash).unpack().cpu().numpy() != g["long_%s_code" % mod]).mean() < 0.002
        for S in (16, 64):
            w = torch.from_numpy(g["trans%d" % S]).cuda().t().contiguous()
            short = ops.pair_softmax(ops.gemm_nt(torch.from_numpy(g["long_%s" % mod]).cuda(), w))
            assert (short.cpu() - torch.from_numpy(g["short%d_%s" % (S, mod)])).abs().max() < 3e-6
            assert (xr.pack_pair_argmax(short).unpack().cpu().numpy() != g["short%d_%s_code" % (S, mod)]).mean() < 0.01


def test_fast_mode_fp16_error_and_bit_agreement(ops, clip_models):
    """fp16 activations -> ~1e-3 embedding error, well under 1 % code-bit flips."""
    g, W, m, _ = clip_models
    image = W.synth_images(3, 16).cuda()
    ref = m.encode_image(image)
    ops.set_precision("f16")
    try:
        fast = m.encode_image(image)
    finally:
        ops.set_precision("f32")
    assert rel(fast, ref) < 1e-2
    proj = torch.randn(512, 64, generator=g_(1)).cuda()
    flips = ((ref @ proj).sign() != (fast @ proj).sign()).float().mean().item()
    assert flips < 0.01


def test_mith_head_matches_reference_goldens(ops):
    from test_oracle_encode import mith_inputs, mith_params
    from xmh.models import weights as W
    from xmh.models.mith import MITHHashLayer
    g = np.load(os.path.join(GOLDEN, "encode_mith.npz"))
    seed = int(g["seed"])
    cls_i, tok_i, cls_t, tok_t, mask = mith_inputs(W, seed)
    for K in (16, 64):
        head = MITHHashLayer(512, K)
        hp = mith_params(W, seed, K)
        sd = head.state_dict()
        for k in sd:
            src = k.replace("gcl_t.", "gcl_i.")
            assert src in hp, k
            sd[k] = hp[src]
        head.load_state_dict(sd)
        head = head.cuda().eval()
        _, ch_i, th_i, _ = head.encode_img(cls_i.cuda(), tok_i.cuda())
        _, ch_t, th_t, _ = head.encode_txt(cls_t.cuda(), tok_t.cuda(), mask.cuda())
        for got, key in ((ch_i, "cls_hash_i"), (th_i, "tok_hash_i"), (ch_t, "cls_hash_t"), (th_t, "tok_hash_t")):
            assert (got.cpu() - torch.from_numpy(g["k%d_%s" % (K, key)])).abs().max() < 2e-5, (K, key)
        assert ((ch_i + th_i).sign().cpu().numpy() != g["k%d_code_i" % K]).mean() < 0.01
        assert ((ch_t + th_t).sign().cpu().numpy() != g["k%d_code_t" % K]).mean() < 0.01


def test_mith_head_entry_point_equals_the_primitive_chain(ops):
    """xmh_head_mith against the per-primitive chain, bit for bit, both modalities (text with a token mask), all precisions;
    an in-place weight update must reach the cached descriptor."""
    import xmh.models.clip as C
    from test_oracle_encode import mith_inputs
    from xmh.models import weights as W
    from xmh.models.mith import MITHHashLayer
    cls_i, tok_i, cls_t, tok_t, mask = (t.cuda() for t in mith_inputs(W, 3))
    torch.manual_seed(4)
    head = MITHHashLayer(512, 32).cuda().eval()

    def both():
        _, a, b, _ = head.encode_img(cls_i, tok_i)
        _, c, d, _ = head.encode_txt(cls_t, tok_t, mask)
        return [a, b, c, d]

    before = ops.get_precision()
    try:
        for prec in ("f32", "f32x", "f16"):
            ops.set_precision(prec)
            native = both()
            C.NATIVE_FORWARD = False
            try:
                chain = both()
            finally:
                C.NATIVE_FORWARD = True
            for x, y in zip(native, chain):
                assert x.shape == y.shape and torch.equal(x, y), prec
        ops.set_precision("f32")
        first = both()
        with torch.no_grad():
            head.lct_i.hashing.fc_list[3].weight.mul_(-1.0)
        second = both()
        assert not torch.equal(first[1], second[1]) and torch.equal(first[3], second[3])      # image tokens_hash changed, text untouched
    finally:
        ops.set_precision(before)


def test_lta_edge_cases_against_oracle(ops):
    """tokens with fewer than top-k positive concepts, fully masked concepts (NaN -> 0), exact ties."""
    from oracle import encode as enc
    B, L, K, D = 2, 9, 16, 64
    gen = torch.Generator().manual_seed(4)
    S = torch.tanh(torch.randn(B, L, K, generator=gen))
    S[0, :, 3] = -0.5                   # concept 3 never positive for sample 0 -> all-zero aggregate row
    S[1, 2, :] = 0.25                   # a token whose scores all tie
    S[1, 4, :] = -0.1                   # a token with no positive concept at all
    X = torch.randn(B, L, D, generator=gen)
    mask = torch.zeros(B, L, dtype=torch.bool)
    mask[0, 7:] = True
    got = ops.lta_aggregate(S.cuda(), X.cuda(), mask.cuda(), None, 8).cpu()
    sim = S.permute(1, 0, 2).clone() + torch.where(mask, float("-inf"), 0.0).t()[:, :, None]
    sim = torch.where(sim > 0, sim, torch.full_like(sim, float("-inf")))
    kth = torch.topk(sim, k=8, dim=-1).values.min(dim=-1, keepdim=True).values
    sim = torch.where(sim >= kth, sim, torch.full_like(sim, float("-inf")))
    att = torch.softmax(sim, dim=0)
    att = torch.where(torch.isnan(att), torch.zeros_like(att), att)
    want = torch.bmm(att.permute(1, 2, 0), X)
    assert (got - want).abs().max() < 1e-5 and got[0, 3].abs().max() == 0
    del enc


def test_fast_mode_gemm_paths_and_cast(ops):
    """fp16-input MFMA GEMM (K % 32 == 0) and its fp32-operand fallback (other K) agree with an fp16-rounded reference."""
    for M, N, K in ((300, 200, 768), (77, 130, 96), (64, 64, 40), (5, 512, 3072)):
        A, W = torch.randn(M, K, generator=g_(M)), (torch.randn(N, K, generator=g_(N)) * 0.05).half().float()
        b = torch.randn(N, generator=g_(K))
        want = A.half().double() @ W.double().t() + b.double()
        got = ops.gemm_nt(A.cuda(), W.cuda(), b.cuda(), precision=ops.PREC_F16)
        assert rel(got, want) < 5e-6, (M, N, K)
    x = torch.randn(33, 64, generator=g_(9))
    assert torch.equal(ops.cast_f16(x.cuda()).cpu(), x.half())
