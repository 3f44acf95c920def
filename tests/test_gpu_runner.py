"""End-to-end drop-in check on the GPU: a registered runner built from a config runs get_code + valid and writes
the reference's artefacts; its mAP equals the oracle's calc_map_k port on the codes it produced."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_cfg(tmp_path, arch, runner, K, layers=2):
    from xmh.utils.config import Config
    return Config({
        "model": {"arch": arch, "clip_path": "synthetic:1814:vision_layers=%d,transformer_layers=%d" % (layers, layers)},
        "dataset": {"arch": "synthetic", "name": "synth", "num_classes": 24, "retrieval_num": 230, "max_word": 32, "image_resolution": 224},
        "run": {"arch": runner, "output_dim": K, "device": 0, "batch_size": 32, "num_workers": 0, "is_train": False, "query_num": 50,
                "train_num": 60, "save_dir": str(tmp_path), "log_dir": str(tmp_path), "seed": 1814},
    })


@pytest.mark.parametrize("arch,runner,K", [("DCMHT", "DCMHTTrainer", 16), ("DSPH", "DSPHTrainer", 128), ("MITH", "MITHTrainer", 64)])
def test_runner_valid_matches_oracle(tmp_path, arch, runner, K):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import scipy.io as scio
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from oracle import retrieval as orc
    from xmh.common.register import registry
    cfg = make_cfg(tmp_path, arch, runner, K)
    trainer = registry.get_runner_class(runner).from_config(cfg=cfg, autorun=False)
    q_img, q_txt = trainer.get_code(trainer.query_loader, trainer.query_num)
    r_img, r_txt = trainer.get_code(trainer.retrieval_loader, trainer.retrieval_num)
    trainer.encode_fuse = 1                                  # one loader batch per forward, like the reference: same codes
    q_img1, q_txt1 = trainer.get_code(trainer.query_loader, trainer.query_num)
    # a different row count picks different GEMM tiles: the only admissible difference is a flipped near-zero logit
    assert (q_img1 != q_img).float().mean() < 2e-3 and (q_txt1 != q_txt).float().mean() < 2e-3
    trainer.encode_fuse = 4
    assert q_img.shape == (50, K) and r_txt.shape == (230, K) and q_img.dtype == torch.float32
    assert set(np.unique(q_img.cpu().numpy())) <= {-1.0, 0.0, 1.0}
    maps = trainer.valid(0, k=None)
    qL, rL = trainer.query_labels, trainer.retrieval_labels
    want = [orc.map_k(a.cpu(), b.cpu(), qL, rL, None, stable=True) for a, b in ((q_img, r_txt), (q_txt, r_img), (q_img, r_img), (q_txt, r_txt))]
    for got, w in zip(maps, want):
        assert abs(got - float(w)) < 1e-6
    # artefacts (runners/base.py:322-336, :379-405)
    mat = scio.loadmat(os.path.join(str(tmp_path), "mat_files", "last.mat"))
    assert set(["q_img", "q_txt", "r_img", "r_txt", "q_l", "r_l"]) <= set(mat)
    assert mat["q_img"].dtype == np.float32 and mat["q_img"].shape == (50, K) and np.array_equal(mat["r_txt"], r_txt.cpu().numpy())
    assert mat["q_l"].dtype == np.int64 and np.array_equal(mat["r_l"], rL.numpy())
    assert os.path.exists(os.path.join(str(tmp_path), "mat_files", "i2t-best.mat"))
    pth = os.path.join(str(tmp_path), "model-0.pth")
    assert os.path.exists(pth)
    sd = torch.load(pth, map_location="cpu")
    assert any(k.startswith("backbone.visual.transformer.resblocks.0.attn.in_proj_weight") for k in sd)
    assert any(k.startswith("hash.") for k in sd)
    # the .pth was serialised on a host thread under the encode loop: the same file torch.save(model.state_dict(), path) writes
    live = trainer.model.state_dict()
    assert list(sd) == list(live) and all(torch.equal(sd[k], live[k].cpu()) for k in live)
    assert next(iter(torch.load(pth).values())).device == next(iter(live.values())).device      # saved from the device, like the reference's
    # i2t-best / t2i-best / last of one epoch are names of one file; each reads back complete
    for other in ("i2t-best.mat", "t2i-best.mat"):
        m2 = scio.loadmat(os.path.join(str(tmp_path), "mat_files", other))
        assert all(np.array_equal(m2[k], mat[k]) for k in ("q_img", "q_txt", "r_img", "r_txt", "q_l", "r_l"))
    # mAP@k path and the calc_map_k injection seam
    m50 = trainer.valid(1, k=50)
    assert abs(m50[0] - float(orc.map_k(q_img.cpu(), r_txt.cpu(), qL, rL, 50, stable=True))) < 1e-6
    calls = []

    def spy(qB, rB, qLx, rLx, k=None):
        calls.append(qB.shape)
        return orc.map_k(qB.cpu(), rB.cpu(), qLx, rLx, k, stable=True)
    trainer.calc_map_k = spy
    m_spy = trainer.valid(2, k=None)
    assert len(calls) == 4 and abs(m_spy[0] - maps[0]) < 1e-6
    # a saved checkpoint round-trips through resume_model -> test()
    cfg.run.resume_model = pth
    t2 = registry.get_runner_class(runner).from_config(cfg=cfg, autorun=False)
    t2.top_k = None
    assert abs(t2.test()[0] - maps[0]) < 1e-6
    assert os.path.exists(os.path.join(str(tmp_path), "mat_files", "test.mat"))


def test_runner_takes_raw_uint8_photos(tmp_path):
    """a dataset that yields undecoded-photo-like RGB bytes ([H, W, 3] uint8) instead of transformed floats;
    the runner resizes/normalises on the GPU (Pillow-exact) and produces the codes the host-side transform would."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from oracle import preprocess as O
    from xmh.common.register import registry
    cfg = make_cfg(tmp_path, "DCMHT", "DCMHTTrainer", 16, layers=1)
    cfg.dataset.raw_image_hw = [96, 130]
    trainer = registry.get_runner_class("DCMHTTrainer").from_config(cfg=cfg, autorun=False)
    q_img, _ = trainer.get_code(trainer.query_loader, trainer.query_num)
    ds = trainer.query_loader.dataset
    assert ds[0][0].dtype == torch.uint8 and tuple(ds[0][0].shape) == (96, 130, 3)
    host = torch.from_numpy(np.stack([O.eval_transform(ds[i][0].numpy()) for i in range(8)])).cuda()
    want = trainer.make_hash_code(trainer.model.encode_image(host))
    assert torch.equal(q_img[:8].cpu(), want.float().cpu())


def test_twdh_runner_long_and_short_codes(tmp_path):
    """TwDHTrainer encodes once and evaluates the 512-bit long code and two short codes; every mAP equals the
    oracle's calc_map_k port on the codes it produced (the long one through the long-code scan kernels)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from oracle import retrieval as orc
    from xmh.common.register import registry
    cfg = make_cfg(tmp_path, "TwDH", "TwDHTrainer", 16, layers=1)
    cfg.model.long_dim = 512
    cfg.model.trans_matrix = "synthetic"
    cfg.model.short_dims = [16, 64]
    trainer = registry.get_runner_class("TwDHTrainer").from_config(cfg=cfg, autorun=False)
    assert sorted(trainer.model.get_short_dims()) == [16, 64]
    ql_img, ql_txt, qs_img, qs_txt = trainer.get_code(trainer.query_loader, trainer.query_num)
    rl_img, rl_txt, rs_img, rs_txt = trainer.get_code(trainer.retrieval_loader, trainer.retrieval_num)
    assert ql_img.shape == (50, 512) and rs_txt["64"].shape == (230, 64) and set(qs_img) == {"16", "64"}
    maps = trainer.valid(0, k=None)
    qL, rL = trainer.query_labels, trainer.retrieval_labels
    for name, (qi, qt, ri, rt) in (("long", (ql_img, ql_txt, rl_img, rl_txt)), ("16", (qs_img["16"], qs_txt["16"], rs_img["16"], rs_txt["16"])),
                                   ("64", (qs_img["64"], qs_txt["64"], rs_img["64"], rs_txt["64"]))):
        want = [orc.map_k(a.cpu(), b.cpu(), qL, rL, None, stable=True) for a, b in ((qi, rt), (qt, ri), (qi, ri), (qt, rt))]
        for got, w in zip(maps[name], want):
            assert abs(got - float(w)) < 1e-6, name
    files = set(os.listdir(os.path.join(str(tmp_path), "mat_files")))
    assert {"i2t-long.mat", "t2i-long.mat", "i2t-short-16.mat", "t2i-short-64.mat"} <= files


def test_runner_on_mat_files_and_photos_of_mixed_sizes(tmp_path):
    """SURVEY 8f-2 end to end: the reference's file layout (index/caption/label .mat + image files) through the
    ``transformer_dataset`` mirror -- PIL only decodes, photos of different sizes reach the runner as a list and are resized /
    normalised on the GPU.  The BPE merge table does not travel to the GPU box, so a toy tokenizer is registered instead."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import scipy.io as scio
    from PIL import Image
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    import xmh.dataset  # noqa: F401
    from oracle import preprocess as O
    from xmh.common.register import registry

    if registry.get_tokenizer_class("toy_words") is None:
        @registry.register_tokenizer("toy_words")
        class ToyWords:                                           # same two methods the dataset calls
            def tokenize(self, text):
                return str(text).strip().lower().split()

            def convert_tokens_to_ids(self, tokens):
                return [49406 if t == "<|startoftext|>" else 49407 if t == "<|endoftext|>" else 1 + sum(map(ord, t)) % 4000 for t in tokens]

    rng = np.random.default_rng(11)
    root = tmp_path / "data" / "tiny"
    root.mkdir(parents=True)
    n, C = 23, 6
    paths = []
    for i in range(n):
        h, w = [(40, 64), (64, 48), (50, 50)][i % 3]
        p = root / ("img%02d.png" % i)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), mode="RGB").save(p)
        paths.append(str(p))
    labels = (rng.random((n, C)) < 0.4).astype(np.int64)
    labels[:, 0] = 1
    scio.savemat(root / "index.mat", {"index": np.asarray(paths)})
    scio.savemat(root / "caption.mat", {"caption": np.asarray([["photo number %d of a dog" % i] for i in range(n)])})
    scio.savemat(root / "label.mat", {"category": labels})
    cfg = make_cfg(tmp_path, "DCMHT", "DCMHTTrainer", 16, layers=1)
    cfg.dataset.update({"arch": "transformer_dataset", "name": "tiny", "path": str(tmp_path / "data"), "label_file": "label.mat",
                        "tokenizer_arch": "toy_words"})
    cfg.run.query_num, cfg.run.batch_size = 7, 5
    np.random.seed(3)
    trainer = registry.get_runner_class("DCMHTTrainer").from_config(cfg=cfg, autorun=False)
    assert trainer.query_num == 7 and trainer.retrieval_num == n - 7 and trainer.train_loader is None
    r_img, r_txt = trainer.get_code(trainer.retrieval_loader, trainer.retrieval_num)
    ds = trainer.retrieval_loader.dataset
    host = torch.from_numpy(np.stack([O.eval_transform(ds[i][0].numpy()) for i in range(len(ds))])).cuda()
    want = trainer.make_hash_code(trainer.model.encode_image(host)).float().cpu()
    assert r_img.shape == (n - 7, 16) and (r_img.cpu() != want).float().mean() < 0.01       # batch-size dependent GEMM tiling only
    ids = torch.stack([ds[i][1] for i in range(len(ds))]).cuda()
    want_t = trainer.make_hash_code(trainer.model.encode_text(ids)).float().cpu()
    assert (r_txt.cpu() != want_t).float().mean() < 0.01
    maps = trainer.valid(0, k=None)
    assert all(0.0 <= m <= 1.0 for m in maps)


def test_bench_sharded_path_over_rccl_single_rank():
    """bench.py's multi-GPU exchange path (all_gather of histograms, all_reduce of AP sums over RCCL) on one rank:
    must run and give the same mAP as the single-process path."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--Q", "300", "--R", "20000", "--no-cpu-baseline",
            "--no-hbm-regime", "--no-encode"]
    a = json.loads(subprocess.run(base, capture_output=True, text=True, check=True, timeout=600).stdout.strip().splitlines()[-1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    b = json.loads(subprocess.run(base + ["--force-sharded"], capture_output=True, text=True, check=True, timeout=600, env=env).stdout.strip().splitlines()[-1])
    assert abs(a["mAP"] - b["mAP"]) < 1e-9 and b["n_gpus"] == 1 and b["value"] > 0
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in a
    assert a["roofline"]["bound"] in ("valu", "hbm", "mfma") and "frac" in a["roofline"] and "workload" in a["config"]
    assert "hbm_frac" in a["roofline"] and a["rccl_ranks"] == 1 and b["rccl_ranks"] == 1
    # the line on stdout is the compact contract line (bench_emit); the full object of the last run sits beside it
    detail = json.load(open(os.path.join(root, "gpurun_out", "bench_detail.json")))
    assert detail["roofline"]["hbm"]["bound"] == "hbm" and len(json.dumps(a)) < 4096


def test_bench_two_ranks_sharing_the_gpu():
    """`bench.py --gpus 2` END TO END on the one-GPU box ("has never executed with N > 1 on any machine"): the self
    launcher, two ranks, ShardedQueries.gather, the all-to-all exchange of map_k_sharded on workspace views, the fixed-gallery legs
    (configs[2], configs[4] top-k with its all-gather into one tensor and pinned D2H) -- with --share-gpu, i.e. both ranks on cuda:0 and
    gloo moving the device tensors, because RCCL refuses two ranks on one device.  The mAP of the weak-scaling step must be the mAP of
    one scan over both shards."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    Q, Rn = 300, 20000
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "2", "--warmup", "1", "--settle", "0", "--Q", str(Q),
           "--R", str(Rn), "--no-cpu-baseline", "--no-hbm-regime", "--no-encode"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "XMH_BENCH_CHILD")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert any("all_to_all" in c for c in line["config"]["collectives_in_step"])
    detail = json.load(open(os.path.join(root, "gpurun_out", "bench_detail.json")))
    assert detail["rccl_ranks"] == 2 and "share-gpu" in detail["launcher"]
    assert "error" not in detail["strong_scaling"] and "configs4_topk_10M_256bit" in detail["strong_scaling"], detail["strong_scaling"]
    # the same synthetic shards in one process, one scan
    spec = importlib.util.spec_from_file_location("xmh_bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from xmh import retrieval as R
    qB, qL, _, _ = bench.synth(Q, 8, 64, 80, seed=1814, p=0.04)
    shards = [bench.synth(8, Rn, 64, 80, seed=1814 + 1 + rank, p=0.04)[2:] for rank in range(2)]
    rB, rL = torch.cat([s_[0] for s_ in shards]), torch.cat([s_[1] for s_ in shards])
    whole = R.RankingScan(R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda()), 80)
    whole.histograms(False)
    want = float(whole.map_all(None)[0].item())
    assert abs(detail["mAP"] - want) < 1e-9, (detail["mAP"], want)


def test_runner_distributed_code_path_world_size_1(tmp_path):
    """the sharded eval path of the runner (contiguous shard sampler, RCCL all-gather of packed query codes, histogram
    exchange, all-reduce) with a 1-rank process group must reproduce the single-process result."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import torch.distributed as dist
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from xmh.common.register import registry
    cfg = make_cfg(tmp_path, "DSPH", "DSPHTrainer", 64, layers=1)
    single = registry.get_runner_class("DSPHTrainer").from_config(cfg=cfg, autorun=False)
    want = single.valid(0, k=None)
    cfg.run.distributed_addr, cfg.run.distributed_port = "127.0.0.1", 29577
    cfg.run.save_dir = cfg.run.log_dir = str(tmp_path / "dist")
    try:
        shard = registry.get_runner_class("DSPHTrainer").from_config(0, 1, True, cfg, None, autorun=False)
        got = shard.valid(0, k=None)
        q_img, _ = shard.get_code(shard.query_loader, shard.query_num)
        assert q_img.shape == (50, 64)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    for a, b in zip(got, want):
        assert abs(a - b) < 1e-9


def _runner_two_rank_worker(rank, world, port, tmp, want):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "clip-based-cross-modal-hash_amd")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from xmh.common.register import registry
    import pathlib
    cfg = make_cfg(pathlib.Path(tmp), "DSPH", "DSPHTrainer", 64, layers=1)
    cfg.run.distributed_addr, cfg.run.distributed_port, cfg.run.share_gpu = "127.0.0.1", port, True
    cfg.run.save_dir = cfg.run.log_dir = os.path.join(tmp, "dist")
    try:
        shard = registry.get_runner_class("DSPHTrainer").from_config(rank, world, True, cfg, None, autorun=False)
        assert dist.get_world_size() == world and shard.rank == rank
        got = shard.valid(0, k=None)
        lo, hi = shard._shard(shard.retrieval_num)
        assert hi - lo in (shard.retrieval_num // world, shard.retrieval_num - shard.retrieval_num // world) and (lo == 0) == (rank == 0)
        for a, b in zip(got, want):
            assert abs(a - b) < 1e-7, (rank, got, want)      # fp32 credits added per shard first: another order than one scan's
        open(os.path.join(tmp, "ok%d" % rank), "w").write("%r %r" % (lo, hi))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_runner_distributed_two_ranks_sharing_the_gpu(tmp_path):
    """the runner's sharded evaluation (what replaces runners/base.py:82-96,259-264 / main.py:38-51) with world_size = 2 on this one-GPU
    box: two processes on cuda:0, run.share_gpu puts the group on gloo.  Each rank encodes ITS contiguous half of the queries and of the
    gallery, the packed query codes are all-gathered, the totals tables exchanged, the mAP all-reduced: all four mAPs of valid() on every
    rank equal the single-process ones."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    import torch.multiprocessing as mp
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from xmh.common.register import registry
    cfg = make_cfg(tmp_path, "DSPH", "DSPHTrainer", 64, layers=1)
    single = registry.get_runner_class("DSPHTrainer").from_config(cfg=cfg, autorun=False)
    want = [float(m) for m in single.valid(0, k=None)]
    del single
    torch.cuda.empty_cache()
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    mp.spawn(_runner_two_rank_worker, args=(2, port, str(tmp_path), want), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d" % r)) for r in range(2))


def _mask_numbers(line):
    import re
    return re.sub(r"(nan|[-+]?\d+\.\d+(e[-+]?\d+)?)", "#", line)


def _golden_cfg(tmp_path, arch, runner, K, extra_model=None):
    from oracle import runner_fixture as RF
    from xmh.utils.config import Config
    model = {"arch": arch, "clip_path": "synthetic:%d:vision_layers=%d,transformer_layers=%d" % (RF.SEED, RF.CLIP_LAYERS, RF.CLIP_LAYERS)}
    model.update(extra_model or {})
    return Config({
        "model": model,
        "dataset": {"arch": "synthetic", "name": "synth", "num_classes": RF.NUM_CLASSES, "retrieval_num": RF.RETRIEVAL_NUM, "max_word": 32,
                    "image_resolution": 224, "seed": RF.SEED, "p_label": 0.1},
        "run": {"arch": runner, "output_dim": K, "device": 0, "batch_size": RF.BATCH, "num_workers": 0, "is_train": False, "query_num": RF.QUERY_NUM,
                "train_num": RF.RETRIEVAL_NUM, "save_dir": str(tmp_path), "log_dir": str(tmp_path), "seed": RF.SEED, "epochs": 1},
    })


def _load_golden_heads(trainer, tag, golden_keys):
    """same rule, same key names as oracle/make_golden_runner.py applied to the reference model"""
    from oracle import runner_fixture as RF
    from xmh.models import weights as W
    sd = trainer.model.state_dict()
    mine = sorted(k for k in sd if not k.endswith("num_batches_tracked"))
    assert mine == sorted(golden_keys), (set(mine) ^ set(golden_keys))       # reference checkpoints stay loadable
    sd.update({k: v.to(sd[k].device) for k, v in RF.head_state(W, tag, sd).items()})
    trainer.model.load_state_dict(sd)


def _cmp_codes(got, want, what):
    got = got.cpu().numpy()
    assert got.shape == want.shape and got.dtype == want.dtype, what
    flips = float((got != want).mean())
    assert flips <= 0.004, (what, flips)          # a near-zero logit may round the other way (fp32 summation order); nothing else
    return flips


def _check_maps_against_reference_log(mine, ref_logged, trainer, code_pairs, exact):
    """the four mAPs of one valid() against the reference's logged values.  With equal codes the ONLY admissible difference is
    the order of equal distances: the reference's default torch.sort leaves it unspecified (calc_utils.py:77) and
    on 24 items with 16..64-bit codes ties are everywhere.  So: this package's value == the oracle's canonical-order value on
    the reference's codes (1e-6), and BOTH it and the reference's logged value lie inside the oracle's tie-order envelope."""
    from oracle import retrieval as orc
    qL, rL = trainer.query_labels, trainer.retrieval_labels
    for got, ref, (qc, rc) in zip(mine, ref_logged, code_pairs):
        qc, rc = torch.from_numpy(qc), torch.from_numpy(rc)
        lo, hi = orc.map_k_tie_bounds(qc, rc, qL, rL, None)
        assert lo - 1e-6 <= ref <= hi + 1e-6, (ref, lo, hi)
        if exact:
            assert abs(got - float(orc.map_k(qc, rc, qL, rL, None, stable=True))) < 1e-6
            assert lo - 1e-6 <= got <= hi + 1e-6


@pytest.mark.parametrize("arch,runner", [("DCMHT", "DCMHTTrainer"), ("MITH", "MITHTrainer"), ("DSPH", "DSPHTrainer")])
def test_runner_reproduces_reference_runner_golden(tmp_path, arch, runner):
    """SURVEY 8c / VERDICT r1 a-7: get_code buffers, valid() mAPs, log line and .mat arrays of the REFERENCE's own
    DCMHTTrainer / MITHTrainer / DSPHTrainer (run by oracle/make_golden_runner.py on the 8-query / 24-gallery synthetic set) against
    this package's runners on the same weights and data.  DSPH (configs[3]'s method, 128 bit): the reference
    class is constructed with its HyP threshold read from the workbook it ships (oracle/_ref_import.py stands in for xlrd)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import logging
    import scipy.io as scio
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from conftest import GOLDEN
    from oracle import runner_fixture as RF
    from xmh.common.register import registry
    g = np.load(os.path.join(GOLDEN, "runner.npz"))
    K = RF.CASES[arch]
    trainer = registry.get_runner_class(runner).from_config(cfg=_golden_cfg(tmp_path, arch, runner, K), autorun=False)
    _load_golden_heads(trainer, "%s%d" % (arch, K), [str(k) for k in g[arch + "_state_keys"]])
    trainer.encode_fuse = 1                                   # the reference's granularity: one loader batch per forward
    assert np.array_equal(trainer.query_labels.numpy(), g[arch + "_mat_q_l"]) and np.array_equal(trainer.retrieval_labels.numpy(), g[arch + "_mat_r_l"])
    q_img, q_txt = trainer.get_code(trainer.query_loader, trainer.query_num)
    r_img, r_txt = trainer.get_code(trainer.retrieval_loader, trainer.retrieval_num)
    flips = [_cmp_codes(a, g["%s_%s" % (arch, n)], n) for a, n in ((q_img, "q_img"), (q_txt, "q_txt"), (r_img, "r_img"), (r_txt, "r_txt"))]
    lines = []

    class Cap(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    trainer.logger.addHandler(Cap())
    maps = trainer.valid(0, k=None)                           # returns (i2t, t2i, i2i, t2t); the log line orders them i2t, t2i, t2t, i2i
    want = g[arch + "_maps_i2t_t2i_t2t_i2i"]
    _check_maps_against_reference_log((maps[0], maps[1], maps[3], maps[2]), want, trainer,
                                      [(g[arch + "_q_img"], g[arch + "_r_txt"]), (g[arch + "_q_txt"], g[arch + "_r_img"]),
                                       (g[arch + "_q_txt"], g[arch + "_r_txt"]), (g[arch + "_q_img"], g[arch + "_r_img"])], exact=max(flips) == 0)
    line = [ln for ln in lines if ln.startswith(">>>>>> [0/1]")][-1]
    assert _mask_numbers(line) == _mask_numbers(str(g[arch + "_log_line"]))
    mat = scio.loadmat(os.path.join(str(tmp_path), "mat_files", "last.mat"))
    assert str(mat["q_img"].dtype) == str(g[arch + "_mat_q_img_dtype"]) and str(mat["q_l"].dtype) == str(g[arch + "_mat_q_l_dtype"])
    assert np.array_equal(mat["q_l"], g[arch + "_mat_q_l"]) and np.array_equal(mat["r_l"], g[arch + "_mat_r_l"])
    assert np.array_equal(mat["r_img"], r_img.cpu().numpy())
    have = sorted(os.listdir(os.path.join(str(tmp_path), "mat_files"))) + sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".pth"))
    assert have == [str(f) for f in g[arch + "_files"]]


def test_twdh_reproduces_reference_class_on_shipped_matrices(tmp_path):
    """the reference's TwDH class (models/TwDH/TwDH.py:34-85) was instantiated with the transform matrices it
    ships (data/transformer/TwDH/coco/trans/512/{16,32,64}.pkl, carried in the golden as inputs) and driven by its own
    TwDHTrainer.get_code / valid (runners/TwDH/runner.py:145-228); this package's TwDH + TwDHTrainer must give the same long
    and short code buffers, mAPs, log lines and files."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import logging
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from conftest import GOLDEN
    from oracle import runner_fixture as RF
    from xmh.common.register import registry
    g = np.load(os.path.join(GOLDEN, "runner.npz"))
    LONG = RF.CASES["TwDH"]
    shorts = [int(x) for x in g["TwDH_short_dims"]]
    tdir = os.path.join(str(tmp_path), "trans", str(LONG))
    os.makedirs(tdir)
    for s in shorts:
        torch.save(torch.from_numpy(g["TwDH_trans_%d" % s]), os.path.join(tdir, "%d.pkl" % s))
    cfg = _golden_cfg(tmp_path, "TwDH", "TwDHTrainer", 16, {"long_dim": LONG, "trans_matrix": os.path.join(str(tmp_path), "trans")})
    trainer = registry.get_runner_class("TwDHTrainer").from_config(cfg=cfg, autorun=False)
    assert sorted(trainer.model.get_short_dims()) == shorts
    _load_golden_heads(trainer, "TwDH%d" % LONG, [str(k) for k in g["TwDH_state_keys"]])
    trainer.encode_fuse = 1
    ql_i, ql_t, qs_i, qs_t = trainer.get_code(trainer.query_loader, trainer.query_num)
    rl_i, rl_t, rs_i, rs_t = trainer.get_code(trainer.retrieval_loader, trainer.retrieval_num)
    flips = {"long": max(_cmp_codes(a, g["TwDH_long_" + n], "long " + n) for a, n in ((ql_i, "q_img"), (ql_t, "q_txt"), (rl_i, "r_img"), (rl_t, "r_txt")))}
    for s in shorts:
        k = str(s)
        flips[k] = max(_cmp_codes(a, g["TwDH_%s_%s" % (k, n)], k + " " + n)
                       for a, n in ((qs_i[k], "q_img"), (qs_t[k], "q_txt"), (rs_i[k], "r_img"), (rs_t[k], "r_txt")))
    lines = []

    class Cap(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    trainer.logger.addHandler(Cap())
    res = trainer.valid(0, k=None)
    for name, maps in res.items():
        want = g["TwDH_%s_maps_i2t_t2i_t2t_i2i" % name]
        gq_i, gq_t, gr_i, gr_t = (g["TwDH_%s_%s" % (name, n)] for n in ("q_img", "q_txt", "r_img", "r_txt"))
        _check_maps_against_reference_log((maps[0], maps[1], maps[3], maps[2]), want, trainer,
                                          [(gq_i, gr_t), (gq_t, gr_i), (gq_t, gr_t), (gq_i, gr_i)], exact=flips[name] == 0)
        tag = "Long" if name == "long" else "Short, %s Bit" % name
        line = [ln for ln in lines if ln.startswith(">>>>>> [0/1], " + tag)][-1]
        assert _mask_numbers(line) == _mask_numbers(str(g["TwDH_%s_log_line" % name]))
    assert sorted(os.listdir(os.path.join(str(tmp_path), "mat_files"))) == [str(f) for f in g["TwDH_files"]]


def test_mith_runner_topk_mode_with_a_forced_exact_zero(tmp_path):
    """VERDICT r5 missing 1: MITH quantises sign(cls_hash + tokens_hash) (reference runners/MITH/runner.py:125-131, base.py:407-410): an
    activation of exactly 0 stays 0.  Force one in a query row and one in a gallery row: retrieve_topk() must return, for every query, the
    stable sort of the reference's calc_hammingDist on the -1/0/+1 codes the runner itself produced (half-integer distances included)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from oracle import retrieval as orc
    from xmh.common.register import registry
    cfg = make_cfg(tmp_path, "MITH", "MITHTrainer", 64, layers=1)
    trainer = registry.get_runner_class("MITHTrainer").from_config(cfg=cfg, autorun=False)
    inner = trainer.generate_hash

    def with_zeros(image, text, key_padding_mask=None):
        ih, th = inner(image, text, key_padding_mask)
        ih, th = ih.clone(), th.clone()
        ih[0, 3] = 0.0                                       # first row of every forward: query 0 / gallery rows 0, 128, ...
        th[0, 7] = 0.0
        th[1, 7] = 0.0
        return ih, th
    trainer.generate_hash = with_zeros
    q_img, q_txt = trainer.get_code(trainer.query_loader, trainer.query_num)
    r_img, r_txt = trainer.get_code(trainer.retrieval_loader, trainer.retrieval_num)
    assert (q_img == 0).any() and (r_txt == 0).any() and set(np.unique(r_img.cpu().numpy())) <= {-1.0, 0.0, 1.0}
    got = trainer.retrieve_topk(20, tasks=("i2t", "t2i", "i2i", "t2t"))
    halves = False
    for task, (q, r) in {"i2t": (q_img, r_txt), "t2i": (q_txt, r_img), "i2i": (q_img, r_img), "t2t": (q_txt, r_txt)}.items():
        val, order = torch.sort(orc.hamming_dist(q.cpu(), r.cpu()), dim=1, stable=True)       # restates calc_hammingDist, :51-56
        d, i = got[task]
        assert torch.equal(i.long(), order[:, :20]), task
        assert torch.equal(d, val[:, :20]), task
        halves |= bool((d * 2 % 2 == 1).any())
    assert halves                                            # a half-integer distance made it into a list: the ternary kernels ran
    # and the mAP path on the same ternary code sets (the scan's zero planes), against the oracle's stable-order calc_map_k port
    maps = trainer.valid(0, k=None)
    want = orc.map_k(q_img.cpu(), r_txt.cpu(), trainer.query_labels, trainer.retrieval_labels, None, stable=True)
    assert abs(maps[0] - float(want)) < 1e-6
    # k larger than the gallery: unused slots
    d, i = trainer.retrieve_topk(300, tasks=("i2t",))["i2t"]
    assert (i[:, 230:] == -1).all() and torch.isinf(d[:, 230:]).all() and (i[:, :230] >= 0).all()


def test_bench_eight_ranks_sharing_the_gpu():
    """the driver's N = 8 command line (`bench.py --gpus 8`) END TO END without an 8-GPU node: eight ranks on cuda:0
    over gloo (--share-gpu).  The weak-scaling headline's mAP must equal ONE scan over all eight shards, configs[2]'s strong-scaling
    mAP one scan over the whole NUS-WIDE-shaped gallery, and configs[4]'s merged top-100 lists (Q = 1, 8, 64) the lists of ONE top-k
    over all 10 M rows (digest of the (distance, index) arrays)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib.util
    import json
    import subprocess
    import sys
    import zlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world, Q, Rn = 8, 296, 9000
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--share-gpu", "--steps", "2", "--warmup", "1", "--settle", "0", "--Q", str(Q),
           "--R", str(Rn), "--no-cpu-baseline", "--no-hbm-regime", "--no-encode"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "XMH_BENCH_CHILD")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["value"] > 0
    detail = json.load(open(os.path.join(root, "gpurun_out", "bench_detail.json")))
    assert detail["rccl_ranks"] == world and "share-gpu" in detail["launcher"]
    strong = detail["strong_scaling"]
    assert "error" not in strong, strong
    spec = importlib.util.spec_from_file_location("xmh_bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from xmh import retrieval as R
    from xmh import sharded

    def one_scan(qB, qL, rB, rL, C):
        whole = R.RankingScan(R.pack_sign(qB.cuda()), R.pack_labels(qL.cuda()), R.pack_sign(rB.cuda()), R.pack_labels(rL.cuda()), C)
        whole.histograms(False)
        return float(whole.map_all(None)[0].item())
    # weak-scaling headline: the same synthetic shards in one process, one scan
    qB, qL, _, _ = bench.synth(Q, 8, 64, 80, seed=1814, p=0.04)
    shards = [bench.synth(8, Rn, 64, 80, seed=1814 + 1 + rank, p=0.04)[2:] for rank in range(world)]
    want = one_scan(qB, qL, torch.cat([s_[0] for s_ in shards]), torch.cat([s_[1] for s_ in shards]), 80)
    assert abs(detail["mAP"] - want) < 1e-9, (detail["mAP"], want)
    # configs[2], strong scaling: Q 5000 x R 188 000 in total
    b = sharded.shard_bounds(188000, world)
    qB, qL, _, _ = bench.synth(5000, 8, 64, 21, seed=2814, p=0.1)
    shards = [bench.synth(8, b[rank + 1] - b[rank], 64, 21, seed=2815 + rank, p=0.1)[2:] for rank in range(world)]
    want = one_scan(qB, qL, torch.cat([s_[0] for s_ in shards]), torch.cat([s_[1] for s_ in shards]), 21)
    assert abs(strong["configs2_nuswide_map"]["mAP"] - want) < 1e-9, (strong["configs2_nuswide_map"]["mAP"], want)
    # configs[4], strong scaling: top-100 over 10 M x 256 bit in total, merged lists vs ONE top-k over all rows
    b = sharded.shard_bounds(10_000_000, world)
    g = torch.Generator(device="cuda")
    parts = []
    for rank in range(world):
        g.manual_seed(4814 + rank)
        parts.append(torch.randint(-2**31, 2**31 - 1, (b[rank + 1] - b[rank], 8), dtype=torch.int32, device="cuda", generator=g))
    g.manual_seed(4813)
    qall = torch.randint(-2**31, 2**31 - 1, (64, 8), dtype=torch.int32, device="cuda", generator=g)
    gallery = R.PackedCodes(torch.cat(parts), None, 256)
    for Qn in (1, 8, 64):
        d, i = R.hamming_topk(R.PackedCodes(qall[:Qn].contiguous(), None, 256), gallery, 100)
        d, i = (d.cpu().to(torch.int32) & 0xFFFF), i.cpu()
        leg = strong["configs4_topk_10M_256bit"]["legs"]["Q%d" % Qn]
        assert leg["lists_crc32"] == zlib.crc32(d.numpy().tobytes() + i.numpy().tobytes()), Qn
        assert leg["first_hit"] == [int(d[0, 0]), int(i[0, 0])]


def _runner_topk_two_rank_worker(rank, world, port, tmp):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (root, os.path.join(root, "clip-based-cross-modal-hash_amd")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import pathlib
    import torch.distributed as dist
    import xmh.models  # noqa: F401
    import xmh.runners  # noqa: F401
    from xmh.common.register import registry
    cfg = make_cfg(pathlib.Path(tmp), "MITH", "MITHTrainer", 64, layers=1)
    cfg.run.distributed_addr, cfg.run.distributed_port, cfg.run.share_gpu = "127.0.0.1", port, True
    cfg.run.save_dir = cfg.run.log_dir = os.path.join(tmp, "dist")
    try:
        shard = registry.get_runner_class("MITHTrainer").from_config(rank, world, True, cfg, None, autorun=False)
        inner = shard.generate_hash

        def with_zero(image, text, key_padding_mask=None):          # rank 1 only: its shard alone holds exact zeros
            ih, th = inner(image, text, key_padding_mask)
            if rank == 1:
                ih, th = ih.clone(), th.clone()
                ih[0, 5] = 0.0
                th[0, 9] = 0.0
            return ih, th
        shard.generate_hash = with_zero
        got = shard.retrieve_topk(15)
        torch.save({k: (d, i) for k, (d, i) in got.items()}, os.path.join(tmp, "lists%d.pt" % rank))
        q_img, q_txt = shard.get_code(shard.query_loader, shard.query_num)
        r_img, r_txt = shard.get_code(shard.retrieval_loader, shard.retrieval_num)
        torch.save((q_img.cpu(), q_txt.cpu(), r_img.cpu(), r_txt.cpu()), os.path.join(tmp, "codes%d.pt" % rank))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_runner_topk_two_ranks_sharing_the_gpu_with_zeros_on_one_rank(tmp_path):
    """retrieve_topk() sharded (north_star: rank r holds its shard's codes, query codes all-gathered, partial lists merged on the host)
    with world = 2 on the one GPU, and exact zeros in rank 1's rows only: both ranks must rank in half units (the reduced value flags)
    and return the same lists = the stable sort of the reference's distance over the whole code sets."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    import torch.multiprocessing as mp
    from oracle import retrieval as orc
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    mp.spawn(_runner_topk_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    lists = [torch.load(os.path.join(str(tmp_path), "lists%d.pt" % r)) for r in range(2)]
    q_img, q_txt, r_img, r_txt = torch.load(os.path.join(str(tmp_path), "codes0.pt"))
    assert (r_img == 0).any() or (q_img == 0).any()
    for task, (q, r) in {"i2t": (q_img, r_txt), "t2i": (q_txt, r_img)}.items():
        val, order = torch.sort(orc.hamming_dist(q, r), dim=1, stable=True)
        for rank in range(2):
            d, i = lists[rank][task]
            assert torch.equal(i.long(), order[:, :15]), (task, rank)
            assert torch.equal(d, val[:, :15]), (task, rank)


def test_roctx_ranges_mark_the_phases_of_valid(tmp_path):
    """VERDICT r5 missing 5 / SURVEY 5 "Build adds": with xmh_prof_enable(2) every phase of valid() pushes a roctx range (tower forward,
    head, pack, pass 1, pass 2, top-k phases).  The roctx library is interposed by a recorder: LD_PRELOAD of a tiny shared object that
    logs roctxRangePushA / roctxRangePop, built here with gcc."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "fake_roctx.c"
    src.write_text(textwrap.dedent('''
        #include <stdio.h>
        #include <stdlib.h>
        static FILE* f;
        static int depth;
        static void open_log(void) { if (!f) f = fopen(getenv("XMH_ROCTX_LOG"), "a"); }
        int roctxRangePushA(const char* m) { open_log(); fprintf(f, "%d push %s\\n", depth, m); fflush(f); return depth++; }
        int roctxRangePop(void) { open_log(); --depth; fprintf(f, "%d pop\\n", depth); fflush(f); return depth; }
    '''))
    so = tmp_path / "librocprofiler-sdk-roctx.so.1"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    log = tmp_path / "roctx.log"
    code = textwrap.dedent('''
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
        import pathlib, torch
        from test_gpu_runner import make_cfg
        import xmh.models, xmh.runners
        from xmh import _lib
        from xmh.common.register import registry
        t = registry.get_runner_class("MITHTrainer").from_config(cfg=make_cfg(pathlib.Path(%r), "MITH", "MITHTrainer", 64, layers=1), autorun=False)
        t.valid(0, k=None)                                  # warm: nothing logged while ranges are off
        _lib.prof_enable(2)
        t.valid(1, k=None)
        t.retrieve_topk(5, tasks=("i2t",))
        _lib.prof_enable(0)
        t.valid(2, k=None)
    ''') % (root, os.path.join(root, "clip-based-cross-modal-hash_amd"), os.path.join(root, "tests"), str(tmp_path / "run"))
    env = dict(os.environ, XMH_ROCTX_LOG=str(log), LD_LIBRARY_PATH=str(tmp_path) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = log.read_text().splitlines()
    pushes = [ln.split(" push ", 1)[1] for ln in lines if " push " in ln]
    assert len(pushes) == sum(" pop" in ln for ln in lines) and lines[-1] == "0 pop"       # balanced, closed
    for must in ("encode: towers + heads", "xmh_vit_b32_forward (image tower)", "xmh_text_forward", "xmh_head_mith", "encode: quantise + pack",
                 "xmh_pack_sign", "xmh_hamming_hist (pass 1)", "xmh_hamming_map (pass 2 + mean)", "topk: sample + threshold pick",
                 "topk: streaming filter", "topk: select"):
        assert any(p.startswith(must) for p in pushes), (must, sorted(set(pushes)))
    # nesting: the tower entry points sit inside the runner's encode range
    depth_of = {ln.split(" push ", 1)[1]: int(ln.split(" ", 1)[0]) for ln in lines if " push " in ln}
    assert depth_of["xmh_head_mith"] >= 1
