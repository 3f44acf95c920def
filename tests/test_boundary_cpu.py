"""Host-side boundary logic that needs no GPU: registry semantics, config objects, synthetic dataset contract,
shard arithmetic."""
import os

import pytest
import torch

from xmh.common.register import registry
from xmh.utils.config import Config, load_yaml


def test_registry_contract():
    import xmh.models  # noqa: F401  (registers DCMHT / DSPH / MITH)
    import xmh.runners  # noqa: F401
    from xmh.models.base import BaseModel
    from xmh.runners.base import BaseTrainer
    for name in ("DCMHT", "DSPH", "MITH"):
        assert issubclass(registry.get_model_class(name), BaseModel)
    for name in ("DCMHTTrainer", "DSPHTrainer", "MITHTrainer"):
        assert issubclass(registry.get_runner_class(name), BaseTrainer)
    assert registry.get_model_class("nope") is None and registry.get_runner_class("nope") is None
    with pytest.raises(KeyError):
        registry.register_model("DCMHT")(registry.get_model_class("DCMHT"))
    with pytest.raises(AssertionError):
        registry.register_model("NotAModel")(dict)
    with pytest.raises(AssertionError):
        registry.register_runner("NotARunner")(dict)

    @registry.register_tokenizer("unit_test_tokenizer")
    class Tok:
        pass
    assert isinstance(registry.get_tokenizer_class("unit_test_tokenizer")(), Tok)
    assert "DCMHT" in registry.list_models() and "DCMHTTrainer" in registry.list_runners()


def test_config_object(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text("model:\n  arch: DCMHT\n  clip_path: synthetic\nrun:\n  output_dim: 64\n  save_dir: ./x\ndataset:\n  name: coco\n")
    cfg = load_yaml(str(p), save_dir=str(tmp_path))
    assert cfg.model.arch == "DCMHT" and cfg.run.get("top_k", None) is None and cfg.run.output_dim == 64
    assert cfg.run.save_dir == str(tmp_path) and cfg.run.log_dir == str(tmp_path)
    assert isinstance(cfg.dataset, Config) and cfg["dataset"]["name"] == "coco"


def test_synthetic_dataset_tuple_contract():
    from xmh.dataset import SyntheticPairs
    d = SyntheticPairs(7, num_classes=24, resolution=32, max_words=32, seed=3)
    image, ids, mask, label, index = d[4]
    assert image.shape == (3, 32, 32) and image.dtype == torch.float32
    assert ids.shape == (32,) and ids.dtype == torch.int64 and ids[0] == 49406 and ids.max() == 49407
    assert mask.dtype == torch.bool and torch.equal(mask, ids == 0)
    assert label.shape == (24,) and label.dtype == torch.int64 and label.sum() >= 1 and index == 4
    assert d.get_all_label().shape == (7, 24)
    again = d[4]
    assert torch.equal(again[0], image) and torch.equal(again[1], ids)          # deterministic


def test_shard_bounds_and_rank_offsets():
    from xmh import sharded
    assert sharded.shard_bounds(10, 3) == [0, 4, 7, 10]
    assert sharded.shard_bounds(8, 8) == list(range(9))
    ha = torch.tensor([[[1, 2, 0]], [[0, 1, 3]]])        # [world=2, Q=1, nb=3]
    hr = torch.tensor([[[1, 0, 0]], [[0, 1, 1]]])
    b0 = sharded.rank_offsets(ha, hr, 0)
    b1 = sharded.rank_offsets(ha, hr, 1)
    assert b0[0].tolist() == [[0, 1, 4]] and b1[0].tolist() == [[1, 3, 4]]
    assert b0[1].tolist() == [[0, 1, 2]] and b1[1].tolist() == [[1, 1, 2]]
    assert b0[2].tolist() == [3] and b1[2].tolist() == [3]


def test_load_backbone_from_checkpoint_file_and_weight_rounding(tmp_path):
    """models/base.py:18-31 file branch (torch.jit.load fails on a plain state_dict -> torch.load), architecture
    inference from shapes (model.py:438-489) and the fp16 rounding of convert_weights (:415-436)."""
    import xmh.models  # noqa: F401
    from xmh.models import weights as W
    sd = W.synth_clip_state_dict(3, vision_layers=1, transformer_layers=2, vocab_size=1000)
    sd["input_resolution"], sd["context_length"], sd["vocab_size"] = torch.tensor(224), torch.tensor(77), torch.tensor(1000)
    path = str(tmp_path / "clip_small.pt")
    torch.save(dict(sd), path)
    model = registry.get_model_class("DCMHT").from_config(Config({"clip_path": path}), output_dim=32)
    bb = model.backbone
    assert len(bb.visual.transformer.resblocks) == 1 and len(bb.transformer.resblocks) == 2 and bb.vocab_size == 1000
    w = bb.visual.transformer.resblocks[0].attn.in_proj_weight.detach()
    assert torch.equal(w, sd["visual.transformer.resblocks.0.attn.in_proj_weight"].half().float())      # fp16-rounded
    ln = bb.visual.ln_pre.weight.detach()
    assert torch.equal(ln, sd["visual.ln_pre.weight"])                                                    # LayerNorm stays fp32
    keys = set(model.state_dict().keys())
    assert "backbone.visual.conv1.weight" in keys and "hash.img_hash.fc2.weight" in keys and "hash.txt_hash.norm.weight" in keys
    assert model.hash.img_hash.fc2.weight.shape == (64, 512)                                             # 2K outputs (hash_scale = 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.encode_image(torch.zeros(1, 3, 224, 224))                                                  # product path needs the GPU
    model.freezen()
    assert not any(p.requires_grad for p in model.parameters())
    model.unfreezen()
    assert all(p.requires_grad for p in model.parameters())


def test_tower_helper_runs_back_to_back_without_a_gpu(monkeypatch):
    """xmh.towers.run_both: image tower first, text tower second, results in (image, text) order; XMH_TOWER_STREAMS=0 (and a machine
    without HIP) selects the single-stream path."""
    from xmh import towers
    order = []
    monkeypatch.setenv("XMH_TOWER_STREAMS", "0")
    out = towers.run_both(lambda: order.append("image") or "I", lambda: order.append("text") or "T")
    assert out == ("I", "T") and order == ["image", "text"]


def test_loss_display_line_is_the_reference_format():
    """runners/base.py:359-377 print_loss_dict on the dictionary our_loss returns (models/DCMHT/DCMHT.py:126-146)"""
    import logging

    from xmh.runners.methods import DCMHTTrainer

    class Opt:
        def get_lr(self):
            return [1e-3, 1e-5, 1e-3]

    lines = []
    handler = logging.Handler()
    handler.emit = lambda rec: lines.append(rec.getMessage())
    t = DCMHTTrainer.__new__(DCMHTTrainer)
    t.logger = logging.getLogger("xmh-test-display")
    t.logger.setLevel(logging.INFO)
    t.logger.addHandler(handler)
    t.loss_type, t.epochs, t.train_loader, t.optimizer = "l1", 100, [0] * 79, Opt()
    t.print_loss_dict({"All loss": 1.5, "Intra": {"Positive": 0.25, "Negative": 0.5},
                       "Inter": {"Positive": {"i2t": 1, "t2i": 2}, "Negative": {"i2t": 3, "t2i": 4}}, "Quan": {"Image": 0.4, "Text": 0.3}},
                      bits=16, epoch=3, times=40)
    assert lines == [">>>>>> Display (l1 loss-16) >>>>>> [3/100], [40/79]: All loss: 1.5, Intra: Positive: 0.25, Negative: 0.5, "
                     "Inter: Positive: i2t: 1, t2i: 2, Negative: i2t: 3, t2i: 4, Quan: Image: 0.4, Text: 0.3, "
                     "lr: 0.000010000-0.001000000"]


def _scripted_archive(state_dict, path):
    """a TorchScript archive whose ``state_dict()`` has exactly these dotted names -- what OpenAI's ViT-B-32.pt is to the reference
    (models/base.py:21-23: ``torch.jit.load(clipPath).state_dict()``; every shipped config names ./ViT-B-32.pt)"""
    class Holder(torch.nn.Module):
        def forward(self):
            return torch.zeros(1)

    root = Holder()
    for key, value in state_dict.items():
        node, parts = root, key.split(".")
        for name in parts[:-1]:
            if not hasattr(node, name):
                node.add_module(name, Holder())
            node = getattr(node, name)
        if value.is_floating_point() and value.dim() > 0:
            node.register_parameter(parts[-1], torch.nn.Parameter(value.clone(), requires_grad=False))
        else:
            node.register_buffer(parts[-1], value.clone())
    torch.jit.save(torch.jit.script(root), path)


def test_load_backbone_takes_the_torchscript_branch(tmp_path):
    """models/base.py:21-23 -- the branch every shipped config takes.  A scripted module holding a 1-layer
    CLIP-shaped state_dict (plus the three scalar buffers OpenAI's archive carries, which build_model deletes, model.py:471-473) is
    saved with torch.jit.save and loaded through load_backbone; the plain-state_dict file of the same tensors must give the same model."""
    import xmh.models  # noqa: F401
    from xmh.models import weights as W
    sd = W.synth_clip_state_dict(5, vision_layers=1, transformer_layers=1, vocab_size=600)
    sd["input_resolution"], sd["context_length"], sd["vocab_size"] = torch.tensor(224), torch.tensor(77), torch.tensor(600)
    jit_path, plain_path = str(tmp_path / "ViT-tiny.pt"), str(tmp_path / "plain.pt")
    _scripted_archive(sd, jit_path)
    torch.save(dict(sd), plain_path)
    loaded = torch.jit.load(jit_path, map_location="cpu").state_dict()              # the archive really is TorchScript and carries the names
    assert set(loaded) == set(sd) and torch.equal(loaded["visual.conv1.weight"], sd["visual.conv1.weight"])
    with pytest.raises(RuntimeError):
        torch.jit.load(plain_path)                                                  # ... and the plain file really takes the other branch
    a = registry.get_model_class("DSPH").from_config(Config({"clip_path": jit_path}), output_dim=32)
    b = registry.get_model_class("DSPH").from_config(Config({"clip_path": plain_path}), output_dim=32)
    sa, sb = a.backbone.state_dict(), b.backbone.state_dict()
    assert set(sa) == set(sb) and len(a.backbone.visual.transformer.resblocks) == 1 and a.backbone.vocab_size == 600
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert torch.equal(sa["visual.conv1.weight"], sd["visual.conv1.weight"].half().float())


def test_dsph_state_dict_has_the_reference_checkpoint_keys():
    """the key list of the REFERENCE's DSPH class (tests/golden/runner.npz, DSPH_state_keys, written by
    oracle/make_golden_runner.py from models/DSPH/DSPH.py) -- including ``hyp.proxies`` of its loss module -- equals this package's,
    so a reference DSPH checkpoint loads strictly (runners/base.py:103-105)."""
    import numpy as np
    import xmh.models  # noqa: F401
    from conftest import GOLDEN
    from oracle import runner_fixture as RF
    g = np.load(os.path.join(GOLDEN, "runner.npz"))
    want = sorted(str(k) for k in g["DSPH_state_keys"])
    K = RF.CASES["DSPH"]
    model = registry.get_model_class("DSPH").from_config(
        Config({"clip_path": "synthetic:%d:vision_layers=%d,transformer_layers=%d" % (RF.SEED, RF.CLIP_LAYERS, RF.CLIP_LAYERS), "numclass": RF.NUM_CLASSES}), output_dim=K)
    sd = model.state_dict()
    assert sorted(k for k in sd if not k.endswith("num_batches_tracked")) == want and "hyp.proxies" in want
    assert tuple(sd["hyp.proxies"].shape) == (RF.NUM_CLASSES, K)
    model.load_state_dict({k: torch.zeros_like(v) for k, v in sd.items()}, strict=True)


@pytest.mark.timeout(900)
def test_the_library_builds_from_scratch_with_hipcc(tmp_path):
    """build() normally finds the in-tree libxmh.so newer than its sources and `make` does nothing.  Here every
    csrc/*.hip is compiled for gfx950 from scratch into a temporary directory (hipcc cross-compiles without a GPU), linked, and the fresh
    library must export exactly what include/xmh.h declares -- the build is exercised, not assumed."""
    import ctypes
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "clip-based-cross-modal-hash_amd")
    so = tmp_path / "libxmh_fresh.so"
    r = subprocess.run(["make", "-C", pkg, "-j", str(min(8, os.cpu_count() or 1)), "BUILD=%s" % (tmp_path / "obj"), "TARGET=%s" % so],
                       capture_output=True, text=True, timeout=880)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.count("--offload-arch=gfx950") >= len([f for f in os.listdir(os.path.join(pkg, "csrc")) if f.endswith(".hip")])
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "xmh.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(xmh_[a-z0-9_]+)\s*\(", src)))
    import torch  # noqa: F401  (one HIP runtime per process: torch's, loaded first -- see xmh/_lib.py)
    lib = ctypes.CDLL(str(so))
    assert not [n for n in declared if not hasattr(lib, n)]
    assert lib.xmh_version() >= 100
