"""Shared between oracle/make_golden_runner.py (build container, feeds the REFERENCE's runners) and tests/test_gpu_runner.py
(GPU box, feeds this package's runners): the deterministic inputs of the runner golden.  TEST INFRASTRUCTURE ONLY.

Both sides build their model with its own random head initialisation and then overwrite every ``hash.*`` tensor of its
state_dict by the rule below, keyed by the state_dict KEY NAME -- which the plugin contract requires to be identical on both
sides (reference checkpoints must stay loadable).  The CLIP backbone comes from xmh.models.weights (a 2-layer
ViT-B/32-shaped state_dict, written to a file for the reference's load_backbone, named by a "synthetic:" path here), the
dataset from xmh.dataset.synthetic.SyntheticPairs."""
import torch

SEED = 1814
CLIP_LAYERS = 2
QUERY_NUM, RETRIEVAL_NUM, NUM_CLASSES, BATCH = 8, 24, 24, 5          # "a 8-query/24-gallery synthetic set"
CASES = {"DCMHT": 16, "MITH": 64, "TwDH": 512, "DSPH": 128}


def head_state(W, tag, state_dict):
    """deterministic replacement for every ``hash.*`` entry of a model's state_dict"""
    new = {}
    for k, v in state_dict.items():
        if not k.startswith("hash."):
            continue
        name = "runner.%s.%s" % (tag, k.replace("gcl_t.", "gcl_i."))            # MITH: gcl_t aliases gcl_i (models/MITH/hash/hash.py:218)
        if k.endswith("num_batches_tracked") or k.endswith("position.pe"):
            new[k] = v.clone()
        elif k.endswith("running_var"):
            new[k] = W.synth_tensor(SEED, name, v.shape, 0.2).abs() + 0.5
        elif k.endswith("weight") and v.dim() == 1:                              # LayerNorm / BatchNorm gains
            new[k] = 1.0 + W.synth_tensor(SEED, name, v.shape, 0.05)
        else:
            new[k] = W.synth_tensor(SEED, name, v.shape, 0.05 if v.dim() > 1 else 0.02)
    return new


def datasets():
    """(query, retrieval) SyntheticPairs with the index / label conventions of dataset/builder.py"""
    from xmh.dataset.synthetic import SyntheticPairs
    mk = lambda n, off: SyntheticPairs(n, NUM_CLASSES, 224, 32, SEED, 0.1, index_offset=off)      # noqa: E731
    return mk(QUERY_NUM, 0), mk(RETRIEVAL_NUM, QUERY_NUM)


def clip_overrides():
    return dict(vision_layers=CLIP_LAYERS, transformer_layers=CLIP_LAYERS)
