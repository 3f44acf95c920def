"""Generate tests/golden/encode_twdh.npz by RUNNING the reference's pieces of the TwDH forward (models/TwDH/TwDH.py:66-85):
the DCMHT HashLayer at long_dim = 512 (models/DCMHT/hash/hash.py) and ``quantization(long_hash.matmul(trans))`` for two
short lengths, plus the runner's quantiser (runners/DCMHT/runner.py:82-95), on a seeded transform matrix (head-level
golden, B = 24).  The TwDH CLASS itself -- instantiated with the centre / transform matrices the reference ships under
data/transformer/TwDH/coco -- and TwDHTrainer.get_code / valid are pinned by oracle/make_golden_runner.py -> runner.npz.
TEST INFRASTRUCTURE ONLY; runs in the build container.   python oracle/make_golden_twdh.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_import  # noqa: E402
from oracle.make_golden_encode import load_weights_module  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1814
LONG = 512


def main():
    _ref_import.setup()
    Wt = load_weights_module()
    from models.DCMHT.hash.hash import HashLayer
    from runners.DCMHT.runner import DCMHTTrainer
    ref = HashLayer(feature_size=512, outputDim=LONG, num_heads=8, batch_first=True, hash_func_="softmax").eval()
    new = {}
    for k, v in ref.state_dict().items():
        if k.endswith("num_batches_tracked"):
            new[k] = v
        elif k.endswith("running_var"):
            new[k] = Wt.synth_tensor(SEED, "twdh%d." % LONG + k, v.shape, 0.2).abs() + 0.5
        elif k.endswith("norm.weight"):
            new[k] = 1.0 + Wt.synth_tensor(SEED, "twdh%d." % LONG + k, v.shape, 0.1)
        else:
            new[k] = Wt.synth_tensor(SEED, "twdh%d." % LONG + k, v.shape, 0.05 if v.dim() > 1 else 0.02)
    ref.load_state_dict(new)
    emb = Wt.synth_tensor(SEED, "twdh_input", (24, 512), 0.5)
    rec = {"emb": emb.numpy()}
    with torch.no_grad():
        long_i, long_t = ref.encode_img(emb), ref.encode_txt(emb)
        rec["long_img"], rec["long_txt"] = long_i.numpy(), long_t.numpy()
        rec["long_img_code"] = DCMHTTrainer.make_hash_code(long_i.clone()).numpy()
        rec["long_txt_code"] = DCMHTTrainer.make_hash_code(long_t.clone()).numpy()
        for S in (16, 64):
            trans = Wt.synth_tensor(SEED, "twdh_trans%d" % S, (2 * LONG, 2 * S), 0.2)
            rec["trans%d" % S] = trans.numpy()
            for name, lh in (("img", long_i), ("txt", long_t)):
                short = ref.quantization(lh.matmul(trans))                      # models/TwDH/TwDH.py:73 / :83
                rec["short%d_%s" % (S, name)] = short.numpy()
                rec["short%d_%s_code" % (S, name)] = DCMHTTrainer.make_hash_code(short.clone()).numpy()
    np.savez_compressed(os.path.join(OUT, "encode_twdh.npz"), seed=SEED, **rec)
    print("twdh goldens:", {k: v.shape for k, v in rec.items()})


if __name__ == "__main__":
    main()
