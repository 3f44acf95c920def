"""Generate tests/golden/runner.npz by RUNNING the reference's own runners and models on the 8-query / 24-gallery synthetic set
(SURVEY 8c): DCMHTTrainer.get_code + valid (runners/base.py:242-266, :307-339), MITHTrainer's generate_hash override
(runners/MITH/runner.py:125-131) through the same get_code/valid, DSPHTrainer (runners/DSPH/runner.py; models/DSPH/DSPH.py:15-48) likewise, and the TwDH class (models/TwDH/TwDH.py:34-85, instantiated
with the centre / transform matrices the reference SHIPS under data/transformer/TwDH/coco) through TwDHTrainer.get_code / valid
(runners/TwDH/runner.py:145-228).  Recorded per method: the four code buffers, the four mAPs of the log line, the log line
itself, the arrays of last.mat; for TwDH also the three shipped [1024, 2*short] transform matrices (inputs of the GPU test).

The trainers are created with object.__new__ (their constructors build datasets from files that are not in the tree and start
the training loop); every attribute get_code / valid read is set by hand, the methods themselves run unmodified.
TEST INFRASTRUCTURE ONLY; runs in the build container.   python oracle/make_golden_runner.py"""
import logging
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "clip-based-cross-modal-hash_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from oracle import _ref_import  # noqa: E402
from oracle import runner_fixture as RF  # noqa: E402
from oracle.make_golden_encode import load_weights_module  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TWDH_DATA = os.path.join(_ref_import.REF, "data", "transformer", "TwDH", "coco")


class Capture(logging.Handler):
    def __init__(self):
        super().__init__()
        self.lines = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def make_logger(name):
    lg = logging.getLogger("golden-" + name)
    lg.setLevel(logging.DEBUG)
    lg.propagate = False
    cap = Capture()
    lg.handlers = [cap]
    return lg, cap


def bare_trainer(cls, model, K, tmp, name):
    t = object.__new__(cls)
    q, r = RF.datasets()
    lg, cap = make_logger(name)
    t.cfg, t.device, t.distributed, t.model_ddp, t.world_size = None, "cpu", False, None, 1
    t.model = model.float().eval()
    t.output_dim, t.query_num, t.retrieval_num = K, len(q), len(r)
    t.query_labels, t.retrieval_labels = q.get_all_label(), r.get_all_label()
    t.query_loader = DataLoader(q, batch_size=RF.BATCH, shuffle=False)
    t.retrieval_loader = DataLoader(r, batch_size=RF.BATCH, shuffle=False)
    t.save_dir, t.logger, t.epochs = tmp, lg, 1
    t.max_mapi2t = t.max_mapt2i = 0
    t.best_epoch_i = t.best_epoch_t = 0
    from common.calc_utils import calc_map_k
    t.calc_map_k = calc_map_k
    t.hash_scale = 2
    return t, cap


def maps_from(line):
    g = re.search(r"MAP\(i->t\): ([\d.eE+-]+|nan), MAP\(t->i\): ([\d.eE+-]+|nan), MAP\(t->t\): ([\d.eE+-]+|nan), MAP\(i->i\): ([\d.eE+-]+|nan)", line)
    return np.array([float(x) for x in g.groups()], dtype=np.float64)            # order of the log line: i2t, t2i, t2t, i2i


def main():
    _ref_import.setup()
    for pkg in ("models.TwDH", "runners.TwDH"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(_ref_import.REF, *pkg.split("."))]
            sys.modules[pkg] = m
    # the reference imports these at module level; neither is installed here and neither is on the eval path
    for missing in ("tqdm",):
        try:
            __import__(missing)
        except ImportError:
            tm = types.ModuleType(missing)
            tm.tqdm = lambda x, *a, **k: x
            sys.modules[missing] = tm
    W = load_weights_module()
    import scipy.io as scio
    torch.set_num_threads(8)
    rec = {"seed": RF.SEED}
    with tempfile.TemporaryDirectory() as tmp:
        clip_file = os.path.join(tmp, "clip_synth.pt")
        torch.save(W.synth_clip_state_dict(RF.SEED, **RF.clip_overrides()), clip_file)

        # ---- DCMHT and MITH: BaseTrainer.get_code / valid --------------------------------------------------------------
        from models.DCMHT.DCMHT import DCMHT
        from models.MITH.MITH import MITH
        from runners.DCMHT.runner import DCMHTTrainer
        from runners.MITH.runner import MITHTrainer

        class MITHLoadable(MITH):
            """the reference's MITH.__init__ unpacks two values from load_backbone(return_patches=True), which returns three
            (models/base.py:26 vs models/MITH/MITH.py:34): as shipped the class cannot be constructed.  This shim drops the
            middle value (the token count, unused by MITH); everything else is the reference's code."""

            def load_backbone(self, clipPath, return_patches=False):
                r = super().load_backbone(clipPath, return_patches)
                return r[0], r[-1]

        # DSPH (configs[3]'s method): the reference's own class and trainer; its constructor reads the HyP loss threshold
        # out of models/DSPH/loss/codetable.xlsx through xlrd, which is not in the image -- _ref_import stands in for that one call chain
        # over the real workbook.  DSPHTrainer has no overrides on the evaluation path: BaseTrainer.generate_hash / make_hash_code (sign_).
        from models.DSPH.DSPH import DSPH
        from runners.DSPH.runner import DSPHTrainer

        for arch, mcls, tcls in (("DCMHT", DCMHT, DCMHTTrainer), ("MITH", MITHLoadable, MITHTrainer), ("DSPH", DSPH, DSPHTrainer)):
            K = RF.CASES[arch]
            cfg_model = {"clip_path": clip_file, "numclass": RF.NUM_CLASSES} if arch == "DSPH" else {"clip_path": clip_file}
            model = mcls.from_config(cfg_model, output_dim=K, train_num=RF.RETRIEVAL_NUM)
            sd = model.state_dict()
            sd.update(RF.head_state(W, "%s%d" % (arch, K), sd))
            model.load_state_dict(sd)
            t, cap = bare_trainer(tcls, model, K, os.path.join(tmp, arch), arch)
            os.makedirs(t.save_dir, exist_ok=True)
            with torch.no_grad():
                q_img, q_txt = t.get_code(t.query_loader, t.query_num)
                r_img, r_txt = t.get_code(t.retrieval_loader, t.retrieval_num)
                t.valid(0, k=None)
            line = [ln for ln in cap.lines if ln.startswith(">>>>>> [0/1]")][-1]
            mat = scio.loadmat(os.path.join(t.save_dir, "mat_files", "last.mat"))
            assert np.array_equal(mat["q_img"], q_img.numpy()) and np.array_equal(mat["r_txt"], r_txt.numpy())
            rec.update({arch + "_q_img": q_img.numpy(), arch + "_q_txt": q_txt.numpy(), arch + "_r_img": r_img.numpy(), arch + "_r_txt": r_txt.numpy(),
                        arch + "_maps_i2t_t2i_t2t_i2i": maps_from(line), arch + "_log_line": np.array(line),
                        arch + "_state_keys": np.array(sorted(k for k in sd if not k.endswith("num_batches_tracked"))),
                        arch + "_mat_q_l": mat["q_l"], arch + "_mat_r_l": mat["r_l"], arch + "_mat_q_img_dtype": np.array(str(mat["q_img"].dtype)),
                        arch + "_mat_q_l_dtype": np.array(str(mat["q_l"].dtype)),
                        arch + "_files": np.array(sorted(os.listdir(os.path.join(t.save_dir, "mat_files"))) + sorted(f for f in os.listdir(t.save_dir) if f.endswith(".pth")))})
            print(arch, K, line)

        # ---- TwDH: the reference class on the shipped centre / transform matrices --------------------------------------
        import importlib
        sys.modules.pop("models.DCMHT.hash", None)          # _ref_import's bare namespace: TwDH imports HashLayer from the package itself
        importlib.import_module("models.DCMHT.hash")
        from models.TwDH.TwDH import TwDH
        from runners.TwDH.runner import TwDHTrainer
        LONG = RF.CASES["TwDH"]
        model = TwDH(cfg=None, long_dim=LONG, short_dim=16, clipPath=clip_file, train_num=RF.RETRIEVAL_NUM, hash_func="softmax",
                     long_center=os.path.join(TWDH_DATA, "long", "%d.pkl" % LONG), short_center=os.path.join(TWDH_DATA, "short"),
                     trans=os.path.join(TWDH_DATA, "trans", str(LONG)))
        sd = model.state_dict()
        sd.update(RF.head_state(W, "TwDH%d" % LONG, sd))
        model.load_state_dict(sd)
        t, cap = bare_trainer(TwDHTrainer, model, LONG, os.path.join(tmp, "TwDH"), "TwDH")
        os.makedirs(t.save_dir, exist_ok=True)
        t.long_dim = LONG
        t.max_short = {d: {"i2t": 0, "t2i": 0} for d in model.get_short_dims()}
        t.best_epoch_short = {d: {"i2t": 0, "t2i": 0} for d in model.get_short_dims()}
        # the reference keys max_short by int (runner.py:44-47) and looks it up by the str key of the code dict (:209): KeyError in
        # its own valid(); accept both spellings so that the unmodified method runs
        for d in list(t.max_short):
            t.max_short[str(d)] = t.max_short[d]
            t.best_epoch_short[str(d)] = t.best_epoch_short[d]
        with torch.no_grad():
            ql_i, ql_t, qs_i, qs_t = t.get_code(t.query_loader, t.query_num)
            rl_i, rl_t, rs_i, rs_t = t.get_code(t.retrieval_loader, t.retrieval_num)
            t.valid(0, k=None)
        rec.update({"TwDH_short_dims": np.array(sorted(int(k) for k in qs_i)), "TwDH_long_q_img": ql_i.numpy(), "TwDH_long_q_txt": ql_t.numpy(),
                    "TwDH_long_r_img": rl_i.numpy(), "TwDH_long_r_txt": rl_t.numpy()})
        for k in qs_i:
            rec.update({"TwDH_%s_q_img" % k: qs_i[k].numpy(), "TwDH_%s_q_txt" % k: qs_t[k].numpy(), "TwDH_%s_r_img" % k: rs_i[k].numpy(),
                        "TwDH_%s_r_txt" % k: rs_t[k].numpy(), "TwDH_trans_%s" % k: model.trans[k].numpy()})
        for ln in cap.lines:
            if ln.startswith(">>>>>> [0/1], Long"):
                rec["TwDH_long_maps_i2t_t2i_t2t_i2i"], rec["TwDH_long_log_line"] = maps_from(ln), np.array(ln)
            m = re.match(r">>>>>> \[0/1\], Short, (\d+) Bit", ln)
            if m:
                rec["TwDH_%s_maps_i2t_t2i_t2t_i2i" % m.group(1)], rec["TwDH_%s_log_line" % m.group(1)] = maps_from(ln), np.array(ln)
            print(ln) if ln.startswith(">>>>>>") else None
        rec["TwDH_state_keys"] = np.array(sorted(k for k in sd if not k.endswith("num_batches_tracked")))
        rec["TwDH_files"] = np.array(sorted(os.listdir(os.path.join(t.save_dir, "mat_files"))))
    np.savez_compressed(os.path.join(OUT, "runner.npz"), **rec)
    print("runner golden:", os.path.getsize(os.path.join(OUT, "runner.npz")), "bytes;", len(rec), "arrays")


if __name__ == "__main__":
    main()
