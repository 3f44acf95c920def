"""Per-method runners registered under the reference's names (runners/<METHOD>/runner.py).

``from_config(rank, world_size, distributed, cfg, logger)`` keeps the positional order mp.spawn relies on
(main.py:44-49, runners/DCMHT/runner.py:40-80); like in the reference, constructing a runner builds dataset and
model and then calls ``run()``."""
from __future__ import annotations

import os

import torch

from .. import retrieval as R
from .. import towers
from ..common.register import registry
from .base import BaseTrainer


class _MethodTrainer(BaseTrainer):
    MODEL_ARCH = None

    def __init__(self, cfg, is_train=True, logger=None, device=None, world_size=torch.cuda.device_count(), output_dim=16,
                 train_num=10000, query_num=5000, epochs=100, save_dir="./result", batch_size=128, num_workers=4, pin_memory=True,
                 shuffle=True, display_step=20, top_k=5000, model_state="", distributed=False, autorun=True, **kwags):
        super().__init__(cfg=cfg, is_train=is_train, logger=logger, device=device, output_dim=output_dim, train_num=train_num,
                         distributed=distributed, query_num=query_num, epochs=epochs, save_dir=save_dir, display_step=display_step,
                         top_k=top_k, model_state=model_state, batch_size=batch_size, world_size=world_size)
        self.build_dataset(cfg.dataset, train_num=train_num, query_num=query_num, batch_size=batch_size, num_workers=num_workers,
                           pin_memory=pin_memory, shuffle=shuffle)
        self.build_model(cfg.model, output_dim=output_dim)
        if autorun:
            self.run()

    @classmethod
    def from_config(cls, rank=0, world_size=torch.cuda.device_count(), distributed=False, cfg=None, logger=None, **extra):
        assert cfg is not None, "config is None!"
        run = cfg.run
        return cls(cfg, is_train=run.get("is_train", False), logger=logger, device=rank if distributed else run.get("device", 0),
                   output_dim=run.get("output_dim", 16), train_num=run.get("train_num", 10000), query_num=run.get("query_num", 5000),
                   epochs=run.get("epochs", 10), save_dir=run.get("save_dir", "./result"), batch_size=run.get("batch_size", 128),
                   num_workers=run.get("num_workers", 4), pin_memory=run.get("pin_memory", True), shuffle=run.get("shuffle", True),
                   model_state=run.get("resume_model", ""), display_step=run.get("display_step", 20), top_k=run.get("top_k", None),
                   world_size=world_size, distributed=distributed, **extra)


@registry.register_runner("DCMHTTrainer")
class DCMHTTrainer(_MethodTrainer):
    """runners/DCMHT/runner.py: the model emits 2K pair probabilities; bit j = +1 iff p[j,1] > p[j,0] (:82-95)."""

    def __init__(self, cfg, *a, **k):
        self.hash_func = cfg.model.get("hash_func", "softmax")
        assert self.hash_func == "softmax", "DCMHT must adopt the 'softmax' hash technique."
        self.hash_scale = 2
        self.loss_type = k.get("loss_type", "l1")               # runners/DCMHT/runner.py:26-30: only named in the display line
        super().__init__(cfg, *a, **k)

    @classmethod
    def make_hash_code(cls, code):
        if isinstance(code, list):
            code = torch.stack(code).permute(1, 0, 2)
        return R.pack_pair_argmax(code.reshape(code.shape[0], -1)).unpack()

    @classmethod
    def pack_hash_code(cls, code, out, row_index, flags):
        R.pack_pair_argmax(code.reshape(code.shape[0], -1), out=out, row_index=row_index)

    def compute_loss(self, img_hash=None, txt_hash=None, label=None, index=None, epoch=0, times=0, global_step=0, **kwags):
        """runners/DCMHT/runner.py:97-105: the objective of one batch.  The returned loss is differentiable with respect to
        img_hash / txt_hash (xmh_loss.hip behind torch.autograd); the display line of :101-103 needs the training loop's
        loader and optimiser, which this package does not build, and is skipped without them."""
        all_loss, loss_dict = self.model.object_function(img_hash=img_hash, txt_hash=txt_hash, labels=label, indexs=index, **kwags)
        if global_step % self.display_step == 0 and getattr(self, "train_loader", None) is not None and getattr(self, "optimizer", None) is not None:
            self.print_loss_dict(loss_dict, bits=img_hash.shape[-1] // self.hash_scale, epoch=epoch, times=times)
        return all_loss


@registry.register_runner("DSPHTrainer")
class DSPHTrainer(_MethodTrainer):
    """runners/DSPH/runner.py: base generate_hash + sign quantiser."""


@registry.register_runner("MITHTrainer")
class MITHTrainer(_MethodTrainer):
    """runners/MITH/runner.py:125-131: code = sign(cls_hash + tokens_hash); the text tower gets the padding mask."""

    def generate_hash(self, image, text, key_padding_mask=None):
        _, img_cls_hash, tokens_hash_i, _, _, txt_cls_hash, tokens_hash_t, _ = self.model(image, text, key_padding_mask=key_padding_mask)
        return img_cls_hash + tokens_hash_i, txt_cls_hash + tokens_hash_t


@registry.register_runner("TwDHTrainer")
class TwDHTrainer(DCMHTTrainer):
    """runners/TwDH/runner.py: one forward yields the long code and every short code (:138-143); evaluation reports the four
    mAPs per code length (:181-228).  Code streams: "long" (long_dim bits) and one per short length, all quantised by the
    DCMHT pair-argmax (:82-95 of the DCMHT runner, inherited)."""

    def __init__(self, cfg, *a, **k):
        self.long_dim = cfg.model.get("long_dim", 512)
        self.max_short, self.best_epoch_short = {}, {}
        super().__init__(cfg, *a, **k)

    def build_model(self, cfg_model, output_dim=16):
        super().build_model(cfg_model, output_dim=output_dim)
        for item in self.model.get_short_dims():
            self.max_short[str(item)] = {"i2t": 0, "t2i": 0}
            self.best_epoch_short[str(item)] = {"i2t": 0, "t2i": 0}

    def generate_hash(self, image, text, key_padding_mask=None):
        (long_image_hash, short_image_hash), (long_text_hash, short_text_hash) = towers.run_both(
            lambda: self.model.encode_image(image), lambda: self.model.encode_text(text))
        return long_image_hash, short_image_hash, long_text_hash, short_text_hash

    def generate_hashes(self, image, text, key_padding_mask=None):
        li, si, lt, st = self.generate_hash(image, text, key_padding_mask)
        out = {"long": (li, lt)}
        for key in si:
            out[key] = (si[key], st[key])
        return out

    def code_streams(self):
        dims = {"long": self.long_dim}
        for d in self.model.get_short_dims():
            dims[str(d)] = int(d)
        return dims

    def encode_shard(self, data_loader, length):
        return self.encode_streams(data_loader, length)["long"]

    def get_code(self, data_loader, length):
        """reference :145-179 -- (long_img, long_txt, {short: img}, {short: txt}) fp32 +-1 buffers, complete on every rank."""
        streams = self.encode_streams(data_loader, length)
        full = {n: (self._gather_packed(i, length).unpack(), self._gather_packed(t, length).unpack()) for n, (i, t) in streams.items()}
        return (full["long"][0], full["long"][1], {n: v[0] for n, v in full.items() if n != "long"},
                {n: v[1] for n, v in full.items() if n != "long"})

    def _evaluate_streams(self, k):
        self._qlab = self._rlab = None
        qs = self.encode_streams(self.query_loader, self.query_num)
        rs = self.encode_streams(self.retrieval_loader, self.retrieval_num)
        out = {}
        for name in qs:
            q_img, q_txt = self._gather_packed(qs[name][0], self.query_num), self._gather_packed(qs[name][1], self.query_num)
            r_img, r_txt = rs[name]
            maps = (self._map(q_img, r_txt, k), self._map(q_txt, r_img, k), self._map(q_img, r_img, k), self._map(q_txt, r_txt, k))
            out[name] = (maps, (q_img, q_txt, r_img, r_txt))
        return out

    def valid(self, epoch, k=None):
        assert self.query_loader is not None and self.retrieval_loader is not None
        save_dir = os.path.join(self.save_dir, "mat_files")
        os.makedirs(save_dir, exist_ok=True)
        self.logger.info("Valid.")
        results = self._evaluate_streams(k)
        for name, ((mAPi2t, mAPt2i, mAPi2i, mAPt2t), codes) in results.items():
            bits = codes[0].K
            if name == "long":                                   # reference valid_each, short is None (:194-208)
                if self.max_mapi2t < mAPi2t:
                    self.best_epoch_i = epoch
                    self._save_codes(codes, os.path.join(save_dir, "i2t-long.mat"))
                    if self._is_writer():
                        self.save_model(save_dir=self.save_dir, epoch=epoch)
                self.max_mapi2t = max(self.max_mapi2t, mAPi2t)
                if self.max_mapt2i < mAPt2i:
                    self.best_epoch_t = epoch
                    self._save_codes(codes, os.path.join(save_dir, "t2i-long.mat"))
                    if self._is_writer():
                        self.save_model(save_dir=self.save_dir, epoch=epoch)
                self.max_mapt2i = max(self.max_mapt2i, mAPt2i)
                self.logger.info(f">>>>>> [{epoch}/{self.epochs}], Long, {bits} Bit, MAP(i->t): {mAPi2t}, MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, "
                                 f"MAP(i->i): {mAPi2i}, MAX MAP(i->t): {self.max_mapi2t}, epoch: {self.best_epoch_i}, MAX MAP(t->i): {self.max_mapt2i}, "
                                 f"epoch: {self.best_epoch_t}")
            else:                                                # short codes (:209-228)
                if self.max_short[name]["i2t"] < mAPi2t:
                    self.best_epoch_short[name]["i2t"] = epoch
                    self._save_codes(codes, os.path.join(save_dir, f"i2t-short-{name}.mat"))
                self.max_short[name]["i2t"] = max(self.max_short[name]["i2t"], mAPi2t)
                if self.max_short[name]["t2i"] < mAPt2i:
                    self.best_epoch_short[name]["t2i"] = epoch
                    self._save_codes(codes, os.path.join(save_dir, f"t2i-short-{name}.mat"))
                self.max_short[name]["t2i"] = max(self.max_short[name]["t2i"], mAPt2i)
                self.logger.info(f">>>>>> [{epoch}/{self.epochs}], Short, {bits} Bit, MAP(i->t): {mAPi2t}, MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, "
                                 f"MAP(i->i): {mAPi2i}, MAX MAP(i->t): {self.max_short[name]['i2t']}, epoch: {self.best_epoch_short[name]['i2t']}, "
                                 f"MAX MAP(t->i): {self.max_short[name]['t2i']}, epoch: {self.best_epoch_short[name]['t2i']}")
        return {name: maps for name, (maps, _) in results.items()}
