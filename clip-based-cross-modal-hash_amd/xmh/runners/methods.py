"""Per-method runners registered under the reference's names (runners/<METHOD>/runner.py).

``from_config(rank, world_size, distributed, cfg, logger)`` keeps the positional order mp.spawn relies on
(main.py:44-49, runners/DCMHT/runner.py:40-80); like in the reference, constructing a runner builds dataset and
model and then calls ``run()``."""
from __future__ import annotations

import torch

from .. import retrieval as R
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
        super().__init__(cfg, *a, **k)

    @classmethod
    def make_hash_code(cls, code):
        if isinstance(code, list):
            code = torch.stack(code).permute(1, 0, 2)
        return R.pack_pair_argmax(code.reshape(code.shape[0], -1)).unpack()

    @classmethod
    def pack_hash_code(cls, code, out, row_index, flags):
        R.pack_pair_argmax(code.reshape(code.shape[0], -1), out=out, row_index=row_index)


@registry.register_runner("DSPHTrainer")
class DSPHTrainer(_MethodTrainer):
    """runners/DSPH/runner.py: base generate_hash + sign quantiser."""


@registry.register_runner("MITHTrainer")
class MITHTrainer(_MethodTrainer):
    """runners/MITH/runner.py:125-131: code = sign(cls_hash + tokens_hash); the text tower gets the padding mask."""

    def generate_hash(self, image, text, key_padding_mask=None):
        _, img_cls_hash, tokens_hash_i, _, _, txt_cls_hash, tokens_hash_t, _ = self.model(image, text, key_padding_mask=key_padding_mask)
        return img_cls_hash + tokens_hash_i, txt_cls_hash + tokens_hash_t
