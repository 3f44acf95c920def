"""``BaseTrainer``: the encode-and-retrieve driver with the reference's runner surface (runners/base.py:22-415).

Same constructor arguments, the same overridable seams (``generate_hash``, ``make_hash_code``, ``get_code``,
``valid``, ``test``, the instance attribute ``self.calc_map_k``), the same outputs (log line :338/:357, ``.mat``
keys and dtypes :397-404, ``model-<epoch>.pth`` :379-384).  What changes underneath:

* codes never exist as fp32 on the hot path: the quantiser writes bit-packed rows straight into the code buffer
  at ``index`` (xmh_pack_sign / xmh_pack_pair_argmax with row_index) and retrieval consumes the packed buffers;
  ``get_code`` still hands back ``[N, K]`` fp32 +-1 tensors to callers that ask for them (unpacked on demand);
* distributed eval keeps the gallery sharded (contiguous index ranges, one per rank) instead of all-reducing dense
  zero-initialised buffers (runners/base.py:259-264): packed query codes are all-gathered, per-shard bucket
  histograms are exchanged, every rank ranks its own shard (xmh/sharded.py);
* optimisation (``train_epoch``) is outside the encode-and-retrieve path and raises, exactly like the reference's
  own base class does (:296-297).
"""
from __future__ import annotations

import io
import os
import threading

import numpy as np
import torch
from torch import distributed as dist
from torch.utils.data import DataLoader, Sampler

from .. import _lib
from .. import retrieval as R
from .. import sharded
from .. import towers
from ..common.calc_utils import calc_map_k
from ..common.register import registry
from ..utils.logger import get_color_logger


def set_seed(seed=1814):
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class ContiguousShardSampler(Sampler):
    """rank r iterates dataset rows [bounds[r], bounds[r+1]) in order -- the contiguous partition of SURVEY 8e
    (the reference's strided DistributedSampler, runners/base.py:180-191, is an internal detail it replaces)."""

    def __init__(self, n: int, rank: int, world: int):
        b = sharded.shard_bounds(n, world)
        self.lo, self.hi = b[rank], b[rank + 1]

    def __iter__(self):
        return iter(range(self.lo, self.hi))

    def __len__(self):
        return self.hi - self.lo


class BaseTrainer:

    def __init__(self, cfg, is_train=True, device=None, world_size=torch.cuda.device_count(), output_dim=16, train_num=10000,
                 query_num=5000, epochs=100, save_dir="./result", display_step=20, top_k=5000, model_state="", batch_size=128,
                 distributed=False, logger=None, **kwags) -> None:
        set_seed(seed=cfg.run.get("seed", 1814))
        self.cfg = cfg
        self.is_train = is_train
        log_dir = cfg.run.get("log_dir", save_dir)
        name = cfg.dataset.get("name", "data") + "-" + str(device)
        self.logger = logger or get_color_logger(log_dir, name, display=(not distributed) or device == 0)
        self.rank = 0
        if distributed:
            self._init_distribution(rank=device, world_size=world_size)
        self.logger.info(f"parameters: {dict(cfg)}")
        self.device = 0 if distributed and cfg.run.get("share_gpu") else device       # the reference: device == rank (runners/base.py:82-96)
        if not distributed and torch.cuda.is_available() and device is not None:
            # libxmh launches on the CURRENT device's stream: make run.device current (the reference reaches any device
            # through .to(self.device), runners/base.py:105-107)
            # "cuda" without an index means the current device; a non-CUDA device (the reference accepts anything .to() does)
            # is left to the ops, which report that they need a GPU tensor
            d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
            if d.type == "cuda":
                torch.cuda.set_device(d.index if d.index is not None else torch.cuda.current_device())
        self.output_dim, self.train_num, self.query_num = output_dim, train_num, query_num
        self.epochs, self.display_step, self.top_k = epochs, display_step, top_k
        self.model_state, self.batch_size, self.save_dir = model_state, batch_size, save_dir
        self.encode_fuse = int(cfg.run.get("encode_fuse", 4))      # loader batches per evaluation forward (encode_shard)
        self.image_resolution = int(cfg.dataset.get("image_resolution", 224))
        os.makedirs(save_dir, exist_ok=True)
        self.global_step = 0
        self.max_mapi2t = self.max_mapt2i = 0
        self.best_epoch_i = self.best_epoch_t = 0
        self.calc_map_k = calc_map_k                     # instance attribute: the injection point (runners/base.py:78)
        self.distributed = distributed
        self.world_size = world_size if distributed else 1
        self.model_ddp = None
        self._qlab = self._rlab = None

    # ---- distributed ------------------------------------------------------------------------------------
    def _init_distribution(self, rank=0, world_size=4):
        self.rank, self.world_size = rank, world_size
        self.logger.info("Initializing distributed")
        assert self.cfg.run.get("distributed_addr"), "DDP needs the 'distributed_addr' field."
        assert self.cfg.run.get("distributed_port"), "DDP needs the 'distributed_port' field"
        os.environ["MASTER_ADDR"] = str(self.cfg.run.distributed_addr)
        os.environ["MASTER_PORT"] = str(self.cfg.run.distributed_port)
        # run.share_gpu -- a TEST HOOK that is not in the reference's configs: every rank on cuda:0 and the group over gloo (which takes
        # device tensors; RCCL refuses two ranks on one device), so the sharded evaluation runs with world_size > 1 on a one-GPU box
        share = bool(self.cfg.run.get("share_gpu"))
        torch.cuda.set_device(0 if share else rank)
        if not dist.is_initialized():
            dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world_size)     # "nccl" is RCCL on ROCm

    # ---- builders ---------------------------------------------------------------------------------------
    def build_model(self, cfg_model, output_dim=16, **kwags):
        arch = cfg_model.get("arch", "DCMHT")
        cls = registry.get_model_class(arch)
        assert cls is not None, "model '%s' is not registered (known: %s)" % (arch, registry.list_models())
        self.model = cls.from_config(cfg_model, output_dim=output_dim, train_num=self.train_num)
        if os.path.isfile(self.model_state):
            self.logger.info("loading model...")
            self.model.load_state_dict(torch.load(self.model_state, map_location=f"cuda:{self.device}"))
        self.model.float()
        self.model.to(self.device)
        self.logger.info("Building model!")
        self.logger.info(f"Output dim: {self.output_dim}")

    def build_optimizer(self, cfg_optimizer=None, parameters=None):
        raise NotImplementedError("optimisation is outside the encode-and-retrieve path (SURVEY 2.1 #13)")

    def build_dataset(self, cfg, train_num=10000, query_num=5000, batch_size=128, num_workers=4, pin_memory=True, shuffle=True):
        arch = cfg.get("arch", "transformer_dataset")
        self.logger.info(f"Using {cfg.get('name', 'synthetic')} dataset.")
        if arch in ("synthetic", "synthetic_pairs"):
            from ..dataset import build_synthetic_splits
            train_data, query_data, retrieval_data = build_synthetic_splits(cfg, train_num, query_num)
        else:
            builder = registry.get_dataset_class(arch)
            if builder is None:
                raise NotImplementedError("dataset arch '%s' is not registered (known: %s)" % (arch, registry.list_datasets()))
            if hasattr(builder, "build"):
                train_data, query_data, retrieval_data = builder.build(cfg, train_num=train_num, query_num=query_num)
            else:                                            # the reference's file layout (runners/base.py:145-159)
                from ..dataset import build_dataloader
                dataname, path = cfg.get("name", "mirflickr25k"), cfg.get("path", "./data")
                tok = registry.get_tokenizer_class(cfg.get("tokenizer_arch", "clip_tokenizer"))
                assert tok is not None, "tokenizer '%s' is not registered" % cfg.get("tokenizer_arch", "clip_tokenizer")
                train_data, query_data, retrieval_data = build_dataloader(
                    captionFile=os.path.join(path, dataname, cfg.get("txt_file", "caption.mat")),
                    indexFile=os.path.join(path, dataname, cfg.get("img_file", "index.mat")),
                    labelFile=os.path.join(path, dataname, cfg.get("label_file", "caption.mat")),
                    imageResolution=cfg.get("image_resolution", 224), maxWords=cfg.get("max_word", 32), query_num=query_num,
                    train_num=train_num, dataset_cls=arch, tokenizer=tok())
        self.build_loader(train_data, query_data, retrieval_data, batch_size, num_workers, pin_memory, shuffle)

    def build_loader(self, train_data, query_data, retrieval_data, batch_size, num_workers, pin_memory, shuffle, drop_last=False):
        self.train_labels = train_data.get_all_label() if train_data is not None else None     # no training split on this path
        self.query_labels = query_data.get_all_label()
        self.retrieval_labels = retrieval_data.get_all_label()
        self.retrieval_num = len(self.retrieval_labels)
        for nm, t in (("train", self.train_labels), ("query", self.query_labels), ("retrieval", self.retrieval_labels)):
            if t is not None:
                self.logger.info(f"{nm} shape: {tuple(t.shape)}")
        qs = rs = None
        if self.distributed:
            qs = ContiguousShardSampler(len(query_data), self.rank, self.world_size)
            rs = ContiguousShardSampler(len(retrieval_data), self.rank, self.world_size)
            batch_size = max(1, batch_size // self.world_size)
        mk = lambda d, s, sh: DataLoader(d, batch_size=batch_size, num_workers=num_workers, pin_memory=pin_memory, sampler=s,   # noqa: E731
                                         shuffle=sh, drop_last=False, collate_fn=getattr(d, "collate", None))
        self.train_loader = mk(train_data, None, shuffle and not self.distributed) if train_data is not None else None
        self.query_loader = mk(query_data, qs, False)
        self.retrieval_loader = mk(retrieval_data, rs, False)

    # ---- run --------------------------------------------------------------------------------------------
    def run(self):
        if self.is_train:
            self.train()
        else:
            self.test()

    def train(self):
        for epoch in range(self.epochs):
            self.train_epoch(epoch=epoch)
            self.valid(epoch, k=self.top_k)
        self.logger.info(f">>>>>>> FINISHED >>>>>> Best epoch, I-T: {self.best_epoch_i}, mAP: {self.max_mapi2t}, T-I: {self.best_epoch_t}, mAP: {self.max_mapt2i}")

    def train_epoch(self, epoch: int):
        raise NotImplementedError("training is outside the encode-and-retrieve path this package implements")

    def compute_loss(self, *a, **k):
        raise NotImplementedError("training is outside the encode-and-retrieve path this package implements")

    def print_loss_dict(self, loss_dict, bits=16, epoch=0, times=0):
        """runners/base.py:359-377 -- the nested loss dictionary flattened depth-first into one display line"""
        def flat(key, value):
            if isinstance(value, dict):
                return f"{key}: " + "".join(flat(k, v) for k, v in value.items())
            return f"{key}: {value}, "

        rates = "-".join("%.9f" % r for r in sorted(set(self.optimizer.get_lr())))
        self.logger.info(f">>>>>> Display ({self.loss_type} loss-{bits}) >>>>>> [{epoch}/{self.epochs}], [{times}/{len(self.train_loader)}]: "
                         + "".join(flat(k, v) for k, v in loss_dict.items()) + f"lr: {rates}")

    def change_state(self, mode):
        if mode == "train":
            self.model.train()
            self.model.unfreezen()
        else:
            self.model.eval()
            self.model.freezen()

    # ---- encode ------------------------------------------------------------------------------------------
    def generate_hash(self, image, text, key_padding_mask=None):
        return towers.run_both(lambda: self.model.encode_image(image), lambda: self.model.encode_text(text))

    @classmethod
    def make_hash_code(cls, code):
        """reference :407-410 -- sign -> -1/0/+1 fp32 (computed by xmh_pack_sign + xmh_unpack_pm1)."""
        return R.pack_sign(code).unpack()

    @classmethod
    def pack_hash_code(cls, code, out, row_index, flags):
        """quantise a batch of model outputs straight into packed rows ``row_index`` of ``out``."""
        R.pack_sign(code, out=out, row_index=row_index, flags=flags)

    def _shard(self, length):
        if not self.distributed:
            return 0, length
        b = sharded.shard_bounds(length, self.world_size)
        return b[self.rank], b[self.rank + 1]

    def _image_transform(self):
        t = self.__dict__.get("_gpu_eval_transform")
        if t is None:
            from ..dataset.preprocess import GpuEvalTransform
            t = GpuEvalTransform(int(getattr(self, "image_resolution", 224) or 224))
            self.__dict__["_gpu_eval_transform"] = t
        return t

    def generate_hashes(self, image, text, key_padding_mask=None) -> dict:
        """name -> (image_hash, text_hash) for every code stream of one forward; the base methods have one stream, ""
        (TwDH emits a long code and several short ones from the same forward, runners/TwDH/runner.py:138-143)."""
        return {"": self.generate_hash(image=image, text=text, key_padding_mask=key_padding_mask)}

    def code_streams(self) -> dict:
        """name -> code length K of every stream generate_hashes returns."""
        return {"": self.output_dim}

    def encode_streams(self, data_loader, length: int) -> dict:
        """HOT LOOP 1 (runners/base.py:250-257): name -> this rank's packed (image, text) code rows."""
        self.change_state(mode="valid")
        lo, hi = self._shard(length)
        dev = torch.device("cuda", self.device) if isinstance(self.device, int) else torch.device(self.device)
        dims = self.code_streams()
        bufs = {n: (R.empty_packed(hi - lo, K, dev, with_zero=True), R.empty_packed(hi - lo, K, dev, with_zero=True)) for n, K in dims.items()}
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        # Evaluation is per-sample independent, so several loader batches are fused into one forward: the encoder's GEMMs
        # sit on an L2-bandwidth roofline that rises with the number of rows (B=100 -> 22.5k images/s,
        # B=400 -> 28k).  `run.encode_fuse` loader batches per forward (default 4; 1 = the reference's granularity).
        fuse = max(1, int(getattr(self, "encode_fuse", 4) or 1))
        pending = []

        def flush():
            if not pending:
                return
            if isinstance(pending[0][0], list):              # undecoded photos of different sizes: one list for the group
                image = [im for p in pending for im in p[0]]
            else:
                image = torch.cat([p[0] for p in pending]) if len(pending) > 1 else pending[0][0]
            text = torch.cat([p[1] for p in pending]) if len(pending) > 1 else pending[0][1]
            kpm = None
            if pending[0][2] is not None:
                kpm = torch.cat([p[2] for p in pending]) if len(pending) > 1 else pending[0][2]
            rows = torch.cat([p[3] for p in pending]) if len(pending) > 1 else pending[0][3]
            pending.clear()
            if isinstance(image, list) or image.dtype == torch.uint8:     # raw RGB bytes: the eval transform runs on the GPU
                image = self._image_transform()(image)       # (dataset/transformer_dataset.py:38-42, Pillow-exact)
            with _lib.prof_range("encode: towers + heads (%d rows)" % rows.shape[0]):
                hashes = self.generate_hashes(image, text, kpm)
            with _lib.prof_range("encode: quantise + pack"):
                for name, (image_hash, text_hash) in hashes.items():
                    self.pack_hash_code(image_hash, bufs[name][0], rows, flags)
                    self.pack_hash_code(text_hash, bufs[name][1], rows, flags)

        # Host-to-device copies go on their own HIP stream: with pinned loader batches (pin_memory=True) the copy of the next
        # group overlaps the forward of the current one (60 MB of fp32 pixels per 100 images is 2-3 ms of PCIe time against a
        # 4.3 ms forward).  The compute stream waits on one event per loader batch; record_stream keeps the caching allocator
        # from recycling a buffer the other stream still reads.
        compute = torch.cuda.current_stream(dev)
        copy_stream = compute if os.environ.get("XMH_NO_COPY_STREAM") else self.__dict__.setdefault("_h2d_stream", torch.cuda.Stream(device=dev))

        def upload(t):
            if t is None or t.device == dev:
                return t
            t = t.to(dev, non_blocking=True)
            t.record_stream(compute)
            return t

        with torch.no_grad():
            for image, text, key_padding_mask, label, index in data_loader:
                as_list = isinstance(image, (list, tuple))
                if pending and (as_list != isinstance(pending[0][0], list) or
                                (not as_list and tuple(image.shape[1:]) != tuple(pending[0][0].shape[1:]))):
                    flush()                                  # stacked batches of another size / kind start a new group
                with torch.cuda.stream(copy_stream):
                    image = [upload(im) for im in image] if as_list else upload(image)
                    text, key_padding_mask, index = upload(text), upload(key_padding_mask), upload(index)
                    ready = torch.cuda.Event()
                    ready.record(copy_stream)
                compute.wait_event(ready)                    # stream-side wait: the host does not block
                pending.append((image, text, key_padding_mask, (index - lo).to(torch.int64)))
                if len(pending) == fuse:
                    flush()
            flush()
        if self.distributed:                                 # ternary-ness is a property of the whole code set: every rank must
            sharded.reduce_flags(flags)                      # rank its shard in the same bucket units (2K+1 vs K+1 buckets)
        if not (int(flags.item()) & 1):                      # no exact zero anywhere: drop the zero planes
            for img, txt in bufs.values():
                img.zero = txt.zero = None
        return bufs

    def encode_shard(self, data_loader, length: int):
        """this rank's packed (image, text) code rows of the single-stream methods."""
        return self.encode_streams(data_loader, length)[""]

    def _gather_packed(self, p: R.PackedCodes, length: int) -> R.PackedCodes:
        if not self.distributed:
            return p
        b = sharded.shard_bounds(length, self.world_size)
        counts = [b[r + 1] - b[r] for r in range(self.world_size)]
        has_zero = torch.tensor([0 if p.zero is None else 1], device=p.bits.device)
        dist.all_reduce(has_zero, op=dist.ReduceOp.MAX)
        bits = sharded.all_gather_rows(p.bits, counts)
        zero = None
        if int(has_zero.item()):
            zero = sharded.all_gather_rows(R.zero_plane_or_default(p), counts)
        return R.PackedCodes(bits, zero, p.K, p.flags)

    def get_code(self, data_loader, length: int):
        """reference :242-266 -- two [length, K] fp32 buffers of -1/0/+1, complete on every rank."""
        img, txt = self.encode_shard(data_loader, length)
        return self._gather_packed(img, length).unpack(), self._gather_packed(txt, length).unpack()

    # ---- retrieve ----------------------------------------------------------------------------------------
    def _map(self, q: R.PackedCodes, r_shard: R.PackedCodes, k):
        """one calc_map_k: full query set against this rank's gallery shard (all of it when not distributed)."""
        C = self.query_labels.shape[1]
        dev = q.bits.device
        if self._qlab is None:
            self._qlab = R.pack_labels(self.query_labels.to(dev))
            lo, hi = self._shard(self.retrieval_num)
            self._rlab = R.pack_labels(self.retrieval_labels[lo:hi].to(dev))
        if not self.distributed:
            if self.calc_map_k is calc_map_k:
                return float(R.map_k_packed(q, r_shard, self._qlab, self._rlab, C, k).item())
            return float(self.calc_map_k(q.unpack(), r_shard.unpack(), self.query_labels, self.retrieval_labels, k))
        if self.calc_map_k is not calc_map_k:
            # an injected calc_map_k sees what the reference's would: the complete code matrices on every rank
            # (runners/base.py:259-264 gathers them), at the price of the gather the built-in sharded scan avoids
            lo, hi = self._shard(self.retrieval_num)
            full = self._gather_packed(r_shard, self.retrieval_num)
            return float(self.calc_map_k(q.unpack(), full.unpack(), self.query_labels, self.retrieval_labels, k))
        ops = sharded.HipShardOps(q, self._qlab, r_shard, self._rlab, C)
        return float(sharded.map_k_sharded(ops, k, map_only=True)[0].item())

    def _evaluate(self, k):
        self._qlab = self._rlab = None
        q_img, q_txt = self.encode_shard(self.query_loader, self.query_num)
        r_img, r_txt = self.encode_shard(self.retrieval_loader, self.retrieval_num)
        q_img, q_txt = self._gather_packed(q_img, self.query_num), self._gather_packed(q_txt, self.query_num)
        maps = (self._map(q_img, r_txt, k), self._map(q_txt, r_img, k), self._map(q_img, r_img, k), self._map(q_txt, r_txt, k))
        return maps, (q_img, q_txt, r_img, r_txt)

    def retrieve_topk(self, k: int, tasks=("i2t", "t2i")):
        """north_star retrieval mode on the runner: encode both sets like valid() (get_code x 2, reference :309-310), then the exact
        k nearest gallery items of every query under (distance, gallery index) order instead of the mAP -- per rank on its gallery shard,
        the lists merged on the host (sharded.topk_sharded).  Returns {task: (dist float32 [Q, k], idx int32 [Q, k])} CPU tensors,
        identical on every rank: ``dist`` are the reference's calc_hammingDist values 0.5 * (K - q.r) (common/calc_utils.py:51-56;
        halves appear when sign_() left an exact 0 in a code, :407-410), unused slots (fewer than k gallery rows) hold inf / -1.
        Tasks: i2t / t2i / i2i / t2t as in valid()."""
        self._qlab = self._rlab = None
        q_img, q_txt = self.encode_shard(self.query_loader, self.query_num)
        r_img, r_txt = self.encode_shard(self.retrieval_loader, self.retrieval_num)
        q_img, q_txt = self._gather_packed(q_img, self.query_num), self._gather_packed(q_txt, self.query_num)
        pairs = {"i2t": (q_img, r_txt), "t2i": (q_txt, r_img), "i2i": (q_img, r_img), "t2t": (q_txt, r_txt)}
        lo, _ = self._shard(self.retrieval_num)
        # one unit for every rank and task: encode_streams reduced the quantiser's value flags over the ranks already
        tern = any(p.zero is not None for p in (q_img, q_txt, r_img, r_txt))
        out = {}
        for task in tasks:
            q, r = pairs[task]
            if self.distributed:
                d, i = sharded.topk_sharded(q, r, int(k), lo, ternary=tern)
            else:
                d, i = R.hamming_topk(q, r, int(k), 0, ternary=tern)
                d, i = d.cpu().to(torch.int32) & 0xFFFF, i.cpu()
            unused = i < 0
            d = d.to(torch.float32) * (0.5 if tern else 1.0)
            d[unused] = float("inf")
            out[task] = (d, i)
        return out

    def _is_writer(self):
        return not self.distributed or self.rank == 0

    def _gather_packed_to_writer(self, p: R.PackedCodes, length: int):
        """the complete packed code set on rank 0 only (None elsewhere): the .mat writer is the only consumer"""
        if not self.distributed:
            return p
        b = sharded.shard_bounds(length, self.world_size)
        counts = [b[r + 1] - b[r] for r in range(self.world_size)]
        bits = sharded.gather_rows_to(p.bits, counts, dst=0)
        zero = sharded.gather_rows_to(p.zero, counts, dst=0) if p.zero is not None else None     # zero planes are global (reduce_flags)
        return R.PackedCodes(bits, zero, p.K, p.flags) if bits is not None else None

    def _save_codes(self, codes, save_file):
        """one .mat of the reference's layout (:386-405).  valid() may write the same codes up to three times (i2t-best, t2i-best,
        last): the gather to the writer and the unpack to fp32 [N, K] are done once per code set, the matrices go to the file writer as
        device tensors (xmh/utils/matfile.py transposes them to the format's column-major order on the GPU), the label matrices -- the
        same every epoch -- are kept in file form, and the second and third file are further names of the first (hard links; the writer
        never rewrites a file in place)."""
        from ..utils import matfile
        memo = self.__dict__.get("_mat_memo")
        if memo is None or memo[0] is not codes:
            q_img, q_txt, r_img, r_txt = codes
            r_img, r_txt = self._gather_packed_to_writer(r_img, self.retrieval_num), self._gather_packed_to_writer(r_txt, self.retrieval_num)
            arrays = None
            if self._is_writer():
                arrays = tuple(t.unpack() for t in (q_img, q_txt, r_img, r_txt))
            memo = self._mat_memo = [codes, arrays, None]
        if self._is_writer():
            plain = type(self).save_mat.__func__ is BaseTrainer.save_mat.__func__
            if memo[2] is not None and os.path.exists(memo[2]) and plain:
                matfile.link_or_copy(memo[2], save_file)
                return
            a = memo[1]
            ql, rl = self.query_labels, self.retrieval_labels
            if plain:                                        # the label matrices in file form, made once (keyed on the objects: a new dataset is a new key)
                lab = self.__dict__.get("_mat_labels")
                if lab is None or lab[0] is not self.query_labels or lab[1] is not self.retrieval_labels:
                    try:
                        dev = a[0].device
                        on = [t.to(dev) if isinstance(t, torch.Tensor) else t for t in (self.query_labels, self.retrieval_labels)]
                        lab = self._mat_labels = (self.query_labels, self.retrieval_labels, matfile.prepare(on[0]), matfile.prepare(on[1]))
                    except matfile.UnsupportedMatValue:
                        lab = self._mat_labels = (self.query_labels, self.retrieval_labels, self.query_labels, self.retrieval_labels)
                ql, rl = lab[2], lab[3]
            self.save_mat(a[0], a[1], ql, a[2], a[3], rl, save_file=save_file)
            memo[2] = save_file

    def valid(self, epoch, k=None):
        assert self.query_loader is not None and self.retrieval_loader is not None
        save_dir = os.path.join(self.save_dir, "mat_files")
        os.makedirs(save_dir, exist_ok=True)
        self.logger.info("Valid.")
        self._start_model_bytes()                            # the .pth a new best would write is serialised under the encode loop
        try:
            (mAPi2t, mAPt2i, mAPi2i, mAPt2t), codes = self._evaluate(k)
        except BaseException:
            self._drop_model_bytes()                         # never leave the serialising thread behind an exception
            raise
        saved = False                                        # both bests in one epoch name the same model-<epoch>.pth: written once
        if self.max_mapi2t < mAPi2t:
            self.best_epoch_i = epoch
            self._save_codes(codes, os.path.join(save_dir, "i2t-best.mat"))
            if self._is_writer():
                self.save_model(save_dir=self.save_dir, epoch=epoch)
                saved = True
        self.max_mapi2t = max(self.max_mapi2t, mAPi2t)
        if self.max_mapt2i < mAPt2i:
            self.best_epoch_t = epoch
            self._save_codes(codes, os.path.join(save_dir, "t2i-best.mat"))
            if self._is_writer() and not saved:
                self.save_model(save_dir=self.save_dir, epoch=epoch)
        self.max_mapt2i = max(self.max_mapt2i, mAPt2i)
        self._save_codes(codes, os.path.join(save_dir, "last.mat"))
        self._mat_memo = None
        self._drop_model_bytes()
        self.logger.info(f">>>>>> [{epoch}/{self.epochs}], MAP(i->t): {mAPi2t}, MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, MAP(i->i): {mAPi2i}, "
                         f"MAX MAP(i->t): {self.max_mapi2t}, epoch: {self.best_epoch_i}, MAX MAP(t->i): {self.max_mapt2i}, epoch: {self.best_epoch_t}")
        return mAPi2t, mAPt2i, mAPi2i, mAPt2t

    def test(self):
        assert not self.model_state == "", "test step must provide the model file!"
        self.logger.info("Test.")
        save_dir = os.path.join(self.save_dir, "mat_files")
        os.makedirs(save_dir, exist_ok=True)
        (mAPi2t, mAPt2i, mAPi2i, mAPt2t), codes = self._evaluate(self.top_k)
        self._save_codes(codes, os.path.join(save_dir, "test.mat"))
        self._mat_memo = None
        self.logger.info(f">>>>>> TEST, MAP(i->t): {mAPi2t}, MAP(t->i): {mAPt2i}, MAP(t->t): {mAPt2t}, MAP(i->i): {mAPi2i}")
        return mAPi2t, mAPt2i, mAPi2i, mAPt2t

    # ---- outputs -----------------------------------------------------------------------------------------
    def save_model(self, save_dir, epoch, other=""):
        path = os.path.join(save_dir, "model-" + other + str(epoch) + ".pth")
        data = self._take_model_bytes()
        if data is not None:                                 # torch.save(self.model.state_dict(), <memory>) done under the encode loop of this valid()
            with open(path, "wb") as f:
                f.write(data.getbuffer())
        else:
            torch.save(self.model.state_dict(), path)
        self.logger.info("save mode to {}".format(path))

    # valid() saves the model when an epoch is a new best (reference :318-330): 0.6 GB of ViT-B/32 weights through pickle, 0.17 s at the
    # end of an evaluation whose GPU work the host only enqueues.  The weights do not change inside valid(), so the same torch.save runs on
    # a host thread while the encode loop keeps the GPU busy (device-to-host copies on a stream of its own) and save_model() only writes
    # the bytes.  Not a new best: the bytes are dropped.
    def _start_model_bytes(self):
        self._model_bytes = None
        if not self._is_writer() or type(self).save_model is not BaseTrainer.save_model:
            return
        box = {}
        model = self.model
        dev = next(model.parameters()).device

        side = None
        if dev.type == "cuda":
            # valid() runs straight behind train_epoch(): the copies must not start before the optimizer step / load_state_dict kernels the
            # caller's stream still holds -- the side stream is ordered behind it here, on the caller's thread
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))

        def work():
            try:
                buf = io.BytesIO()
                if side is not None:
                    with torch.cuda.device(dev), torch.cuda.stream(side):
                        torch.save(model.state_dict(), buf)
                else:
                    torch.save(model.state_dict(), buf)
                box["bytes"] = buf
            except Exception as e:                           # save_model() then serialises in line, and any real error shows there
                box["error"] = e
        th = threading.Thread(target=work, name="xmh-model-bytes", daemon=True)
        th.start()
        self._model_bytes = (th, box)

    def _take_model_bytes(self):
        pending = self.__dict__.get("_model_bytes")
        if pending is None:
            return None
        pending[0].join()
        return pending[1].get("bytes")                       # kept: a second best of the same epoch names the same file and is skipped by valid()

    def _drop_model_bytes(self):
        pending = self.__dict__.get("_model_bytes")
        if pending is not None:
            pending[0].join()
        self._model_bytes = None

    @classmethod
    def save_mat(cls, query_img, query_txt, query_labels, retrieval_img, retrieval_txt, retrieval_labels, save_file="i2t"):
        """reference :386-405 -- same keys; codes fp32 [N,K], labels as stored by the dataset (int64).  Level-5 MAT-file as
        scipy.io.savemat writes it, through xmh/utils/matfile.py (tensors are put in the file's column-major order on their own
        device); whatever that writer does not cover goes through scipy itself."""
        from ..utils import matfile
        names = ("q_img", "q_txt", "r_img", "r_txt", "q_l", "r_l")
        values = (query_img, query_txt, retrieval_img, retrieval_txt, query_labels, retrieval_labels)
        try:
            matfile.write_mat5(os.path.join(save_file), dict(zip(names, values)))
            return
        except matfile.UnsupportedMatValue:
            pass
        import scipy.io as scio

        def arr(t):
            if isinstance(t, matfile.Prepared):
                return np.ascontiguousarray(t.host.T)
            return t.cpu().detach().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        # scipy rewrites its target in place; valid() hard-links last.mat / *-best.mat to one file, so the fallback also writes
        # a new inode and renames it over the name, like write_mat5
        path = os.path.join(save_file)
        tmp = "%s.tmp%d" % (path, os.getpid())
        try:
            with open(tmp, "wb") as f:
                scio.savemat(f, dict(zip(names, (arr(v) for v in values))))
            os.replace(tmp, path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    @classmethod
    def from_config(cls, cfg, logger=None):
        raise NotImplementedError()
