"""TwDH model wrapper (reference models/TwDH/TwDH.py:10-125): CLIP backbone + the DCMHT hash layer at ``long_dim`` bits, and
for every configured short length a fixed ``[2*long, 2*short]`` transform followed by the pair softmax
(``quantization(long_hash.matmul(trans))``, :66-85).  Registered as "TwDH".

The reference torch.load()s the centre / transform tensors from ``<long_center>/<long>.pkl``,
``<short_center>/<short>.pkl`` and ``<trans_matrix>/<long>/<short>.pkl``; the same layout is read here.  Without such files
(benchmarks, tests) ``trans_matrix: synthetic`` + ``short_dims: [..]`` builds seeded transforms instead.  The centres only
enter the training loss and are not loaded."""
import os

import torch

from .. import ops
from ..common.register import registry
from .base import BaseModel
from .heads import DCMHTHashLayer
from .weights import _gen


@registry.register_model("TwDH")
class TwDH(BaseModel):
    def __init__(self, cfg, long_dim=512, short_dim=16, clipPath="./ViT-B-32.pt", train_num=10000, hash_func="softmax",
                 trans="./data/transformer/TwDH/center/trans", short_dims=None, quan_alpha=0.5, low_rate=0):
        super().__init__(cfg)
        embed_dim, self.backbone = self.load_backbone(clipPath=clipPath, return_patches=False)
        self.hash = DCMHTHashLayer(feature_size=embed_dim, outputDim=long_dim, num_heads=8, batch_first=True, hash_func_=hash_func)
        self.output_dim = self.long_dim = long_dim
        self.quan_alpha, self.low_rate = quan_alpha, low_rate
        self.trans = {}                                         # key (short length as str) -> [2*long, 2*short] fp32
        if trans == "synthetic":
            for sd in (short_dims or [short_dim]):
                g = _gen(1814, "twdh_trans/%d/%d" % (long_dim, sd))
                self.trans[str(sd)] = torch.randn(2 * long_dim, 2 * sd, generator=g) * (2 * long_dim) ** -0.5
        elif os.path.isfile(trans):
            self.trans[os.path.basename(trans).strip().split(".")[0]] = torch.load(trans, map_location="cpu").float()
        else:
            for item in sorted(os.listdir(trans)):
                self.trans[item.strip().split(".")[0]] = torch.load(os.path.join(trans, item), map_location="cpu").float()
        self.short_dims = [int(k) for k in self.trans]
        self._trans_t = {}                                      # device copies, transposed to the nn.Linear layout gemm_nt takes

    def get_short_dims(self):
        return self.short_dims

    def _short(self, long_hash):
        out = {}
        for k, v in self.trans.items():
            w = self._trans_t.get(k)
            if w is None or w.device != long_hash.device:
                w = v.to(long_hash.device).t().contiguous()
                self._trans_t[k] = w
            out[k] = ops.pair_softmax(ops.gemm_nt(long_hash, w))  # quantization(long_hash.matmul(v)), TwDH.py:73,83
        return out

    def encode_image(self, image):
        long_hash = self.hash.encode_img(self.backbone.encode_image(image))
        return long_hash, self._short(long_hash)

    def encode_text(self, text):
        long_hash = self.hash.encode_txt(self.backbone.encode_text(text))
        return long_hash, self._short(long_hash)

    def object_function(self, *a, **k):
        raise NotImplementedError("training losses are outside the encode-and-retrieve path (SURVEY 2.1 #7)")

    @classmethod
    def from_config(cls, cfg, output_dim=16, train_num=10000):
        long_dim = cfg.get("long_dim", 512)
        trans = cfg.get("trans_matrix", "./data/transformer/TwDH/center/trans")
        if trans != "synthetic":
            trans = os.path.join(trans, str(long_dim))
        return cls(cfg=cfg, long_dim=long_dim, short_dim=output_dim, clipPath=cfg.get("clip_path", "./ViT-B-32.pt"), train_num=train_num,
                   hash_func=cfg.get("hash_func", "softmax"), trans=trans, short_dims=cfg.get("short_dims", None),
                   quan_alpha=cfg.get("quan_alpha", 0.5), low_rate=cfg.get("low_rate", 0))
