"""DSPH model wrapper (reference models/DSPH/DSPH.py:13-60): backbone + Linear/tanh head, registered as "DSPH".
The reference reads ``loss/codetable.xlsx`` at construction for its HyP loss threshold (:33-35); that is a
training-only input and is not needed here.  What IS part of the contract is the parameter the loss module
owns: a reference DSPH checkpoint carries ``hyp.proxies`` [numclass, K] (models/DSPH/loss/HyP.py:15-16) beside ``backbone.*``
and ``hash.*`` -- pinned by tests/golden/runner.npz (DSPH_state_keys, from the reference's own class) -- so the module tree
here has it too and ``load_state_dict`` of such a checkpoint succeeds strictly."""
import torch
import torch.nn as nn

from ..common.register import registry
from .base import BaseModel
from .heads import DSPHHashLayer


class HyPProxies(nn.Module):
    """parameter container of the reference's HyP loss (models/DSPH/loss/HyP.py:8-16): ``proxies`` [numclass, output_dim],
    randn then kaiming_normal_(fan_out).  The reference seeds the GLOBAL generator with ``hypseed`` to draw them; a private
    generator is used here so that constructing a model has no side effect on the caller's random stream (values of a freshly
    constructed model therefore differ from the reference's; a loaded checkpoint overwrites them either way).  The loss itself
    is training code (out of scope)."""

    def __init__(self, numclass=80, output_dim=16, hypseed=0, alpha=0.8, threshold=None):
        super().__init__()
        self.alpha, self.threshold = alpha, threshold
        g = torch.Generator().manual_seed(int(hypseed))
        std = (2.0 / max(1, numclass)) ** 0.5                          # kaiming_normal_, mode="fan_out" of a [numclass, output_dim] matrix
        self.proxies = nn.Parameter(torch.randn(numclass, output_dim, generator=g) * std)


@registry.register_model("DSPH")
class DSPH(BaseModel):
    def __init__(self, cfg, outputDim=16, clipPath="./ViT-B-32.pt", train_num=10000, numclass=80, hypseed=1, alpha=0):
        super().__init__(cfg)
        embed_dim, self.backbone = self.load_backbone(clipPath=clipPath, return_patches=False)
        self.hash = DSPHHashLayer(inputDim=embed_dim, outputDim=outputDim)
        self.output_dim, self.numclass, self.hypseed, self.alpha = outputDim, numclass, hypseed, alpha
        self.hyp = HyPProxies(numclass=numclass, output_dim=outputDim, hypseed=hypseed, alpha=alpha)

    def encode_image(self, image):
        return self.hash.encode_img(self.backbone.encode_image(image))

    def encode_text(self, text):
        return self.hash.encode_txt(self.backbone.encode_text(text))

    def object_function(self, *a, **k):
        raise NotImplementedError("training losses are outside the encode-and-retrieve path (SURVEY 2.1 #9)")

    @classmethod
    def from_config(cls, cfg, output_dim=16, train_num=10000):
        return cls(cfg=cfg, outputDim=output_dim, clipPath=cfg.get("clip_path", "./ViT-B-32.pt"), train_num=train_num,
                   numclass=cfg.get("numclass", 80), hypseed=cfg.get("hypseed", 0), alpha=cfg.get("alpha", 0.8))
