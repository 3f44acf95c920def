"""DSPH model wrapper (reference models/DSPH/DSPH.py:13-60): backbone + Linear/tanh head, registered as "DSPH".
The reference reads ``loss/codetable.xlsx`` at construction for its HyP loss threshold (:33-35); that is a
training-only input and is not needed here."""
from ..common.register import registry
from .base import BaseModel
from .heads import DSPHHashLayer


@registry.register_model("DSPH")
class DSPH(BaseModel):
    def __init__(self, cfg, outputDim=16, clipPath="./ViT-B-32.pt", train_num=10000, numclass=80, hypseed=1, alpha=0):
        super().__init__(cfg)
        embed_dim, self.backbone = self.load_backbone(clipPath=clipPath, return_patches=False)
        self.hash = DSPHHashLayer(inputDim=embed_dim, outputDim=outputDim)
        self.output_dim, self.numclass, self.hypseed, self.alpha = outputDim, numclass, hypseed, alpha

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
