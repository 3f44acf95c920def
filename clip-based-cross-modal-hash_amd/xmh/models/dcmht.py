"""DCMHT model wrapper (reference models/DCMHT/DCMHT.py:11-70): backbone + DCMHT head, registered as "DCMHT"."""
from ..common.register import registry
from .base import BaseModel
from .heads import DCMHTHashLayer


@registry.register_model("DCMHT")
class DCMHT(BaseModel):
    def __init__(self, cfg, outputDim=16, clipPath="./ViT-B-32.pt", train_num=10000, hash_func="softmax", vartheta=0.75,
                 threshold=0.1, quan_alpha=0.001, similarity_function="euclidean"):
        super().__init__(cfg)
        embed_dim, self.backbone = self.load_backbone(clipPath=clipPath, return_patches=False)
        self.hash = DCMHTHashLayer(feature_size=embed_dim, outputDim=outputDim, num_heads=8, batch_first=True, hash_func_=hash_func)
        self.output_dim, self.hash_func = outputDim, hash_func
        self.vartheta, self.threshold, self.quan_alpha = vartheta, threshold, quan_alpha
        self.similarity_function = similarity_function

    def encode_image(self, image):
        return self.hash.encode_img(self.backbone.encode_image(image))

    def encode_text(self, text):
        return self.hash.encode_txt(self.backbone.encode_text(text))

    def object_function(self, *a, **k):
        raise NotImplementedError("training losses are outside the encode-and-retrieve path (SURVEY 2.1 #7)")

    @classmethod
    def from_config(cls, cfg, output_dim=16, train_num=10000):
        return cls(cfg=cfg, outputDim=output_dim, clipPath=cfg.get("clip_path", "./ViT-B-32.pt"), train_num=train_num,
                   hash_func=cfg.get("hash_func", "softmax"), vartheta=cfg.get("vartheta", 0.75), threshold=cfg.get("threshold", 0.1),
                   quan_alpha=cfg.get("quan_alpha", 0.001), similarity_function=cfg.get("similarity_function", "euclidean"))
