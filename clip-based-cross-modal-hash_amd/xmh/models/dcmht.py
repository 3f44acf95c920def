"""DCMHT model wrapper (reference models/DCMHT/DCMHT.py:11-70): backbone + DCMHT head, registered as "DCMHT"; the loss FORWARD
(:72-155: similarity_loss / soft_argmax_hash_loss / our_loss / object_function) through xmh_loss.hip -- no autograd graph: the
backward pass of the training step is outside this path."""
import torch

from .. import retrieval as R
from .._lib import check, current_stream, lib, ptr
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

    # ---- loss forward (reference :72-155) ------------------------------------------------------------------------------------
    @staticmethod
    def _codes(x):
        if not x.is_cuda:
            raise RuntimeError("xmh losses need CUDA/HIP tensors (got %s); there is no CPU fallback" % x.device)
        return x.detach().float().reshape(x.shape[0], -1).contiguous()

    def similarity_loss(self, a, b, labels):
        """(positive_loss, negative_loss) of reference similarity_loss(a, b, calc_label_sim(labels, labels)) (:72-98): 0-dim fp32
        device tensors.  `labels` is the [B, C] multi-hot matrix itself (packed on the device), not the [B, B] label_sim."""
        a, b = self._codes(a), self._codes(b)
        B, D = a.shape
        if b.shape != a.shape or labels.shape[0] != B:
            raise ValueError("similarity_loss: a %s, b %s, labels %s" % (tuple(a.shape), tuple(b.shape), tuple(labels.shape)))
        lab = R.pack_labels(labels.to(a.device))
        out = torch.empty(2, dtype=torch.float64, device=a.device)
        cosine = self.similarity_function == "cosine"
        if not cosine and self.similarity_function != "euclidean":
            raise ValueError("similarity_function must be 'euclidean' or 'cosine', got %r" % (self.similarity_function,))
        max_value = float(self.output_dim * 2 * self.vartheta) ** 0.5
        check(lib.xmh_pair_similarity_loss(ptr(a), ptr(b), B, D, ptr(lab), labels.shape[1], int(cosine), max_value, float(self.threshold),
                                           ptr(out), current_stream()), "xmh_pair_similarity_loss")
        out = out.float()
        return out[0], out[1]

    def soft_argmax_hash_loss(self, code):
        """reference :100-105 -- 1 - mean((2 code - 1)^2)"""
        c = self._codes(code)
        out = torch.empty(1, dtype=torch.float64, device=c.device)
        check(lib.xmh_quant_loss(ptr(c), c.numel(), ptr(out), current_stream()), "xmh_quant_loss")
        return out.float()[0]

    def our_loss(self, image, text, labels=None, indexs=None, **kwags):
        """reference :107-149 -- (loss, loss_dict) with the same keys"""
        intra_p, intra_n = self.similarity_loss(image, text, labels)
        inter_p_i, inter_n_i = self.similarity_loss(image, image, labels)
        inter_p_t, inter_n_t = self.similarity_loss(text, text, labels)
        quan_i, quan_t = self.soft_argmax_hash_loss(image), self.soft_argmax_hash_loss(text)
        intra = intra_p + intra_n
        inter = inter_p_t + inter_p_i + inter_n_i + inter_n_t
        loss = inter + intra + self.quan_alpha * ((quan_i + quan_t) / 2)
        loss_dict = {"All loss": loss, "Intra": {"Positive": intra_p, "Negative": intra_n},
                     "Inter": {"Positive": {"i2t": inter_p_i, "t2i": inter_p_t}, "Negative": {"i2t": inter_n_i, "t2i": inter_n_t}},
                     "Quan": {"Image": quan_i, "Text": quan_t}}
        return loss, loss_dict

    def object_function(self, img_hash, txt_hash, labels=None, indexs=None, **kwags):
        """reference :151-155 -- without labels every sample is its own class"""
        if labels is None:
            labels = torch.ones([img_hash.shape[0]], dtype=torch.int).diag()
        return self.our_loss(img_hash, txt_hash, labels, indexs, **kwags)

    @classmethod
    def from_config(cls, cfg, output_dim=16, train_num=10000):
        return cls(cfg=cfg, outputDim=output_dim, clipPath=cfg.get("clip_path", "./ViT-B-32.pt"), train_num=train_num,
                   hash_func=cfg.get("hash_func", "softmax"), vartheta=cfg.get("vartheta", 0.75), threshold=cfg.get("threshold", 0.1),
                   quan_alpha=cfg.get("quan_alpha", 0.001), similarity_function=cfg.get("similarity_function", "euclidean"))
