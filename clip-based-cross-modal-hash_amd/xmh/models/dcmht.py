"""DCMHT model wrapper (reference models/DCMHT/DCMHT.py:11-70): backbone + DCMHT head, registered as "DCMHT"; the loss
(:72-155: similarity_loss / soft_argmax_hash_loss / our_loss / object_function) through xmh_loss.hip.  `our_loss` /
`object_function` are differentiable with respect to the two code matrices (`_Objective`: the gradient kernels of xmh_loss.hip
behind torch.autograd, i.e. what `loss.backward()` of runners/DCMHT/runner.py:124 hands to the hash heads); the backward of
the encoders themselves is outside this path."""
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

    # ---- loss (reference :72-155) ---------------------------------------------------------------------------------------------
    @staticmethod
    def _codes(x):
        if not x.is_cuda:
            raise RuntimeError("xmh losses need CUDA/HIP tensors (got %s); there is no CPU fallback" % x.device)
        return x.detach().float().reshape(x.shape[0], -1).contiguous()

    def _branch(self):
        cosine = self.similarity_function == "cosine"
        if not cosine and self.similarity_function != "euclidean":
            raise ValueError("similarity_function must be 'euclidean' or 'cosine', got %r" % (self.similarity_function,))
        return int(cosine), float(self.output_dim * 2 * self.vartheta) ** 0.5, float(self.threshold)

    def _pair(self, a, b, lab, C):
        cosine, max_value, threshold = self._branch()
        out = torch.empty(2, dtype=torch.float64, device=a.device)
        check(lib.xmh_pair_similarity_loss(ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(lab), C, cosine, max_value, threshold,
                                           ptr(out), current_stream()), "xmh_pair_similarity_loss")
        out = out.float()
        return out[0].clone(), out[1].clone()                    # separate tensors: they become outputs of an autograd Function

    def _pair_grad(self, a, b, lab, C, scale, upstream, grad, accumulate):
        cosine, max_value, threshold = self._branch()
        check(lib.xmh_pair_similarity_loss_grad(ptr(a), ptr(b), a.shape[0], a.shape[1], ptr(lab), C, cosine, max_value, threshold,
                                                float(scale), ptr(upstream), ptr(grad), int(accumulate), current_stream()),
              "xmh_pair_similarity_loss_grad")

    def _quant(self, c):
        out = torch.empty(1, dtype=torch.float64, device=c.device)
        check(lib.xmh_quant_loss(ptr(c), c.numel(), ptr(out), current_stream()), "xmh_quant_loss")
        return out.float().reshape(())

    def similarity_loss(self, a, b, labels):
        """(positive_loss, negative_loss) of reference similarity_loss(a, b, calc_label_sim(labels, labels)) (:72-98): 0-dim fp32
        device tensors (no graph behind them: differentiate through our_loss).  `labels` is the [B, C] multi-hot matrix itself
        (packed on the device), not the [B, B] label_sim."""
        a, b = self._codes(a), self._codes(b)
        if b.shape != a.shape or labels.shape[0] != a.shape[0]:
            raise ValueError("similarity_loss: a %s, b %s, labels %s" % (tuple(a.shape), tuple(b.shape), tuple(labels.shape)))
        return self._pair(a, b, R.pack_labels(labels.to(a.device)), labels.shape[1])

    def soft_argmax_hash_loss(self, code):
        """reference :100-105 -- 1 - mean((2 code - 1)^2)"""
        return self._quant(self._codes(code))

    def our_loss(self, image, text, labels=None, indexs=None, **kwags):
        """reference :107-149 -- (loss, loss_dict) with the same keys; `loss` carries the autograd graph back to image / text,
        the entries of loss_dict are detached like the reference's `.data`"""
        if image.shape[0] != text.shape[0] or labels.shape[0] != image.shape[0]:
            raise ValueError("our_loss: image %s, text %s, labels %s" % (tuple(image.shape), tuple(text.shape), tuple(labels.shape)))
        (loss, intra_p, intra_n, inter_p_i, inter_n_i, inter_p_t, inter_n_t, quan_i, quan_t) = _Objective.apply(self, image, text, labels)
        loss_dict = {"All loss": loss.detach(), "Intra": {"Positive": intra_p, "Negative": intra_n},
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


class _Objective(torch.autograd.Function):
    """our_loss (reference :107-149) with its gradient: forward = the nine scalars from xmh_pair_similarity_loss / xmh_quant_loss,
    backward = xmh_pair_similarity_loss_grad / xmh_quant_loss_grad accumulated per code matrix.  d loss / d image =
    d intra(image, text) + 2 d inter(image, image) [both argument slots are the same tensor] + quan_alpha / 2 d quan(image);
    text alike with the roles exchanged (the pair terms are symmetric)."""

    @staticmethod
    def forward(ctx, model, image, text, labels):
        a, b = model._codes(image), model._codes(text)
        if a.shape != b.shape:
            raise ValueError("our_loss: image codes %s, text codes %s" % (tuple(a.shape), tuple(b.shape)))
        if (image.requires_grad or text.requires_grad) and (a.shape[0] + a.shape[1]) * 4 > 64 * 1024:
            # the gradient kernel keeps one code row and one row of pair weights in LDS: say so now, not at backward() time
            raise ValueError("our_loss: batch %d x code width %d exceeds the gradient kernel's (B + D) * 4 <= 65536 bytes of LDS"
                             % (a.shape[0], a.shape[1]))
        C = labels.shape[1]
        lab = R.pack_labels(labels.to(a.device))
        intra_p, intra_n = model._pair(a, b, lab, C)
        inter_p_i, inter_n_i = model._pair(a, a, lab, C)
        inter_p_t, inter_n_t = model._pair(b, b, lab, C)
        quan_i, quan_t = model._quant(a), model._quant(b)
        loss = (inter_p_t + inter_p_i + inter_n_i + inter_n_t) + (intra_p + intra_n) + model.quan_alpha * ((quan_i + quan_t) / 2)
        ctx.model, ctx.C = model, C
        ctx.shapes = (image.shape, image.dtype, text.shape, text.dtype)
        ctx.save_for_backward(a, b, lab)
        parts = (intra_p, intra_n, inter_p_i, inter_n_i, inter_p_t, inter_n_t, quan_i, quan_t)
        ctx.mark_non_differentiable(*parts)
        return (loss,) + parts

    @staticmethod
    @torch.autograd.function.once_differentiable      # the gradient kernels are not themselves differentiable: fail loudly on double backward
    def backward(ctx, g, *_):
        a, b, lab = ctx.saved_tensors
        m, C = ctx.model, ctx.C
        up = g.detach().float().reshape(1).contiguous()
        out = [None, None]
        for slot, (x, y) in enumerate(((a, b), (b, a))):
            if not ctx.needs_input_grad[1 + slot]:
                continue
            gx = torch.empty_like(x)
            m._pair_grad(x, y, lab, C, 1.0, up, gx, False)                     # intra term, this matrix in the first slot
            m._pair_grad(x, x, lab, C, 2.0, up, gx, True)                      # inter term: both slots
            check(lib.xmh_quant_loss_grad(ptr(x), x.numel(), float(m.quan_alpha) / 2, ptr(up), ptr(gx), 1, current_stream()),
                  "xmh_quant_loss_grad")
            shape, dtype = ctx.shapes[2 * slot], ctx.shapes[2 * slot + 1]
            out[slot] = gx.reshape(shape).to(dtype)
        return None, out[0], out[1], None
