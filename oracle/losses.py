"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the DCMHT loss (reference models/DCMHT/DCMHT.py:72-155) in float64 torch,
and its gradient with respect to the two code matrices by autograd over that restatement (what loss.backward() of
runners/DCMHT/runner.py:124 produces); pinned against the reference's own `our_loss` / its backward by
tests/golden/loss_dcmht.npz (oracle/make_golden_loss.py).  Only tests/ may import this module."""
import torch


def label_sim(labels):
    """common/calc_utils.py:8-10"""
    l = labels.double()
    return (l @ l.t() > 0).double()


def similarity_loss(a, b, lsim, output_dim, vartheta=0.75, threshold=0.1, similarity_function="euclidean"):
    """models/DCMHT/DCMHT.py:72-98 -> (positive_loss, negative_loss)"""
    a, b = a.double(), b.double()
    if similarity_function == "euclidean":
        s = torch.cdist(a, b, p=2.0, compute_mode="donot_use_mm_for_euclid_dist")        # :78 (the reference's cdist may take the mm route)
        pos = s * lsim                                                                    # :81
        neg = s * (1 - lsim)                                                              # :82
        m = float(output_dim * 2 * vartheta) ** 0.5                                       # :83
        neg = neg.clip(max=m)                                                             # :84
        neg = m * (1 - lsim) - neg                                                        # :85
        return pos.pow(2).mean(), neg.pow(2).mean()                                       # :87-88
    s = (a / a.norm(dim=-1, keepdim=True)) @ (b / b.norm(dim=-1, keepdim=True)).t()       # calc_utils.py:38-49
    s = s.clip(min=threshold).clip(max=1 - threshold)                                     # :93
    l = (-lsim * torch.log(s) - (1 - lsim) * torch.log(1 - s)).mean()                    # :94
    return l, l


def soft_argmax_hash_loss(code):
    """:100-105"""
    return 1 - (2 * code.double() - 1).pow(2).mean()


def our_loss(image, text, labels, output_dim, vartheta=0.75, threshold=0.1, quan_alpha=0.001, similarity_function="euclidean"):
    """:107-149 -> dict of the nine scalars"""
    ls = label_sim(labels)
    kw = dict(output_dim=output_dim, vartheta=vartheta, threshold=threshold, similarity_function=similarity_function)
    ip, in_ = similarity_loss(image, text, ls, **kw)
    pi, ni = similarity_loss(image, image, ls, **kw)
    pt, nt = similarity_loss(text, text, ls, **kw)
    qi, qt = soft_argmax_hash_loss(image), soft_argmax_hash_loss(text)
    loss = (pt + pi + ni + nt) + (ip + in_) + quan_alpha * (qi + qt) / 2
    return {"loss": loss, "intra_pos": ip, "intra_neg": in_, "inter_pos_i": pi, "inter_neg_i": ni, "inter_pos_t": pt, "inter_neg_t": nt,
            "quan_i": qi, "quan_t": qt}


def our_loss_grad(image, text, labels, output_dim, **kw):
    """(d loss / d image, d loss / d text) in float64: autograd over our_loss above (runners/DCMHT/runner.py:124 loss.backward())"""
    img = image.double().clone().requires_grad_(True)
    txt = text.double().clone().requires_grad_(True)
    our_loss(img, txt, labels, output_dim, **kw)["loss"].backward()
    return img.grad, txt.grad
