from __future__ import annotations

import torch
from torch.utils.data import Dataset

from ..common.register import registry
from ..models.weights import _gen


@registry.register_dataset("synthetic_pairs")
class SyntheticPairs(Dataset):
    """Seeded image/caption/label triples of the reference's shapes: images randn [3,R,R], captions
    [SOS, tokens, EOS, 0...] padded to max_words with mask ids == 0, multi-hot labels with >= 1 class."""

    def __init__(self, n: int, num_classes: int = 24, resolution: int = 224, max_words: int = 32, seed: int = 1814,
                 p_label: float = 0.1, vocab: int = 49408, index_offset: int = 0, raw_hw=None):
        self.n, self.res, self.L, self.seed, self.vocab, self.offset = n, resolution, max_words, seed, vocab, index_offset
        self.raw_hw = None if raw_hw is None else (int(raw_hw[0]), int(raw_hw[1]))     # yield undecoded-photo-like uint8 [H, W, 3]
        g = _gen(seed, "labels/%d" % index_offset)
        L = torch.rand(n, num_classes, generator=g) < p_label
        L[torch.arange(n), torch.randint(0, num_classes, (n,), generator=g)] = True
        self.labels = L.to(torch.int64)

    def __len__(self):
        return self.n

    def get_all_label(self):
        return self.labels

    def __getitem__(self, i):
        g = _gen(self.seed, "item/%d" % (self.offset + i))
        if self.raw_hw is None:
            image = torch.randn(3, self.res, self.res, generator=g)
        else:                                            # what PIL would hand to the transform: RGB bytes, any size
            image = torch.randint(0, 256, (self.raw_hw[0], self.raw_hw[1], 3), generator=g, dtype=torch.uint8)
        ids = torch.zeros(self.L, dtype=torch.int64)
        n = int(torch.randint(4, self.L - 1, (1,), generator=g))
        ids[0] = self.vocab - 2
        ids[1:1 + n] = torch.randint(1, self.vocab - 3, (n,), generator=g)
        ids[1 + n] = self.vocab - 1
        return image, ids, ids == 0, self.labels[i], i


def build_synthetic_splits(cfg, train_num, query_num):
    """(train, query, retrieval) like dataset/builder.py:34-104 returns; retrieval_num from cfg (default 4x query)."""
    C = cfg.get("num_classes", 24)
    res, L, seed = cfg.get("image_resolution", 224), cfg.get("max_word", 32), cfg.get("seed", 1814)
    rnum = cfg.get("retrieval_num", 4 * query_num)
    mk = lambda n, off: SyntheticPairs(n, C, res, L, seed, cfg.get("p_label", 0.1), index_offset=off, raw_hw=cfg.get("raw_image_hw"))   # noqa: E731
    return mk(min(train_num, rnum), query_num), mk(query_num, 0), mk(rnum, query_num)
