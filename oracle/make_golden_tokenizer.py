"""Generate tests/golden/tokenizer.npz by RUNNING the reference's SimpleTokenizer (models/CLIP/simple_tokenizer.py) and the
caption packing of its dataset (dataset/transformer_dataset.py:67-87) on a fixed list of captions.
``ftfy`` is not installed in the build container; the reference module is imported with an identity ``ftfy.fix_text``
(what ftfy does to text that needs no repair -- the captions below are clean apart from HTML entities).
TEST INFRASTRUCTURE ONLY; runs in the build container.   python oracle/make_golden_tokenizer.py"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import _ref_import  # noqa: E402

CAPTIONS = [
    "A photo of a cat, sitting on the sofa!",
    "two dogs playing in the park",
    "It's a man's world -- they've said; I'm sure she'll agree, he'd too.",
    "sunset beach 2019 hdr nikon d7000",
    "An   oddly\tspaced \n caption   with   tabs",
    "café au lait &amp; crème brûlée",
    "emoji \U0001f600 and symbols #hashtag @user 100% $5.99",
    "UPPER lower MiXeD case",
    "supercalifragilisticexpialidocious antidisestablishmentarianism",
    "a " + "very " * 40 + "long caption that must be cut to the maximum number of words",
    "",
    "&lt;b&gt;bold&lt;/b&gt; &quot;quoted&quot;",
]
MAX_WORDS = 32


def main():
    _ref_import.setup()
    if "ftfy" not in sys.modules:
        f = types.ModuleType("ftfy")
        f.fix_text = lambda s: s
        sys.modules["ftfy"] = f
    from models.CLIP.simple_tokenizer import SimpleTokenizer
    tok = SimpleTokenizer()
    ids, packed = [], []
    for cap in CAPTIONS:
        words = tok.tokenize(cap)
        ids.append(np.asarray(tok.convert_tokens_to_ids(words), np.int64))
        w = ["<|startoftext|>"] + words                       # dataset/transformer_dataset.py:72-84
        if len(w) > MAX_WORDS - 1:
            w = w[:MAX_WORDS - 1]
        w = w + ["<|endoftext|>"]
        c = tok.convert_tokens_to_ids(w)
        while len(c) < MAX_WORDS:
            c.append(0)
        packed.append(c)
    out = {"captions": np.asarray(CAPTIONS, dtype=object), "packed": np.asarray(packed, np.int64), "max_words": MAX_WORDS}
    for i, a in enumerate(ids):
        out["ids%d" % i] = a
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tokenizer.npz"), **out)
    print("wrote tokenizer.npz", len(CAPTIONS), "captions; decode check:", tok.decode(ids[0].tolist()))


if __name__ == "__main__":
    main()
