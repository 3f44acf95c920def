"""CPU: the host side of the evaluation input pipeline -- the CLIP BPE tokenizer against token ids produced by
the reference's SimpleTokenizer (oracle/make_golden_tokenizer.py), and the .mat-driven dataset mirror on a tiny
generated dataset.  The BPE merge table is a data file of the reference that this repository does not ship: the tokenizer
tests run where it can be found (the build container) and skip elsewhere."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

VOCAB_CANDIDATES = [os.environ.get("XMH_BPE_VOCAB"), "/root/reference/models/CLIP/bpe_simple_vocab_16e6.txt.gz"]


@pytest.fixture(scope="module")
def tok():
    from xmh.dataset.tokenizer import ClipTokenizer
    for c in VOCAB_CANDIDATES:
        if c and os.path.isfile(c):
            return ClipTokenizer(c)
    pytest.skip("bpe_simple_vocab_16e6.txt.gz not available here")


def test_tokenizer_is_registered_like_the_reference(monkeypatch, tmp_path):
    import xmh.dataset  # noqa: F401
    from xmh.common.register import registry
    from xmh.dataset.tokenizer import ClipTokenizer, find_vocab
    assert registry.get_tokenizer_class("clip_tokenizer") is ClipTokenizer
    monkeypatch.delenv("XMH_BPE_VOCAB", raising=False)
    monkeypatch.chdir(tmp_path)                                # no ./models/CLIP/ here, nothing shipped next to the module
    with pytest.raises(FileNotFoundError, match="XMH_BPE_VOCAB"):
        find_vocab(None)


def test_tokenizer_matches_reference_ids(tok):
    g = np.load(os.path.join(GOLDEN, "tokenizer.npz"), allow_pickle=True)
    caps = [str(c) for c in g["captions"]]
    assert len(tok.encoder) == 49408 and tok.encoder["<|startoftext|>"] == 49406 and tok.encoder["<|endoftext|>"] == 49407
    for i, cap in enumerate(caps):
        assert tok.encode(cap) == g["ids%d" % i].tolist(), cap
        assert tok.convert_tokens_to_ids(tok.tokenize(cap)) == g["ids%d" % i].tolist()
    assert tok.decode(tok.encode("two dogs playing in the park")) == "two dogs playing in the park "


def _tiny_dataset(tmp_path, n=14, C=5):
    import scipy.io as scio
    from PIL import Image
    rng = np.random.default_rng(3)
    root = tmp_path / "data" / "tiny"
    root.mkdir(parents=True)
    paths, caps = [], []
    for i in range(n):
        h, w = (40 + 3 * i, 64) if i % 2 else (48, 50 + 2 * i)
        p = root / ("img%02d.png" % i)
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), mode="RGB").save(p)
        paths.append(str(p))
        caps.append(["photo number %d of a dog" % i, "a second caption, #%d!" % i])
    labels = (rng.random((n, C)) < 0.4).astype(np.int64)
    labels[:, 0] = 1
    scio.savemat(root / "index.mat", {"index": np.asarray(paths)})                 # char matrix, one padded row per image
    scio.savemat(root / "caption.mat", {"caption": np.asarray(caps)})
    scio.savemat(root / "label.mat", {"category": labels})
    return root, paths, caps, labels


def test_dataset_mirror_on_generated_mat_files(tok, tmp_path):
    """dataset/builder.py:34-104 + dataset/transformer_dataset.py: key names, split sizes, sample tuple, caption packing."""
    from PIL import Image
    from xmh.dataset import build_dataloader
    root, paths, caps, labels = _tiny_dataset(tmp_path)
    np.random.seed(5)
    train, query, retrieval = build_dataloader(str(root / "caption.mat"), str(root / "index.mat"), str(root / "label.mat"), query_num=4, train_num=6,
                                               dataset_cls="transformer_dataset", tokenizer=tok, maxWords=32)
    assert train is None and len(query) == 4 and len(retrieval) == 10
    assert query.get_all_label().dtype == torch.int64 and tuple(retrieval.get_all_label().shape) == (10, 5)
    g = np.load(os.path.join(GOLDEN, "tokenizer.npz"), allow_pickle=True)
    image, caption, kpm, label, index = query[1]
    src = str(query.indexs[1]).strip()
    assert image.dtype == torch.uint8 and np.array_equal(image.numpy(), np.asarray(Image.open(src).convert("RGB")))
    assert caption.shape == (32,) and caption[0] == 49406 and int((caption == 49407).sum()) == 1 and torch.equal(kpm, caption == 0)
    assert index == 1 and torch.equal(label, torch.from_numpy(labels[paths.index(src)]))
    # caption packing equals the reference's for the golden captions (CLS + tokens cut to 31 + SEP + zero padding)
    query.captions = np.asarray([[str(c)] for c in g["captions"]])
    for i in range(len(g["captions"])):
        assert query._load_text(i)[0].tolist() == g["packed"][i].tolist()
    batch = query.collate([query[i] for i in range(4)])
    assert isinstance(batch[0], list) and batch[1].shape == (4, 32) and batch[4].tolist() == [0, 1, 2, 3]
