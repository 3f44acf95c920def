"""The reference's evaluation input pipeline behind its own names: ``build_dataloader`` (dataset/builder.py:34-104:
.mat / .npy / .txt parsing and the query / train / retrieval split) and ``transformer_dataset``
(dataset/transformer_dataset.py:11-107: the sample tuple ``(image, caption, key_padding_mask, label, index)``).

What differs, deliberately: ``_load_image`` only DECODES (PIL -> RGB bytes ``[H, W, 3] uint8``); the resize + ToTensor +
Normalize of the reference's eval transform runs on the GPU, Pillow-exact, in ``BaseTrainer.encode_streams`` through
``xmh.dataset.preprocess.GpuEvalTransform``.  Photos of different sizes cannot be stacked, so ``collate`` keeps the images
of a batch as a list.  The training transform (random crop / flip) belongs to training, which is out of scope."""
from __future__ import annotations

import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ..common.register import registry

SPECIAL_TOKEN = {"CLS_TOKEN": "<|startoftext|>", "SEP_TOKEN": "<|endoftext|>", "MASK_TOKEN": "[MASK]", "UNK_TOKEN": "[UNK]", "PAD_TOKEN": "[PAD]"}


@registry.register_dataset("transformer_dataset")
class TransformerDataset(Dataset):
    def __init__(self, captions, indexs, labels, is_train=True, imageResolution=224, tokenizer=None, maxWords=32, npy=False, **kwags):
        super().__init__()
        if is_train:
            raise NotImplementedError("the training transform (RandomResizedCrop / flip) is outside the encode-and-retrieve path")
        self.captions, self.indexs, self.labels = captions, indexs, labels
        self.is_train, self.npy = is_train, npy
        self.imageResolution, self.maxWords, self.tokenizer = imageResolution, maxWords, tokenizer
        self._length = len(self.indexs)

    def __len__(self):
        return self._length

    def _load_image(self, index: int) -> torch.Tensor:
        """RGB bytes [H, W, 3] uint8 (reference :57-65 minus the transform, which runs on the GPU)."""
        if self.npy:
            arr = np.asarray(self.indexs[index], dtype=np.uint8)
        else:
            from PIL import Image
            arr = np.asarray(Image.open(str(self.indexs[index]).strip()).convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(np.ascontiguousarray(arr))

    def _load_text(self, index: int):
        """reference :67-87: one caption (random among several), CLS + BPE tokens cut to maxWords - 1, SEP, zero padding."""
        captions = self.captions[index]
        use_cap = captions[random.randint(0, len(captions) - 1)]
        words = [SPECIAL_TOKEN["CLS_TOKEN"]] + self.tokenizer.tokenize(str(use_cap))
        if len(words) > self.maxWords - 1:
            words = words[: self.maxWords - 1]
        caption = self.tokenizer.convert_tokens_to_ids(words + [SPECIAL_TOKEN["SEP_TOKEN"]])
        caption = torch.tensor(caption + [0] * (self.maxWords - len(caption)))
        return caption, caption == 0

    def _load_label(self, index: int) -> torch.Tensor:
        return torch.from_numpy(np.asarray(self.labels[index]))

    def get_all_label(self):
        labels = torch.zeros([self._length, len(self.labels[0])], dtype=torch.int64)
        for i, item in enumerate(self.labels):
            labels[i] = torch.from_numpy(np.asarray(item))
        return labels

    def get_tag_length(self):
        return self.captions.shape[-1]

    def __getitem__(self, index):
        image = self._load_image(index)
        caption, key_padding_mask = self._load_text(index)
        return image, caption, key_padding_mask, self._load_label(index), index

    @staticmethod
    def collate(batch):
        """images stay a list when their sizes differ (undecoded photos); everything else stacks as usual."""
        images = [b[0] for b in batch]
        same = all(tuple(i.shape) == tuple(images[0].shape) for i in images)
        return (torch.stack(images) if same else images, torch.stack([b[1] for b in batch]), torch.stack([b[2] for b in batch]),
                torch.stack([b[3] for b in batch]), torch.tensor([b[4] for b in batch], dtype=torch.int64))


def split_data(captions, indexs, labels, query_num=5000, train_num=10000, random_index=None):
    """reference dataset/builder.py:8-32: one permutation; queries first, training set = head of the retrieval set."""
    if random_index is None:
        random_index = np.random.permutation(range(len(indexs)))
    q, t, r = random_index[:query_num], random_index[query_num: query_num + train_num], random_index[query_num:]
    return (indexs[q], indexs[t], indexs[r]), (captions[q], captions[t], captions[r]), (labels[q], labels[t], labels[r])


def _first_key(mat: dict, keys, what: str):
    for k in keys:
        if k in mat:
            return mat[k]
    raise RuntimeError("%s file is not support, we only read the keys of %s." % (what, list(keys)))


def build_dataloader(captionFile: str, indexFile: str, labelFile: str, imageResolution=224, query_num=5000, train_num=10000,
                     dataset_cls=None, **kwargs):
    """reference dataset/builder.py:34-104 -> (train, query, retrieval) datasets; the training split is returned as None
    (training is out of scope), the other two are evaluation datasets."""
    import scipy.io as scio
    assert dataset_cls is not None, "'dataset_cls' must be provided!"
    dataset = registry.get_dataset_class(dataset_cls)
    if captionFile.endswith("mat"):
        captions = _first_key(scio.loadmat(captionFile), ("caption", "tags", "YAll"), "text")
        captions = captions[0] if captions.shape[0] == 1 else captions
    elif captionFile.endswith("txt"):
        with open(captionFile, "r") as f:
            captions = np.asarray([[item.strip()] for item in f.readlines()])
    else:
        raise ValueError("the format of 'captionFile' doesn't support, only support [txt, mat] format.")
    if indexFile.endswith("mat"):
        npy, indexs = False, _first_key(scio.loadmat(indexFile), ("index", "imgs", "FAll"), "image")
    elif indexFile.endswith("npy"):
        npy, indexs = True, np.load(indexFile)
    else:
        raise RuntimeError("index file is not support, we only read the keys of [*.mat, *.npy].")
    labels = _first_key(scio.loadmat(labelFile), ("category", "LAll", "labels"), "label")
    s_idx, s_cap, s_lab = split_data(captions, indexs, labels, query_num=query_num, train_num=train_num)
    kwargs.pop("img_train_transform", None)
    mk = lambda i: dataset(captions=s_cap[i], indexs=s_idx[i], labels=s_lab[i], imageResolution=imageResolution, is_train=False, npy=npy, **kwargs)   # noqa: E731
    return None, mk(0), mk(2)
