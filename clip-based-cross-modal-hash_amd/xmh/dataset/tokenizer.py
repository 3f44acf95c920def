"""CLIP byte-pair tokenizer behind the reference's registry name ``clip_tokenizer`` (reference
models/CLIP/simple_tokenizer.py:63-146: ``tokenize`` -> BPE token strings, ``convert_tokens_to_ids``, ``encode``,
``decode``; instantiated with no arguments by runners/base.py:158).  Host-side text plumbing, not on the GPU path.

The merge table is OpenAI's ``bpe_simple_vocab_16e6.txt.gz`` -- a data file the reference keeps next to its tokenizer and
that this repository does not ship.  It is looked up, in this order, at: the constructor argument, ``$XMH_BPE_VOCAB``,
``./models/CLIP/bpe_simple_vocab_16e6.txt.gz`` (a reference checkout as working directory) and next to this module.
``ftfy`` is used for mojibake repair when it is installed, exactly like the reference; without it only HTML entities
are unescaped (identical for clean captions)."""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache

import regex as re

from ..common.register import registry

try:                                       # optional, like every other text-repair dependency of the host pipeline
    import ftfy
except ImportError:                        # pragma: no cover
    ftfy = None

VOCAB_NAME = "bpe_simple_vocab_16e6.txt.gz"
N_MERGES = 49152 - 256 - 2                 # merges used by CLIP's 49408-entry vocabulary


def find_vocab(path: str = None) -> str:
    candidates = [path, os.environ.get("XMH_BPE_VOCAB"), os.path.join("models", "CLIP", VOCAB_NAME),
                  os.path.join(os.path.dirname(os.path.abspath(__file__)), VOCAB_NAME)]
    for c in candidates:
        if c and os.path.isfile(c):
            return c
    raise FileNotFoundError("CLIP BPE merge table %s not found; pass its path or set XMH_BPE_VOCAB (the reference keeps it "
                            "in models/CLIP/)" % VOCAB_NAME)


@lru_cache()
def byte_alphabet() -> dict:
    """byte value -> printable unicode character: printable latin-1 bytes map to themselves, the rest to 256, 257, ..."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    return table


def _clean(text: str) -> str:
    if ftfy is not None:
        text = ftfy.fix_text(text)
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


@registry.register_tokenizer("clip_tokenizer")
class ClipTokenizer:
    def __init__(self, bpe_path: str = None):
        alphabet = byte_alphabet()
        self.byte_encoder = alphabet
        self.byte_decoder = {c: b for b, c in alphabet.items()}
        with gzip.open(find_vocab(bpe_path)) as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(line.split()) for line in lines[1:N_MERGES + 1]]
        # vocabulary order: single symbols in alphabet-construction order, the same with the end-of-word mark, the merges, specials
        order = [alphabet[b] for b in list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))]
        order += [alphabet[b] for b in range(256) if alphabet[b] not in order]
        vocab = order + [s + "</w>" for s in order] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self._pairs = merges
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)

    def bpe(self, token: str) -> str:
        """greedy lowest-rank merging of one pre-token (already mapped through the byte alphabet) -> space-joined symbols."""
        hit = self.cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            ranked = [(self.bpe_ranks.get((a, b), None), i) for i, (a, b) in enumerate(zip(word, word[1:]))]
            ranked = [r for r in ranked if r[0] is not None]
            if not ranked:
                break
            best = min(ranked)[0]
            first, second = self._pairs[best]
            merged, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self.cache[token] = out
        return out

    def tokenize(self, text: str) -> list:
        tokens = []
        for piece in re.findall(self.pat, _clean(text).lower()):
            piece = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
            tokens.extend(self.bpe(piece).split(" "))
        return tokens

    def convert_tokens_to_ids(self, tokens) -> list:
        return [self.encoder[t] for t in tokens]

    def encode(self, text: str) -> list:
        return self.convert_tokens_to_ids(self.tokenize(text))

    def decode(self, ids) -> str:
        text = "".join(self.decoder[i] for i in ids)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")
