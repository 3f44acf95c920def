"""Datasets.  The reference's .mat/PIL/BPE pipeline is host-side I/O outside the hot path;
what the runner depends on is the sample tuple ``(image, caption, key_padding_mask, label, index)``
(dataset/transformer_dataset.py:102-107) and ``get_all_label()`` (:95-100).  ``SyntheticPairs`` provides exactly
that from seeded generators."""
from .synthetic import SyntheticPairs, build_synthetic_splits  # noqa: F401
