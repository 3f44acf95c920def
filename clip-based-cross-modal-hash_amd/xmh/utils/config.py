"""Minimal stand-in for the OmegaConf objects the reference passes around (main.py:15-22): attribute access,
``.get(key, default)`` and item access over nested dicts loaded from the same YAML layout
(``model`` / ``optimizer`` / ``dataset`` / ``run`` sections)."""
from __future__ import annotations

import yaml


class Config(dict):
    def __init__(self, data=None, **kw):
        super().__init__()
        for k, v in dict(data or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def load_yaml(path: str, save_dir=None) -> Config:
    """OmegaConf.load + the two fields main.py injects (run.save_dir, run.log_dir; main.py:20,36)."""
    with open(path) as f:
        cfg = Config(yaml.safe_load(f))
    if save_dir is not None:
        cfg.run.save_dir = save_dir
    if "run" in cfg and "log_dir" not in cfg.run:
        cfg.run.log_dir = cfg.run.get("save_dir", "./result")
    return cfg
