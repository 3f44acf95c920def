"""Import helpers for the READ-ONLY reference checkout (this container only).

TEST INFRASTRUCTURE ONLY -- used by ``oracle/make_golden*.py`` to generate the
fixtures under ``tests/golden/``.  ``/root/reference`` does not exist on the GPU
box; nothing at test/bench/smoke run time imports this module.

The reference's package ``__init__`` files pull in dependencies that are not in
the image (torchvision, omegaconf, ftfy, termcolor, xlrd).  We pre-seed
``sys.modules`` with bare namespace packages pointing at the real directories so
the individual modules we need import unmodified.
"""
import os
import sys
import types

REF = os.environ.get("XMH_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "common"))


def _ns(name: str, path: str):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def setup():
    """Make ``common.*``, ``models.*`` and ``runners.*`` of the reference importable."""
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for pkg in ("models", "models.CLIP", "models.DSPH", "models.DCMHT", "models.MITH", "models.DSPH.hash",
                "models.DCMHT.hash", "dataset", "runners", "runners.DCMHT", "runners.MITH", "runners.DSPH", "utils"):
        _ns(pkg, os.path.join(REF, *pkg.split(".")))
    if "termcolor" not in sys.modules:
        tc = types.ModuleType("termcolor")
        tc.colored = lambda s, *a, **k: s
        sys.modules["termcolor"] = tc
