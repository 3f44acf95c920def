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
    try:
        import xlrd  # noqa: F401
    except ImportError:
        sys.modules["xlrd"] = _xlrd_stand_in()


def _xlrd_stand_in():
    """The one xlrd call chain the reference makes -- ``xlrd.open_workbook(path).sheet_by_index(0).row(i)[j].value``
    (models/DSPH/DSPH.py:33-35: the HyP loss threshold out of models/DSPH/loss/codetable.xlsx) -- over the real workbook: an .xlsx is
    a zip of XML, its numeric cells are read with zipfile + a regular expression.  Rows and columns are 0-based as in xlrd; an absent
    cell has the value ''.  xlrd itself is not in the image (and has not read .xlsx since 2.0)."""
    import re
    import zipfile

    class Cell:
        def __init__(self, value):
            self.value = value

    class Sheet:
        def __init__(self, xml):
            self.cells = {}
            for col, row, val in re.findall(r'<c r="([A-Z]+)(\d+)"[^>]*><v>([^<]*)</v>', xml):
                c = 0
                for ch in col:
                    c = c * 26 + (ord(ch) - 64)
                self.cells[(int(row) - 1, c - 1)] = float(val)
            self.ncols = 1 + max(c for _, c in self.cells)

        def row(self, i):
            return [Cell(self.cells.get((i, j), "")) for j in range(self.ncols)]

    class Book:
        def __init__(self, path):
            with zipfile.ZipFile(path) as z:
                self.sheets = [Sheet(z.read("xl/worksheets/sheet1.xml").decode())]

        def sheet_by_index(self, i):
            return self.sheets[i]

    m = types.ModuleType("xlrd")
    m.open_workbook = Book
    return m
