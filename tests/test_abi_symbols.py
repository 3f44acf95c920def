"""The C-ABI library must load (no GPU needed) and export every function include/xmh.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "xmh.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xmh_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("xmh_pack_sign", "xmh_pack_pair_argmax", "xmh_pack_labels", "xmh_hamming_dist", "xmh_hamming_hist",
                 "xmh_hamming_ap", "xmh_hamming_topk", "xmh_map_finalize", "xmh_last_error", "xmh_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from xmh import _lib
    so = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(so, n)]
    assert not missing, missing
    assert set(_lib.PROTOTYPES) == set(_declared()), set(_lib.PROTOTYPES) ^ set(_declared())
    assert so.xmh_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    from xmh import _lib
    rc = _lib.lib.xmh_pack_sign(None, 4, 0, None, None, None, None, None)
    assert rc == -22 and b"xmh_pack_sign" in _lib.lib.xmh_last_error()
    plan = _lib.ScanPlan()
    assert _lib.lib.xmh_scan_plan_make(0, 10, 64, 0, ctypes.byref(plan)) == -22
    assert _lib.lib.xmh_scan_plan_make(5000, 117218, 64, 0, ctypes.byref(plan)) == 0
    assert plan.nbuckets == 65 and plan.qpad == 5120 and plan.chunk * plan.nchunk >= 117218      # whole 128-query blocks of pass 1
    assert _lib.lib.xmh_scan_plan_make(10, 1000, 2048, 0, ctypes.byref(plan)) == 0 and plan.nbuckets == 2049    # long binary codes
    assert _lib.lib.xmh_scan_plan_make(10, 1000, 4096, 0, ctypes.byref(plan)) == -95      # beyond 2048 bits
    assert _lib.lib.xmh_scan_plan_make(10, 1000, 512, 1, ctypes.byref(plan)) == -95       # zero planes only up to 256 bits
    assert b"at most 2048" in _lib.lib.xmh_last_error()
    assert _lib.lib.xmh_topk_ws_bytes(8, 1000, 64, 0) == 0
