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


def test_the_binding_brings_torch_in_before_the_library():
    """One HIP runtime per process: libxmh.so loaded before PyTorch initialises /opt/rocm's runtime beside the one torch ships, and the
    second one to be used reports "no ROCm-capable device" (build() followed by smoke() in one process did).  xmh/_lib.py therefore
    imports torch above its CDLL call -- checked in a fresh interpreter, where nothing else has imported torch yet."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import xmh._lib as L; "
            "assert 'torch' in sys.modules and L.lib.xmh_version() >= 100; print('ok')") % os.path.join(ROOT, "clip-based-cross-modal-hash_amd")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-500:]


def test_the_in_tree_library_is_a_build_of_the_sources_beside_it():
    """libxmh.so carries the sha256 of the sources it was compiled from (xmh_build_id): a stale in-tree library -- a source edited and
    the library not rebuilt -- fails here instead of travelling to the GPU box"""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    info = bench.build_info()
    assert info["lib_is_a_build_of_these_sources"], info
    assert len(info["build_id"]) == 16
