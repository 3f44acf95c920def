"""ctypes wrapper of oracle/libxmh_oracle.so (plain-C integer oracle).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libxmh_oracle.so")


def _load():
    if not os.path.exists(_SO):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return C.CDLL(_SO)


_lib = _load()
_p = C.c_void_p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def hamming(qbits, rbits):
    qbits, rbits = _u32(qbits), _u32(rbits)
    out = np.empty((qbits.shape[0], rbits.shape[0]), dtype=np.uint16)
    _lib.orc_hamming(_ptr(qbits), _ptr(rbits), C.c_int64(qbits.shape[0]), C.c_int64(rbits.shape[0]), qbits.shape[1], _ptr(out))
    return out


def hist(qbits, qlab, rbits, rlab, nb):
    qbits, qlab, rbits, rlab = map(_u32, (qbits, qlab, rbits, rlab))
    Q = qbits.shape[0]
    ha = np.empty((Q, nb), dtype=np.uint32)
    hr = np.empty((Q, nb), dtype=np.uint32)
    _lib.orc_hist(_ptr(qbits), _ptr(qlab), _ptr(rbits), _ptr(rlab), C.c_int64(Q), C.c_int64(rbits.shape[0]),
                  qbits.shape[1], qlab.shape[1], nb, _ptr(ha), _ptr(hr))
    return ha, hr


def ap(qbits, qlab, rbits, rlab, nb, k=None, base_all=None, base_rel=None, nrel_total=None):
    qbits, qlab, rbits, rlab = map(_u32, (qbits, qlab, rbits, rlab))
    Q = qbits.shape[0]
    if base_all is not None:
        base_all, base_rel, nrel_total = _u32(base_all), _u32(base_rel), _u32(nrel_total)
    s = np.empty(Q, dtype=np.float64)
    cap = np.empty(Q, dtype=np.int32)
    _lib.orc_ap(_ptr(qbits), _ptr(qlab), _ptr(rbits), _ptr(rlab), C.c_int64(Q), C.c_int64(rbits.shape[0]), qbits.shape[1],
                qlab.shape[1], nb, _ptr(base_all), _ptr(base_rel), _ptr(nrel_total), C.c_int64(0 if k is None else k),
                _ptr(s), _ptr(cap))
    return s, cap


def topk(qbits, rbits, nb, k, base_index=0):
    qbits, rbits = _u32(qbits), _u32(rbits)
    Q = qbits.shape[0]
    d = np.empty((Q, k), dtype=np.uint16)
    i = np.empty((Q, k), dtype=np.int32)
    _lib.orc_topk(_ptr(qbits), _ptr(rbits), C.c_int64(Q), C.c_int64(rbits.shape[0]), qbits.shape[1], nb, k,
                  C.c_int64(base_index), _ptr(d), _ptr(i))
    return d, i


def topk_ternary(qbits, qzero, rbits, rzero, K, k, base_index=0):
    """top-k of ternary codes (bits + zero planes, padding bits set in the zero planes); distances in half units K - q.r"""
    qbits, qzero, rbits, rzero = map(_u32, (qbits, qzero, rbits, rzero))
    Q = qbits.shape[0]
    d = np.empty((Q, k), dtype=np.uint16)
    i = np.empty((Q, k), dtype=np.int32)
    _lib.orc_topk_ternary(_ptr(qbits), _ptr(qzero), _ptr(rbits), _ptr(rzero), C.c_int64(Q), C.c_int64(rbits.shape[0]), qbits.shape[1],
                          int(K), int(k), C.c_int64(base_index), _ptr(d), _ptr(i))
    return d, i
