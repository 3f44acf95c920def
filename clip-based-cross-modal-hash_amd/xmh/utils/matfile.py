"""Writer for the `.mat` result files of `BaseTrainer.save_mat` (reference runners/base.py:386-405 calls `scipy.io.savemat`).

Level-5 MAT-file, uncompressed -- what `scipy.io.savemat` writes by default and what `scipy.io.loadmat` / MATLAB read.  A result file is
six real 2-D matrices (four code matrices fp32 [N, K], two label matrices as the dataset stores them), ~140 MB at the COCO shape; the
format stores them column-major, and scipy gets there with a strided host copy of every matrix (0.2 s per file, most of it the 75 MB
of int64 labels).  Here the transposition happens where the matrix lives -- on the GPU for the codes the evaluation just produced --
and the host only writes headers and bytes.

    write_mat5(path, {"q_img": tensor_or_array, ...})

Anything this writer does not cover (more than two dimensions, complex, strings, objects, > 4 GB per matrix) raises
`UnsupportedMatValue`; `BaseTrainer.save_mat` falls back to scipy then.  File layout (MAT-File Format, "Level 5 MAT-File Format"):
128-byte header, then per variable one miMATRIX element = array flags, dimensions, name, real part."""
import os
import struct
import time

import numpy as np
import torch

# numpy dtype -> (mx class of the array flags, mi type of the data element)
_CLASSES = {
    np.dtype(np.float64): (6, 9), np.dtype(np.float32): (7, 7),
    np.dtype(np.int8): (8, 1), np.dtype(np.uint8): (9, 2), np.dtype(np.int16): (10, 3), np.dtype(np.uint16): (11, 4),
    np.dtype(np.int32): (12, 5), np.dtype(np.uint32): (13, 6), np.dtype(np.int64): (14, 12), np.dtype(np.uint64): (15, 13),
}
_MI_INT8, _MI_INT32, _MI_UINT32, _MI_MATRIX = 1, 5, 6, 14


class UnsupportedMatValue(TypeError):
    pass


class Prepared:
    """a matrix already in file form (`prepare(value)`): what a caller keeps for matrices it writes again and again, e.g. the label
    matrices of an evaluation, which do not change between epochs"""
    __slots__ = ("rows", "cols", "dtype", "host")

    def __init__(self, rows, cols, dtype, host):
        self.rows, self.cols, self.dtype, self.host = rows, cols, dtype, host


def prepare(value):
    return value if isinstance(value, Prepared) else Prepared(*_column_major(value))


def _column_major(value):
    """-> (rows, cols, numpy dtype, C-contiguous host array [cols, rows] holding the column-major bytes of the [rows, cols] matrix)"""
    if isinstance(value, Prepared):
        return value.rows, value.cols, value.dtype, value.host
    if isinstance(value, torch.Tensor):
        t = value.detach()
        if t.dtype == torch.bool:
            t = t.to(torch.uint8)                                   # scipy stores bool arrays as uint8 as well
        if t.dim() == 0:
            t = t.reshape(1, 1)
        elif t.dim() == 1:
            t = t.reshape(1, -1)                                    # savemat's default oned_as="row"
        if t.dim() != 2 or t.is_complex():
            raise UnsupportedMatValue("tensor of shape %s / dtype %s" % (tuple(value.shape), value.dtype))
        rows, cols = int(t.shape[0]), int(t.shape[1])
        host = t.t().contiguous().cpu().numpy()                     # transposed where the data lives, one device-to-host copy
    else:
        a = np.asarray(value)
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(1, -1)
        if a.ndim != 2:
            raise UnsupportedMatValue("array of shape %s" % (a.shape,))
        rows, cols = int(a.shape[0]), int(a.shape[1])
        host = np.ascontiguousarray(a.T)
    if host.dtype.byteorder not in ("=", "|", "<"):                # big-endian input: convert the BYTES, not only the label
        host = host.astype(host.dtype.newbyteorder("="))
    dt = host.dtype
    if np.dtype(dt) not in _CLASSES:
        raise UnsupportedMatValue("dtype %s" % host.dtype)
    if host.nbytes >= (1 << 32) - 64:
        raise UnsupportedMatValue("matrix of %d bytes (the level-5 element size is 32 bit)" % host.nbytes)
    return rows, cols, np.dtype(dt), host


def _pad8(n):
    return (8 - n % 8) % 8


def write_mat5(path, variables):
    """Write `variables` (name -> torch tensor on any device / numpy array / nested lists, real, at most 2-D) as a level-5 MAT-file.
    The file appears under its name only when complete (written beside it, then renamed): a reader or a hard link to an older
    file of that name never sees a partial write."""
    elements = []
    for name, value in variables.items():
        nm = name.encode("latin1")
        if not nm or len(nm) > 63 or name.startswith("_"):
            raise UnsupportedMatValue("variable name %r" % name)
        rows, cols, dt, host = _column_major(value)
        mx, mi = _CLASSES[dt]
        head = struct.pack("<IIII", _MI_UINT32, 8, mx, 0)                         # array flags: class, no complex / global / logical bits
        head += struct.pack("<IIii", _MI_INT32, 8, rows, cols)                    # dimensions
        head += struct.pack("<II", _MI_INT8, len(nm)) + nm + b"\0" * _pad8(len(nm))
        head += struct.pack("<II", mi, host.nbytes)                               # real part: tag, then the column-major bytes
        pad = _pad8(host.nbytes)
        elements.append((struct.pack("<II", _MI_MATRIX, len(head) + host.nbytes + pad) + head, host, pad))
    text = ("MATLAB 5.0 MAT-file Platform: posix, Created on: %s (xmh.utils.matfile)" % time.asctime()).encode("latin1")[:116]
    header = text + b" " * (116 - len(text)) + b"\0" * 8 + struct.pack("<H", 0x0100) + b"IM"
    tmp = "%s.tmp%d" % (path, os.getpid())
    try:
        with open(tmp, "wb") as f:
            f.write(header)
            for head, host, pad in elements:
                f.write(head)
                f.write(host.reshape(-1).view(np.uint8).data)
                if pad:
                    f.write(b"\0" * pad)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):                                     # a full disk or an interrupt leaves no half-written file behind
            os.remove(tmp)
        raise


def link_or_copy(src, dst):
    """`dst` becomes another name of the finished file `src` (hard link: no bytes move) or, where the filesystem has no links, a copy.
    Safe against later rewrites of either name because write_mat5 never writes in place."""
    import shutil
    tmp = "%s.tmp%d" % (dst, os.getpid())
    try:
        if os.path.exists(tmp):
            os.remove(tmp)
        os.link(src, tmp)
    except OSError:
        shutil.copyfile(src, tmp)
    os.replace(tmp, dst)
