"""
JPEG table helpers with the reference's names (compression/jpeg_helpers.py:253-310).  jpeg_qtable goes through the
native library (nimg_jpeg_qtable) so the tables the kernels see are the ones tested bit-exactly against the reference.
The libjpeg batch codec / bit-stream parser of the reference file are CPU validation tooling and out of scope.
"""
import numpy as np

from .. import ops


def jpeg_qtable(quality, channel=0):
    """DCT quantisation matrix for a quality level 1..100; channel 0 = luma, >0 = chroma. Returns (8,8) float32."""
    return ops.qtable(int(np.maximum(np.minimum(100, quality), 1)), int(channel))


def zigzag(n):
    """Zig-zag scan index matrix (jpeg_helpers.py:253-261): entry [r, c] = position of coefficient (r, c) in the scan.  The
    scan walks the anti-diagonals d = r + c in order; odd diagonals run top-right -> bottom-left (r ascending), even ones the
    other way."""
    r, c = np.divmod(np.arange(n * n), n)
    d = r + c
    order = np.lexsort((np.where(d % 2 == 1, r, -r), d))          # primary key d, secondary the walking direction
    zz = np.empty(n * n, dtype=np.uint16)
    zz[order] = np.arange(n * n, dtype=np.uint16)
    return zz.reshape(n, n)


def jpeg_qf_estimation(q_mtx, channel=0):
    """The IJG quality 1..100 whose table of `channel` is closest (mean absolute difference) to q_mtx; ties -> the lowest."""
    tables = np.stack([jpeg_qtable(qf, channel) for qf in range(1, 101)]).astype(np.float64)
    return 1 + int(np.abs(tables - np.asarray(q_mtx, dtype=np.float64)).mean(axis=(1, 2)).argmin())
