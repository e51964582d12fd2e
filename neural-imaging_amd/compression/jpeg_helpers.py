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
    """Zig-zag scan index matrix (jpeg_helpers.py:253-261)."""
    def compare(xy):
        x, y = xy
        return (x + y, -y if (x + y) % 2 else y)
    zz = np.zeros((n, n), dtype=np.uint16)
    for i, (x, y) in enumerate(sorted(((x, y) for x in range(n) for y in range(n)), key=compare)):
        zz[x, y] = i
    return zz


def jpeg_qf_estimation(q_mtx, channel=0):
    errors = [np.mean(np.abs(jpeg_qtable(qf, channel) - q_mtx)) for qf in range(1, 101)]
    return int(np.argmin(errors) + 1)
