"""
Tensor-level wrappers over the C ABI (include/nimg.h).  torch is used ONLY for device memory (torch.empty / views)
and the current HIP stream; all arithmetic happens inside libnimg.so.  Every function requires contiguous float32
CUDA(HIP) tensors in NHWC and raises otherwise - there is no CPU path.

TF padding semantics live here (same_pads): SAME is asymmetric for strided convolutions (extra sample after).
"""
import math
import os

import numpy as np
import torch

from . import _lib

LRELU_ALPHA = 0.2          # tf.keras.layers.LeakyReLU(alpha=0.2), helpers/tf_helpers.py:23

# Arithmetic type of the convolution GEMMs: 'f32' = exact float32 MFMA (parity mode, default) | 'bf16' = bf16 operands
# with float32 accumulation on the matrix cores (throughput mode; tensors in HBM stay float32).
COMPUTE = 'f32'


# Throughput mode keeps the FAN-internal tensors (pooled activations, un-pooled / pooled gradients) in HBM as bf16: their
# consumers round to bf16 anyway, so results are bit-identical (test_fan_bf16_storage_is_bit_neutral); False = float32.
STORE_BF16 = True


def set_compute(mode):
    global COMPUTE
    if mode not in ('f32', 'bf16'):
        raise ValueError('compute mode must be f32 or bf16')
    COMPUTE = mode


_WB_REGISTRY = {}        # data_ptr of a registered 4-D weight -> {'shape': ..., 0: image, 1: image} (see WeightImages)


class WeightImages(object):
    """All bf16 weight images (both layouts) of one model's convolution kernels, rebuilt by ONE launch per step
    (refresh(), called at the top of the model's forward) instead of one tiny launch per layer and pass."""

    def __init__(self, weights, device):
        lib = _lib.load()
        self.entries, sizes = [], []
        for w in weights:
            kh, kw, cin, cout = w.shape
            for mode in (0, 1):
                sizes.append(int(lib.nimg_conv_weights_bf16_bytes(kh, kw, cin, cout, mode)))
                self.entries.append((w, mode))
        offs = np.concatenate([[0], np.cumsum([-(-s // 256) * 256 for s in sizes])]).astype(np.int64)
        self.buf = torch.empty(int(offs[-1]) or 256, dtype=torch.uint8, device=device)
        table = np.zeros((len(self.entries), 4), np.int64)
        for i, (w, mode) in enumerate(self.entries):
            kh, kw, cin, cout = w.shape
            view = self.buf[int(offs[i]):int(offs[i]) + sizes[i]]
            table[i] = (w.data_ptr(), view.data_ptr(), ((kh * kw) << 32) | mode, (cin << 32) | cout)
            if _WB_REGISTRY.get(w.data_ptr(), {}).get('shape') != tuple(w.shape):
                _WB_REGISTRY[w.data_ptr()] = {'shape': tuple(w.shape)}
            _WB_REGISTRY[w.data_ptr()][mode] = view
        self.table = torch.from_numpy(table).to(device)
        # addresses are recycled by the caching allocator: drop the registry entries with the model
        import weakref
        keys = [(w.data_ptr(), tuple(w.shape)) for w in weights]
        weakref.finalize(self, WeightImages._forget, keys, self.buf.data_ptr(), self.buf.numel())

    @staticmethod
    def _forget(keys, base, size):
        for ptr, shape in keys:
            reg = _WB_REGISTRY.get(ptr)
            if reg is not None and reg['shape'] == shape and \
                    all(base <= reg[m].data_ptr() < base + size for m in (0, 1) if m in reg):
                del _WB_REGISTRY[ptr]

    def refresh(self):
        if COMPUTE == 'bf16' and len(self.entries):
            _lib.call('nimg_conv_weights_bf16_batch', _p(self.table), len(self.entries), _stream())


def weights_bf16(w, mode):
    """bf16 weight image for the throughput-mode kernels (re-made every step from the float32 master weights): the
    model-wide pre-built image if the weight is registered (WeightImages), else converted here."""
    reg = _WB_REGISTRY.get(w.data_ptr())
    if reg is not None and reg['shape'] == tuple(w.shape) and mode in reg:
        return reg[mode]
    kh, kw, cin, cout = w.shape
    nbytes = int(_lib.load().nimg_conv_weights_bf16_bytes(kh, kw, cin, cout, mode))
    wb = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _lib.call('nimg_conv_weights_bf16', _p(w), _p(wb), kh, kw, cin, cout, mode, _stream())
    return wb


# The raw handle of the current stream of the current device.  torch.cuda.current_stream() builds a Stream object through four
# Python layers (device-index resolution, is_available(), an os.environ look-up ...): ~7 us, 230 times per training step = 1.6 of
# the 3.5 ms of host time an eager step costs (tools/host_profile.py); the two C entry points below answer in ~0.3 us.
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError('nimg ops need contiguous device tensors (got device={}, contiguous={})'.format(
                t.device, t.is_contiguous()))


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError('nimg ops need float32 tensors, got {}'.format(t.dtype))
    _chk(*ts)


D2S_EPILOGUE = os.environ.get('NIMG_NO_D2S_OUT') is None          # A/B switch: depth_to_space as a separate pass
BF16_IN, BF16_OUT, BF16_MASK, BF16_DZ, D2S_OUT, S2D_OUT, COPY_LRELU, MASK_CONV = 1, 2, 4, 8, 16, 32, 64, 128          # include/nimg.h NIMG_BF16_*, NIMG_D2S_OUT


def _fb(*ts):
    """float32 or bfloat16 device tensors (bf16 STORAGE of FAN-internal tensors in throughput mode)."""
    for t in ts:
        if t is not None and t.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError('nimg ops need float32 / bfloat16 tensors, got {}'.format(t.dtype))
    _chk(*ts)


def _is_bf16(t):
    return t is not None and t.dtype == torch.bfloat16


def same_pads(size, k, s):
    """TF SAME padding: total = max((ceil(in/s)-1)*s + k - in, 0); before = total // 2 (rest after)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


# Buffers that live OUTSIDE a step's own allocations (the grow-only workspaces below, the manipulations' filter-tap cache)
# are baked into a captured HIP graph as raw pointers.  While a pin log is open (graphs.CapturedStep opens one around its
# warm-up and capture) every such buffer handed out is also recorded there; the capture keeps the list, so a later eager call
# that re-grows a workspace or evicts a cached table replaces the OWNER's reference only - the memory the graph replays on
# stays allocated for as long as the captured step lives.
_PIN_LOG = None


def begin_pin_log():
    global _PIN_LOG
    _PIN_LOG = []


def end_pin_log():
    global _PIN_LOG
    log, _PIN_LOG = _PIN_LOG or [], None
    seen, out = set(), []
    for t in log:
        if id(t) not in seen:
            seen.add(id(t))
            out.append(t)
    return out


def pin(t):
    if _PIN_LOG is not None and t is not None:
        _PIN_LOG.append(t)
    return t


class Workspace(object):
    """A grow-only scratch buffer (split-K slabs, reduction partials)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        nbytes = max(int(nbytes), 256)
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return pin(self.buf)

    # deferred slab reductions (DEFER_REDUCE): every pending weight gradient keeps its OWN piece of an arena until the batched
    # reduction has been issued; the arena grows by replacement (the pending pieces hold the old one alive) and is re-used from
    # the start after each flush
    arena, used, pending = None, 0, ()

    def claim(self, nbytes, device):
        nbytes = (max(int(nbytes), 256) + 255) // 256 * 256
        if self.arena is None or self.arena.device != device or self.used + nbytes > self.arena.numel():
            size = max(2 * nbytes, 64 << 20, 0 if self.arena is None else 2 * self.arena.numel())
            if self.arena is not None:
                self.pending = self.pending + (self.arena,)
            self.arena, self.used = torch.empty(size, dtype=torch.uint8, device=device), 0
        piece = self.arena[self.used:self.used + nbytes]
        self.used += nbytes
        pin(self.arena)
        return piece

    def release(self):
        self.used, self.pending = 0, ()


_ws = Workspace()

# Weight / bias gradients of a layer and its input gradient depend only on the same dz, and many of them (the UNet
# levels, 512 workgroups each) cannot fill 256 CUs on their own: layers launch their parameter gradients on side HIP
# streams so the kernels share the machine.  A side stream first waits for everything queued on the launch stream,
# the tensors it reads are pinned against early reuse by the caching allocator (record_stream), and every consumer of
# the gradient buffers (optimiser, all-reduce, NaN check, end of each model's backward) joins them.
# NIMG_SIDE_STREAMS=n (default 2): the parameter gradients of different layers are independent of each other as well - on
# ONE side stream the UNet's ~25 weight-gradient + slab-reduction pairs (20 - 70 us each, a dependent launch gap between
# them) finish ~0.25 ms after the input-gradient chain they run beside (profiles/r03_ao_c4_launch_trace.txt).  Round 3 spread
# them over three; since the FAN's weight gradients are issued late (models/forensics.py LATE_PARAMS) three streams put three
# chip-filling 5x5 weight gradients beside the launch stream at once (profiles/r04_final_c4_launch_trace.txt: djpeg_bwd waits
# 1.2 ms for a CU) - with two, every workload is 1 - 4 % faster and the captured two-branch graph replays as fast as eager
# launches (profiles/r04_side_streams_2_vs_3.txt).  A gradient buffer always goes to the same stream (`key`), so accumulating
# launches stay ordered; each stream has its own split-K scratch.  NIMG_NO_SIDE_STREAM=1 keeps everything on one stream.
import os as _os

_SIDE = {'streams': [], 'ws': [Workspace()], 'n': max(1, int(_os.environ.get('NIMG_SIDE_STREAMS', '2'))), 'next': 0, 'keys': {},
         'enabled': _os.environ.get('NIMG_NO_SIDE_STREAM') is None, 'dirty': set()}
_ws_side = _SIDE['ws'][0]       # split-K scratch of the kernels that run on the (first) side stream


def _side_index(key):
    """The side stream of a gradient buffer: assigned round-robin when the buffer is first seen, the same one ever after."""
    if key is None or _SIDE['n'] == 1:
        return 0
    k = _SIDE['keys'].get(key)
    if k is None:
        k = _SIDE['keys'][key] = _SIDE['next'] % _SIDE['n']
        _SIDE['next'] += 1
    return k


# Arrival counters of the in-kernel split-K finish (include/nimg.h nimg_bind_tickets, csrc/common.h ticket_finish): one zeroed
# 64 KB buffer per (device, stream) that launches weight gradients, bound the first time that stream asks for its scratch buffer.
# The library leaves the counters zero after every launch.  OPT-IN (NIMG_TICKETS=1) because it LOSES (profiles/r06_tickets_ab.txt:
# UNet backward 1667 -> 2304 us, C4 step 8.02 -> 8.68 ms): the last-arriving workgroup of a dw tile reads splits x tile bytes
# (2.4 MB for a 256 -> 256 layer) ALONE at ~100 GB/s while the rest of the chip has nothing left to do - a serial 20 - 30 us tail per
# launch - where the separate reduction spreads the same 38 MB over every CU in ~10 us.  Without it every weight gradient keeps its
# separate reduction launch; the sums differ in the order of the additions only.
_TICKETS = {}
TICKETS = _os.environ.get('NIMG_TICKETS') is not None


def _bind_tickets(device, handle):
    key = (device.index, handle)
    buf = _TICKETS.get(key)
    if buf is None:
        buf = _TICKETS[key] = torch.zeros(_lib.TICKET_BYTES, dtype=torch.uint8, device=device)
        _lib.call('nimg_bind_tickets', handle, buf.data_ptr(), buf.numel())
    pin(buf)


def _ws_current(device):
    """The scratch buffer of the stream the caller launches on."""
    cur = _stream() if (_raw_device is not None and device.index in (None, _raw_device())) else \
        torch.cuda.current_stream(device).cuda_stream
    if TICKETS:
        _bind_tickets(device, cur)
    for k, st in enumerate(_SIDE['streams']):
        if st is not None and st.device == device and cur == st.cuda_stream:
            return _SIDE['ws'][k]
    return _ws


# NIMG_SIDE_CUS=n: the side streams run on n compute units only (nimg_stream_create_cu_mask) - partitions the chip between the
# parameter-gradient kernels and the launch stream's kernels instead of interleaving them workgroup by workgroup.
SIDE_CUS = int(_os.environ.get('NIMG_SIDE_CUS', '0'))


def _new_side_stream(dev):
    if SIDE_CUS <= 0:
        return torch.cuda.Stream(device=dev)
    import ctypes
    h = ctypes.c_void_p()
    with torch.cuda.device(dev):
        _lib.call('nimg_stream_create_cu_mask', SIDE_CUS, ctypes.byref(h))
    return torch.cuda.ExternalStream(h.value, device=dev)


class _on_side_stream(object):
    def __init__(self, *tensors, key=None):
        self.tensors = [t for t in tensors if t is not None]
        self.k = _side_index(key)

    def __enter__(self):
        dev = self.tensors[0].device
        streams, k = _SIDE['streams'], self.k
        while len(streams) <= k:
            streams.append(None)
            if len(_SIDE['ws']) < len(streams):
                _SIDE['ws'].append(Workspace())
        if streams[k] is None or streams[k].device != dev:
            streams[k] = _new_side_stream(dev)
        side = streams[k]
        self.prev = torch.cuda.current_stream(dev)
        fork = _SIDE.get('fork')
        if fork is not None and fork[1] == self.prev:
            if k not in fork[2]:                # inside `with one_fork()`: the group's single marker on the launch stream,
                side.wait_event(fork[0])        # waited for ONCE per side stream (its later launches are behind that wait)
                fork[2].add(k)
        else:
            side.wait_stream(self.prev)
        for t in self.tensors:
            t.record_stream(side)
        torch.cuda.set_stream(side)             # (torch.cuda.stream(side) does the same behind two more Python layers)
        _SIDE['dirty'].add(k)
        return side

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


# Every fork costs the LAUNCH stream: side.wait_stream(launch) records an event there - a marker packet between two of its
# kernels - and the next kernel starts 6 - 8 us after the previous one ended instead of ~2 (profiles/r06_fork_markers.txt: 25 such
# gaps in the UNet's input-gradient chain, eager launches and graph replay alike).  `with one_fork():` records ONE event; every
# side-stream block entered inside waits on it.  Contract: nothing launched inside may depend on launch-stream work queued after
# the fork - i.e. the body consists of side-stream launches only (parameter gradients deferred by their model, see
# models/pipelines.py UNet.backward).  NIMG_NO_FORK_GROUPS=1: every block forks for itself, as before round 6.
FORK_GROUPS = _os.environ.get('NIMG_NO_FORK_GROUPS') is None


class one_fork(object):
    def __enter__(self):
        if FORK_GROUPS and _SIDE['enabled'] and _SIDE.get('fork') is None and torch.cuda.is_available():
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            _SIDE['fork'] = (ev, cur, set())
            self.mine = True
        else:
            self.mine = False
        return self

    def __exit__(self, *exc):
        if self.mine:
            _SIDE['fork'] = None
        return False


class ParamGroup(object):
    """Parameter-gradient launches of a few consecutive layers, deferred and issued behind ONE fork:
        g = ops.ParamGroup(); g.add(lambda: layer.backward_params(...)); ...input-gradient launches...; g.flush()
    The closures must hold tensors no later launch-stream kernel of the group overwrites."""

    def __init__(self, defer=True):
        self.fns = []
        self.defer = defer

    def add(self, fn):
        if self.defer and FORK_GROUPS and _SIDE['enabled']:
            self.fns.append(fn)
        else:
            fn()

    def flush(self):
        if self.fns:
            with one_fork():
                for fn in self.fns:
                    fn()
            self.fns = []


class side_stream(object):
    """with side_stream(t1, t2, ..., key=buffer address): launches inside run on a side stream (after everything queued so far
    on the launch stream; the tensors are pinned against early reuse), or in place when side streams are disabled.  Tensors
    ALLOCATED inside belong to that stream's pool - keep them local to the block."""

    def __init__(self, *tensors, key=None):
        self.ctx = _on_side_stream(*tensors, key=key) if _SIDE['enabled'] else None

    def __enter__(self):
        return self.ctx.__enter__() if self.ctx is not None else None

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc) if self.ctx is not None else False


# The split-K slab reductions of the weight gradients issued with side=True are DEFERRED: the partial-sum kernels run where they
# were issued, the reductions they owe are collected per side stream and run as ONE batched launch when somebody joins the side
# streams (or the batch is full) - ~30 launches of 4 - 19 us per training step become 2 - 4, with the same sums to the bit
# (csrc/common.h ReduceEntry).  OPT-IN (NIMG_DEFER_REDUCE=1), because it LOSES: UNet backward 1690 -> 1744 us, C4 step 8.06 -> 8.21 ms
# (profiles/r05_deferred_reduce_ab.txt).  A reduction issued right behind its producer finds the ~38 MB of slabs in the 256 MB
# Infinity Cache and overlaps with the next layers; the batched one reads ~1 GB of slabs from HBM at the tail of the side stream,
# on the step's critical path.  The launches were never the cost - VERDICT r04 item 4 is answered by measurement, not by count.
DEFER_REDUCE = _os.environ.get('NIMG_DEFER_REDUCE') is not None
_DEFER = {}


def _defer_state(k):
    st = _DEFER.get(k)
    if st is None:
        import ctypes
        lib = _lib.load()
        nb, cap = int(lib.nimg_reduce_entry_bytes()), int(lib.nimg_reduce_batch_max())
        st = _DEFER[k] = {'buf': (ctypes.c_char * (nb * cap))(), 'n': 0, 'nb': nb, 'cap': cap}
    return st


def _flush_deferred(k, on_its_stream=True):
    """Issue the batched reduction owed on side stream k (on that stream) and hand its scratch arena back."""
    st = _DEFER.get(k)
    if st is None or st['n'] == 0:
        return
    side = _SIDE['streams'][k]
    if on_its_stream and side is not None and torch.cuda.current_stream(side.device) != side:
        prev = torch.cuda.current_stream(side.device)
        torch.cuda.set_stream(side)
        try:
            _lib.call('nimg_reduce_slabs_batch', st['buf'], st['n'], _stream())
        finally:
            torch.cuda.set_stream(prev)
    else:
        _lib.call('nimg_reduce_slabs_batch', st['buf'], st['n'], _stream())
    st['n'] = 0
    _SIDE['ws'][k].release()


# NIMG_REDUCE_STREAM=1: the split-K slab reduction of a side=True weight gradient runs on a stream of its OWN (one per side stream),
# behind an event of the side stream, instead of between two weight gradients on the side stream: the next weight gradient starts
# as soon as its predecessor ends, the reduction (a small grid that waits long for free CUs beside the chip-filling kernels: 28 us
# on average inside a C4 step against 7 us alone) runs beside it.  Every pending weight gradient keeps its own piece of the side
# stream's arena until the join.  Same kernels, same sums.
# NIMG_CHAIN_REDUCE=0|1: the slab reduction of a side=True weight gradient runs in the PROLOGUE of the next weight-gradient kernel of its
# side stream (nimg_conv2d_wgrad_bf16_chained; the last one of a chain as a launch of its own at the join): ~25 launches fewer per
# step on the side queues, which end later than the launch queue in the UNet's backward pass (profiles/r06_fork_markers.txt).
CHAIN_REDUCE = _os.environ.get('NIMG_CHAIN_REDUCE', '0') == '1'
_CHAIN = {}                                   # side stream index -> the entry its last chained launch filled (ctypes buffer)
REDUCE_STREAM = _os.environ.get('NIMG_REDUCE_STREAM', '0') == '1'
_RSTREAM = {'streams': {}, 'dirty': set()}


def _reduce_behind(k, entry, device):
    """Issue the reduction described by `entry` (owed by the launch just made on side stream k, the current stream) on k's
    reduction stream."""
    side = torch.cuda.current_stream(device)
    rs = _RSTREAM['streams'].get(k)
    if rs is None or rs.device != device:
        rs = _RSTREAM['streams'][k] = torch.cuda.Stream(device=device)
    ev = torch.cuda.Event()
    ev.record(side)
    rs.wait_event(ev)
    torch.cuda.set_stream(rs)
    try:
        _lib.call('nimg_reduce_slabs_batch', entry, 1, _stream())
    finally:
        torch.cuda.set_stream(side)
    _RSTREAM['dirty'].add(k)


def _flush_chain(k):
    """The reduction the last chained weight gradient of side stream k still owes: as a launch of its own, on that stream."""
    entry = _CHAIN.pop(k, None)
    if entry is None:
        return
    side = _SIDE['streams'][k]
    prev = torch.cuda.current_stream(side.device)
    torch.cuda.set_stream(side)
    try:
        _lib.call('nimg_reduce_slabs_batch', entry, 1, _stream())
    finally:
        torch.cuda.set_stream(prev)
    _SIDE['ws'][k].release()               # (later users of the arena are ordered behind the join that called this)


def join_side_stream():
    """Make the current stream wait for the parameter-gradient kernels launched on the side streams."""
    for k in sorted(_SIDE['dirty']):
        st = _SIDE['streams'][k]
        if st is not None:
            _flush_deferred(k)
            _flush_chain(k)
            torch.cuda.current_stream(st.device).wait_stream(st)
    _SIDE['dirty'].clear()
    for k in sorted(_RSTREAM['dirty']):
        rs = _RSTREAM['streams'][k]
        torch.cuda.current_stream(rs.device).wait_stream(rs)
        _SIDE['ws'][k].release()           # every later user of the arena is ordered behind this join
    _RSTREAM['dirty'].clear()


# ----------------------------------------------------------------------------------------------------------------
# differentiable JPEG
def qtable(quality, channel):
    out = np.zeros(64, np.float32)
    _lib.call('nimg_jpeg_qtable', int(quality), int(channel), out.ctypes.data)
    return out.reshape(8, 8)


def qtables_device(quality, device):
    q = np.stack([qtable(quality, 0), qtable(quality, 1), qtable(quality, 1)]) if quality is not None \
        else np.ones((3, 8, 8), np.float32)
    return torch.from_numpy(q).to(device)


ROUNDING = {'round': 0, 'soft': 1, 'sin': 2, 'harmonic': 3, 'identity': 4}


def djpeg_fwd(x, qtab, rounding='soft', want_mask=True, want_idx=False, want_xdq=False, out=None):
    _f32(x, qtab, out)
    n, h, w, c = x.shape
    if c != 3:
        raise ValueError('dJPEG expects NHW3 images')
    y = torch.empty_like(x) if out is None else out
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if want_mask else None
    idx = torch.empty((n, 3, h // 8, w // 8, 8, 8), dtype=torch.int16, device=x.device) if want_idx else None
    xdq = torch.empty((n, 3, h // 8, w // 8, 8, 8), dtype=torch.float32, device=x.device) if want_xdq else None
    _lib.call('nimg_djpeg_fwd', _p(x), _p(y), _p(qtab), _p(mask), _p(idx), _p(xdq), n, h, w, ROUNDING[rounding],
              _stream())
    return y, mask, idx, xdq


def djpeg_bwd(x, gy, mask, qtab, rounding='soft', out=None, dq=None, accumulate=False):
    """gx = d loss / d x; with dq (a (2,8,8) or 128-element float32 buffer) also the gradient of trainable quantisation
    tables, [luma, chroma (Cb + Cr)], (+)= when accumulate."""
    _f32(x, gy, qtab, dq)
    _chk(mask)
    n, h, w, _ = x.shape
    gx = torch.empty_like(x) if out is None else out
    if dq is None:
        _lib.call('nimg_djpeg_bwd', _p(x), _p(gy), _p(mask), _p(qtab), _p(gx), n, h, w, ROUNDING[rounding], _stream())
        return gx
    if dq.numel() != 128 or not dq.is_contiguous():
        raise ValueError('dq: 128 contiguous float32 values (2 x 8 x 8) expected')
    need = _lib.load().nimg_djpeg_dq_workspace_bytes(n, h, w)
    ws = _ws_current(x.device).get(need, x.device)
    _lib.call('nimg_djpeg_bwd_dq', _p(x), _p(gy), _p(mask), _p(qtab), _p(gx), _p(dq), n, h, w, ROUNDING[rounding],
              1 if accumulate else 0, _p(ws), need, _stream())
    return gx


# ----------------------------------------------------------------------------------------------------------------
# convolutions
def conv2d(x, w, bias=None, x2=None, stride=1, padding='SAME', act=None, pad_mode=0, out=None, out2=None,
           act_mask=None, pads=None, out_hw=None, _wmode=0, _f32_only=False, mask_alpha=None, out_bf16=False, residual=None,
           bf16_copy=False, d2s_out=False, s2d_out=False, copy_lrelu=False, mask_conv_layout=False):
    """x (N,H,W,C1) [+ x2 (N,H,W,C2)], w (k,k,C1+C2,Cout) HWIO.  padding 'SAME' (TF) | 'VALID' | explicit pads/out_hw.
    d2s_out: the result is returned as its depth_to_space(2) image (N, 2 Hout, 2 Wout, Cout / 4) - written in that layout by
    the 3x3 throughput-mode kernel (act_mask then has that shape too - or, with mask_conv_layout, the convolution's own
    (N, Hout, Wout, Cout) shape: the masking activation was stored as a space-to-depth image), convolution + d2s_clip
    (+ lrelu_bwd) elsewhere.
    s2d_out: the result is returned as its space_to_depth(2) image (N, Hout / 2, Wout / 2, 4 Cout), i.e. the gradient at the
    input of a depth_to_space layer (act_mask / residual keep the (N, Hout, Wout, Cout) layout); 3x3 throughput-mode kernel
    (bf16 if out_bf16), convolution + d2s_clip_bwd elsewhere.
    out/out2: optional pre-allocated outputs (out2 splits the output channels: Cout = out.C + out2.C).
    residual: float32 tensor of the output's shape added to the result (a residual block's skip connection; fused into the
    3x3 throughput-mode kernel, a separate add elsewhere).
    bf16_copy: return (out, copy) with `copy` = the float32 result rounded to bf16 by the same kernel (None where the fused
    kernel does not apply, i.e. in float32 mode): what the bf16 kernels downstream would round it to anyway, at half the
    bytes.  copy_lrelu: the copy holds LeakyReLU(0.2) of the (activation-free) result instead."""
    copy = None
    if mask_conv_layout and not d2s_out:
        raise ValueError('mask_conv_layout only has a meaning with d2s_out')
    if w.shape[0] not in (1, 2, 3, 5):
        # 4x4, 6x6 ... 11x11 (FAN `kernel`, forensics.py:51; demosaicing filters, pipelines.py:242,419): the float32 matrix-core
        # kernels in either compute mode - the throughput-mode kernels are built for the channel's own 1x1 / 2x2 / 3x3 / 5x5 layers
        if w.shape[0] not in BIG_KERNELS or stride != 1:
            raise NotImplementedError('convolution kernel size {} (stride {}) is not built: 1 ... 11 at stride 1'.format(
                w.shape[0], stride))
        _f32_only = True
    if s2d_out:
        co_ = w.shape[3] if _wmode == 0 else w.shape[2]
        fused = COMPUTE == 'bf16' and not _f32_only and x2 is None and out is None and out2 is None and stride == 1 and \
            w.shape[0] == 3 and co_ % 4 == 0 and co_ >= 8 and x.shape[3] % 4 == 0 and x.shape[3] >= 8 and \
            (x.shape[3] % 8 == 0 or (not _is_bf16(x) and residual is None)) and D2S_EPILOGUE and not bf16_copy and \
            not d2s_out
        if not fused:
            y = conv2d(x, w, bias, stride=stride, padding=padding, act=act, pad_mode=pad_mode, act_mask=act_mask, pads=pads,
                       out_hw=out_hw, _wmode=_wmode, _f32_only=_f32_only, mask_alpha=mask_alpha, residual=residual)
            y = d2s_clip_bwd(y, 1.0)
            return y.to(torch.bfloat16) if out_bf16 and COMPUTE == 'bf16' else y      # the caller's storage choice survives
    if d2s_out:
        co_ = w.shape[3] if _wmode == 0 else w.shape[2]
        fused = COMPUTE == 'bf16' and not _f32_only and x2 is None and out is None and out2 is None and stride == 1 and \
            w.shape[0] == 3 and co_ % 16 == 0 and x.shape[3] % 8 == 0 and D2S_EPILOGUE
        if residual is not None:
            raise NotImplementedError('d2s_out with a residual')
        if not fused:
            if mask_conv_layout:
                raise RuntimeError('a space-to-depth-stored mask needs the fused depth-to-space epilogue')
            y = conv2d(x, w, bias, stride=stride, padding=padding, act=act, pad_mode=pad_mode, pads=pads, out_hw=out_hw,
                       _wmode=_wmode, _f32_only=_f32_only)
            y = d2s_clip(y, 1.0, 0.0, False)
            y = y if act_mask is None else lrelu_bwd(y, act_mask, out=y, alpha=mask_alpha)
            if out_bf16 and COMPUTE == 'bf16':
                y = y.to(torch.bfloat16)
            return (y, y.to(torch.bfloat16) if COMPUTE == 'bf16' else None) if bf16_copy else y
    if residual is not None or bf16_copy:
        _f32(residual)
        fused = COMPUTE == 'bf16' and not _f32_only and x2 is None and out2 is None and \
            ((stride == 1 and w.shape[0] == 3) or (residual is None and (w.shape[0], stride) in ((1, 1), (5, 1), (5, 2)))) and \
            (s2d_out or (not out_bf16 and (out is None or out.dtype == torch.float32))) and x.shape[3] % 8 == 0 and \
            not (copy_lrelu and act is not None) and \
            (w.shape[3] if _wmode == 0 else w.shape[2]) % 4 == 0 and (w.shape[3] if _wmode == 0 else w.shape[2]) >= 8
        if not fused:
            y = conv2d(x, w, bias, x2=x2, stride=stride, padding=padding, act=act, pad_mode=pad_mode, out=out, out2=out2,
                       act_mask=act_mask, pads=pads, out_hw=out_hw, _wmode=_wmode, _f32_only=_f32_only, mask_alpha=mask_alpha,
                       out_bf16=out_bf16)
            if residual is not None:
                add(residual, y, out=y)
            return (y, None) if bf16_copy else y
    _f32(w, bias)
    _fb(x, x2, out, out2, act_mask)
    if (x2 is not None and x2.dtype != x.dtype) or (out2 is not None and out is not None and out2.dtype != out.dtype):
        raise RuntimeError('the two halves of a split input / output must be stored alike (both float32 or both bfloat16)')
    n, h, wd, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    ks = w.shape[0]
    cout = w.shape[3] if _wmode == 0 else w.shape[2]         # _wmode 1 = input gradient: (k,k,Cin,Cout) read backwards
    if (w.shape[2] if _wmode == 0 else w.shape[3]) != c1 + c2 or w.shape[1] != ks:
        raise ValueError('weight shape {} does not match input channels {}+{}'.format(tuple(w.shape), c1, c2))
    if pads is not None:
        pt, pl = pads
        ho, wo = out_hw
    elif padding == 'SAME':
        ho, pt = same_pads(h, ks, stride)
        wo, pl = same_pads(wd, ks, stride)
    elif padding == 'VALID':
        pt = pl = 0
        ho, wo = (h - ks) // stride + 1, (wd - ks) // stride + 1
    else:
        raise ValueError(padding)
    if s2d_out and (ho % 2 or wo % 2):
        raise ValueError('s2d_out: even output size expected')
    if out is None:
        shape = (n, 2 * ho, 2 * wo, cout // 4) if d2s_out else ((n, ho // 2, wo // 2, 4 * cout) if s2d_out else (n, ho, wo, cout))
        out = torch.empty(shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    o1 = cout if (d2s_out or s2d_out) else out.shape[3]
    o2 = 0 if out2 is None else out2.shape[3]
    if o1 + o2 != cout or (not (d2s_out or s2d_out) and tuple(out.shape[:3]) != (n, ho, wo)):
        raise ValueError('output shape mismatch')
    if d2s_out and act_mask is not None and \
            tuple(act_mask.shape) != ((n, ho, wo, cout) if mask_conv_layout else tuple(out.shape)):
        raise ValueError('d2s_out: the mask has the shape of the depth-to-space output (or of the convolution, mask_conv_layout)')
    # activation: LeakyReLU(0.2) | ReLU (= slope 0); act_mask multiplies by the same-slope derivative (mask_alpha
    # overrides the slope for an input-gradient pass behind a ReLU layer)
    if act not in (None, 'leaky_relu', 'relu'):
        raise ValueError('unsupported activation {}'.format(act))
    act_id = 0 if act is None else 1
    alpha = 0.0 if act == 'relu' else (LRELU_ALPHA if mask_alpha is None else float(mask_alpha))
    if COMPUTE == 'bf16' and not _f32_only and _wmode == 0 and c2 == 0 and out2 is None and act_mask is None and \
            c1 in (3, 4) and ks in (3, 5) and stride == 1 and cout >= 8 and (ho, wo) == (h, wd) and \
            (pt, pl) == ((ks - 1) // 2, (ks - 1) // 2):
        _f32(x)
        if ROWS_CONV and c1 == 4 and ks == 3 and cout == 32 and wd == 128 and h % 4 == 0 and h >= 4 and pad_mode == 0 and \
                _is_bf16(out) and act in (None, 'leaky_relu') and n * h * wd * 16 < (1 << 31) - 65536:
            # the UNet's first layer at 128-pixel rows: row-streaming form (csrc/conv3_rows.hip), same bits
            _lib.call('nimg_conv3_rows_c4_bf16', _p(x), _p(w), _p(bias), _p(out), n, h, wd, act_id, alpha, _stream())
            return out
        _lib.call('nimg_conv2d_fwd_smallc_bf16_ex', _p(x), c1, _p(w), _p(bias), _p(out), cout, n, h, wd, ks, pad_mode,
                  act_id, alpha, BF16_OUT if _is_bf16(out) else 0, _stream())
        return out
    if COMPUTE == 'bf16' and not _f32_only and _wmode == 1 and c2 == 0 and out2 is None and act_mask is None and \
            c1 == 32 and cout == 3 and ks == 5 and stride == 1 and (ho, wo) == (h, wd) and (pt, pl) == (2, 2) and \
            pad_mode == 0 and bias is None and act is None and not _is_bf16(x) and not _is_bf16(out):
        _lib.call('nimg_conv2d_dgrad_fewin_bf16', _p(x), _p(w), _p(out), 3, 32, n, h, wd, 5, _stream())
        return out
    if COMPUTE == 'bf16' and not _f32_only and residual is None and not bf16_copy and not d2s_out and not s2d_out and \
            rows_conv_ok(x, x2, ks, stride, cout, (ho, wo), (pt, pl), pad_mode, out, out2, act_mask, act, bias):
        # the UNet's level-1 layers: row-band streaming kernel (csrc/conv3_rows.hip), same bits as the tile kernels
        _lib.call('nimg_conv3_rows_bf16', _p(x), c1, _p(x2), c2, _p(weights_bf16(w, _wmode)), _p(bias), _p(act_mask), _p(out),
                  _p(out2), None, n, h, wd, cout, act_id, alpha, 0 if _is_bf16(out) else 1, _stream())
        return out if out2 is None else (out, out2)
    if COMPUTE == 'bf16' and not _f32_only and c2 % 8 == 0 and cout >= 8 and \
            (c1 % 8 == 0 or (c2 == 0 and c1 % 4 == 0 and c1 >= 8 and not _is_bf16(x))):
        wb = weights_bf16(w, _wmode)
        flags = (BF16_IN if _is_bf16(x) else 0) | (BF16_OUT if _is_bf16(out) else 0) | \
            (BF16_MASK if _is_bf16(act_mask) else 0) | (D2S_OUT if d2s_out else 0) | (S2D_OUT if s2d_out else 0) | \
            (MASK_CONV if (mask_conv_layout and d2s_out) else 0)
        if residual is not None or bf16_copy:
            if residual is not None and tuple(residual.shape) != (n, ho, wo, cout):
                raise ValueError('residual: the shape of the output expected')
            if bf16_copy:
                copy = torch.empty(out.shape, dtype=torch.bfloat16, device=out.device)
            _lib.call('nimg_conv2d_fwd_bf16_res', _p(x), c1, _p(wb), _p(bias), _p(out), o1, _p(act_mask), _p(residual), _p(copy),
                      n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, act_id, alpha, flags | (COPY_LRELU if copy_lrelu else 0),
                      _stream())
            return (out, copy) if bf16_copy else out
        _lib.call('nimg_conv2d_fwd_bf16_ex', _p(x), c1, _p(x2), c2, _p(wb), _p(bias), _p(out), o1, _p(out2), o2,
                  _p(act_mask), n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, act_id, alpha, flags, _stream())
        return out if out2 is None else (out, out2)
    if _is_bf16(x) or _is_bf16(out) or _is_bf16(act_mask) or _is_bf16(x2) or _is_bf16(out2):
        raise RuntimeError('bf16-stored tensor reached a float32 convolution path')
    if _wmode == 1:
        w = flip_weights(w)
    _lib.call('nimg_conv2d_fwd', _p(x), c1, _p(x2), c2, _p(w), _p(bias), _p(out), o1, _p(out2), o2, _p(act_mask),
              n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, act_id, alpha, _stream())
    return out if out2 is None else (out, out2)


# Row-band streaming form of the 3x3 convolutions with 32 output channels over 128-pixel-wide bf16-stored images - the UNet's
# level-1 layers at the channel's 128 x 128 RAW patches (NIMG_NO_ROWS_CONV=1: the tile kernels, A/B runs)
ROWS_CONV = _os.environ.get('NIMG_NO_ROWS_CONV') is None
# ... and the 64-pixel-wide, 64-channel second level: OPT-IN (NIMG_ROWS_LEVEL2=1).  Stand-alone the streaming form wins there too
# (32 -> 64: 27 -> 22 us, 64 -> 64: 39 -> 34 us at B = 64) but the UNet step does not move (forward 755 vs 761 us, step 2354 vs 2347 us):
# these layers are no longer byte-bound (73 KB of weights staged per workgroup, one 143 KB workgroup per CU leaves the side
# streams' weight gradients no room) - profiles/r05_unet_rows_ab.txt section 6
ROWS_LEVEL2 = _os.environ.get('NIMG_ROWS_LEVEL2') is not None


def rows_conv_ok(x, x2, ks, stride, cout, out_hw, pads, pad_mode, out, out2, act_mask, act, bias=None):
    n, h, wd, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    if ROWS_CONV and ROWS_LEVEL2 and ks == 3 and stride == 1 and wd == 64 and h % 4 == 0 and h >= 4 and tuple(out_hw) == (h, wd) and \
            tuple(pads) == (1, 1) and pad_mode == 0 and _is_bf16(x) and x2 is None and act in (None, 'leaky_relu') and \
            n * h * wd * max(c1, 64) * 2 < (1 << 31) - 65536:        # 32-bit buffer descriptors, as the C entry checks
        # the UNet's second level: 32 | 64 -> 64 channels at 64-pixel rows, one bf16 tensor out
        return cout == 64 and c1 in (32, 64) and out2 is None and _is_bf16(out) and out.shape[3] == 64 and \
            (act_mask is None or (_is_bf16(act_mask) and tuple(act_mask.shape) == (n, h, wd, 64)))
    if not (ROWS_CONV and ks == 3 and stride == 1 and wd == 128 and h % 4 == 0 and h >= 4 and tuple(out_hw) == (h, wd) and
            tuple(pads) == (1, 1) and pad_mode == 0 and _is_bf16(x) and (x2 is None or _is_bf16(x2)) and
            act in (None, 'leaky_relu') and n * h * wd * max(c1, c2) * 2 < (1 << 31) - 65536):
        return False
    if cout == 64:      # an input gradient that leaves as two 32-channel tensors (a decoder layer's [up, skip])
        return (c1, c2) == (32, 0) and out2 is not None and _is_bf16(out) and _is_bf16(out2) and out.shape[3] == 32 and \
            out2.shape[3] == 32 and act_mask is None and bias is None and act is None
    return cout == 32 and out2 is None and (c1, c2) in ((32, 0), (64, 0), (32, 32)) and \
        (act_mask is None or (_is_bf16(act_mask) and tuple(act_mask.shape) == (n, h, wd, 32)))      # out: bf16 or float32


def rows_d2s_ok(x, w):
    n, h, wd, c = x.shape
    return ROWS_CONV and COMPUTE == 'bf16' and _is_bf16(x) and tuple(w.shape) == (3, 3, 32, 12) and c == 32 and wd == 128 and \
        h % 4 == 0 and h >= 4 and n * h * wd * 64 < (1 << 31) - 65536


def conv3_rows_d2s(x, w, bias, out=None):
    """clip(depth_to_space(conv2d(x, w, bias), 2), 0, 1) in one pass (the UNet's last layer, where rows_d2s_ok): x (n, h, 128, 32)
    bf16 -> (n, 2 h, 256, 3) float32; the same bits as conv2d + d2s_clip."""
    if not rows_d2s_ok(x, w):
        raise ValueError('conv3_rows_d2s: unsupported shape / mode')
    _f32(w, bias, out)
    n, h, wd, _ = x.shape
    if out is not None and (tuple(out.shape) != (n, 2 * h, 2 * wd, 3) or not out.is_contiguous()):
        raise ValueError('conv3_rows_d2s: output shape mismatch')
    y = torch.empty((n, 2 * h, 2 * wd, 3), dtype=torch.float32, device=x.device) if out is None else out
    _lib.call('nimg_conv3_rows_d2s_bf16', _p(x), 32, _p(weights_bf16(w, 0)), _p(bias), _p(y), n, h, wd, _stream())
    return y


def flip_weights(w, out=None):
    """(k,k,Cin,Cout) -> spatially flipped (k,k,Cout,Cin) for the input-gradient pass."""
    _f32(w, out)
    kh, kw, cin, cout = w.shape
    wt = torch.empty((kh, kw, cout, cin), dtype=torch.float32, device=w.device) if out is None else out
    _lib.call('nimg_conv_flip_weights', _p(w), _p(wt), kh, kw, cin, cout, _stream())
    return wt


def conv2d_dgrad(dz, w, in_hw, stride=1, padding='SAME', act_mask=None, out=None, out2=None, mask_alpha=None,
                 out_bf16=False, residual=None, bf16_copy=False, d2s_out=False, s2d_out=False, mask_conv_layout=False):
    """Input gradient of conv2d (stride 1, odd kernel): correlation of dz with the flipped kernel."""
    if stride != 1:
        raise NotImplementedError('strided dgrad is expressed by the caller (see models/compression.py)')
    ks = w.shape[0]
    h, wd = in_hw
    if padding == 'SAME':
        _, pt = same_pads(h, ks, 1)
        _, pl = same_pads(wd, ks, 1)
    else:
        pt = pl = 0
    # forward used pad (pt, pl); the gradient correlation needs ks-1-pt / ks-1-pl; the kernel is read flipped/transposed
    return conv2d(dz, w, None, pads=(ks - 1 - pt, ks - 1 - pl), out_hw=(h, wd), act_mask=act_mask, out=out,
                  out2=out2, _wmode=1, mask_alpha=mask_alpha, out_bf16=out_bf16, residual=residual, bf16_copy=bf16_copy,
                  d2s_out=d2s_out, s2d_out=s2d_out, mask_conv_layout=mask_conv_layout)


def conv2d_wgrad(x, dz, ks, x2=None, stride=1, padding='SAME', pad_mode=0, pads=None, dw=None, accumulate=False,
                 db=None, side=False, _defer_k=None):
    """dw (k,k,C1+C2,Cout) = sum over pixels of x (x) dz;  db (optional, Cout) = fused bias gradient.
    side=True: launch on the side stream (see _on_side_stream); dw / db must then be persistent buffers."""
    if side and _SIDE['enabled'] and dw is not None:
        ctx = _on_side_stream(x, dz, x2, key=dw.data_ptr())
        with ctx:
            return conv2d_wgrad(x, dz, ks, x2=x2, stride=stride, padding=padding, pad_mode=pad_mode, pads=pads, dw=dw,
                                accumulate=accumulate, db=db, side=False, _defer_k=ctx.k)
    _f32(dw, db)
    _fb(x, x2, dz)
    if x2 is not None and x2.dtype != x.dtype:
        raise RuntimeError('the two halves of a split input must be stored alike (both float32 or both bfloat16)')
    n, h, wd, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[3]
    _, ho, wo, cout = dz.shape
    if pads is not None:
        pt, pl = pads
    elif padding == 'SAME':
        _, pt = same_pads(h, ks, stride)
        _, pl = same_pads(wd, ks, stride)
    else:
        pt = pl = 0
    if dw is None:
        dw = torch.empty((ks, ks, c1 + c2, cout), dtype=torch.float32, device=x.device)
    packed_ok = c2 == 0 and c1 in (3, 4) and stride == 1 and ks in (3, 5)
    if ks not in (1, 2, 3, 5) and (ks not in BIG_KERNELS or stride != 1):
        raise NotImplementedError('weight gradient for kernel size {} (stride {}) is not built'.format(ks, stride))
    if COMPUTE == 'bf16' and ks in (1, 2, 3, 5) and (packed_ok or (c1 % 4 == 0 and c2 % 4 == 0 and cout % 4 == 0 and c1 + c2 >= 8 and
                                            (c2 == 0 or c1 % 8 == 0))):
        need = _lib.load().nimg_conv2d_wgrad_bf16_workspace_bytes(c1 + c2, cout, ks, ks, n, ho, wo)
        flags = (BF16_IN if _is_bf16(x) else 0) | (BF16_DZ if _is_bf16(dz) else 0)
        if _defer_k is not None and DEFER_REDUCE and not accumulate:
            _wgrad_deferred(_defer_k, x, c1, x2, c2, dz, None, cout, dw, db, n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, need, flags)
            return dw
        if _defer_k is not None and REDUCE_STREAM and not accumulate:
            _wgrad_reduce_stream(_defer_k, x, c1, x2, c2, dz, None, cout, dw, db, n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, need, flags)
            return dw
        if _defer_k is not None and CHAIN_REDUCE and not accumulate:
            _wgrad_chained(_defer_k, x, c1, x2, c2, dz, None, cout, dw, db, n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, need, flags)
            return dw
        if _defer_k is not None and CHAIN_REDUCE:
            _flush_chain_here(_defer_k)        # an accumulating launch follows the chain of its stream
        if _defer_k is not None:
            _flush_deferred(_defer_k, on_its_stream=False)      # an accumulating launch follows the pending ones of its stream
        ws = _ws_current(x.device).get(need, x.device)
        _lib.call('nimg_conv2d_wgrad_bf16_ex', _p(x), c1, _p(x2), c2, _p(dz), cout, _p(dw), _p(db), n, h, wd, ks, stride,
                  pt, pl, pad_mode, ho, wo, 1 if accumulate else 0, _p(ws), ws.numel(), flags, _stream())
        return dw
    if _is_bf16(x) or _is_bf16(dz):
        raise RuntimeError('bf16-stored tensor reached a float32 weight-gradient path')
    need = _lib.load().nimg_conv2d_wgrad_workspace_bytes(c1 + c2, cout, ks, ks, n, ho, wo)
    ws = _ws_current(x.device).get(need, x.device)
    _lib.call('nimg_conv2d_wgrad', _p(x), c1, _p(x2), c2, _p(dz), cout, _p(dw), _p(db), n, h, wd, ks, stride, pt, pl,
              pad_mode, ho, wo, 1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return dw


def _wgrad_deferred(k, x, c1, x2, c2, dz, idx, cout, dw, db, n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, need, flags):
    """One weight gradient whose slab reduction joins side stream k's pending batch (the caller is ON that stream)."""
    import ctypes
    st = _defer_state(k)
    if st['n'] == st['cap']:
        _flush_deferred(k, on_its_stream=False)
    ws = _SIDE['ws'][k].claim(need, x.device)
    entry = ctypes.byref(st['buf'], st['n'] * st['nb'])
    _lib.call('nimg_conv2d_wgrad_bf16_deferred', _p(x), c1, _p(x2), c2, _p(dz), _p(idx), cout, _p(dw), _p(db), n, h, wd, ks, stride,
              pt, pl, pad_mode, ho, wo, _p(ws), ws.numel(), flags, entry, _stream())
    st['n'] += 1


def _flush_chain_here(k):
    entry = _CHAIN.pop(k, None)
    if entry is not None:
        _lib.call('nimg_reduce_slabs_batch', entry, 1, _stream())


def _wgrad_chained(k, x, c1, x2, c2, dz, idx, cout, dw, db, n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, need, flags):
    """One weight gradient on side stream k (the current stream): it runs the reduction its predecessor on k owes and leaves its
    own to its successor (or to the join)."""
    import ctypes
    nb = int(_lib.load().nimg_reduce_entry_bytes())
    pre = _CHAIN.get(k)
    entry = (ctypes.c_char * nb)()
    ws = _SIDE['ws'][k].claim(need, x.device)
    _lib.call('nimg_conv2d_wgrad_bf16_chained', _p(x), c1, _p(x2), c2, _p(dz), _p(idx), cout, _p(dw), _p(db), n, h, wd, ks, stride,
              pt, pl, pad_mode, ho, wo, _p(ws), ws.numel(), flags, pre, entry, _stream())
    _CHAIN[k] = entry


def _wgrad_reduce_stream(k, x, c1, x2, c2, dz, idx, cout, dw, db, n, h, wd, ks, stride, pt, pl, pad_mode, ho, wo, need, flags):
    """One weight gradient on side stream k (the current stream) whose slab reduction goes to k's reduction stream."""
    import ctypes
    nb = int(_lib.load().nimg_reduce_entry_bytes())
    entry = (ctypes.c_char * nb)()
    ws = _SIDE['ws'][k].claim(need, x.device)
    _lib.call('nimg_conv2d_wgrad_bf16_deferred', _p(x), c1, _p(x2), c2, _p(dz), _p(idx), cout, _p(dw), _p(db), n, h, wd, ks, stride,
              pt, pl, pad_mode, ho, wo, _p(ws), ws.numel(), flags, entry, _stream())
    _reduce_behind(k, entry, x.device)


def bias_grad(dz, db=None, accumulate=False, side=False):
    if side and _SIDE['enabled'] and db is not None:
        with _on_side_stream(dz, key=db.data_ptr()):
            return bias_grad(dz, db=db, accumulate=accumulate, side=False)
    _f32(db)
    _fb(dz)
    cout = dz.shape[-1]
    npix = dz.numel() // cout
    if db is None:
        db = torch.empty((cout,), dtype=torch.float32, device=dz.device)
    need = _lib.load().nimg_bias_grad_workspace_bytes(npix, cout)
    ws = _ws_current(dz.device).get(need, dz.device)
    _lib.call('nimg_bias_grad_ex', _p(dz), _p(db), npix, cout, 1 if accumulate else 0, _p(ws), ws.numel(),
              BF16_DZ if _is_bf16(dz) else 0, _stream())
    return db


def convt2x2(x, w, bias, out_bf16=False):
    """Conv2DTranspose(k=2,s=2); w (2,2,Cout,Cin).  Throughput mode: x may be stored as bf16, y can be (out_bf16)."""
    _f32(w, bias)
    _fb(x)
    n, h, wd, cin = x.shape
    cout = w.shape[2]
    y = torch.empty((n, 2 * h, 2 * wd, cout), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    if COMPUTE == 'bf16' and cin % 8 == 0 and cout >= 8:
        wb = weights_bf16(w, 1)                # (2,2,Cout,Cin) read as HWIO with the channel roles swapped
        _lib.call('nimg_convt2x2_fwd_bf16_ex', _p(x), _p(wb), _p(bias), _p(y), n, h, wd, cin, cout,
                  (BF16_IN if _is_bf16(x) else 0) | (BF16_OUT if out_bf16 else 0), _stream())
        return y
    if _is_bf16(x) or out_bf16:
        raise RuntimeError('bf16-stored tensor reached a float32 transposed-convolution path')
    _lib.call('nimg_convt2x2_fwd', _p(x), _p(w), _p(bias), _p(y), n, h, wd, cin, cout, _stream())
    return y


def convt2x2_dgrad(dy, w, act_mask=None, out_bf16=False):
    """d input of Conv2DTranspose = strided 2x2 conv of dy with the same kernel viewed as (2,2,Cin'=Cout,Cout'=Cin)."""
    n, h2, w2, _ = dy.shape
    return conv2d(dy, w, None, stride=2, pads=(0, 0), out_hw=(h2 // 2, w2 // 2), act_mask=act_mask, out_bf16=out_bf16)


def convt2x2_wgrad(x, dy, dw=None, side=False):
    """dw (2,2,Cout,Cin) = wgrad with the roles swapped: 'input' = dy (stride 2), 'output gradient' = x."""
    return conv2d_wgrad(dy, x, 2, stride=2, pads=(0, 0), dw=dw, side=side)


# ----------------------------------------------------------------------------------------------------------------
# pooling / layout / element-wise
def maxpool2(x):
    """MaxPool2D(2); a bf16-stored tensor (even h, w; c % 8 == 0) is pooled as bf16."""
    _fb(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    if _is_bf16(x):
        _lib.call('nimg_maxpool2_fwd_bf16', _p(x), _p(y), n, h, w, c, _stream())
        return y
    _lib.call('nimg_maxpool2_fwd', _p(x), _p(y), n, h, w, c, _stream())
    return y


def maxpool2_bwd(dp, yact, add=None, apply_mask=True, out=None):
    _fb(dp, yact, add, out)
    n, h, w, c = yact.shape
    dz = torch.empty_like(yact) if out is None else out
    if _is_bf16(yact):
        if any(t is not None and not _is_bf16(t) for t in (dp, add, dz)):
            raise RuntimeError('maxpool2_bwd on a bf16-stored activation needs bf16-stored gradients')
        _lib.call('nimg_maxpool2_bwd_bf16', _p(dp), _p(yact), _p(add), _p(dz), n, h, w, c, 1 if apply_mask else 0,
                  LRELU_ALPHA, _stream())
        return dz
    _f32(dp, add, dz)
    _lib.call('nimg_maxpool2_bwd', _p(dp), _p(yact), _p(add), _p(dz), n, h, w, c, 1 if apply_mask else 0,
              LRELU_ALPHA, _stream())
    return dz


def conv2d_pool(x, w, bias=None, act='leaky_relu', want_idx=True, out_bf16=False):
    """Conv2D(SAME, stride 1) -> [LeakyReLU] -> MaxPool2D(2) in one pass (the FAN feature extractor): returns the pooled
    activation and the arg-max bytes; the full-resolution activation is never written.  Throughput mode: x may be stored as
    bf16 and the pooled activation can be (out_bf16) - its consumers round it to bf16 anyway."""
    _f32(w, bias)
    _fb(x)
    n, h, wd, cin = x.shape
    ks, cout = w.shape[0], w.shape[3]
    if w.shape[2] != cin or (h & 1) or (wd & 1) or (cout & 3) or ks not in (3, 5):
        raise ValueError('conv2d_pool: unsupported shape')
    bf16_path = COMPUTE == 'bf16' and (cin in (3, 4) or cin % 8 == 0)
    if (out_bf16 or _is_bf16(x)) and not bf16_path:
        raise RuntimeError('bf16 storage needs the throughput-mode convolution path')
    pooled = torch.empty((n, h // 2, wd // 2, cout), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
    idx = torch.empty((n, h // 2, wd // 2, cout), dtype=torch.uint8, device=x.device) if want_idx else None
    a = 1 if act == 'leaky_relu' else 0
    if bf16_path:
        wb = None if cin in (3, 4) else weights_bf16(w, 0)
        flags = (BF16_IN if _is_bf16(x) else 0) | (BF16_OUT if out_bf16 else 0)
        _lib.call('nimg_conv2d_pool_fwd_bf16_ex', _p(x), cin, _p(w), _p(wb), _p(bias), _p(pooled), _p(idx), cout, n, h, wd,
                  ks, a, LRELU_ALPHA, flags, _stream())
    else:
        _lib.call('nimg_conv2d_pool_fwd', _p(x), cin, _p(w), _p(bias), _p(pooled), _p(idx), cout, n, h, wd, ks, a,
                  LRELU_ALPHA, _stream())
    return pooled, idx


# The UNet's encoder levels write the pooled tensor from the second convolution's epilogue (NIMG_NO_POOL_ALSO=1: a separate
# max-pool pass over the stored activation, A/B runs).
POOL_ALSO = _os.environ.get('NIMG_NO_POOL_ALSO') is None


_EPI8 = {}


def _epi8_built(entry):
    """Does the bound library carry the 8-wide epilogue branch `entry` needs?  An A/B build with -DNIMG_NO_EPI8 answers an empty
    call (n = 0) with NIMG_ERR_ARG instead of NIMG_OK, and the callers fall back to the two-pass forms."""
    if entry not in _EPI8:
        lib = _lib.load()
        if entry == 'nimg_conv2d_fwd_pool_also_bf16':
            rc = lib.nimg_conv2d_fwd_pool_also_bf16(None, 8, None, None, None, None, None, 8, 0, 16, 16, 1, LRELU_ALPHA, None)
        else:
            rc = lib.nimg_conv2d_dgrad_unpool_out_bf16(None, 8, None, None, None, None, 8, 0, 16, 16, 1, LRELU_ALPHA, None)
        _EPI8[entry] = rc == 0
    return _EPI8[entry]


def conv2d_and_pool_ok(x, w):
    n, h, wd, cin = x.shape
    return POOL_ALSO and _epi8_built('nimg_conv2d_fwd_pool_also_bf16') and COMPUTE == 'bf16' and _is_bf16(x) and w.shape[0] == 3 and w.shape[1] == 3 and cin % 8 == 0 and \
        w.shape[3] % 8 == 0 and h % 2 == 0 and wd % 2 == 0 and h > 8 and wd > 8


def conv2d_and_pool(x, w, bias=None, act='leaky_relu'):
    """3x3 SAME Conv2D -> [LeakyReLU] stored as bf16 AND its MaxPool2D(2), one pass (where conv2d_and_pool_ok): returns
    (activation, pooled activation) - bit-identical to conv2d(..., out_bf16=True) followed by maxpool2."""
    _f32(w, bias)
    if not conv2d_and_pool_ok(x, w):
        raise ValueError('conv2d_and_pool: unsupported shape / mode')
    n, h, wd, cin = x.shape
    cout = w.shape[3]
    y = torch.empty((n, h, wd, cout), dtype=torch.bfloat16, device=x.device)
    pooled = torch.empty((n, h // 2, wd // 2, cout), dtype=torch.bfloat16, device=x.device)
    if wd == 128 and rows_conv_ok(x, None, 3, 1, cout, (h, wd), (1, 1), 0, y, None, None, act):       # (the pooled tensor: level 1 only)
        _lib.call('nimg_conv3_rows_bf16', _p(x), cin, None, 0, _p(weights_bf16(w, 0)), _p(bias), None, _p(y), None, _p(pooled), n, h,
                  wd, cout, 1 if act == 'leaky_relu' else 0, LRELU_ALPHA, 0, _stream())
        return y, pooled
    _lib.call('nimg_conv2d_fwd_pool_also_bf16', _p(x), cin, _p(weights_bf16(w, 0)), _p(bias), _p(y), _p(pooled), None, cout,
              n, h, wd, 1 if act == 'leaky_relu' else 0, LRELU_ALPHA, _stream())
    return y, pooled


# ... and their backward: the input gradient of the NEXT level's first convolution written through the max-pool
# (NIMG_NO_DGRAD_UNPOOL_OUT=1: input gradient, then maxpool2_bwd; A/B runs)
DGRAD_UNPOOL_OUT = _os.environ.get('NIMG_NO_DGRAD_UNPOOL_OUT') is None


def conv2d_dgrad_unpool_out_ok(dz, w, act, skip):
    return DGRAD_UNPOOL_OUT and _epi8_built('nimg_conv2d_dgrad_unpool_out_bf16') and COMPUTE == 'bf16' and _is_bf16(dz) and _is_bf16(act) and (skip is None or _is_bf16(skip)) and \
        w.shape[0] == 3 and w.shape[1] == 3 and w.shape[2] % 8 == 0 and w.shape[3] % 8 == 0 and \
        act.shape[1] == 2 * dz.shape[1] and act.shape[2] == 2 * dz.shape[2] and act.shape[3] == w.shape[2]


def conv2d_dgrad_unpool_out(dz, w, act, skip=None, apply_mask=True, out=None):
    """maxpool2_bwd(conv2d_dgrad(dz, w) stored as bf16, act, add=skip, apply_mask) in one pass: the gradient at the INPUT of the
    2x2 max-pool that fed a 3x3 SAME convolution with kernel w (k,k,cin,cout); dz (n,h,w,cout), act / skip / result
    (n,2h,2w,cin), all bf16; out may be skip.  Same bits as the two passes."""
    if not conv2d_dgrad_unpool_out_ok(dz, w, act, skip):
        raise ValueError('conv2d_dgrad_unpool_out: unsupported shape / mode')
    _f32(w)
    n, h, wd, cz = dz.shape
    cin = w.shape[2]
    if out is None:
        out = torch.empty_like(act)
    _lib.call('nimg_conv2d_dgrad_unpool_out_bf16', _p(dz), cz, _p(weights_bf16(w, 1)), _p(act), _p(skip), _p(out), cin, n, h, wd,
              1 if apply_mask else 0, LRELU_ALPHA, _stream())
    return out


def maxpool2_unpool(dp, idx, pooled, apply_mask=True, out=None, out_bf16=False):
    """Backward of conv2d_pool's epilogue: the pre-activation gradient at full resolution (optionally stored as bf16:
    its consumers - the bf16 weight / input gradient kernels - round it to bf16 anyway)."""
    _f32(pooled)
    _fb(dp, out)
    _chk(idx)
    n, ho, wo, c = dp.shape
    if out is None:
        out = torch.empty((n, 2 * ho, 2 * wo, c), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dp.device)
    flags = (BF16_IN if _is_bf16(dp) else 0) | (BF16_OUT if _is_bf16(out) else 0)
    _lib.call('nimg_maxpool2_unpool_ex', _p(dp), _p(idx), _p(pooled), _p(out), n, ho, wo, c, 1 if apply_mask else 0,
              LRELU_ALPHA, flags, _stream())
    return out


# The FAN's conv2..4 backward reads the pooled gradient directly (un-pooled while staging); NIMG_NO_UNPOOL_FOLD=1 materialises
# the full-resolution gradient first (A/B runs).
UNPOOL_FOLD = _os.environ.get('NIMG_NO_UNPOOL_FOLD') is None


def unpool_fold_ok(x, g, cin, cout, ks):
    """True when the 5x5 backward kernels can take (pooled gradient, arg-max bytes) instead of the un-pooled gradient."""
    # the C side (nimg_conv2d_fwd_bf16_unpool / nimg_conv2d_wgrad_bf16_unpool) needs its buffer-load variants: descriptors
    # address < 2 GB, the input-gradient pass writes cin channels in 32-wide tiles, NIMG_NO_BUFFER_LOADS leaves them out
    small = max(x.numel(), g.numel()) * 2 < (1 << 31) - 65536 and 25 * cin * cout * 2 < (1 << 31) - 65536
    return UNPOOL_FOLD and COMPUTE == 'bf16' and ks == 5 and _is_bf16(x) and _is_bf16(g) and cin % 32 == 0 and \
        cout % 64 == 0 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and small and \
        _os.environ.get('NIMG_NO_BUFFER_LOADS') is None


def conv2d_dgrad_unpool(g, idx, w, act_mask=None, out_bf16=False):
    """Input gradient of a fused 5x5 conv + pool layer from the pooled gradient g (N,H/2,W/2,Cout) bf16 + arg-max bytes."""
    _f32(w)
    _fb(g, act_mask)
    _chk(idx)
    n, hp, wp, cz = g.shape
    ks, ci = w.shape[0], w.shape[2]
    out = torch.empty((n, 2 * hp, 2 * wp, ci), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=g.device)
    flags = (BF16_OUT if out_bf16 else 0) | (BF16_MASK if _is_bf16(act_mask) else 0)
    if SPARSE_DGRAD and ks == 5 and ci % 32 == 0 and cz % 8 == 0 and _is_bf16(g) and g.numel() * 2 < (1 << 31) - 65536:
        # the pooled gradient as the compressed operand of the structured-sparsity matrix instruction (csrc/dgrad5s.hip)
        img = torch.empty(int(_lib.load().nimg_conv5_dgrad_sparse_image_bytes(ci, cz)), dtype=torch.uint8, device=g.device)
        _lib.call('nimg_conv5_dgrad_sparse_weights', _p(w), _p(img), ci, cz, _stream())
        _lib.call('nimg_conv5_dgrad_sparse', _p(g), _p(idx), cz, _p(img), _p(out), ci, _p(act_mask), n, 2 * hp, 2 * wp, LRELU_ALPHA,
                  flags, _stream())
        return out
    wb = weights_bf16(w, 1)
    _lib.call('nimg_conv2d_fwd_bf16_unpool', _p(g), _p(idx), cz, _p(wb), None, _p(out), ci, _p(act_mask), n, 2 * hp, 2 * wp, ks,
              ks - 1 - (ks - 1) // 2, ks - 1 - (ks - 1) // 2, 2 * hp, 2 * wp, 0, LRELU_ALPHA, flags, _stream())
    return out


def conv2d_wgrad_unpool(x, g, idx, ks, dw, db=None, side=False, _defer_k=None):
    """Weight / bias gradient of a fused 5x5 conv + pool layer from its bf16 input x and the pooled gradient + arg-max bytes."""
    if side and _SIDE['enabled'] and dw is not None:
        ctx = _on_side_stream(x, g, idx, key=dw.data_ptr())
        with ctx:
            return conv2d_wgrad_unpool(x, g, idx, ks, dw, db=db, side=False, _defer_k=ctx.k)
    _f32(dw, db)
    _fb(x, g)
    _chk(idx)
    n, h, wd, cin = x.shape
    cout = g.shape[3]
    need = _lib.load().nimg_conv2d_wgrad_bf16_workspace_bytes(cin, cout, ks, ks, n, h, wd)
    if _defer_k is not None and DEFER_REDUCE and ks == 5 and h % 2 == 0 and wd % 2 == 0:
        _wgrad_deferred(_defer_k, x, cin, None, 0, g, idx, cout, dw, db, n, h, wd, ks, 1, 2, 2, 0, h, wd, need, BF16_IN | BF16_DZ)
        return dw
    if _defer_k is not None and CHAIN_REDUCE and ks == 5 and h % 2 == 0 and wd % 2 == 0:
        _wgrad_chained(_defer_k, x, cin, None, 0, g, idx, cout, dw, db, n, h, wd, ks, 1, 2, 2, 0, h, wd, need, BF16_IN | BF16_DZ)
        return dw
    if _defer_k is not None and REDUCE_STREAM and ks == 5 and h % 2 == 0 and wd % 2 == 0:
        _wgrad_reduce_stream(_defer_k, x, cin, None, 0, g, idx, cout, dw, db, n, h, wd, ks, 1, 2, 2, 0, h, wd, need, BF16_IN | BF16_DZ)
        return dw
    ws = _ws_current(x.device).get(need, x.device)
    _lib.call('nimg_conv2d_wgrad_bf16_unpool', _p(x), cin, _p(g), _p(idx), cout, _p(dw), _p(db), n, h, wd, ks, 0, _p(ws),
              ws.numel(), _stream())
    return dw


def pooled_backward_ok(cin, cout, ks):
    """True when the backward of a fused conv+pool layer can consume the pooled gradient directly (no un-pooling
    pass): throughput mode, the FAN conv1 shape class."""
    return COMPUTE == 'bf16' and cin == 3 and cout == 32 and ks == 5


def conv2d_wgrad_pooled(x, g, idx, ks, dw=None, db=None, side=False):
    """Weight / bias gradient of conv2d_pool from the pooled gradient g (already x LeakyReLU') and the 2-bit arg-max codes."""
    if side and _SIDE['enabled'] and dw is not None:
        with _on_side_stream(x, g, idx, key=dw.data_ptr()):
            return conv2d_wgrad_pooled(x, g, idx, ks, dw=dw, db=db, side=False)
    _f32(x, dw, db)
    _fb(g)
    _chk(idx)
    n, h, wd, cin = x.shape
    cout = g.shape[3]
    if dw is None:
        dw = torch.empty((ks, ks, cin, cout), dtype=torch.float32, device=x.device)
    need = _lib.load().nimg_conv2d_wgrad_bf16_workspace_bytes(cin, cout, ks, ks, n, h, wd)
    ws = _ws_current(x.device).get(need, x.device)
    _lib.call('nimg_conv2d_wgrad_pooled_bf16_ex', _p(x), cin, _p(g), _p(idx), cout, _p(dw), _p(db), n, h, wd, ks, 0, _p(ws),
              ws.numel(), BF16_DZ if _is_bf16(g) else 0, _stream())
    return dw


def conv2d_dgrad_pooled(g, idx, w, out=None):
    """Input gradient of conv2d_pool (3 <- 32 channels, 5x5) from the pooled gradient and the arg-max bytes."""
    _f32(w, out)
    _fb(g)
    _chk(idx)
    n, hp, wp, cz = g.shape
    ks, ci = w.shape[0], w.shape[2]
    if out is None:
        out = torch.empty((n, 2 * hp, 2 * wp, ci), dtype=torch.float32, device=g.device)
    _lib.call('nimg_conv2d_dgrad_fewin_pooled_bf16_ex', _p(g), _p(idx), _p(w), _p(out), ci, cz, n, 2 * hp, 2 * wp, ks,
              BF16_DZ if _is_bf16(g) else 0, _stream())
    return out


def d2s_clip(x, scale=1.0, shift=0.0, clip=True, out=None):
    _f32(x, out)
    n, h, w, c4 = x.shape
    if out is not None and tuple(out.shape) != (n, 2 * h, 2 * w, c4 // 4):
        raise ValueError('d2s_clip: output shape mismatch')
    y = torch.empty((n, 2 * h, 2 * w, c4 // 4), dtype=torch.float32, device=x.device) if out is None else out
    _lib.call('nimg_d2s_clip_fwd', _p(x), _p(y), n, h, w, c4 // 4, float(scale), float(shift), 1 if clip else 0,
              _stream())
    return y


def d2s_clip_bwd(dy, scale=1.0):
    _f32(dy)
    n, h2, w2, c = dy.shape
    dx = torch.empty((n, h2 // 2, w2 // 2, 4 * c), dtype=torch.float32, device=dy.device)
    _lib.call('nimg_d2s_clip_bwd', _p(dy), _p(dx), n, h2 // 2, w2 // 2, c, float(scale), _stream())
    return dx


def lrelu_bwd(dy, yact, out=None, alpha=None):
    """dz = dy * act'(y) for LeakyReLU(alpha) (default 0.2); alpha = 0 is the ReLU derivative."""
    _f32(dy, yact, out)
    dz = torch.empty_like(dy) if out is None else out
    _lib.call('nimg_lrelu_bwd', _p(dy), _p(yact), _p(dz), dy.numel(), LRELU_ALPHA if alpha is None else float(alpha),
              _stream())
    return dz


def add(a, b, out=None):
    _f32(a, b, out)
    o = torch.empty_like(a) if out is None else out
    _lib.call('nimg_add', _p(a), _p(b), _p(o), a.numel(), _stream())
    return o


def add_n(tensors, out=None):
    """out = sum of 2..6 same-sized float32 tensors in one pass (out may be one of them)."""
    import ctypes
    _f32(*tensors)
    _f32(out)
    n = tensors[0].numel()
    if any(t.numel() != n for t in tensors) or (out is not None and out.numel() != n):
        raise ValueError('add_n: the tensors differ in size')
    if len(tensors) == 1:
        return tensors[0] if out is None else out.copy_(tensors[0])
    unaligned = any(t.data_ptr() % 16 for t in tensors) or (out is not None and out.data_ptr() % 16)
    if len(tensors) > 6 or (n & 3) or unaligned:          # pairwise chain (the one-pass kernel moves 16-byte vectors)
        o = add(tensors[0], tensors[1], out=out)           # starts from tensors[0] whatever `out` holds
        for t in tensors[2:]:
            o = add(o, t, out=o)
        return o
    o = torch.empty_like(tensors[0]) if out is None else out
    ptrs = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    _lib.call('nimg_add_n', ptrs, len(tensors), _p(o), n, _stream())
    return o


def avgpool(x, f):
    _f32(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h // f, w // f, c), dtype=torch.float32, device=x.device)
    _lib.call('nimg_avgpool_fwd', _p(x), _p(y), n, h, w, c, f, _stream())
    return y


def avgpool_bwd(dy, f):
    _f32(dy)
    n, ho, wo, c = dy.shape
    dx = torch.empty((n, ho * f, wo * f, c), dtype=torch.float32, device=dy.device)
    _lib.call('nimg_avgpool_bwd', _p(dy), _p(dx), n, ho * f, wo * f, c, f, _stream())
    return dx


# ----------------------------------------------------------------------------------------------------------------
# losses / head / optimiser
def mse255(a, b, grad_scale=None, grad_out=None, accumulate=False):
    """Returns (loss[1], grad wrt a or None).  grad = grad_scale * d mse255 / d a."""
    _f32(a, b, grad_out)
    loss = torch.empty((1,), dtype=torch.float32, device=a.device)
    g = None
    if grad_scale is not None:
        g = torch.empty_like(a) if grad_out is None else grad_out
    need = _lib.load().nimg_mse255_workspace_bytes()
    ws = _ws.get(need, a.device)
    _lib.call('nimg_mse255', _p(a), _p(b), _p(loss), _p(g), a.numel(), float(grad_scale or 0.0),
              1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return loss, g


# The workflow hands the UNet its output gradient as (manipulation gradients..., L2 term): one pass that sums them and writes
# the gradient behind depth_to_space + clip (NIMG_NO_FUSED_HEAD_GRAD=1: the three separate passes, A/B runs).
FUSED_HEAD_GRAD = _os.environ.get('NIMG_NO_FUSED_HEAD_GRAD') is None


def mse255_sum_s2d3(parts, y, target, grad_scale):
    """(mse255(y, target)[1], space_to_depth(sum(parts) + grad_scale * d mse255 / d y)) - parts, y, target (n,2h,2w,3) float32;
    bit-identical to add_n -> mse255(accumulate=True) -> d2s_clip_bwd(scale 1)."""
    import ctypes
    _f32(y, target, *parts)
    n, h2, w2, c = y.shape
    if c != 3 or (h2 & 1) or (w2 & 1) or not 1 <= len(parts) <= 6 or any(t.shape != y.shape for t in parts) or \
            target.shape != y.shape:
        raise ValueError('mse255_sum_s2d3: unsupported shapes')
    loss = torch.empty((1,), dtype=torch.float32, device=y.device)
    dz = torch.empty((n, h2 // 2, w2 // 2, 12), dtype=torch.float32, device=y.device)
    need = _lib.load().nimg_mse255_workspace_bytes()
    ws = _ws.get(need, y.device)
    ptrs = (ctypes.c_void_p * len(parts))(*[t.data_ptr() for t in parts])
    _lib.call('nimg_mse255_sum_s2d3', ptrs, len(parts), _p(y), _p(target), _p(loss), _p(dz), n, h2 // 2, w2 // 2,
              float(grad_scale), _p(ws), ws.numel(), _stream())
    return loss, dz


def fan_head_fwd(act, w, b, labels=None, loss_scale=1.0):
    _f32(act, w, b)
    n, h, wd, c = act.shape
    k = w.shape[1]
    dev = act.device
    gap = torch.empty((n, c), dtype=torch.float32, device=dev)
    probs = torch.empty((n, k), dtype=torch.float32, device=dev)
    loss_per = dlogits = None
    if labels is not None:
        _chk(labels)
        if labels.dtype != torch.int32:
            raise RuntimeError('labels must be int32')
        loss_per = torch.empty((n,), dtype=torch.float32, device=dev)
        dlogits = torch.empty((n, k), dtype=torch.float32, device=dev)
    _lib.call('nimg_fan_head_fwd', _p(act), _p(w), _p(b), _p(labels), _p(gap), _p(probs), _p(loss_per), _p(dlogits),
              n, h * wd, c, k, float(loss_scale), _stream())
    return gap, probs, loss_per, dlogits


def fan_head_bwd(act, gap, w, dlogits, loss_per, loss_scale, dw, db, alpha=None):
    """alpha: slope of the LeakyReLU whose derivative the kernel applies to the gradient it returns (default 0.2; 1.0 = none)."""
    _f32(act, gap, w, dlogits, loss_per, dw, db)
    n, h, wd, c = act.shape
    k = w.shape[1]
    dact = torch.empty_like(act)
    loss = torch.empty((1,), dtype=torch.float32, device=act.device)
    _lib.call('nimg_fan_head_bwd', _p(act), _p(gap), _p(w), _p(dlogits), _p(loss_per), _p(dact), _p(dw), _p(db),
              _p(loss), n, h * wd, c, k, float(loss_scale), LRELU_ALPHA if alpha is None else float(alpha), _stream())
    return dact, loss


# ---- fused FAN head, throughput mode (csrc/head.hip): 1x1 conv + LeakyReLU + global average pooling in one pass; the backward
# pass builds the pooled gradient's spread-out form inside the 1x1 layer's input-gradient kernel from one bit per activation.
# NIMG_NO_HEAD_FUSED=1 keeps the generic 1x1 / pooling kernels (A/B switch).
HEAD_FUSED = _os.environ.get('NIMG_NO_HEAD_FUSED') is None


def head_fused_ok(x, cout):
    """The fused head applies: throughput mode, bf16-stored input (N, H, W, C) with C = cout in {64, 128, 256} and
    H * W in {64, 128, 256}."""
    return bool(HEAD_FUSED and COMPUTE == 'bf16' and STORE_BF16 and x.dim() == 4 and _is_bf16(x) and x.shape[3] == cout and
                x.shape[0] > 0 and _lib.load().nimg_head_fused_ok(int(x.shape[1] * x.shape[2]), int(cout)))


def head_fwd(x, w, b, want_mask=True):
    """x (N,H,W,C) bf16, w (1,1,C,C), b (C) -> gap (N,C) float32, mask (N*H*W, C/32), mask_p (N, H*W/32, C): int32 sign bits of
    the activation in the two layouts the backward kernels read (or None, None)."""
    _chk(x, w, b)
    n, h, wd, c = x.shape
    gap = torch.empty((n, c), dtype=torch.float32, device=x.device)
    mask = mask_p = None
    if want_mask:           # the same sign bits twice: channel-major words per pixel (input gradient), pixel-major per channel (weights)
        mask = torch.empty((n * h * wd, c // 32), dtype=torch.int32, device=x.device)
        mask_p = torch.empty((n, h * wd // 32, c), dtype=torch.int32, device=x.device)
    _lib.call('nimg_head_fwd', _p(x), _p(weights_bf16(w, 0)), _p(b), _p(mask), _p(mask_p), _p(gap), n, h * wd, c, LRELU_ALPHA,
              _stream())
    return gap, mask, mask_p


def head_wgrad(x, mask_p, dlogits, wdense, dw, db=None, accumulate=False):
    """Weight (+ bias) gradient of the fused head's 1x1 layer from its bf16 input and the pixel-major sign words."""
    _chk(x, mask_p, dlogits, wdense)
    _f32(dw, db)
    n, h, wd, c = x.shape
    need = _lib.load().nimg_head_wgrad_workspace_bytes(n, c)
    ws = _ws_current(x.device).get(need, x.device)
    _lib.call('nimg_head_wgrad', _p(x), _p(mask_p), _p(dlogits), _p(wdense), dlogits.shape[1], _p(dw), _p(db), n, h * wd, c,
              LRELU_ALPHA, 1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return dw


def fan_dense_fwd(gap, w, b, labels=None, loss_scale=1.0):
    """The classifier on a pooled feature: probs (+ per-image loss and dlogits with labels)."""
    _f32(gap, w, b)
    n, c = gap.shape
    k = w.shape[1]
    probs = torch.empty((n, k), dtype=torch.float32, device=gap.device)
    loss_per = dlogits = None
    if labels is not None:
        _chk(labels)
        if labels.dtype != torch.int32:
            raise RuntimeError('labels must be int32')
        loss_per = torch.empty((n,), dtype=torch.float32, device=gap.device)
        dlogits = torch.empty((n, k), dtype=torch.float32, device=gap.device)
    _lib.call('nimg_fan_dense_fwd', _p(gap), _p(w), _p(b), _p(labels), _p(probs), _p(loss_per), _p(dlogits), n, c, k,
              float(loss_scale), _stream())
    return probs, loss_per, dlogits


def fan_dense_bwd(gap, dlogits, loss_per, loss_scale, dw, db):
    _f32(gap, dlogits, loss_per, dw, db)
    n, c = gap.shape
    loss = torch.empty((1,), dtype=torch.float32, device=gap.device)
    _lib.call('nimg_fan_dense_bwd', _p(gap), _p(dlogits), _p(loss_per), _p(dw), _p(db), _p(loss), n, c, dlogits.shape[1],
              float(loss_scale), _stream())
    return loss


def head_dgrad(mask, dlogits, wdense, w, in_mask, shape):
    """Gradient at the INPUT of the fused head's 1x1 layer, bf16 (N,H,W,C): from the classifier's dlogits, the activation's sign
    bits and (optionally) the bf16 tensor whose sign gates the LeakyReLU of the layer below."""
    _chk(mask, dlogits, wdense, w, in_mask)
    n, h, wd, c = shape
    dx = torch.empty((n, h, wd, c), dtype=torch.bfloat16, device=mask.device)
    if in_mask is not None and (not _is_bf16(in_mask) or tuple(in_mask.shape) != tuple(shape)):
        raise RuntimeError('head_dgrad: the mask tensor is the bf16-stored input of the 1x1 layer')
    _lib.call('nimg_head_dgrad', _p(mask), _p(dlogits), _p(wdense), dlogits.shape[1], _p(weights_bf16(w, 1)), _p(in_mask), _p(dx),
              n, h * wd, c, LRELU_ALPHA, _stream())
    return dx


def head_dact(mask, dlogits, wdense, shape):
    """Gradient at the fused head's 1x1 pre-activation as a bf16 tensor (N,H,W,C) - what its weight gradient reads."""
    _chk(mask, dlogits, wdense)
    n, h, wd, c = shape
    dact = torch.empty((n, h, wd, c), dtype=torch.bfloat16, device=mask.device)
    _lib.call('nimg_head_dact', _p(mask), _p(dlogits), _p(wdense), dlogits.shape[1], _p(dact), n, h * wd, c, LRELU_ALPHA, _stream())
    return dact


def adam_lr_t(lr, step, beta1=0.9, beta2=0.999):
    """Keras Adam's bias-corrected rate of iteration `step` (1-based), exactly as nimg_adam_step computes it: double
    arithmetic on the float32 values of lr / beta1 / beta2 that cross the C ABI."""
    lr, beta1, beta2 = (float(np.float32(v)) for v in (lr, beta1, beta2))
    return lr * math.sqrt(1.0 - math.pow(beta2, float(int(step)))) / (1.0 - math.pow(beta1, float(int(step))))


def adam_step(params, grads, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, skip_flag=None,
              lr_t_dev=None):
    """lr_t_dev: optional 1-element device tensor holding the bias-corrected rate (captured-graph replays refresh it)."""
    join_side_stream()
    _f32(params, grads, m, v, lr_t_dev)
    if lr_t_dev is not None:
        _lib.call('nimg_adam_step_dev', _p(params), _p(grads), _p(m), _p(v), params.numel(), _p(lr_t_dev), float(beta1),
                  float(beta2), float(eps), float(grad_scale), _p(skip_flag), _stream())
        return
    _lib.call('nimg_adam_step', _p(params), _p(grads), _p(m), _p(v), params.numel(), float(lr), float(beta1),
              float(beta2), float(eps), int(step), float(grad_scale), _p(skip_flag), _stream())


def int_fill(t, value=0):
    """t[...] = value for an int32 device tensor (the step's flag words) on the library's own kernel."""
    _chk(t)
    if t.dtype != torch.int32 or not t.is_contiguous():
        raise ValueError('int_fill: contiguous int32 tensor expected')
    _lib.call('nimg_int_words', _p(t), None, t.numel(), int(value), 0, _stream())
    return t


def int_max_(dst, src):
    """dst = max(dst, src) element-wise, int32, in place."""
    _chk(dst, src)
    if dst.dtype != torch.int32 or src.dtype != torch.int32 or dst.numel() != src.numel():
        raise ValueError('int_max_: two int32 tensors of one size expected')
    _lib.call('nimg_int_words', _p(dst), _p(src), dst.numel(), 0, 1, _stream())
    return dst


def float_fill(t, value):
    _f32(t)
    _lib.call('nimg_float_fill', _p(t), t.numel(), float(value), _stream())
    return t


def nan_flag(g, flag):
    join_side_stream()
    _f32(g)
    _lib.call('nimg_nan_flag', _p(g), g.numel(), _p(flag), _stream())


# ----------------------------------------------------------------------------------------------------------------
# constrained conv pieces, manipulations
def constrained_kernel(kernel, strength=100.0):
    _f32(kernel)
    nf = torch.empty_like(kernel)
    _lib.call('nimg_constrained_kernel_fwd', _p(kernel), _p(nf), kernel.shape[0], kernel.shape[2], float(strength),
              _stream())
    return nf


def constrained_kernel_bwd(kernel, dnf, dk, strength=100.0):
    _f32(kernel, dnf, dk)
    _lib.call('nimg_constrained_kernel_bwd', _p(kernel), _p(dnf), _p(dk), kernel.shape[0], kernel.shape[2],
              float(strength), _stream())
    return dk


def cconv3(x, w, pad_mode=1, out=None, want_f32=True, want_c4=False):
    """5x5 convolution 3 -> 3 on the VALU (the ConstrainedConv2D core): x (N,H,W,3), w (5,5,3,3) HWIO device tensor.
    Returns (y float32 (N,H,W,3) | None, c4 bf16 (N,H,W,4) = {y0, y1, y2, 1} | None)."""
    _f32(x, w, out)
    n, h, wd, c = x.shape
    if c != 3 or tuple(w.shape) != (5, 5, 3, 3):
        raise ValueError('cconv3: (N,H,W,3) input and a (5,5,3,3) kernel expected')
    y = (torch.empty_like(x) if out is None else out) if want_f32 else None
    c4 = torch.empty((n, h, wd, 4), dtype=torch.bfloat16, device=x.device) if want_c4 else None
    _lib.call('nimg_cconv3', _p(x), _p(w), _p(y), _p(c4), n, h, wd, int(pad_mode), _stream())
    return y, c4


# The row-band front-end kernels (csrc/frontend.hip) serve the FAN's standard first layer in throughput mode; NIMG_OLD_FRONTEND=1
# keeps the generic small-channel kernels (A/B runs).
FRONT_END = _os.environ.get('NIMG_OLD_FRONTEND') is None
CCONV_DGRAD_MFMA = _os.environ.get('NIMG_NO_CCONV_DGRAD_MFMA') is None       # A/B switch
# 5x5 stride-2 layers as 3x3 stride-1 layers over the space-to-depth image (throughput mode); NIMG_NO_S2D_CONV=1: the strided kernels
S2D_CONV = _os.environ.get('NIMG_NO_S2D_CONV') is None
# 5x5 input gradients of the fused conv + pool layers on the sparse matrix instruction (csrc/dgrad5s.hip): correct, but at
# parity with the ring kernels (its dense operand is twice the LDS bytes per matrix cycle) - off unless NIMG_SPARSE_DGRAD=1
SPARSE_DGRAD = _os.environ.get('NIMG_SPARSE_DGRAD') is not None


def front_end_ok(cin, cout, ks, h, w, n=None):
    """The FAN front-end kernels (csrc/frontend.hip) take this first layer; with `n` also: its backward entry points address
    their tensors through 32-bit buffer descriptors (the forward chunks the batch, they do not)."""
    small = n is None or n * h * w * 16 < (1 << 31) - 65536            # pooled gradient (n,h/2,w/2,32) bf16, pixels (n,h,w,4) bf16
    return FRONT_END and COMPUTE == 'bf16' and cin == 3 and cout == 32 and ks == 5 and h % 2 == 0 and w % 2 == 0 and \
        h >= 4 and w >= 4 and small


def conv1_pool_c4(c4, w, bias, act='leaky_relu', want_idx=True, out_bf16=True):
    """Conv2D(32, 5x5, SAME) + bias + activation + MaxPool2D(2) over the bf16 {c0,c1,c2,1} pixels written by cconv3 (the FAN's
    first convolution, throughput mode).  Returns (pooled (N,H/2,W/2,32), arg-max codes (N,H/2,W/2,8) uint8 | None): 2 bits per
    channel, channel c in byte c >> 2 at bits 2 (c & 3) (include/nimg.h nimg_conv1_pool_fwd_c4)."""
    _f32(w, bias)
    _chk(c4)
    n, h, wd, c = c4.shape
    if c4.dtype != torch.bfloat16 or c != 4 or tuple(w.shape) != (5, 5, 3, 32) or (h & 1) or (wd & 1):
        raise ValueError('conv1_pool_c4: (N,H,W,4) bf16 pixels, a (5,5,3,32) kernel and even sizes expected')
    pooled = torch.empty((n, h // 2, wd // 2, 32), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=c4.device)
    idx = torch.empty((n, h // 2, wd // 2, 8), dtype=torch.uint8, device=c4.device) if want_idx else None
    _lib.call('nimg_conv1_pool_fwd_c4', _p(c4), _p(w), _p(bias), _p(pooled), _p(idx), n, h, wd,
              LRELU_ALPHA if act == 'leaky_relu' else 1.0, 1 if out_bf16 else 0, _stream())
    return pooled, idx


def conv1_wgrad_c4(c4, g, idx, dw=None, db=None, accumulate=False, side=False):
    """Weight / bias gradient of conv1_pool_c4 from the pooled gradient g (already x LeakyReLU') and the 2-bit arg-max codes."""
    if side and _SIDE['enabled'] and dw is not None:
        with _on_side_stream(c4, g, idx, key=dw.data_ptr()):
            return conv1_wgrad_c4(c4, g, idx, dw=dw, db=db, accumulate=accumulate, side=False)
    _f32(dw, db)
    _fb(g)
    _chk(c4, idx)
    n, h, wd, _ = c4.shape
    if c4.dtype != torch.bfloat16 or tuple(g.shape) != (n, h // 2, wd // 2, 32) or tuple(idx.shape) != (n, h // 2, wd // 2, 8) or \
            idx.dtype != torch.uint8:
        raise ValueError('conv1_wgrad_c4: (N,H,W,4) bf16 pixels, a (N,H/2,W/2,32) gradient and (N,H/2,W/2,8) arg-max codes expected')
    if dw is None:
        dw = torch.empty((5, 5, 3, 32), dtype=torch.float32, device=c4.device)
    need = _lib.load().nimg_conv1_wgrad_c4_workspace_bytes()
    ws = _ws_current(c4.device).get(need, c4.device)
    _lib.call('nimg_conv1_wgrad_c4', _p(c4), _p(g), _p(idx), _p(dw), _p(db), n, h, wd, 1 if _is_bf16(g) else 0,
              1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return dw


def conv1_dgrad_pooled(g, idx, w, out=None):
    """Input gradient (N,H,W,3) of conv1_pool_c4 from the pooled gradient g (N,H/2,W/2,32) and the 2-bit arg-max codes."""
    _f32(w, out)
    _fb(g)
    _chk(idx)
    n, hp, wp, c = g.shape
    if c != 32 or tuple(w.shape) != (5, 5, 3, 32) or tuple(idx.shape) != (n, hp, wp, 8) or idx.dtype != torch.uint8:
        raise ValueError('conv1_dgrad_pooled: a (N,H/2,W/2,32) gradient, (N,H/2,W/2,8) arg-max codes and a (5,5,3,32) kernel expected')
    if out is None:
        out = torch.empty((n, 2 * hp, 2 * wp, 3), dtype=torch.float32, device=g.device)
    _lib.call('nimg_conv1_dgrad_pooled', _p(g), _p(idx), _p(w), _p(out), n, 2 * hp, 2 * wp, 1 if _is_bf16(g) else 0, _stream())
    return out


def cconv3_dgrad(dy, nf):
    """Input gradient of the SYMMETRIC-padded 5x5 3 -> 3 filter nf: zero-padded correlation with the flipped / transposed
    filter plus the mirror terms of the two outermost rows / columns."""
    _f32(dy, nf)
    n, h, wd, _ = dy.shape
    if COMPUTE == 'bf16' and wd % 64 == 0 and CCONV_DGRAD_MFMA:      # throughput mode: the main term on the matrix core
        dx = torch.empty_like(dy)
        _lib.call('nimg_conv5c3_bf16', _p(dy), _p(flip_weights(nf)), _p(dx), n, h, wd, _stream())
    else:
        dx, _ = cconv3(dy, flip_weights(nf), pad_mode=0)
    _lib.call('nimg_cconv3_dgrad_border', _p(dy), _p(nf), _p(dx), n, h, wd, _stream())
    return dx


def fold_pad(dpad, pad, pad_mode):
    _f32(dpad)
    n, hp, wp, c = dpad.shape
    dx = torch.empty((n, hp - 2 * pad, wp - 2 * pad, c), dtype=torch.float32, device=dpad.device)
    _lib.call('nimg_fold_pad', _p(dpad), _p(dx), n, hp - 2 * pad, wp - 2 * pad, c, pad, pad_mode, _stream())
    return dx


def gaussian_fwd(x, gk25, out=None, clip=True, want_mask=True):
    _f32(x, gk25, out)
    n, h, w, _ = x.shape
    y = torch.empty_like(x) if out is None else out
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if want_mask else None
    _lib.call('nimg_gaussian_fwd', _p(x), _p(y), _p(mask), _p(gk25), n, h, w, 1 if clip else 0, _stream())
    return y, mask


def gaussian_bwd(dy, mask, gk25):
    _f32(dy, gk25)
    n, h, w, _ = dy.shape
    dx = torch.empty_like(dy)
    _lib.call('nimg_gaussian_bwd', _p(dy), _p(mask), _p(dx), _p(gk25), n, h, w, _stream())
    return dx


def dwfilter_fwd(x, taps, k, pad_mode, out=None, clip=True, want_mask=True):
    """Any odd k x k per-channel filter, mirrored border (pad_mode 'SYMMETRIC' | 'REFLECT'), optional clip + pass mask."""
    _f32(x, taps, out)
    n, h, w, _ = x.shape
    if taps.numel() != k * k:
        raise ValueError('taps: k*k floats')
    y = torch.empty_like(x) if out is None else out
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if want_mask else None
    _lib.call('nimg_dwfilter_fwd', _p(x), _p(y), _p(mask), _p(taps), int(k), PAD_MODES[pad_mode], n, h, w,
              1 if clip else 0, _stream())
    return y, mask


def dwfilter_bwd(dy, mask, taps, k, pad_mode):
    _f32(dy, taps)
    n, h, w, _ = dy.shape
    dx = torch.empty_like(dy)
    _lib.call('nimg_dwfilter_bwd', _p(dy), _p(mask), _p(dx), _p(taps), int(k), PAD_MODES[pad_mode], n, h, w, _stream())
    return dx


def sharpen_fwd(x, gk9, out=None, want_aux=True):
    _f32(x, gk9, out)
    n, h, w, _ = x.shape
    y = torch.empty_like(x) if out is None else out
    aux = torch.empty_like(x) if want_aux else None
    mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if want_aux else None
    _lib.call('nimg_sharpen_fwd', _p(x), _p(y), _p(aux), _p(mask), _p(gk9), n, h, w, _stream())
    return y, aux, mask


def sharpen_bwd(x, dy, aux, mask, gk9):
    _f32(x, dy, aux, gk9)
    n, h, w, _ = x.shape
    dx = torch.empty_like(x)
    _lib.call('nimg_sharpen_bwd', _p(x), _p(dy), _p(aux), _p(mask), _p(dx), _p(gk9), n, h, w, _stream())
    return dx


class AxisOperator(object):
    """A banded linear operator along a spatial axis, stored as CSR (+ its transpose) on the device."""

    def __init__(self, dense, device):
        dense = np.asarray(dense, np.float64)
        self.out_size, self.in_size = dense.shape
        self.fwd = self._csr(dense, device)
        self.bwd = self._csr(dense.T, device)

    @staticmethod
    def _csr(m, device):
        rowptr, col, val = [0], [], []
        for r in range(m.shape[0]):
            nz = np.nonzero(m[r])[0]
            col.extend(nz.tolist())
            val.extend(m[r, nz].tolist())
            rowptr.append(len(col))
        return (torch.tensor(rowptr, dtype=torch.int32, device=device),
                torch.tensor(col, dtype=torch.int32, device=device),
                torch.tensor(np.asarray(val, np.float32), dtype=torch.float32, device=device))


def sparse_axis_apply(x, csr, axis, out_size, out=None):
    _f32(x, out)
    rowptr, col, val = csr
    n, h, w, c = x.shape
    shape = (n, out_size, w, c) if axis == 0 else (n, h, out_size, c)
    y = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
    _lib.call('nimg_sparse_axis_apply', _p(x), _p(y), _p(rowptr), _p(col), _p(val), n, h, w, c, axis, out_size,
              _stream())
    return y


# ----------------------------------------------------------------------------------------------------------------
# learned codec pieces
def affine(x, a, b, out=None):
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_affine', _p(x), _p(y), x.numel(), float(a), float(b), _stream())
    return y


def lrelu(x):
    _f32(x)
    y = torch.empty_like(x)
    _lib.call('nimg_lrelu_fwd', _p(x), _p(y), x.numel(), LRELU_ALPHA, _stream())
    return y


def zero_insert2(x):
    _f32(x)
    n, h, w, c = x.shape
    y = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
    _lib.call('nimg_zero_insert2', _p(x), _p(y), n, h, w, c, _stream())
    return y


def s2d2_affine(x, a=1.0, b=0.0, cp=None):
    """bf16 space-to-depth image (N,H/2,W/2,cp) of a x + b: block channel (2 pr + pc) C + ci = pixel (2by + pr, 2bx + pc)."""
    _f32(x)
    n, h, w, c = x.shape
    cp = (4 * c + 15) // 16 * 16 if cp is None else cp
    y = torch.empty((n, h // 2, w // 2, cp), dtype=torch.bfloat16, device=x.device)
    _lib.call('nimg_s2d2_affine_bf16', _p(x), _p(y), n, h, w, c, cp, float(a), float(b), _stream())
    return y


def s2d_conv_weights(w5, cp=None):
    """(5,5,Cin,Cout) kernel of a stride-2 SAME convolution -> the (3,3,cp,Cout) kernel of the equivalent stride-1 convolution
    over the space-to-depth image."""
    _f32(w5)
    if w5.shape[0] != 5 or w5.shape[1] != 5:
        raise ValueError('s2d_conv_weights: a 5x5 kernel expected')
    cin, cout = w5.shape[2], w5.shape[3]
    cp = (4 * cin + 15) // 16 * 16 if cp is None else cp
    w3 = torch.empty((3, 3, cp, cout), dtype=torch.float32, device=w5.device)
    _lib.call('nimg_s2d_conv_weights', _p(w5), _p(w3), cin, cp, cout, _stream())
    return w3


def s2d_conv_weights_bwd(dw3, dw5, accumulate=False):
    _f32(dw3, dw5)
    cp, cout = dw3.shape[2], dw3.shape[3]
    cin = dw5.shape[2]
    if tuple(dw3.shape[:2]) != (3, 3) or tuple(dw5.shape) != (5, 5, cin, cout) or cp < 4 * cin:
        raise ValueError('s2d_conv_weights_bwd: (3,3,cp,Cout) -> (5,5,Cin,Cout) with cp >= 4 Cin')
    _lib.call('nimg_s2d_conv_weights_bwd', _p(dw3), _p(dw5), cin, cp, cout, 1 if accumulate else 0, _stream())
    return dw5


def d2s2_scale(xs, c, scale=1.0):
    """(N,H/2,W/2,cp) float32 block image -> scale * its depth-to-space image (N,H,W,c)."""
    _f32(xs)
    n, hb, wb, cp = xs.shape
    x = torch.empty((n, 2 * hb, 2 * wb, c), dtype=torch.float32, device=xs.device)
    _lib.call('nimg_d2s2_scale', _p(xs), _p(x), n, 2 * hb, 2 * wb, c, cp, float(scale), _stream())
    return x


def s2d_conv_ok(ks, stride, h, w, cin):
    """A 5x5 stride-2 SAME layer over an even-sized image can run as a 3x3 stride-1 layer over the space-to-depth image."""
    return COMPUTE == 'bf16' and S2D_CONV and ks == 5 and stride == 2 and h % 2 == 0 and w % 2 == 0 and cin >= 1


def conv2d_dgrad_strided2(dz, w, in_hw, scale=1.0, act_mask=None, out_bf16=False, mask_s2d=False):
    """Input gradient of a stride-2 TF-SAME convolution (times LeakyReLU'(act_mask) if given).  Throughput mode: the 3x3
    stride-1 input gradient of the equivalent convolution over the space-to-depth image (s2d_conv_weights), written straight
    in the depth-to-space layout - and masked - by the kernel's epilogue (a separate d2s2_scale pass where the block channels
    are padded, i.e. the 3-channel image layer).  Parity mode: zero insertion + stride-1 correlation with the flipped kernel
    (pad = ks-1-pad_before)."""
    ks = w.shape[0]
    h, wd = in_hw
    cin = w.shape[2]
    if s2d_conv_ok(ks, 2, h, wd, cin) and (not _is_bf16(dz) or dz.shape[3] % 8 == 0):
        w3 = s2d_conv_weights(w)
        if w3.shape[2] == 4 * cin and cin % 4 == 0 and scale == 1.0 and dz.shape[3] % 8 == 0 and D2S_EPILOGUE:
            # (out_bf16: the gradient is stored as bf16 - for a tensor that only feeds matrix-core operands)
            # (mask_s2d: act_mask is the space-to-depth image of the masking activation, i.e. in THIS convolution's layout)
            return conv2d_dgrad(dz, w3, (h // 2, wd // 2), act_mask=act_mask, d2s_out=True, out_bf16=out_bf16,
                                mask_conv_layout=mask_s2d)
        if mask_s2d:
            raise RuntimeError('a space-to-depth-stored mask needs the fused depth-to-space epilogue')
        dxs = conv2d_dgrad(dz, w3, (h // 2, wd // 2))
        d = d2s2_scale(dxs, cin, scale)
    else:
        _, pt = same_pads(h, ks, 2)
        _, pl = same_pads(wd, ks, 2)
        up = zero_insert2(dz)
        if up.shape[1] != h or up.shape[2] != wd:
            raise NotImplementedError('strided dgrad is built for even input sizes')
        d = conv2d(up, w, None, pads=(ks - 1 - pt, ks - 1 - pl), out_hw=(h, wd), _wmode=1)
        d = d if scale == 1.0 else affine(d, scale, 0.0)
    return d if act_mask is None else lrelu_bwd(d, act_mask, out=d)


class LatentWorkspace(object):
    """float64 scratch of the DiscreteLatent kernels; must survive from forward to backward."""

    def __init__(self, k, device):
        self.k = k
        self.buf = torch.empty(int(_lib.load().nimg_latent_workspace_bytes(k)), dtype=torch.uint8, device=device)

    def hist_sums(self):
        """view of the K float64 histogram sums (the unit to all-reduce under data parallelism)"""
        return self.buf.view(torch.float64)[1024 * self.k:1024 * self.k + self.k]


LATENT_ROUNDING = {'identity': 0, 'soft': 1, 'sin': 2}       # bits 2-3 of the kernels' flag word when soft_codebook is off


def _latent_flags(soft_codebook, unit_codebook, rounding):
    if soft_codebook:
        return 1 | (2 if unit_codebook else 0)
    return (2 if unit_codebook else 0) | (LATENT_ROUNDING[rounding] << 2)


def latent_fwd(z, scale, codebook, ws, v=50.0, gamma=25.0, soft_codebook=True, count_global=0, finalize=True,
               unit_codebook=False, rounding='identity'):
    """unit_codebook: the caller's promise that codebook[k] = codebook[0] + k (include/nimg.h: the kernels then evaluate only
    the centres whose weight can reach the float64 sums).  rounding (soft_codebook=False): identity | soft | sin
    (models/layers.py:118-134)."""
    _f32(z, scale, codebook)
    latent = torch.empty_like(z)
    entropy = torch.empty((1,), dtype=torch.float32, device=z.device)
    _lib.call('nimg_latent_fwd', _p(z), _p(scale), _p(codebook), codebook.numel(), float(v), float(gamma),
              _latent_flags(soft_codebook, unit_codebook, rounding), _p(latent), _p(entropy), z.numel(), int(count_global),
              _p(ws.buf),
              ws.buf.numel(), 1 if finalize else 0, _stream())
    return latent, entropy


def latent_entropy_finalize(ws, count_global, entropy):
    _lib.call('nimg_latent_entropy_finalize', ws.k, int(count_global), _p(entropy), _p(ws.buf), _stream())


def latent_bwd(z, scale, latent, dlatent, entropy_coef, codebook, ws, dscale=None, v=50.0, gamma=25.0,
               soft_codebook=True, accumulate_dscale=False, unit_codebook=False, rounding='identity'):
    _f32(z, scale, latent, dlatent, codebook, dscale)
    dz = torch.empty_like(z)
    _lib.call('nimg_latent_bwd', _p(z), _p(scale), _p(latent), _p(dlatent), float(entropy_coef), _p(codebook),
              codebook.numel(), float(v), float(gamma), _latent_flags(soft_codebook, unit_codebook, rounding), _p(dz),
              _p(dscale),
              1 if accumulate_dscale else 0, z.numel(), _p(ws.buf), ws.buf.numel(), _stream())
    return dz


def l2_loss(target, y, grad_scale=None, grad_out=None, accumulate=False):
    _f32(target, y, grad_out)
    loss = torch.empty((1,), dtype=torch.float32, device=y.device)
    g = None
    if grad_scale is not None:
        g = torch.empty_like(y) if grad_out is None else grad_out
    ws = _ws.get(_lib.load().nimg_l2_loss_workspace_bytes(), y.device)
    _lib.call('nimg_l2_loss', _p(target), _p(y), _p(loss), _p(g), y.numel(), float(grad_scale or 0.0),
              1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return loss, g


# ----------------------------------------------------------------------------------------------------------------
# awgn / gamma / median manipulations
def awgn_fwd(x, noise, strength, out=None, want_mask=True):
    _f32(x, noise, out)
    y = torch.empty_like(x) if out is None else out
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_mask else None
    _lib.call('nimg_awgn_fwd', _p(x), _p(noise), _p(y), _p(mask), x.numel(), float(strength), _stream())
    return y, mask


def awgn_bwd(x, noise, dy, mask, strength):
    _f32(x, noise, dy)
    dx = torch.empty_like(x)
    _lib.call('nimg_awgn_bwd', _p(x), _p(noise), _p(dy), _p(mask), _p(dx), x.numel(), float(strength), _stream())
    return dx


def gamma_fwd(x, g, out=None):
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_gamma_fwd', _p(x), _p(y), x.numel(), float(g), _stream())
    return y


def gamma_bwd(x, dy, g):
    _f32(x, dy)
    dx = torch.empty_like(x)
    _lib.call('nimg_gamma_bwd', _p(x), _p(dy), _p(dx), x.numel(), float(g), _stream())
    return dx


def median_fwd(x, kernel, out=None, want_sel=True):
    _f32(x, out)
    n, h, w, _ = x.shape
    y = torch.empty_like(x) if out is None else out
    sel = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_sel else None
    _lib.call('nimg_median_fwd', _p(x), _p(y), _p(sel), n, h, w, int(kernel), _stream())
    return y, sel


def median_bwd(dy, sel, kernel):
    _f32(dy)
    n, h, w, _ = dy.shape
    dx = torch.zeros_like(dy)
    _lib.call('nimg_median_bwd', _p(dy), _p(sel), _p(dx), n, h, w, int(kernel), _stream())
    return dx


# ----------------------------------------------------------------------------------------------------------------
# image quality metrics
_SSIM_GAUSS = {}


def ssim(a, b, mode='skimage', max_val=1.0):
    """Per-image SSIM of two (N,H,W,C) batches. mode 'skimage' = helpers/metrics.ssim of the reference (7x7 uniform
    window), 'tf' = tf.image.ssim (11x11 Gaussian, sigma 1.5)."""
    _f32(a, b)
    n, h, w, c = a.shape
    m = {'skimage': 0, 'tf': 1}[mode]
    gk = None
    if m == 1:
        gk = _ssim_window(a.device)
    out = torch.empty((n,), dtype=torch.float32, device=a.device)
    ws = _ws.get(_lib.load().nimg_ssim_workspace_bytes(n), a.device)
    _lib.call('nimg_ssim', _p(a), _p(b), _p(out), n, h, w, c, m, float(max_val), _p(gk), _p(ws), ws.numel(), _stream())
    return out


def _ssim_window(device):
    key = str(device)
    if key not in _SSIM_GAUSS:           # tf _fspecial_gauss: softmax of -(x^2 + y^2) / (2 sigma^2)
        co = np.arange(11, dtype=np.float64) - 5.0
        g = -0.5 * (co[:, None] ** 2 + co[None, :] ** 2) / 1.5 ** 2
        g = np.exp(g - g.max())
        _SSIM_GAUSS[key] = torch.from_numpy((g / g.sum()).astype(np.float32).ravel()).to(device)
    return _SSIM_GAUSS[key]


def mae255(a, b, grad_scale=None, grad_out=None, accumulate=False):
    """helpers/tf_helpers.py:35-36 mean |255a - 255b|.  Returns (loss[1], grad wrt a or None), like mse255."""
    _f32(a, b, grad_out)
    loss = torch.empty((1,), dtype=torch.float32, device=a.device)
    g = None
    if grad_scale is not None:
        g = torch.empty_like(a) if grad_out is None else grad_out
    ws = _ws.get(_lib.load().nimg_mae255_workspace_bytes(), a.device)
    _lib.call('nimg_mae255', _p(a), _p(b), _p(loss), _p(g), a.numel(), float(grad_scale or 0.0),
              1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return loss, g


def ssim_loss(y, t, grad_scale=None, grad_out=None, accumulate=False, max_val=1.0):
    """helpers/tf_helpers.py:39-40 mean 255 (1 - tf.image.ssim(y, t, 1)).  Returns (loss[1], grad wrt y or None)."""
    _f32(y, t, grad_out)
    n, h, w, c = y.shape
    if h < 11 or w < 11:
        raise ValueError('tf.image.ssim needs images of at least 11 x 11 pixels')
    loss = torch.empty((1,), dtype=torch.float32, device=y.device)
    g = None
    if grad_scale is not None:
        g = torch.empty_like(y) if grad_out is None else grad_out
    ws = _ws.get(_lib.load().nimg_ssim_loss_workspace_bytes(n, h, w, c, 0 if g is None else 1), y.device)
    _lib.call('nimg_ssim_loss', _p(y), _p(t), _p(loss), _p(g), n, h, w, c, float(max_val), _p(_ssim_window(y.device)),
              float(grad_scale or 0.0), 1 if accumulate else 0, _p(ws), ws.numel(), _stream())
    return loss, g


MSSSIM_SCALES = 5


def msssim_loss(y, t, grad_scale=None, grad_out=None, accumulate=False, max_val=1.0):
    """helpers/tf_helpers.py:43-44 mean 255 (1 - tf.image.ssim_multiscale(y, t, 1)): contrast-structure means of four scales
    and the SSIM mean of the fifth, each scale a 2x2 average pooling of the previous one, combined by the weighted geometric
    mean per (image, channel).  Returns (loss[1], grad wrt y or None).  Sizes must be multiples of 16 with H/16, W/16 >= 11
    (TF pads odd scales symmetrically instead; not built)."""
    _f32(y, t, grad_out)
    n, h, w, c = y.shape
    div = 1 << (MSSSIM_SCALES - 1)
    if h % div or w % div or h // div < 11 or w // div < 11:
        raise ValueError('MS-SSIM needs image sizes that are multiples of {} and at least {} pixels'.format(div, 11 * div))
    dev = y.device
    gk = _ssim_window(dev)
    planes = n * c
    values = torch.empty((MSSSIM_SCALES, planes), dtype=torch.float32, device=dev)
    items = torch.tensor([float((h // (1 << k) - 10) * (w // (1 << k) - 10)) for k in range(MSSSIM_SCALES)],
                         dtype=torch.float32, device=dev)
    want_grad = grad_scale is not None
    ys, ts, maps = [y], [t], []
    ws = _ws.get(_lib.load().nimg_ssim_planes_workspace_bytes(n, c), dev)
    for k in range(MSSSIM_SCALES):
        if k > 0:
            ys.append(avgpool(ys[-1], 2))
            ts.append(avgpool(ts[-1], 2))
        hk, wk = ys[k].shape[1], ys[k].shape[2]
        last = k == MSSSIM_SCALES - 1
        mp = torch.empty((3, n, hk - 10, wk - 10, c), dtype=torch.float32, device=dev) if want_grad else None
        maps.append(mp)
        _lib.call('nimg_ssim_planes', _p(ys[k]), _p(ts[k]), n, hk, wk, c, float(max_val), _p(gk),
                  _p(values[k]) if last else None, None if last else _p(values[k]), _p(mp),
                  (1 if last else 2) if want_grad else 0, _p(ws), ws.numel(), _stream())
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    coef = torch.empty_like(values) if want_grad else None
    _lib.call('nimg_msssim_combine', _p(values), _p(items), MSSSIM_SCALES, planes, _p(loss), _p(coef), _stream())
    if not want_grad:
        return loss, None
    acc = None
    for k in range(MSSSIM_SCALES - 1, -1, -1):           # coarse to fine: g_k + un-pooled gradient of the coarser scales
        hk, wk = ys[k].shape[1], ys[k].shape[2]
        if k == 0:
            g = torch.empty_like(y) if grad_out is None else grad_out
            acc_flag = accumulate
        else:
            g, acc_flag = torch.empty_like(ys[k]), False
        _lib.call('nimg_ssim_maps_grad', _p(ys[k]), _p(ts[k]), _p(maps[k]), _p(coef[k]), _p(g), n, hk, wk, c, _p(gk),
                  float(grad_scale), 1 if acc_flag else 0, _stream())
        if acc is not None:
            add(g, avgpool_bwd(acc, 2), out=g)
        acc = g
    return loss, acc


IMAGE_LOSSES = {'L2': mse255, 'L1': mae255, 'SSIM': ssim_loss, 'MS-SSIM': msssim_loss}


# ----------------------------------------------------------------------------------------------------------------
# GPU-resident data feed (helpers/dataset.py:89-131, helpers/loading.py:132-211)
DISCARD_MODES = {None: 0, 'flat': 1, 'flat-aggressive': 2, 'dark-n-textured': 3}


def _feed_chk(t, dtype, what):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise TypeError('{}: contiguous {} tensor on the GPU expected, got {} on {}'.format(what, dtype, t.dtype, t.device))


def patch_stats(rgb, image_idx, cand_xy, patch):
    """rgb (n, H, W, 3) uint8, image_idx (B) int32, cand_xy (B, A, 2) int32 -> (var, mean) float64 (B, A)."""
    _feed_chk(rgb, torch.uint8, 'rgb'), _feed_chk(image_idx, torch.int32, 'image_idx'), _feed_chk(cand_xy, torch.int32, 'cand_xy')
    n, h, w, _ = rgb.shape
    b, a = cand_xy.shape[0], cand_xy.shape[1]
    var = torch.empty((b, a), dtype=torch.float64, device=rgb.device)
    mean = torch.empty_like(var)
    _lib.call('nimg_patch_stats', _p(rgb), n, h, w, _p(image_idx), _p(cand_xy), b, a, int(patch), _p(var), _p(mean), _stream())
    return var, mean


def patch_select(cand_xy, uniforms, var, mean, discard, max_attempts):
    """The discard policy over (B, A) candidates -> (chosen_xy (B, 2) int32, candidates consumed (B) int32)."""
    _feed_chk(cand_xy, torch.int32, 'cand_xy'), _feed_chk(uniforms, torch.float32, 'uniforms')
    _feed_chk(var, torch.float64, 'var'), _feed_chk(mean, torch.float64, 'mean')
    b, a = cand_xy.shape[0], cand_xy.shape[1]
    xy = torch.empty((b, 2), dtype=torch.int32, device=cand_xy.device)
    used = torch.empty((b,), dtype=torch.int32, device=cand_xy.device)
    _lib.call('nimg_patch_select', _p(cand_xy), _p(uniforms), _p(var), _p(mean), b, a, int(max_attempts),
              DISCARD_MODES[discard], _p(xy), _p(used), _stream())
    return xy, used


def patch_gather(raw, rgb, image_idx, xy, patch):
    """raw (n, H/2, W/2, 4) uint16 (stored as int16 bits) | None, rgb (n, H, W, 3) uint8 | None, xy (B, 2) int32 ->
    (x (B, p/2, p/2, 4) | None, y (B, p, p, 3) | None) float32."""
    _feed_chk(raw, torch.int16, 'raw'), _feed_chk(rgb, torch.uint8, 'rgb')
    _feed_chk(image_idx, torch.int32, 'image_idx'), _feed_chk(xy, torch.int32, 'xy')
    if raw is None and rgb is None:
        raise ValueError('nothing to gather')
    if rgb is not None:
        n, h, w, _ = rgb.shape
    else:
        n, h, w = raw.shape[0], 2 * raw.shape[1], 2 * raw.shape[2]
    b, p = xy.shape[0], int(patch)
    dev = xy.device
    x = None if raw is None else torch.empty((b, p // 2, p // 2, 4), dtype=torch.float32, device=dev)
    y = None if rgb is None else torch.empty((b, p, p, 3), dtype=torch.float32, device=dev)
    _lib.call('nimg_patch_gather', _p(raw), _p(rgb), n, h, w, _p(image_idx), _p(xy), b, p, _p(x), _p(y), _stream())
    return x, y


# ----------------------------------------------------------------------------------------------------------------
# element-wise pieces of the INet / DNet pipelines
def tanh(x, out=None):
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_tanh_fwd', _p(x), _p(y), x.numel(), _stream())
    return y


def tanh_bwd(dy, y, out=None):
    _f32(dy, y, out)
    dx = torch.empty_like(dy) if out is None else out
    _lib.call('nimg_tanh_bwd', _p(dy), _p(y), _p(dx), dy.numel(), _stream())
    return dx


ACTIVATIONS = {'leaky_relu': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'softsign': 4}     # helpers/tf_helpers.py:22-28
BIG_KERNELS = (4, 6, 7, 8, 9, 10, 11)    # kernel sizes served by the generic float32 kernels only (conv_mfma.hip / conv_wgrad.hip)


def activation(x, kind, out=None):
    """activation_mapping[kind](x), element-wise (out may be x)."""
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_activation_fwd', _p(x), _p(y), x.numel(), ACTIVATIONS[kind], LRELU_ALPHA, _stream())
    return y


def activation_bwd(dy, y, kind, out=None):
    """dy * activation'(.) taken from the stored output y (out may be dy)."""
    _f32(dy, y, out)
    dx = torch.empty_like(dy) if out is None else out
    _lib.call('nimg_activation_bwd', _p(dy), _p(y), _p(dx), dy.numel(), ACTIVATIONS[kind], LRELU_ALPHA, _stream())
    return dx


def clip01(x, out=None):
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_clip01', _p(x), _p(y), x.numel(), _stream())
    return y


# element-wise pieces of ClassicISP (models/pipelines.py:416-453, models/layers.py:206-258)
def isp_residual(x, f, alpha, clip=True, out=None):
    """y = [clip01](x - alpha f); f = None -> y = [clip01](x).  alpha: 1-element device tensor."""
    _f32(x, f, alpha, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_isp_residual_fwd', _p(x), _p(f), _p(alpha), _p(y), x.numel(), 1 if clip else 0, _stream())
    return y


def isp_residual_bwd(dy, f, alpha, dalpha, accumulate=False):
    """df = -alpha dy (returned), dalpha (+)= -sum(dy f); the clip of the forward pass is straight-through."""
    _f32(dy, f, alpha, dalpha)
    ws = _ws.get(_lib.load().nimg_isp_residual_workspace_bytes(), dy.device)
    df = torch.empty_like(dy)
    _lib.call('nimg_isp_residual_bwd', _p(dy), _p(f), _p(alpha), _p(df), _p(dalpha), _p(ws), dy.numel(),
              1 if accumulate else 0, _stream())
    return df


def mask_scale(x, keep, scale, out=None):
    """y = keep ? x * scale : 0 - Keras Dropout with a given uint8 mask (forward and backward)."""
    _f32(x, out)
    if keep.dtype != torch.uint8 or keep.numel() != x.numel() or not keep.is_contiguous():
        raise TypeError('keep: contiguous uint8 mask of the size of x')
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_mask_scale', _p(x), _p(keep), _p(y), x.numel(), float(scale), _stream())
    return y


def confusion_accumulate(probs, labels=None, conf=None, want_pred=True):
    """Decisions (first arg-max per row, int32) of a (n,k) probability batch and, given int32 labels and a (k,k) int64
    device matrix, conf[label][pred] += 1 - all on the device (training/validation.py:163-202 without the per-batch D2H)."""
    _f32(probs)
    n, k = probs.shape
    if labels is not None and (labels.dtype != torch.int32 or labels.numel() != n or not labels.is_cuda):
        raise TypeError('labels: int32 device tensor with one entry per row of probs')
    if conf is not None and (conf.dtype != torch.int64 or tuple(conf.shape) != (k, k) or not conf.is_contiguous()):
        raise TypeError('conf: contiguous (k,k) int64 device tensor')
    pred = torch.empty((n,), dtype=torch.int32, device=probs.device) if want_pred else None
    _lib.call('nimg_confusion_accumulate', _p(probs), _p(labels), _p(pred), _p(conf), n, k, _stream())
    return pred


def sigmoid(x, out=None):
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_sigmoid_fwd', _p(x), _p(y), x.numel(), _stream())
    return y


def sigmoid_bwd(dy, y, out=None):
    _f32(dy, y, out)
    dx = torch.empty_like(dy) if out is None else out
    _lib.call('nimg_sigmoid_bwd', _p(dy), _p(y), _p(dx), dy.numel(), _stream())
    return dx


def gamma_ste(x, lo=1.0 / 255, hi=1.0, exponent=1.0 / 2.2, out=None):
    """y = pow(clip_ste(x, lo, hi), exponent) - the gamma stage of ClassicISP (pipelines.py:449-451)."""
    _f32(x, out)
    y = torch.empty_like(x) if out is None else out
    _lib.call('nimg_gamma_ste_fwd', _p(x), _p(y), x.numel(), lo, hi, exponent, _stream())
    return y


def gamma_ste_bwd(x, dy, lo=1.0 / 255, hi=1.0, exponent=1.0 / 2.2, out=None):
    _f32(x, dy, out)
    dx = torch.empty_like(dy) if out is None else out
    _lib.call('nimg_gamma_ste_bwd', _p(x), _p(dy), _p(dx), dy.numel(), lo, hi, exponent, _stream())
    return dx


PAD_MODES = {'CONSTANT': 0, 'SYMMETRIC': 1, 'REFLECT': 2}


def pad2d(x, pad, mode='REFLECT'):
    """tf.pad on the two spatial axes; the backward pass is fold_pad(d, pad, PAD_MODES[mode])."""
    _f32(x)
    n, h, w, c = x.shape
    y = torch.empty((n, h + 2 * pad, w + 2 * pad, c), dtype=torch.float32, device=x.device)
    _lib.call('nimg_pad2d', _p(x), _p(y), n, h, w, c, pad, PAD_MODES[mode], _stream())
    return y
