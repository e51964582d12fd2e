"""
hipGraph capture of the channel's training step.

A training step is a fixed sequence of ~240 kernel launches (explicit forward / backward, no tape, no host decisions between
launches when nan_check='deferred'), issued through Python + ctypes.  CapturedStep records that sequence ONCE into a HIP graph
(torch.cuda.CUDAGraph drives hipStreamBeginCapture on torch's stream; the library's launches land on that stream, the
weight-gradient side stream is forked / joined inside the capture) and replays it with one call per step.  The only per-step
host input is Keras Adam's bias-corrected rate lr * sqrt(1 - b2^t) / (1 - b1^t): it lives in a one-float device buffer that is
refreshed before every replay (nimg_adam_step_dev reads it), so the captured launches never change.

The graph holds raw addresses.  Buffers allocated inside the capture belong to the graph's private pool; the ones that live
outside it (ops.Workspace scratch, cached filter tables) are pinned by the CapturedStep (ops.begin_pin_log / end_pin_log), so
eager calls that follow - validation at another shape, a second captured step - can re-grow or evict them safely.

Restrictions: fixed batch shape and hyper-parameters (lambda_*, manipulation strengths: augment=False), single process (the
RCCL bucket launches are not captured), nan_check='deferred'.  The private generators of an 'awgn' manipulation and of the FAN's
dropout are registered with the graph (their offsets advance per replay); injected noise / masks cannot be captured.
"""
import torch

from . import ops, parallel


class CapturedStep(object):

    def __init__(self, flow, batch_x, batch_y, learning_rate=1e-4, warmup=3, **kw):
        if parallel.is_distributed():
            raise RuntimeError('CapturedStep: the data-parallel step is not captured (RCCL launches stay eager)')
        if flow._nan_check != 'deferred' or kw.get('augment'):
            raise ValueError('CapturedStep needs nan_check="deferred" and augment=False (no host decisions inside a step)')
        self.flow, self.lr, self.kw = flow, float(learning_rate), kw
        dev = flow.device
        self.x = batch_x.detach().to(dev).clone().contiguous()          # static inputs: refill with load()
        self.y = batch_y.detach().to(dev).clone().contiguous()
        self._rate_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        # warm-up on a side stream (allocator / workspace growth, lazy module state), as torch's capture rules ask
        ops.begin_pin_log()                   # every workspace / cached table the step touches from here on is pinned below
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(max(int(warmup), 1)):
                flow.training_step(self.x, self.y, learning_rate=self.lr, **kw)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.t = flow._step
        flow._lr_t_dev = self._rate_dev
        self.graph = torch.cuda.CUDAGraph()
        # private random streams (awgn noise: helpers/tf_helpers.Awgn._gen; dropout masks: FAN._dropout_gen) were created by the
        # warm-up steps; a capture may only draw from generators whose state the graph owns
        for gen in [getattr(op, '_gen', None) for op in getattr(flow, '_operations', {}).values()] + \
                [getattr(getattr(flow, 'fan', None), '_dropout_gen', None)]:
            if isinstance(gen, torch.Generator):
                self.graph.register_generator_state(gen)
        try:
            self._set_rate(self.t + 1)
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.out = flow.training_step(self.x, self.y, learning_rate=self.lr, **kw)
        finally:
            flow._lr_t_dev = None
            # strong references to the external buffers whose addresses the graph holds (ops.Workspace scratch incl. the side
            # stream's, filter-tap tables, the learned codec's latent workspace): eager calls after the capture may re-grow or
            # evict them, the graph keeps replaying on these
            self._pins = ops.end_pin_log()
            lws = getattr(getattr(flow, 'codec', None), '_lws', None)
            if lws is not None:
                self._pins.append(lws)
            self._pins.append(dict(flow._labels_cache))
        flow._step = self.t                 # the capture itself executed nothing

    def _set_rate(self, t):
        ops.float_fill(self._rate_dev, ops.adam_lr_t(self.lr, t))     # a scalar-argument fill launch, ordered before the replay

    def load(self, batch_x, batch_y):
        """Next batch into the captured step's input buffers (device-to-device or pinned host-to-device copy)."""
        self.x.copy_(batch_x, non_blocking=True)
        self.y.copy_(batch_y, non_blocking=True)

    def step(self):
        self.t += 1
        self._set_rate(self.t)
        self.graph.replay()
        self.flow._step = self.t
        return self.out


class CapturedModelStep(object):
    """The stand-alone training step of ONE model - NIPModel.training_step(x, y) (train_nip.py -> training/pipeline.py:191-247),
    DCN.training_step(x) (train_dcn.py), FAN.training_step(x, labels) - recorded into a HIP graph and replayed.  At the
    reference's own batch sizes these steps are ~150 launches of 5 - 40 us: the host needs ~8 us per launch through Python +
    ctypes, so the eager step is bound by the HOST (config 2: 1.8 ms per step for ~0.9 ms of kernel time); the replay is one call.

    inputs: the positional batch tensors of training_step (device tensors; refill with load()).  kwargs go to training_step
    unchanged (DCN: sync=False is forced - the captured step cannot read losses back).  Same restrictions as CapturedStep."""

    def __init__(self, model, *inputs, learning_rate=1e-4, warmup=3, **kw):
        if parallel.is_distributed():
            raise RuntimeError('CapturedModelStep: the data-parallel step is not captured (RCCL launches stay eager)')
        import inspect
        if 'sync' in inspect.signature(model.training_step).parameters:
            kw['sync'] = False
        self.model, self.lr, self.kw = model, float(learning_rate), kw
        dev = model.device
        self.inputs = [t.detach().to(dev).clone().contiguous() for t in inputs]
        self._rate_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        store = model._model
        ops.begin_pin_log()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(max(int(warmup), 1)):
                model.training_step(*self.inputs, learning_rate=self.lr, **kw)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.t = store.step
        store.lr_t_dev = self._rate_dev
        self.graph = torch.cuda.CUDAGraph()
        try:
            self._set_rate(self.t + 1)
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                self.out = model.training_step(*self.inputs, learning_rate=self.lr, **kw)
        finally:
            store.lr_t_dev = None
            self._pins = ops.end_pin_log()
            lws = getattr(model, '_lws', None)
            if lws is not None:
                self._pins.append(lws)
        store.step = self.t                   # the capture itself executed nothing

    def _set_rate(self, t):
        ops.float_fill(self._rate_dev, ops.adam_lr_t(self.lr, t))

    def load(self, *inputs):
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src, non_blocking=True)

    def step(self):
        self.t += 1
        self._set_rate(self.t)
        self.graph.replay()
        self.model._model.step = self.t
        return self.out


class RunAhead(object):
    """Bounds how far the launching thread runs ahead of the GPU in an eager-launch loop: call it after every step; it waits for the
    step issued `depth` steps earlier.  Without a bound the host (~3 ms of launches per 8 ms step) gets ~16 steps ahead, every step
    in flight holds its temporaries, and the caching allocator has to hipMalloc new segments in the middle of the timed loop - a
    0.3 - 0.75 s stall on boxes where hipMalloc is slow (profiles/r04_zc_stall_probe.txt: always at step 16, inside torch.empty);
    a training loop that reads a loss every step is paced by that read and does not need it.
    Two or three steps in flight keep the GPU fed and the pool at its warm-up size."""

    def __init__(self, depth=3):
        self.depth, self.events = depth, []

    def __call__(self):
        e = torch.cuda.Event()
        e.record()
        self.events.append(e)
        if len(self.events) > self.depth:
            self.events.pop(0).synchronize()
