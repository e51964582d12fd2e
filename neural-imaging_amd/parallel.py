"""
Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over xGMI.

The reference is single-process / single-device; data parallelism is build-side.  The channel shards over
raw patches with ONE exchange step: a sum all-reduce of each model's flat float32 gradient buffer (FAN 4.6 MB first -
its gradients are complete first in the backward pass - then UNet 31 MB), followed by the same fused Adam on every
rank with grad_scale = 1 / world_size (CE and MSE are means over the batch).  No other collective touches the data path.
"""
import os

import torch
import torch.distributed as dist


# A one-rank group has nothing to exchange, so the collectives are skipped at world size 1 - unless they are forced
# (force_collectives() / NIMG_DP_FORCE_COLLECTIVES=1): then every bucket launch, flag reduction, histogram all-reduce and
# broadcast goes through the backend exactly as at N > 1.  That is how the RCCL stream / event hand-off with the library's
# raw-stream launches is exercised on a one-GPU box (tests/test_gpu_models.py::test_data_parallel_step_nccl_world1,
# bench.py `dp1_nccl_*`).
_FORCE = os.environ.get('NIMG_DP_FORCE_COLLECTIVES', '0') == '1'


def force_collectives(on=True):
    global _FORCE
    _FORCE = bool(on)


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise the process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    ws = int(os.environ.get('WORLD_SIZE', '1'))
    if (ws <= 1 and not _FORCE) or (dist.is_available() and dist.is_initialized()):
        return ws
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if ws <= 1:                               # forced one-rank group: no launcher has set the rendezvous variables
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if 'MASTER_PORT' not in os.environ:
            # the port is OURS to pick: between _free_port() closing its probe socket and the store listening on the number,
            # another process can take it (seen on a GPU box: EADDRINUSE) - pick again instead of failing the run
            for attempt in range(8):
                os.environ['MASTER_PORT'] = str(_free_port())
                try:
                    dist.init_process_group(backend=backend)
                    return ws
                except Exception as e:                                    # noqa (DistNetworkError is a RuntimeError subclass)
                    in_use = 'EADDRINUSE' in str(e) or 'address already in use' in str(e).lower()
                    if not in_use or attempt == 7:
                        del os.environ['MASTER_PORT']
                        raise
    dist.init_process_group(backend=backend)
    return ws


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_generator(base_seed, device):
    """A device generator for this rank's random draws (awgn noise, dropout masks): every rank must draw DIFFERENT numbers
    for its shard (identical noise / masks on all ranks would correlate the shards of one global batch, SURVEY 8e caveat 3),
    and rank 0 of any world reproduces the single-process stream: seed = base_seed + 1000003 * rank."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(base_seed) + 1000003 * rank())
    return gen


class GradientBucket(object):
    """Asynchronous sum all-reduce of one flat gradient buffer; wait() before the optimiser touches it."""

    def __init__(self):
        self._work = []

    def launch(self, flat_grad):
        from . import ops
        ops.join_side_stream()               # parameter gradients may still be running on the side stream
        if is_distributed() and flat_grad.numel() > 0:
            self._work.append(dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        for w in self._work:
            w.wait()
        self._work = []


def sync_gradients(flat_grad):
    """Blocking sum all-reduce of one model's flat gradient buffer (the stand-alone training_step of a single model).
    Returns the world size: mean losses scale the Adam step by 1 / world, sum losses (tf.nn.l2_loss) by 1."""
    from . import ops
    ops.join_side_stream()
    if is_distributed() and flat_grad.numel() > 0:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return world_size()


def all_reduce_flag(flag):
    """OR-reduce the NaN flag over ranks (SURVEY 8e caveat 4)."""
    if is_distributed():
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)


def broadcast_floats(values, device=None):
    """Rank 0's host-side random draws for every rank (SURVEY 8e caveat 2: the augmentation strengths of a step are drawn once,
    on the host, for the whole global batch - workflows/...:205)."""
    values = [float(v) for v in values]
    if not is_distributed() or not values:
        return values
    dev = device if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor(values, dtype=torch.float64, device=dev)
    dist.broadcast(t, src=0)
    return t.cpu().tolist()


def shard_batch(batch, rank_, world):
    """Contiguous shard of the global batch for this rank (labels are positional per rank, workflows/...:257-258)."""
    n = batch.shape[0]
    if n % world:
        raise ValueError('a global batch of {} does not split over {} ranks'.format(n, world))
    per = n // world
    return batch[rank_ * per:(rank_ + 1) * per]
