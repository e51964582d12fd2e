"""Worker for the 2-rank gloo test of the data-parallel plumbing (spawned processes import this module first)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def dp_worker(rank, world, port, q):
    import torch
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': str(rank)})
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import parallel as par
    assert par.init_from_env('gloo') == world and par.world_size() == world and par.rank() == rank
    flat = torch.full((1000,), float(rank + 1))
    bucket = par.GradientBucket()
    for lo, hi in ((0, 100), (100, 600), (600, 1000)):              # three buckets in flight, like the channel's FAN / decoder / encoder
        bucket.launch(flat[lo:hi])
    bucket.wait()
    flag = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
    par.all_reduce_flag(flag)
    batch = torch.arange(4 * world).reshape(4 * world, 1)
    shard = par.shard_batch(batch, rank, world)
    drawn = par.broadcast_floats([0.25 + rank, 7.0 * (rank + 1)])
    # this rank's random stream (awgn noise / dropout masks): through the model that draws it, on the CPU device here
    from neural_imaging_amd.helpers import tf_helpers
    awgn = tf_helpers.Awgn(seed=5)
    gen = par.rank_generator(5, torch.device('cpu'))
    noise = torch.randn((6,), generator=gen)
    q.put((rank, [float(flat[0]), float(flat[599]), float(flat[999])], int(flag[0]), shard.flatten().tolist(), drawn,
           noise.tolist(), awgn._seed))
    torch.distributed.destroy_process_group()


def make_channel(codec, dev, nan_check='deferred'):
    """The channel of the data-parallel step test: UNet -> 3 manipulations -> (dJPEG | TwitterDCN) -> FAN at raw 32x32."""
    from neural_imaging_amd.workflows.manipulation_classification import ManipulationClassification
    manips = ['sharpen:1', 'resample:50', 'gaussian:0.83']
    if codec == 'dcn':
        from neural_imaging_amd.models import compression
        dcn = compression.TwitterDCN(patch_size=64, device=dev)
        dist = {'downsampling': 'none', 'compression': 'dcn', 'compression_params': {'model': dcn}}
        return ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip', 'dcn'},
                                          raw_patch_size=32, device=dev, nan_check=nan_check), dict(lambda_nip=0.1, lambda_dcn=0.01)
    dist = {'downsampling': 'none', 'compression': 'jpeg', 'compression_params': {'quality': 80, 'codec': 'soft'}}
    return ManipulationClassification('UNet', manipulations=manips, distribution=dist, trainable={'nip'},
                                      raw_patch_size=32, device=dev, nan_check=nan_check), dict(lambda_nip=0.1)


def channel_state(wf):
    """(flat gradients, flat parameters) of every trainable model, on the host."""
    models = [wf.fan, wf.nip] + ([wf.codec] if wf.is_trainable('dcn') else [])
    return ([m._model.flat_grad.detach().cpu().numpy().copy() for m in models],
            [m._model.flat.detach().cpu().numpy().copy() for m in models])


def dp_step_worker(rank, world, port, q, codec, raw, rgb):
    """One data-parallel training step of the channel on this rank's contiguous shard of the global batch (every rank drives
    cuda:0; gloo carries the collectives, so the test runs on a one-GPU box)."""
    import torch
    os.environ.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'RANK': str(rank),
                       'WORLD_SIZE': str(world), 'LOCAL_RANK': '0'})
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, parallel as par
    _lib.load()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    assert par.init_from_env('gloo') == world
    wf, kw = make_channel(codec, dev)
    bx = par.shard_batch(torch.from_numpy(raw), rank, world)
    by = par.shard_batch(torch.from_numpy(rgb), rank, world)
    loss, parts = wf.training_step(bx, by, learning_rate=1e-4, **kw)
    wf.check_nan()
    grads, params = channel_state(wf)
    q.put((rank, float(parts['ce']), float(parts['nip']), float(parts['dcn']), grads, params))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def dp_nccl_world1_worker(q, codec, raw, rgb, mode):
    """The data-parallel code path through RCCL on ONE GPU: a one-rank `nccl` group with the collectives forced
    (parallel.force_collectives) - three gradient buckets, NaN-flag reduction, the codec's histogram all-reduce all go through
    ProcessGroupNCCL and its stream / event hand-off with the library's raw-stream launches - next to the plain step of an
    identically initialised channel in the same process."""
    import torch
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import _lib, ops, parallel as par
    _lib.load()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    ops.set_compute(mode)
    bx, by = torch.from_numpy(raw).to(dev), torch.from_numpy(rgb).to(dev)
    out = {}
    for leg in ('plain', 'nccl'):
        if leg == 'nccl':
            par.force_collectives(True)
            os.environ.pop('WORLD_SIZE', None)
            par.init_from_env('nccl')
            assert par.is_distributed() and par.world_size() == 1 and torch.distributed.get_backend() == 'nccl'
        wf, kw = make_channel(codec, dev)
        losses = []
        for _ in range(3):                       # three steps: the second and third run on all-reduced, Adam-updated weights
            loss, parts = wf.training_step(bx, by, learning_rate=1e-4, **kw)
            losses.append((float(parts['ce']), float(parts['nip']), float(parts['dcn'])))
        wf.check_nan()
        out[leg] = (losses,) + channel_state(wf)
        del wf
    q.put(out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def forced_world1_worker(q):
    import torch
    importlib.import_module('neural-imaging_amd')
    from neural_imaging_amd import parallel as par
    for k in ('RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        os.environ.pop(k, None)
    assert par.init_from_env('gloo') == 1 and not par.is_distributed()          # a plain one-rank run starts no group
    par.force_collectives(True)
    par.init_from_env('gloo')
    ok = par.is_distributed() and par.world_size() == 1 and par.rank() == 0
    flat = torch.arange(10, dtype=torch.float32)
    b = par.GradientBucket()
    b.launch(flat[:4])
    b.launch(flat[4:])
    b.wait()
    flag = torch.tensor([1], dtype=torch.int32)
    par.all_reduce_flag(flag)
    q.put((ok, flat.tolist(), int(flag[0]), par.broadcast_floats([0.5, 2.0]), par.sync_gradients(flat)))
    torch.distributed.destroy_process_group()
