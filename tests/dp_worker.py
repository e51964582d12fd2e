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
    bucket.launch(flat)
    bucket.wait()
    flag = torch.tensor([1 if rank == 1 else 0], dtype=torch.int32)
    par.all_reduce_flag(flag)
    batch = torch.arange(8).reshape(8, 1)
    shard = par.shard_batch(batch, rank, world)
    drawn = par.broadcast_floats([0.25 + rank, 7.0 * (rank + 1)])
    q.put((rank, float(flat[0]), int(flag[0]), shard.flatten().tolist(), drawn))
    torch.distributed.destroy_process_group()
