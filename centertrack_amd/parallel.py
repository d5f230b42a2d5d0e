"""Multi-GPU execution of the hot path: one process per GPU, streams sharded by batch.

The reference has no distributed inference; frames of different
videos are independent while frames of one video are strictly sequential, so the path
shards by *stream*: rank r owns streams {s : s mod G == r}, holds a full weight replica
(79.8 MB) and its streams' tracker state.  There is no activation exchange; the only
collective is ONE all-gather per step of the packed decode rows ([K, F] floats per
stream, 4-10 KB) over RCCL/xGMI (torch.distributed backend "nccl"), latency-bound by
construction.  On CPU the same code runs over gloo (tests, world_size 2)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank).
    Single-process runs (no RANK in the environment) return (0, 1, 0) without a process group."""
    if 'RANK' not in os.environ or int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return 0, 1, int(os.environ.get('LOCAL_RANK', 0))
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_streams(num_streams, rank, world):
    """Global stream ids owned by ``rank`` (round-robin, so G=1,2,4,8 all balance)."""
    return [s for s in range(num_streams) if s % world == rank]


def gather_detections(local_rows, num_streams, world, rank):
    """All-gather the packed decode rows.  ``local_rows``: [B_local, K, F] tensor of the
    streams returned by ``shard_streams`` (same K, F on every rank; B_local may differ by
    one, shorter shards are zero-padded for the collective).  Returns [num_streams, K, F]
    ordered by global stream id, on every rank."""
    if world == 1:
        return local_rows
    per = (num_streams + world - 1) // world
    K, F = local_rows.shape[1], local_rows.shape[2]
    send = local_rows.new_zeros((per, K, F))
    send[:local_rows.shape[0]] = local_rows
    recv = local_rows.new_empty((world * per, K, F))
    dist.all_gather_into_tensor(recv, send)
    out = local_rows.new_empty((num_streams, K, F))
    recv = recv.view(world, per, K, F)
    for r in range(world):
        ids = shard_streams(num_streams, r, world)
        if ids:
            out[ids] = recv[r, :len(ids)]
    return out


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value):
    """max of a python float over ranks (used for the bench's timed region)."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    """tear the process group down (end of a torchrun job): avoids the NCCL/RCCL "process group was not
    destroyed" teardown path"""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
