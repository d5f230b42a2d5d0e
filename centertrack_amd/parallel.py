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


class DetectionGatherer(object):
    """The one exchange step of the sharded path: all-gather of the packed decode rows, with every buffer
    allocated ONCE (send [per,K,F], recv [world*per,K,F], the globally ordered result [num_streams,K,F]) so that a
    step enqueues exactly one copy-in, one ``all_gather_into_tensor`` and one reorder copy -- no allocation and no
    per-stream Python loop per frame.  Round-robin ownership (``shard_streams``) makes the reorder a transpose:
    global stream s = j*world + r is row j of rank r, i.e. ``out = recv.view(world, per, K, F).transpose(0, 1)``.
    ``steps`` / ``last`` let the caller verify afterwards what was exchanged (``checksums``)."""

    def __init__(self, num_streams, world, rank, K, F, device, dtype=torch.float32):
        self.num_streams, self.world, self.rank = num_streams, world, rank
        self.per = (num_streams + world - 1) // world
        self.n_local = len(shard_streams(num_streams, rank, world))
        self.send = torch.zeros((self.per, K, F), dtype=dtype, device=device)
        self.recv = torch.zeros((world * self.per, K, F), dtype=dtype, device=device)
        self.out = torch.zeros((world * self.per, K, F), dtype=dtype, device=device)
        self.steps = 0

    def __call__(self, local_rows):
        """``local_rows``: [n_local, K, F] rows of this rank's streams (ascending global id).  Returns the
        [num_streams, K, F] block ordered by global stream id (a view of a reused buffer)."""
        if local_rows.shape[0] != self.n_local:
            raise ValueError('rank %d owns %d streams, got %d row blocks' % (self.rank, self.n_local, local_rows.shape[0]))
        self.steps += 1
        if self.world == 1:
            self.out[:self.n_local].copy_(local_rows)
            return self.out[:self.num_streams]
        self.send[:self.n_local].copy_(local_rows)
        dist.all_gather_into_tensor(self.recv, self.send)
        K, F = self.send.shape[1:]
        self.out.view(self.per, self.world, K, F).copy_(self.recv.view(self.world, self.per, K, F).transpose(0, 1))
        return self.out[:self.num_streams]

    @staticmethod
    def _bits_sum(t, dims=None):
        """order-independent exact checksum: the float32 bit patterns summed as int64"""
        v = t.contiguous().view(torch.int32).to(torch.int64)
        return v.sum() if dims is None else v.sum(dim=dims)

    def checksums(self, local_rows):
        """(checksum of this rank's rows, checksums of every rank's block as it arrived here) -- the caller compares
        them across ranks (``verify``) to prove that the collective moved the real rows"""
        mine = int(self._bits_sum(local_rows).item())
        K, F = self.send.shape[1:]
        if self.world == 1:
            return mine, [int(self._bits_sum(self.out[:self.n_local]).item())]
        return mine, self._bits_sum(self.recv.view(self.world, self.per, K, F), (1, 2, 3)).tolist()

    def verify(self, local_rows):
        """After a step: every rank's block in the gathered result must carry the checksum that rank computed over
        its own rows (exchanged with one more small all-gather), and the reordered block must hold this rank's rows
        at its global stream ids.  Returns the number of ranks in the RCCL / gloo group; raises on a mismatch."""
        mine, blocks = self.checksums(local_rows)
        ids = shard_streams(self.num_streams, self.rank, self.world)
        if ids and not torch.equal(self.out[ids], local_rows.to(self.out.dtype)):
            raise RuntimeError('rank %d: its own rows are not at their global stream ids in the gathered block' % self.rank)
        if self.world == 1:
            if blocks[0] != mine:
                raise RuntimeError('gathered block differs from the local rows')
            return 1
        dev = self.send.device
        sums = torch.zeros(self.world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sums, torch.tensor([mine], dtype=torch.int64, device=dev))
        for r, (want, got) in enumerate(zip(sums.tolist(), blocks)):
            if want != got:
                raise RuntimeError('rank %d: block of rank %d arrived with checksum %r, its owner computed %r'
                                   % (self.rank, r, got, want))
        return dist.get_world_size()


def gather_detections(local_rows, num_streams, world, rank):
    """One-shot form of ``DetectionGatherer`` (allocates; tests / tools): ``local_rows`` [B_local, K, F] of the
    streams returned by ``shard_streams`` -> [num_streams, K, F] ordered by global stream id, on every rank."""
    if world == 1:
        return local_rows
    g = DetectionGatherer(num_streams, world, rank, local_rows.shape[1], local_rows.shape[2], local_rows.device,
                          local_rows.dtype)
    return g(local_rows).clone()


def run_steps(det, frame_of, metas, steps, frames_per_step, first=0):
    """The body of bench.py's loops (warm-up and timed alike): ``steps`` steps of ``frames_per_step`` consecutive
    frames of every stream the rank owns -- frame t is ``frame_of(t)`` (what ``det.step`` accepts: a host or device
    [B,3,H,W] tensor) -- through ``det.step``; a ``det.gather_fn`` hook (the all-gather of the packed rows) runs
    inside ``det.step``.  Returns (frames processed per stream, detections returned)."""
    t = first
    ndet = 0
    ahead = bool(getattr(det, 'supports_prefetch', False))    # upload frame t+1 while frame t is computed
    last = first + steps * frames_per_step - 1
    for _ in range(steps):
        for _ in range(frames_per_step):
            if ahead:
                res = det.step(frame_of(t), metas, prefetch=frame_of(t + 1) if t < last else None)
            else:
                res = det.step(frame_of(t), metas)
            ndet += sum(len(r) for r in res)
            t += 1
    return t - first, ndet


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
