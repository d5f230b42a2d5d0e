"""Multi-GPU execution of the hot path: one process per GPU, streams sharded by batch.

The reference has no distributed inference; frames of different
videos are independent while frames of one video are strictly sequential, so the path
shards by *stream*: rank r owns streams {s : s mod G == r}, holds a full weight replica
(79.8 MB) and its streams' tracker state.  There is no activation exchange; the only
collective is ONE all-gather per step of the packed decode rows ([K, F] floats per
stream, 4-10 KB) over RCCL/xGMI (torch.distributed backend "nccl"), latency-bound by
construction.  On CPU the same code runs over gloo (tests, world_size 2).

Round 3 (multi-GPU readiness): ranks prove at start-up that they built IDENTICAL launch plans
(``check_same_plan``: tile shapes, split-K and DCN schedule knobs fix the fp32 summation order, so a rank with a stale
tuning cache would silently produce other last bits); the all-gather runs on a SIDE stream and lands in pinned host
memory, so it overlaps the next frame's launch and host work instead of sitting between graph launch and stream sync;
the gathered block is CONSUMED every step (``DetectionGatherer.consume``: detections of all streams counted on the host
from the block of the previous step), and the count is cross-checked against the ranks' own counts after the loop."""
import hashlib
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment; returns (rank, world, local_rank).
    Single-process runs (no RANK in the environment) return (0, 1, 0) without a process group.  A torchrun job of ONE
    rank (`--nproc-per-node 1`) does get its group: the collectives of the sharded path then really go through RCCL
    (``group_active``), which is how a single-GPU box exercises them (tests/test_hip_rccl.py)."""
    if 'RANK' not in os.environ:
        return 0, 1, int(os.environ.get('LOCAL_RANK', 0))
    rank, world = int(os.environ['RANK']), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', rank))
    if world == 1 and not ('MASTER_ADDR' in os.environ and 'MASTER_PORT' in os.environ) and not dist.is_initialized():
        # RANK exported by something that is not a launcher (a scheduler's task index): a plain single-process run --
        # the one-rank group is only created when torchrun's rendezvous variables are there
        return 0, 1, local
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def group_active():
    """a torch.distributed process group exists (any world size): the exchange step goes through its collectives"""
    return dist.is_available() and dist.is_initialized()


def shard_streams(num_streams, rank, world):
    """Global stream ids owned by ``rank`` (round-robin, so G=1,2,4,8 all balance)."""
    return [s for s in range(num_streams) if s % world == rank]


def check_same_plan(signature, what='launch plan'):
    """Every rank hashes ``signature`` (a string that pins the fp32 summation order of its plan: per launch the tile
    shape / algo / split-K, the DCN schedule knobs -- ``DLASegHIP.plan_signature``); the hashes are all-gathered and a
    mismatch raises on EVERY rank, naming the ranks that differ from rank 0.  Returns the hex digest."""
    digest = hashlib.sha256(signature.encode()).digest()
    hexd = digest[:8].hex()
    if not group_active():
        return hexd
    world = dist.get_world_size()
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    mine = torch.tensor([int.from_bytes(digest[0:7], 'little'), int.from_bytes(digest[7:14], 'little')],
                        dtype=torch.int64, device=dev)
    allh = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allh, mine)
    allh = allh.view(world, 2).tolist()
    bad = [r for r in range(world) if allh[r] != allh[0]]
    if bad:
        raise RuntimeError('%s differs between ranks: rank(s) %s built another plan than rank 0 (stale '
                           'CENTERTRACK_TUNE_CACHE / CENTERTRACK_DCN_KNOBS on that rank?); identical fp32 summation '
                           'orders on every rank need identical plans' % (what, bad))
    return hexd


def build_plans_consistently(build_fn):
    """Run ``build_fn`` (whatever builds this rank's launch plans, e.g. ``det._context(H, W)``) so that every rank ends
    up with rank 0's launch-shape choices: rank 0 builds first (timing whatever the pinned table does not hold), its
    tuning cache is broadcast, then the other ranks build from it without timing anything.  Returns build_fn()'s
    result.  ``check_same_plan`` afterwards proves the outcome."""
    from . import autotune
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return build_fn()
    out = build_fn() if dist.get_rank() == 0 else None
    autotune.share_from_rank0()
    if dist.get_rank() != 0:
        out = build_fn()
    return out


class DetectionGatherer(object):
    """The one exchange step of the sharded path: all-gather of the packed decode rows, with every buffer
    allocated ONCE (send [per,K,F], recv [world*per,K,F], the globally ordered result [num_streams,K,F]) so that a
    step enqueues exactly one copy-in, one ``all_gather_into_tensor`` and one reorder copy -- no allocation and no
    per-stream Python loop per frame.  Round-robin ownership (``shard_streams``) makes the reorder a transpose:
    global stream s = j*world + r is row j of rank r, i.e. ``out = recv.view(world, per, K, F).transpose(0, 1)``.
    ``steps`` / ``last`` let the caller verify afterwards what was exchanged (``checksums``).

    ``overlap`` (CUDA only): the three operations are enqueued on a side stream behind an event of the producing
    stream, the reorder copy goes straight to PINNED HOST memory (two blocks, ping-pong), and ``__call__`` returns
    immediately: the collective of frame t runs beside the host work and the launch of frame t+1.  The producer must
    not overwrite ``local_rows`` before the copy-in ran: ``rows_free`` is the event to wait for (StreamDetector does,
    before its next graph launch).  ``consume()`` hands out the block of the PREVIOUS call (complete by then) as a
    numpy view."""

    def __init__(self, num_streams, world, rank, K, F, device, dtype=torch.float32, overlap=False):
        self.num_streams, self.world, self.rank = num_streams, world, rank
        self.per = (num_streams + world - 1) // world
        self.n_local = len(shard_streams(num_streams, rank, world))
        device = torch.device(device)
        self.send = torch.zeros((self.per, K, F), dtype=dtype, device=device)
        self.recv = torch.zeros((world * self.per, K, F), dtype=dtype, device=device)
        self.out = torch.zeros((world * self.per, K, F), dtype=dtype, device=device)
        self.steps = 0
        # with a process group the exchange is a real collective even for a group of one rank (RCCL on a one-GPU box);
        # without one (plain `python bench.py`) the rows are only copied
        self.collective = world > 1 or group_active()
        if self.collective:
            if not group_active():
                raise RuntimeError('DetectionGatherer(world=%d) needs an initialised torch.distributed group' % world)
            if dist.get_world_size() != world:
                # (recv is sized for `world` ranks: a collective over another group size would fail inside RCCL with a
                # shape error -- say what is wrong instead)
                raise ValueError('DetectionGatherer(world=%d) inside a process group of %d ranks: pass the group\'s '
                                 'world size' % (world, dist.get_world_size()))
        self.overlap = bool(overlap) and device.type == 'cuda'
        self.consumed_steps = 0
        self.consumed_detections = 0
        self._pending = None                      # index of the host block the last call is filling
        if self.overlap:
            self.side = torch.cuda.Stream(device=device)
            self.host = [torch.zeros((world * self.per, K, F), dtype=dtype).pin_memory() for _ in range(2)]
            self.done = [torch.cuda.Event(), torch.cuda.Event()]
            self.rows_ready = torch.cuda.Event()
            self.rows_free = torch.cuda.Event()
        else:
            self.host = [None, None]
            self.rows_free = None

    def __call__(self, local_rows):
        """``local_rows``: [n_local, K, F] rows of this rank's streams (ascending global id).  Returns the
        [num_streams, K, F] block ordered by global stream id (a view of a reused buffer; with ``overlap`` the view is
        only valid once ``wait()`` returned)."""
        if local_rows.shape[0] != self.n_local:
            raise ValueError('rank %d owns %d streams, got %d row blocks' % (self.rank, self.n_local, local_rows.shape[0]))
        self.steps += 1
        K, F = self.send.shape[1:]
        if not self.overlap:
            if not self.collective:
                self.out[:self.n_local].copy_(local_rows)
            else:
                self.send[:self.n_local].copy_(local_rows)
                dist.all_gather_into_tensor(self.recv, self.send)
                self.out.view(self.per, self.world, K, F).copy_(self.recv.view(self.world, self.per, K, F).transpose(0, 1))
            self._pending = -1
            return self.out[:self.num_streams]
        cur = torch.cuda.current_stream()
        self.rows_ready.record(cur)
        slot = self.steps & 1
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.rows_ready)
            if not self.collective:
                self.out[:self.n_local].copy_(local_rows, non_blocking=True)
                self.rows_free.record(self.side)
            else:
                self.send[:self.n_local].copy_(local_rows, non_blocking=True)
                self.rows_free.record(self.side)
                dist.all_gather_into_tensor(self.recv, self.send)
                self.out.view(self.per, self.world, K, F).copy_(self.recv.view(self.world, self.per, K, F).transpose(0, 1))
            self.host[slot].copy_(self.out, non_blocking=True)
            self.done[slot].record(self.side)
        self._pending = slot
        return self.out[:self.num_streams]

    def wait(self):
        """block the host until the last enqueued gather (and its host copy) finished"""
        if self.overlap and self._pending is not None and self._pending >= 0:
            self.done[self._pending].synchronize()

    def host_block(self):
        """the block of the last call as numpy [num_streams, K, F] (host-synchronous: waits for that call)"""
        if self._pending is None:
            return None
        if self.overlap:
            self.wait()
            return self.host[self._pending].numpy()[:self.num_streams]
        return self.out[:self.num_streams].cpu().numpy()

    def consume(self, score_col=0, thresh=0.0):
        """Use the block of the last call on the host: count the detections of ALL streams whose score exceeds
        ``thresh`` (what a rank-0 result writer would keep).  Called right before the next step's gather is enqueued,
        i.e. one step after its collective: the wait is a no-op in steady state.  Returns the count."""
        blk = self.host_block()
        if blk is None:
            return 0
        n = int((blk[:, :, score_col] > thresh).sum())
        self.consumed_steps += 1
        self.consumed_detections += n
        self._pending = None
        return n

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
        if not self.collective:
            return mine, [int(self._bits_sum(self.out[:self.n_local]).item())]
        return mine, self._bits_sum(self.recv.view(self.world, self.per, K, F), (1, 2, 3)).tolist()

    def verify(self, local_rows):
        """After a step: every rank's block in the gathered result must carry the checksum that rank computed over
        its own rows (exchanged with one more small all-gather), and the reordered block must hold this rank's rows
        at its global stream ids.  Returns the number of ranks in the RCCL / gloo group; raises on a mismatch."""
        self.wait()
        if self.overlap:
            torch.cuda.current_stream().wait_stream(self.side)
        mine, blocks = self.checksums(local_rows)
        ids = shard_streams(self.num_streams, self.rank, self.world)
        if ids and not torch.equal(self.out[ids], local_rows.to(self.out.dtype)):
            raise RuntimeError('rank %d: its own rows are not at their global stream ids in the gathered block' % self.rank)
        if not self.collective:
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
                # (frames of one video share their meta: the promise lets the native loop launch frame t+1 in the call
                # that finishes frame t)
                res = det.step(frame_of(t), metas, prefetch=frame_of(t + 1) if t < last else None, prefetch_metas=metas)
            else:
                res = det.step(frame_of(t), metas)
            ndet += sum(len(r) for r in res)
            t += 1
    return t - first, ndet


def sum_over_ranks(value):
    """sum of a python int over ranks"""
    if not (dist.is_available() and dist.is_initialized()):
        return int(value)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


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
