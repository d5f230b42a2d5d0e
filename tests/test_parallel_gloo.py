"""Multi-process (world_size 2, gloo, CPU) tests of the sharded path of centertrack_amd.parallel:
streams are partitioned round-robin, every rank computes only its own streams, and ONE
all-gather of the packed decode rows reproduces exactly what a single process holds."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(num_streams, K, F):
    g = torch.Generator().manual_seed(99)
    return torch.rand((num_streams, K, F), generator=g)


def _worker(rank, world, port, num_streams, K, F, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from centertrack_amd import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    full = _rows(num_streams, K, F)
    mine = parallel.shard_streams(num_streams, rank, world)
    local = full[mine] if mine else full[:0]
    out = parallel.gather_detections(local, num_streams, world, rank)
    t = parallel.max_over_ranks(1.0 + rank)
    parallel.barrier()
    q.put((rank, mine, out.numpy(), t))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('num_streams', [1, 4, 5])
def test_all_gather_of_sharded_streams_equals_single_process(num_streams):
    world, K, F = 2, 7, 14
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, num_streams, K, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _rows(num_streams, K, F).numpy()
    owned = []
    for rank, mine, out, t in got:
        np.testing.assert_array_equal(out, want)            # every rank ends with all streams, in global order
        assert t == 2.0                                      # max over ranks of (1 + rank)
        owned += mine
    assert sorted(owned) == list(range(num_streams))         # a partition: no stream on two ranks


def test_shard_streams_is_a_balanced_partition():
    from centertrack_amd import parallel
    for world in (1, 2, 4, 8):
        for n in (1, 4, 16, 32, 33):
            shards = [parallel.shard_streams(n, r, world) for r in range(world)]
            assert sorted(sum(shards, [])) == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_single_process_is_identity():
    from centertrack_amd import parallel
    x = _rows(3, 5, 10)
    assert parallel.gather_detections(x, 3, 1, 0) is x
    assert parallel.max_over_ranks(3.5) == 3.5


# --------------------------------------------------------------------------------------------------
# the detector-side hook + bench.py's loop body (parallel.run_steps) over a 2-rank gloo group
class _FakeDetector(object):
    """Stands in for StreamDetector on a GPU-less host: ``step`` produces this rank's packed decode rows for frame
    t (a deterministic function of (global stream id, t, frame content)) and, like StreamDetector.step, hands the
    device rows to ``gather_fn`` before returning per-stream results."""

    def __init__(self, stream_ids, K, F):
        self.ids, self.K, self.F = stream_ids, K, F
        self.gather_fn = None
        self.t = 0

    @staticmethod
    def rows_of(sid, t, K, F, scale):
        g = torch.Generator().manual_seed(1000 * sid + t)
        return torch.rand((K, F), generator=g) * scale

    def step(self, frame, metas):
        scale = float(frame.mean())
        rows = torch.stack([self.rows_of(s, self.t, self.K, self.F, scale) for s in self.ids]) if self.ids \
            else torch.zeros((0, self.K, self.F))
        if self.gather_fn is not None:
            self.gather_fn(rows)
        self.t += 1
        return [[0] * (s + 1) for s in self.ids]           # "detections": s+1 per stream


def _bench_loop_worker(rank, world, port, num_streams, K, F, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from centertrack_amd import parallel
    parallel.init_from_env(backend='gloo')
    ids = parallel.shard_streams(num_streams, rank, world)
    det = _FakeDetector(ids, K, F)
    gatherer = parallel.DetectionGatherer(num_streams, world, rank, K, F, torch.device('cpu'))
    last = {}

    def gather(rows):                                      # exactly bench.py's hook
        last['rows'] = rows
        last['all'] = gatherer(rows)
    det.gather_fn = gather
    frames = [torch.full((len(ids), 3, 4, 4), 1.0 + t) for t in range(3)]
    bufs = (gatherer.send.data_ptr(), gatherer.recv.data_ptr(), gatherer.out.data_ptr())
    nfr, ndet = parallel.run_steps(det, lambda t: frames[t % 3], [None] * len(ids), steps=2, frames_per_step=3)
    ranks = gatherer.verify(last['rows'])
    assert bufs == (gatherer.send.data_ptr(), gatherer.recv.data_ptr(), gatherer.out.data_ptr())   # no re-allocation
    tampered = False
    if world > 1:
        gatherer.recv[0, 0, 0] += 1.0                      # a corrupted block must be caught
        try:
            gatherer.verify(last['rows'])
        except RuntimeError:
            tampered = True
    parallel.barrier()
    q.put((rank, nfr, ndet, ranks, gatherer.steps, last['all'].clone().numpy(), tampered))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('num_streams', [4, 5])
def test_bench_loop_body_gathers_every_frame_through_the_detector_hook(num_streams):
    world, K, F = 2, 5, 14
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_loop_worker, args=(r, world, port, num_streams, K, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t_last = 5                                             # 2 steps x 3 frames: the last frame is t = 5, content 1 + 5 % 3
    want = torch.stack([_FakeDetector.rows_of(s, t_last, K, F, 1.0 + t_last % 3) for s in range(num_streams)]).numpy()
    for rank, nfr, ndet, ranks, steps, allrows, tampered in got:
        assert nfr == 6 and steps == 6 and ranks == world and tampered
        mine = [s for s in range(num_streams) if s % world == rank]
        assert ndet == 6 * sum(s + 1 for s in mine)
        np.testing.assert_array_equal(allrows, want)       # global stream order, every rank


def test_gatherer_single_rank_verifies_and_reuses_buffers():
    from centertrack_amd import parallel
    g = parallel.DetectionGatherer(3, 1, 0, 5, 10, torch.device('cpu'))
    x = _rows(3, 5, 10)
    out = g(x)
    assert torch.equal(out, x) and g.verify(x) == 1
    p = out.data_ptr()
    assert g(x * 2).data_ptr() == p
    with pytest.raises(RuntimeError):
        g.verify(x)                                        # the block now holds 2x: checksum mismatch
    with pytest.raises(ValueError):
        g(x[:2])


# --------------------------------------------------------------------------------------------------
# round 3: plan-consistency check across ranks, the gathered block consumed inside the loop
def _plan_check_worker(rank, world, port, same, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from centertrack_amd import parallel
    parallel.init_from_env(backend='gloo')
    sig = 'dcn_knobs=(128, 4, 4, 1);level0:conv:104:1' if (same or rank == 0) else 'dcn_knobs=(0, 8, 2, 2);level0:conv:104:1'
    try:
        q.put((rank, 'ok', parallel.check_same_plan(sig)))
    except RuntimeError as e:
        q.put((rank, 'raised', str(e)))
    parallel.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('same', [True, False])
def test_ranks_compare_their_launch_plans_before_they_start(same):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_check_worker, args=(r, world, port, same, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if same:
        assert [g[1] for g in got] == ['ok', 'ok'] and got[0][2] == got[1][2]
    else:                                                  # EVERY rank raises, and names the rank that differs
        assert [g[1] for g in got] == ['raised', 'raised']
        assert all('rank(s) [1]' in g[2] for g in got)


def test_plan_check_single_process_returns_a_digest():
    from centertrack_amd import parallel
    a, b = parallel.check_same_plan('x'), parallel.check_same_plan('y')
    assert a != b and len(a) == 16


def _consume_worker(rank, world, port, num_streams, K, F, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from centertrack_amd import parallel
    parallel.init_from_env(backend='gloo')
    ids = parallel.shard_streams(num_streams, rank, world)
    det = _FakeDetector(ids, K, F)
    gatherer = parallel.DetectionGatherer(num_streams, world, rank, K, F, torch.device('cpu'), overlap=True)
    assert not gatherer.overlap                            # (no side stream on a CPU device: same call sequence, inline)
    thresh = 0.5

    def gather(rows):                                      # exactly bench.py's round-3 hook
        gatherer.consume(0, thresh)                        # the previous frame's block
        gatherer(rows)
        return gatherer.rows_free
    det.gather_fn = gather
    frames = [torch.full((len(ids), 3, 4, 4), 1.0) for _ in range(3)]
    nfr, _ = parallel.run_steps(det, lambda t: frames[t % 3], [None] * len(ids), steps=2, frames_per_step=2)
    gatherer.consume(0, thresh)                            # the last frame's
    # what every rank must have counted: detections above the threshold of ALL streams over the 4 frames
    want = sum(int((_FakeDetector.rows_of(s, t, K, F, 1.0)[:, 0] > thresh).sum()) for s in range(num_streams) for t in range(nfr))
    local = sum(int((_FakeDetector.rows_of(s, t, K, F, 1.0)[:, 0] > thresh).sum()) for s in ids for t in range(nfr))
    total = parallel.sum_over_ranks(local)
    parallel.barrier()
    q.put((rank, gatherer.consumed_steps, gatherer.consumed_detections, want, total))
    torch.distributed.destroy_process_group()


def test_gathered_block_is_consumed_every_step_and_counts_all_streams():
    world, num_streams, K, F = 2, 5, 6, 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_consume_worker, args=(r, world, port, num_streams, K, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, steps, counted, want, total in got:
        assert steps == 4 and counted == want == total


def _worker_one_rank_group(port, q):
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from centertrack_amd import parallel
    r, w, _ = parallel.init_from_env(backend='gloo')
    rows = _rows(3, 5, 11)
    g = parallel.DetectionGatherer(3, 1, 0, 5, 11, 'cpu')
    out = g(rows).clone()
    h = parallel.check_same_plan('conv:201:1;dcn_knobs=(128, 4, 4, 1, 0, 0)')
    try:        # a gatherer sized for another world than the group's is refused with a clear error, not a shape error
        parallel.DetectionGatherer(4, 2, 0, 5, 11, 'cpu')
        refused = False
    except ValueError as e:
        refused = 'process group of 1 ranks' in str(e)
    q.put((r, w, parallel.group_active(), g.collective, bool(torch.equal(out, rows)), g.verify(rows), len(h), refused))
    torch.distributed.destroy_process_group()


def test_group_of_one_rank_exchanges_through_the_collective():
    """a torchrun job of ONE rank gets its process group, and with a group the exchange step is the real collective
    (send -> all_gather_into_tensor -> reorder), not the copy a plain single process does: how tests/test_hip_rccl.py
    makes RCCL execute on a one-GPU box"""
    from centertrack_amd import parallel
    plain = parallel.DetectionGatherer(3, 1, 0, 5, 11, 'cpu')
    assert not plain.collective                                # no group in this process: rows are only copied
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_worker_one_rank_group, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert got == (0, 1, True, True, True, 1, 16, True)


def test_rank_variable_without_a_rendezvous_is_a_plain_single_process(monkeypatch):
    """RANK exported by something that is not a launcher (WORLD_SIZE 1 or absent, no MASTER_ADDR / MASTER_PORT): no
    process group is created -- init_process_group would fail on the missing rendezvous (advisor, round 4)"""
    from centertrack_amd import parallel
    for k in ('MASTER_ADDR', 'MASTER_PORT', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('RANK', '0')
    assert parallel.init_from_env(backend='gloo') == (0, 1, 0)
    monkeypatch.setenv('WORLD_SIZE', '1')
    assert parallel.init_from_env(backend='gloo') == (0, 1, 0)
    assert not parallel.group_active()
