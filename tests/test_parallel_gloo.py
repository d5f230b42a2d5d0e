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
