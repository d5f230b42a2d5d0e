"""GPU: RCCL really executes on this code once.  No multi-GPU node is available to the builder, so
the N > 1 path is covered by the gloo tests on CPU (tests/test_parallel_gloo.py); what those cannot show is that
``init_process_group('nccl')`` and an ``all_gather_into_tensor`` enqueued on the gatherer's side stream work on an MI355X
under ROCm's RCCL.  A torchrun job of ONE rank does: ``parallel.init_from_env`` creates the group whenever torchrun's
variables are present, and with a group the exchange step of ``bench.py`` is the real collective (plan-hash all-gather,
row all-gather on the side stream into pinned host memory, checksum all-gather of ``verify``), also for one rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_bench_under_torchrun_one_rank_goes_through_rccl(device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
           '--frames-per-step', '4', '--no-cpu-baseline', '--no-roofline', '--no-resident', '--no-extra-configs',
           '--no-box-probes']
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert lines, p.stdout[-2000:]
    j = json.loads(lines[-1])
    assert j['n_gpus'] == 1 and j['process_group'] == 'nccl', j.get('process_group')
    assert j['rccl_ranks'] == 1
    g = j['gathered']
    # every timed frame's rows went through the side-stream all-gather and were consumed from pinned host memory
    assert g['steps_consumed'] == 2 * 4 and g['equal'] and g['detections_counted_from_blocks'] > 0, g
    assert j['value'] > 100, j['value']
