"""world_size-2 gloo test (CPU) of the N > 1 launcher logic bench.py uses: rank -> alignment
assignment, barrier + max-over-ranks timing, aggregate throughput.  The step function here is a
stand-in sleep; the GPU work itself is covered by the -m gpu tests."""
import importlib
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    d = importlib.import_module("founder-sequences_amd.dist")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        assert d.env_rank() == (rank, rank, world)
        mine = d.alignments_for_rank(5, rank, world)
        calls = []

        def step():
            calls.append(1)
            time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one

        dt = d.timed_steps(step, steps=3, warmup=1, dist=dist, device_sync=None,
                           tensor_factory=lambda v: torch.tensor(v, dtype=torch.float64))
        q.put((rank, mine, len(calls), dt, d.seed_for_alignment(0x5EED0002, mine[0])))
    finally:
        dist.destroy_process_group()


def test_two_rank_timing_and_assignment():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, a0, c0, t0, s0), (r1, a1, c1, t1, s1) = out
    assert a0 == [0, 1, 2] and a1 == [3, 4]                 # contiguous, sizes differ by <= 1
    assert c0 == c1 == 4                                    # warmup + steps
    assert abs(t0 - t1) < 1e-9                              # both ranks report the max over ranks
    assert t0 >= 3 * 0.04 * 0.9                             # ... which is the slow rank's time
    assert s0 != s1
    d = importlib.import_module("founder-sequences_amd.dist")
    assert d.aggregate_cells_per_second(10, 3, 2, 2.0) == 30.0


def test_assignment_covers_everything_once():
    d = importlib.import_module("founder-sequences_amd.dist")
    for n in (0, 1, 7, 22, 64):
        for world in (1, 2, 3, 8):
            got = sum((d.alignments_for_rank(n, r, world) for r in range(world)), [])
            assert got == list(range(n))
