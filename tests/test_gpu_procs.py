"""N > 1 as the driver launches it: one PROCESS per rank, the real library, torch.distributed for the exchanges.
The ranks are started as children of the test process (never re-executed from a process that has touched the GPU).
On the one-GPU development box they share cuda:0 and exchange over gloo through a host copy (ShardTransport
via_host); with two or more GPUs the same worker runs over RCCL, one rank per card."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(world, case, backend="gloo", expect="ok", extra_env=None, timeout=240):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), case, backend, expect], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs, codes = [], []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
            codes.append(p.returncode)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                   # (exactly the children started here)
    return codes, outs


@pytest.mark.parametrize("world,case", [(2, "lds"), (2, "streamed"), (3, "sigma16")])
def test_ranks_as_processes_sharing_one_gpu(world, case):
    """Every rank's DP array, traceback and merged segments, and the boundary states it owns, equal the oracle's."""
    codes, outs = run_ranks(world, case)
    assert codes == [0] * world, "\n".join(outs)


def test_a_failing_rank_takes_the_others_with_it():
    """One rank fails on its own (injected after phase A, where an out-of-memory would strike): it reports its error,
    the other rank returns FSEQ_E_PEER from its next exchange instead of waiting in a collective."""
    codes, outs = run_ranks(2, "lds", expect="failure", extra_env={"FSEQ_INJECT_FAILURE_RANK": "1"}, timeout=120)
    assert codes == [0, 0], "\n".join(outs)


def test_two_ranks_over_rccl():
    """The driver's transport: backend "nccl" (= RCCL), in-place all-reduce on the exchange tensor, one rank per GPU."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the development box has one)")
    codes, outs = run_ranks(2, "streamed", backend="nccl")
    assert codes == [0, 0], "\n".join(outs)


def test_one_rank_over_rccl():
    """The driver's backend with the one GPU there is: the exchange primitive and bench.py's timing collectives against the
    real RCCL (tests/rccl_transport_worker.py) -- the dtypes, ops and views the ranks of a sharded run use."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_transport_worker.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=240)
    assert p.returncode == 0 and "rccl transport ok" in p.stdout, p.stdout[-2000:]


def test_bench_in_the_drivers_launch_shape_with_four_ranks():
    """The driver's N > 1 command -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N -- with
    N = 4 (the pool lets six processes open a card: four ranks, their launcher and this test) as a rehearsal on the one GPU (FSEQ_BENCH_REHEARSAL=1: the ranks share cuda:0, the
    timing collectives and the exchanges go over gloo): one JSON line from rank 0, "scaling": "strong", four ranks, and the same
    segmentation as the single-GPU run of the workload (BASELINE C5's alignment: 8,105 merged segments, max size 159) -- through
    the reduced phase C and pass 2 of every rank's own blocks."""
    import json
    env = dict(os.environ, FSEQ_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "4", "--workload", "C5", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["ranks"] == 4 and d["n_gpus"] == 1 and d["scaling"] == "strong" and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["segments"] == 8105 and d["config"]["max_segment_size"] == 159
    assert d["config"]["reduced_blocks"] > 0 and "all-reduces" in d["config"]["parallelism"]
