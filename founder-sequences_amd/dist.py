"""Launcher-side logic for N > 1: one process per GPU, independent alignments per rank.

The segmentation of one alignment is a chain of n/L dependent DP rounds and does not shard; what
shards with no data-path collective is the set of alignments.  This module
holds the pieces bench.py shares with the CPU (gloo) test: rank -> work assignment, the barrier +
max-over-ranks timing bracket, and the aggregate throughput."""
import os
import time


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def alignments_for_rank(n_alignments, rank, world):
    """Contiguous split of alignment ids [0, n_alignments) over ranks (sizes differ by at most 1)."""
    base, rem = divmod(n_alignments, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def seed_for_alignment(base_seed, alignment_id):
    return (base_seed + 0x1000 * alignment_id) & 0xFFFFFFFFFFFFFFFF


def timed_steps(step_fn, steps, warmup, dist=None, device_sync=None, tensor_factory=None):
    """Runs warmup + steps of step_fn between barriers; returns the max-over-ranks wall time of the
    timed region in seconds.  dist: torch.distributed (initialised) or None for a single process."""
    def barrier():
        if dist is not None:
            dist.barrier()
        if device_sync is not None:
            device_sync()

    for _ in range(warmup):
        step_fn()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = tensor_factory([dt])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def aggregate_cells_per_second(cells_per_rank_step, steps, world, seconds):
    return world * cells_per_rank_step * steps / seconds
