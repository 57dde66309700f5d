"""Launcher-side logic for N > 1: one process per GPU.

ONE alignment over the ranks: rank r owns a contiguous range of column blocks; the
library (fseq_set_shard, include/fseq.h) asks for a handful of small all-reduces per step through one callback.
ShardTransport is that callback over torch.distributed: RCCL (backend "nccl") on the exchange tensor when every
rank has its own GPU, or gloo through a pinned host copy when ranks share a card (rehearsals on one GPU, tests).
Also here: the pieces bench.py shares with the CPU (gloo) tests -- rank -> work assignment, the barrier +
max-over-ranks timing bracket, the aggregate throughput."""
import os
import time


class ShardTransport:
    """The exchange buffer of a sharded context and its all-reduce (fseq_allreduce_fn)."""

    def __init__(self, words, device, dist=None, via_host=False):
        import torch
        self.torch = torch
        self.dist = dist
        self.via_host = via_host
        self.buf = torch.zeros(int(words), dtype=torch.int32, device=device)      # uint32 words; sums of one non-zero
        self.words = int(words)                                                    # contribution and maxima of small values
        self.calls = 0                                                             # are the same in int32
        self.words_moved = 0
        self._host = None

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def allreduce(self, offset, count, op):
        torch, dist = self.torch, self.dist
        self.calls += 1
        self.words_moved += count
        if dist is None or count == 0:
            return 0
        view = self.buf[offset:offset + count]
        rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
        # (uint32 words travel as int32: sums of one non-zero contribution are exact whatever the sign; maxima are only
        # taken of small values -- list capacities, flags, presence and status words -- which must stay below 2^31)
        if self.via_host:
            h = view.cpu()
            if op == 1:
                assert int(h.min()) >= 0, "a MAX exchange carries a word >= 2^31"
            dist.all_reduce(h, op=rop)
            view.copy_(h)
        else:
            if op == 1 and count <= 4096:
                assert int(view.min()) >= 0, "a MAX exchange carries a word >= 2^31"
            dist.all_reduce(view, op=rop)
        torch.cuda.synchronize(self.buf.device)
        return 0


class ThreadWorld:
    """W ranks as W threads of one process on one GPU (tests): the all-reduce is a barrier, a reduction by rank 0
    over the ranks' exchange tensors, and a second barrier.  ctypes releases the GIL around the library call and
    takes it again for the callback, so the ranks really run side by side."""

    def __init__(self, world, take_turns=False):
        """take_turns: the ranks hold a common lock whenever they are NOT in an exchange, so one rank at a time has work on
        the card and its phase times are those of a GPU of its own (tools/rehearsal_threads.py --serialize: the per-rank
        times a prediction of T(N) is made from; the caller takes / drops the lock around its own calls with turn())."""
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.bufs = [None] * world
        self.turn_lock = threading.Lock() if take_turns else None
        self.busy = [0.0] * world             # seconds a rank held the card (its own host + device time between exchanges)
        self._since = [0.0] * world

    def _take(self, rank):
        if self.turn_lock is not None:
            self.turn_lock.acquire()
            self._since[rank] = time.perf_counter()

    def _drop(self, rank):
        if self.turn_lock is not None:
            self.busy[rank] += time.perf_counter() - self._since[rank]
            self.turn_lock.release()

    def turn(self, rank):
        """context manager: this rank's turn on the card (a no-op unless take_turns)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._take(rank)
            try:
                yield
            finally:
                self._drop(rank)
        return cm()

    def transport(self, words, rank, device):
        """(ShardTransport, allreduce(offset, count, op)) of one rank: a buffer of `words` words on `device`."""
        tr = ShardTransport(words, device, None)
        self.bufs[rank] = tr.buf
        torch = tr.torch

        def allreduce(offset, count, op, self=self, tr=tr, rank=rank):
            tr.calls += 1
            tr.words_moved += count
            self._drop(rank)                              # (the card is never held while waiting for the others)
            try:
                self.barrier.wait()
                if rank == 0:
                    views = [b[offset:offset + count] for b in self.bufs]
                    acc = views[0].clone()
                    for v in views[1:]:
                        acc = acc + v if op == 0 else torch.maximum(acc, v)
                    for v in views:
                        v.copy_(acc)
                    torch.cuda.synchronize(tr.buf.device)
                self.barrier.wait()
            finally:
                self._take(rank)
            return 0

        return tr, allreduce

    def attach(self, ctx, rank, device):
        tr, allreduce = self.transport(ctx.shard_xbuf_words(self.world), rank, device)
        ctx.set_shard(rank, self.world, tr.ptr, tr.words, allreduce)
        ctx._transport = tr
        return tr


def shard_context(ctx, rank, world, dist, device, via_host=False):
    """Makes ctx one rank of a sharded run; returns the transport (keep it alive with the context)."""
    tr = ShardTransport(ctx.shard_xbuf_words(world), device, dist if world > 1 else None, via_host)
    ctx.set_shard(rank, world, tr.ptr, tr.words, tr.allreduce)
    ctx._transport = tr
    return tr


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
