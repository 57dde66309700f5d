"""One rank of a sharded run as its own PROCESS (tests/test_gpu_procs.py starts world_size of these as children):
the real library, torch.distributed for the exchanges -- gloo through a pinned host copy when the ranks share a card,
RCCL ("nccl") when every rank has its own -- against the oracle's serial walk.  Exit code 0 = this rank agrees."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

CASES = {
    # name: m, n, L, K, Brec, mu, seed, kind, block_len
    "lds": (300, 6000, 25, 8, 200, 2e-3, 51, 0, 50),
    "streamed": (12000, 1500, 20, 12, 120, 3e-4, 53, 0, 30),
    "sigma16": (900, 5000, 100, 10, 300, 1e-3, 52, 1, 64),
}


def main():
    case, backend, expect = sys.argv[1], sys.argv[2], sys.argv[3]
    import numpy as np
    import torch
    import torch.distributed as dist
    import fso
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend="gloo")
    pkg = importlib.import_module("founder-sequences_amd")
    fdist = importlib.import_module("founder-sequences_amd.dist")
    m, n, L, K, Brec, mu, seed, kind, B = CASES[case]
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    ctx = pkg.SegmentationContext(m, n, L, block_len=B, device=local)
    fdist.shard_context(ctx, rank, world, dist, torch.device("cuda", local), via_host=(backend != "nccl"))
    ctx.set_sequences(msa)
    if expect == "failure":
        # FSEQ_INJECT_FAILURE_RANK (set by the test) makes one rank fail after phase A: that rank reports its own error,
        # every other rank FSEQ_E_PEER -- and nobody waits in a collective
        bad = int(os.environ["FSEQ_INJECT_FAILURE_RANK"])
        try:
            ctx.run()
        except pkg.FseqError as e:
            want = 4 if rank == bad else pkg.FSEQ_E_PEER
            print("rank %d: error %d (%s)" % (rank, e.code, e), flush=True)
            return 0 if e.code == want else 1
        print("rank %d: the run did not fail" % rank, flush=True)
        return 1
    ctx.run()
    ref = fso.segment_long(msa, L, keep_dp=True, threads=2)
    lb, mx, sz = ctx.debug_dp()
    from helpers import owned_dp_mask
    written = owned_dp_mask(ctx, n, L)                    # (a rank that keeps windows answers for the entries it computed)
    ok = ctx.result.max_segment_size == ref["max_segment_size"]
    ok = ok and np.array_equal(mx[written], ref["dp"]["segment_max_size"][written]) and np.array_equal(lb[written], ref["dp"]["lb"][written].astype(np.uint32))
    ok = ok and np.array_equal(sz[written], ref["dp"]["segment_size"][written])
    tb, red = ctx.traceback(), ctx.reduced_traceback()
    ok = ok and all(np.array_equal(tb[f], ref["traceback"][f]) for f in ("lb", "rb", "segment_max_size", "segment_size"))
    ok = ok and len(red) == len(ref["reduced"]) and all(np.array_equal(red[f], ref["reduced"][f]) for f in ("lb", "rb", "segment_size"))
    mine = 0
    for i in range(len(red)):
        if ctx.shard_owner(int(red["rb"][i])) == rank:
            a, d = ctx.boundary_state(i)
            ok = ok and np.array_equal(a, ref["a"][i]) and np.array_equal(d, ref["d"][i])
            mine += 1
    tr = ctx._transport
    print("rank %d/%d %s over %s: %s (%d segments, %d boundary states mine, %d exchanges, %d words)"
          % (rank, world, case, backend, "ok" if ok else "MISMATCH", len(red), mine, tr.calls, tr.words_moved), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
