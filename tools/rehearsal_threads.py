#!/usr/bin/env python3
"""Rehearsal of a sharded run with the ranks as THREADS of one process on ONE GPU (the development pool has one card
and lets at most six processes use it, so eight rank processes cannot be rehearsed): the real library, one context per
rank with 1 / W of the card's memory as its budget, the exchanges as a barrier + a device reduction over the ranks'
exchange tensors (founder-sequences_amd/dist.py, ThreadWorld).  What it shows is NOT scaling -- the ranks share one
chip --, it is (a) that W ranks run the whole workload to the same result as one GPU does, bit for bit, (b) how many
exchanges and bytes a step makes, (c) which phases are replicated: a rank's phase times against the single-GPU ones.

    python tools/rehearsal_threads.py C4 8 > profiles/r04_rehearsal_C4_8ranks_one_gpu.json
    python tools/rehearsal_threads.py C4 8 2 --serialize     # one rank at a time on the card: per-rank phase times as on a
                                                              # GPU of its own -> the predicted T(N) of DESIGN.md section 6
"""
import hashlib
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def digest(ctx):
    """sha256 over the traceback and the merged segments of a context"""
    import numpy as np
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(ctx.traceback()).tobytes())
    h.update(np.ascontiguousarray(ctx.reduced_traceback()).tobytes())
    return h.hexdigest()


def main():
    import numpy as np
    import torch
    import bench
    pkg = importlib.import_module("founder-sequences_amd")
    fdist = importlib.import_module("founder-sequences_amd.dist")
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 2
    serialize = "--serialize" in sys.argv
    w = bench.WORKLOADS[name]
    m, n, L = w["m"], w["n"], w["L"]
    dev = torch.device("cuda", 0)
    total = torch.cuda.get_device_properties(0).total_memory
    keys = ("ms_phase_a", "ms_phase_b", "ms_phase_c", "ms_dp", "ms_pass2", "ms_host", "ms_total")

    # ---- one GPU, not sharded: the reference result and phase times
    ctx = pkg.SegmentationContext(m, n, L, device=0)
    ctx.generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
    ctx.run()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run()
    torch.cuda.synchronize()
    single_ms = (time.perf_counter() - t0) / steps * 1e3
    single_t = ctx.timings()
    single = {"ms_per_step": single_ms, "phases_ms": {k: round(single_t[k], 3) for k in keys}, "n_blocks": single_t["n_blocks"], "block_len": single_t["block_len"],
              "digest": digest(ctx), "segments": int(ctx.result.segment_count), "max_segment_size": int(ctx.result.max_segment_size)}
    # boundary states of a few merged segments, to compare with their owners' (first, last, and around every rank border)
    red = ctx.reduced_traceback()
    probe = sorted(set([0, len(red) - 1] + [int(np.searchsorted(red["rb"], n * g // world)) for g in range(1, world)] +
                       [max(0, int(np.searchsorted(red["rb"], n * g // world)) - 1) for g in range(1, world)]))
    probe = [i for i in probe if 0 <= i < len(red)]
    ref_states = {i: tuple(hashlib.sha256(x.tobytes()).hexdigest() for x in ctx.boundary_state(i)) for i in probe}
    ctx.close()
    del ctx
    torch.cuda.empty_cache()

    # ---- W ranks as threads
    tw = fdist.ThreadWorld(world, take_turns=serialize)
    ctxs = [pkg.SegmentationContext(m, n, L, device=0) for _ in range(world)]
    errs = [None] * world
    per_rank = [None] * world
    wall = [0.0] * world

    def work(r):
        try:
            tw.attach(ctxs[r], r, dev)
            ctxs[r].set_memory_budget(int(total * 0.92 / world))
            with tw.turn(r):
                ctxs[r].generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
                ctxs[r].run()                              # warm-up (allocations, list capacity)
            tw.barrier.wait()
            tw.busy[r] = 0.0
            acc = {k: 0.0 for k in keys}
            t0 = time.perf_counter()
            for _ in range(steps):
                with tw.turn(r):
                    ctxs[r].run()
                t = ctxs[r].timings()
                for k in keys:
                    acc[k] += t[k]
            torch.cuda.synchronize()
            tw.barrier.wait()
            wall[r] = (time.perf_counter() - t0) / steps * 1e3
            per_rank[r] = {k: round(v / steps, 3) for k, v in acc.items()}
        except BaseException as e:                         # a failing rank must not leave the others in a barrier for ever
            errs[r] = e
            tw.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    digests = [digest(c) for c in ctxs]
    states_ok = True
    for i in probe:
        owner = ctxs[0].shard_owner(int(red["rb"][i]))
        a, d = ctxs[owner].boundary_state(i)
        if (hashlib.sha256(a.tobytes()).hexdigest(), hashlib.sha256(d.tobytes()).hexdigest()) != ref_states[i]:
            states_ok = False
    tr = ctxs[0]._transport
    runs = steps + 1
    t0_ = ctxs[0].timings()
    out = {
        "what": "rehearsal: %d ranks as threads of one process on ONE MI355X (barrier + device reduction for the all-reduce); NOT a scaling measurement%s"
                % (world, "; ranks take turns on the card (one at a time outside the exchanges): the per-rank phase times are those of a GPU of its own" if serialize else ""),
        "serialized": serialize,
        "workload": name, "m": m, "n": n, "L": L, "n_gpus": 1, "ranks": world, "steps": steps,
        "single_gpu": single,
        "sharded": {
            "ms_per_step_all_ranks_on_one_card": max(wall),
            "n_blocks": t0_["n_blocks"], "block_len": t0_["block_len"], "dp_chunks": t0_["dp_chunks"], "dp_sweeps": t0_["dp_sweeps"],
            "exchanges_per_step": tr.calls // runs, "MB_per_step": round(tr.words_moved * 4 / runs / 1e6, 3),
            "dp_keeps": "windows" if not ctxs[0].debug_dp_owned()[3] else "whole arrays",
            "phases_ms_per_rank": per_rank,
            # --serialize: what a rank spent holding the card per step (its own kernels and host code between the exchanges) --
            # its step on a GPU of its own, exchanges excluded; phases that contain exchanges (B, DP, host) include the
            # other ranks' turns in their event times above and are to be read from this figure minus A, C and pass 2
            "busy_ms_per_rank": [round(b / steps * 1e3, 3) for b in tw.busy] if serialize else None,
            "columns_per_rank": [c.shard_columns() for c in ctxs],
        },
        "bit_identical_to_single_gpu": all(d == single["digest"] for d in digests),
        "boundary_states_probed": len(probe), "boundary_states_identical": states_ok,
        "csrc_sha": bench.csrc_sha(),
    }
    print(json.dumps(out))
    for c in ctxs:
        c.close()
    return 0 if out["bit_identical_to_single_gpu"] and states_ok else 1


if __name__ == "__main__":
    sys.exit(main())
