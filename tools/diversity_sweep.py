#!/usr/bin/env python3
"""Is the path fast only on the survey's friendly generator?  BASELINE C3's and C5's shapes with the founder count K
and the mutation rate mu of the synthetic mosaic swept from the bench's values to "every row
its own founder": per point the step time and its phases, the list capacity X that worked and the retries it took,
the sweeps of the speculative DP, the blocks whose key-space tree had to slice, the bytes of the per-column lists, and
whether the reference would give up (FSEQ_E_NO_REDUCTION, generate_context.cc:192-200).

    python tools/diversity_sweep.py > profiles/r04_diversity_sweep.txt
    python tools/diversity_sweep.py C3 --quick      # fewer points
    python tools/diversity_sweep.py C4 --K=64,1024,4096 --mu=5e-5,1e-3      # chosen points (C4's shape with more founders)
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import bench
    pkg = importlib.import_module("founder-sequences_amd")
    which = [a for a in sys.argv[1:] if not a.startswith("-")] or ["C3", "C5"]
    quick = "--quick" in sys.argv
    print("# tools/diversity_sweep.py on kernel sources %s; one MI355X; step = pass 1 + DP + traceback + merge + pass 2 (bench.py's metric)" % bench.csrc_sha())
    print("# columns: workload K mu | ms/step (x the K = bench point) | A B C D pass2 host ms | X retries dp_sweeps phase_a_fallbacks, blocks the tree gave to the sweep / the trie gave to the tree | "
          "max_segment_size segments | list GB | status")
    for name in which:
        w = bench.WORKLOADS[name]
        m, n, L = w["m"], w["n"], w["L"]
        Ks = [w["K"], 256, 1024, m]
        mus = [1e-4, 1e-3, 1e-2]
        if quick:
            Ks, mus = [w["K"], 1024], [1e-4, 1e-2]
        for a in sys.argv[1:]:                                     # chosen points: --K=64,4096 --mu=5e-5
            if a.startswith("--K="):
                Ks = [int(x) for x in a[4:].split(",")]
            if a.startswith("--mu="):
                mus = [float(x) for x in a[5:].split(",")]
        base_ms = None
        for K in Ks:
            for mu in mus:
                ctx = pkg.SegmentationContext(m, n, L, device=0)
                ctx.generate_synthetic(w["seed"], K, w["B"], mu, w["kind"])
                status = "ok"
                t_first = time.perf_counter()
                try:
                    ctx.run()                                      # finds the list capacity (retries are part of a FIRST run only)
                except pkg.NoReduction:
                    status = "NO_REDUCTION"
                except pkg.FseqError as e:
                    status = "error %d: %s" % (e.code, e)
                first_ms = (time.perf_counter() - t_first) * 1e3
                t_first_run = ctx.timings()
                steps = 3
                acc = {}
                ms = float("nan")
                if not status.startswith("error"):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        try:
                            ctx.run()
                        except pkg.NoReduction:
                            pass
                        t = ctx.timings()
                        for k in ("ms_phase_a", "ms_phase_b", "ms_phase_c", "ms_dp", "ms_pass2", "ms_host"):
                            acc[k] = acc.get(k, 0.0) + t[k] / steps
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / steps * 1e3
                t = ctx.timings()
                if base_ms is None:
                    base_ms = ms
                stride = (t["list_cap_used"] + 3) & ~1
                res = ctx.result
                print("%s K=%-6d mu=%-6g | %9.3f ms (%5.2fx) first run %9.1f ms | A %.2f B %.2f C %.2f D %.2f p2 %.2f host %.2f | X %d retries(first run) %d dp_sweeps %d fallbacks %d given up tree %d trie %d | max %d segments %d | lists %.2f GB | %s"
                      % (name, K, mu, ms, ms / base_ms if base_ms else float("nan"), first_ms,
                         acc.get("ms_phase_a", 0), acc.get("ms_phase_b", 0), acc.get("ms_phase_c", 0), acc.get("ms_dp", 0), acc.get("ms_pass2", 0), acc.get("ms_host", 0),
                         t["list_cap_used"], t_first_run["retries"], t["dp_sweeps"], t["phase_a_fallbacks"], t["phase_a_given_up"], t["phase_a_trie_given_up"],
                         res.max_segment_size if res else -1, res.segment_count if res else -1, n * stride * 8 / 1e9, status), flush=True)
                ctx.close()
                del ctx
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
