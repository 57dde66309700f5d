#!/usr/bin/env python3
"""End-to-end wall split of the drop-in on BASELINE configs: what a founder_sequences run spends OUTSIDE the
segmentation the bench line measures -- the upload of the raw rows (H2D + alphabet scan + encode / transpose on the
device), the boundary states' way back to the host, the joiner (class tables, co-occurrence edges, edge drawing for
greedy; matching for bipartite), the founders file.  north_star: joining "stays on host until it shows in the profile";
this is that profile.

    python tools/e2e_profile.py C2 greedy      > profiles/r03_e2e_C2.json
    python tools/e2e_profile.py C3 bipartite   > profiles/r03_e2e_C3.json
"""
import hashlib
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import numpy as np
    import bench
    pkg = importlib.import_module("founder-sequences_amd")
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    how = sys.argv[2] if len(sys.argv) > 2 else "greedy"
    w = bench.WORKLOADS[name]
    m, n, L = w["m"], w["n"], w["L"]
    # the input as a caller has it: raw row-major bytes on the host (generated on the device, fetched once)
    gen = pkg.SegmentationContext(m, n, L)
    gen.generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
    msa = np.ascontiguousarray(gen.get_sequences())
    gen.close()
    out = {"workload": "%s: m=%d x n=%d, L=%d, joining=%s" % (name, m, n, L, how), "runs": []}
    for rep in range(3):                                      # first = cold (allocations, list capacity estimate)
        ctx = pkg.SegmentationContext(m, n, L)
        t0 = time.perf_counter()
        ctx.set_sequences(msa)
        t1 = time.perf_counter()
        ctx.run()
        t2 = time.perf_counter()
        if rep:                                               # steady state of the segmentation itself
            ctx.run()
            t2b = time.perf_counter()
        perm = {"greedy": ctx.join_greedy, "bipartite": ctx.join_bipartite, "random": lambda: ctx.join_random(1)}[how]()
        t3 = time.perf_counter()
        jp = ctx.join_profile()
        with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False) as f:
            path = f.name
        ctx.write_founders_device(perm, path)                   # [r5] the lines put together on the device, where the alignment is
        t4 = time.perf_counter()
        sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
        size = os.path.getsize(path)
        os.unlink(path)
        th0 = time.perf_counter()
        ctx.write_founders(msa, perm, path)                      # (the host writer, from the raw rows: same bytes)
        ms_host_writer = (time.perf_counter() - th0) * 1e3
        assert hashlib.sha256(open(path, "rb").read()).hexdigest() == sha
        os.unlink(path)
        t = ctx.timings()
        run = {
            "ms_upload_encode": (t1 - t0) * 1e3,
            "ms_segmentation_first_run": (t2 - t1) * 1e3,
            "ms_segmentation": ((t2b - t2) * 1e3) if rep else None,
            "ms_join": (t3 - (t2b if rep else t2)) * 1e3,
            "join_profile": jp,
            "ms_write_founders": (t4 - t3) * 1e3, "ms_write_founders_host_writer": ms_host_writer,
            "founders_bytes": size, "founders_sha256": sha,
            "segments": int(ctx.result.segment_count), "max_segment_size": int(ctx.result.max_segment_size),
            "phases_ms": {k: t[k] for k in ("ms_phase_a", "ms_phase_b", "ms_phase_c", "ms_dp", "ms_pass2", "ms_host", "ms_total")},
        }
        out["runs"].append(run)
        ctx.close()
    last = out["runs"][-1]
    seg = last["ms_segmentation"]
    out["summary"] = {
        "ms_upload_encode": last["ms_upload_encode"], "ms_segmentation": seg, "ms_boundary_states_d2h": last["join_profile"]["ms_d2h"],
        "ms_join_host": last["join_profile"]["ms_total"] - last["join_profile"]["ms_d2h"], "ms_write_founders": last["ms_write_founders"],
        "join_over_segmentation": last["ms_join"] / seg,
        "raw_input_bytes": int(m) * int(n), "boundary_state_bytes": last["join_profile"]["bytes_d2h"],
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
