#!/usr/bin/env python3
"""The clock the chip holds inside phase C's kernel (diagnostic build: FSEQ_HIPCC_FLAGS=-DFSEQ_CLOCK_STAMPS), after
>= 2 s of back-to-back steps: d(s_memtime) / d(s_memrealtime) x 100 MHz around every workgroup, the median
(MI355X_MICROARCH.md "DVFS give-back" item 6).  bench.py prices the VALU issue share of the dominant kernel with the
EFFECTIVE clock of the profiled launch (GRBM_GUI_ACTIVE / 8 / duration); this is the check of that figure.

    FSEQ_HIPCC_FLAGS=-DFSEQ_CLOCK_STAMPS python founder-sequences_amd/build.py --force && python tools/clock_probe.py C3 C5 C4cols50k
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
    import bench
    pkg = importlib.import_module("founder-sequences_amd")
    for name in (sys.argv[1:] or ["C3"]):
        w = bench.WORKLOADS[name]
        ctx = pkg.SegmentationContext(w["m"], w["n"], w["L"], device=0)
        ctx.generate_synthetic(w["seed"], w["K"], w["B"], w["mu"], w["kind"])
        ctx.run()
        t0 = time.perf_counter()
        steps = 0
        while time.perf_counter() - t0 < 2.5 or steps < 3:
            ctx.run()
            steps += 1
        ghz, n = ctx.debug_clock()
        t = ctx.timings()
        print(json.dumps({"workload": name, "in_kernel_clock_ghz": round(ghz, 4), "workgroups_stamped": n, "steps": steps,
                          "ms_phase_c": round(t["ms_phase_c"], 3), "note": "median over phase C's workgroups of d(s_memtime)/d(s_memrealtime) x 100 MHz, diagnostic build"}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
