"""Segmentation throughput at other row counts than the BASELINE configs (one kernel configuration each):
    python tools/shape_sweep.py > profiles/r03_shape_sweep.txt"""
import importlib, sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("founder-sequences_amd")
for (m, n, L) in [(1200, 500000, 60), (2200, 500000, 100), (3400, 300000, 100), (4800, 300000, 100), (5008, 300000, 100), (7000, 200000, 100), (9000, 150000, 100)]:
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(0x5EED0000 + m, 24, 2000, 1e-4, 0)
    best = None
    for i in range(4):
        ctx.run()
        t = ctx.timings()
        if best is None or t["ms_total"] < best["ms_total"]:
            best = dict(t)
    print("m=%d n=%d: total %.3f ms  A %.3f B %.3f C %.3f D %.3f p2 %.3f  -> %.1f G cells/s" % (m, n, best["ms_total"], best["ms_phase_a"], best["ms_phase_b"], best["ms_phase_c"], best["ms_dp"], best["ms_pass2"], m * n / best["ms_total"] / 1e6), flush=True)
    ctx.close()
