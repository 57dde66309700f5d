#!/usr/bin/env python3
"""sigma = 16 alignments in which most columns carry few codes (real gapped alignments; the survey's generator draws all 16
codes in nearly every column): BASELINE C5's rows on 200,000 columns, a share of the columns cut down to 1 .. 4 codes by an
order-preserving image, timed with the one-pass columns of round 4 and with every column in two digit passes.

    python tools/gapped_probe.py > profiles/r04_gapped_sigma16.txt
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def gapped(msa, share, seed):
    codes = np.unique(msa)
    rng = np.random.default_rng(seed)
    for k in np.nonzero(rng.random(msa.shape[1]) < share)[0]:
        pick = np.sort(rng.choice(len(codes), size=int(rng.integers(1, 5)), replace=False))
        col = np.searchsorted(codes, msa[:, k])
        msa[:, k] = codes[pick[col % len(pick)]]
    return msa


def main():
    import fso
    import bench
    pkg = importlib.import_module("founder-sequences_amd")
    m, n, L = 10000, 200000, 100
    print("# tools/gapped_probe.py on kernel sources %s: m = %d, n = %d, L = %d, sigma = 16; ms per step (phases A B C D pass 2)" % (bench.csrc_sha(), m, n, L))
    base = fso.synth_msa(fso.synth_spec(0x5EED0005, 32, 5000, 1e-4, 1), m, n)
    for share in (0.0, 0.5, 0.9):
        msa = gapped(base.copy(), share, 7)
        for env in ("", "1"):
            if env:
                os.environ["FSEQ_NO_DENSE_COLUMNS"] = env
            else:
                os.environ.pop("FSEQ_NO_DENSE_COLUMNS", None)
            ctx = pkg.SegmentationContext(m, n, L)
            ctx.set_sequences(msa)
            ctx.run()
            t0 = time.perf_counter()
            for _ in range(5):
                ctx.run()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            t = ctx.timings()
            print("columns with <= 4 codes: %3.0f %% | %-22s | %8.3f ms | A %.2f B %.2f C %.2f D %.2f p2 %.2f | max segment size %d"
                  % (share * 100, "two passes everywhere" if env else "one pass where dense", ms, t["ms_phase_a"], t["ms_phase_b"], t["ms_phase_c"], t["ms_dp"],
                     t["ms_pass2"], ctx.result.max_segment_size), flush=True)
            ctx.close()
    os.environ.pop("FSEQ_NO_DENSE_COLUMNS", None)


if __name__ == "__main__":
    main()
