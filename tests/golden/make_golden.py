#!/usr/bin/env python3
"""Writes tests/golden/vectors.json: fixed inputs (generator parameters of SURVEY.md Appendix E, or explicit
small matrices) and the expected outputs of the segmentation path.

WHERE THE EXPECTED VALUES COME FROM.  The reference cannot be built here (libbio / sdsl-lite / Lemon are not
in the tree), so these are NOT reference outputs, with one exception:
  * "reference_known_answers" holds the one DP step SURVEY.md Appendix D.3 recorded from the reference's own
    calculate_segmentation_lp_dp_arg + rmq.hh (the survey's stub build, not repeated in this round);
  * everything under "cases" was produced by THIS script from oracle/fseq_oracle.c (the CPU restatement).  It
    pins the oracle against regressions and gives the GPU tests a target that needs no oracle run; it does not
    pin anything to the reference (parity stays "unpinned").

    python tests/golden/make_golden.py        # rewrites vectors.json
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import fso  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


CASES = [
    # name, m, n, L, K, Brec, mu, seed, kind
    ("C1", 8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 0),
    ("C2_prefix", 2500, 3000, 50, 16, 2000, 1e-4, 0x5EED0002, 0),
    ("C5_prefix", 10000, 1200, 100, 32, 5000, 1e-4, 0x5EED0005, 1),
    ("sigma16_small", 30, 300, 8, 4, 50, 1e-2, 5, 1),
    ("L1", 12, 200, 1, 3, 20, 2e-2, 15, 0),
    ("n_equals_2L", 16, 64, 32, 2, 30, 1e-2, 13, 0),
    ("one_row", 1, 50, 5, 1, 10, 0.0, 26, 0),
    ("wide", 700, 900, 9, 6, 45, 8e-3, 21, 0),
]


def main():
    out = {
        "generated_by": "tests/golden/make_golden.py from oracle/fseq_oracle.c (NOT reference outputs; see the script's header)",
        "reference_known_answers": [{
            "source": "SURVEY.md Appendix D.3: the reference's calculate_segmentation_lp_dp_arg (segmentation_lp_context.cc:393-481) + rmq.hh",
            "m": 6, "L": 2, "lb": 0, "text_pos": 5, "dp_segment_max_size": [3, 2, 4, 2],
            "values": [0, 3, 5, 6], "counts": [2, 1, 1, 2], "expected": {"lb": 3, "rb": 6, "segment_max_size": 3, "segment_size": 3},
        }],
        "cases": [],
    }
    for name, m, n, L, K, Brec, mu, seed, kind in CASES:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        r = fso.segment_long(msa, L, keep_dp=True, threads=4)
        c = {"name": name, "m": m, "n": n, "L": L, "K": K, "Brec": Brec, "mu": mu, "seed": seed, "kind": kind,
             "msa_sha256": sha(msa), "status": int(r["status"]), "max_segment_size": int(r["max_segment_size"]),
             "traceback": {f: [int(x) for x in r["traceback"][f]] for f in ("lb", "rb", "segment_max_size", "segment_size")},
             "dp_sha256": {"segment_max_size": sha(r["dp"]["segment_max_size"].astype(np.uint32)), "lb": sha(r["dp"]["lb"].astype(np.uint32)),
                           "segment_size": sha(r["dp"]["segment_size"].astype(np.uint32))}}
        if r["status"] == 0:
            c["reduced"] = {f: [int(x) for x in r["reduced"][f]] for f in ("lb", "rb", "segment_size")}
            c["boundary_a_sha256"] = sha(np.asarray(r["a"], dtype=np.uint32))
            c["boundary_d_sha256"] = sha(np.asarray(r["d"], dtype=np.uint32))
        out["cases"].append(c)
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %d cases" % len(out["cases"]))


if __name__ == "__main__":
    main()
