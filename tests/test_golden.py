"""Committed fixtures (tests/golden/vectors.json, written by tests/golden/make_golden.py).

They are oracle outputs, not reference outputs (the reference cannot be built here), except the one DP step
recorded from the reference's own code in SURVEY.md Appendix D.3.  CPU: the oracle must keep reproducing
them (regression pin) and the recorded reference answer.  GPU: the HIP path is checked against the
fixtures directly, with no oracle in the loop."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

import fso

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "vectors.json")) as _f:
    GOLDEN = json.load(_f)
CASES = GOLDEN["cases"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _msa(c):
    msa = fso.synth_msa(fso.synth_spec(c["seed"], c["K"], c["Brec"], c["mu"], c["kind"]), c["m"], c["n"])
    assert sha(msa) == c["msa_sha256"]                       # the generator is part of the fixture
    return msa


def test_reference_known_answer_dp_step():
    k = GOLDEN["reference_known_answers"][0]
    dp = np.zeros(len(k["dp_segment_max_size"]) + 1, dtype=fso.DP_DTYPE)
    dp["segment_max_size"] = k["dp_segment_max_size"] + [0xFFFFFFFF]
    dp["segment_size"] = 0xFFFFFFFF
    L_ = fso.lib(True)
    h = L_.fso_rmq_new(dp.ctypes.data + 16, dp.dtype.itemsize, len(dp), 64)
    got = fso.dp_step(k["values"], k["counts"], dp, h, k["m"], k["L"], k["lb"], k["text_pos"], (0, k["m"], k["m"], k["m"]))
    L_.fso_rmq_free(h)
    e = k["expected"]
    assert got == (e["lb"], e["rb"], e["segment_max_size"], e["segment_size"])


@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_fixtures(c):
    r = fso.segment_long(_msa(c), c["L"], keep_dp=True, threads=4)
    assert int(r["status"]) == c["status"] and int(r["max_segment_size"]) == c["max_segment_size"]
    for f in ("lb", "rb", "segment_max_size", "segment_size"):
        assert [int(x) for x in r["traceback"][f]] == c["traceback"][f]
    for f in ("segment_max_size", "lb", "segment_size"):
        assert sha(r["dp"][f].astype(np.uint32)) == c["dp_sha256"][f]
    if c["status"] == 0:
        for f in ("lb", "rb", "segment_size"):
            assert [int(x) for x in r["reduced"][f]] == c["reduced"][f]
        assert sha(np.asarray(r["a"], dtype=np.uint32)) == c["boundary_a_sha256"]
        assert sha(np.asarray(r["d"], dtype=np.uint32)) == c["boundary_d_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=[c["name"] for c in CASES])
def test_hip_path_matches_fixtures(c):
    pkg = importlib.import_module("founder-sequences_amd")
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["Brec"], c["mu"], c["kind"])      # the device generator
    assert sha(np.ascontiguousarray(ctx.get_sequences())) == c["msa_sha256"]
    try:
        ctx.run()
    except pkg.NoReduction:
        assert c["status"] != 0
    assert ctx.result.max_segment_size == c["max_segment_size"]
    tb = ctx.traceback()
    for f in ("lb", "rb", "segment_max_size", "segment_size"):
        assert [int(x) for x in tb[f]] == c["traceback"][f]
    # (the DP-array hashes of the fixture include the entries no path writes, lp.cc parts 3 / 4: CPU test only)
    if c["status"] == 0:
        red = ctx.reduced_traceback()
        for f in ("lb", "rb", "segment_size"):
            assert [int(x) for x in red[f]] == c["reduced"][f]
        A = np.stack([ctx.boundary_state(i)[0] for i in range(len(red))])
        D = np.stack([ctx.boundary_state(i)[1] for i in range(len(red))])
        assert sha(A.astype(np.uint32)) == c["boundary_a_sha256"]
        assert sha(D.astype(np.uint32)) == c["boundary_d_sha256"]
