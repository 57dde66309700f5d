"""The column-block decomposition (tests/proto_blocks.py, the model of the HIP kernels) must
reproduce the oracle bit for bit: DP array, traceback, merged segments, boundary states."""
import numpy as np
import pytest

import fso
import proto_blocks as pb


def _codes(msa):
    u = np.unique(msa)
    return np.searchsorted(u, msa).astype(np.int64)


def _compare(msa, L, B, X, digit_bits=4):
    ref = fso.segment_long(msa, L, keep_dp=True, debug=True)
    got = pb.segment(_codes(msa), L, B, X, digit_bits)
    m, n = msa.shape
    if got["overflow"]:
        return False
    dp = ref["dp"]
    written = np.ones(len(dp), dtype=bool)
    written[n - 2 * L + 1:n - L] = False
    assert np.array_equal(got["M"][written], dp["segment_max_size"][written].astype(np.int64))
    assert np.array_equal(got["LB"][written], dp["lb"][written].astype(np.int64))
    assert np.array_equal(got["SZ"][written], dp["segment_size"][written].astype(np.int64))
    assert got["max_segment_size"] == ref["max_segment_size"]
    tb = ref["traceback"]
    assert [(int(x["lb"]), int(x["rb"]), int(x["segment_max_size"]), int(x["segment_size"])) for x in tb] == got["traceback"]
    if ref["status"] == 0:
        red = ref["reduced"]
        assert [(int(x["lb"]), int(x["rb"]), int(x["segment_size"])) for x in red] == got["reduced"]
        for i, (a, d) in enumerate(got["snaps"]):
            assert np.array_equal(a, ref["a"][i].astype(np.int64))
            assert np.array_equal(d, ref["d"][i].astype(np.int64))
    return True


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,B,X", [
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 16, 255),      # C1
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 1000, 255),    # one block
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 1, 255),       # B = 1
    (24, 400, 7, 4, 60, 1e-2, 11, 37, 255),
    (40, 300, 20, 5, 50, 5e-3, 12, 64, 255),
    (16, 64, 32, 2, 30, 1e-2, 13, 10, 255),                # n == 2L
    (12, 200, 1, 3, 20, 2e-2, 15, 7, 255),                 # L = 1
    (70, 500, 9, 6, 45, 8e-3, 21, 50, 255),                # > 16 distinct keys per block: 2 digit passes
])
def test_blocks_match_oracle(m, n, L, K, Brec, mu, seed, B, X):
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu), m, n)
    assert _compare(msa, L, B, X)


def test_blocks_sigma16_and_2bit_digits():
    msa = fso.synth_msa(fso.synth_spec(5, 4, 50, 1e-2, kind=1), 30, 300)
    assert _compare(msa, 8, 32, 255)
    assert _compare(msa, 8, 32, 255, digit_bits=2)


def test_truncated_lists_are_exact_or_flag_overflow():
    msa = fso.synth_msa(fso.synth_spec(77, 4, 60, 1e-2), 60, 400)
    ref = fso.segment_long(msa, 10, debug=True)
    mx = ref["max_segment_size"]
    saw_ok = saw_overflow = False
    for X in (1, 2, 4, mx - 1, mx, mx + 1, 2 * mx, 60):
        ok = _compare(msa, 10, 40, X)
        saw_ok |= ok
        saw_overflow |= not ok
    assert saw_ok and saw_overflow
    assert _compare(msa, 10, 40, 60)        # X >= m: lists are complete, never overflows


def test_device_rmq_equals_restated_rmq():
    rng = np.random.default_rng(3)
    for trial in range(3):
        n = 64 * 9 + 17
        vals = rng.integers(0, 6 if trial else 1000, size=n).astype(np.uint32)
        r = fso.Rmq(vals)
        dr = pb.DeviceRmq(n)
        for i in range(n):
            dr.append(i, int(vals[i]))
            r.update(i)
        for _ in range(4000):
            b = int(rng.integers(0, n - 1))
            e = int(rng.integers(b + 1, n + 1))
            assert dr.query(b, e) == r.query(b, e), (b, e)
