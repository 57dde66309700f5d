"""CPU tests of the restatement in oracle/ (no GPU).

The reference holds no tests or golden vectors for this path and cannot be compiled
here (F2), so these tests pin the restatement to (i) the one known-answer case recorded from the
reference's own code in SURVEY.md Appendix D.3, (ii) the documented quirks of rmq.hh (F4), and
(iii) brute-force invariants of the pBWT / segmentation definitions.
"""
import ctypes as C

import numpy as np
import pytest

import fso
from helpers import brute_pbwt, distinct_count, optimal_max_segment_size

U32MAX = 0xFFFFFFFF


# ------------------------------------------------------------------ rmq.hh restatement

def _filled_rmq(vals, bs=64):
    r = fso.Rmq(vals, bs)
    for i in range(len(vals)):
        r.update(i)
    return r


def test_rmq_short_ranges_are_naive_first_min():
    # beg_block >= end_block -> std::min_element, i.e. the FIRST minimum (rmq.hh:90-91,116)
    rng = np.random.default_rng(1)
    vals = rng.integers(0, 5, size=300).astype(np.uint32)
    r = _filled_rmq(vals)
    for _ in range(2000):
        b = int(rng.integers(0, 299))
        e = int(rng.integers(b + 1, min(300, b + 64) + 1))
        if b // 64 + 1 < e // 64:
            continue
        got = r.query(b, e)
        assert got == b + int(np.argmin(vals[b:e]))


def test_rmq_value_minimal_below_four_blocks():
    # Levels 0 and 1 of the sparse table are built correctly; only levels >= 2 skip blocks
    # (rmq.hh:76-79).  A query whose whole-block span is < 4 blocks uses levels 0/1 only.
    rng = np.random.default_rng(2)
    vals = rng.integers(0, 1000, size=64 * 12).astype(np.uint32)
    r = _filled_rmq(vals)
    n = len(vals)
    checked = 0
    for _ in range(5000):
        b = int(rng.integers(0, n - 1))
        e = int(rng.integers(b + 1, n + 1))
        if e // 64 - (b // 64 + 1) >= 4:
            continue
        got = r.query(b, e)
        assert b <= got < e
        assert vals[got] == vals[b:e].min()
        checked += 1
    assert checked > 500


def test_rmq_level2_bug_is_reproduced():
    # F4: level-2 entry j covers blocks {j, j+1, j+3} only.  Put the unique minimum in block j+2
    # of a 4-block span: the reference returns a non-minimal element; so must the restatement.
    vals = np.full(64 * 8, 100, dtype=np.uint32)
    vals[64 * 3 + 5] = 1          # block 3 = the skipped one for the span of blocks 1..4
    vals[64 * 1 + 7] = 50
    r = _filled_rmq(vals)
    got = r.query(64, 64 * 5)     # beg_block = 2?  beg=64 -> beg_block = 2, end_block = 5
    # whole blocks 2,3,4 -> pow2 = hi(3) = 1 -> level 1 only: correct here
    assert vals[got] == 1
    got = r.query(60, 64 * 5)     # beg_block = 1, end_block = 5 -> pow2 = 2 -> level 2 [1] = blocks {1,2,4}
    assert got == 64 * 1 + 7      # block 3's minimum (value 1) is missed
    assert vals[got] == 50


def test_rmq_tie_rules():
    # operator() prefers the table sample over left_smp on ties and smp1 over smp2 (rmq.hh:96-98),
    # so the returned index is not the leftmost minimum.
    vals = np.full(64 * 4, 7, dtype=np.uint32)
    r = _filled_rmq(vals)
    # beg=10 -> beg_block=1; end=200 -> end_block=3; table sample = first elem of block 1 = 64.
    assert r.query(10, 200) == 64
    # exact block end: early return (rmq.hh:100-101)
    assert r.query(10, 192) == 64
    # strictly smaller on the left wins
    vals2 = vals.copy()
    vals2[20] = 6
    r2 = _filled_rmq(vals2)
    assert r2.query(10, 200) == 20
    # strictly smaller on the right wins; equal on the right does not
    vals3 = vals.copy()
    vals3[195] = 6
    r3 = _filled_rmq(vals3)
    assert r3.query(10, 200) == 195
    assert r3.query(10, 195) == 64


def test_rmq_update_only_on_block_completion():
    vals = np.arange(200, 0, -1).astype(np.uint32)
    r = fso.Rmq(vals)
    for i in range(130):
        r.update(i)
    # blocks 0 and 1 complete; a query spanning them works, elements 128.. are scanned naively
    assert r.query(0, 130) == 129
    assert r.query(0, 128) == 127


# ------------------------------------------------------------------ DP step

def _dp_array(maxes):
    dp = np.zeros(len(maxes), dtype=fso.DP_DTYPE)
    dp["segment_max_size"] = maxes
    dp["segment_size"] = maxes
    return dp


def test_dp_step_known_answer_from_reference():
    # obtained from the reference's own calculate_segmentation_lp_dp_arg
    # (segmentation_lp_context.cc:393-481) + rmq.hh: m=6, L=2, text_pos=5, DP max = {3,2,4,2},
    # pairs {(0,2),(3,1),(5,1),(6,2)} -> lb=3 rb=6 max=3 size=3.
    dp = _dp_array([3, 2, 4, 2, U32MAX])
    L_ = fso.lib(True)
    h = L_.fso_rmq_new(dp.ctypes.data + 16, dp.dtype.itemsize, len(dp), 64)
    got = fso.dp_step([0, 3, 5, 6], [2, 1, 1, 2], dp, h, 6, 2, 0, 5, (0, 6, 6, 6))
    L_.fso_rmq_free(h)
    assert got == (3, 6, 3, 3)


def test_dp_step_first_strictly_smaller_wins():
    # Two candidate ranges with the same value: the one visited first (smaller divergence value)
    # must be kept (segmentation_lp_context.cc:420,472 use operator<).
    dp = _dp_array([2, 2, 2, 2, 2, 2, 2, 2, U32MAX, U32MAX])
    L_ = fso.lib(True)
    h = L_.fso_rmq_new(dp.ctypes.data + 16, dp.dtype.itemsize, len(dp), 64)
    # m=10, L=2, text_pos=9: values 3,5,7,10 ; counts 6,1,1,2 -> ranges [3,5): rhs 4, [5,7): rhs 3, [7,9): rhs 2
    got = fso.dp_step([3, 5, 7, 10], [6, 1, 1, 2], dp, h, 10, 2, 0, 9, (0, 10, 10, 10))
    L_.fso_rmq_free(h)
    # candidates: max(2,4)=4 @ idx 1 ; max(2,3)=3 @ idx 3 ; max(2,2)=2 @ idx 5 -> last is strictly smaller
    assert got == (7, 10, 2, 2)


def test_dp_step_skips_range_after_zero():
    # When the smallest value equals lb (0), the range (v0, v1) is never evaluated (:416-428).
    dp = _dp_array([1, 1, 1, 1, 1, 1, U32MAX, U32MAX, U32MAX])
    L_ = fso.lib(True)
    h = L_.fso_rmq_new(dp.ctypes.data + 16, dp.dtype.itemsize, len(dp), 64)
    # m=5, L=2, text_pos=7; values 0,4,8 counts 2,1,2: whole range -> 3; range (0,4) skipped; range (4,8) clipped to [4,7) -> rhs = 5-3 = 2
    got = fso.dp_step([0, 4, 8], [2, 1, 2], dp, h, 5, 2, 0, 7, (0, 8, 5, 5))
    L_.fso_rmq_free(h)
    assert got == (4, 8, 2, 2)
    dp = _dp_array([1, 1, 1, 1, 1, 1, U32MAX, U32MAX, U32MAX])
    h = L_.fso_rmq_new(dp.ctypes.data + 16, dp.dtype.itemsize, len(dp), 64)
    # only values 0 and 8: the sole range (0,8) is skipped -> whole-range candidate stays
    got = fso.dp_step([0, 8], [3, 2], dp, h, 5, 2, 0, 7, (0, 8, 5, 5))
    L_.fso_rmq_free(h)
    assert got == (0, 8, 2, 2)


# ------------------------------------------------------------------ pBWT restatement vs brute force

@pytest.mark.parametrize("sigma,m,n,seed", [(2, 13, 40, 0), (4, 32, 60, 1), (16, 64, 50, 2), (4, 1, 20, 3), (3, 50, 30, 4)])
def test_pbwt_matches_brute_force(sigma, m, n, seed):
    rng = np.random.default_rng(seed)
    # mosaic-ish: few founders with noise so that long matches exist
    founders = rng.integers(0, sigma, size=(3, n))
    pick = rng.integers(0, 3, size=m)
    msa = founders[pick].astype(np.uint8)
    noise = rng.random((m, n)) < 0.05
    msa[noise] = rng.integers(0, sigma, size=int(noise.sum()))
    msa = (msa + 65).astype(np.uint8)
    for order in ("C", "F"):
        x = np.array(msa, order=order)
        p = fso.Pbwt(x)
        for k in range(n + 1):
            a, d = brute_pbwt(msa, k)
            assert p.idx == k
            assert np.array_equal(p.a, a), (k, order)
            assert np.array_equal(p.d, d), (k, order)
            v, c = p.counts()
            uv, uc = np.unique(d, return_counts=True)
            assert np.array_equal(v, uv) and np.array_equal(c, uc)
            for lb in (0, k // 2, k):
                if lb <= k:
                    assert p.unique_substring_count_lhs(lb) == (distinct_count(msa, lb, k) if lb < k else int((d > lb).sum()))
            if k < n:
                p.step()


def test_pbwt_runs_lhs():
    rng = np.random.default_rng(7)
    msa = (rng.integers(0, 2, size=(20, 12)) + 65).astype(np.uint8)
    p = fso.Pbwt(msa)
    for _ in range(12):
        p.step()
    f, r = p.unique_substring_count_idxs_lhs(6)
    assert r.sum() == 20
    a = p.a
    pos = 0
    for fi, rl in zip(f, r):
        assert a[pos] == fi
        rows = a[pos:pos + rl]
        assert len({bytes(msa[x, 6:12]) for x in rows}) == 1
        pos += rl
    assert len(f) == distinct_count(msa, 6, 12)


def test_pbwt_set_state_roundtrip():
    spec = fso.synth_spec(5, 3, 40, 0.01)
    msa = fso.synth_msa(spec, 17, 100)
    p = fso.Pbwt(msa)
    for _ in range(30):
        p.step()
    a, d = p.a, p.d
    q = fso.Pbwt(msa)
    q.set_state(a, d, 30)
    for _ in range(25):
        p.step()
        q.step()
    assert np.array_equal(p.a, q.a) and np.array_equal(p.d, q.d)
    assert all(np.array_equal(x, y) for x, y in zip(p.counts(), q.counts()))


# ------------------------------------------------------------------ long path, end to end

def _check_segmentation(msa, L, res, check_optimum):
    m, n = msa.shape
    tb = res["traceback"]
    assert tb[0]["lb"] == 0 and tb[-1]["rb"] == n
    for i in range(len(tb)):
        assert tb[i]["rb"] - tb[i]["lb"] >= L
        if i:
            assert tb[i]["lb"] == tb[i - 1]["rb"]
        assert tb[i]["segment_size"] == distinct_count(msa, int(tb[i]["lb"]), int(tb[i]["rb"]))
    mx = max(int(x) for x in tb["segment_size"])
    assert res["max_segment_size"] == mx == tb[-1]["segment_max_size"]
    if check_optimum:
        assert mx == optimal_max_segment_size(msa, L)
    red = res["reduced"]
    assert red[0]["lb"] == 0 and red[-1]["rb"] == n
    for i in range(len(red)):
        if i:
            assert red[i]["lb"] == red[i - 1]["rb"]
        lb, rb = int(red[i]["lb"]), int(red[i]["rb"])
        assert red[i]["segment_size"] == distinct_count(msa, lb, rb) <= mx
        assert red[i]["segment_max_size"] == U32MAX
        a, d = brute_pbwt(msa, rb)
        assert np.array_equal(res["a"][i], a)
        assert np.array_equal(res["d"][i], d)
    # merged boundaries are a subset of the DP boundaries
    assert set(red["rb"].tolist()) <= set(tb["rb"].tolist())


@pytest.mark.parametrize("m,n,L,K,B,mu,seed", [
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001),     # BASELINE config C1
    (24, 400, 7, 4, 60, 1e-2, 11),
    (40, 300, 20, 5, 50, 5e-3, 12),
    (16, 64, 32, 2, 30, 1e-2, 13),               # n == 2L: no part 2, one part-3 column
    (16, 65, 32, 2, 30, 1e-2, 14),
    (12, 200, 1, 3, 20, 2e-2, 15),               # L = 1
])
def test_long_path_invariants(m, n, L, K, B, mu, seed):
    spec = fso.synth_spec(seed, K, B, mu)
    msa = fso.synth_msa(spec, m, n)
    res = fso.segment_long(msa, L, keep_dp=True, debug=True)
    if res["status"] != 0:
        assert res["max_segment_size"] >= m
        return
    _check_segmentation(msa, L, res, check_optimum=(n <= 400))


def test_long_path_sampling_and_threads_do_not_change_results():
    spec = fso.synth_spec(99, 5, 80, 4e-3)
    msa = fso.synth_msa(spec, 30, 1500)
    base = fso.segment_long(msa, 12, sample_rate=1501, threads=1, debug=True)
    assert base["n_samples"] == 1
    for sr, th in ((7, 1), (40, 3), (155, 8), (1500, 2)):
        r = fso.segment_long(msa, 12, sample_rate=sr, threads=th, debug=True)
        assert r["max_segment_size"] == base["max_segment_size"]
        assert np.array_equal(r["traceback"], base["traceback"])
        assert np.array_equal(r["reduced"], base["reduced"])
        assert np.array_equal(r["a"], base["a"]) and np.array_equal(r["d"], base["d"])
    # row-major and column-major views of the same alignment agree
    r = fso.segment_long(np.ascontiguousarray(msa), 12, debug=True)
    assert np.array_equal(r["reduced"], base["reduced"]) and np.array_equal(r["a"], base["a"])


def test_long_path_unreducible_input_reports_failure():
    # iid symbols: every row distinct in every window -> max segment size == m (F8)
    rng = np.random.default_rng(0)
    msa = (rng.integers(0, 4, size=(6, 200)) + 65).astype(np.uint8)
    res = fso.segment_long(msa, 20, debug=True)
    assert res["status"] == 1 and res["max_segment_size"] >= 6


def test_short_path_distinct_rows():
    spec = fso.synth_spec(3, 2, 1000, 0.0)
    msa = fso.synth_msa(spec, 10, 30)
    f, r = fso.segment_short(msa)
    assert r.sum() == 10
    assert len(f) == distinct_count(msa, 0, 30) <= 2
    a, _ = brute_pbwt(msa, 30)
    assert f[0] == a[0]


# ------------------------------------------------------------------ generator

def test_generator_is_stateless_and_sliceable():
    spec = fso.config_spec("C1")
    full = fso.synth_msa(spec, 8, 1000)
    part = fso.synth_msa(spec, 8, 100, c0=300)
    assert np.array_equal(full[:, 300:400], part)
    assert set(np.unique(full).tolist()) <= set(b"ACGT")
    spec5 = fso.config_spec("C5")
    x = fso.synth_msa(spec5, 50, 2000)
    assert set(np.unique(x).tolist()) <= set(b"ACGTRYSWKMBDHVN-")
    frac = np.isin(x, np.frombuffer(b"ACGT", dtype=np.uint8)).mean()
    assert 0.85 < frac < 0.95


def test_config_c1_is_reducible():
    c = fso.CONFIGS["C1"]
    msa = fso.synth_msa(fso.config_spec("C1"), c["m"], c["n"])
    res = fso.segment_long(msa, c["L"], debug=True)
    assert res["status"] == 0 and res["max_segment_size"] < c["m"]
