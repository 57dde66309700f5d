"""CPU proof of round 2's two algorithms on their Python model (tests/proto_shard.py) against the oracle:
the chunk-speculative DP for any chunk length, and the sharded run -- here with the ranks as threads of one
process; tests/test_dist_gloo.py runs the same model as two gloo processes."""
import threading

import numpy as np
import pytest

import fso
import proto_blocks as pb
import proto_shard as ps


def codes_of(msa):
    vals = np.unique(msa)
    return np.searchsorted(vals, msa).astype(np.int64)


def oracle(msa, L):
    return fso.segment_long(msa, L, keep_dp=True, threads=2)


def lists_of(codes, L, B, X):
    m, n = codes.shape
    a, d = np.arange(m, dtype=np.int64), np.zeros(m, dtype=np.int64)
    lists = []
    for k0 in range(0, n, B):
        recs, a2 = pb.phase_c(codes, a, d, k0, min(B, n - k0), X, L)
        lists.extend(recs)
        for k in range(k0, min(n, k0 + B)):
            a, d = pb.colstep(a, d, codes[a, k], k + 1)
    return lists


@pytest.mark.parametrize("m,n,L,seed,forced", [(12, 700, 5, 3, 1), (12, 700, 5, 3, 4), (20, 900, 10, 4, 3), (30, 600, 7, 5, 2),
                                              (16, 1300, 100, 6, 1), (8, 1000, 10, 0x5EED0001, 5)])
def test_speculative_dp_equals_serial_walk(m, n, L, seed, forced):
    """Whatever the chunk length: the fixed point of the sweeps is the oracle's DP array (max, lb, size)."""
    msa = fso.synth_msa(fso.synth_spec(seed, 3, 60, 5e-3, 0), m, n)
    codes = codes_of(msa)
    ref = oracle(msa, L)
    S = ps.dp_schedule(L, n)
    lists = lists_of(codes, L, 64, m)
    r0, mine = ps.chunk_plan(S, [(0, n)], 0, forced=forced)
    stats = {}
    M, LB, SZ, ovf = ps.speculative_dp(lists, S, m, r0, mine, stats=stats)
    assert not ovf
    w = np.ones(n - L + 1, dtype=bool)
    w[n - 2 * L + 1:n - L] = False
    assert np.array_equal(M[w], ref["dp"]["segment_max_size"][w])
    assert np.array_equal(LB[w], ref["dp"]["lb"][w].astype(np.int64))
    assert np.array_equal(SZ[w], ref["dp"]["segment_size"][w])
    assert 2 <= stats["sweeps"] <= len(r0) + 1


def test_rebuilt_rmq_state_equals_incremental():
    """k_spec_rebuild / k_spec_table: masks and the (quirky, rmq.hh:76-79) sparse table in closed form from the keys."""
    rng = np.random.default_rng(7)
    for n in (64, 65, 500, 1500):
        M = rng.integers(0, 6, size=n).astype(np.int64)
        inc = pb.DeviceRmq(n)
        for t in range(n):
            inc.append(t, int(M[t]))
        reb = ps.rmq_from_keys(M, n, n)
        assert reb.K == inc.K and reb.T == inc.T
        for _ in range(300):
            b = int(rng.integers(0, n))
            e = int(rng.integers(b + 1, n + 1))
            assert reb.query(b, e) == inc.query(b, e)


def test_merge_by_thresholds_equals_list_walk():
    """find_segments_greedy through one threshold per boundary == the walk over the lists (proto_blocks.segment)."""
    for seed, m, n, L, X in [(11, 24, 500, 7, 3), (12, 40, 400, 20, 63), (13, 12, 300, 5, 1)]:
        msa = fso.synth_msa(fso.synth_spec(seed, 4, 60, 1e-2, 0), m, n)
        codes = codes_of(msa)
        r = pb.segment(codes, L, 37, X)
        tb, lists = r["traceback"], r["lists"]
        tau = [ps.seg_tau(lists[e[1] - 1], r["max_segment_size"]) for e in tb]
        red, ov = ps.merge_by_thresholds(tb, r["max_segment_size"], tau, lambda qs: [ps.seg_count(lists[c], lb) for c, lb in qs])
        # the walk over the lists themselves (round 1's host loop; lp.cc:335-390)
        want, want_ov = [], False
        cur_lb, prev_size, prev = 0, tb[0][3], 0
        for j in range(1, len(tb)):
            rec = lists[tb[j][1] - 1]
            cnt, known = 0, rec["complete"]
            for v, c in zip(rec["vals"], rec["cnts"]):
                if v > cur_lb:
                    cnt += c
                else:
                    known = True
                    break
            if not known and cnt <= r["max_segment_size"]:
                want_ov = True
                break
            if cnt <= r["max_segment_size"]:
                prev_size = cnt
            else:
                want.append((cur_lb, tb[prev][1], prev_size))
                prev_size, cur_lb = tb[j][3], tb[prev][1]
            prev = j
        want.append((cur_lb, tb[prev][1], prev_size))
        assert ov == want_ov
        if not ov:
            assert red == want


class ThreadRanks:
    """W ranks as threads; allreduce = barrier, reduce, barrier (what ThreadWorld does on the GPU side)."""

    def __init__(self, world):
        self.world, self.bar, self.slots = world, threading.Barrier(world), [None] * world

    def fn(self, rank):
        def allreduce(arr, op):
            self.slots[rank] = arr
            self.bar.wait()
            acc = self.slots[0].copy()
            for s in self.slots[1:]:
                acc = acc + s if op == 0 else np.maximum(acc, s)
            self.bar.wait()
            return acc
        return allreduce


def run_ranks(codes, L, B, X, world, forced=0):
    tr = ThreadRanks(world)
    out, errs = [None] * world, []

    def work(r):
        try:
            out[r] = ps.segment_sharded(codes, L, B, X, r, world, tr.fn(r), forced_rounds=forced)
        except BaseException as e:
            errs.append(e)
            tr.bar.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if errs:
        raise errs[0]
    return out


def check_ranks(out, ref, n, L):
    w = np.ones(n - L + 1, dtype=bool)
    w[n - 2 * L + 1:n - L] = False
    seen = set()
    for r in out:
        assert r["max_segment_size"] == ref["max_segment_size"]
        assert np.array_equal(r["M"][w], ref["dp"]["segment_max_size"][w])
        assert np.array_equal(r["LB"][w], ref["dp"]["lb"][w].astype(np.int64))
        assert np.array_equal(r["SZ"][w], ref["dp"]["segment_size"][w])
        assert [(a, b, c) for a, b, c in r["reduced"]] == [(int(x["lb"]), int(x["rb"]), int(x["segment_size"])) for x in ref["reduced"]]
        for i, (sa, sd) in r["snaps"].items():
            assert i not in seen
            seen.add(i)
            assert np.array_equal(sa, ref["a"][i]) and np.array_equal(sd, ref["d"][i])
    assert seen == set(range(len(ref["reduced"])))          # every boundary state on exactly one rank


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("m,n,L,B,X,seed", [(12, 700, 5, 16, 63, 21), (20, 800, 10, 25, 2, 22), (9, 1100, 30, 40, 63, 23)])
def test_sharded_model_matches_oracle(world, m, n, L, B, X, seed):
    msa = fso.synth_msa(fso.synth_spec(seed, 3, 70, 5e-3, 0), m, n)
    codes = codes_of(msa)
    ref = oracle(msa, L)
    if ref["status"] != 0:
        pytest.skip("input cannot be reduced")
    out = run_ranks(codes, L, B, X, world, forced=2)
    check_ranks(out, ref, n, L)
    cols = out[0]["geometry"]["cols"]
    assert cols[0][0] == 0 and cols[-1][1] == n and all(a[1] == b[0] for a, b in zip(cols, cols[1:]))
