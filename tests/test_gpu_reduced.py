"""[r5] Phase C and pass 2 on the blocks' REPRESENTATIVE rows (csrc/fseq_reduced.hpp) against the oracle on all rows.

The library takes the representatives where they are clearly fewer than the rows (the founder mosaics of BASELINE.json);
here FSEQ_REDUCED_ALWAYS forces them wherever a block has fewer representatives than rows, so that every regime of the
reduced path meets the oracle on small inputs too: one-wave and multi-wave configurations, the list wave, 2 / 4 / 8 bits per
symbol, 16-bit LDS words, streamed rows (the reduced alignment), blocks that run on all rows beside reduced ones, lists that
reach below what the representatives vouch for (the block is flagged and run again on all rows), the cached plan of a
second run, the list-capacity retries, pass 2 from the reduced stride states, and a sharded run.
DP array, traceback, merged segments, boundary states (compare_long) and the per-column lists are bit-exact."""
import importlib

import numpy as np
import pytest

import fso
from test_gpu_parity import compare_long, run_gpu
from test_gpu_shard import run_world, check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("founder-sequences_amd")


@pytest.fixture()
def always(monkeypatch):
    monkeypatch.setenv("FSEQ_REDUCED_ALWAYS", "1")
    return monkeypatch


SHAPES = [
    # m, n, L, K, Brec, mu, seed, kind, block_len
    (200, 3000, 15, 6, 400, 1e-3, 61, 0, 100),           # one-wave configurations
    (600, 4000, 30, 8, 600, 5e-4, 62, 0, 128),
    (2500, 6000, 50, 16, 2000, 1e-4, 0x5EED0002, 0, 0),   # BASELINE C2 / C3 rows
    (2504, 5000, 100, 24, 2500, 1e-4, 0x5EED0003, 0, 245),
    (900, 3000, 40, 10, 500, 5e-4, 63, 1, 100),          # sigma = 16: two digit passes
    (10000, 3000, 100, 32, 1500, 1e-4, 0x5EED0005, 1, 200),   # BASELINE C5 rows
    (9000, 2000, 40, 200, 400, 2e-4, 64, 0, 128),        # hundreds of founders: 256 .. 1024-thread configurations
    (12000, 2400, 20, 12, 300, 3e-4, 65, 0, 100),        # streamed rows: the reduced alignment, the streamed chain step
    (30000, 1500, 25, 40, 500, 1e-4, 66, 0, 150),
    (13000, 1200, 20, 12, 300, 3e-4, 69, 1, 100),        # streamed rows of 4-bit symbols
    (70000, 600, 20, 30, 200, 1e-4, 70, 0, 100),         # a column of more than 16 KB: two LDS-DMA pieces per lane in k_reduce_msa_lds
    (300, 2000, 3, 5, 100, 2e-3, 67, 0, 16),             # L = 3: thresholds right behind the column
    (64, 1500, 200, 4, 300, 1e-3, 68, 0, 50),            # L >> block length: every early block is exact (vmin = 1)
]


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,kind,B", SHAPES)
def test_reduced_run_matches_oracle(pkg, always, m, n, L, K, Brec, mu, seed, kind, B):
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    ctx, ref = compare_long(pkg, msa, L, block_len=B)
    t = ctx.timings()
    assert t["reduced_blocks"] > 0, t
    assert t["reduced_rows_mean"] < m
    # a second run takes the cached plan (no read-back of the counts) and must give the same
    tb = ctx.traceback().copy()
    ctx.run()
    assert np.array_equal(ctx.traceback(), tb) and ctx.timings()["reduced_blocks"] == t["reduced_blocks"]
    for i in (0, len(ctx.reduced_traceback()) // 2, len(ctx.reduced_traceback()) - 1):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(a, ref["a"][i]) and np.array_equal(d, ref["d"][i])


def test_reduced_alignment_by_gathers(pkg, always):
    """FSEQ_REDUCED_MSA_GATHER: the representatives' columns gathered from memory (the form for columns that do not fit two LDS
    buffers) instead of through LDS."""
    always.setenv("FSEQ_REDUCED_MSA_GATHER", "1")
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(12000, 2400, 20, 12, 300, 3e-4, 65, 0, 100), (13000, 1200, 20, 12, 300, 3e-4, 69, 1, 100)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctx, _ = compare_long(pkg, msa, L, block_len=B)
        assert ctx.timings()["reduced_blocks"] > 0


def _lists_match(ctx, msa, L, every=5):
    m, n = msa.shape
    p = fso.Pbwt(msa, debug=False)
    X = ctx.timings()["list_cap_used"]
    for k in range(n):
        p.step()
        if k % every and k < n - 3:
            continue
        v, c = p.counts()
        gv, gc, cnt0, complete = ctx.debug_column_list(k)
        thr = max(0, k + 2 - L)
        rec = v >= thr
        ev = np.concatenate([[k + 1], v[~rec][::-1]])
        ec = np.concatenate([[c[rec].sum()], c[~rec][::-1]])
        assert complete or gc[1:].sum() > X, k
        assert np.array_equal(gv, ev[:len(gv)]) and np.array_equal(gc, ec[:len(gc)]), k
        assert complete == (len(gv) == len(ev)), k
        if complete:
            assert cnt0 == (c[0] if v[0] == 0 else 0), k


@pytest.mark.parametrize("ew", [False, True])
def test_reduced_lists_match_oracle(pkg, always, ew):
    if ew:
        always.setenv("FSEQ_REDUCED_EW", "1")            # small blocks on two waves: the list on a wave of its own
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(200, 1500, 15, 6, 300, 2e-3, 71, 0, 100), (700, 1200, 25, 8, 400, 1e-3, 72, 1, 64),
                                                   (3000, 900, 30, 60, 300, 3e-4, 73, 0, 128)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctx = run_gpu(pkg, msa, L, block_len=B)
        assert ctx.timings()["reduced_blocks"] > 0
        _lists_match(ctx, msa, L)


def test_lists_below_the_floor_send_the_block_to_all_rows(pkg, always):
    """No margin beyond the list capacity in the choice of vmin, and a capacity the lists outgrow: lists reach below what the
    representatives vouch for, the blocks are flagged, the attempt runs again with them on all rows -- and the result is
    the oracle's (exactness never rests on the choice of vmin)."""
    always.setenv("FSEQ_REDUCED_MARGIN", "0")
    redone = 0
    for (m, n, L, K, Brec, mu, seed, B, cap) in [(400, 4000, 20, 6, 150, 3e-3, 81, 100, 8), (1200, 3000, 30, 10, 200, 2e-3, 82, 128, 24),
                                                  (12000, 1800, 20, 12, 150, 1e-3, 83, 100, 16)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, 0), m, n)
        ctx, _ = compare_long(pkg, msa, L, block_len=B, list_cap=cap)
        t = ctx.timings()
        redone += t["reduced_redone"]
        _lists_match(ctx, msa, L, every=11)
    assert redone > 0
    # an attempt that runs again leaves its phase range like any other (roctx pushes and pops stay balanced)
    pushes, pops, _ = pkg.debug_ranges()
    assert pushes == pops


def test_blocks_with_too_many_representatives_run_on_all_rows(pkg, always):
    """FSEQ_REDUCED_CAP: blocks with more representatives than the cap run on all rows beside the reduced ones (lists, and
    their boundaries in pass 2 from the block's start), LDS-resident and streamed."""
    for (m, n, L, K, Brec, mu, seed, B) in [(500, 5000, 20, 8, 700, 1e-3, 91, 100), (12000, 2000, 20, 10, 500, 2e-4, 92, 100)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, 0), m, n)
        always.delenv("FSEQ_REDUCED_CAP", raising=False)
        mean = run_gpu(pkg, msa, L, block_len=B).timings()["reduced_rows_mean"]
        always.setenv("FSEQ_REDUCED_CAP", str(mean))         # about half of the blocks have more
        ctx, _ = compare_long(pkg, msa, L, block_len=B)
        t = ctx.timings()
        assert 0 < t["reduced_blocks"] < t["n_blocks"], t
        _lists_match(ctx, msa, L, every=13)


def test_reduced_run_and_the_run_on_all_rows_agree_at_depth(pkg, monkeypatch):
    """A long input with the library's own choice (the representatives on a founder mosaic) against FSEQ_NO_REDUCED: the same
    traceback, segments and boundary states, lists included."""
    m, n, L = 2504, 60000, 100
    spec = fso.synth_spec(0x5EED0003, 24, 5000, 1e-4, 0)
    msa = fso.synth_msa(spec, m, n)
    ctx = run_gpu(pkg, msa, L)
    assert ctx.timings()["reduced_blocks"] == ctx.timings()["n_blocks"]
    monkeypatch.setenv("FSEQ_NO_REDUCED", "1")
    ref = run_gpu(pkg, msa, L)
    assert ref.timings()["reduced_blocks"] == 0
    assert np.array_equal(ctx.traceback(), ref.traceback()) and np.array_equal(ctx.reduced_traceback(), ref.reduced_traceback())
    for k in range(0, n, 997):
        a, b = ctx.debug_column_list(k), ref.debug_column_list(k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3], k
    S = len(ctx.reduced_traceback())
    for i in list(range(0, S, max(1, S // 25))) + [S - 1]:
        a, d = ctx.boundary_state(i)
        ra, rd = ref.boundary_state(i)
        assert np.array_equal(a, ra) and np.array_equal(d, rd), i


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reduced_run_matches_oracle(pkg, always, world):
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(600, 8000, 30, 8, 600, 5e-4, 101, 0, 100), (12000, 3000, 20, 12, 300, 3e-4, 102, 0, 60)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctxs = run_world(pkg, world, lambda c: c.set_sequences(msa), m, n, L, block_len=B)
        check_against_oracle(pkg, ctxs, msa, L)
        assert all(c.timings()["reduced_blocks"] > 0 for c in ctxs)
