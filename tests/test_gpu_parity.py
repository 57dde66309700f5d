"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.
Bit-exact: DP array, traceback, merged segments, boundary (a, d) states, block states, lists."""
import importlib

import numpy as np
import pytest

import fso

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("founder-sequences_amd")


def run_gpu(pkg, msa, L, **kw):
    m, n = msa.shape
    ctx = pkg.SegmentationContext(m, n, L, **kw)
    ctx.set_sequences(msa)
    try:
        ctx.run()
    except pkg.NoReduction:
        pass
    return ctx


def compare_long(pkg, msa, L, check_dp=True, **kw):
    m, n = msa.shape
    ref = fso.segment_long(msa, L, keep_dp=check_dp, threads=4)
    ctx = run_gpu(pkg, msa, L, **kw)
    assert ctx.result.short_path == 0
    assert ctx.result.max_segment_size == ref["max_segment_size"]
    if check_dp:
        lb, mx, sz = ctx.debug_dp()
        dp = ref["dp"]
        written = np.ones(len(dp), dtype=bool)
        written[n - 2 * L + 1:n - L] = False
        assert np.array_equal(mx[written], dp["segment_max_size"][written])
        assert np.array_equal(lb[written], dp["lb"][written].astype(np.uint32))
        assert np.array_equal(sz[written], dp["segment_size"][written])
    tb = ctx.traceback()
    assert len(tb) == len(ref["traceback"])
    for f in ("lb", "rb", "segment_max_size", "segment_size"):
        assert np.array_equal(tb[f], ref["traceback"][f]), f
    if ref["status"] != 0:
        assert ctx.result.segment_count == 0
        return ctx, ref
    red = ctx.reduced_traceback()
    assert len(red) == len(ref["reduced"])
    for f in ("lb", "rb", "segment_size"):
        assert np.array_equal(red[f], ref["reduced"][f]), f
    for i in range(len(red)):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(a, ref["a"][i]), i
        assert np.array_equal(d, ref["d"][i]), i
    return ctx, ref


CASES = [
    # m, n, L, K, Brec, mu, seed, kind, block_len
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 0, 0),       # BASELINE config C1
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 0, 1),       # B = 1
    (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 0, 1000),    # one block
    (24, 400, 7, 4, 60, 1e-2, 11, 0, 37),
    (40, 300, 20, 5, 50, 5e-3, 12, 0, 64),
    (16, 64, 32, 2, 30, 1e-2, 13, 0, 10),                # n == 2L
    (12, 200, 1, 3, 20, 2e-2, 15, 0, 7),                 # L = 1
    (70, 500, 9, 6, 45, 8e-3, 21, 0, 50),                # 64 < m <= 448 : one wave, 7 rows per lane
    (300, 2000, 25, 8, 200, 2e-3, 22, 0, 0),             # T = 256, E = 5
    (1000, 3000, 30, 10, 300, 1e-3, 23, 0, 100),
    (2500, 4000, 50, 16, 2000, 1e-4, 0x5EED0002, 0, 0),  # C2 rows, shortened columns (T=256, E=11)
    (3000, 1500, 40, 12, 500, 5e-4, 24, 0, 128),         # T = 1024, E = 7
    (30, 300, 8, 4, 50, 1e-2, 5, 1, 32),                 # sigma = 16
    (600, 1500, 20, 8, 300, 2e-3, 25, 1, 64),            # sigma = 16, T = 256
    (1, 50, 5, 1, 10, 0.0, 26, 0, 8),                    # single row: cannot reduce
    (9000, 1200, 40, 16, 400, 2e-4, 27, 0, 128),         # 16-bit LDS state (m > 7168), sigma = 4
    (10000, 2500, 100, 32, 500, 1e-4, 0x5EED0005, 1, 0), # BASELINE config C5 rows (m = 10,000, sigma = 16), shortened columns
    (300, 5000, 10, 8, 200, 2e-3, 28, 0, 16),            # 313 blocks: three-level phase B (7 x 7 x 7)
    (64, 4000, 6, 5, 90, 5e-3, 29, 0, 4),                # 1000 blocks of 4 columns: three levels, ragged last groups
]


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,kind,B", CASES)
def test_long_path_matches_oracle(pkg, m, n, L, K, Brec, mu, seed, kind, B):
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    compare_long(pkg, msa, L, block_len=B)


@pytest.mark.parametrize("sigma,m,n,L", [(3, 90, 600, 10), (40, 120, 800, 12), (200, 300, 500, 8), (256, 64, 300, 5)])
def test_wide_and_odd_alphabets(pkg, sigma, m, n, L):
    """Any byte alphabet: a column is ceil(log2(sigma)/2) two-bit partition passes."""
    rng = np.random.default_rng(sigma)
    founders = rng.integers(0, sigma, size=(5, n))
    pick = rng.integers(0, 5, size=(m, (n + 99) // 100))
    msa = np.empty((m, n), dtype=np.uint8)
    for b in range(pick.shape[1]):
        msa[:, b * 100:(b + 1) * 100] = founders[pick[:, b], b * 100:(b + 1) * 100]
    noise = rng.random((m, n)) < 2e-3
    msa[noise] = rng.integers(0, sigma, size=int(noise.sum()))
    if sigma < 256:
        msa = (msa.astype(np.uint16) * (255 // sigma)).astype(np.uint8)      # spread over the byte range
    compare_long(pkg, msa, L, block_len=48)


@pytest.mark.parametrize("m", [2, 63, 64, 65, 448, 449, 1280, 1281, 2240, 2241, 2560, 2561, 3584, 3585, 4800, 4801, 5120, 5121, 6720, 6721, 7168, 7169,
                               8640, 8641, 9216, 9217, 9600, 9601, 10240, 10241, 10560, 10561, 11264, 11265])
def test_kernel_configuration_boundaries(pkg, m):
    """Row counts on both sides of every <T,E> capacity (64, 448, 1280, 2560, 3584, 7168 | packed) and of the
    configurations that keep a wave free for the per-column lists (448 x 5 = 2240, 448 x 6 > 2560, 960 x 5 = 4800, 960 x 7 = 6720);
    16-bit LDS state: 1024 (960 with the list wave) x 9, 10, 11 rows, then the streamed kernels."""
    n, L = 160, 8
    msa = fso.synth_msa(fso.synth_spec(1000 + m, 7, 40, 4e-3), m, n)
    compare_long(pkg, msa, L, check_dp=True, block_len=33)


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,kind,B", [
    (12000, 500, 20, 12, 120, 3e-4, 41, 0, 64),          # three tiles of 4096 rows
    (20000, 300, 15, 16, 100, 2e-4, 42, 1, 50),          # five tiles, sigma = 16 (two digit passes per column)
    (70000, 200, 12, 20, 64, 1e-4, 43, 0, 40),           # more than 65535 rows: 32-bit tile carry
    (100000, 160, 10, 64, 50, 5e-5, 0x5EED0004, 0, 0),   # BASELINE config C4 rows, shortened columns
    (200001, 48, 8, 30, 16, 5e-5, 44, 0, 0),             # 2-bit packed columns: 50,001 bytes staged per column
    (11300, 2400, 4, 3, 40, 2e-3, 45, 0, 1200),          # two blocks, many segments: pass 2 in several launches
    (12000, 3000, 10, 12, 200, 3e-4, 46, 0, 12),         # 250 blocks: three-level phase B on the streamed kernels
    (540000, 40, 6, 40, 10, 2e-5, 47, 0, 0),             # value ids beyond 2^19: the keyed scan of the tiles does not apply
])
def test_streamed_state_for_large_m(pkg, m, n, L, K, Brec, mu, seed, kind, B):
    """m > 11,264: the block order streams through HBM/L2 in tiles (fseq_stream.hpp)."""
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    compare_long(pkg, msa, L, block_len=B)


@pytest.mark.parametrize("wide", [False, True])
def test_streamed_phase_a_id_width(pkg, monkeypatch, wide):
    """The streamed key-space tree keeps its ids as halfwords and runs a block again with 32-bit ids once a range has
    more than 65,536 distinct keys: 70,000 random rows are all distinct after a dozen columns (the fallback), a
    mosaic of few founders never gets there; FSEQ_BLOCKKEYS_WIDE forces 32 bits from the start."""
    monkeypatch.setenv("FSEQ_NO_BLOCKTRIE", "1")             # (the mosaic would be the trie's otherwise)
    if wide:
        monkeypatch.setenv("FSEQ_BLOCKKEYS_WIDE", "1")
    rng = np.random.default_rng(11)
    msa = (rng.integers(0, 4, size=(70000, 64)) + 65).astype(np.uint8)
    compare_long(pkg, msa, 8, block_len=32)
    msa = fso.synth_msa(fso.synth_spec(48, 20, 64, 1e-4, 0), 70000, 96)
    compare_long(pkg, msa, 8, block_len=40)


def test_streamed_phase_a_leaves_one_by_one(pkg, monkeypatch):
    """The streamed key-space tree ranks two leaves (16 columns of 2-bit symbols) at once from the packed columns
    (bk_pair_leaf); FSEQ_BLOCKKEYS_SINGLE keeps the leaves one by one, which is also what a pair falls back to when its
    D_lo x D_hi bitmap does not fit (random rows: 4^8 distinct words per leaf) and what wider symbols use."""
    monkeypatch.setenv("FSEQ_NO_BLOCKTRIE", "1")
    monkeypatch.setenv("FSEQ_BLOCKKEYS_SINGLE", "1")
    msa = fso.synth_msa(fso.synth_spec(48, 20, 64, 1e-4, 0), 30000, 200)
    compare_long(pkg, msa, 8, block_len=72)
    monkeypatch.delenv("FSEQ_BLOCKKEYS_SINGLE")
    rng = np.random.default_rng(13)
    msa = (rng.integers(0, 4, size=(30000, 80)) + 65).astype(np.uint8)             # every pair falls back
    compare_long(pkg, msa, 8, block_len=40)
    msa = fso.synth_msa(fso.synth_spec(49, 20, 64, 1e-4, 0), 30000, 100)            # odd leaf counts, a last leaf of 4 columns
    compare_long(pkg, msa, 8, block_len=44)


def test_streamed_phase_a_with_more_rows_than_bitmap_bits(pkg, monkeypatch):
    """A merge of the streamed key-space tree is sliced by whole `hi` values, so one value's Dlo <= m keys must fit the
    LDS bitmap: with a forced small bitmap (2048 words = 65,536 bits) and 70,000 all-distinct rows phase A must take the
    column sweep instead (it used to overrun the bitmap)."""
    monkeypatch.setenv("FSEQ_BLOCKKEYS_CAP", "2048")
    rng = np.random.default_rng(12)
    msa = (rng.integers(0, 4, size=(70000, 48)) + 65).astype(np.uint8)
    ctx, _ = compare_long(pkg, msa, 8, block_len=24)
    assert ctx.timings()["phase_a_fallbacks"] == 0           # (the key-space kernel did not run at all)


def _block_states_match(ctx, msa, every=1):
    n = msa.shape[1]
    bl, nbk = ctx.timings()["block_len"], ctx.timings()["n_blocks"]
    p = fso.Pbwt(msa)
    for b in range(0, nbk + 1, every):
        while p.idx < min(n, b * bl):
            p.step()
        a, d = ctx.debug_block_state(b)
        assert np.array_equal(a, p.a) and np.array_equal(d, p.d), b


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,B,given_up", [
    (12000, 500, 20, 12, 120, 3e-4, 41, 64, 0),           # SL = 2 (750 words of 16 rows), blocks of four groups
    (20003, 330, 15, 16, 100, 2e-4, 61, 50, 0),           # m not a multiple of 16, a last group of 2 columns, a last block of 30
    (40000, 260, 12, 20, 1000, 1e-4, 62, 100, 0),         # SL = 4
    (40000, 260, 12, 20, 64, 1e-4, 62, 100, 0),           # ... every row changes its founder at column 128: 8,000 nodes with siblings in that level
    (100000, 200, 10, 64, 50, 5e-5, 0x5EED0004, 0, None), # BASELINE C4's rows: SL = 8, the library's own block length
    (131072, 96, 8, 30, 1000, 5e-5, 63, 37, 0),           # the most rows the trie takes
])
def test_streamed_phase_a_trie(pkg, monkeypatch, m, n, L, K, Brec, mu, seed, B, given_up):
    """Streamed rows at 2 bits per symbol: phase A sorts the rows into their key classes with a trie over 16-column words
    (an exact hash table of (class, word) pairs per level) and ranks the trie instead of the rows (fseq_blocktrie.hpp).
    Every block boundary state against the oracle's pBWT, the run against the oracle, the same again without the trie; the
    second run on a context launches the trie alone (nothing was given up)."""
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, 0), m, n)
    ctx, ref = compare_long(pkg, msa, L, block_len=B)
    t = ctx.timings()
    assert (given_up is None or t["phase_a_trie_given_up"] == given_up) and t["phase_a_given_up"] == 0, t
    _block_states_match(ctx, msa)
    tb = ctx.traceback().copy()
    ctx.run()                                                # the trie alone (if nothing was given up)
    assert np.array_equal(ctx.traceback(), tb) and ctx.timings()["phase_a_trie_given_up"] == t["phase_a_trie_given_up"]
    _block_states_match(ctx, msa, every=3)
    monkeypatch.setenv("FSEQ_NO_BLOCKTRIE", "1")
    ctx2 = run_gpu(pkg, msa, L, block_len=B)
    assert np.array_equal(ctx2.traceback(), tb)


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,kind,B", [
    (2504, 3000, 20, 16, 400, 5e-4, 81, 0, 245),          # 2 bits, T = 256 (BASELINE C3's rows), a last group of 5 columns
    (2504, 1200, 20, 2504, 500, 1e-2, 91, 0, 245),        # ... every row its own founder
    (5000, 900, 15, 30, 200, 3e-4, 82, 1, 100),           # 4 bits (sigma = 16): groups of 8 columns, T = 512
    (10000, 700, 30, 24, 300, 2e-4, 83, 1, 200),          # BASELINE C5's rows: T = 1024, the trie by itself
    (10000, 500, 30, 10000, 500, 1e-3, 84, 1, 200),       # ... all rows distinct: more new classes in a group than the pair table holds
    (3000, 400, 10, 40, 100, 5e-4, 85, 2, 64),            # 8 bits per symbol: groups of 4 columns
])
def test_phase_a_trie_on_lds_resident_rows(pkg, monkeypatch, m, n, L, K, Brec, mu, seed, kind, B):
    """The trie of fseq_blocktrie.hpp in front of the LDS-resident key-space tree (by itself from 6,145 rows on; here on every
    row count): 2, 4 and 8 bits per symbol, 256 / 512 / 1024 threads; what it gives up is the tree's (and what the tree gives
    up the sweep's).  Block boundary states against the oracle's pBWT, the run against the oracle, a second run identical."""
    monkeypatch.setenv("FSEQ_BLOCKTRIE_ALWAYS", "1")
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, min(kind, 1)), m, n)
    if kind == 2:
        msa = msa.copy()
        msa[:, 1::2] += 32                                   # (up to 32 characters: 8 bits per symbol)
    ctx, ref = compare_long(pkg, msa, L, check_dp=False, block_len=B)
    t = ctx.timings()
    _block_states_match(ctx, msa, every=2)
    tb = ctx.traceback().copy()
    try:
        ctx.run()
    except pkg.NoReduction:
        pass
    t2 = ctx.timings()
    assert np.array_equal(ctx.traceback(), tb) and t2["phase_a_trie_given_up"] in (t["phase_a_trie_given_up"], 0), (t, t2)


def _gapped_sigma16(m, n, seed, dense_share):
    """A sigma = 16 alignment in which a share of the columns carries at most four codes (an order-preserving image of the
    generator's column: the codes' ranks in the column, cut to 0..3, mapped into four codes spread over the alphabet)."""
    msa = fso.synth_msa(fso.synth_spec(seed, 24, 300, 2e-4, 1), m, n).copy()
    codes = np.unique(msa)
    assert len(codes) > 8
    rng = np.random.default_rng(seed)
    for k in np.nonzero(rng.random(n) < dense_share)[0]:
        pick = np.sort(rng.choice(len(codes), size=int(rng.integers(1, 5)), replace=False))      # 1 .. 4 codes, any digits
        col = np.searchsorted(codes, msa[:, k])
        msa[:, k] = codes[pick[col % len(pick)]]
    return msa


@pytest.mark.parametrize("m,n,L,share,B", [
    (10000, 900, 30, 0.8, 200),          # BASELINE C5's rows (the second digit rides with the row where a column takes two passes)
    (3000, 700, 12, 0.5, 64),            # 32-bit LDS state
    (10000, 400, 20, 1.0, 100),          # every column dense
])
def test_columns_with_at_most_four_codes_take_one_pass(pkg, monkeypatch, m, n, L, share, B):
    """4-bit symbols: a column with at most four present codes is partitioned ONCE, by the codes' ranks among the present ones
    (k_column_presence, k_columns); the remap keeps the codes' order, so the result is the two digit passes'.  The run against the
    oracle (DP array, traceback, segments, boundary states), block states against its pBWT, and the same with every column in two
    passes (FSEQ_NO_DENSE_COLUMNS)."""
    msa = _gapped_sigma16(m, n, 300 + m // 1000, share)
    ctx, ref = compare_long(pkg, msa, L, block_len=B)
    _block_states_match(ctx, msa, every=2)
    tb = ctx.traceback().copy()
    ctx.run()
    assert np.array_equal(ctx.traceback(), tb)
    monkeypatch.setenv("FSEQ_NO_DENSE_COLUMNS", "1")
    ctx2 = run_gpu(pkg, msa, L, block_len=B)
    assert np.array_equal(ctx2.traceback(), tb)


def test_streamed_phase_a_trie_gives_blocks_up(pkg):
    """What does not fit the trie's table goes to the key-space tree, block by block: columns 0..95 of 30,000 rows are a
    mosaic of few founders (the trie's), columns 96..191 random (every row its own key after eight columns: more than 12,288
    nodes in a level), then a mosaic again.  A level with thousands of siblings under one parent (~2,900 distinct words in the last
    group of a block, each row otherwise like its founder) is ranked by the trie: its list goes through LDS in pieces."""
    m = 30000
    rng = np.random.default_rng(71)
    a = fso.synth_msa(fso.synth_spec(72, 10, 40, 2e-4, 0), m, 288)
    msa = a.copy()
    codes = np.unique(a)
    msa[:, 96:192] = codes[rng.integers(0, len(codes), size=(m, 96))]
    ctx, _ = compare_long(pkg, msa, 8, block_len=48)
    t = ctx.timings()
    assert t["n_blocks"] == 6 and t["phase_a_trie_given_up"] == 2, t
    _block_states_match(ctx, msa)
    tb = ctx.traceback().copy()
    try:
        ctx.run()                                            # (trie, then the tree on the two blocks, as the first time)
    except pkg.NoReduction:
        pass
    assert np.array_equal(ctx.traceback(), tb) and ctx.timings()["phase_a_trie_given_up"] == 2
    # many siblings under one parent
    msa = fso.synth_msa(fso.synth_spec(73, 6, 400, 0.0, 0), m, 96)
    codes = np.unique(msa)
    words = rng.integers(0, len(codes), size=(3000, 8))
    msa[:3000, 88:96] = codes[words]
    ctx, _ = compare_long(pkg, msa, 8, block_len=96)
    assert ctx.timings()["phase_a_trie_given_up"] == 0
    _block_states_match(ctx, msa)


@pytest.mark.parametrize("fan", [0, 2, 3, 5, 64])
def test_phase_b_recursion_with_any_group_size(pkg, monkeypatch, fan):
    """Phase B composes groups of G key blocks level after level until at most G are left (G = 4 by itself): 600
    and 37 blocks with G = 2 (nine levels), 3, 5, 64 (one level, then a chain of ten), on the LDS-resident and on the
    streamed kernels; every block boundary state is compared with the oracle's pBWT."""
    if fan:
        monkeypatch.setenv("FSEQ_CHAIN_FAN", str(fan))
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(300, 6000, 25, 8, 200, 2e-3, 51, 0, 10), (900, 600, 12, 10, 50, 1e-3, 52, 1, 16),
                                                  (12000, 900, 10, 12, 200, 3e-4, 46, 0, 12)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctx, _ = compare_long(pkg, msa, L, block_len=B)
        bl, nbk = ctx.timings()["block_len"], ctx.timings()["n_blocks"]
        p = fso.Pbwt(msa)
        for b in range(0, nbk + 1, max(1, nbk // 7)):
            while p.idx < min(n, b * bl):
                p.step()
            a, d = ctx.debug_block_state(b)
            assert np.array_equal(a, p.a) and np.array_equal(d, p.d), (m, n, b, fan)


@pytest.mark.parametrize("form", ["FSEQ_CHAIN_STREAM_SINGLE", "FSEQ_CHAIN_STREAM_PASSES"])
def test_streamed_phase_b_forms(pkg, monkeypatch, form):
    """Streamed rows: a chain step of phase B is a stable radix sort by block rank + range maxima spread over the chip
    (fseq_chainsort.hpp, the default); the same step on one workgroup per chain, and the two-bit digit passes it replaced,
    stay as tested alternatives -- block boundary states against the oracle's pBWT in every form, few and many keys per
    block (a mosaic of few founders; random rows: every row its own key, 15-bit ranks = two radix passes)."""
    monkeypatch.setenv(form, "1")
    rng = np.random.default_rng(17)
    for msa, L, B in [(fso.synth_msa(fso.synth_spec(46, 12, 200, 3e-4, 0), 12000, 900), 10, 12),
                      ((rng.integers(0, 4, size=(30000, 160)) + 65).astype(np.uint8), 8, 16),
                      (fso.synth_msa(fso.synth_spec(47, 3, 400, 0.0, 0), 11300, 300), 10, 20)]:
        ctx, _ = compare_long(pkg, msa, L, check_dp=False, block_len=B)
        m, n = msa.shape
        bl = ctx.timings()["block_len"]
        p = fso.Pbwt(msa)
        for b in range(0, ctx.timings()["n_blocks"] + 1, max(1, ctx.timings()["n_blocks"] // 6)):
            while p.idx < min(n, b * bl):
                p.step()
            a, d = ctx.debug_block_state(b)
            assert np.array_equal(a, p.a) and np.array_equal(d, p.d), (form, m, n, b)


def test_streamed_pass_2_from_absolute_states(pkg, monkeypatch):
    """Streamed rows: pass 2 replays the columns from phase C's stride states in id form on phase C's own tile step (the
    default, every other streamed test); FSEQ_SS_ABSOLUTE keeps the states as divergences and pass 2 on the first form's
    tile step -- same boundary states, two blocks with many segments, sigma = 16 and a shape with few stride states."""
    monkeypatch.setenv("FSEQ_SS_ABSOLUTE", "1")
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(11300, 2400, 4, 3, 40, 2e-3, 45, 0, 1200), (20000, 300, 15, 16, 100, 2e-4, 42, 1, 50), (12000, 3000, 10, 12, 200, 3e-4, 46, 0, 12)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        compare_long(pkg, msa, L, check_dp=False, block_len=B)


def test_phase_b_and_pass_2_with_the_plain_scan(pkg, monkeypatch):
    """Phase B and pass 2 scan keys (count << shift | divergence) while n fits the shift of their configuration -- every
    test shape does; FSEQ_PLAIN_SCAN keeps the has-based scan, which long inputs (BASELINE C5: n = 10^6 on the
    1024 x 11 kernels) use, covered on the small shapes too."""
    monkeypatch.setenv("FSEQ_PLAIN_SCAN", "1")
    for (m, n, L, K, Brec, mu, seed, kind, B) in SPEC_SHAPES[:2] + [(2500, 3000, 30, 16, 500, 1e-3, 91, 0, 100), (9000, 1200, 40, 16, 400, 2e-4, 27, 0, 128),
                                                                     (5000, 700, 20, 9, 100, 1e-3, 92, 1, 64)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        compare_long(pkg, msa, L, block_len=B)


def test_phase_b_and_pass_2_with_occurrence_keys(pkg, monkeypatch):
    """n >= 2^shift (BASELINE C5: n = 10^6 on the 1024 x 10 kernels, 18 bits) takes keys whose count part counts lanes and
    waves that hold a bucket instead of rows (partition_step KO, any n < 2^25); forced here on shapes of every kernel
    family, one-wave workgroups included.  The streamed kernels always take them: their has-based scan under
    FSEQ_PLAIN_SCAN is the second half."""
    shapes = SPEC_SHAPES[:2] + [(2500, 3000, 30, 16, 500, 1e-3, 91, 0, 100), (9000, 1200, 40, 16, 400, 2e-4, 27, 0, 128),
                                (5000, 700, 20, 9, 100, 1e-3, 92, 1, 64), (60, 900, 12, 5, 50, 2e-3, 93, 0, 32), (400, 800, 16, 7, 80, 1e-3, 94, 0, 40)]
    monkeypatch.setenv("FSEQ_OCCURRENCE_KEYS", "1")
    for (m, n, L, K, Brec, mu, seed, kind, B) in shapes:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        compare_long(pkg, msa, L, block_len=B)
    monkeypatch.delenv("FSEQ_OCCURRENCE_KEYS")
    monkeypatch.setenv("FSEQ_PLAIN_SCAN", "1")
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(12000, 500, 20, 12, 120, 3e-4, 41, 0, 64), (20000, 300, 15, 16, 100, 2e-4, 42, 1, 50)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        compare_long(pkg, msa, L, block_len=B)


def test_streamed_phase_c_with_the_plain_scan(pkg, monkeypatch):
    """The streamed tiles scan their running maxima as keys (count << 19 | value id) when the ids allow it; the
    has-based scan they fall back to otherwise must give the same results."""
    monkeypatch.setenv("FSEQ_STREAM_PLAIN_SCAN", "1")
    for (m, n, L, K, Brec, mu, seed, kind, B) in [(12000, 500, 20, 12, 120, 3e-4, 41, 0, 64), (20000, 300, 15, 16, 100, 2e-4, 42, 1, 50)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        compare_long(pkg, msa, L, block_len=B)


def test_unsupported_shape_fails_loudly(pkg):
    ctx = pkg.SegmentationContext(600000, 64, 8)         # more rows than one staged column allows (147,456 bytes at 2 bits per row)
    ctx.generate_synthetic(1, 4, 16, 1e-3, 0)
    with pytest.raises(pkg.FseqError) as e:
        ctx.run()
    assert e.value.code == pkg.FSEQ_E_UNSUPPORTED


def test_block_states_and_lists_match_oracle(pkg):
    m, n, L, B = 200, 1200, 15, 100
    msa = fso.synth_msa(fso.synth_spec(31, 6, 150, 3e-3), m, n)
    ctx = run_gpu(pkg, msa, L, block_len=B)
    p = fso.Pbwt(msa, debug=False)
    nblocks = (n + B - 1) // B
    for k in range(n + 1):
        if k % B == 0 or k == n:
            b = k // B if k < n else nblocks
            a, d = ctx.debug_block_state(b)
            assert np.array_equal(a, p.a), k
            assert np.array_equal(d, p.d), k
        if k < n:
            p.step()
            if k % 7 == 0 or k > n - 5:
                v, c = p.counts()
                gv, gc, cnt0, complete = ctx.debug_column_list(k)
                assert complete or gc[1:].sum() > ctx.timings()["list_cap_used"]
                # entry 0 lumps the values >= (k+1)+1-L; the rest are the distinct values below, descending
                thr = max(0, k + 2 - L)
                rec = v >= thr
                ev = np.concatenate([[k + 1], v[~rec][::-1]])
                ec = np.concatenate([[c[rec].sum()], c[~rec][::-1]])
                assert np.array_equal(gv, ev[:len(gv)]), k
                assert np.array_equal(gc, ec[:len(gc)]), k
                assert complete == (len(gv) == len(ev))
                if complete:                                     # (what the DP reads of a complete list; an open one leaves it 0)
                    assert cnt0 == (c[0] if v[0] == 0 else 0)


def test_small_list_cap_retries_to_exact_result(pkg):
    msa = fso.synth_msa(fso.synth_spec(77, 4, 60, 1e-2), 60, 400)
    ctx, ref = compare_long(pkg, msa, 10, block_len=40, list_cap=2)
    t = ctx.timings()
    assert t["retries"] >= 1 and t["list_cap_used"] > 2


def test_unreducible_input_reports_no_reduction(pkg):
    rng = np.random.default_rng(0)
    msa = (rng.integers(0, 4, size=(6, 200)) + 65).astype(np.uint8)
    ctx = pkg.SegmentationContext(6, 200, 20)
    ctx.set_sequences(msa)
    with pytest.raises(pkg.NoReduction):
        ctx.run()
    ref = fso.segment_long(msa, 20)
    assert ref["status"] == 1 and ctx.result.max_segment_size == ref["max_segment_size"]


SHORT_SHAPES = [
    # m, n, L, K, Brec, mu, seed, kind
    (50, 30, 20, 3, 1000, 1e-3, 3, 0),
    (300, 999, 500, 6, 200, 2e-3, 41, 0),
    (700, 1200, 900, 8, 150, 3e-3, 42, 1),               # sigma = 16
    (9000, 2000, 1500, 16, 400, 2e-4, 43, 0),            # 16-bit LDS state (m > 7168)
    (12000, 1500, 800, 12, 300, 3e-4, 44, 0),            # streamed rows
    (10000, 70000, 40000, 32, 5000, 1e-4, 45, 1),        # m > 7168 AND more than 65535 columns in the one block (16-bit divergences would wrap)
]


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed,kind", SHORT_SHAPES)
def test_short_path_matches_oracle(pkg, m, n, L, K, Brec, mu, seed, kind):
    """segmentation_sp_context::process (n < 2L): the distinct rows with their representatives and copy numbers."""
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.set_sequences(msa)
    try:
        res = ctx.run()
    except pkg.NoReduction:
        res = ctx.result
    assert res.short_path == 1
    f, r = fso.segment_short(msa)
    gf, gr = ctx.short_path_runs()
    assert res.max_segment_size == len(f)
    assert np.array_equal(gf, f) and np.array_equal(gr, r)


def test_short_path_by_column_sweep_refuses_what_would_wrap(pkg, monkeypatch):
    """The classic per-column sweep keeps block-relative divergences in 16 bits for 7168 < m <= 11264: one block of
    more than 65535 columns must fail loudly there instead of returning a wrong count (round-1 advice)."""
    monkeypatch.setenv("FSEQ_PHASE_A_CLASSIC", "1")
    ctx = pkg.SegmentationContext(8000, 66000, 40000)
    ctx.generate_synthetic(7, 8, 5000, 1e-4, 0)
    with pytest.raises(pkg.FseqError) as e:
        ctx.run()
    assert e.value.code == pkg.FSEQ_E_UNSUPPORTED
    # ... and still runs the shapes it can
    msa = fso.synth_msa(fso.synth_spec(41, 6, 200, 2e-3, 0), 300, 999)
    c2 = pkg.SegmentationContext(300, 999, 500)
    c2.set_sequences(msa)
    c2.run()
    f, r = fso.segment_short(msa)
    gf, gr = c2.short_path_runs()
    assert np.array_equal(gf, f) and np.array_equal(gr, r)


def test_device_generator_matches_host_generator(pkg):
    for name, m, n in (("C1", 8, 1000), ("C2", 100, 3000), ("C5", 64, 2000)):
        c = fso.CONFIGS[name]
        ctx = pkg.SegmentationContext(m, n, c["L"])
        ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
        host = fso.synth_msa(fso.config_spec(name), m, n)
        assert np.array_equal(ctx.get_sequences(), host)
        assert np.array_equal(ctx.get_sequences(10, 50), host[:, 10:50])


@pytest.mark.parametrize("m,n", [(1, 1), (7, 130), (65, 64), (100, 1000), (257, 333)])
def test_device_input_path_roundtrip(pkg, m, n):
    """Row-major input: alphabet scan + encode + transpose on the device; reading the alignment
    back must give the same bytes, for any byte alphabet and sizes off the 64 x 64 tile grid."""
    rng = np.random.default_rng(m * 1000 + n)
    msa = rng.choice(np.array([0, 1, 45, 65, 67, 71, 84, 200, 255], dtype=np.uint8), size=(m, n))
    ctx = pkg.SegmentationContext(m, n, max(1, n // 4))
    ctx.set_sequences(np.ascontiguousarray(msa))          # C order -> device path
    assert np.array_equal(ctx.get_sequences(), msa)
    ctx.set_sequences(np.asfortranarray(msa))             # F order -> host path
    assert np.array_equal(ctx.get_sequences(), msa)


def test_row_major_and_column_major_inputs_agree(pkg):
    msa = fso.synth_msa(fso.synth_spec(9, 4, 80, 5e-3), 33, 700)
    c1 = run_gpu(pkg, np.asfortranarray(msa), 12)
    c2 = run_gpu(pkg, np.ascontiguousarray(msa), 12)
    assert np.array_equal(c1.reduced_traceback(), c2.reduced_traceback())
    a1, d1 = c1.boundary_state(0)
    a2, d2 = c2.boundary_state(0)
    assert np.array_equal(a1, a2) and np.array_equal(d1, d2)


def test_config_c2_full_size_properties(pkg):
    """BASELINE config C2 at full size: size-independent properties + oracle parity of the result."""
    c = fso.CONFIGS["C2"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    res = ctx.run()
    tb = ctx.traceback()
    red = ctx.reduced_traceback()
    # segments tile [0, n), each at least L long, merged boundaries are DP boundaries
    assert tb["lb"][0] == 0 and tb["rb"][-1] == n
    assert np.array_equal(tb["lb"][1:], tb["rb"][:-1]) and (tb["rb"] - tb["lb"]).min() >= L
    assert red["lb"][0] == 0 and red["rb"][-1] == n and np.array_equal(red["lb"][1:], red["rb"][:-1])
    assert set(red["rb"].tolist()) <= set(tb["rb"].tolist())
    assert res.max_segment_size == tb["segment_size"].max() == tb["segment_max_size"][-1] < m
    assert red["segment_size"].max() <= res.max_segment_size
    # boundary states are permutations and their divergences reproduce the segment sizes
    for i in (0, len(red) // 2, len(red) - 1):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(np.sort(a), np.arange(m))
        assert int((d > red["lb"][i]).sum()) == red["segment_size"][i]
        assert d[0] == red["rb"][i]
    # idempotence
    res2 = ctx.run()
    assert np.array_equal(ctx.reduced_traceback(), red) and res2.max_segment_size == res.max_segment_size
    # the oracle on the same input (a few seconds of CPU)
    msa = fso.synth_msa(fso.config_spec("C2"), m, n)
    ref = fso.segment_long(msa, L, threads=8)
    assert ref["max_segment_size"] == res.max_segment_size
    for f in ("lb", "rb", "segment_size"):
        assert np.array_equal(red[f], ref["reduced"][f])
    for i in (0, len(red) // 3, len(red) - 1):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(a, ref["a"][i]) and np.array_equal(d, ref["d"][i])


def test_list_capacity_estimate_and_hint(pkg):
    """Segments larger than the default list capacity (255): the capacity comes from the block boundary
    states (no retry), a second run on the same context starts from it, and a tiny explicit capacity
    still reaches the same result through retries."""
    m, n, L = 3000, 1500, 30
    msa = fso.synth_msa(fso.synth_spec(88, 300, 100, 4e-3), m, n)      # 300 founders: segment sizes of several hundred
    ctx, ref = compare_long(pkg, msa, L, check_dp=True)
    assert ref["max_segment_size"] > 255
    t = ctx.timings()
    assert t["list_cap_used"] > 255 and t["retries"] == 0
    red = ctx.reduced_traceback().copy()
    ctx.run()
    t2 = ctx.timings()
    assert t2["retries"] == 0 and t2["list_cap_used"] == t["list_cap_used"]
    assert np.array_equal(ctx.reduced_traceback(), red)
    ctx3, _ = compare_long(pkg, msa, L, check_dp=False, list_cap=16)
    assert ctx3.timings()["retries"] >= 4


def test_concurrent_contexts_on_one_device(pkg):
    """Several alignments in flight on one GPU (one context, stream and host thread each, as bench.py's
    batched figure runs them): every context still gets exactly its own oracle result."""
    import threading
    cases = []
    for j, (m, n, L) in enumerate([(300, 4000, 25), (1000, 3000, 30), (2500, 6000, 50), (70, 2000, 9)]):
        msa = fso.synth_msa(fso.synth_spec(500 + j, 8 + j, 200, 2e-3), m, n)
        ctx = pkg.SegmentationContext(m, n, L)
        ctx.set_sequences(msa)
        cases.append((ctx, fso.segment_long(msa, L, threads=4)))
    errors = []

    def work(ctx):
        try:
            for _ in range(3):
                ctx.run()
        except Exception as e:                           # noqa: BLE001 - reported below
            errors.append(e)

    ths = [threading.Thread(target=work, args=(c,)) for c, _ in cases]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    for ctx, ref in cases:
        red = ctx.reduced_traceback()
        assert ctx.result.max_segment_size == ref["max_segment_size"]
        for f in ("lb", "rb", "segment_size"):
            assert np.array_equal(red[f], ref["reduced"][f]), f
        a, d = ctx.boundary_state(len(red) - 1)
        assert np.array_equal(a, ref["a"][len(red) - 1]) and np.array_equal(d, ref["d"][len(red) - 1])


def test_config_c3_full_size_matches_oracle(pkg):
    """BASELINE config C3 at full size (m = 2,504 x n = 1,000,000, L = 100: the pipelined DP schedule, 10^6
    DP cells, ~10,000 segments): merged segments and sampled boundary states against the CPU oracle
    (about 20 s of CPU), plus the size-independent properties."""
    c = fso.CONFIGS["C3"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    res = ctx.run()
    tb = ctx.traceback()
    red = ctx.reduced_traceback()
    assert tb["lb"][0] == 0 and tb["rb"][-1] == n
    assert np.array_equal(tb["lb"][1:], tb["rb"][:-1]) and (tb["rb"] - tb["lb"]).min() >= L
    assert red["lb"][0] == 0 and red["rb"][-1] == n and np.array_equal(red["lb"][1:], red["rb"][:-1])
    assert res.max_segment_size == tb["segment_size"].max() == tb["segment_max_size"][-1] < m
    msa = ctx.get_sequences()                                 # the device generator's alignment (checked equal to the host one elsewhere)
    ref = fso.segment_long(msa, L, threads=8)
    assert ref["max_segment_size"] == res.max_segment_size
    for f in ("lb", "rb", "segment_size"):
        assert np.array_equal(red[f], ref["reduced"][f])
    for i in (0, len(red) // 7, len(red) // 2, len(red) - 1):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(a, ref["a"][i]) and np.array_equal(d, ref["d"][i])


def test_config_c3_shape_at_high_diversity_matches_oracle(pkg):
    """BASELINE C3's shape at FULL length (m = 2,504 x n = 1,000,000, L = 100) with 1,024 founders and ten times the bench's
    mutation rate -- a point of profiles/r04_diversity_sweep.txt where the per-column lists grow to ~1,500 entries, most
    merges of the key-space tree would slice (its blocks go to the column sweep) and the maximum segment size is ~1,200
    instead of 68: the whole run against the CPU oracle (traceback, merged segments, sampled boundary states)."""
    c = fso.CONFIGS["C3"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], 1024, c["B"], 1e-3, c["kind"])
    res = ctx.run()
    t = ctx.timings()
    assert t["list_cap_used"] > 1000, t                  # (the key-space tree would give most blocks up; the trie ranks them: 1,024 founders are few classes)
    tb = ctx.traceback()
    red = ctx.reduced_traceback()
    assert tb["lb"][0] == 0 and tb["rb"][-1] == n and np.array_equal(tb["lb"][1:], tb["rb"][:-1]) and (tb["rb"] - tb["lb"]).min() >= L
    assert 500 < res.max_segment_size < m
    msa = ctx.get_sequences()
    ref = fso.segment_long(msa, L, threads=8)
    assert ref["max_segment_size"] == res.max_segment_size
    for f in ("lb", "rb", "segment_max_size", "segment_size"):
        assert np.array_equal(tb[f], ref["traceback"][f]), f
    for f in ("lb", "rb", "segment_size"):
        assert np.array_equal(red[f], ref["reduced"][f])
    for i in (0, len(red) // 7, len(red) // 2, len(red) - 1):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(a, ref["a"][i]) and np.array_equal(d, ref["d"][i])
    ctx.run()                                                 # (the second run launches what the first one learned: same result)
    assert np.array_equal(ctx.traceback(), tb)


@pytest.mark.parametrize("name", ["one_symbol", "identical_rows", "two_rows", "two_groups", "column_of_gaps"])
def test_degenerate_alignments(pkg, name):
    """Alphabets of one symbol, identical rows, two rows, two constant groups, a rare symbol in one column."""
    rng = np.random.default_rng(3)
    if name == "one_symbol":
        msa = np.full((20, 300), ord("A"), dtype=np.uint8)
    elif name == "identical_rows":
        msa = np.tile(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(1, 400)), (50, 1))
    elif name == "two_rows":
        msa = rng.choice(np.frombuffer(b"AC", dtype=np.uint8), size=(2, 500))
        msa[1] = msa[0]
        msa[1, 250] = ord("G")
    elif name == "two_groups":
        msa = np.empty((64, 256), dtype=np.uint8)
        msa[:32] = ord("A")
        msa[32:] = ord("T")
    else:
        msa = np.tile(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(1, 300)), (30, 1))
        msa[7, 150] = ord("-")
    compare_long(pkg, np.ascontiguousarray(msa), 10, block_len=37)


def _free_hbm_bytes():
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    return free


def check_full_size_properties(pkg, ctx, res, m, n, L, prefix_cols):
    """What holds at any size: the traceback tiles [0, n) with segments of length >= L, the merged segments are a
    coarsening of it, sizes are what the boundary states say (#{d > lb}), and -- a DP cell depends on nothing behind
    its own column -- the DP array on a column prefix equals the oracle's run on that prefix alone."""
    tb = ctx.traceback()
    red = ctx.reduced_traceback()
    assert tb["lb"][0] == 0 and tb["rb"][-1] == n
    assert np.array_equal(tb["lb"][1:], tb["rb"][:-1]) and (tb["rb"] - tb["lb"]).min() >= L
    assert red["lb"][0] == 0 and red["rb"][-1] == n and np.array_equal(red["lb"][1:], red["rb"][:-1])
    assert set(red["rb"].tolist()) <= set(tb["rb"].tolist())
    assert res.max_segment_size == tb["segment_size"].max() == tb["segment_max_size"][-1] < m
    assert red["segment_size"].max() <= res.max_segment_size
    for i in (0, len(red) // 3, len(red) - 1):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(np.sort(a), np.arange(m))
        assert int((d > red["lb"][i]).sum()) == red["segment_size"][i]
        assert d[0] == red["rb"][i]
    lb, mx, sz = ctx.debug_dp()
    ref = fso.segment_long(ctx.get_sequences(0, prefix_cols), L, keep_dp=True, threads=8)
    k = prefix_cols - 2 * L                                  # DP entries of regular cells of the prefix run
    assert np.array_equal(mx[:k], ref["dp"]["segment_max_size"][:k])
    assert np.array_equal(lb[:k], ref["dp"]["lb"][:k].astype(np.uint32))
    assert np.array_equal(sz[:k], ref["dp"]["segment_size"][:k])


def check_depth_against_oracle(pkg, ctx, m, n, L, blocks, list_every=1, cells_per_block=150, seed=1):
    """Oracle evidence at depth, at full size.  For every block b of `blocks`: the oracle's pBWT starts from the
    GPU's boundary state of b (debug_block_state) and runs the block's columns.  On the way
      * every merged boundary inside the block must equal fseq_boundary_state (pass 2 from the stride states),
      * the per-column list of every `list_every`-th column (and of the block's last 8) must be the top of the
        oracle's divergence-value counts,
      * `cells_per_block` sampled DP cells are evaluated again by the oracle's calculate_segmentation_lp_dp_arg on the
        oracle's counts of that column, over the GPU's own DP array with the oracle's rmq built on it: each must
        reproduce (lb, max, size) -- the fixed-point property of segmentation_lp_context.cc:393-481,
    and the state the oracle ends in must equal the GPU's boundary state of b + 1."""
    t = ctx.timings()
    B, nblocks, X = t["block_len"], t["n_blocks"], t["list_cap_used"]
    lb, mx, sz = ctx.debug_dp()
    k_dp = n - L + 1
    mx = mx.copy()
    mx[n - 2 * L + 1:n - L] = 0xFFFFFFFF                       # entries no cell writes (the oracle's initial value)
    dp = np.zeros(k_dp, dtype=fso.DP_DTYPE)
    dp["lb"], dp["rb"], dp["segment_max_size"], dp["segment_size"] = lb, np.arange(k_dp, dtype=np.uint64) + L, mx, sz
    rmq = fso.Rmq(mx, debug=False)
    for i in range(63, k_dp, 64):
        rmq.update(i)                                          # (append-only tables: a query only reads what was final when its cell ran)
    red = ctx.reduced_traceback()
    rb_index = {int(rb): i for i, rb in enumerate(red["rb"])}
    part2_limit, part3_limit = min(2 * L, n - L) - 1, n - L
    rng = np.random.default_rng(seed)
    checked = dict(blocks=0, boundaries=0, lists=0, cells=0)
    for b in sorted(set(int(x) for x in blocks)):
        assert 0 <= b < nblocks
        c0, c1 = b * B, min(n, (b + 1) * B)
        sub = ctx.get_sequences(c0, c1)
        p = fso.Pbwt(sub, debug=False, col0=c0)
        a, d = ctx.debug_block_state(b)
        p.set_state(a, d, c0)
        lo, hi = max(c0, part2_limit), min(c1, part3_limit)
        cell_cols = set(rng.choice(np.arange(lo, hi), size=min(cells_per_block, hi - lo), replace=False).tolist()) if hi > lo else set()
        for k in range(c0, c1):
            p.step()
            if k + 1 in rb_index:
                ga, gd = ctx.boundary_state(rb_index[k + 1])
                assert np.array_equal(ga, p.a) and np.array_equal(gd, p.d), ("boundary", b, k + 1)
                checked["boundaries"] += 1
            want_list = (k - c0) % list_every == 0 or k >= c1 - 8
            if want_list or k in cell_cols:
                v, c = p.counts()
            if want_list:
                gv, gc, cnt0, complete = ctx.debug_column_list(k)
                thr = max(0, k + 2 - L)
                rec = v >= thr
                ev = np.concatenate([[k + 1], v[~rec][::-1]])
                ec = np.concatenate([[c[rec].sum()], c[~rec][::-1]])
                assert complete or gc[1:].sum() > X, ("list", b, k)
                assert np.array_equal(gv, ev[:len(gv)]) and np.array_equal(gc, ec[:len(gc)]), ("list", b, k)
                assert complete == (len(gv) == len(ev)), ("list", b, k)
                # (the count of zeros is what the DP reads of a COMPLETE list; an open list of the reduced phase C leaves it 0)
                assert not complete or cnt0 == (c[0] if v[0] == 0 else 0), ("list", b, k)
                checked["lists"] += 1
            if k in cell_cols:
                got = fso.dp_step(v, c, dp, rmq.h, m, L, 0, k, (0, k + 1, m, m), debug=False)
                tt = k + 1 - L
                assert got == (int(lb[tt]), k + 1, int(mx[tt]), int(sz[tt])), ("cell", b, k, got)
                checked["cells"] += 1
        a, d = ctx.debug_block_state(b + 1)
        assert np.array_equal(a, p.a) and np.array_equal(d, p.d), ("block end", b)
        checked["blocks"] += 1
    return checked


def depth_blocks(t, n):
    """Blocks worth a replay: the first ones, both sides of every boundary between groups of 4^k blocks that is still
    inside the alignment for the largest k (phase B's levels), both sides of a boundary between two DP chunks in the
    middle of the alignment, a few arbitrary ones, the last two."""
    nb, B = t["n_blocks"], t["block_len"]
    pick = {0, 1, nb - 2, nb - 1, nb // 3, (2 * nb) // 3 + 1}
    g = 1
    while g * 4 < nb:
        g *= 4
    for q in (g // 4, g):
        for x in (q - 1, q):
            if 0 <= x < nb:
                pick.add(x)
    if t["dp_chunks"] > 1:
        col = n * (t["dp_chunks"] // 2) // t["dp_chunks"]      # about where chunk dp_chunks / 2 begins
        for x in (col // B - 1, col // B, col // B + 1):
            if 0 <= x < nb:
                pick.add(x)
    return sorted(pick)


def test_config_c5_full_size_properties(pkg):
    """BASELINE config C5 at full size (m = 10,000 x n = 1,000,000, sigma = 16, 4 bits per cell, 16-bit LDS state)."""
    c = fso.CONFIGS["C5"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    res = ctx.run()
    t = ctx.timings()
    assert t["dp_chunks"] > 100 and t["dp_sweeps"] < 1000
    check_full_size_properties(pkg, ctx, res, m, n, L, 6000)
    blocks = depth_blocks(t, n) + list(range(37, t["n_blocks"], 41))
    got = check_depth_against_oracle(pkg, ctx, m, n, L, blocks, list_every=1, cells_per_block=40)
    assert got["blocks"] >= 30 and got["cells"] >= 1000 and got["boundaries"] >= 32 and got["lists"] >= 20000, got


def test_config_c5_full_size_matches_oracle(pkg):
    """The whole of BASELINE config C5 through the oracle (pass 1 + DP on one thread, as the reference runs them: about
    a minute of CPU): DP array, traceback, merged segments and every boundary state, as for C3."""
    c = fso.CONFIGS["C5"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    ctx.run()
    msa = ctx.get_sequences()
    ref = fso.segment_long(msa, L, keep_dp=True, threads=8)
    lb, mx, sz = ctx.debug_dp()
    dp = ref["dp"]
    written = np.ones(len(dp), dtype=bool)
    written[n - 2 * L + 1:n - L] = False
    assert np.array_equal(mx[written], dp["segment_max_size"][written])
    assert np.array_equal(lb[written], dp["lb"][written].astype(np.uint32))
    assert np.array_equal(sz[written], dp["segment_size"][written])
    tb, red = ctx.traceback(), ctx.reduced_traceback()
    for f in ("lb", "rb", "segment_max_size", "segment_size"):
        assert np.array_equal(tb[f], ref["traceback"][f]), f
    for f in ("lb", "rb", "segment_size"):
        assert np.array_equal(red[f], ref["reduced"][f]), f
    for i in range(len(red)):
        a, d = ctx.boundary_state(i)
        assert np.array_equal(a, ref["a"][i]) and np.array_equal(d, ref["d"][i]), i


def test_config_c4_full_size_properties(pkg):
    """BASELINE config C4 at full size on one GPU (m = 100,000 x n = 5,000,000, 2 bits per cell, streamed block
    state): the size-independent properties, and oracle parity of the DP on a column prefix of the same alignment.
    Needs ~250 GB of free HBM (125 GB alignment + lists + stride states): skipped on a card that has less free."""
    if _free_hbm_bytes() < 200e9:
        pytest.skip("less than 200 GB of HBM free")
    c = fso.CONFIGS["C4"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    res = ctx.run()
    check_full_size_properties(pkg, ctx, res, m, n, L, 2400)
    t = ctx.timings()
    got = check_depth_against_oracle(pkg, ctx, m, n, L, depth_blocks(t, n), list_every=4, cells_per_block=100)       # (blocks of ~800 columns since round 5)
    assert got["blocks"] >= 8 and got["cells"] >= 1000 and got["boundaries"] >= 32 and got["lists"] >= 2000, got
    ctx.close()


@pytest.mark.parametrize("bits", [8, 4, 2])
def test_borrowed_device_columns(pkg, bits):
    """Input already in HBM (fseq_set_device_columns / _packed): column-major dense codes in a caller-owned
    buffer, one per byte or packed 4 / 2 bits per code the way the library stores its own uploads."""
    import torch
    m, n, L = 333, 900, 12
    sigma = {8: 7, 4: 13, 2: 4}[bits]
    rng = np.random.default_rng(bits)
    founders = rng.integers(0, sigma, size=(6, n))
    pick = rng.integers(0, 6, size=(m, (n + 89) // 90))
    codes = np.empty((m, n), dtype=np.uint8)
    for b in range(pick.shape[1]):
        codes[:, b * 90:(b + 1) * 90] = founders[pick[:, b], b * 90:(b + 1) * 90]
    noise = rng.random((m, n)) < 3e-3
    codes[noise] = rng.integers(0, sigma, size=int(noise.sum()))
    spb = 8 // bits
    ld = ((m + spb - 1) // spb + 15) // 16 * 16
    cols = np.zeros((n, ld), dtype=np.uint8)
    for r in range(m):
        cols[:, r // spb] |= (codes[r] << ((r % spb) * bits)).astype(np.uint8)
    dev = torch.from_numpy(cols).to("cuda")
    ctx = pkg.SegmentationContext(m, n, L)
    if bits == 8:
        ctx.set_device_columns(dev.data_ptr(), ld, sigma, keepalive=dev)
    else:
        ctx.set_device_columns_packed(dev.data_ptr(), ld, sigma, bits, keepalive=dev)
    assert np.array_equal(ctx.get_sequences(), codes)         # codes come back as they are (identity byte map)
    ctx.run()
    ref = fso.segment_long(codes, L, threads=4)
    red = ctx.reduced_traceback()
    assert ctx.result.max_segment_size == ref["max_segment_size"]
    for f in ("lb", "rb", "segment_size"):
        assert np.array_equal(red[f], ref["reduced"][f])
    a, d = ctx.boundary_state(len(red) - 1)
    assert np.array_equal(a, ref["a"][len(red) - 1]) and np.array_equal(d, ref["d"][len(red) - 1])


def test_borrowed_columns_with_a_code_beyond_sigma_are_refused(pkg):
    """fseq_set_device_columns trusts nothing: a borrowed code >= sigma would lose its high digits in the 2-bit digit
    passes and merge with another symbol (round-1 advice) -- one pass over the columns finds it, FSEQ_E_ARG."""
    import torch
    m, n = 100, 64
    cols = np.zeros((n, 112), dtype=np.uint8)
    cols[:, :m] = np.random.default_rng(1).integers(0, 5, size=(n, m))
    cols[17, 33] = 5                                          # sigma = 5: codes 0..4 only
    dev = torch.from_numpy(cols).to("cuda")
    ctx = pkg.SegmentationContext(m, n, 8)
    with pytest.raises(pkg.FseqError) as e:
        ctx.set_device_columns(dev.data_ptr(), 112, 5, keepalive=dev)
    assert e.value.code == pkg.FSEQ_E_ARG and "code 5" in str(e.value)
    cols[17, 33] = 4
    dev2 = torch.from_numpy(cols).to("cuda")
    ctx.set_device_columns(dev2.data_ptr(), 112, 5, keepalive=dev2)
    # padding behind the last row of a packed column is not a code: m = 5 rows at 4 bits, garbage in the spare nibble
    pk = np.zeros((16, 16), dtype=np.uint8)
    pk[:, 2] = 0xF2                                           # row 4 = code 2, spare high nibble = 15
    devp = torch.from_numpy(pk).to("cuda")
    c2 = pkg.SegmentationContext(5, 16, 4)
    c2.set_device_columns_packed(devp.data_ptr(), 16, 3, 4, keepalive=devp)


def _random_case(rng):
    L = int(rng.integers(1, 41))
    n = int(rng.integers(2 * L, 2 * L + 1200))
    m = int(rng.choice([2, 3, 5, 17, 64, 65, 130, 300, 449, 700]))
    sigma = int(rng.choice([2, 3, 4, 5, 16, 17, 60]))
    k = int(rng.integers(1, 9))
    brec = int(rng.integers(5, 200))
    founders = rng.integers(0, sigma, size=(k, n))
    pick = rng.integers(0, k, size=(m, (n + brec - 1) // brec))
    msa = np.empty((m, n), dtype=np.uint8)
    for b in range(pick.shape[1]):
        msa[:, b * brec:(b + 1) * brec] = founders[pick[:, b], b * brec:(b + 1) * brec]
    noise = rng.random((m, n)) < float(rng.choice([0.0, 1e-3, 1e-2]))
    msa[noise] = rng.integers(0, sigma, size=int(noise.sum()))
    kw = {"block_len": int(rng.choice([0, 0, 1, 7, 16, 33, 100, 5000]))}
    if rng.random() < 0.3:
        kw["list_cap"] = int(rng.choice([1, 2, 5, 17]))
    return np.ascontiguousarray(msa + 33), L, kw


@pytest.mark.parametrize("chunk", range(4))
def test_random_shapes_match_oracle(pkg, chunk):
    """Differential test on random shapes: rows, columns, L, alphabet, recombination, noise, block length
    and list capacity drawn at random (fixed seeds); DP array, traceback, merged segments and every
    boundary state must equal the oracle's."""
    import os
    rng = np.random.default_rng(1000 + chunk)
    for _ in range(int(os.environ.get("FSEQ_RANDOM_CASES", "10"))):
        msa, L, kw = _random_case(rng)
        compare_long(pkg, msa, L, check_dp=True, **kw)


def test_random_shapes_with_many_rows(pkg):
    """The same differential test on the row counts of the larger configurations (list wave on 512 / 1024 threads,
    16-bit state with 9 / 10 / 11 rows per thread, streamed tiles, halfword and 32-bit ids of the streamed phase A)."""
    import os
    rng = np.random.default_rng(2024)
    for _ in range(3 * int(os.environ.get("FSEQ_RANDOM_CASES", "10"))):
        L = int(rng.integers(2, 60))
        n = int(rng.integers(2 * L, 2 * L + 500))
        m = int(rng.choice([1300, 2240, 2241, 2504, 2560, 3000, 4000, 4800, 4801, 5008, 6721, 8000, 8640, 9300, 9601, 10000, 10300,
                            10800, 11264, 11265, 20000, 66000]))
        sigma = int(rng.choice([2, 4, 4, 5, 16, 20]))
        k = int(rng.integers(2, 40))
        brec = int(rng.integers(5, 300))
        founders = rng.integers(0, sigma, size=(k, n))
        pick = rng.integers(0, k, size=(m, (n + brec - 1) // brec))
        msa = np.empty((m, n), dtype=np.uint8)
        for b in range(pick.shape[1]):
            msa[:, b * brec:(b + 1) * brec] = founders[pick[:, b], b * brec:(b + 1) * brec]
        noise = rng.random((m, n)) < float(rng.choice([0.0, 1e-4, 1e-3]))
        msa[noise] = rng.integers(0, sigma, size=int(noise.sum()))
        kw = {"block_len": int(rng.choice([0, 0, 16, 33, 100, 5000]))}
        if rng.random() < 0.2:
            kw["list_cap"] = int(rng.choice([2, 17]))
        compare_long(pkg, np.ascontiguousarray(msa + 33), L, check_dp=True, **kw)


def test_native_batch_runner(pkg):
    """fseq_run_segmentation_batch: several contexts in flight from native threads; per-context results and
    return codes (one of the inputs cannot be reduced)."""
    cases = []
    for j, (m, n, L) in enumerate([(300, 3000, 25), (900, 2000, 30), (64, 1500, 9)]):
        msa = fso.synth_msa(fso.synth_spec(700 + j, 6 + j, 150, 2e-3), m, n)
        ctx = pkg.SegmentationContext(m, n, L)
        ctx.set_sequences(msa)
        cases.append((ctx, fso.segment_long(msa, L, threads=4)))
    bad = np.ascontiguousarray((np.random.default_rng(5).integers(0, 4, size=(6, 300)) + 65).astype(np.uint8))
    cbad = pkg.SegmentationContext(6, 300, 20)
    cbad.set_sequences(bad)
    rcs = pkg.run_batch([c for c, _ in cases] + [cbad])
    assert rcs[:3] == [pkg.FSEQ_OK] * 3 and rcs[3] == pkg.FSEQ_E_NO_REDUCTION
    for ctx, ref in cases:
        red = ctx.reduced_traceback()
        assert ctx.result.max_segment_size == ref["max_segment_size"]
        for f in ("lb", "rb", "segment_size"):
            assert np.array_equal(red[f], ref["reduced"][f])


SPEC_SHAPES = [(300, 6000, 25, 8, 200, 2e-3, 51, 0, 50), (900, 5000, 100, 10, 300, 1e-3, 52, 1, 64),
               (12000, 1500, 20, 12, 120, 3e-4, 53, 0, 30), (40, 9000, 7, 4, 60, 1e-2, 54, 0, 0),
               (200, 4000, 250, 6, 400, 2e-3, 55, 0, 100), (120, 3000, 140, 5, 300, 3e-3, 56, 0, 64),
               (2500, 20000, 50, 16, 2000, 1e-4, 0x5EED0002, 0, 0)]


@pytest.mark.parametrize("rounds,max_sweeps,win", [(0, 0, 0), (1, 0, 0), (3, 0, 0), (7, 0, 0), (40, 0, 0), (5, 2, 0), (2, 1, 0), (4, 0, 1), (4, 3, 5000)])
def test_speculative_dp_matches_serial_walk(pkg, monkeypatch, rounds, max_sweeps, win):
    """Phase D as chunk-speculative sweeps (fseq_dpspec.hpp): whatever the chunk length (rounds = 1: every round
    its own chunk), the tail window (1: the lifts pick up every spike; 5000: whole chunks) and the sweep budget
    (1, 2, 3: the serial kernel takes over behind the first dirty chunk), the whole DP array, the traceback, the
    merged segments and the boundary states equal the oracle's serial walk.  rounds = 0: the plan the library picks."""
    monkeypatch.setenv("FSEQ_POISON_LISTS", "1")
    if rounds:
        monkeypatch.setenv("FSEQ_DP_SPEC_ROUNDS", str(rounds))
    if max_sweeps:
        monkeypatch.setenv("FSEQ_DP_SPEC_MAX_SWEEPS", str(max_sweeps))
    if win:
        monkeypatch.setenv("FSEQ_DP_SPEC_WIN", str(win))
    for (m, n, L, K, Brec, mu, seed, kind, B) in SPEC_SHAPES:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctx, _ = compare_long(pkg, msa, L, block_len=B)
        t = ctx.timings()
        if rounds:
            assert t["dp_chunks"] >= 2, t
        if max_sweeps and rounds:
            assert t["dp_sweeps"] <= max_sweeps + 1 or t["dp_sweeps"] >= 1000, t


@pytest.mark.parametrize("mode", ["tree", "tiny_bitmap", "classic"])
def test_phase_a_in_key_space_matches_the_column_sweep(pkg, monkeypatch, mode):
    """Phase A ranks the block keys as a tree over the columns (fseq_blockkeys.hpp) instead of sweeping the pBWT over
    the block; merges too large for its LDS bitmap run in slices (forced here with a bitmap of 2048 words, and on
    random rows where every row is its own key), FSEQ_PHASE_A_CLASSIC runs the sweep everywhere.  Block boundary
    states, lists, DP, traceback, segments and boundary states are compared with the oracle in every mode."""
    if mode == "tiny_bitmap":
        monkeypatch.setenv("FSEQ_BLOCKKEYS_CAP", "2048")
    if mode == "classic":
        monkeypatch.setenv("FSEQ_PHASE_A_CLASSIC", "1")
    fallbacks = 0
    for (m, n, L, K, Brec, mu, seed, kind, B) in SPEC_SHAPES + [(64, 4000, 6, 5, 90, 5e-3, 29, 0, 4), (8, 1000, 10, 3, 100, 5e-3, 0x5EED0001, 0, 0),
                                                                   (700, 1500, 12, 300, 40, 3e-2, 77, 1, 333)]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctx, _ = compare_long(pkg, msa, L, block_len=B)
        fallbacks += ctx.timings()["phase_a_fallbacks"]
        bl = ctx.timings()["block_len"]
        p = fso.Pbwt(msa)
        for b in range(0, ctx.timings()["n_blocks"] + 1, max(1, ctx.timings()["n_blocks"] // 5)):
            while p.idx < min(n, b * bl):
                p.step()
            a, d = ctx.debug_block_state(b)
            assert np.array_equal(a, p.a) and np.array_equal(d, p.d), (m, n, b)
    # random rows: every row its own key in every block -- far beyond a 2048-word bitmap
    rng = np.random.default_rng(3)
    msa = (rng.integers(0, 4, size=(900, 600)) + 65).astype(np.uint8)
    ctx = run_gpu(pkg, msa, 8, block_len=150)
    ref = fso.segment_long(msa, 8, keep_dp=True, threads=4)
    lb, mx, sz = ctx.debug_dp()
    assert np.array_equal(mx[:600 - 16 + 1], ref["dp"]["segment_max_size"][:600 - 16 + 1])
    fallbacks += ctx.timings()["phase_a_fallbacks"]
    if mode == "tiny_bitmap":
        assert fallbacks > 0
    if mode == "classic":
        assert fallbacks == 0


@pytest.mark.parametrize("shape", [
    # m, n, L, K, mu, kind, block_len: every row its own founder -- every merge of the key-space tree would slice many times over
    (2504, 2000, 20, 2504, 1e-2, 0, 245),                 # LDS-resident rows (BASELINE C3's), 32-bit words
    (10000, 1400, 30, 10000, 1e-3, 1, 200),               # sigma = 16, 16-bit LDS state (BASELINE C5's rows)
    (30000, 400, 10, 30000, 1e-3, 0, 80),                 # streamed rows
])
def test_diverse_blocks_go_to_the_column_sweep(pkg, monkeypatch, shape):
    """A block whose merges would slice past its budget is given up by the key-space tree and ranked by the column sweep
    (profiles/r04_diversity_sweep.txt: with all rows distinct the slices made phase A 25 - 150 times slower).  First run:
    tree + filtered sweep; the second run knows that most blocks were given up and runs the sweep alone; a friendly input
    on the same shape gives nothing up and its second run launches the tree alone.  All bit-identical to the oracle."""
    m, n, L, K, mu, kind, B = shape
    monkeypatch.setenv("FSEQ_NO_BLOCKTRIE", "1")             # (the trie in front of the tree: test_phase_a_trie_on_lds_resident_rows)
    msa = fso.synth_msa(fso.synth_spec(91, K, 500, mu, kind), m, n)
    ctx, ref = compare_long(pkg, msa, L, check_dp=False, block_len=B)
    t = ctx.timings()
    assert t["phase_a_given_up"] > t["n_blocks"] // 2, t
    tb0 = ctx.traceback().copy()
    try:
        ctx.run()                                                    # the sweep alone
    except pkg.NoReduction:
        pass
    assert ctx.timings()["phase_a_given_up"] == t["n_blocks"]
    assert np.array_equal(ctx.traceback(), tb0)
    bl = t["block_len"]
    p = fso.Pbwt(msa)
    for b in (1, t["n_blocks"] // 2, t["n_blocks"]):
        while p.idx < min(n, b * bl):
            p.step()
        a, d = ctx.debug_block_state(b)
        assert np.array_equal(a, p.a) and np.array_equal(d, p.d), b
    # a friendly mosaic on the same shape: nothing given up, and the second run (tree alone) agrees with the first
    msa2 = fso.synth_msa(fso.synth_spec(92, 12, 300, 2e-4, kind), m, n)
    ctx2, _ = compare_long(pkg, msa2, L, check_dp=False, block_len=B)
    assert ctx2.timings()["phase_a_given_up"] == 0
    tb2 = ctx2.traceback().copy()
    ctx2.run()
    assert ctx2.timings()["phase_a_given_up"] == 0 and np.array_equal(ctx2.traceback(), tb2)
    # mixed: the first half of the columns friendly, the second half all distinct -> some blocks given up, not most... or most:
    # either way the run after it repeats the outcome exactly
    msa3 = msa2.copy()
    msa3[:, n // 2 + n // 8:] = msa[:, n // 2 + n // 8:]
    ctx3, _ = compare_long(pkg, msa3, L, check_dp=False, block_len=B)
    g3 = ctx3.timings()["phase_a_given_up"]
    assert 0 < g3 < t["n_blocks"]
    tb3 = ctx3.traceback().copy()
    try:
        ctx3.run()
    except pkg.NoReduction:
        pass
    assert np.array_equal(ctx3.traceback(), tb3)


def test_repeated_runs_are_bit_identical(pkg):
    """The same context run again and again (phase A's staged leaf columns, the emitter wave, the speculative DP and
    the benign races they allow must never show in a result): traceback, segments and a boundary state each time."""
    for (m, n, L, K, Brec, mu, seed, kind) in [(10000, 1200, 100, 32, 5000, 1e-4, 0x5EED0005, 1), (2500, 20000, 50, 16, 2000, 1e-4, 0x5EED0002, 0)]:
        ctx = pkg.SegmentationContext(m, n, L)
        ctx.generate_synthetic(seed, K, Brec, mu, kind)
        ctx.run()
        tb0, red0 = ctx.traceback().copy(), ctx.reduced_traceback().copy()
        a0, d0 = ctx.boundary_state(len(red0) // 2)
        b0 = ctx.debug_block_state(ctx.timings()["n_blocks"] // 2)
        for _ in range(12):
            ctx.run()
            assert np.array_equal(ctx.traceback(), tb0) and np.array_equal(ctx.reduced_traceback(), red0)
            a, d = ctx.boundary_state(len(red0) // 2)
            assert np.array_equal(a, a0) and np.array_equal(d, d0)
            b = ctx.debug_block_state(ctx.timings()["n_blocks"] // 2)
            assert np.array_equal(b[0], b0[0]) and np.array_equal(b[1], b0[1])


def test_speculative_dp_with_short_lists_retries(pkg, monkeypatch):
    """A list too short to prove a cell is reported per chunk by the sweep that last ran the chunk; the run
    retries with longer lists and stays exact."""
    monkeypatch.setenv("FSEQ_DP_SPEC_ROUNDS", "3")
    msa = fso.synth_msa(fso.synth_spec(61, 8, 200, 2e-3, 0), 300, 6000)
    ctx, _ = compare_long(pkg, msa, 25, block_len=50, list_cap=2)
    assert ctx.timings()["retries"] >= 1 and ctx.timings()["dp_chunks"] >= 2


def test_progress_counters_and_callback(pkg):
    """fseq_set_progress / fseq_step_max / fseq_current_step (segmentation_lp_context.hh:122-127): the stages come in the
    reference's order -- generate_traceback (steps = columns), find_segments_greedy (traceback entries),
    update_samples_to_traceback_positions (merged boundaries) -- every stage ends at its step_max, and the counters read
    what the last callback reported."""
    m, n, L = 300, 6000, 25
    msa = fso.synth_msa(fso.synth_spec(51, 8, 200, 2e-3, 0), m, n)
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.set_sequences(msa)
    seen = []
    ctx.set_progress(lambda stage, cur, mx: seen.append((stage, cur, mx)))
    ctx.run()
    stages = [s for s, _, _ in seen]
    assert stages == sorted(stages) and set(stages) == {pkg.STAGE_TRACEBACK, pkg.STAGE_MERGE, pkg.STAGE_SAMPLES}
    assert all(cur <= mx for _, cur, mx in seen)
    last = {s: (cur, mx) for s, cur, mx in seen}
    assert last[pkg.STAGE_TRACEBACK] == (n, n)
    assert last[pkg.STAGE_MERGE] == (ctx.result.dp_segment_count, ctx.result.dp_segment_count)
    assert last[pkg.STAGE_SAMPLES] == (ctx.result.segment_count, ctx.result.segment_count)
    assert (ctx.current_step(), ctx.step_max()) == (seen[-1][1], seen[-1][2])
    ctx.set_progress(None)
    ctx.run()
    assert len(seen) == len(stages)


def test_phase_ranges_are_pushed_and_balanced(pkg):
    """Every phase of a long-path run is an roctx range (rocprofv3 --marker-trace): four pushes and four pops per run, by
    the library's own count; the default build links roctx."""
    msa = fso.synth_msa(fso.synth_spec(22, 8, 200, 2e-3, 0), 300, 2000)
    ctx = pkg.SegmentationContext(300, 2000, 25)
    ctx.set_sequences(msa)
    p0, q0, with_roctx = pkg.debug_ranges()
    ctx.run()
    p1, q1, _ = pkg.debug_ranges()
    assert p1 - p0 == 4 and q1 - q0 == 4
    ctx.run()
    p2, q2, _ = pkg.debug_ranges()
    assert p2 - p1 == 4 and q2 - q1 == 4 and p2 == q2
    import os
    if os.path.exists("/opt/rocm/lib/librocprofiler-sdk-roctx.so") and os.path.exists("/opt/rocm/include/rocprofiler-sdk-roctx/roctx.h"):
        assert with_roctx


def test_tuning_is_per_context_and_read_once(pkg, monkeypatch):
    """The FSEQ_* knobs are read from the environment when a context is created (never on the run path) and can be
    set per context: a context created under FSEQ_PHASE_A_CLASSIC keeps the column sweep after the variable is gone,
    another one gets it through set_tuning; unknown names are refused."""
    m, n, L = 12000, 600, 12
    msa = fso.synth_msa(fso.synth_spec(53, 12, 120, 3e-4, 0), m, n)
    ref = fso.segment_long(msa, L, threads=4)
    monkeypatch.setenv("FSEQ_BLOCKKEYS_CAP", "2048")              # (a bitmap so small that the key-space tree has to slice: phase_a_fallbacks > 0)
    a = pkg.SegmentationContext(m, n, L, block_len=64)
    monkeypatch.delenv("FSEQ_BLOCKKEYS_CAP")
    b = pkg.SegmentationContext(m, n, L, block_len=64)
    c = pkg.SegmentationContext(m, n, L, block_len=64)
    c.set_tuning("FSEQ_PHASE_A_CLASSIC")
    with pytest.raises(pkg.FseqError):
        c.set_tuning("FSEQ_NO_SUCH_KNOB")
    for ctx in (a, b, c):
        ctx.set_sequences(msa)
        ctx.run()
        assert np.array_equal(ctx.reduced_traceback()["rb"], ref["reduced"]["rb"])
    ta, tb, tc = a.timings(), b.timings(), c.timings()
    assert tb["phase_a_fallbacks"] == 0 and tc["phase_a_fallbacks"] == 0
