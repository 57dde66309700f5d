"""One alignment over several ranks (fseq_set_shard): every rank holds its own column blocks, the exchange steps go
through the caller's all-reduce.  Here the ranks are threads of this process on the one GPU (ThreadWorld); the
results on EVERY rank -- DP array, traceback, merged segments -- and the boundary states on their owners must be
bit-identical to the oracle's serial walk, for rank counts that do and do not divide the blocks evenly."""
import importlib
import threading

import numpy as np
import pytest

import fso
from helpers import owned_dp_mask

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("founder-sequences_amd")


def run_world(pkg, world, make_input, m, n, L, **kw):
    fdist = importlib.import_module("founder-sequences_amd.dist")
    tw = fdist.ThreadWorld(world)
    ctxs = [pkg.SegmentationContext(m, n, L, **kw) for _ in range(world)]
    errs = [None] * world

    def work(r):
        try:
            tw.attach(ctxs[r], r, "cuda:0")
            make_input(ctxs[r])
            try:
                ctxs[r].run()
            except pkg.NoReduction:
                pass
        except BaseException as e:          # a failing rank must not leave the others in a barrier for ever
            errs[r] = e
            tw.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errs:
        if e is not None:
            raise e
    return ctxs


def check_against_oracle(pkg, ctxs, msa, L):
    m, n = msa.shape
    ref = fso.segment_long(msa, L, keep_dp=True, threads=4)
    covered = np.zeros(n - L + 1, dtype=bool)
    for ctx in ctxs:
        assert ctx.result.max_segment_size == ref["max_segment_size"]
        lb, mx, sz = ctx.debug_dp()
        written = owned_dp_mask(ctx, n, L)                # (a rank that keeps windows answers for the entries it computed)
        covered |= written
        assert np.array_equal(mx[written], ref["dp"]["segment_max_size"][written])
        assert np.array_equal(lb[written], ref["dp"]["lb"][written].astype(np.uint32))
        assert np.array_equal(sz[written], ref["dp"]["segment_size"][written])
        tb = ctx.traceback()
        for f in ("lb", "rb", "segment_max_size", "segment_size"):
            assert np.array_equal(tb[f], ref["traceback"][f]), f
        if ref["status"] != 0:
            assert ctx.result.segment_count == 0
            continue
        red = ctx.reduced_traceback()
        assert len(red) == len(ref["reduced"])
        for f in ("lb", "rb", "segment_size"):
            assert np.array_equal(red[f], ref["reduced"][f]), f
    all_written = np.ones(n - L + 1, dtype=bool)
    all_written[n - 2 * L + 1:n - L] = False
    assert np.array_equal(covered, all_written)           # the ranks' parts tile the DP array
    if ref["status"] != 0:
        return
    red = ctxs[0].reduced_traceback()
    owners = set()
    for i in range(len(red)):
        owner = ctxs[0].shard_owner(int(red["rb"][i]))
        owners.add(owner)
        a, d = ctxs[owner].boundary_state(i)
        assert np.array_equal(a, ref["a"][i]), i
        assert np.array_equal(d, ref["d"][i]), i
        other = (owner + 1) % len(ctxs)
        if other != owner:
            with pytest.raises(pkg.FseqError):
                ctxs[other].boundary_state(i)
    return owners


SHAPES = [
    # m, n, L, K, Brec, mu, seed, kind, block_len
    (300, 6000, 25, 8, 200, 2e-3, 51, 0, 50),
    (900, 5000, 100, 10, 300, 1e-3, 52, 1, 64),          # sigma = 16, pipelined DP schedule
    (40, 9000, 7, 4, 60, 1e-2, 54, 0, 0),
    (2500, 20000, 50, 16, 2000, 1e-4, 0x5EED0002, 0, 0),  # BASELINE C2 rows
    (12000, 3000, 20, 12, 120, 3e-4, 53, 0, 30),          # streamed block state
    (64, 4000, 6, 5, 90, 5e-3, 29, 0, 4),                 # 1000 blocks of 4 columns
]


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("shape", SHAPES)
def test_sharded_run_matches_oracle(pkg, world, shape):
    m, n, L, K, Brec, mu, seed, kind, B = shape
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    ctxs = run_world(pkg, world, lambda c: c.set_sequences(msa), m, n, L, block_len=B)
    owners = check_against_oracle(pkg, ctxs, msa, L)
    assert len(owners) == world                      # every rank produced boundary states
    cols = [c.shard_columns() for c in ctxs]
    assert cols[0][0] == 0 and cols[-1][1] == n
    for (a0, a1), (b0, b1) in zip(cols, cols[1:]):
        assert a0 < b0 <= a1 < b1 + 1                 # contiguous shares, the halo reaches into the next rank's
    calls = {c._transport.calls for c in ctxs}
    assert len(calls) == 1                            # every rank made the same exchanges


@pytest.mark.parametrize("fan", [2, 3, 8])
def test_sharded_phase_b_recursion(pkg, monkeypatch, fan):
    """A rank composes its hyper key block level by level (groups of F key blocks; F = 4 by itself): 1000 and 600 blocks
    over 2 and 3 ranks with F = 2 (seven or eight levels per rank), 3, 8 -- block states included."""
    monkeypatch.setenv("FSEQ_CHAIN_FAN", str(fan))
    for world, (m, n, L, K, Brec, mu, seed, kind, B) in [(2, (64, 4000, 6, 5, 90, 5e-3, 29, 0, 4)), (3, (300, 6000, 25, 8, 200, 2e-3, 51, 0, 10)),
                                                          (2, (12000, 900, 10, 12, 200, 3e-4, 46, 0, 12))]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctxs = run_world(pkg, world, lambda c: c.set_sequences(msa), m, n, L, block_len=B)
        check_against_oracle(pkg, ctxs, msa, L)


def test_sharded_device_generator_and_short_lists(pkg, monkeypatch):
    """Device-side generation of the rank's own columns; a list capacity too small for the DP is found out by all
    ranks together and retried."""
    monkeypatch.setenv("FSEQ_DP_SPEC_ROUNDS", "5")
    m, n, L = 300, 8000, 25
    spec = (61, 8, 200, 2e-3, 0)
    msa = fso.synth_msa(fso.synth_spec(*spec), m, n)
    ctxs = run_world(pkg, 3, lambda c: c.generate_synthetic(*spec), m, n, L, block_len=40, list_cap=2)
    check_against_oracle(pkg, ctxs, msa, L)
    assert all(c.timings()["retries"] >= 1 for c in ctxs)
    # a rank holds (and can hand back) its own columns only
    c0, c1 = ctxs[1].shard_columns()
    assert np.array_equal(ctxs[1].get_sequences(c0, c1), msa[:, c0:c1])
    with pytest.raises(pkg.FseqError):
        ctxs[1].get_sequences(0, 10)


@pytest.mark.parametrize("mode", ["window_1024", "window_64", "full"])
def test_sharded_dp_exchange_modes(pkg, monkeypatch, mode):
    """The sharded DP keeps, of the other ranks' keys, a window in front of its own entries and follows the traceback rank by
    rank.  A window that holds everything the sweeps read (1,024 entries here), one that does not (64: a sweep reports that
    it read below it and the run starts again with whole-array exchanges), and the whole-array form asked for outright --
    all bit-identical to the oracle, DP entries included."""
    if mode == "full":
        monkeypatch.setenv("FSEQ_SHARD_DP_FULL", "1")
    else:
        monkeypatch.setenv("FSEQ_SHARD_DP_WINDOW", mode.split("_")[1])
    monkeypatch.setenv("FSEQ_DP_SPEC_ROUNDS", "7")
    fell_back = []
    for world, (m, n, L, K, Brec, mu, seed, kind, B) in [(3, (300, 24000, 25, 8, 200, 2e-3, 51, 0, 50)), (4, (2500, 20000, 50, 16, 2000, 1e-4, 0x5EED0002, 0, 0)),
                                                          (2, (12000, 6000, 20, 12, 120, 3e-4, 53, 0, 30))]:
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        ctxs = run_world(pkg, world, lambda c: c.set_sequences(msa), m, n, L, block_len=B)
        check_against_oracle(pkg, ctxs, msa, L)
        whole = {c.debug_dp_owned()[3] for c in ctxs}
        assert len(whole) == 1                            # every rank took the same way
        if mode == "full":
            assert whole == {True}
        if mode == "window_1024":
            assert whole == {False}
        fell_back.append(whole == {True})
        assert len({c._transport.calls for c in ctxs}) == 1
    if mode == "window_64":
        #  the window that does not hold: on at least one of the inputs a sweep read below it and the run started
        # again with whole-array exchanges -- the restart is what makes the window exact, so it must have been taken
        assert any(fell_back), fell_back


def test_too_many_ranks_fail_together(pkg):
    m, n, L = 16, 250, 100                       # two DP rounds in all: three ranks with columns cannot all own one
    fdist = importlib.import_module("founder-sequences_amd.dist")
    tw = fdist.ThreadWorld(4)
    ctx = pkg.SegmentationContext(m, n, L)
    with pytest.raises(pkg.FseqError):
        tw.attach(ctx, 3, "cuda:0")


def test_a_failing_rank_fails_every_rank(pkg, monkeypatch):
    """A rank that fails on its own (injected after phase A, where an out-of-memory would strike) posts its error code
    in the status word every exchange starts with: it reports its own error, the others FSEQ_E_PEER -- all from matching
    exchanges, so nobody is left waiting (the harness would hang here otherwise: no barrier is aborted for this)."""
    monkeypatch.setenv("FSEQ_INJECT_FAILURE_RANK", "1")
    fdist = importlib.import_module("founder-sequences_amd.dist")
    m, n, L = 300, 6000, 25
    msa = fso.synth_msa(fso.synth_spec(51, 8, 200, 2e-3, 0), m, n)
    world = 3
    tw = fdist.ThreadWorld(world)
    ctxs = [pkg.SegmentationContext(m, n, L, block_len=50) for _ in range(world)]
    codes = [None] * world

    def work(r):
        tw.attach(ctxs[r], r, "cuda:0")
        ctxs[r].set_sequences(msa)
        try:
            ctxs[r].run()
            codes[r] = 0
        except pkg.FseqError as e:
            codes[r] = e.code

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths)
    assert codes == [pkg.FSEQ_E_PEER, 4, pkg.FSEQ_E_PEER]
    assert len({c._transport.calls for c in ctxs}) == 1           # the same exchanges on every rank, the failing one included


def test_sharded_borrowed_device_columns(pkg):
    """Borrowed columns under sharding: every rank passes the columns [first, last) fseq_shard_columns reports, column
    `first` at the base pointer (byte codes and 2-bit packed ones)."""
    import torch
    m, n, L, B = 300, 6000, 25, 50
    msa = fso.synth_msa(fso.synth_spec(51, 8, 200, 2e-3, 0), m, n)
    alphabet = np.sort(np.unique(msa))
    lut = np.zeros(256, dtype=np.uint8)
    lut[alphabet] = np.arange(len(alphabet), dtype=np.uint8)
    codes = lut[msa]                                               # dense codes in byte order, as the library assigns them
    for bits in (8, 2):
        keep = []

        def make_input(c):
            c0, c1 = c.shard_columns()
            if bits == 8:
                ld = (m + 15) // 16 * 16                                                       # column-major, one code per byte
                colmajor = np.zeros((c1 - c0, ld), dtype=np.uint8)
                colmajor[:, :m] = codes[:, c0:c1].T
                t = torch.from_numpy(colmajor).to("cuda:0")
                keep.append(t)
                c.set_device_columns(t.data_ptr(), ld, len(alphabet), keepalive=t)
            else:
                packed, ld = pkg.pack_columns(codes[:, c0:c1], 2)
                t = torch.from_numpy(packed).to("cuda:0")
                keep.append(t)
                c.set_device_columns_packed(t.data_ptr(), ld, len(alphabet), 2, keepalive=t)

        ctxs = run_world(pkg, 2, make_input, m, n, L, block_len=B)
        check_against_oracle(pkg, ctxs, msa, L)
