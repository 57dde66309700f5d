"""The north-star partition as a conformance path (fseq_rowshard_pbwt, include/fseq.h): the POSITIONS of the pBWT
order sharded over ranks, every column two all-reduces (the column + the scattered (a, d); one summary per rank:
bucket histogram, divergence carry).  The ranks are threads of this process on the one GPU (ThreadWorld); every rank
is given the symbols of its own ROWS only -- the rest of its device columns is zeroed -- so a result that matches the
oracle's serial pBWT proves the column exchange, the histogram / carry exchange and the scatter."""
import importlib
import threading

import numpy as np
import pytest

import fso

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("founder-sequences_amd")


def run_rowshard(pkg, msa, ncols, world):
    import torch
    fdist = importlib.import_module("founder-sequences_amd.dist")
    m, n = msa.shape
    alphabet = np.unique(msa)
    sigma = len(alphabet)
    bits = 2 if sigma <= 4 else 4 if sigma <= 16 else 8
    codes = np.searchsorted(alphabet, msa).astype(np.uint8)
    packed, ld = pkg.pack_columns(codes[:, :max(ncols, 1)], bits)
    tw = fdist.ThreadWorld(world)
    words = pkg.rowshard_xbuf_words(m, bits, world)
    out = [None] * world
    errs = [None] * world

    def work(r):
        try:
            lo, hi = pkg.rowshard_rows(m, bits, r, world)
            mine = np.zeros_like(packed)
            per = 8 // bits
            mine[:, lo // per:(hi + per - 1) // per] = packed[:, lo // per:(hi + per - 1) // per]      # my rows' bytes only
            cols = torch.from_numpy(mine).to("cuda:0")
            tr, allreduce = tw.transport(words, r, "cuda:0")
            out[r] = pkg.rowshard_pbwt(cols.data_ptr(), ld, m, sigma, bits, ncols, r, world, tr.ptr, tr.words,
                                       allreduce if world > 1 else None) + (tr.calls,)
        except BaseException as e:
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
    a = np.zeros(m, dtype=np.uint32)
    d = np.zeros(m, dtype=np.uint32)
    covered = np.zeros(m, dtype=np.int32)
    for (ar, dr, lo, hi, ms, nex, calls) in out:
        a[lo:hi] = ar[lo:hi]
        d[lo:hi] = dr[lo:hi]
        covered[lo:hi] += 1
        if world > 1:
            assert calls == nex
    assert np.all(covered == 1)                       # every position on exactly one rank
    npass = 1 if sigma <= 4 else 2 if sigma <= 16 else 4
    assert out[0][5] == (1 + 2 * npass * ncols if ncols else 0)      # X0 once, then X1+X2 and X3(+X0) per partition
    return a, d


@pytest.mark.parametrize("world", [1, 2, 3, 4])
def test_row_sharded_pbwt_matches_the_serial_sweep(pkg, world):
    for (m, n, K, Brec, mu, seed, kind, ncols) in [(300, 40, 8, 200, 2e-2, 61, 0, 40), (2504, 24, 24, 5000, 1e-3, 62, 0, 24),
                                                    (20000, 12, 12, 100, 3e-3, 63, 0, 12), (900, 16, 10, 300, 1e-2, 64, 1, 16),
                                                    (7, 30, 3, 10, 5e-2, 65, 0, 30)]:
        if m < world * 2:
            continue
        msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
        a, d = run_rowshard(pkg, msa, ncols, world)
        p = fso.Pbwt(msa)
        while p.idx < ncols:
            p.step()
        assert np.array_equal(a, p.a), (m, world)
        assert np.array_equal(d, p.d), (m, world)


def test_row_sharded_state_equals_the_block_boundary_state_of_the_product_path(pkg):
    """The conformance path and the product path (phase A-C of fseq_run_segmentation) meet at a block boundary."""
    m, n, L = 2500, 4000, 50
    msa = fso.synth_msa(fso.synth_spec(0x5EED0002, 16, 2000, 1e-4, 0), m, n)
    ctx = pkg.SegmentationContext(m, n, L, block_len=64)
    ctx.set_sequences(msa)
    try:
        ctx.run()
    except pkg.NoReduction:
        pass
    a_ref, d_ref = ctx.debug_block_state(3)
    a, d = run_rowshard(pkg, msa, 3 * 64, 2)
    assert np.array_equal(a, a_ref) and np.array_equal(d, d_ref)


def test_row_sharded_arguments_are_checked(pkg):
    import torch
    buf = torch.zeros(4096, dtype=torch.int32, device="cuda:0")
    cols = torch.zeros(16 * 4, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(pkg.FseqError):        # exchange buffer too small
        pkg.rowshard_pbwt(cols.data_ptr(), 16, 40, 4, 2, 4, 0, 1, buf.data_ptr(), 8)
    with pytest.raises(pkg.FseqError):        # two ranks need an all-reduce
        pkg.rowshard_pbwt(cols.data_ptr(), 16, 40, 4, 2, 4, 0, 2, buf.data_ptr(), 4096)
    with pytest.raises(pkg.FseqError):        # 3 bits per symbol is no layout
        pkg.rowshard_pbwt(cols.data_ptr(), 16, 40, 4, 3, 4, 0, 1, buf.data_ptr(), 4096)
    assert pkg.rowshard_xbuf_words(40, 2, 1) > 0 and pkg.rowshard_xbuf_words(40, 3, 1) == 0
