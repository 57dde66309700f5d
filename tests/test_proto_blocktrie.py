"""The trie of phase A (tests/proto_blocktrie.py, the model of csrc/fseq_blocktrie.hpp) against the column sweep of the
block model (tests/proto_blocks.py phase_a, itself checked against the oracle): ranks, key divergences, number of keys."""
import numpy as np
import pytest

import fso
import proto_blocks as pb
import proto_blocktrie as pt


def _codes(msa):
    u = np.unique(msa)
    return np.searchsorted(u, msa).astype(np.int64)


def _check(codes, k0, nb, bits, row_order=None):
    rank, keyd, D = pb.phase_a(codes, k0, min(nb, codes.shape[1] - k0))
    r2, k2, D2 = pt.blocktrie(codes, k0, nb, bits, row_order)
    assert D == D2
    assert np.array_equal(rank, r2)
    assert np.array_equal(keyd, k2)


@pytest.mark.parametrize("m,n,K,Brec,mu,seed,kind,bits,B", [
    (300, 200, 8, 60, 5e-3, 101, 0, 2, 64),        # whole groups of 16 columns
    (300, 200, 8, 60, 5e-3, 101, 0, 2, 37),        # a last group of 5 columns, a last block of 15
    (257, 120, 6, 40, 1e-2, 102, 1, 4, 20),        # sigma = 16: groups of 8 columns
    (200, 90, 30, 25, 2e-2, 103, 1, 8, 11),        # 8 bits per symbol: groups of 4 columns
    (120, 64, 120, 500, 5e-2, 104, 0, 2, 32),      # every row its own founder: every class splits in the first groups
])
def test_trie_matches_the_column_sweep(m, n, K, Brec, mu, seed, kind, bits, B):
    codes = _codes(fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n))
    assert codes.max() < (1 << bits)
    for k0 in range(0, n, B):
        _check(codes, k0, B, bits)


def test_trie_does_not_depend_on_the_order_of_the_rows():
    """Which row registers its word in direct[] first, and which ids the pair table hands out, is a race between 1024 threads:
    ranks and divergences must come out the same for any order."""
    codes = _codes(fso.synth_msa(fso.synth_spec(105, 10, 30, 1e-2, 0), 400, 96))
    rng = np.random.default_rng(5)
    for _ in range(4):
        _check(codes, 16, 64, 2, rng.permutation(400))


def test_trie_on_equal_and_on_random_rows():
    codes = np.zeros((50, 40), dtype=np.int64)
    _check(codes, 0, 40, 2)                                        # one class throughout
    rng = np.random.default_rng(6)
    codes = rng.integers(0, 4, size=(90, 70)).astype(np.int64)      # all rows distinct after a few columns
    _check(codes, 3, 48, 2)
    codes[:, 20:36] = 3                                             # a group whose word is 0xFFFFFFFF (direct[]'s "nobody here yet")
    _check(codes, 4, 48, 2)
