"""CPU proof of the reduced phase C / pass 2 (tests/proto_reduced.py) against the oracle's pBWT on all rows."""
import numpy as np
import pytest

from oracle import fso
import proto_reduced as pr


def _states(msa, B):
    m, n = msa.shape
    p = fso.Pbwt(msa)
    out = []
    for b in range((n + B - 1) // B + 1):
        k = min(b * B, n)
        while p.idx < k:
            p.step()
        out.append((p.a, p.d))
    return out


SHAPES = [
    # m, n, L, B, X, K, Brec, mu, seed, kind
    (200, 1200, 15, 100, 15, 6, 150, 3e-3, 31, 0),
    (600, 2400, 40, 160, 31, 12, 500, 1e-3, 7, 0),
    (400, 1500, 25, 128, 20, 8, 300, 2e-3, 9, 1),        # sigma = 16
    (64, 600, 10, 50, 4, 3, 100, 5e-3, 5, 0),
]


@pytest.mark.parametrize("m,n,L,B,X,K,Brec,mu,seed,kind", SHAPES)
@pytest.mark.parametrize("margin", [None, 0])
def test_reduced_lists_equal_full_lists(m, n, L, B, X, K, Brec, mu, seed, kind, margin):
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    st = _states(msa, B)
    nb = (n + B - 1) // B
    checked = invalid = reduced_rows = 0
    for b in range(nb):
        k0, k1 = b * B, min(n, (b + 1) * B)
        a0, d0 = st[b]
        a1, d1 = st[b + 1]
        full, (fa, fd) = pr.block_lists_full(msa, k0, k1, a0, d0, L, X)
        assert np.array_equal(fa, a1) and np.array_equal(fd, d1)
        red, Lr, vmin = pr.block_lists_reduced(msa, k0, k1, a0, d0, a1, d1, L, X, margin=margin)
        reduced_rows += Lr
        for f, r in zip(full, red):
            if not r[4]:
                invalid += 1
                continue
            checked += 1
            assert f[0] == r[0] and f[2] == r[2] and f[3] == r[3]        # entries, complete, cum
            if f[2] or max(vmin, 1) == 1:
                assert f[1] == r[1]                                        # cnt0: what the DP reads of a complete list
    assert checked > n // 2
    assert reduced_rows < nb * m


def test_a_list_below_vmin_is_flagged_not_wrong():
    # a floor chosen far too high (every value but the newest is below it): the lists must be flagged, never different
    m, n, L, B, X = 300, 900, 20, 100, 25
    msa = fso.synth_msa(fso.synth_spec(3, 6, 200, 2e-3), m, n)
    st = _states(msa, B)
    b = 5
    k0, k1 = b * B, (b + 1) * B
    a0, d0 = st[b]
    a1, d1 = st[b + 1]
    full, _ = pr.block_lists_full(msa, k0, k1, a0, d0, L, X)
    vmin = k0 - 5
    a_red, d_red = pr.reduce_state(a0, d0, a1, d1, vmin)
    p = fso.Pbwt(np.ascontiguousarray(msa[a_red, k0:k1]), col0=k0)
    p.set_state(np.arange(len(a_red), dtype=np.uint32), d_red, k0)
    flagged = 0
    for k in range(k0, k1):
        p.step()
        v, c = p.counts()
        r = pr.emit_list(v, c, k, L, X, m_true=m, v_min=vmin)
        f = full[k - k0]
        if r[4]:
            assert f[0] == r[0] and f[3] == r[3]
        else:
            flagged += 1
    assert flagged > 0


@pytest.mark.parametrize("m,n,L,B,X,K,Brec,mu,seed,kind", SHAPES[:3])
def test_boundary_state_is_one_chain_step_from_the_block_state(m, n, L, B, X, K, Brec, mu, seed, kind):
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    st = _states(msa, B)
    rng = np.random.default_rng(seed)
    for b in (int(x) for x in rng.choice((n + B - 1) // B, size=4, replace=False)):
        k0, k1 = b * B, min(n, (b + 1) * B)
        a0, d0 = st[b]
        a1, d1 = st[b + 1]
        # block keys (phase A's ranks): position i of (a1, d1) starts a key iff d1 > k0
        head = d1 > k0
        head[0] = True
        leaf_of_row = np.zeros(m, dtype=np.int64)
        leaf_of_row[a1] = np.cumsum(head) - 1
        vmin = max(pr.choose_vmin(d0, k0, L, X + X // 4 + 8), 1)
        a_red, d_red = pr.reduce_state(a0, d0, a1, d1, vmin)
        sub = np.ascontiguousarray(msa[a_red, k0:k1])
        pred = fso.Pbwt(sub, col0=k0)
        pred.set_state(np.arange(len(a_red), dtype=np.uint32), d_red, k0)
        pall = fso.Pbwt(msa[:, k0:k1], col0=k0)
        pall.set_state(a0, d0, k0)
        for k in range(k0 + 1, k1 + 1):
            pred.step()
            pall.step()
            if (k - k0) % 17 and k != k1:
                continue
            cls_of_leaf, headd, ncls = pr.class_tables(a_red[pred.a], pred.d, leaf_of_row, k0)
            a, d = pr.chain_step(a0, d0, cls_of_leaf[leaf_of_row], headd)
            assert np.array_equal(a, pall.a), (b, k)
            assert np.array_equal(d, pall.d), (b, k)
