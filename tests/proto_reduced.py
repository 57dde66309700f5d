"""Python model of the REDUCED phase C (round 5): the per-column lists of a column block computed on a subset of the rows.

Claim.  Let (a0, d0) be the exact state in front of a block [k0, k1), (a1, d1) the exact state behind it, and v_min any
value.  The rows  S = { a1[i] : d1[i] >= v_min }  hold one representative of every class of rows that agree on the columns
[v_min - 1, k1).  The pBWT of the sub-alignment of those rows, started from the restriction of (a0, d0) to S (order kept,
divergence = the maximum of d0 over the positions skipped since the last kept row), has at every column k of the block the
same number of values >= v as the full pBWT, for every v >= v_min  (both count the distinct row substrings over
[v - 1, k]).  So every (value, count) entry of a per-column list with value >= v_min -- and the lump of entry 0 when
thr >= v_min -- is the full run's entry; a list that takes an entry below v_min (or runs out of values while rows are
missing) is flagged and the block is redone on all rows.

This file states that in numpy over the oracle's pBWT (tests/test_proto_reduced.py proves it against the full lists).
TEST INFRASTRUCTURE (uses oracle/).
"""
import numpy as np

from oracle import fso


def emit_list(values, counts, k, L, X, m_true=None, v_min=0):
    """The list of column k as k_columns emits it (csrc/fseq_kernels.hpp): entry 0 = (k + 1, count of the values
    >= thr = k + 2 - L), then the distinct values below thr, descending, while the below-thr counts in front do not
    exceed X.  values/counts ascending (the histogram after column k).  Returns (entries, cnt0, complete, cum, valid):
    valid is False when an entry below v_min was taken or the list ran out of values with rows missing."""
    m_red = int(counts.sum())
    if m_true is None:
        m_true = m_red
    thr = k + 2 - L if k + 2 > L else 0
    rec = values >= thr
    R = int(counts[rec].sum())
    ents = [(k + 1, R)]
    cumN = 0
    lump_ok = thr >= v_min
    vs = values[~rec][::-1]
    cs = counts[~rec][::-1]
    dug = False
    for v, c in zip(vs, cs):
        if cumN > X:
            break
        if v < v_min:
            dug = True
        ents.append((int(v), int(c)))
        cumN += int(c)
    cum = R + cumN
    if cumN <= X and m_red != m_true:
        # every value of the reduced histogram taken: the rows that are not there would have followed
        dug = True
    valid = lump_ok and not dug
    cnt0_red = int(counts[0]) if len(values) and values[0] == 0 else 0
    cnt0 = m_true - (m_red - cnt0_red)         # exact when v_min <= 1
    return ents, cnt0, (cum == m_true and m_red == m_true), cum, valid


def choose_vmin(d0, k0, L, Xp, W=4096):
    """Largest v with #{ v <= d0 < thr0 } > Xp, thr0 = the threshold of the block's first column; searched over the W
    values below thr0 (v_min = max(thr0 - W, 0) when there are not that many: the run then flags what it cannot prove)."""
    thr0 = k0 + 2 - L if k0 + 2 > L else 0
    lo = max(thr0 - W, 0)
    if thr0 == 0:
        return 0
    below = d0[(d0 < thr0) & (d0 >= lo)]
    hist = np.bincount(thr0 - 1 - below, minlength=thr0 - lo)      # bin j: value thr0 - 1 - j
    cum = np.cumsum(hist)
    idx = np.searchsorted(cum, Xp + 1)                             # first bin where the cumulative count exceeds Xp
    if idx >= len(cum):
        return lo
    return thr0 - 1 - int(idx)


def reduce_state(a0, d0, a1, d1, v_min):
    """Representatives and the reduced start state."""
    m = len(a0)
    keep_row = np.zeros(m, dtype=bool)
    keep_row[a1[d1 >= v_min]] = True
    kept = keep_row[a0]
    # divergence of a kept row = max of d0 since the last kept row (positions skipped + its own)
    a_red = a0[kept]
    idx = np.nonzero(kept)[0]
    starts = np.r_[0, idx[:-1] + 1]
    d_red = np.array([d0[s:e + 1].max() for s, e in zip(starts, idx)], dtype=np.uint32)
    return a_red.astype(np.uint32), d_red


def block_lists_full(msa, k0, k1, a0, d0, L, X):
    """Lists of the columns [k0, k1) from the full state; returns (lists, (a1, d1))."""
    m = msa.shape[0]
    p = fso.Pbwt(msa[:, k0:k1], col0=k0)
    p.set_state(a0, d0, k0)
    out = []
    for k in range(k0, k1):
        p.step()
        v, c = p.counts()
        out.append(emit_list(v, c, k, L, X, m_true=m))
    return out, (p.a, p.d)


def block_lists_reduced(msa, k0, k1, a0, d0, a1, d1, L, X, margin=None, W=4096):
    """Lists of the columns [k0, k1) from the reduced state.  Returns (lists, n_reduced_rows, v_min)."""
    m = msa.shape[0]
    Xp = X + (X // 4 + 8 if margin is None else margin)
    v_min = choose_vmin(d0, k0, L, Xp, W)
    a_red, d_red = reduce_state(a0, d0, a1, d1, v_min)
    sub = np.ascontiguousarray(msa[a_red, k0:k1])
    p = fso.Pbwt(sub, col0=k0)
    p.set_state(np.arange(len(a_red), dtype=np.uint32), d_red, k0)
    out = []
    for k in range(k0, k1):
        p.step()
        v, c = p.counts()
        out.append(emit_list(v, c, k, L, X, m_true=m, v_min=v_min))
    return out, len(a_red), v_min


def class_tables(a_red, d_red, leaf_of_row, k0):
    """Pass 2 from a reduced state at a column k of the block: the class of every leaf (block key) and the divergence in
    front of every class.  Heads = entries whose divergence lies inside the block (> k0)."""
    head = d_red > k0
    head[0] = True
    cls_of_entry = np.cumsum(head) - 1
    ncls = int(cls_of_entry[-1]) + 1
    cls_of_leaf = np.full(int(leaf_of_row.max()) + 1, -1, dtype=np.int64)
    cls_of_leaf[leaf_of_row[a_red]] = cls_of_entry
    headd = d_red[head]
    return cls_of_leaf, headd, ncls


def chain_step(a0, d0, key_of_row, keyd):
    """One chain step (phase B's): stable sort of a0 by key; a row whose predecessor has another key takes keyd[key], one
    whose predecessor has the same key the maximum of d0 over the old positions between them."""
    keys = key_of_row[a0]
    order = np.argsort(keys, kind="stable")
    a = a0[order]
    d = np.zeros(len(a0), dtype=np.uint32)
    ks = keys[order]
    for i in range(len(a0)):
        if i == 0 or ks[i] != ks[i - 1]:
            d[i] = keyd[ks[i]]
        else:
            d[i] = d0[order[i - 1] + 1: order[i] + 1].max()
    return a.astype(np.uint32), d
