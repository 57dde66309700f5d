"""Model of csrc/fseq_blocktrie.hpp (phase A as a trie over 32-bit group words), statement by statement what the kernel
does, in plain Python: tests/test_proto_blocktrie.py checks it against the column sweep of tests/proto_blocks.py (which
is checked against the oracle's pBWT).

phase 1  groups of N = 32 / bits columns from the LAST to the first; a class keeps its id while it does not split
         (direct[id] = the word registered in this group), every other (class, word) pair gets a new id
phase 2  the nodes of a level {id, parent id, word} ranked in the order (rank of the parent, word); single children take
         their parent's place and divergence; siblings are ordered by word, a sibling's divergence is the last column in which
         it differs from the next smaller one
"""
import numpy as np

SENT = 0xFFFFFFFF        # direct[]: "nobody here yet"; a row whose word is this value always goes through the pair table


def group_words(codes, k0, kend, bits):
    """[levels][m] group words, level t = group levels - 1 - t; column c of a group in bits [bits * c, bits * (c + 1))."""
    N = 32 // bits
    levels = (kend - k0 + N - 1) // N
    m = codes.shape[0]
    out = []
    for t in range(levels):
        g = levels - 1 - t
        w = np.zeros(m, dtype=np.uint64)
        for c in range(N):
            k = k0 + N * g + c
            if k < kend:
                w |= codes[:, k].astype(np.uint64) << np.uint64(bits * c)
        out.append((g, w))
    return out


def blocktrie(codes, k0, nb, bits, row_order=None):
    """-> rank[m] (by row), keyd[D], D  -- as proto_blocks.phase_a.  row_order: the order in which the rows reach a group
    (the kernel's is whatever the hardware makes of 1024 threads: the result must not depend on it)."""
    m, n = codes.shape
    kend = min(n, k0 + nb)
    N = 32 // bits
    ids = np.zeros(m, dtype=np.int64)
    next_id = 1
    levels = []
    rows = np.arange(m) if row_order is None else row_order
    for g, w in group_words(codes, k0, kend, bits):
        direct, pairs = {}, {}
        for r in rows:
            p, x = int(ids[r]), int(w[r])
            if p not in direct and x != SENT:
                direct[p] = x                              # the first row of the class in this group registers its word
            if direct.get(p) == x:
                continue                                   # ... and who carries it keeps the id
            if (p, x) not in pairs:
                pairs[(p, x)] = next_id
                next_id += 1
            ids[r] = pairs[(p, x)]
        nodes = [(p, p, x) for p, x in direct.items()] + [(i, p, x) for (p, x), i in pairs.items()]
        levels.append((g, nodes))
    # phase 2
    R = {0: 0}
    D = [kend - k0]
    for g, nodes in levels:
        nprev = len(D)
        cnt = [0] * nprev
        for (i, p, x) in nodes:
            cnt[R[p]] += 1
        base, run = [], 0
        for c in cnt:
            base.append(run)
            run += c
        Rn, Dn = {}, [None] * len(nodes)
        lst = []
        for (i, p, x) in nodes:
            r = R[p]
            if cnt[r] == 1:
                Rn[i] = base[r]
                Dn[base[r]] = D[r]
            else:
                lst.append((r, i, x))
        for (r, i, x) in lst:
            below = [y for (r2, _, y) in lst if r2 == r and y < x]
            at = base[r] + len(below)
            Rn[i] = at
            if not below:
                Dn[at] = D[r]
            else:
                diff = x ^ max(below)
                Dn[at] = N * g + (diff.bit_length() - 1) // bits + 1
        R, D = Rn, Dn
    rank = np.array([R[int(i)] for i in ids], dtype=np.int64)
    keyd = np.array([k0 + v for v in D], dtype=np.int64)
    return rank, keyd, len(D)
