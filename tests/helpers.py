"""Brute-force reference computations used by the CPU tests (numpy / pure Python, small sizes)."""
import numpy as np


def brute_pbwt(msa, k):
    """State before column k: a_k = stable co-lex order of the prefixes [0,k); d_k[i] = leftmost
    column from which row a_k[i] equals row a_k[i-1] up to column k-1 (k when they differ at k-1,
    and for i == 0).  SURVEY.md Appendix B A1."""
    m = msa.shape[0]
    if k == 0:
        return np.arange(m, dtype=np.uint32), np.zeros(m, dtype=np.uint32)
    keys = [bytes(msa[r, :k][::-1]) for r in range(m)]
    a = sorted(range(m), key=lambda r: (keys[r], r))
    d = np.zeros(m, dtype=np.uint32)
    d[0] = k
    for i in range(1, m):
        x, y = msa[a[i], :k], msa[a[i - 1], :k]
        j = k
        while j > 0 and x[j - 1] == y[j - 1]:
            j -= 1
        d[i] = j
    return np.array(a, dtype=np.uint32), d


def distinct_count(msa, lb, rb):
    return len({bytes(msa[r, lb:rb]) for r in range(msa.shape[0])})


def optimal_max_segment_size(msa, L):
    """Independent O(n^2 m) min-max segmentation with every segment >= L columns."""
    m, n = msa.shape
    INF = 10 ** 9
    best = [INF] * (n + 1)
    best[0] = 0
    for end in range(L, n + 1):
        # extend leftwards, tracking distinct count of [t, end) incrementally via hashing suffixes
        cur = INF
        for t in range(end - L, -1, -1):
            if t != 0 and (t < L or best[t] >= INF):
                continue
            c = distinct_count(msa, t, end)
            v = max(best[t], c) if t else c
            if v < cur:
                cur = v
            if c >= cur:
                # distinct count only grows as t decreases
                break
        best[end] = cur
    return best[n]


def owned_dp_mask(ctx, n, L):
    """DP entries a context answers for in fseq_debug_dp: the entries some cell writes (the L - 1 in front of the final
    cell's are never written), restricted -- on a rank of a sharded run that keeps windows, not whole arrays -- to the
    entries the rank computed itself."""
    import numpy as np
    written = np.ones(n - L + 1, dtype=bool)
    written[n - 2 * L + 1:n - L] = False
    lo, hi, final_cell, whole = ctx.debug_dp_owned()
    if whole:
        return written
    mine = np.zeros(n - L + 1, dtype=bool)
    mine[lo:hi] = True
    if final_cell:
        mine[n - L] = True
    return written & mine
