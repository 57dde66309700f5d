"""join_oracle.py -- TEST INFRASTRUCTURE: independent Python restatement of the rules the non-greedy
joiners follow, used only by tests/ to check founder-sequences_amd/csrc/fseq_join.hpp.

Restated from /root/reference: join_context.cc:63-126 (copy-number preparation),
create_segment_texts_task.cc:15-81 (segment texts and their copies), merge_segments_task.cc:133-195
(intersection weights) and bipartite_matcher.cc:95-151 (permutations follow the matchings).
PARITY UNPINNED: the reference's matching comes from Lemon 1.3.1 (absent) and its tie orders from
std::sort / std::shuffle; the checks below are therefore order-free (multisets, optimal weights)."""
import math
from collections import Counter

import numpy as np


def classes(m, lb, a, d):
    """Runs of rows that agree on [lb, rb), in pBWT order (unique_substring_count_idxs_lhs, Appendix B A7):
    list of row-id lists."""
    out = []
    for i in range(m):
        if i == 0 or d[i] > lb:
            out.append([])
        out[-1].append(int(a[i]))
    return out


def bipartite_copy_multiset(m, X, cls):
    """Multiset of (class size, number of slots) create_segment_texts_task produces: every class once,
    then copies in descending size order, ceil(size / m * remaining) each, then round-robin."""
    sizes = sorted((len(c) for c in cls), reverse=True)
    slots = [1] * len(sizes)
    remaining = X - len(sizes)
    for i, sz in enumerate(sizes):
        if remaining == 0:
            break
        cn = min(remaining, math.ceil(1.0 * sz / m * remaining))
        slots[i] += cn
        remaining -= cn
    while remaining:
        for i in range(len(sizes)):
            if remaining == 0:
                break
            slots[i] += 1
            remaining -= 1
    return Counter(zip(sizes, slots))


def random_copy_multiset(X, cls):
    """Multiset of (class size, copy number) of join_context.cc:88-114 (ascending sort, proportional
    distribution from the largest down with the class count as divisor, then round-robin)."""
    sizes = sorted(len(c) for c in cls)
    k = len(sizes)
    empty = X - k
    remaining = empty
    cn = [0] * k
    for i in range(k - 1, -1, -1):
        add = min(remaining, math.ceil(1.0 * sizes[i] / k * empty))
        cn[i] = 1 + add
        remaining -= add
    while remaining:
        for i in range(k - 1, -1, -1):
            if remaining == 0:
                break
            cn[i] += 1
            remaining -= 1
    return Counter(zip(sizes, cn))


def slot_classes(perm_row, cls):
    """Class index of every slot of one segment's permutation row (slots hold the class's smallest row id)."""
    rep = {min(c): i for i, c in enumerate(cls)}
    return [rep[int(r)] for r in perm_row]


def optimal_weight(slots_l, cls_l, slots_r, cls_r):
    """Maximum total |rows_l n rows_r| over perfect matchings between two slot lists (scipy's
    linear_sum_assignment as the independent solver)."""
    from scipy.optimize import linear_sum_assignment
    sets_l = [set(c) for c in cls_l]
    sets_r = [set(c) for c in cls_r]
    base = np.array([[len(x & y) for y in sets_r] for x in sets_l], dtype=np.int64)
    w = base[np.ix_(slots_l, slots_r)]
    r, c = linear_sum_assignment(w, maximize=True)
    return int(w[r, c].sum()), base
