"""Pure-Python restatement of the reference's greedy joiner, for small cases.

TEST INFRASTRUCTURE ONLY (oracle for SURVEY.md row N1).  Written from
founder-sequences/greedy_matcher.cc:31-465 and founder-sequences/join_context.cc:333-356, independently
of the C++ in founder-sequences_amd/csrc/fseq_join.hpp.  PARITY UNPINNED: the reference has no tests and
cannot be built here; libbio's radix_sort<true> is assumed descending and stable.
"""
import math
from collections import deque


def update_string_mappings(seq_count, seg_start_pos, permutation, divergence):
    """greedy_matcher.cc:31-68 -> (distinct, seq_mapping, inverse_mapping, run_lengths[1-based])."""
    seq_mapping, inverse, run_lengths = [], [None] * seq_count, [0]
    cur = 0
    for i in range(seq_count):
        s = int(permutation[i])
        if seg_start_pos < int(divergence[i]):
            run_lengths[len(seq_mapping)] = cur
            run_lengths.append(0)
            seq_mapping.append(s)
            cur = 0
        assert seq_mapping
        inverse[s] = len(seq_mapping) - 1
        cur += 1
    run_lengths[len(seq_mapping)] = cur
    return len(seq_mapping), seq_mapping, inverse, run_lengths


def copies(distinct, run_lengths, max_segment_size, seq_count):
    """update_seq_occurrences + radix_sort<true> + update_copies (greedy_matcher.cc:71-105)."""
    occ = sorted(((i, run_lengths[1 + i]) for i in range(distinct)), key=lambda p: -p[1])   # stable, descending
    cn = [1] * max_segment_size
    to_fill = max_segment_size - len(occ)
    rem = to_fill
    while True:
        for idx, count in occ:
            c = min(rem, int(math.ceil(1.0 * count / seq_count * to_fill)))
            rem -= c
            cn[idx] += c
            if rem == 0:
                return cn


def greedy_match(seq_count, max_segment_size, segments, A, D):
    """segments: list of (lb, rb); A, D: per segment arrays of length seq_count.
    Returns permutations[s][row]."""
    X = max_segment_size
    perms = [[0] * X for _ in segments]
    seg_start = 0
    ld, lmap, linv, lrl = update_string_mappings(seq_count, seg_start, A[0], D[0])
    seg_start = segments[0][1]
    lcn = copies(ld, lrl, X, seq_count)
    lslots = [deque() for _ in range(X)]
    i = 0
    for cls in range(ld):
        for j in range(lcn[cls]):
            lslots[cls].append(i + j)
            perms[0][i + j] = lmap[cls]
        i += lcn[cls]
    for t in range(1, len(segments)):
        rd, rmap, rinv, rrl = update_string_mappings(seq_count, seg_start, A[t], D[t])
        rcn = copies(rd, rrl, X, seq_count)
        pairs = sorted((linv[int(s)], rinv[int(s)]) for s in A[t])
        by_count = {}
        k = 0
        while k < len(pairs):
            j = k
            while j < len(pairs) and pairs[j] == pairs[k]:
                j += 1
            by_count.setdefault(j - k, []).append(pairs[k])
            k = j
        rrc = list(rcn)
        rslots = [deque() for _ in range(X)]

        def draw(l, r):
            slot = lslots[l].popleft()
            rslots[r].append(slot)
            perms[t][slot] = rmap[r]

        drew = True
        while drew:
            drew = False
            for count in sorted(by_count, reverse=True):
                keep = []
                for (l, r) in by_count[count]:
                    if lcn[l] and rrc[r]:
                        drew = True
                        lcn[l] -= 1
                        rrc[r] -= 1
                        draw(l, r)
                        keep.append((l, r))
                by_count[count] = keep
            by_count = {c: v for c, v in by_count.items() if v}
        i = j = 0
        while True:
            while i < ld and lcn[i] == 0:
                i += 1
            if i == ld:
                break
            while j < rd and rrc[j] == 0:
                j += 1
            if j == rd:
                break
            draw(i, j)
            lcn[i] -= 1
            rrc[j] -= 1
        ld, lmap, linv, lrl, lcn, lslots = rd, rmap, rinv, rrl, rcn, rslots
        seg_start = segments[t][1]
    return perms


def founders(msa, segments, perms, max_segment_size):
    """join_context::output_in_permutation_order (join_context.cc:333-356) -> list of bytes lines."""
    out = []
    for row in range(max_segment_size):
        parts = []
        for s, (lb, rb) in enumerate(segments):
            parts.append(bytes(msa[perms[s][row], lb:rb]))
        out.append(b"".join(parts))
    return out
