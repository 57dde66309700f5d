"""Pure-Python model of the column-block decomposition the HIP kernels implement.

TEST INFRASTRUCTURE: a slow, readable statement of the *device algorithm* (phases A/B/C/D), used by
the CPU tests to prove that the decomposition reproduces the oracle bit for bit before any of it
runs on a GPU.  It is not a fallback: the product never imports it.

Phase A  per column block [k0,k0+nb): pBWT from the identity -> dense rank of every row's block key
         (co-lex order) + the in-block divergence in front of each distinct key.
Phase B  serial over blocks: stable sort of the boundary order by block rank, as LSD digit passes of
         the same sigma-bucket partition step; divergences ride along; bucket firsts take keyd.
Phase C  per block from its exact boundary state: true per-column updates; divergences replaced by
         order-preserving ids; per column the top of the divergence histogram (descending values
         until the cumulative count exceeds X).
Phase D  DP over columns from the top lists, candidates visited in descending value order with
         exact pruning; rmq.hh semantics through prefix/suffix first-min arrays + the (quirky)
         sparse table.
"""
import numpy as np

U32MAX = 0xFFFFFFFF


def colstep(a, d, sym, first_val):
    """One stable sigma-bucket partition step. sym[i] = symbol of a[i]."""
    m = len(a)
    order = np.argsort(sym, kind="stable")
    a2 = a[order]
    d2 = np.empty(m, dtype=np.int64)
    last = {}
    dst_of = np.empty(m, dtype=np.int64)
    dst_of[order] = np.arange(m)
    for i in range(m):
        c = int(sym[i])
        if c in last:
            p = last[c]
            d2[dst_of[i]] = d[p + 1:i + 1].max()
        else:
            d2[dst_of[i]] = first_val
        last[c] = i
    return a2, d2


def phase_a(codes, k0, nb):
    """codes: [m, n] dense symbols. Returns rank[m] (by row), keyd[D], D."""
    m = codes.shape[0]
    a = np.arange(m, dtype=np.int64)
    d = np.full(m, k0, dtype=np.int64)
    for j in range(nb):
        k = k0 + j
        a, d = colstep(a, d, codes[a, k], k + 1)
    first = np.ones(m, dtype=bool)
    first[1:] = d[1:] > k0
    r_of_pos = np.cumsum(first) - 1
    rank = np.empty(m, dtype=np.int64)
    rank[a] = r_of_pos
    keyd = d[first]
    return rank, keyd, int(first.sum())


def phase_b_step(a, d, rank, keyd, D, digit_bits=4):
    """Boundary state at k0 -> boundary state at k0+nb, using only the block's rank/keyd."""
    ndig = 1
    while (1 << (digit_bits * ndig)) < D:
        ndig += 1
    mask = (1 << digit_bits) - 1
    for p in range(ndig):
        sym = (rank[a] >> (digit_bits * p)) & mask
        a, d = colstep(a, d, sym, -1)          # firsts get garbage (-1): never consumed, fixed below
    r = rank[a]
    first = np.ones(len(a), dtype=bool)
    first[1:] = r[1:] != r[:-1]
    d = d.copy()
    d[first] = keyd[r[first]]
    return a, d


def phase_c(codes, a, d, k0, nb, X, L):
    """Returns per column (k0..k0+nb-1) a dict(vals, cnts, cnt0, complete): entry 0 lumps every
    value >= thr = (k+1)+1-L (the DP clips them all to the same cut bound, lp.cc:444-445); then the
    distinct values below thr, descending, truncated once their cumulative count exceeds X."""
    V = np.unique(d)
    D0 = len(V)
    ids = np.searchsorted(V, d)
    cnt = np.zeros(D0 + nb, dtype=np.int64)
    np.add.at(cnt, ids, 1)
    out = []
    for j in range(nb):
        k = k0 + j
        a, ids2 = colstep(a, ids, codes[a, k], D0 + j)
        # incremental table update (what the kernel does with LDS atomics) == full recount
        cnt[:] = 0
        np.add.at(cnt, ids2, 1)
        ids = ids2
        thr = max(0, k + 2 - L)
        vals, cnts, cum, complete = [k + 1], [0], 0, True      # cum: counts below thr only
        for i in range(D0 + j, -1, -1):
            if cnt[i] == 0:
                continue
            v = int(V[i]) if i < D0 else k0 + (i - D0) + 1
            if v >= thr:
                cnts[0] += int(cnt[i])
                continue
            if cum > X:
                complete = False
                break
            vals.append(v)
            cnts.append(int(cnt[i]))
            cum += int(cnt[i])
        cnt0 = int(cnt[0]) if V[0] == 0 else 0
        out.append(dict(vals=vals, cnts=cnts, cnt0=cnt0, complete=complete))
    return out, a


class DeviceRmq:
    """rmq.hh semantics (block 64) over M[] with O(1) partial-block answers.

    K[t] is a 64-bit mask over the block of t: bit p (p <= t mod 64) is set iff
    M[p] <= min(M[p+1..t]) (the monotonic stack after scanning the block up to t, popping only
    strictly greater keys).  The first minimum of [b, t] inside one block is the lowest set bit of
    K[t] at or above b: stack keys are non-decreasing bottom to top and equal keys all stay, so the
    lowest surviving position >= b is the leftmost minimum (std::min_element, rmq.hh:116)."""

    def __init__(self, n):
        self.M = np.full(n, U32MAX, dtype=np.int64)
        self.K = [0] * n
        self.T = [[]]
        self.filled = 0

    def append(self, t, v):
        assert t == self.filled
        M = self.M
        M[t] = v
        base = t - (t % 64)
        mask = 1 << (t - base)
        runmin = v
        for p in range(t - 1, base - 1, -1):
            if M[p] <= runmin:
                mask |= 1 << (p - base)
            runmin = min(runmin, M[p])
        self.K[t] = mask
        self.filled += 1
        if (t + 1) % 64 == 0:
            bnum = (t + 1) // 64
            new_smp = base + self._ctz(mask)
            self.T[0].append(new_smp)
            p = 1
            while (1 << p) <= bnum:
                if len(self.T) <= p:
                    self.T.append([])
                smp = self.T[p - 1][bnum - (1 << p)]
                self.T[p].append(new_smp if M[new_smp] < M[smp] else smp)
                p += 1

    @staticmethod
    def _ctz(x):
        return (x & -x).bit_length() - 1

    def inblock(self, b, e):
        """first minimum of [b, e), b and e-1 in the same block"""
        assert b // 64 == (e - 1) // 64
        return b + self._ctz(self.K[e - 1] >> (b % 64))

    def naive(self, b, e):
        M = self.M
        bb, eb = b // 64, (e - 1) // 64
        if bb == eb:
            return self.inblock(b, e)
        assert eb == bb + 1
        left, right = self.inblock(b, bb * 64 + 64), self.inblock(eb * 64, e)
        return right if M[right] < M[left] else left

    def query(self, beg, end):
        M = self.M
        beg_block = beg // 64 + 1
        end_block = end // 64
        if beg_block >= end_block:
            return self.naive(beg, end)
        pow2 = (end_block - beg_block).bit_length() - 1
        smp1 = self.T[pow2][beg_block]
        smp2 = self.T[pow2][end_block - (1 << pow2)]
        smp = smp2 if M[smp2] < M[smp1] else smp1
        left = self.inblock(beg, beg_block * 64)
        smp = left if M[left] < M[smp] else smp
        if end == end_block * 64:
            return smp
        right = self.inblock(end_block * 64, end)
        return right if M[right] < M[smp] else smp


def dp_from_list(rec, rmq, LB, m, L, end):
    """One DP cell from a descending top list; returns (lb, max, size, overflow)."""
    vals, cnts = rec["vals"], rec["cnts"]
    best = None   # (value, lb, size)
    cum = 0
    stopped = False
    for i in range(len(vals)):
        if vals[i] == 0:
            break                         # bottom entry: its count belongs to the whole-range candidate
        cum += cnts[i]
        if best is not None and cum > best[0]:
            stopped = True
            break
        if i + 1 >= len(vals):
            break
        lo, hi = vals[i + 1], vals[i]
        if lo == 0:
            continue                      # range (v0 == 0, v1) is never a candidate (lp.cc:416-428)
        c = min(hi, end + 1 - L)
        if lo < L:
            if L < c:
                lo = L
            else:
                continue
        if lo < c:
            idx = rmq.query(lo - L, c - L)
            v = max(int(rmq.M[idx]), cum)
            if best is None or v <= best[0]:
                best = (v, idx + L, cum)
    overflow = False
    if not stopped:
        if not rec["complete"]:
            overflow = True               # ran out of list without proving the rest irrelevant
        elif rec["cnt0"] > 0:
            v = m - rec["cnt0"]
            if best is None or v <= best[0]:
                best = (v, 0, v)
    if best is None or m <= best[0]:
        best = (m, 0, m)
    return best[1], best[0], best[2], overflow


def segment(codes, L, B, X, digit_bits=4):
    """Full long path through phases A-D + traceback + merge + snapshots. codes: [m,n] dense."""
    m, n = codes.shape
    nblocks = (n + B - 1) // B
    blocks = [(b * B, min(B, n - b * B)) for b in range(nblocks)]
    ranks = [phase_a(codes, k0, nb) for k0, nb in blocks]
    # phase B
    a = np.arange(m, dtype=np.int64)
    d = np.zeros(m, dtype=np.int64)
    bstate = []
    for (k0, nb), (rank, keyd, D) in zip(blocks, ranks):
        bstate.append((a, d))
        a, d = phase_b_step(a, d, rank, keyd, D, digit_bits)
    bstate.append((a, d))
    # phase C
    lists = []
    for (k0, nb), (sa, sd) in zip(blocks, bstate):
        recs, a_end = phase_c(codes, sa, sd, k0, nb, X, L)
        lists.extend(recs)
    # phase D
    dp_size = n - L + 1
    rmq = DeviceRmq(dp_size)
    LB = np.zeros(dp_size, dtype=np.int64)
    SZ = np.full(dp_size, U32MAX, dtype=np.int64)
    p2lim = min(2 * L, n - L) - 1
    overflow = False
    for end in range(L, n - L + 1):
        rec = lists[end - 1]
        if end <= p2lim:
            lb, mx, sz = 0, m - rec["cnt0"], m - rec["cnt0"]
        else:
            lb, mx, sz, ov = dp_from_list(rec, rmq, LB, m, L, end)
            overflow |= ov
        t = end - L
        LB[t], SZ[t] = lb, sz
        rmq.append(t, mx)
    lb, mx, sz, ov = dp_from_list(lists[n - 1], rmq, LB, m, L, n)
    overflow |= ov
    LB[n - L], SZ[n - L] = lb, sz
    rmq.M[n - L] = mx
    M = rmq.M
    # traceback
    tb = []
    t = n - L
    while True:
        tb.append((int(LB[t]), t + L, int(M[t]), int(SZ[t])))
        if LB[t] == 0:
            break
        t = int(LB[t]) - L
    tb.reverse()
    max_seg = tb[-1][2]
    # merge from the lists
    red = []
    cur_lb, prev_size, prev = 0, tb[0][3], 0
    for j in range(1, len(tb)):
        rec = lists[tb[j][1] - 1]
        cnt, known = 0, rec["complete"]
        for v, c in zip(rec["vals"], rec["cnts"]):
            if v > cur_lb:
                cnt += c
            else:
                known = True
                break
        if not known and cnt <= max_seg:
            overflow = True
        if cnt <= max_seg:
            prev_size = cnt
        else:
            red.append((cur_lb, tb[prev][1], prev_size))
            prev_size = tb[j][3]
            cur_lb = tb[prev][1]
        prev = j
    red.append((cur_lb, tb[prev][1], prev_size))
    # snapshots from block boundary states
    snaps = []
    for (_, rb, _) in red:
        b = min(rb // B, nblocks)
        sa, sd = bstate[b]
        for k in range(b * B, rb):
            sa, sd = colstep(sa, sd, codes[sa, k], k + 1)
        snaps.append((sa, sd))
    return dict(LB=LB, M=M, SZ=SZ, traceback=tb, max_segment_size=max_seg, reduced=red,
                snaps=snaps, overflow=overflow, bstate=bstate, lists=lists)
