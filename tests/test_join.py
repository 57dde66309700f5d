"""Greedy segment joining + founders writer, host logic.
CPU: the C++ matcher behind the C ABI against the pure-Python restatement in oracle/greedy_oracle.py
and against structural invariants.  GPU: the same through a real segmentation."""
import importlib
import os

import numpy as np
import pytest

import fso
import greedy_oracle as go
from helpers import distinct_count


@pytest.fixture(scope="module")
def pkg():
    build = importlib.import_module("founder-sequences_amd.build")
    build.build()
    return importlib.import_module("founder-sequences_amd")


def _segment(m, n, L, K, Brec, mu, seed, kind=0):
    msa = fso.synth_msa(fso.synth_spec(seed, K, Brec, mu, kind), m, n)
    res = fso.segment_long(msa, L, debug=True)
    assert res["status"] == 0
    return np.ascontiguousarray(msa), res


CASES = [(8, 1000, 10, 3, 100, 5e-3, 0x5EED0001), (24, 400, 7, 4, 60, 1e-2, 11), (40, 300, 20, 5, 50, 5e-3, 12),
         (70, 500, 9, 6, 45, 8e-3, 21), (200, 1200, 15, 6, 150, 3e-3, 31), (33, 700, 12, 4, 80, 5e-3, 9)]


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed", CASES)
def test_greedy_matcher_matches_python_oracle_and_invariants(pkg, m, n, L, K, Brec, mu, seed):
    msa, res = _segment(m, n, L, K, Brec, mu, seed)
    red = res["reduced"]
    X = res["max_segment_size"]
    segs = [(int(x["lb"]), int(x["rb"])) for x in red]
    perm = pkg.greedy_match_host(m, X, red["lb"], red["rb"], res["a"], res["d"])
    ref = go.greedy_match(m, X, segs, res["a"], res["d"])
    assert perm.tolist() == ref
    # invariants: X founders; in every segment every distinct substring of the input is present,
    # nothing else is; copy numbers are at least 1
    assert perm.shape == (len(segs), X) and perm.max() < m
    for s, (lb, rb) in enumerate(segs):
        have = {bytes(msa[r, lb:rb]) for r in perm[s]}
        want = {bytes(msa[r, lb:rb]) for r in range(m)}
        assert have == want
        assert len(want) == red["segment_size"][s] <= X
    lines = go.founders(msa, segs, perm, X)
    assert len(lines) == X and all(len(x) == n for x in lines)


def test_single_segment_and_full_copy_numbers(pkg):
    # one segment: classes in pBWT run order, copies proportional to class size (greedy_matcher.cc:81-160)
    m = 10
    a = np.array([[3, 7, 1, 0, 2, 4, 5, 6, 8, 9]], dtype=np.uint32)
    d = np.array([[20, 0, 0, 5, 0, 0, 0, 9, 0, 0]], dtype=np.uint32)      # classes {3,7,1} {0,2,4,5} {6,8,9}
    perm = pkg.greedy_match_host(m, 6, [0], [20], a, d)
    # to_fill = 3; sorted by size desc (stable): class1 (4), class0 (3), class2 (3): ceil(4/10*3)=2, ceil(3/10*3)=1 -> done
    assert perm.tolist() == [[3, 3, 0, 0, 0, 6]]
    assert go.greedy_match(m, 6, [(0, 20)], a, d) == perm.tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("host_front", [False, True])
def test_founders_through_the_gpu_path(pkg, tmp_path, monkeypatch, host_front):
    """fseq_join_greedy: class tables and co-occurrence edges built on the device (fseq_joinprep.hpp), copies and edge
    drawing on the host -- and, with FSEQ_JOIN_HOST, everything on the host from the boundary states: both equal the
    independent oracle."""
    if host_front:
        monkeypatch.setenv("FSEQ_JOIN_HOST", "1")
    for (m, n, L, K, Brec, mu, seed) in CASES[:4] + [(300, 2000, 25, 8, 200, 2e-3, 22), (2500, 4000, 50, 16, 2000, 1e-4, 0x5EED0002)]:
        msa, res = _segment(m, n, L, K, Brec, mu, seed)
        ctx = pkg.SegmentationContext(m, n, L)
        ctx.set_sequences(msa)
        ctx.run()
        perm = ctx.join_greedy()
        segs = [(int(x["lb"]), int(x["rb"])) for x in res["reduced"]]
        ref = go.greedy_match(m, res["max_segment_size"], segs, res["a"], res["d"])
        assert perm.tolist() == ref
        path = str(tmp_path / ("founders_%d.txt" % seed))
        ctx.write_founders(msa, perm, path)
        got = open(path, "rb").read().split(b"\n")
        assert got[-1] == b"" and got[:-1] == go.founders(msa, segs, ref, res["max_segment_size"])


# ---- non-greedy joiners: order-free checks against oracle/join_oracle.py -------------
import join_oracle as jo
from collections import Counter

N3_CASES = CASES + [(120, 900, 10, 9, 70, 1e-2, 77)]


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed", N3_CASES)
def test_bipartite_joiner_slots_and_optimal_matchings(pkg, m, n, L, K, Brec, mu, seed):
    msa, res = _segment(m, n, L, K, Brec, mu, seed)
    red, X = res["reduced"], res["max_segment_size"]
    perm, weights = pkg.bipartite_match_host(m, X, red["lb"], red["rb"], res["a"], res["d"])
    assert perm.shape == (len(red), X) and len(weights) == len(red) - 1
    cls = [jo.classes(m, int(red["lb"][s]), res["a"][s], res["d"][s]) for s in range(len(red))]
    slots = []
    for s in range(len(red)):
        assert len(cls[s]) == red["segment_size"][s]
        sl = jo.slot_classes(perm[s], cls[s])                   # every slot shows a class representative
        cnt = Counter(sl)
        assert set(cnt) == set(range(len(cls[s])))               # every distinct substring is present
        assert Counter((len(cls[s][i]), c) for i, c in cnt.items()) == jo.bipartite_copy_multiset(m, X, cls[s])
        slots.append(sl)
    for s in range(1, len(red)):
        best, base = jo.optimal_weight(slots[s - 1], cls[s - 1], slots[s], cls[s])
        realised = sum(int(base[l, r]) for l, r in zip(slots[s - 1], slots[s]))   # founder i keeps slot i
        assert realised == best == int(weights[s - 1])


@pytest.mark.parametrize("m,n,L,K,Brec,mu,seed", N3_CASES[:5])
def test_random_joiner_copy_numbers_and_seed(pkg, m, n, L, K, Brec, mu, seed):
    msa, res = _segment(m, n, L, K, Brec, mu, seed)
    red, X = res["reduced"], res["max_segment_size"]
    p0 = pkg.random_join_host(m, X, red["lb"], red["rb"], res["a"], res["d"], 0)
    p0b = pkg.random_join_host(m, X, red["lb"], red["rb"], res["a"], res["d"], 0)
    p1 = pkg.random_join_host(m, X, red["lb"], red["rb"], res["a"], res["d"], 12345)
    assert np.array_equal(p0, p0b)
    for s in range(len(red)):
        cls = jo.classes(m, int(red["lb"][s]), res["a"][s], res["d"][s])
        first = {c[0]: len(c) for c in cls}                      # random slots hold the run's first row in pBWT order
        for p in (p0, p1):
            cnt = Counter(int(r) for r in p[s])
            assert set(cnt) == set(first)
            assert Counter((first[r], c) for r, c in cnt.items()) == jo.random_copy_multiset(X, cls)
    if X > 2 and len(red) > 1:
        assert not np.array_equal(p0, p1)


def test_kuhn_munkres_small_known_answer(pkg):
    # two segments, three classes each; the identity pairing is the unique optimum (weights 3 + 2 + 2)
    m = 7
    a = np.array([[0, 1, 2, 3, 4, 5, 6], [0, 1, 2, 5, 6, 3, 4]], dtype=np.uint32)
    d = np.array([[10, 0, 0, 5, 0, 7, 0], [20, 10, 10, 15, 10, 18, 10]], dtype=np.uint32)   # classes {0,1,2} {3,4} {5,6} | {0,1,2} {5,6} {3,4}
    perm, weights = pkg.bipartite_match_host(m, 3, [0, 10], [10, 20], a, d)
    assert weights.tolist() == [7]
    assert perm.tolist() == [[0, 3, 5], [0, 3, 5]]


@pytest.mark.gpu
def test_nongreedy_joiners_through_the_gpu_path(pkg, tmp_path):
    m, n, L, K, Brec, mu, seed = CASES[4]
    msa, res = _segment(m, n, L, K, Brec, mu, seed)
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.set_sequences(msa)
    ctx.run()
    red, X = res["reduced"], res["max_segment_size"]
    host_perm, _ = pkg.bipartite_match_host(m, X, red["lb"], red["rb"], res["a"], res["d"])
    assert np.array_equal(ctx.join_bipartite(), host_perm)
    assert np.array_equal(ctx.join_random(7), pkg.random_join_host(m, X, red["lb"], red["rb"], res["a"], res["d"], 7))
    # founders: every line is a concatenation of input substrings, one per segment
    path = str(tmp_path / "founders_bp.txt")
    ctx.write_founders(msa, host_perm, path)
    lines = open(path, "rb").read().split(b"\n")[:-1]
    assert len(lines) == X and all(len(x) == n for x in lines)
    # [r5] the device writer (the lines put together from the resident packed alignment): the same bytes
    pdev = str(tmp_path / "founders_bp_dev.txt")
    ctx.write_founders_device(host_perm, pdev)
    assert open(pdev, "rb").read() == open(path, "rb").read()
    # segments files: bipartite = X lines per segment (texts, row lists, copied-from); random = one line per class
    pb = str(tmp_path / "segments_bp.txt")
    ctx.write_segments(msa, pkg.JOIN_BIPARTITE, pb)
    rows = [x.split("\t") for x in open(pb).read().split("\n")[:-1]]
    assert rows[0] == ["SEGMENT", "LB", "RB", "SIZE", "SUBSEQUENCE", "SEQUENCES", "COPIED_FROM"]
    assert len(rows) - 1 == X * len(red)
    for r in rows[1:]:
        s = int(r[0])
        assert (int(r[1]), int(r[2]), int(r[3])) == (int(red["lb"][s]), int(red["rb"][s]), int(red["segment_size"][s]))
        if r[6] == "-":
            ids = [int(x) for x in r[5].split(",")]
            assert ids == sorted(ids) and all(bytes(msa[i, int(r[1]):int(r[2])]).decode() == r[4] for i in ids)
        else:
            assert r[5] == ""
    pr = str(tmp_path / "segments_rnd.txt")
    ctx.write_segments(msa, pkg.JOIN_RANDOM, pr)
    rows = [x.split("\t") for x in open(pr).read().split("\n")[:-1]]
    assert rows[0] == ["SEGMENT", "LB", "RB", "SIZE", "SUBSEQUENCE_NUMBER", "COPY_NUMBER", "SUBSEQUENCE"]
    per_seg = Counter(int(r[0]) for r in rows[1:])
    assert all(per_seg[s] == red["segment_size"][s] for s in range(len(red)))
    assert all(sum(int(r[5]) for r in rows[1:] if int(r[0]) == s) == X for s in range(len(red)))
    pg = str(tmp_path / "segments_greedy.txt")
    ctx.write_segments(msa, pkg.JOIN_GREEDY, pg)
    assert open(pg).read() == "SEGMENT\tLB\tRB\tSIZE\tSUBSEQUENCE_NUMBER\tCOPY_NUMBER\tSUBSEQUENCE\n"


@pytest.mark.gpu
def test_config_c2_full_size_founders_match_the_oracle():
    """BASELINE config C2 at full size (m = 2,500 x n = 100,000, L = 50, greedy joining): the founders the library
    writes are byte for byte what the oracle's segmentation + the independent greedy oracle give (SHA-256 of the file)."""
    import hashlib
    import importlib
    import os
    import tempfile
    pkg = importlib.import_module("founder-sequences_amd")
    c = fso.CONFIGS["C2"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    ctx.run()
    msa = np.ascontiguousarray(ctx.get_sequences())
    perm = ctx.join_greedy()
    with tempfile.NamedTemporaryFile(delete=False) as f:
        path = f.name
    ctx.write_founders(msa, perm, path)
    got = hashlib.sha256(open(path, "rb").read()).hexdigest()
    ctx.write_founders_device(perm, path)                      # [r5] from the resident alignment: the same file
    assert hashlib.sha256(open(path, "rb").read()).hexdigest() == got
    os.unlink(path)
    jp = ctx.join_profile()
    # (the class tables and the co-occurrence edges come from the device: a fraction of the S' x m x 8 bytes of states)
    assert 0 < jp["bytes_d2h"] < ctx.result.segment_count * m * 8 // 4 and jp["ms_total"] >= jp["ms_d2h"] > 0
    ref = fso.segment_long(msa, L, threads=8)
    segs = [(int(x["lb"]), int(x["rb"])) for x in ref["reduced"]]
    operm = go.greedy_match(m, ref["max_segment_size"], segs, ref["a"], ref["d"])
    want = hashlib.sha256(b"".join(x + b"\n" for x in go.founders(msa, segs, operm, ref["max_segment_size"]))).hexdigest()
    assert got == want


@pytest.mark.gpu
def test_config_c3_full_size_bipartite_joiner_against_independent_checks(tmp_path):
    """BASELINE config C3 -- the configuration that names --segment-joining=bipartite-matching -- at full size (m = 2,504 x
    n = 1,000,000, L = 100) through the library's own path (device segmentation, fseq_join_bipartite), checked against things
    the library did not compute: for 240 adjacent segment pairs spread over the alignment the total |rows_l n rows_r| its
    chained permutations realise equals the OPTIMAL weight of the pair's assignment problem by scipy's solver on classes
    rebuilt from the boundary states with oracle/join_oracle.py (merge_segments_task.cc:62-67,103-131); every slot of every
    sampled segment shows a class representative and the copies follow create_segment_texts_task.cc:43-75's multiset; and
    the --output-segments file's (SEGMENT, LB, RB, SIZE, row lists) of the sampled segments are the oracle's classes
    (segmentation_dp_arg.cc:59-111).  Lemon's tie order among equal-weight matchings is parity-unpinned, so the
    checks are order-free."""
    import importlib
    from collections import Counter
    pkg = importlib.import_module("founder-sequences_amd")
    c = fso.CONFIGS["C3"]
    m, n, L = c["m"], c["n"], c["L"]
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.generate_synthetic(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
    res = ctx.run()
    X, S = int(res.max_segment_size), int(res.segment_count)
    red = ctx.reduced_traceback()
    perm = ctx.join_bipartite()
    assert perm.shape == (S, X)
    rng = np.random.default_rng(5)
    pairs = sorted(set(int(x) for x in rng.integers(1, S, size=260)) | {1, S - 1})[:240]
    assert len(pairs) >= 200
    cache = {}

    def seg(s):
        if s not in cache:
            a, d = ctx.boundary_state(s)
            cl = jo.classes(m, int(red["lb"][s]), a, d)
            assert len(cl) == int(red["segment_size"][s])
            sl = jo.slot_classes(perm[s], cl)                      # every slot shows a class representative (its smallest row)
            cnt = Counter(sl)
            assert set(cnt) == set(range(len(cl)))
            assert Counter((len(cl[i]), k) for i, k in cnt.items()) == jo.bipartite_copy_multiset(m, X, cl)
            cache[s] = (cl, sl)
        return cache[s]

    for s in pairs:
        cl_l, sl_l = seg(s - 1)
        cl_r, sl_r = seg(s)
        best, base = jo.optimal_weight(sl_l, cl_l, sl_r, cl_r)
        realised = sum(int(base[l, r]) for l, r in zip(sl_l, sl_r))       # founder i keeps slot i across the boundary
        assert realised == best, s
        cache.pop(s - 2, None)
    # the segments file of the same run: the sampled segments' lines against the oracle's classes
    msa = np.ascontiguousarray(ctx.get_sequences())
    path = str(tmp_path / "segments_c3.txt")
    ctx.write_segments(msa, pkg.JOIN_BIPARTITE, path)
    want = {}
    for s in pairs[:40]:
        a, d = ctx.boundary_state(s)
        want[s] = sorted(sorted(cl) for cl in jo.classes(m, int(red["lb"][s]), a, d))
    got = {s: [] for s in want}
    nlines = 0
    with open(path) as f:
        header = f.readline().rstrip("\n").split("\t")
        assert header == ["SEGMENT", "LB", "RB", "SIZE", "SUBSEQUENCE", "SEQUENCES", "COPIED_FROM"]
        for line in f:
            nlines += 1
            r = line.rstrip("\n").split("\t")
            s = int(r[0])
            if s in got:
                assert (int(r[1]), int(r[2]), int(r[3])) == (int(red["lb"][s]), int(red["rb"][s]), int(red["segment_size"][s]))
                if r[6] == "-":
                    ids = [int(x) for x in r[5].split(",")]
                    assert bytes(msa[ids[0], int(r[1]):int(r[2])]).decode() == r[4]
                    got[s].append(ids)
    assert nlines == X * S
    for s in want:
        assert sorted(got[s]) == want[s], s


@pytest.mark.gpu
@pytest.mark.parametrize("kind,m,n,L", [(0, 300, 2000, 25), (1, 700, 1500, 20), (0, 12000, 900, 10)])
def test_device_founders_writer_matches_the_host_writer(pkg, tmp_path, kind, m, n, L):
    """fseq_write_founders_device (k_founders: the lines from the resident packed alignment, 2 / 4 bits per symbol, LDS-resident and
    streamed row counts, a slot without a row printing '-') against fseq_write_founders from the raw host rows."""
    msa = np.ascontiguousarray(fso.synth_msa(fso.synth_spec(7 + kind, 8, 300, 2e-3, kind), m, n))
    ctx = pkg.SegmentationContext(m, n, L)
    ctx.set_sequences(msa)
    ctx.run()
    perm = ctx.join_greedy()
    a, b = str(tmp_path / "h.txt"), str(tmp_path / "d.txt")
    ctx.write_founders(msa, perm, a)
    ctx.write_founders_device(perm, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    # a slot without a row (join_context.cc:348-349: m_permutation_max): '-' over the segment
    perm2 = perm.copy()
    perm2[0, 0] = (1 << int(m).bit_length()) - 1
    ctx.write_founders_device(perm2, b)
    red = ctx.reduced_traceback()
    first = open(b, "rb").read().split(b"\n")[0]
    assert first[:int(red["rb"][0])] == b"-" * int(red["rb"][0]) and first[int(red["rb"][0]):] == open(a, "rb").read().split(b"\n")[0][int(red["rb"][0]):]
