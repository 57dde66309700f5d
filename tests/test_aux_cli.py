"""The three auxiliary tools (host only, no GPU): remove_identity_columns,
insert_identity_columns, match_founder_sequences -- against Python restatements of
remove-identity-columns/main.cc, insert-identity-columns/main.cc and
match-sequences-to-founders/match_founder_sequences.cc:163-218, and as a round trip."""
import importlib
import os
import subprocess

import numpy as np
import pytest


@pytest.fixture(scope="module")
def tools():
    b = importlib.import_module("founder-sequences_amd.build")
    return {os.path.basename(p): p for p in b.build_aux()}


def _texts(rng, m, n, identity_frac):
    base = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    t = np.tile(base, (m, 1))
    vary = rng.random(n) >= identity_frac
    for c in np.nonzero(vary)[0]:
        t[:, c] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=m)
        if len(set(t[:, c].tolist())) == 1:
            t[0, c] = ord("A") if t[0, c] != ord("A") else ord("C")
    return t


def test_remove_identity_columns_with_more_texts_than_descriptors(tools, tmp_path):
    """Thousands of haplotype files must not need thousands of open descriptors (the reference reopens every file
    per 32 KiB chunk, remove-identity-columns/main.cc fill_buffers; round-1 advice): 200 texts under a limit of 64."""
    import resource
    rng = np.random.default_rng(9)
    m, n = 200, 40000
    t = _texts(rng, m, n, 0.5)
    src, work = tmp_path / "src", tmp_path / "work"
    src.mkdir()
    work.mkdir()
    names = []
    for i in range(m):
        (src / ("h%d.txt" % i)).write_bytes(bytes(t[i]))
        names.append(str(src / ("h%d.txt" % i)))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")

    def limit():
        resource.setrlimit(resource.RLIMIT_NOFILE, (64, 64))

    r = subprocess.run([tools["remove_identity_columns"], "-i", str(tmp_path / "list.txt")], cwd=work, capture_output=True, timeout=300, preexec_fn=limit)
    assert r.returncode == 0, r.stderr
    want_mask = "".join("1" if len(set(t[:, c].tolist())) == 1 else "0" for c in range(n))
    assert r.stdout.decode().strip() == want_mask
    keep = np.array([ch == "0" for ch in want_mask])
    for i in (0, 77, m - 1):
        assert (work / ("h%d.txt" % i)).read_bytes() == bytes(t[i][keep])


@pytest.mark.parametrize("m,n", [(5, 300), (12, 70000), (3, 32768)])      # 70000 > two 32 KiB chunks
def test_remove_then_insert_identity_columns_round_trip(tools, tmp_path, m, n):
    rng = np.random.default_rng(n)
    t = _texts(rng, m, n, 0.6)
    src = tmp_path / "src"
    work = tmp_path / "work"
    back = tmp_path / "back"
    for d in (src, work, back):
        d.mkdir()
    names = []
    for i in range(m):
        (src / ("seq%d.txt" % i)).write_bytes(bytes(t[i]))
        names.append(str(src / ("seq%d.txt" % i)))
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    r = subprocess.run([tools["remove_identity_columns"], "-i", str(tmp_path / "list.txt")], cwd=work, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    mask = r.stdout.decode().strip()
    want_mask = "".join("1" if len(set(t[:, c].tolist())) == 1 else "0" for c in range(n))
    assert mask == want_mask
    keep = np.array([ch == "0" for ch in want_mask])
    for i in range(m):
        assert (work / ("seq%d.txt" % i)).read_bytes() == bytes(t[i][keep])
    # a second run must refuse to overwrite, --overwrite must not
    r2 = subprocess.run([tools["remove_identity_columns"], "-i", str(tmp_path / "list.txt")], cwd=work, capture_output=True, timeout=120)
    assert r2.returncode != 0
    r3 = subprocess.run([tools["remove_identity_columns"], "--overwrite"], cwd=work, input=("\n".join(names) + "\n").encode(), capture_output=True, timeout=120)
    assert r3.returncode == 0 and r3.stdout.decode().strip() == want_mask
    # put the columns back: list-file form (outputs named after the inputs) and text form (outputs 1 .. m)
    (tmp_path / "mask.txt").write_text(mask + "\n")
    reduced = [str(work / ("seq%d.txt" % i)) for i in range(m)]
    (tmp_path / "reduced_list.txt").write_text("\n".join(reduced) + "\n")
    r = subprocess.run([tools["insert_identity_columns"], "-i", str(tmp_path / "reduced_list.txt"), "-f", "list-file", "-r", names[0],
                        "-d", str(tmp_path / "mask.txt")], cwd=back, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for i in range(m):
        assert (back / ("seq%d.txt" % i)).read_bytes() == bytes(t[i])
    text = tmp_path / "founders.txt"
    text.write_bytes(b"".join(bytes(t[i][keep]) + b"\n" for i in range(m)))
    back2 = tmp_path / "back2"
    back2.mkdir()
    r = subprocess.run([tools["insert_identity_columns"], "--input", str(text), "--reference", names[1], "--identity-columns",
                        str(tmp_path / "mask.txt")], cwd=back2, capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for i in range(m):
        assert (back2 / str(i + 1)).read_bytes() == bytes(t[i])


def _match_oracle(seq, founders, min_len):
    """match_context::match_sequence_and_report, restated."""
    out, errs = [], 0
    cur = list(range(len(founders)))
    lb, count, chr_idx = 0, len(cur), 0
    for c in seq:
        recheck = False
        if min_len and min_len <= chr_idx - lb:
            recheck = True
        else:
            dst = [f for f in cur if founders[f][chr_idx] == c]
            if not dst:
                recheck = True
        if recheck:
            out.append((lb, chr_idx, list(cur)))
            lb = chr_idx
            cur = list(range(len(founders)))
            dst = [f for f in cur if founders[f][chr_idx] == c]
            if not dst:
                errs += 1
        count, cur = len(dst), dst
        chr_idx += 1
    if count:
        out.append((lb, chr_idx, list(cur)))
    return out, errs


@pytest.mark.parametrize("min_len,fmt", [(0, "list-file"), (0, "text"), (7, "FASTA")])
def test_match_founder_sequences(tools, tmp_path, min_len, fmt):
    rng = np.random.default_rng(11 + min_len)
    n, k, m = 400, 5, 9
    founders = [bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)) for _ in range(k)]
    seqs = []
    for _ in range(m):                                          # mosaics of the founders, plus one foreign character
        s = bytearray()
        while len(s) < n:
            f = founders[int(rng.integers(k))]
            step = int(rng.integers(20, 90))
            s += f[len(s):len(s) + step]
        seqs.append(bytes(s[:n]))
    seqs[3] = seqs[3][:100] + b"N" + seqs[3][101:]
    paths = []
    for i, s in enumerate(seqs):
        (tmp_path / ("s%d.txt" % i)).write_bytes(s)
        paths.append(str(tmp_path / ("s%d.txt" % i)))
    (tmp_path / "seqs.txt").write_text("\n".join(paths) + "\n")
    if fmt == "list-file":
        fp = []
        for i, f in enumerate(founders):
            (tmp_path / ("f%d.txt" % i)).write_bytes(f)
            fp.append(str(tmp_path / ("f%d.txt" % i)))
        (tmp_path / "founders.in").write_text("\n".join(fp) + "\n")
    elif fmt == "text":
        (tmp_path / "founders.in").write_bytes(b"".join(f + b"\n" for f in founders))
    else:
        (tmp_path / "founders.in").write_bytes(b"".join(b">f%d\n" % i + f[:150] + b"\n" + f[150:] + b"\n" for i, f in enumerate(founders)))
    cmd = [tools["match_founder_sequences"], "--sequences", str(tmp_path / "seqs.txt"), "--founders", str(tmp_path / "founders.in"),
           "--founders-format", fmt, "--min-segment-length", str(min_len)]
    r = subprocess.run(cmd + ["--single-threaded"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # one task per sequence on the host's threads (match_founder_sequences.cc:218-252): the same report, in input order
    rt = subprocess.run(cmd, capture_output=True, timeout=120)
    assert rt.returncode == 0 and rt.stdout == r.stdout and rt.stderr == r.stderr
    lines = r.stdout.decode().split("\n")
    assert lines[0] == "SEQUENCE_INDEX\tLB\tRB\tFOUNDER_INDICES" and lines[-1] == ""
    want, errs = [], 0
    for i, s in enumerate(seqs):
        o, e = _match_oracle(s, founders, min_len)
        errs += e
        want += ["%d\t%d\t%d\t%s" % (i, lb, rb, ",".join(map(str, idx))) for lb, rb, idx in o]
    assert lines[1:-1] == want
    assert r.stderr.decode().count("not found in the founders") == errs >= 1
