"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/fseq.h declares.  No compute is called here (no GPU in this container)."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    build = importlib.import_module("founder-sequences_amd.build")
    build.build()
    return importlib.import_module("founder-sequences_amd")


def _declared_symbols(header="fseq.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fseq_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    """include/fseq.h is the drop-in boundary, include/fseq_debug.h the intermediate state for tests: the binding names
    every symbol of both, and nothing of the debug header is declared in the boundary header."""
    assert sorted(_declared_symbols() + _declared_symbols("fseq_debug.h")) == sorted(pkg.EXPORTS)
    assert _declared_symbols("fseq_debug.h") == sorted(pkg.DEBUG_EXPORTS)
    assert not [s for s in _declared_symbols() if s.startswith("fseq_debug")]


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    for name in _declared_symbols() + _declared_symbols("fseq_debug.h"):
        assert hasattr(lib, name), name
    assert lib.fseq_abi_version() == 5
    assert lib.fseq_strerror(pkg.FSEQ_E_PEER).decode().startswith("another rank")
    assert lib.fseq_strerror(2).decode().startswith("unable to reduce")


def test_struct_layouts_match_header(pkg):
    # sizes the C compiler gives the header's structs (natural alignment, x86-64)
    assert C.sizeof(pkg.Params) == 48
    assert C.sizeof(pkg.SynthSpec) == 32
    assert C.sizeof(pkg.Segment) == 24 == pkg.SEGMENT_DTYPE.itemsize
    assert C.sizeof(pkg.DpArg) == 24 == pkg.DPARG_DTYPE.itemsize
    assert C.sizeof(pkg.Result) == 24
    assert C.sizeof(pkg.Timings) == 8 * 8 + 3 * 8 + 12 * 4                # (twelve 32-bit fields: a multiple of the doubles' alignment)


def test_header_compiles_as_plain_c(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "fseq.h"\n#include "fseq_debug.h"\nint main(void){ fseq_params p; (void)p; return sizeof(fseq_segment) == 24 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_bad_arguments_fail_without_touching_a_device(pkg):
    lib = pkg.load_library()
    h = C.c_void_p()
    assert lib.fseq_create(None, C.byref(h)) == pkg.FSEQ_E_ARG
    p = pkg.Params(0, 10, 2, 0, 0, 0, 0)
    assert lib.fseq_create(C.byref(p), C.byref(h)) == pkg.FSEQ_E_ARG
    assert lib.fseq_last_error(None).decode() == "null context"


def test_dp_schedule_and_prefix_rounds(pkg):
    """Host logic behind the resumed DP launches (fseq_dp.hpp dp_schedule / dp_rounds_within): cells L .. n-L in
    rounds of RL, a drain round when pipelined, the final cell; a round belongs to the column prefix [0, col_hi)
    iff the lists of all its cells (column end - 1) are inside it."""
    import numpy as np
    rng = np.random.default_rng(0)
    for _ in range(300):
        L = int(rng.integers(1, 300))
        n = int(rng.integers(2 * L, 2 * L + 5000))
        nr, RL, _, pipe = pkg.dp_schedule(L, n, n)
        assert pipe == (L >= 96)
        assert RL == ((min(L // 2, 48) // 12) * 12 if pipe else min(L, 56))
        nreg = (n - 2 * L) // RL + 1
        assert nr == nreg + (2 if pipe else 1)
        # rounds: r < nreg -> cells end = L + r*RL + i, i < min(RL, n - L - e0 + 1); drain: none; final: end = n
        def needs(r):
            if r == nr - 1:
                return n - 1
            if r >= nreg:
                return -1
            e0 = L + r * RL
            return e0 + min(RL, n - L - e0 + 1) - 2
        for col_hi in [0, 1, L, L + 1, n // 3, n // 2, n - L, n - 1, n, n + 5] + [int(x) for x in rng.integers(0, n + 1, size=5)]:
            got = pkg.dp_schedule(L, n, col_hi)[2]
            want = nr if col_hi >= n else next((r for r in range(nr) if needs(r) >= col_hi), nr)
            assert got == want, (L, n, col_hi, got, want)
