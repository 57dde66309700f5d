"""ctypes binding for the CPU restatement in oracle/fseq_oracle.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.  PARITY UNPINNED (see fseq_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class DpArg(C.Structure):
    _fields_ = [("lb", C.c_uint64), ("rb", C.c_uint64),
                ("segment_max_size", C.c_uint32), ("segment_size", C.c_uint32)]


DP_DTYPE = np.dtype([("lb", "<u8"), ("rb", "<u8"), ("segment_max_size", "<u4"), ("segment_size", "<u4")])


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int),
        ("max_segment_size", C.c_uint32),
        ("n_dp_segments", C.c_uint64),
        ("traceback", C.POINTER(DpArg)),
        ("n_segments", C.c_uint64),
        ("reduced", C.POINTER(DpArg)),
        ("a", C.POINTER(C.c_uint32)),
        ("d", C.POINTER(C.c_uint32)),
        ("dp_size", C.c_uint64),
        ("dp", C.POINTER(DpArg)),
        ("n_samples", C.c_uint64),
        ("pass2_cells", C.c_uint64),
        ("dp_pairs_total", C.c_uint64),
        ("t_pass1", C.c_double), ("t_traceback", C.c_double), ("t_pass2", C.c_double), ("t_merge", C.c_double),
        ("pass2_threads", C.c_int),
    ]


class SynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_founders", C.c_uint32), ("block_len", C.c_uint32),
                ("mut_threshold", C.c_uint64), ("kind", C.c_uint32)]


def build(debug=False):
    target = "libfseq_oracle_dbg.so" if debug else "libfseq_oracle.so"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)
    return os.path.join(_HERE, target)


_libs = {}


def lib(debug=False):
    if debug in _libs:
        return _libs[debug]
    path = os.path.join(_HERE, "libfseq_oracle_dbg.so" if debug else "libfseq_oracle.so")
    src = os.path.join(_HERE, "fseq_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        path = build(debug)
    L = C.CDLL(path)
    vp, u8p, u32p, u64, sz = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.c_uint64, C.c_size_t
    L.fso_rmq_new.restype = vp
    L.fso_rmq_new.argtypes = [vp, sz, sz, C.c_uint]
    L.fso_rmq_free.argtypes = [vp]
    L.fso_rmq_update.argtypes = [vp, sz]
    L.fso_rmq_query.restype = sz
    L.fso_rmq_query.argtypes = [vp, sz, sz]
    L.fso_dp_step.argtypes = [vp, vp, sz, vp, vp, u64, u64, u64, u64, C.POINTER(DpArg)]
    L.fso_pbwt_new.restype = vp
    L.fso_pbwt_new.argtypes = [vp, sz, sz, C.c_uint32, u64, C.c_int]
    L.fso_pbwt_free.argtypes = [vp]
    L.fso_pbwt_prepare.argtypes = [vp]
    L.fso_pbwt_set_state.argtypes = [vp, vp, vp, u64]
    L.fso_pbwt_step.argtypes = [vp]
    L.fso_pbwt_idx.restype = u64
    L.fso_pbwt_idx.argtypes = [vp]
    L.fso_pbwt_a.restype = u32p
    L.fso_pbwt_a.argtypes = [vp]
    L.fso_pbwt_d.restype = u32p
    L.fso_pbwt_d.argtypes = [vp]
    L.fso_pbwt_counts.restype = sz
    L.fso_pbwt_counts.argtypes = [vp, vp, vp]
    L.fso_pbwt_unique_substring_count_lhs.restype = C.c_uint32
    L.fso_pbwt_unique_substring_count_lhs.argtypes = [vp, u64]
    L.fso_pbwt_unique_substring_count_idxs_lhs.restype = sz
    L.fso_pbwt_unique_substring_count_idxs_lhs.argtypes = [vp, u64, vp, vp]
    L.fso_segment_long.restype = C.c_int
    L.fso_segment_long.argtypes = [vp, sz, sz, C.c_uint32, u64, u64, u64, C.c_int, C.c_int, C.POINTER(Result)]
    L.fso_result_free.argtypes = [C.POINTER(Result)]
    L.fso_segment_short.restype = sz
    L.fso_segment_short.argtypes = [vp, sz, sz, C.c_uint32, u64, vp, vp]
    L.fso_synth_threshold.restype = u64
    L.fso_synth_threshold.argtypes = [C.c_double]
    L.fso_synth_byte.restype = C.c_uint8
    L.fso_synth_byte.argtypes = [C.POINTER(SynthSpec), u64, u64]
    L.fso_synth_fill.argtypes = [C.POINTER(SynthSpec), C.c_uint32, u64, u64, vp, sz, sz]
    _libs[debug] = L
    return L


def _strides(msa):
    """msa: 2-D uint8 array indexed [row, col] (any memory order)."""
    assert msa.dtype == np.uint8 and msa.ndim == 2
    return msa.ctypes.data, msa.strides[0], msa.strides[1]


class Rmq:
    """rmq.hh restatement over a uint32 key array (kept alive by this object)."""

    def __init__(self, keys, block_size=64, debug=True):
        self.L = lib(debug)
        self.keys = np.ascontiguousarray(keys, dtype=np.uint32)
        self.h = self.L.fso_rmq_new(self.keys.ctypes.data, 4, len(self.keys), block_size)

    def update(self, last_idx):
        self.L.fso_rmq_update(self.h, last_idx)

    def query(self, beg, end):
        return self.L.fso_rmq_query(self.h, beg, end)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.fso_rmq_free(self.h)
            self.h = None


class Pbwt:
    def __init__(self, msa, with_counts=True, debug=True, col0=0):
        """col0: msa holds the columns [col0, col0 + msa.shape[1]) of a longer alignment (a replay from a state set
        with set_state(a, d, idx >= col0): the oracle only reads columns >= idx)."""
        self.L = lib(debug)
        self.msa = msa
        base, rs, cs = _strides(msa)
        self.m, self.n = msa.shape[0], col0 + msa.shape[1]
        self.h = self.L.fso_pbwt_new(base - col0 * cs, rs, cs, self.m, self.n, 1 if with_counts else 0)
        self.L.fso_pbwt_prepare(self.h)

    def set_state(self, a, d, idx):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        d = np.ascontiguousarray(d, dtype=np.uint32)
        self.L.fso_pbwt_set_state(self.h, a.ctypes.data, d.ctypes.data, idx)

    def step(self):
        self.L.fso_pbwt_step(self.h)

    @property
    def idx(self):
        return self.L.fso_pbwt_idx(self.h)

    @property
    def a(self):
        return np.ctypeslib.as_array(self.L.fso_pbwt_a(self.h), shape=(self.m,)).copy()

    @property
    def d(self):
        return np.ctypeslib.as_array(self.L.fso_pbwt_d(self.h), shape=(self.m,)).copy()

    def counts(self):
        v = np.zeros(self.m + 1, dtype=np.uint32)
        c = np.zeros(self.m + 1, dtype=np.uint32)
        k = self.L.fso_pbwt_counts(self.h, v.ctypes.data, c.ctypes.data)
        return v[:k].copy(), c[:k].copy()

    def unique_substring_count_lhs(self, lb):
        return self.L.fso_pbwt_unique_substring_count_lhs(self.h, lb)

    def unique_substring_count_idxs_lhs(self, lb):
        f = np.zeros(self.m, dtype=np.uint32)
        r = np.zeros(self.m, dtype=np.uint32)
        k = self.L.fso_pbwt_unique_substring_count_idxs_lhs(self.h, lb, f.ctypes.data, r.ctypes.data)
        return f[:k].copy(), r[:k].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.fso_pbwt_free(self.h)
            self.h = None


def dp_step(values, counts, dp, rmq_handle, m, L, lb, text_pos, init, debug=True):
    """dp: structured array DP_DTYPE; rmq_handle built over dp['segment_max_size'] by caller."""
    lib_ = lib(debug)
    values = np.ascontiguousarray(values, dtype=np.uint32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    arg = DpArg(*init)
    lib_.fso_dp_step(values.ctypes.data, counts.ctypes.data, len(values), dp.ctypes.data, rmq_handle,
                     m, L, lb, text_pos, C.byref(arg))
    return (arg.lb, arg.rb, arg.segment_max_size, arg.segment_size)


def sample_rate_for(n, q=4):
    """generate_context.cc:113-124: ceil(q*sqrt(n)); q == 0 -> n + 1."""
    import math
    if q == 0:
        return n + 1
    return int(math.ceil(q * math.sqrt(n)))


def segment_long(msa, L, sample_rate=None, threads=1, keep_dp=False, debug=False):
    """Runs the long path.  Returns a dict of numpy arrays (copied out of the C result)."""
    lib_ = lib(debug)
    base, rs, cs = _strides(msa)
    m, n = msa.shape
    if sample_rate is None:
        sample_rate = sample_rate_for(n)
    res = Result()
    rc = lib_.fso_segment_long(base, rs, cs, m, n, L, sample_rate, threads, 1 if keep_dp else 0, C.byref(res))
    if rc < 0:
        raise ValueError("fso_segment_long: bad arguments")

    def dparr(ptr, k):
        if not ptr or k == 0:
            return np.zeros(0, dtype=DP_DTYPE)
        buf = (DpArg * k).from_address(C.addressof(ptr.contents))
        return np.frombuffer(buf, dtype=DP_DTYPE).copy()

    out = {
        "status": res.status,
        "max_segment_size": res.max_segment_size,
        "traceback": dparr(res.traceback, res.n_dp_segments),
        "reduced": dparr(res.reduced, res.n_segments),
        "n_samples": res.n_samples,
        "pass2_cells": res.pass2_cells,
        "dp_pairs_total": res.dp_pairs_total,
        "t_pass1": res.t_pass1, "t_traceback": res.t_traceback, "t_pass2": res.t_pass2, "t_merge": res.t_merge,
        "pass2_threads": res.pass2_threads,
    }
    if res.n_segments:
        S = res.n_segments
        out["a"] = np.ctypeslib.as_array(res.a, shape=(S, m)).copy()
        out["d"] = np.ctypeslib.as_array(res.d, shape=(S, m)).copy()
    else:
        out["a"] = np.zeros((0, m), dtype=np.uint32)
        out["d"] = np.zeros((0, m), dtype=np.uint32)
    if keep_dp:
        out["dp"] = dparr(res.dp, res.dp_size)
    lib_.fso_result_free(C.byref(res))
    return out


def segment_short(msa, debug=False):
    lib_ = lib(debug)
    base, rs, cs = _strides(msa)
    m, n = msa.shape
    f = np.zeros(m, dtype=np.uint32)
    r = np.zeros(m, dtype=np.uint32)
    k = lib_.fso_segment_short(base, rs, cs, m, n, f.ctypes.data, r.ctypes.data)
    return f[:k].copy(), r[:k].copy()


def synth_spec(seed, n_founders, block_len, mu, kind=0):
    L = lib()
    return SynthSpec(seed, n_founders, block_len, L.fso_synth_threshold(mu), kind)


def synth_msa(spec, m, n, c0=0, order="F"):
    """Returns the raw-byte MSA [m, n] for columns [c0, c0+n); order 'F' = column-major."""
    L = lib()
    out = np.zeros((m, n), dtype=np.uint8, order=order)
    L.fso_synth_fill(C.byref(spec), m, c0, c0 + n, out.ctypes.data, out.strides[0], out.strides[1])
    return out


# The five BASELINE.json configurations: generator parameters.
CONFIGS = {
    "C1": dict(m=8, n=1000, L=10, K=3, B=100, mu=5e-3, seed=0x5EED0001, kind=0),
    "C2": dict(m=2500, n=100000, L=50, K=16, B=2000, mu=1e-4, seed=0x5EED0002, kind=0),
    "C3": dict(m=2504, n=1000000, L=100, K=24, B=5000, mu=1e-4, seed=0x5EED0003, kind=0),
    "C4": dict(m=100000, n=5000000, L=200, K=64, B=10000, mu=5e-5, seed=0x5EED0004, kind=0),
    "C5": dict(m=10000, n=1000000, L=100, K=32, B=5000, mu=1e-4, seed=0x5EED0005, kind=1),
}


def config_spec(name):
    c = CONFIGS[name]
    return synth_spec(c["seed"], c["K"], c["B"], c["mu"], c["kind"])
