"""founder-sequences_amd: MI355X-native segmentation engine (host-side Python mirror of the C ABI).

The compute lives in libfseq_hip.so (hand-written HIP for gfx950, built in-tree by build.py) behind
the C ABI declared in include/fseq.h.  This module is a thin ctypes mirror of that ABI, named after
the reference's interface for the path (segmentation_lp_context / segmentation_container,
include/founder_sequences/segmentation_lp_context.hh:45-121, segmentation_container.hh:15-20).

There is NO CPU fallback: if the library is missing, or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfseq_hip.so")

FSEQ_OK, FSEQ_E_ARG, FSEQ_E_NO_REDUCTION, FSEQ_E_HIP, FSEQ_E_OOM, FSEQ_E_UNSUPPORTED = range(6)


class FseqError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fseq error %d: %s" % (code, msg))
        self.code = code


class NoReduction(FseqError):
    """generate_context.cc:192-200: max segment size equals the number of input sequences."""


class Params(C.Structure):
    _fields_ = [("m", C.c_uint32), ("n", C.c_uint64), ("segment_length", C.c_uint64),
                ("pbwt_sample_rate", C.c_uint64), ("block_len", C.c_uint32), ("list_cap", C.c_uint32),
                ("device", C.c_int32)]


class SynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_founders", C.c_uint32), ("block_len", C.c_uint32),
                ("mut_threshold", C.c_uint64), ("kind", C.c_uint32)]


class Segment(C.Structure):
    _fields_ = [("lb", C.c_uint64), ("rb", C.c_uint64), ("segment_size", C.c_uint32), ("reserved", C.c_uint32)]


class DpArg(C.Structure):
    _fields_ = [("lb", C.c_uint64), ("rb", C.c_uint64), ("segment_max_size", C.c_uint32), ("segment_size", C.c_uint32)]


class Result(C.Structure):
    _fields_ = [("max_segment_size", C.c_uint32), ("short_path", C.c_uint32),
                ("dp_segment_count", C.c_uint64), ("segment_count", C.c_uint64)]


class Timings(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_phase_a", C.c_double), ("ms_phase_b", C.c_double),
                ("ms_phase_c", C.c_double), ("ms_dp", C.c_double), ("ms_pass2", C.c_double), ("ms_host", C.c_double),
                ("ms_colstep_kernels", C.c_double), ("colstep_launches", C.c_uint64), ("colstep_cells", C.c_uint64),
                ("pass2_cells", C.c_uint64), ("list_cap_used", C.c_uint32), ("retries", C.c_uint32),
                ("block_len", C.c_uint32), ("n_blocks", C.c_uint32),
                ("dp_chunks", C.c_uint32), ("dp_sweeps", C.c_uint32), ("phase_a_fallbacks", C.c_uint32), ("phase_a_given_up", C.c_uint32), ("phase_a_trie_given_up", C.c_uint32),
                ("reduced_blocks", C.c_uint32), ("reduced_rows_mean", C.c_uint32), ("reduced_redone", C.c_uint32)]


class JoinProfile(C.Structure):
    _fields_ = [("ms_d2h", C.c_double), ("ms_classes", C.c_double), ("ms_edges", C.c_double), ("ms_draw", C.c_double), ("ms_total", C.c_double),
                ("bytes_d2h", C.c_uint64)]


SEGMENT_DTYPE = np.dtype([("lb", "<u8"), ("rb", "<u8"), ("segment_size", "<u4"), ("reserved", "<u4")])
DPARG_DTYPE = np.dtype([("lb", "<u8"), ("rb", "<u8"), ("segment_max_size", "<u4"), ("segment_size", "<u4")])

# every symbol include/fseq.h declares
EXPORTS = [
    "fseq_abi_version", "fseq_strerror", "fseq_create", "fseq_destroy", "fseq_last_error",
    "fseq_set_rows", "fseq_set_matrix", "fseq_set_device_columns", "fseq_generate_synthetic", "fseq_get_matrix",
    "fseq_run_segmentation", "fseq_get_traceback", "fseq_get_segments", "fseq_boundary_state",
    "fseq_short_path_runs", "fseq_join_greedy", "fseq_greedy_match_host", "fseq_write_founders", "fseq_write_founders_device", "fseq_debug_dp", "fseq_debug_block_state", "fseq_debug_column_list", "fseq_get_timings",
    "fseq_rowshard_xbuf_words", "fseq_rowshard_rows", "fseq_rowshard_pbwt",
    "fseq_debug_rmq", "fseq_shard_xbuf_words", "fseq_set_shard", "fseq_shard_columns", "fseq_shard_owner",
    "fseq_set_device_columns_packed", "fseq_debug_dp_schedule", "fseq_run_segmentation_batch", "fseq_join_bipartite", "fseq_join_random", "fseq_bipartite_match_host", "fseq_random_join_host", "fseq_write_segments",
    "fseq_set_progress", "fseq_step_max", "fseq_current_step", "fseq_set_memory_budget", "fseq_write_segments_host", "fseq_get_join_profile", "fseq_debug_set_tuning",
    "fseq_shard_abort", "fseq_debug_dp_owned", "fseq_debug_clock", "fseq_debug_ranges",
]
# ... of which include/fseq_debug.h declares these (intermediate state for tests, not part of the drop-in boundary)
DEBUG_EXPORTS = ["fseq_debug_dp", "fseq_debug_dp_owned", "fseq_debug_clock", "fseq_debug_ranges", "fseq_debug_block_state", "fseq_debug_column_list", "fseq_debug_rmq", "fseq_debug_dp_schedule", "fseq_debug_set_tuning"]

FSEQ_E_PEER = 6
STAGE_TRACEBACK, STAGE_MERGE, STAGE_SAMPLES = 0, 1, 2
# fseq_progress_fn (include/fseq.h): (user, stage, current_step, step_max)
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64)

JOIN_GREEDY, JOIN_BIPARTITE, JOIN_RANDOM = 0, 1, 2

# fseq_allreduce_fn (include/fseq.h): all-reduce xbuf[offset .. offset + count) over the ranks, op 0 = sum, 1 = max
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int)

_lib = None


def load_library():
    """Loads libfseq_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback for the segmentation path" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, sz = C.c_void_p, C.c_uint64, C.c_size_t
    L.fseq_abi_version.restype = C.c_uint32
    L.fseq_strerror.restype = C.c_char_p
    L.fseq_strerror.argtypes = [C.c_int]
    L.fseq_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.fseq_destroy.argtypes = [vp]
    L.fseq_destroy.restype = None
    L.fseq_last_error.restype = C.c_char_p
    L.fseq_last_error.argtypes = [vp]
    L.fseq_set_rows.argtypes = [vp, C.POINTER(vp)]
    L.fseq_set_matrix.argtypes = [vp, vp, sz, sz]
    L.fseq_set_device_columns.argtypes = [vp, vp, sz, C.c_uint32]
    L.fseq_set_device_columns_packed.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint32]
    L.fseq_generate_synthetic.argtypes = [vp, C.POINTER(SynthSpec)]
    L.fseq_get_matrix.argtypes = [vp, u64, u64, vp, sz, sz]
    L.fseq_run_segmentation.argtypes = [vp, C.POINTER(Result)]
    L.fseq_get_traceback.argtypes = [vp, vp]
    L.fseq_get_segments.argtypes = [vp, vp]
    L.fseq_boundary_state.argtypes = [vp, u64, vp, vp]
    L.fseq_short_path_runs.argtypes = [vp, vp, vp]
    L.fseq_join_greedy.argtypes = [vp, vp]
    L.fseq_greedy_match_host.argtypes = [C.c_uint32, C.c_uint32, u64, vp, vp, vp, vp, vp]
    L.fseq_write_founders.argtypes = [vp, C.POINTER(vp), vp, C.c_char_p]
    L.fseq_write_founders_device.argtypes = [vp, vp, C.c_char_p]
    L.fseq_run_segmentation_batch.argtypes = [vp, sz, vp, vp]
    L.fseq_debug_dp_schedule.argtypes = [u64, u64, u64, vp, vp, vp, vp]
    L.fseq_join_bipartite.argtypes = [vp, vp]
    L.fseq_join_random.argtypes = [vp, C.c_uint32, vp]
    L.fseq_bipartite_match_host.argtypes = [C.c_uint32, C.c_uint32, u64, vp, vp, vp, vp, vp, vp]
    L.fseq_random_join_host.argtypes = [C.c_uint32, C.c_uint32, u64, vp, vp, vp, vp, C.c_uint32, vp]
    L.fseq_write_segments.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_char_p]
    L.fseq_debug_dp.argtypes = [vp, vp, vp, vp]
    L.fseq_debug_block_state.argtypes = [vp, u64, vp, vp]
    L.fseq_debug_column_list.argtypes = [vp, u64, vp, vp, vp, vp, vp]
    L.fseq_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.fseq_debug_rmq.argtypes = [C.c_int, vp, C.c_uint32, vp, vp, C.c_uint32, vp, vp]
    L.fseq_shard_xbuf_words.restype = u64
    L.fseq_shard_xbuf_words.argtypes = [vp, C.c_uint32]
    L.fseq_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32, vp, u64, ALLREDUCE_FN, vp]
    L.fseq_shard_columns.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.fseq_shard_owner.argtypes = [vp, u64, C.POINTER(C.c_uint32)]
    L.fseq_rowshard_xbuf_words.restype = u64
    L.fseq_rowshard_xbuf_words.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.fseq_rowshard_rows.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.fseq_rowshard_pbwt.argtypes = [vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(u64)]
    L.fseq_write_segments_host.argtypes = [vp, C.POINTER(vp), C.c_int, vp, vp, C.c_char_p]
    L.fseq_get_join_profile.argtypes = [vp, C.POINTER(JoinProfile)]
    L.fseq_debug_set_tuning.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.fseq_set_progress.argtypes = [vp, PROGRESS_FN, vp]
    L.fseq_step_max.restype = u64
    L.fseq_step_max.argtypes = [vp]
    L.fseq_current_step.restype = u64
    L.fseq_current_step.argtypes = [vp]
    L.fseq_set_memory_budget.argtypes = [vp, u64]
    L.fseq_shard_abort.argtypes = [vp, C.c_int]
    L.fseq_debug_dp_owned.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.fseq_debug_clock.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.fseq_debug_ranges.argtypes = [C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_int)]
    _lib = L
    return L


def synth_threshold(mu):
    """mu * 2^64 as the generator's integer threshold (same rounding as the C implementations)."""
    if mu <= 0.0:
        return 0
    if mu >= 1.0:
        return 0xFFFFFFFFFFFFFFFF
    return int(mu * 18446744073709551616.0)


def greedy_match_host(m, max_segment_size, lb, rb, a, d):
    """greedy_matcher::match on caller-supplied boundary states (host only, no GPU needed)."""
    L = load_library()
    lb = np.ascontiguousarray(lb, dtype=np.uint64)
    rb = np.ascontiguousarray(rb, dtype=np.uint64)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    d = np.ascontiguousarray(d, dtype=np.uint32)
    perm = np.zeros((len(lb), max_segment_size), dtype=np.uint32)
    rc = L.fseq_greedy_match_host(m, max_segment_size, len(lb), lb.ctypes.data, rb.ctypes.data, a.ctypes.data, d.ctypes.data, perm.ctypes.data)
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    return perm


def debug_ranges():
    """(pushes, pops, built_with_roctx): the phase ranges this process has opened and closed so far."""
    a, b, w = C.c_uint64(), C.c_uint64(), C.c_int()
    load_library().fseq_debug_ranges(C.byref(a), C.byref(b), C.byref(w))
    return a.value, b.value, bool(w.value)


def debug_rmq(keys, beg, end, device=0):
    """rmq.hh as the device routines restate it, on caller keys: (index by the HBM path, index by the LDS path)."""
    L = load_library()
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    beg = np.ascontiguousarray(beg, dtype=np.uint32)
    end = np.ascontiguousarray(end, dtype=np.uint32)
    a = np.zeros(len(beg), dtype=np.uint32)
    b = np.zeros(len(beg), dtype=np.uint32)
    rc = L.fseq_debug_rmq(device, keys.ctypes.data, len(keys), beg.ctypes.data, end.ctypes.data, len(beg), a.ctypes.data, b.ctypes.data)
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    return a, b


class RowShard(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_uint32), ("world", C.c_uint32), ("m", C.c_uint32), ("sigma", C.c_uint32),
                ("bits", C.c_uint32), ("ncols", C.c_uint64), ("d_cols", C.c_void_p), ("ld", C.c_size_t), ("xbuf", C.c_void_p),
                ("xbuf_words", C.c_uint64), ("fn", ALLREDUCE_FN), ("user", C.c_void_p)]


def pack_columns(msa_codes, bits):
    """Row-major codes (m x n, values < 2**bits) -> column-major packed bytes (n x ld, ld a multiple of 16): the layout
    fseq_set_device_columns_packed and fseq_rowshard_pbwt read (row r of a column in byte r * bits / 8)."""
    m, n = msa_codes.shape
    per = 8 // bits
    ld = ((m + per - 1) // per + 15) // 16 * 16
    cols = np.zeros((n, ld * per), dtype=np.uint8)
    cols[:, :m] = msa_codes.T
    out = np.zeros((n, ld), dtype=np.uint8)
    for j in range(per):
        out |= (cols[:, j::per] << (bits * j)).astype(np.uint8)
    return out, ld


def rowshard_xbuf_words(m, bits, world):
    return int(load_library().fseq_rowshard_xbuf_words(m, bits, world))


def rowshard_rows(m, bits, rank, world):
    lo, hi = C.c_uint32(), C.c_uint32()
    rc = load_library().fseq_rowshard_rows(m, bits, rank, world, C.byref(lo), C.byref(hi))
    if rc != FSEQ_OK:
        raise FseqError(rc, load_library().fseq_strerror(rc).decode())
    return lo.value, hi.value


def rowshard_pbwt(d_cols_ptr, ld, m, sigma, bits, ncols, rank, world, xbuf_ptr, xbuf_words, allreduce=None, device=0):
    """fseq_rowshard_pbwt (the north-star row split, include/fseq.h): returns (a, d, pos_lo, pos_hi, ms, exchanges);
    a / d are m long with this rank's positions [pos_lo, pos_hi) filled in.  allreduce(offset, count, op) -> 0."""
    L = load_library()

    def _cb(_user, off, cnt, op):
        try:
            return int(allreduce(int(off), int(cnt), int(op)) or 0)
        except Exception:                           # an exception must not unwind through the C frame
            import traceback
            traceback.print_exc()
            return 1
    cb = ALLREDUCE_FN(_cb) if allreduce is not None else ALLREDUCE_FN()
    args = RowShard(device, rank, world, m, sigma, bits, ncols, d_cols_ptr, ld, xbuf_ptr, xbuf_words, cb, None)
    a = np.zeros(m, dtype=np.uint32)
    d = np.zeros(m, dtype=np.uint32)
    lo, hi, ms, nex = C.c_uint32(), C.c_uint32(), C.c_double(), C.c_uint64()
    rc = L.fseq_rowshard_pbwt(C.byref(args), a.ctypes.data, d.ctypes.data, C.byref(lo), C.byref(hi), C.byref(ms), C.byref(nex))
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    return a, d, lo.value, hi.value, ms.value, nex.value


def dp_schedule(segment_length, n, col_hi):
    """(rounds, cells per round, leading rounds that only need columns < col_hi, pipelined) of the DP schedule."""
    L = load_library()
    a, b, c_, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_int()
    rc = L.fseq_debug_dp_schedule(segment_length, n, col_hi, C.byref(a), C.byref(b), C.byref(c_), C.byref(d))
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    return a.value, b.value, c_.value, bool(d.value)


def run_batch(contexts):
    """fseq_run_segmentation_batch: all contexts in flight together (one native host thread and stream each).
    Returns the list of per-context return codes; every context's .result is filled in."""
    L = load_library()
    n = len(contexts)
    handles = (C.c_void_p * n)(*[c.h for c in contexts])
    results = (Result * n)()
    rcs = (C.c_int * n)()
    rc = L.fseq_run_segmentation_batch(handles, n, results, rcs)
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    for i, c in enumerate(contexts):
        r = Result()
        C.memmove(C.byref(r), C.byref(results[i]), C.sizeof(Result))
        c.result = r
    return list(rcs)


def bipartite_match_host(m, max_segment_size, lb, rb, a, d):
    """bipartite_matcher::match on caller-supplied boundary states (host only).  Returns (permutations,
    total matching weight between segments s and s+1)."""
    L = load_library()
    lb = np.ascontiguousarray(lb, dtype=np.uint64)
    rb = np.ascontiguousarray(rb, dtype=np.uint64)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    d = np.ascontiguousarray(d, dtype=np.uint32)
    perm = np.zeros((len(lb), max_segment_size), dtype=np.uint32)
    weights = np.zeros(max(1, len(lb) - 1), dtype=np.int64)
    rc = L.fseq_bipartite_match_host(m, max_segment_size, len(lb), lb.ctypes.data, rb.ctypes.data, a.ctypes.data, d.ctypes.data,
                                     perm.ctypes.data, weights.ctypes.data)
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    return perm, weights[:len(lb) - 1]


def random_join_host(m, max_segment_size, lb, rb, a, d, seed):
    """join_context::join_random_order_and_output on caller-supplied boundary states (host only)."""
    L = load_library()
    lb = np.ascontiguousarray(lb, dtype=np.uint64)
    rb = np.ascontiguousarray(rb, dtype=np.uint64)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    d = np.ascontiguousarray(d, dtype=np.uint32)
    perm = np.zeros((len(lb), max_segment_size), dtype=np.uint32)
    rc = L.fseq_random_join_host(m, max_segment_size, len(lb), lb.ctypes.data, rb.ctypes.data, a.ctypes.data, d.ctypes.data, seed, perm.ctypes.data)
    if rc != FSEQ_OK:
        raise FseqError(rc, L.fseq_strerror(rc).decode())
    return perm


class SegmentationContext:
    """Mirror of segmentation_lp_context / segmentation_sp_context behind the C ABI.

    generate_traceback + update_samples_to_traceback_positions + find_segments_greedy
    (segmentation_lp_context.hh:119-121) are one call here: run().  The result fields mirror
    segmentation_container (reduced_traceback, reduced_pbwt_samples via boundary_state(),
    max_segment_size)."""

    def __init__(self, m, n, segment_length, pbwt_sample_rate=0, block_len=0, list_cap=0, device=0):
        self.L = load_library()
        self.m, self.n, self.segment_length = int(m), int(n), int(segment_length)
        p = Params(self.m, self.n, self.segment_length, int(pbwt_sample_rate), int(block_len), int(list_cap), int(device))
        h = C.c_void_p()
        rc = self.L.fseq_create(C.byref(p), C.byref(h))
        if rc != FSEQ_OK:
            raise FseqError(rc, self.L.fseq_strerror(rc).decode())
        self.h = h
        self.result = None
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.fseq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == FSEQ_OK:
            return
        msg = self.L.fseq_last_error(self.h).decode() or self.L.fseq_strerror(rc).decode()
        if rc == FSEQ_E_NO_REDUCTION:
            raise NoReduction(rc, msg)
        raise FseqError(rc, msg)

    # ---- one alignment over several ranks (fseq_set_shard; the transport lives in founder-sequences_amd/dist.py)
    def set_shard(self, rank, world, xbuf_ptr, xbuf_words, allreduce):
        """allreduce(offset_words, count_words, op) -> 0: all-reduces that slice of the exchange buffer in place.
        Call before the input is set; every rank then holds its own columns only."""
        def _cb(_user, off, cnt, op):
            try:
                return int(allreduce(int(off), int(cnt), int(op)) or 0)
            except Exception:                      # never let an exception cross the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._shard_cb = ALLREDUCE_FN(_cb)         # keep the thunk alive as long as the context
        self._check(self.L.fseq_set_shard(self.h, rank, world, xbuf_ptr, xbuf_words, self._shard_cb, None))
        self.rank, self.world = rank, world

    def shard_abort(self, code=FSEQ_E_HIP):
        """This rank's host has failed and makes no further calls: the other ranks return FSEQ_E_PEER from their next exchange."""
        self._check(self.L.fseq_shard_abort(self.h, int(code)))

    def set_memory_budget(self, nbytes):
        """Device memory this context may hold in all (ranks that share a card); 0 = whatever is free."""
        self._check(self.L.fseq_set_memory_budget(self.h, int(nbytes)))

    def set_progress(self, fn):
        """fn(stage, current_step, step_max) at the phase boundaries of run() (None: off); step_max() / current_step()
        may be polled from another thread."""
        self._progress = PROGRESS_FN(lambda _u, stage, cur, mx: fn(stage, cur, mx)) if fn else PROGRESS_FN()
        self._check(self.L.fseq_set_progress(self.h, self._progress, None))

    def step_max(self):
        return int(self.L.fseq_step_max(self.h))

    def current_step(self):
        return int(self.L.fseq_current_step(self.h))

    def shard_xbuf_words(self, world):
        return int(self.L.fseq_shard_xbuf_words(self.h, world))

    def shard_columns(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._check(self.L.fseq_shard_columns(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def shard_owner(self, rb):
        r = C.c_uint32()
        self._check(self.L.fseq_shard_owner(self.h, int(rb), C.byref(r)))
        return r.value

    # ---- input (delegate->sequences(), delegate->alphabet())
    def set_sequences(self, msa):
        """msa: uint8 array [m, n] of raw symbol bytes, any memory order."""
        assert msa.dtype == np.uint8 and msa.shape == (self.m, self.n)
        self._check(self.L.fseq_set_matrix(self.h, msa.ctypes.data, msa.strides[0], msa.strides[1]))

    def set_device_columns(self, ptr, ld, sigma, keepalive=None):
        self._keep = keepalive
        self._check(self.L.fseq_set_device_columns(self.h, ptr, ld, sigma))

    def set_device_columns_packed(self, ptr, ld_bytes, sigma, bits, keepalive=None):
        self._keep = keepalive
        self._check(self.L.fseq_set_device_columns_packed(self.h, ptr, ld_bytes, sigma, bits))

    def generate_synthetic(self, seed, n_founders, block_len, mu, kind=0):
        s = SynthSpec(seed, n_founders, block_len, synth_threshold(mu), kind)
        self._check(self.L.fseq_generate_synthetic(self.h, C.byref(s)))

    def get_sequences(self, c0=0, c1=None):
        c1 = self.n if c1 is None else c1
        out = np.zeros((self.m, c1 - c0), dtype=np.uint8, order="F")
        self._check(self.L.fseq_get_matrix(self.h, c0, c1, out.ctypes.data, out.strides[0], out.strides[1]))
        return out

    # ---- the path
    def run(self):
        res = Result()
        rc = self.L.fseq_run_segmentation(self.h, C.byref(res))
        self.result = res
        self._check(rc)
        return res

    @property
    def max_segment_size(self):
        return self.result.max_segment_size

    def traceback(self):
        out = np.zeros(self.result.dp_segment_count, dtype=DPARG_DTYPE)
        if len(out):
            self._check(self.L.fseq_get_traceback(self.h, out.ctypes.data))
        return out

    def reduced_traceback(self):
        out = np.zeros(self.result.segment_count, dtype=SEGMENT_DTYPE)
        if len(out):
            self._check(self.L.fseq_get_segments(self.h, out.ctypes.data))
        return out

    def boundary_state(self, i):
        """(input_permutation, input_divergence) of reduced_pbwt_samples[i]."""
        a = np.zeros(self.m, dtype=np.uint32)
        d = np.zeros(self.m, dtype=np.uint32)
        self._check(self.L.fseq_boundary_state(self.h, i, a.ctypes.data, d.ctypes.data))
        return a, d

    def short_path_runs(self):
        k = self.result.max_segment_size
        f = np.zeros(k, dtype=np.uint32)
        r = np.zeros(k, dtype=np.uint32)
        self._check(self.L.fseq_short_path_runs(self.h, f.ctypes.data, r.ctypes.data))
        return f, r

    # ---- joining (join_context / greedy_matcher, host side)
    def join_greedy(self):
        """permutations[s, r]: input row whose segment-s substring is founder r's content."""
        perm = np.zeros((self.result.segment_count, self.result.max_segment_size), dtype=np.uint32)
        self._check(self.L.fseq_join_greedy(self.h, perm.ctypes.data))
        return perm

    def join_bipartite(self):
        perm = np.zeros((self.result.segment_count, self.result.max_segment_size), dtype=np.uint32)
        self._check(self.L.fseq_join_bipartite(self.h, perm.ctypes.data))
        return perm

    def join_random(self, seed=0):
        perm = np.zeros((self.result.segment_count, self.result.max_segment_size), dtype=np.uint32)
        self._check(self.L.fseq_join_random(self.h, seed, perm.ctypes.data))
        return perm

    def write_segments(self, msa, joining, path):
        """--output-segments for the joining method (JOIN_GREEDY / JOIN_BIPARTITE / JOIN_RANDOM)."""
        assert msa.dtype == np.uint8 and msa.flags["C_CONTIGUOUS"] and msa.shape == (self.m, self.n)
        rows = (C.c_void_p * self.m)(*[msa.ctypes.data + r * msa.strides[0] for r in range(self.m)])
        self._check(self.L.fseq_write_segments(self.h, rows, joining, path.encode() if path else None))

    def write_founders_device(self, permutations, path):
        """--output-founders from the alignment resident on the device (no host rows)."""
        perm = np.ascontiguousarray(permutations, dtype=np.uint32)
        self._check(self.L.fseq_write_founders_device(self.h, perm.ctypes.data, path.encode() if path else None))

    def write_founders(self, msa, permutations, path):
        """msa: the raw input rows as a C-contiguous uint8 array [m, n]."""
        assert msa.dtype == np.uint8 and msa.flags["C_CONTIGUOUS"] and msa.shape == (self.m, self.n)
        rows = (C.c_void_p * self.m)(*[msa.ctypes.data + r * msa.strides[0] for r in range(self.m)])
        perm = np.ascontiguousarray(permutations, dtype=np.uint32)
        self._check(self.L.fseq_write_founders(self.h, rows, perm.ctypes.data, path.encode() if path else None))

    # ---- debug / parity of intermediate state
    def debug_dp(self):
        k = self.n - self.segment_length + 1
        lb = np.zeros(k, dtype=np.uint32)
        mx = np.zeros(k, dtype=np.uint32)
        sz = np.zeros(k, dtype=np.uint32)
        self._check(self.L.fseq_debug_dp(self.h, lb.ctypes.data, mx.ctypes.data, sz.ctypes.data))
        return lb, mx, sz

    def debug_dp_owned(self):
        """(first, last, owns_final_cell, whole_arrays): the DP entries this rank computed (everything when not sharded)."""
        a, b = C.c_uint64(), C.c_uint64()
        f, w = C.c_int(), C.c_int()
        self._check(self.L.fseq_debug_dp_owned(self.h, C.byref(a), C.byref(b), C.byref(f), C.byref(w)))
        return a.value, b.value, bool(f.value), bool(w.value)

    def debug_clock(self):
        """(GHz, workgroups) of phase C's kernel in the last run -- diagnostic builds (-DFSEQ_CLOCK_STAMPS) only."""
        g, n = C.c_double(), C.c_uint32()
        self._check(self.L.fseq_debug_clock(self.h, C.byref(g), C.byref(n)))
        return g.value, n.value

    def debug_block_state(self, b):
        a = np.zeros(self.m, dtype=np.uint32)
        d = np.zeros(self.m, dtype=np.uint32)
        self._check(self.L.fseq_debug_block_state(self.h, b, a.ctypes.data, d.ctypes.data))
        return a, d

    def debug_column_list(self, col):
        cap = self.m + 2
        v = np.zeros(cap, dtype=np.uint32)
        c = np.zeros(cap, dtype=np.uint32)
        ne, c0, comp = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.L.fseq_debug_column_list(self.h, col, v.ctypes.data, c.ctypes.data,
                                                  C.byref(ne), C.byref(c0), C.byref(comp)))
        return v[:ne.value].copy(), c[:ne.value].copy(), c0.value, bool(comp.value)

    def set_tuning(self, name, value="1"):
        """One of the library's FSEQ_* diagnostic knobs for this context (value None = off); the environment is only
        read when a context is created."""
        self._check(self.L.fseq_debug_set_tuning(self.h, name.encode(), None if value is None else str(value).encode()))

    def join_profile(self):
        """Host time of the last join_*() call: {ms_d2h, ms_classes, ms_edges, ms_draw, ms_total, bytes_d2h}."""
        jp = JoinProfile()
        self._check(self.L.fseq_get_join_profile(self.h, C.byref(jp)))
        return {k: getattr(jp, k) for k, _ in JoinProfile._fields_}

    def timings(self):
        t = Timings()
        self._check(self.L.fseq_get_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in Timings._fields_}
