/* fseq_debug.h -- intermediate state of the HIP segmentation path, for parity tests and diagnostics.
 * Not part of the drop-in boundary (include/fseq.h): nothing here replaces a call of the reference; the entry points
 * are exported by the same library. */
#ifndef FSEQ_DEBUG_H
#define FSEQ_DEBUG_H

#include "fseq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The DP's round schedule (debug / tests of the host logic; no device needed): number of rounds for (L, n),
 * cells per round, and how many leading rounds only need the lists of columns < col_hi (what the host hands
 * to a resumed DP launch while later columns are still being produced). */
int  fseq_debug_dp_schedule(uint64_t segment_length, uint64_t n, uint64_t col_hi, uint32_t *n_rounds, uint32_t *cells_per_round,
                            uint32_t *rounds_within, int *pipelined);
/* Whole DP array (debug / parity of intermediate state): n - L + 1 entries, rb = index + L.
 * A rank of a sharded run computes the entries [*first, *last) (and the final cell's, entry n - L, when *final_cell);
 * since round 4 it holds the other ranks' entries only for a window in front of its own (*whole_arrays == 0), so
 * fseq_debug_dp is meaningful there for its own entries only (FSEQ_SHARD_DP_FULL keeps whole arrays on every rank). */
int  fseq_debug_dp(fseq_ctx *ctx, uint32_t *lb, uint32_t *max_size, uint32_t *size);
int  fseq_debug_dp_owned(fseq_ctx *ctx, uint64_t *first, uint64_t *last, int *final_cell, int *whole_arrays);
/* Exact (a,d) at column block_idx*block_len (block_idx <= n_blocks). */
int  fseq_debug_block_state(fseq_ctx *ctx, uint64_t block_idx, uint32_t *a_out, uint32_t *d_out);
/* Per-column divergence list after column c: descending (value,count), up to list_cap+1 entries;
 * *n_entries, *cnt0 (count of value 0) and *complete (list reaches the smallest value). */
int  fseq_debug_column_list(fseq_ctx *ctx, uint64_t c, uint32_t *values, uint32_t *counts,
                            uint32_t *n_entries, uint32_t *cnt0, uint32_t *complete);

/* The restated rmq.hh (rmq<..., 64>, include/founder_sequences/rmq.hh:61-118, quirks included) on caller-supplied
 * keys, straight on the device routines the DP uses (debug / tests): the stack masks and the sparse table are
 * built in closed form from the keys, every query [beg, end) is answered by the HBM path (index_hbm) and, when
 * the array fits the LDS rings (count <= 4096), by the LDS path (index_lds, else 0xFFFFFFFF). */
int  fseq_debug_rmq(int device, uint32_t const *keys, uint32_t count, uint32_t const *beg, uint32_t const *end, uint32_t n_queries,
                    uint32_t *index_hbm, uint32_t *index_lds);

/* Phase ranges (roctx: rocprofv3 --marker-trace) pushed and popped by this process so far, and whether the library was built
 * against roctx: a run of the long path adds four of each (phases A + B, phases C + D, traceback + merge, pass 2). */
int  fseq_debug_ranges(uint64_t *pushes, uint64_t *pops, int *with_roctx);

/* Diagnostic builds only (-DFSEQ_CLOCK_STAMPS; FSEQ_E_UNSUPPORTED otherwise): the clock phase C's kernel held in the last run,
 * d(s_memtime) / d(s_memrealtime) x 100 MHz stamped once around every workgroup, the median over the workgroups
 * (MI355X_MICROARCH.md, "DVFS give-back" item 6).  tools/clock_probe.py. */
int  fseq_debug_clock(fseq_ctx *ctx, double *ghz, uint32_t *workgroups);

/* The library's diagnostic knobs (FSEQ_* names, listed in csrc/fseq_api.hip `struct Tuning`): a context reads them
 * from the environment once, at fseq_create; this sets one afterwards (value NULL = off).  Every knob selects among
 * exact alternatives; results never depend on them.  Call before the first fseq_run_segmentation (a later call drops the work buffers and
 * the result of the context: the geometry may change); on a sharded context before the input is set, identically on
 * every rank (FSEQ_E_ARG afterwards: the input is laid out for the block partition in force when it was set). */
int  fseq_debug_set_tuning(fseq_ctx *ctx, char const *name, char const *value);

#ifdef __cplusplus
}
#endif
#endif
