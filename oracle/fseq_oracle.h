/*
 * fseq_oracle.h -- CPU restatement of the founder-sequences segmentation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU baseline
 * ("port") for the MI355X implementation.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library never links it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this
 * path, and its own sources for the path cannot be compiled in this
 * image without stand-ins for absent headers (rmq.hh includes <sdsl/bits.hpp>; the
 * pBWT arithmetic lives in the un-vendored libbio submodule).  The
 * restatement below follows the reference text line by line where the text exists
 * (every function cites file:line under /root/reference) and SURVEY.md Appendix B's
 * assumptions A1-A7 where it does not (libbio::pbwt::pbwt_context).
 *
 * All functions are plain C99, no dependencies beyond libc/pthreads.
 */
#ifndef FSEQ_ORACLE_H
#define FSEQ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* segmentation_dp_arg, include/founder_sequences/segmentation_dp_arg.hh:18-23.
 * operator< compares segment_max_size only (:53). Defaults: lb=rb=0, both sizes UINT32_MAX. */
typedef struct fso_dp_arg {
	uint64_t lb;                 /* inclusive */
	uint64_t rb;                 /* exclusive */
	uint32_t segment_max_size;
	uint32_t segment_size;
} fso_dp_arg;

/* ---- rmq <vector<dp_arg>, less<>, 64>: include/founder_sequences/rmq.hh:22-118 ---- */
/* Keys are read as *(uint32_t const *)((char const *) base + i * stride): works for a plain
 * uint32 array (stride 4) and for fso_dp_arg::segment_max_size (stride sizeof(fso_dp_arg)). */
typedef struct fso_rmq fso_rmq;
fso_rmq *fso_rmq_new(void const *key_base, size_t stride, size_t capacity, unsigned block_size);
void     fso_rmq_free(fso_rmq *r);
void     fso_rmq_update(fso_rmq *r, size_t last_idx);                 /* rmq.hh:61-81  */
size_t   fso_rmq_query(fso_rmq const *r, size_t beg, size_t end);     /* rmq.hh:85-105 */

/* ---- calculate_segmentation_lp_dp_arg: founder-sequences/segmentation_lp_context.cc:393-481 ---- */
/* pairs: ascending (value,count) list, values[i] / counts[i], npairs entries (Appendix B A6). */
void fso_dp_step(
	uint32_t const *values, uint32_t const *counts, size_t npairs,
	fso_dp_arg const *dp, fso_rmq const *rmq,
	uint64_t seq_count, uint64_t segment_length, uint64_t lb, uint64_t text_pos,
	fso_dp_arg *min_arg);

/* ---- libbio::pbwt::pbwt_context restatement ---- */
typedef struct fso_pbwt fso_pbwt;
/* sym(r,c) = base[r*row_stride + c*col_stride]; only the relative order of symbol bytes matters. */
fso_pbwt *fso_pbwt_new(uint8_t const *base, size_t row_stride, size_t col_stride,
                       uint32_t m, uint64_t n, int with_counts);
void      fso_pbwt_free(fso_pbwt *p);
void      fso_pbwt_prepare(fso_pbwt *p);                      /* A3: identity, d=0, H={0:m}, idx=0 */
void      fso_pbwt_set_state(fso_pbwt *p, uint32_t const *a, uint32_t const *d, uint64_t idx);
void      fso_pbwt_step(fso_pbwt *p);                         /* one column: Appendix A step 2 */
uint64_t  fso_pbwt_idx(fso_pbwt const *p);                    /* sequence_idx() */
uint32_t const *fso_pbwt_a(fso_pbwt const *p);                /* input_permutation() */
uint32_t const *fso_pbwt_d(fso_pbwt const *p);                /* input_divergence() */
/* divergence value counts, ascending; returns npairs (buffers need room for m entries, may be NULL). */
size_t    fso_pbwt_counts(fso_pbwt const *p, uint32_t *values, uint32_t *counts);
uint32_t  fso_pbwt_unique_substring_count_lhs(fso_pbwt const *p, uint64_t lb);   /* A7 */
/* A7: one (first row id of run, run length) entry per run in pBWT order; returns number of runs. */
size_t    fso_pbwt_unique_substring_count_idxs_lhs(fso_pbwt const *p, uint64_t lb,
                                                   uint32_t *first_idx, uint32_t *run_len);

/* ---- the long path: segmentation_lp_context.cc:26-390 + update_pbwt_task.cc:13-35 ---- */
typedef struct fso_result {
	int       status;              /* 0 ok; 1 max_segment_size >= m (generate_context.cc:192-200) */
	uint32_t  max_segment_size;
	uint64_t  n_dp_segments;       /* S: traceback entries before merging */
	fso_dp_arg *traceback;         /* S entries, left to right */
	uint64_t  n_segments;          /* S': after find_segments_greedy */
	fso_dp_arg *reduced;           /* S' entries {lb, rb, UINT32_MAX, size} */
	uint32_t *a;                   /* S' x m : input_permutation at each reduced rb */
	uint32_t *d;                   /* S' x m : input_divergence  at each reduced rb */
	uint64_t  dp_size;             /* n - L + 1 */
	fso_dp_arg *dp;                /* whole DP array (debug / parity of intermediate state) */
	uint64_t  n_samples;           /* pBWT samples taken in pass 1 */
	uint64_t  pass2_cells;         /* R: cells re-processed in pass 2 */
	uint64_t  dp_pairs_total;      /* sum over DP steps of the (value,count) list length */
	double    t_pass1, t_traceback, t_pass2, t_merge;   /* seconds */
	int       pass2_threads;
} fso_result;

/* sample_rate: columns between pBWT samples (generate_context.cc:113-124 computes ceil(q*sqrt(n))
 * or n+1 for "no sampling").  threads: pass-2 worker threads (pass 1 is always one thread, F6).
 * keep_dp: keep the whole DP array in the result.  Long path requires n >= 2L. */
int  fso_segment_long(uint8_t const *base, size_t row_stride, size_t col_stride,
                      uint32_t m, uint64_t n, uint64_t segment_length, uint64_t sample_rate,
                      int threads, int keep_dp, fso_result *out);
void fso_result_free(fso_result *r);

/* ---- the short path: segmentation_sp_context.cc:21-28 ---- */
/* returns number of distinct rows over [0,n); first_idx/run_len sized m. */
size_t fso_segment_short(uint8_t const *base, size_t row_stride, size_t col_stride,
                         uint32_t m, uint64_t n, uint32_t *first_idx, uint32_t *run_len);

/* ---- synthetic founder-mosaic generator; same arithmetic as the device one ---- */
typedef struct fso_synth_spec {
	uint64_t seed;
	uint32_t n_founders;        /* K */
	uint32_t block_len;         /* B: recombination block length in columns */
	uint64_t mut_threshold;     /* mutation iff h(3,r,c) < mut_threshold ( = mu * 2^64 ) */
	uint32_t kind;              /* 0: uniform over "ACGT"; 1: "ACGTRYSWKMBDHVN-" with P(ACGT)=0.9 */
} fso_synth_spec;
uint64_t fso_synth_threshold(double mu);
uint8_t  fso_synth_byte(fso_synth_spec const *s, uint64_t r, uint64_t c);
/* fills out[r*row_stride + c*col_stride] for r<m, c in [c0,c1) with the raw ASCII byte */
void     fso_synth_fill(fso_synth_spec const *s, uint32_t m, uint64_t c0, uint64_t c1,
                        uint8_t *out, size_t row_stride, size_t col_stride);

#ifdef __cplusplus
}
#endif
#endif
