/*
 * fseq.h -- C ABI of the MI355X-native segmentation engine (libfseq_hip.so).
 *
 * Drop-in boundary for ONE path of tsnorri/founder-sequences: everything
 * founder_sequences::segmentation_lp_context does (pBWT pass 1 + minimum-segmentation DP +
 * traceback + pass 2 + greedy segment merge) and segmentation_sp_context::process.
 * The reference has no FFI for this path; its boundary is the C++ delegate pair
 *   segmentation_lp_context_delegate          include/founder_sequences/segmentation_lp_context.hh:45-58
 *   segmentation_context / _delegate          include/founder_sequences/segmentation_context.hh:14-33
 * and the hand-off type
 *   segmentation_container                    include/founder_sequences/segmentation_container.hh:15-20
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes
 * only; no exceptions and no exit() cross this ABI.  All paths are relative to /root/reference.
 */
#ifndef FSEQ_H
#define FSEQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: fseq_timings grew (dp_chunks .. reserved), FSEQ_E_PEER, fseq_set_memory_budget, fseq_set_progress /
 *    fseq_step_max / fseq_current_step; the fseq_debug_* entry points moved to include/fseq_debug.h */
/* 3: fseq_shard_abort */
/* 4: fseq_timings grew (phase_a_trie_given_up: 120 -> 128 bytes): a host that calls fseq_get_timings must be rebuilt */
/* 5: fseq_timings grew (reduced_blocks, reduced_rows_mean, reduced_redone: 128 -> 136 bytes) */
#define FSEQ_ABI_VERSION 5

enum {
	FSEQ_OK             = 0,
	FSEQ_E_ARG          = 1,   /* bad argument / call order */
	FSEQ_E_NO_REDUCTION = 2,   /* max segment size >= m: generate_context.cc:192-200 */
	FSEQ_E_HIP          = 3,   /* HIP runtime error, see fseq_last_error */
	FSEQ_E_OOM          = 4,   /* a device (or pinned host) allocation failed; fseq_last_error has the sizes.  Typically the
	                            * per-column lists of a very diverse input: n x (X + 3) x 8 bytes at list capacity X (DESIGN.md
	                            * section 7).  Nothing was written; other contexts of the process are not affected */
	FSEQ_E_UNSUPPORTED  = 5,   /* shape outside what this build's kernels cover (fails loudly, no CPU fallback) */
	FSEQ_E_PEER         = 6    /* sharded run: another rank failed; every rank returns from the same exchange */
};

typedef struct fseq_ctx fseq_ctx;

/* Construction parameters = what segmentation_lp_context reads from its delegate
 * (segmentation_lp_context.hh:47-50): sequence_count(), sequences().front().size(),
 * segment_length(), pbwt_sample_rate(). */
typedef struct fseq_params {
	uint32_t m;                 /* sequence_count()                                   */
	uint64_t n;                 /* sequence length (columns)                           */
	uint64_t segment_length;    /* L, --segment-length-bound (cmdline.ggo:16-17)       */
	uint64_t pbwt_sample_rate;  /* accepted for interface parity; results do not depend on it. The
	                               engine keeps an exact (a,d) state every block_len columns instead. */
	uint32_t block_len;         /* column-block length B; 0 = choose automatically     */
	uint32_t list_cap;          /* X: per-column divergence list covers the X largest entries; 0 = default.
	                               Too small a value is detected and retried internally (results stay exact). */
	int32_t  device;            /* HIP device ordinal                                  */
} fseq_params;

/* Synthetic founder-mosaic input, generated on the device. */
typedef struct fseq_synth_spec {
	uint64_t seed;
	uint32_t n_founders;
	uint32_t block_len;         /* recombination block length */
	uint64_t mut_threshold;     /* mutation iff hash < mut_threshold (= mu * 2^64) */
	uint32_t kind;              /* 0: "ACGT" uniform; 1: "ACGTRYSWKMBDHVN-", P(ACGT)=0.9 */
} fseq_synth_spec;

/* One reduced segment: segmentation_container::reduced_traceback entry
 * (segmentation_dp_arg.hh:18-23 built by the 3-argument constructor at
 * segmentation_lp_context.cc:368,380). */
typedef struct fseq_segment {
	uint64_t lb;                /* inclusive */
	uint64_t rb;                /* exclusive */
	uint32_t segment_size;
	uint32_t reserved;
} fseq_segment;

/* One traceback entry before merging (segmentation_lp_context.cc:191-224). */
typedef struct fseq_dp_arg {
	uint64_t lb, rb;
	uint32_t segment_max_size, segment_size;
} fseq_dp_arg;

typedef struct fseq_result {
	uint32_t max_segment_size;      /* segmentation_container::max_segment_size           */
	uint32_t short_path;            /* 1 if n < 2L (generate_context.cc:386-389)          */
	uint64_t dp_segment_count;      /* S reported by context_did_finish_traceback (:221-223) */
	uint64_t segment_count;         /* S' = reduced_traceback.size()                       */
} fseq_result;

typedef struct fseq_timings {
	double ms_total;                /* wall, whole fseq_run_segmentation call                       */
	double ms_phase_a, ms_phase_b, ms_phase_c, ms_dp, ms_pass2, ms_host;   /* device phases: HIP event time */
	double ms_colstep_kernels;      /* summed HIP-event time of the column-update kernels (A + C + pass 2) */
	uint64_t colstep_launches;      /* number of those launches                            */
	uint64_t colstep_cells;         /* cells (rows x columns) those launches processed     */
	uint64_t pass2_cells;           /* R: cells re-processed for the boundary snapshots    */
	uint32_t list_cap_used;         /* X that produced the result                          */
	uint32_t retries;               /* list_cap retries                                    */
	uint32_t block_len;             /* B in use                                            */
	uint32_t n_blocks;
	uint32_t dp_chunks;             /* chunks of the speculative DP (0: the serial kernel ran)               */
	uint32_t dp_sweeps;             /* sweeps it compared until no key changed (>= 1000: serial fallback took over) */
	uint32_t phase_a_fallbacks;     /* blocks in which a merge of the key-space tree exceeded the LDS bitmap and ran in slices */
	uint32_t phase_a_given_up;      /* blocks the key-space tree handed to the column sweep (their merges would have sliced past the budget) */
	uint32_t phase_a_trie_given_up; /* streamed rows: blocks the trie over 16-column words handed to the key-space tree (too many distinct keys) */
	uint32_t reduced_blocks;        /* [ABI 5] blocks whose column updates ran on their representative rows (0: every block on all rows) */
	uint32_t reduced_rows_mean;     /* ... representatives per such block, mean                                  */
	uint32_t reduced_redone;        /* ... blocks run again on all rows because a list could not be proven on the representatives */
} fseq_timings;

uint32_t    fseq_abi_version(void);
char const *fseq_strerror(int code);

/* replaces: new segmentation_lp_context(delegate, ...)  generate_context.cc:184 */
int  fseq_create(fseq_params const *params, fseq_ctx **out);
/* replaces: segmentation_lp_context::cleanup  segmentation_lp_context.hh:117 */
void fseq_destroy(fseq_ctx *ctx);
char const *fseq_last_error(fseq_ctx const *ctx);

/* Input = delegate->sequences() (m spans of n raw bytes, founder_sequences.hh:38-40) plus
 * delegate->alphabet() (generate_context.cc:135-147).  rows[r] points at n bytes.  The bytes
 * are mapped to dense codes in ascending byte order and stored column-major in HBM. */
int  fseq_set_rows(fseq_ctx *ctx, uint8_t const *const *rows);
/* Same, from one host buffer: sym(r,c) = base[r*row_stride + c*col_stride]. */
int  fseq_set_matrix(fseq_ctx *ctx, uint8_t const *base, size_t row_stride, size_t col_stride);
/* Input already resident in HBM: column-major dense codes, one per byte, column c at d_codes + c*ld,
 * ld >= m.  sigma = number of codes (codes are < sigma: checked with one pass over the columns, FSEQ_E_ARG if a
 * larger code is found).  The buffer is borrowed, not copied (and not repacked: inputs the library uploads itself
 * are stored at 2 / 4 / 8 bits per cell by sigma).  The code check is one pass over the columns plus a stream
 * synchronisation per call, and only when sigma is below what the code width can hold (sigma = 2^bits: no pass). */
int  fseq_set_device_columns(fseq_ctx *ctx, void const *d_codes, size_t ld, uint32_t sigma);
/* The same for columns that are already packed the way the library stores them: bits = 2, 4 or 8 per
 * code (codes < sigma <= 2^bits), row r of a column in byte r / (8 / bits) at bit (r mod (8 / bits)) * bits,
 * column c at d_packed + c * ld_bytes, ld_bytes a multiple of 16 and >= ceil(m * bits / 8). */
int  fseq_set_device_columns_packed(fseq_ctx *ctx, void const *d_packed, size_t ld_bytes, uint32_t sigma, uint32_t bits);
/* Generate the alignment on the device (bench / large configs). */
int  fseq_generate_synthetic(fseq_ctx *ctx, fseq_synth_spec const *spec);
/* Copy columns [c0,c1) back as raw bytes, out[r*row_stride + (c-c0)*col_stride] (tests, writers). */
int  fseq_get_matrix(fseq_ctx *ctx, uint64_t c0, uint64_t c1, uint8_t *out, size_t row_stride, size_t col_stride);

/* ---- one alignment over several GPUs (one process per GPU) ----
 * replaces: the reference's only parallel axis for this path, independent update_pbwt_task's over column ranges
 * on the global concurrent queue (segmentation_lp_context.cc:319-332, update_pbwt_task.cc:13-35), extended to
 * pass 1: rank r of `world` owns a contiguous range of column blocks -- its share of the alignment, of phases A
 * and C, of the DP chunks and of pass 2.  The exchange steps (the W composite key blocks of phase B, the DP keys
 * after every sweep, the merge thresholds: a few MB in all) go through ONE caller-supplied all-reduce over a
 * caller-owned device buffer; a C++ host passes a function that calls ncclAllReduce on it, the Python mirror one
 * that calls torch.distributed.all_reduce (RCCL) on the tensor that owns the buffer.
 *   fn(user, offset_words, count_words, op): all-reduce xbuf[offset .. offset + count) (uint32 words; op 0 = sum,
 *   1 = max) over all ranks, IN PLACE, and return 0 once the result is visible to work submitted afterwards on
 *   any stream of this device (the library has synchronised its own stream before the call).
 * Call before the input is set (each rank then generates / uploads / borrows only its own columns; borrowed
 * device columns hold the columns [first, last) fseq_shard_columns reports, column `first` at the base pointer).
 * Every rank must make the same calls in the same order; results (fseq_result, traceback, segments, timings)
 * are identical on all ranks, fseq_boundary_state(i) answers on the rank that owns segments[i].rb
 * (fseq_shard_owner) and returns FSEQ_E_ARG elsewhere. */
typedef int (*fseq_allreduce_fn)(void *user, uint64_t offset_words, uint64_t count_words, int op);
/* device words (uint32) the exchange buffer must hold for this context's shape on `world` ranks */
uint64_t fseq_shard_xbuf_words(fseq_ctx const *ctx, uint32_t world);
int  fseq_set_shard(fseq_ctx *ctx, uint32_t rank, uint32_t world, void *xbuf_device, uint64_t xbuf_words,
                    fseq_allreduce_fn fn, void *user);
/* A rank whose HOST failed between two library calls (it could not read its input, allocate a buffer of its own, ...) and
 * will make no further calls on this context: tells the other ranks, who learn of it in the exchange they make next and
 * return FSEQ_E_PEER instead of waiting in a collective.  One status exchange, at most once per context; failures inside
 * the library's own calls (fseq_set_rows, fseq_generate_synthetic, fseq_run_segmentation) are posted by the library
 * itself.  code: what the others see as this rank's error code (0 = FSEQ_E_HIP).  Nothing to do on unsharded contexts. */
int  fseq_shard_abort(fseq_ctx *ctx, int code);
/* columns [*first, *last) this rank holds (its block range plus the few columns of the next rank that its last
 * DP round reads) */
int  fseq_shard_columns(fseq_ctx const *ctx, uint64_t *first, uint64_t *last);
/* rank that owns the boundary state at column rb */
int  fseq_shard_owner(fseq_ctx const *ctx, uint64_t rb, uint32_t *rank);

/* ---- Row-sharded pBWT sweep: BASELINE.json north_star's partition, as a conformance path ----------------------
 * "Rows shard across the GPUs with a per-column sigma-bucket-histogram all-reduce and a boundary exchange for the
 * divergence scan": rank g of `world` owns the positions [m g / world, m (g + 1) / world) of the order (a_k, d_k)
 * and supplies the symbols of the rows fseq_rowshard_rows reports; every column (and 2-bit digit of a wider
 * alphabet) costs two all-reduces through the same fseq_allreduce_fn as above -- the column itself plus the scattered
 * (a, d), and one 12-word summary per rank (bucket counts, running maxima, symbols seen).  The sweep therefore runs
 * at the latency of the collective (DESIGN.md section 6 has the measured curve); the production split of one
 * alignment over GPUs is fseq_set_shard's column blocks, which exchange per phase, not per column.
 * Computes what the per-column update of libbio::pbwt_context computes when founder_sequences.hh:56-65 drives it
 *  over the columns [0, ncols) from the identity order: bit-identical to the single-GPU path.
 * d_cols: device, column-major packed codes < sigma (the layout of fseq_set_device_columns_packed: `bits` per row,
 * row r of a column in byte r * bits / 8, ld bytes from column to column, ld a multiple of 4); only this rank's rows
 * are read.  a_out / d_out: host, m entries each; the entries [*pos_lo, *pos_hi) are written. */
typedef struct fseq_rowshard {
	int32_t  device;
	uint32_t rank, world;
	uint32_t m, sigma, bits;
	uint64_t ncols;
	void const *d_cols;
	size_t   ld;
	void    *xbuf;                 /* device, fseq_rowshard_xbuf_words uint32 words; every rank its own */
	uint64_t xbuf_words;
	fseq_allreduce_fn fn;          /* may be NULL when world == 1 */
	void    *user;
} fseq_rowshard;
uint64_t fseq_rowshard_xbuf_words(uint32_t m, uint32_t bits, uint32_t world);
int  fseq_rowshard_rows(uint32_t m, uint32_t bits, uint32_t rank, uint32_t world, uint32_t *row_lo, uint32_t *row_hi);
int  fseq_rowshard_pbwt(fseq_rowshard const *args, uint32_t *a_out, uint32_t *d_out, uint32_t *pos_lo, uint32_t *pos_hi,
                        double *ms, uint64_t *n_exchanges);

/* replaces: generate_traceback + update_samples_to_traceback_positions + find_segments_greedy
 * (segmentation_lp_context.cc:26-390) or segmentation_sp_context::process
 * (segmentation_sp_context.cc:21-28) when n < 2L.  Returns FSEQ_E_NO_REDUCTION exactly when the
 * reference would print "Unable to reduce the number of sequences" (the traceback is still
 * available then). */
int  fseq_run_segmentation(fseq_ctx *ctx, fseq_result *res);
/* Several independent alignments (chromosomes, windows) at once: one host thread and one stream per
 * context, all in flight together -- the DP of one alignment keeps a single CU busy, so the chip is
 * shared (measured: 8 alignments of BASELINE C2 shape in flight give 3.5x the throughput of one after
 * the other; beyond 4 in flight set GPU_MAX_HW_QUEUES -- the HIP runtime's default of 4 hardware queues
 * makes contexts that share a queue take turns).  results[i] / return_codes[i] are what fseq_run_segmentation(ctxs[i], ...) gives; the
 * return value is FSEQ_OK when every call was made (look at return_codes for their outcomes). */
int  fseq_run_segmentation_batch(fseq_ctx *const *ctxs, size_t count, fseq_result *results, int *return_codes);

/* segmentation_lp_context::m_segmentation_traceback_res (before merging), dp_segment_count entries. */
int  fseq_get_traceback(fseq_ctx *ctx, fseq_dp_arg *out);
/* segmentation_container::reduced_traceback, segment_count entries. */
int  fseq_get_segments(fseq_ctx *ctx, fseq_segment *out);
/* segmentation_container::reduced_pbwt_samples[i]: input_permutation() / input_divergence() at
 * sequence_idx() == segments[i].rb (greedy_matcher.cc:242-257, join_context.cc:73).  m uint32 each. */
int  fseq_boundary_state(fseq_ctx *ctx, uint64_t i, uint32_t *a_out, uint32_t *d_out);
/* Short path result (segmentation_sp_context.hh:28): one (first row id, copy number) per distinct
 * row in pBWT order; returns max_segment_size entries. */
int  fseq_short_path_runs(fseq_ctx *ctx, uint32_t *first_idx, uint32_t *run_len);

/* ---- segment joining on the host ----
 * replaces: join_context::join_greedy -> greedy_matcher::match (join_context.cc:211-229,
 * greedy_matcher.cc:204-465).  permutations: segment_count x max_segment_size uint32, row-major;
 * permutations[s][r] = index of the input sequence whose [lb_s, rb_s) substring is founder r's
 * content in segment s. */
int  fseq_join_greedy(fseq_ctx *ctx, uint32_t *permutations);
/* The same matcher on caller-supplied boundary states (no context, no device): segments given as
 * lb[i], rb[i]; a, d: n_segments x m.  Used by the CPU tests of the host logic. */
int  fseq_greedy_match_host(uint32_t m, uint32_t max_segment_size, uint64_t n_segments, uint64_t const *lb, uint64_t const *rb,
                            uint32_t const *a, uint32_t const *d, uint32_t *permutations);
/* ---- the non-greedy joiners, host C++ as in the reference ----
 * replaces: join_context::join_with_bipartite_matching -> bipartite_matcher::match (join_context.cc:231-236,
 * bipartite_matcher.cc:17-151, create_segment_texts_task.cc:15-81, merge_segments_task.cc:133-195 with
 * INTERSECTION scoring, main.cc:126).  Lemon's MaxWeightedPerfectMatching is replaced by an own
 * Kuhn-Munkres: every matching has the optimal total weight; which optimal matching Lemon picks is
 * not reproduced (parity unpinned).  Texts of equal size keep their classes' pBWT order
 * (the reference's std::sort leaves it to its standard library).  [r5] fseq_join_bipartite runs on the
 * device where the boundary states are (csrc/fseq_joinbip.hpp) while a segment has at most 181 texts and
 * the run is not sharded -- the same permutations, entry for entry, as the host form below. */
int  fseq_join_bipartite(fseq_ctx *ctx, uint32_t *permutations);
/* replaces: join_context::join_random_order_and_output (join_context.cc:259-289): std::mt19937(seed),
 * one std::shuffle per segment. */
int  fseq_join_random(fseq_ctx *ctx, uint32_t seed, uint32_t *permutations);
/* The same joiners on caller-supplied boundary states (no context, no device).  weights (optional):
 * n_segments - 1 total matching weights. */
int  fseq_bipartite_match_host(uint32_t m, uint32_t max_segment_size, uint64_t n_segments, uint64_t const *lb, uint64_t const *rb,
                               uint32_t const *a, uint32_t const *d, uint32_t *permutations, int64_t *weights);
int  fseq_random_join_host(uint32_t m, uint32_t max_segment_size, uint64_t n_segments, uint64_t const *lb, uint64_t const *rb,
                           uint32_t const *a, uint32_t const *d, uint32_t seed, uint32_t *permutations);
/* replaces: join_context::output_segments -> output_segments (segmentation_dp_arg.cc:13-104): the
 * --output-segments text file for the given joining method (greedy: the header line only, SURVEY.md
 * F5; bipartite: one line per segment text; random: one line per distinct substring with its copy
 * number).  path NULL or "-" = stdout. */
enum { FSEQ_JOIN_GREEDY = 0, FSEQ_JOIN_BIPARTITE = 1, FSEQ_JOIN_RANDOM = 2 };
int  fseq_write_segments(fseq_ctx *ctx, uint8_t const *const *rows, int joining, char const *path);
/* The same with the boundary states supplied by the caller (S' x m words each, segment-major): a sharded run, where
 * they sit on their owner ranks (fseq_shard_owner / fseq_boundary_state) and the host collects them. */
int  fseq_write_segments_host(fseq_ctx *ctx, uint8_t const *const *rows, int joining, uint32_t const *a, uint32_t const *d, char const *path);
/* replaces: join_context::output_in_permutation_order (join_context.cc:333-356): max_segment_size
 * lines; line r = concatenation over segments of rows[permutations[s][r]][lb_s, rb_s).  rows = the
 * raw input sequences.  path NULL or "-" = stdout. */
int  fseq_write_founders(fseq_ctx *ctx, uint8_t const *const *rows, uint32_t const *permutations, char const *path);
/* [ABI 5] the same from the alignment the device holds (uploaded, generated or borrowed): the lines are put together on the
 * device and leave in one copy per batch of rows; no host rows needed.  Not for sharded runs (a rank holds its own columns). */
int  fseq_write_founders_device(fseq_ctx *ctx, uint32_t const *permutations, char const *path);

int  fseq_get_timings(fseq_ctx const *ctx, fseq_timings *out);

/* Host time of the last fseq_join_* call on this context (wall, milliseconds): the boundary states' way to the host,
 * then -- greedy joiner -- the class tables (greedy_matcher.cc:31-68), the co-occurrence edges (:295-343) and the rounds
 * that draw them (:353-439); the other joiners report their matching under ms_draw. */
typedef struct fseq_join_profile {
	double ms_d2h, ms_classes, ms_edges, ms_draw, ms_total;
	uint64_t bytes_d2h;
} fseq_join_profile;
int  fseq_get_join_profile(fseq_ctx const *ctx, fseq_join_profile *out);

/* replaces: segmentation_lp_context::step_max() / current_step() (segmentation_lp_context.hh:122-127), which the
 * reference's progress indicator polls from another thread, and the per-stage timestamps of generate_context.cc.
 * The stages are the reference's: generate_traceback (pass 1 + DP; steps = columns), find_segments_greedy (steps =
 * traceback entries), update_samples_to_traceback_positions (steps = merged boundaries).  Pass 1 runs as phases over
 * all columns at once, so current_step moves at the phase boundaries (A, B, C, DP), scaled to columns.  Both counters
 * may be read from any thread while fseq_run_segmentation runs; the optional callback is made from the calling thread
 * at the same points (fn == NULL: none).  With the library built against roctx (the default build), every phase is
 * also an roctx range (rocprofv3 --marker-trace). */
enum { FSEQ_STAGE_TRACEBACK = 0, FSEQ_STAGE_MERGE = 1, FSEQ_STAGE_SAMPLES = 2 };
typedef void (*fseq_progress_fn)(void *user, int stage, uint64_t current_step, uint64_t step_max);
int      fseq_set_progress(fseq_ctx *ctx, fseq_progress_fn fn, void *user);
uint64_t fseq_step_max(fseq_ctx const *ctx);
uint64_t fseq_current_step(fseq_ctx const *ctx);

/* A context that shares its device with other contexts or ranks keeps its device allocations -- the pass-2 stride
 * states are sized from what is free -- inside `bytes` in all (0 = whatever is free on the device, the default). */
int  fseq_set_memory_budget(fseq_ctx *ctx, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif
