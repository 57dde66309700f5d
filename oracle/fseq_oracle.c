/*
 * fseq_oracle.c -- CPU restatement of the founder-sequences segmentation hot path.
 *
 * TEST INFRASTRUCTURE ONLY (parity oracle + timed CPU baseline).  PARITY UNPINNED: see
 * fseq_oracle.h.  File:line citations are relative to /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "fseq_oracle.h"

#include <assert.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

/* ------------------------------------------------------------------------------------------
 * rmq.hh:22-118, t_block_size given at construction (64 for the DP array,
 * segmentation_dp_arg.hh:65), t_cmp = std::less on the uint32 key.
 * Kept on purpose: the smp2 index at rmq.hh:77 equals smp1's and the tie order
 * of operator() (rmq.hh:96,98,104).
 * ---------------------------------------------------------------------------------------- */
struct fso_rmq {
	char const *base;
	size_t      stride;
	size_t      capacity;
	unsigned    bs;
	size_t      nlev;        /* allocated levels */
	size_t    **lev;         /* m_precalc */
	size_t     *lev_size;
	size_t      lev_cap;     /* entries per level */
};

static inline uint32_t rmq_key(fso_rmq const *r, size_t i)
{
	return *(uint32_t const *) (r->base + i * r->stride);
}

fso_rmq *fso_rmq_new(void const *key_base, size_t stride, size_t capacity, unsigned block_size)
{
	fso_rmq *r = (fso_rmq *) calloc(1, sizeof(*r));
	r->base = (char const *) key_base;
	r->stride = stride;
	r->capacity = capacity;
	r->bs = block_size;
	r->lev_cap = capacity / block_size + 2;
	size_t nlev = 1;
	while (((size_t) 1 << nlev) <= r->lev_cap) ++nlev;
	r->nlev = nlev + 1;
	r->lev = (size_t **) calloc(r->nlev, sizeof(size_t *));
	r->lev_size = (size_t *) calloc(r->nlev, sizeof(size_t));
	return r;
}

void fso_rmq_free(fso_rmq *r)
{
	if (!r) return;
	for (size_t i = 0; i < r->nlev; ++i) free(r->lev[i]);
	free(r->lev);
	free(r->lev_size);
	free(r);
}

static void rmq_push(fso_rmq *r, size_t level, size_t v)
{
	assert(level < r->nlev);
	if (!r->lev[level]) r->lev[level] = (size_t *) malloc(r->lev_cap * sizeof(size_t));
	assert(r->lev_size[level] < r->lev_cap);
	r->lev[level][r->lev_size[level]++] = v;
}

/* rmq.hh:108-118: std::min_element = first minimal element of [first,last). */
static size_t rmq_naive_min(fso_rmq const *r, size_t first, size_t last)
{
	size_t best = first;
	uint32_t bv = rmq_key(r, first);
	for (size_t i = first + 1; i < last; ++i)
	{
		uint32_t const v = rmq_key(r, i);
		if (v < bv) { bv = v; best = i; }
	}
	return best;
}

/* rmq.hh:61-81 */
void fso_rmq_update(fso_rmq *r, size_t const last_idx)
{
	unsigned const bs = r->bs;
	if (((1 + last_idx) & (bs - 1)) != 0)
		return;

	size_t const bnum = 1 + last_idx / bs;
	size_t const new_smp = rmq_naive_min(r, (bnum - 1) * bs, bnum * bs);
	uint32_t const new_val = rmq_key(r, new_smp);
	rmq_push(r, 0, new_smp);
	for (size_t pow2 = 1; ((size_t) 1u << pow2) <= bnum; ++pow2)
	{
		size_t const smp1 = r->lev[pow2 - 1][bnum - ((size_t) 1u << pow2)];
		size_t const smp2 = r->lev[pow2 - 1][bnum - (((size_t) 1u << pow2) - 1) - 1];   /* == smp1's index (rmq.hh:77) */
		size_t const smp = (rmq_key(r, smp2) < rmq_key(r, smp1)) ? smp2 : smp1;
		rmq_push(r, pow2, (new_val < rmq_key(r, smp)) ? new_smp : smp);
	}
}

/* sdsl::bits::hi: position of the most significant set bit (0 for x == 0). */
static inline unsigned bits_hi(uint64_t x) { return x ? 63u - (unsigned) __builtin_clzll(x) : 0u; }

/* rmq.hh:85-105 */
size_t fso_rmq_query(fso_rmq const *r, size_t const beg, size_t const end)
{
	unsigned const bs = r->bs;
	size_t const beg_block = beg / bs + 1;
	size_t const end_block = end / bs;

	if (beg_block >= end_block)
		return rmq_naive_min(r, beg, end);

	size_t const pow2 = bits_hi(end_block - beg_block);
	size_t const smp1 = r->lev[pow2][beg_block];
	size_t const smp2 = r->lev[pow2][end_block - ((size_t) 1u << pow2)];
	size_t smp = (rmq_key(r, smp2) < rmq_key(r, smp1)) ? smp2 : smp1;
	size_t const left_smp = rmq_naive_min(r, beg, beg_block * bs);
	smp = (rmq_key(r, left_smp) < rmq_key(r, smp)) ? left_smp : smp;

	if (end == end_block * bs)
		return smp;

	size_t const right_smp = rmq_naive_min(r, end_block * bs, end);
	return (rmq_key(r, right_smp) < rmq_key(r, smp)) ? right_smp : smp;
}

/* ------------------------------------------------------------------------------------------
 * calculate_segmentation_lp_dp_arg, segmentation_lp_context.cc:393-481.
 * ---------------------------------------------------------------------------------------- */
static inline int dp_less(fso_dp_arg const *x, fso_dp_arg const *y)
{
	return x->segment_max_size < y->segment_max_size;   /* segmentation_dp_arg.hh:53 */
}

void fso_dp_step(
	uint32_t const *values, uint32_t const *counts, size_t const npairs,
	fso_dp_arg const *dp, fso_rmq const *rmq,
	uint64_t const seq_count, uint64_t const segment_length, uint64_t const lb, uint64_t const text_pos,
	fso_dp_arg *min_arg)
{
	size_t it = 0;                                  /* :405 */
	size_t const end = npairs;                      /* :406 */
	assert(it != end);

	uint64_t segment_size_diff = counts[it];        /* :410 */
	uint64_t dp_lb = lb;                            /* :412 */
	uint64_t dp_rb = values[it];                    /* :413 */

	if (lb == dp_rb)                                /* :416 */
	{
		uint64_t const segment_size = seq_count - segment_size_diff;
		fso_dp_arg const current_arg = { lb, 1 + text_pos, (uint32_t) segment_size, (uint32_t) segment_size };
		if (dp_less(&current_arg, min_arg))         /* :420 */
			*min_arg = current_arg;

		++it;                                       /* :423 */
		assert(it != end);                          /* value text_pos+1 > 0 is always present */
		dp_rb = values[it];
		segment_size_diff += counts[it];
	}

	++it;                                           /* :428 */
	while (1)
	{
		if (it == end)                              /* :431 */
			break;

		dp_lb = dp_rb;                              /* :435 */
		dp_rb = values[it];
		uint64_t dp_rb_c = dp_rb;
		assert(0 != values[it]);
		assert(dp_lb < dp_rb);
		assert(dp_rb <= 1 + text_pos);

		if (text_pos + 2 - segment_length < dp_rb_c)    /* :444 */
			dp_rb_c = text_pos + 2 - segment_length;

		if (dp_lb < lb + segment_length)                /* :449 */
		{
			if (lb + segment_length < dp_rb_c)
				dp_lb = lb + segment_length;
			else
				goto continue_loop;
		}

		if (dp_lb < dp_rb_c)                            /* :458 */
		{
			uint64_t const dp_lb_tb = dp_lb - segment_length;
			uint64_t const dp_rb_tb = dp_rb_c - segment_length;
			size_t const idx = fso_rmq_query(rmq, dp_lb_tb, dp_rb_tb);      /* :465 */

			fso_dp_arg const *boundary_segment = &dp[idx];
			uint32_t const lhs = boundary_segment->segment_max_size;
			uint32_t const rhs = (uint32_t) (seq_count - segment_size_diff);

			fso_dp_arg const current_arg = { idx + segment_length, 1 + text_pos, lhs > rhs ? lhs : rhs, rhs };   /* :471 */
			if (dp_less(&current_arg, min_arg))
				*min_arg = current_arg;
		}

	continue_loop:
		segment_size_diff += counts[it];                /* :478 */
		++it;
	}
}

/* ------------------------------------------------------------------------------------------
 * libbio::pbwt::pbwt_context (un-vendored).  Semantics per SURVEY.md Appendix A step 2 and
 * Appendix B A1-A7: Durbin's one-pass update with a per-symbol running maximum (values are
 * identical to the RMQ formulation of founder_sequences.hh:56-65).
 * ---------------------------------------------------------------------------------------- */
struct fso_pbwt {
	uint8_t const *base;
	size_t   row_stride, col_stride;
	uint32_t m;
	uint64_t n;
	int      with_counts;
	uint64_t idx;            /* sequence_idx */
	uint32_t *a, *d;         /* input permutation / divergence (state before column idx) */
	uint32_t *a2, *d2;       /* output buffers */
	uint8_t  *symbuf;        /* gathered column symbols in pBWT order */
	/* divergence value counts: cnt[v] plus an ascending doubly linked list of live values;
	 * HEAD = n + 1, TAIL = n + 2. */
	uint32_t *cnt;
	uint32_t *nxt, *prv;
	uint64_t live;           /* number of live distinct values */
};

#define HEAD(p) ((uint32_t) ((p)->n + 1))
#define TAIL(p) ((uint32_t) ((p)->n + 2))

fso_pbwt *fso_pbwt_new(uint8_t const *base, size_t row_stride, size_t col_stride,
                       uint32_t m, uint64_t n, int with_counts)
{
	fso_pbwt *p = (fso_pbwt *) calloc(1, sizeof(*p));
	p->base = base; p->row_stride = row_stride; p->col_stride = col_stride;
	p->m = m; p->n = n; p->with_counts = with_counts;
	p->a  = (uint32_t *) malloc(sizeof(uint32_t) * m);
	p->d  = (uint32_t *) malloc(sizeof(uint32_t) * m);
	p->a2 = (uint32_t *) malloc(sizeof(uint32_t) * m);
	p->d2 = (uint32_t *) malloc(sizeof(uint32_t) * m);
	p->symbuf = (uint8_t *) malloc(m);
	if (with_counts)
	{
		p->cnt = (uint32_t *) calloc(n + 3, sizeof(uint32_t));
		p->nxt = (uint32_t *) malloc(sizeof(uint32_t) * (n + 3));
		p->prv = (uint32_t *) malloc(sizeof(uint32_t) * (n + 3));
	}
	return p;
}

void fso_pbwt_free(fso_pbwt *p)
{
	if (!p) return;
	free(p->a); free(p->d); free(p->a2); free(p->d2); free(p->symbuf);
	free(p->cnt); free(p->nxt); free(p->prv);
	free(p);
}

static void hist_link_tail(fso_pbwt *p, uint32_t v)
{
	uint32_t const t = TAIL(p), last = p->prv[t];
	p->nxt[last] = v; p->prv[v] = last; p->nxt[v] = t; p->prv[t] = v;
	++p->live;
}

static void hist_unlink(fso_pbwt *p, uint32_t v)
{
	uint32_t const a = p->prv[v], b = p->nxt[v];
	p->nxt[a] = b; p->prv[b] = a;
	--p->live;
}

static void hist_rebuild(fso_pbwt *p)
{
	/* from scratch out of p->d (used by prepare / set_state) */
	memset(p->cnt, 0, sizeof(uint32_t) * (p->n + 3));
	p->nxt[HEAD(p)] = TAIL(p); p->prv[TAIL(p)] = HEAD(p); p->live = 0;
	for (uint32_t i = 0; i < p->m; ++i) ++p->cnt[p->d[i]];
	for (uint64_t v = 0; v <= p->n; ++v)
		if (p->cnt[v]) hist_link_tail(p, (uint32_t) v);
}

void fso_pbwt_prepare(fso_pbwt *p)
{
	for (uint32_t i = 0; i < p->m; ++i) { p->a[i] = i; p->d[i] = 0; }
	p->idx = 0;
	if (p->with_counts)
	{
		/* cnt is all-zero only on first use; rebuild handles both */
		hist_rebuild(p);
	}
}

void fso_pbwt_set_state(fso_pbwt *p, uint32_t const *a, uint32_t const *d, uint64_t idx)
{
	memcpy(p->a, a, sizeof(uint32_t) * p->m);
	memcpy(p->d, d, sizeof(uint32_t) * p->m);
	p->idx = idx;
	if (p->with_counts) hist_rebuild(p);
}

void fso_pbwt_step(fso_pbwt *p)
{
	uint32_t const m = p->m;
	uint64_t const k = p->idx;
	assert(k < p->n);
	uint8_t const *col = p->base + k * p->col_stride;
	size_t const rs = p->row_stride;

	/* (1) gather symbols in pBWT order, count buckets */
	uint32_t count[256];
	memset(count, 0, sizeof(count));
	uint8_t *sym = p->symbuf;
	for (uint32_t i = 0; i < m; ++i)
	{
		uint8_t const c = col[(size_t) p->a[i] * rs];
		sym[i] = c;
		++count[c];
	}

	/* (2) exclusive offsets, ascending symbol order (A2); list of present symbols */
	uint32_t offset[256];
	uint8_t present[256];
	unsigned npresent = 0;
	{
		uint32_t acc = 0;
		for (unsigned c = 0; c < 256; ++c)
		{
			offset[c] = acc;
			acc += count[c];
			if (count[c]) present[npresent++] = (uint8_t) c;
		}
	}

	/* (3) stable scatter with per-symbol running max (A1).  run[c] = max d since the last row
	 * with symbol c; UINT32 sentinel "first" handled by seen[]. */
	uint32_t run[256];
	uint8_t  seen[256];
	for (unsigned j = 0; j < npresent; ++j) { run[present[j]] = 0; seen[present[j]] = 0; }
	uint32_t const first_val = (uint32_t) (k + 1);
	for (uint32_t i = 0; i < m; ++i)
	{
		uint32_t const di = p->d[i];
		for (unsigned j = 0; j < npresent; ++j)
		{
			uint8_t const c = present[j];
			if (run[c] < di) run[c] = di;
		}
		uint8_t const c = sym[i];
		uint32_t const dst = offset[c]++;
		p->a2[dst] = p->a[i];
		p->d2[dst] = seen[c] ? run[c] : first_val;
		seen[c] = 1;
		run[c] = 0;
	}

	/* (4) divergence value counts: increments first, then decrements (a value present on both
	 * sides must not be unlinked in between). */
	if (p->with_counts)
	{
		if (0 == p->cnt[first_val]) hist_link_tail(p, first_val);   /* k+1 is the largest value so far */
		for (uint32_t i = 0; i < m; ++i) ++p->cnt[p->d2[i]];
		for (uint32_t i = 0; i < m; ++i)
		{
			uint32_t const v = p->d[i];
			if (0 == --p->cnt[v]) hist_unlink(p, v);
		}
	}

	/* (6) swap, ++sequence_idx */
	uint32_t *t;
	t = p->a; p->a = p->a2; p->a2 = t;
	t = p->d; p->d = p->d2; p->d2 = t;
	++p->idx;
}

uint64_t fso_pbwt_idx(fso_pbwt const *p) { return p->idx; }
uint32_t const *fso_pbwt_a(fso_pbwt const *p) { return p->a; }
uint32_t const *fso_pbwt_d(fso_pbwt const *p) { return p->d; }

size_t fso_pbwt_counts(fso_pbwt const *p, uint32_t *values, uint32_t *counts)
{
	size_t k = 0;
	for (uint32_t v = p->nxt[HEAD(p)]; v != TAIL(p); v = p->nxt[v])
	{
		if (values) values[k] = v;
		if (counts) counts[k] = p->cnt[v];
		++k;
	}
	return k;
}

uint32_t fso_pbwt_unique_substring_count_lhs(fso_pbwt const *p, uint64_t lb)
{
	uint32_t c = 0;
	for (uint32_t i = 0; i < p->m; ++i) c += (p->d[i] > lb);
	return c;
}

size_t fso_pbwt_unique_substring_count_idxs_lhs(fso_pbwt const *p, uint64_t lb,
                                                uint32_t *first_idx, uint32_t *run_len)
{
	size_t runs = 0;
	for (uint32_t i = 0; i < p->m; ++i)
	{
		if (p->d[i] > lb)
		{
			first_idx[runs] = p->a[i];
			run_len[runs] = 1;
			++runs;
		}
		else
		{
			assert(runs);
			++run_len[runs - 1];
		}
	}
	return runs;
}

/* ------------------------------------------------------------------------------------------
 * Long path driver: segmentation_lp_context.cc:26-390, update_pbwt_task.cc:13-35.
 * ---------------------------------------------------------------------------------------- */
typedef struct sample_t { uint64_t idx; uint32_t *a, *d; } sample_t;

typedef struct pass2_task {
	sample_t  sample;            /* moved-in sample */
	uint64_t *rbs; size_t n_rbs; /* sorted right bounds */
	size_t    first_out;         /* index of the first snapshot in the global order */
} pass2_task;

typedef struct pass2_shared {
	uint8_t const *base; size_t rs, cs; uint32_t m; uint64_t n;
	pass2_task *tasks; size_t n_tasks;
	size_t next;                 /* next task to take */
	pthread_mutex_t mu;
	uint32_t *snap_a, *snap_d;   /* S x m */
	uint64_t cells;
} pass2_shared;

/* update_pbwt_task::execute, update_pbwt_task.cc:13-35 */
static void *pass2_worker(void *arg)
{
	pass2_shared *sh = (pass2_shared *) arg;
	fso_pbwt *ctx = fso_pbwt_new(sh->base, sh->rs, sh->cs, sh->m, sh->n, 0);
	uint64_t cells = 0;
	while (1)
	{
		pthread_mutex_lock(&sh->mu);
		size_t const ti = sh->next++;
		pthread_mutex_unlock(&sh->mu);
		if (ti >= sh->n_tasks) break;
		pass2_task *t = &sh->tasks[ti];
		fso_pbwt_set_state(ctx, t->sample.a, t->sample.d, t->sample.idx);
		for (size_t j = 0; j < t->n_rbs; ++j)
		{
			uint64_t const rb = t->rbs[j];
			while (ctx->idx < rb) { fso_pbwt_step(ctx); cells += sh->m; }      /* :19 */
			size_t const o = t->first_out + j;                                    /* :23-30 */
			memcpy(sh->snap_a + o * sh->m, ctx->a, sizeof(uint32_t) * sh->m);
			memcpy(sh->snap_d + o * sh->m, ctx->d, sizeof(uint32_t) * sh->m);
		}
	}
	fso_pbwt_free(ctx);
	pthread_mutex_lock(&sh->mu);
	sh->cells += cells;
	pthread_mutex_unlock(&sh->mu);
	return NULL;
}

int fso_segment_long(uint8_t const *base, size_t row_stride, size_t col_stride,
                     uint32_t m, uint64_t n, uint64_t L, uint64_t sample_rate,
                     int threads, int keep_dp, fso_result *out)
{
	memset(out, 0, sizeof(*out));
	if (!(L > 0 && 2 * L <= n && m > 0 && sample_rate > 0)) return -1;
	uint64_t const lb = 0, rb = n;

	double t0 = now_s();

	/* ---- pass 1: generate_traceback + parts 2-4, segmentation_lp_context.cc:26-188 ---- */
	fso_pbwt *ctx = fso_pbwt_new(base, row_stride, col_stride, m, n, 1);
	fso_pbwt_prepare(ctx);                                                          /* :34 */
	uint64_t const dp_size = n - L + 1;                                             /* :39 */
	fso_dp_arg *dp = (fso_dp_arg *) malloc(sizeof(fso_dp_arg) * dp_size);
	for (uint64_t i = 0; i < dp_size; ++i)
	{
		dp[i].lb = 0; dp[i].rb = 0; dp[i].segment_max_size = UINT32_MAX; dp[i].segment_size = UINT32_MAX;
	}
	fso_rmq *rmq = fso_rmq_new(&dp[0].segment_max_size, sizeof(fso_dp_arg), dp_size, 64);

	size_t samples_cap = (size_t) (n / sample_rate + 2), n_samples = 0;
	sample_t *samples = (sample_t *) calloc(samples_cap, sizeof(sample_t));
	uint32_t *vals = (uint32_t *) malloc(sizeof(uint32_t) * ((size_t) m + 1));
	uint32_t *cnts = (uint32_t *) malloc(sizeof(uint32_t) * ((size_t) m + 1));
	uint64_t pairs_total = 0;

	uint64_t const part2_limit = (2 * L < rb - L ? 2 * L : rb - L) - 1;            /* :72 */
	uint64_t const part3_limit = rb - L;                                            /* :113 */

	for (uint64_t k = 0; k < n; ++k)
	{
		if (0 == k % sample_rate)                                                   /* A4 */
		{
			assert(n_samples < samples_cap);
			sample_t *s = &samples[n_samples++];
			s->idx = k;
			s->a = (uint32_t *) malloc(sizeof(uint32_t) * m);
			s->d = (uint32_t *) malloc(sizeof(uint32_t) * m);
			memcpy(s->a, ctx->a, sizeof(uint32_t) * m);
			memcpy(s->d, ctx->d, sizeof(uint32_t) * m);
		}

		fso_pbwt_step(ctx);     /* afterwards ctx->idx == k + 1 and the counts are H_{k+1} (A5) */
		uint64_t const idx = k;

		if (idx < L - 1)
		{
			/* part 1 (:52-58): nothing */
		}
		else if (idx < part2_limit)
		{
			/* part 2 (:76-98) */
			uint32_t const first = ctx->nxt[HEAD(ctx)];
			uint64_t segment_size_diff = 0;
			if (first != TAIL(ctx) && 0 == first)
				segment_size_diff = ctx->cnt[0];
			uint64_t const tb_idx = idx + 1 - L;
			uint32_t const segment_size = (uint32_t) (m - segment_size_diff);
			fso_dp_arg const cur = { lb, 1 + idx, segment_size, segment_size };
			dp[tb_idx] = cur;
			fso_rmq_update(rmq, tb_idx);
		}
		else if (idx < part3_limit)
		{
			/* part 3 (:117-141) */
			size_t const np = fso_pbwt_counts(ctx, vals, cnts);
			pairs_total += np;
			fso_dp_arg min_arg = { lb, 1 + idx, m, m };
			fso_dp_step(vals, cnts, np, dp, rmq, m, L, lb, idx, &min_arg);
			uint64_t const tb_idx = idx + 1 - L;
			dp[tb_idx] = min_arg;
			fso_rmq_update(rmq, tb_idx);
		}
		/* part 4 loop (:156-163): nothing */
	}
	{
		/* part 4 tail (:165-183) */
		size_t const np = fso_pbwt_counts(ctx, vals, cnts);
		pairs_total += np;
		uint64_t const idx = ctx->idx;      /* == n */
		fso_dp_arg min_arg = { lb, idx, m, m };
		fso_dp_step(vals, cnts, np, dp, rmq, m, L, lb, idx - 1, &min_arg);
		uint64_t const tb_idx = dp_size - 1;
		assert(rb - L == tb_idx);
		dp[tb_idx] = min_arg;
	}
	double t1 = now_s();

	/* ---- follow_traceback, :191-224 ---- */
	size_t tb_cap = (size_t) (n / L + 2), S = 0;
	fso_dp_arg *tb = (fso_dp_arg *) malloc(sizeof(fso_dp_arg) * tb_cap);
	{
		uint64_t arg_idx = dp_size - 1;
		while (1)
		{
			fso_dp_arg const *cur = &dp[arg_idx];
			assert(S < tb_cap);
			tb[S++] = *cur;
			uint64_t const next_pos = cur->lb;
			if (0 == next_pos) break;
			assert(L <= next_pos);
			arg_idx = next_pos - L;
		}
		for (size_t i = 0; i < S / 2; ++i) { fso_dp_arg t = tb[i]; tb[i] = tb[S - 1 - i]; tb[S - 1 - i] = t; }
	}
	uint32_t const max_segment_size = tb[S - 1].segment_max_size;                  /* :218 */
	double t2 = now_s();

	out->max_segment_size = max_segment_size;
	out->n_dp_segments = S;
	out->traceback = tb;
	out->dp_size = dp_size;
	out->n_samples = n_samples;
	out->dp_pairs_total = pairs_total;
	out->t_pass1 = t1 - t0;
	out->t_traceback = t2 - t1;

	fso_pbwt_free(ctx);
	fso_rmq_free(rmq);
	free(vals); free(cnts);
	if (keep_dp) out->dp = dp; else free(dp);

	if (!(max_segment_size < m))                                                   /* generate_context.cc:192-200 */
	{
		for (size_t i = 0; i < n_samples; ++i) { free(samples[i].a); free(samples[i].d); }
		free(samples);
		out->status = 1;
		return 1;
	}

	/* ---- update_samples_to_traceback_positions, :229-316 ---- */
	pass2_task *tasks = (pass2_task *) calloc(n_samples + 1, sizeof(pass2_task));
	size_t n_tasks = 0;
	{
		size_t tb_it = 0;
		size_t i = 1;
		uint64_t *right_bounds = (uint64_t *) malloc(sizeof(uint64_t) * S);
		size_t n_rb = 0, first_out = 0;
		while (i < n_samples)
		{
			if (tb_it == S) break;                                                 /* :250 */
			while (tb_it < S && tb[tb_it].rb < samples[i].idx)                     /* :257 */
				right_bounds[n_rb++] = tb[tb_it++].rb;
			if (n_rb)                                                              /* :265 */
			{
				pass2_task *t = &tasks[n_tasks++];
				t->sample = samples[i - 1];
				samples[i - 1].a = NULL; samples[i - 1].d = NULL;                  /* moved (:276) */
				t->rbs = (uint64_t *) malloc(sizeof(uint64_t) * n_rb);
				memcpy(t->rbs, right_bounds, sizeof(uint64_t) * n_rb);
				t->n_rbs = n_rb;
				t->first_out = first_out;
				first_out += n_rb;
				n_rb = 0;
			}
			++i;
		}
		while (tb_it < S) right_bounds[n_rb++] = tb[tb_it++].rb;                   /* :289-291 */
		if (n_rb)
		{
			pass2_task *t = &tasks[n_tasks++];
			assert(samples[i - 1].a);                                              /* :296 */
			assert(samples[i - 1].idx <= right_bounds[0]);                         /* :298 */
			t->sample = samples[i - 1];
			samples[i - 1].a = NULL; samples[i - 1].d = NULL;
			t->rbs = (uint64_t *) malloc(sizeof(uint64_t) * n_rb);
			memcpy(t->rbs, right_bounds, sizeof(uint64_t) * n_rb);
			t->n_rbs = n_rb;
			t->first_out = first_out;
			first_out += n_rb;
		}
		free(right_bounds);
		assert(first_out == S);
	}
	for (size_t i = 0; i < n_samples; ++i) { free(samples[i].a); free(samples[i].d); }   /* :310 */
	free(samples);

	pass2_shared sh;
	memset(&sh, 0, sizeof(sh));
	sh.base = base; sh.rs = row_stride; sh.cs = col_stride; sh.m = m; sh.n = n;
	sh.tasks = tasks; sh.n_tasks = n_tasks;
	pthread_mutex_init(&sh.mu, NULL);
	sh.snap_a = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) m * S);
	sh.snap_d = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) m * S);
	if (threads < 1) threads = 1;
	if ((size_t) threads > n_tasks) threads = (int) (n_tasks ? n_tasks : 1);
	if (threads == 1)
		pass2_worker(&sh);
	else
	{
		pthread_t *th = (pthread_t *) malloc(sizeof(pthread_t) * threads);
		for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, pass2_worker, &sh);
		for (int i = 0; i < threads; ++i) pthread_join(th[i], NULL);
		free(th);
	}
	pthread_mutex_destroy(&sh.mu);
	for (size_t i = 0; i < n_tasks; ++i) { free(tasks[i].sample.a); free(tasks[i].sample.d); free(tasks[i].rbs); }
	free(tasks);
	double t3 = now_s();

	/* ---- find_segments_greedy, :335-390 ---- */
	fso_dp_arg *red = (fso_dp_arg *) malloc(sizeof(fso_dp_arg) * S);
	size_t *red_src = (size_t *) malloc(sizeof(size_t) * S);
	size_t n_red = 0;
	{
		uint64_t current_lb = 0;                                                    /* :346, lb == 0 */
		uint64_t prev_size = tb[0].segment_size;                                    /* :347 */
		size_t prev = 0;                                                            /* :348 */
		for (size_t j = 1; j < S; ++j)                                              /* :357 */
		{
			uint32_t const *dj = sh.snap_d + j * (size_t) m;
			uint32_t sample_size = 0;                                               /* :363, A7 */
			for (uint32_t r = 0; r < m; ++r) sample_size += (dj[r] > current_lb);
			if (sample_size <= max_segment_size)                                    /* :364 */
				prev_size = sample_size;
			else
			{
				fso_dp_arg const e = { current_lb, tb[prev].rb, UINT32_MAX, (uint32_t) prev_size };   /* :368, 3-arg ctor */
				red[n_red] = e; red_src[n_red] = prev; ++n_red;
				prev_size = tb[j].segment_size;                                     /* :369 */
				current_lb = tb[prev].rb;                                           /* :371 */
			}
			prev = j;                                                               /* :375 */
		}
		fso_dp_arg const e = { current_lb, tb[prev].rb, UINT32_MAX, (uint32_t) prev_size };           /* :380 */
		red[n_red] = e; red_src[n_red] = prev; ++n_red;
	}
	out->n_segments = n_red;
	out->reduced = red;
	out->a = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) m * n_red);
	out->d = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) m * n_red);
	for (size_t i = 0; i < n_red; ++i)
	{
		memcpy(out->a + i * (size_t) m, sh.snap_a + red_src[i] * (size_t) m, sizeof(uint32_t) * m);
		memcpy(out->d + i * (size_t) m, sh.snap_d + red_src[i] * (size_t) m, sizeof(uint32_t) * m);
	}
	free(red_src);
	free(sh.snap_a); free(sh.snap_d);
	double t4 = now_s();

	out->pass2_cells = sh.cells;
	out->pass2_threads = threads;
	out->t_pass2 = t3 - t2;
	out->t_merge = t4 - t3;
	out->status = 0;
	return 0;
}

void fso_result_free(fso_result *r)
{
	if (!r) return;
	free(r->traceback); free(r->reduced); free(r->a); free(r->d); free(r->dp);
	memset(r, 0, sizeof(*r));
}

/* segmentation_sp_context::process, segmentation_sp_context.cc:21-28 */
size_t fso_segment_short(uint8_t const *base, size_t row_stride, size_t col_stride,
                         uint32_t m, uint64_t n, uint32_t *first_idx, uint32_t *run_len)
{
	fso_pbwt *ctx = fso_pbwt_new(base, row_stride, col_stride, m, n, 0);
	fso_pbwt_prepare(ctx);
	while (ctx->idx < n) fso_pbwt_step(ctx);
	size_t const runs = fso_pbwt_unique_substring_count_idxs_lhs(ctx, 0, first_idx, run_len);
	fso_pbwt_free(ctx);
	return runs;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic founder-mosaic generator, SURVEY.md Appendix E.
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t x)
{
	x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
	x ^= x >> 27; x *= 0x94d049bb133111ebULL;
	x ^= x >> 31;
	return x;
}

static inline uint64_t synth_h(uint64_t seed, uint64_t tag, uint64_t r, uint64_t c)
{
	return mix64(seed ^ mix64(tag) ^ mix64(r * 0x9E3779B97F4A7C15ULL + c));
}

static char const ALPHA_DNA[]   = "ACGT";
static char const ALPHA_IUPAC[] = "ACGTRYSWKMBDHVN-";

static inline uint32_t synth_pick(uint32_t kind, uint64_t h)
{
	if (0 == kind) return (uint32_t) (h & 3u);
	uint32_t const u = (uint32_t) (h % 1000u);
	uint64_t const hi = h >> 32;
	if (u < 900u) return (uint32_t) (hi & 3u);
	return 4u + (uint32_t) (hi % 12u);
}

uint64_t fso_synth_threshold(double mu)
{
	if (mu <= 0.0) return 0;
	if (mu >= 1.0) return UINT64_MAX;
	return (uint64_t) (mu * 18446744073709551616.0);
}

uint8_t fso_synth_byte(fso_synth_spec const *s, uint64_t r, uint64_t c)
{
	uint32_t const sigma = (0 == s->kind) ? 4u : 16u;
	uint64_t const b = c / s->block_len;
	uint64_t const f = synth_h(s->seed, 2, r, b) % s->n_founders;
	uint32_t sym = synth_pick(s->kind, synth_h(s->seed, 1, f, c));
	if (synth_h(s->seed, 3, r, c) < s->mut_threshold)
		sym = (sym + 1u + (uint32_t) (synth_h(s->seed, 4, r, c) % (sigma - 1u))) % sigma;
	return (uint8_t) ((0 == s->kind) ? ALPHA_DNA[sym] : ALPHA_IUPAC[sym]);
}

void fso_synth_fill(fso_synth_spec const *s, uint32_t m, uint64_t c0, uint64_t c1,
                    uint8_t *out, size_t row_stride, size_t col_stride)
{
	for (uint64_t c = c0; c < c1; ++c)
		for (uint32_t r = 0; r < m; ++r)
			out[(size_t) r * row_stride + (size_t) (c - c0) * col_stride] = fso_synth_byte(s, r, c);
}
