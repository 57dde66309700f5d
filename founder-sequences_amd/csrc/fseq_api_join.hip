// fseq_api_join.hip -- the part of the C ABI (include/fseq.h) behind the segmentation: segment joining and the output files.
//
// replaces: join_context (join_context.cc:51-356) with greedy_matcher (greedy_matcher.cc:31-465), bipartite_matcher
// (bipartite_matcher.cc:17-151, create_segment_texts_task.cc, merge_segments_task.cc) and the random joiner, and the writers
// of --output-founders / --output-segments (join_context.cc:333-356, segmentation_dp_arg.cc:13-104).  Host C++ as in the
// reference (fseq_join.hpp), with the greedy joiner's data-parallel front -- class tables and co-occurrence edges -- on the
// device where the boundary states are (fseq_joinprep.hpp).  A translation unit of its own since round 4: nothing here
// touches the column kernels, and csrc/fseq_api.hip no longer carries the joiners.
#include "fseq_ctx.hpp"
#include "fseq_join.hpp"
#include "fseq_joinprep.hpp"
#include "fseq_joinbip.hpp"

using namespace fseq;

extern "C" {

int fseq_get_join_profile(fseq_ctx const *c, fseq_join_profile *out)
{
	if (!c || !out) return FSEQ_E_ARG;
	*out = c->jp;
	return FSEQ_OK;
}

int fseq_join_greedy(fseq_ctx *c, uint32_t *permutations)
{
	if (!c || !permutations || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	if (c->segments.empty()) return fail(c, FSEQ_E_ARG, "no segments to join (segmentation failed or was not run)");
	if (c->sh.on) return fail(c, FSEQ_E_UNSUPPORTED, "sharded run: collect the boundary states (fseq_boundary_state on their owners) and use fseq_greedy_match_host");
	(void) hipSetDevice(c->p.device);
	size_t const m = c->p.m, S = c->segments.size();
	double const t0 = now_ms();
	uint32_t const X = c->res.max_segment_size;
	// class tables and co-occurrence edges where the boundary states are (fseq_joinprep.hpp); the host hands out
	// the copies and draws the edges (the serial part of greedy_matcher.cc).  Falls through to the all-host joiner
	// below when the edge array cannot be allocated or the tables come back implausible.
	while (X <= JP_MAX_CLASSES && m <= 0xFFFFFFFFull && !c->tune.join_host)
	{
		hipStream_t st = c->stream;
		uint16_t *d_of = nullptr;
		uint32_t *d_rep = nullptr, *d_size = nullptr, *d_count = nullptr, *d_off = nullptr, *d_ne = nullptr;
		uint64_t *d_rb = nullptr;
		uint2 *d_edges = nullptr;
		unsigned long long *d_cursor = nullptr;
		// (offsets into the edge array are 32-bit words on the way to the host: the capacity stays below 2^32)
		uint64_t const cap_total = std::min<uint64_t>((uint64_t) (S > 1 ? S - 1 : 0) * std::min<uint64_t>(m, (uint64_t) X * X) + 1, 0xFFFFFFFFull);
		int rc;
		auto cleanup = [&]() { dev_free(c, &d_of); dev_free(c, &d_rep); dev_free(c, &d_size); dev_free(c, &d_count); dev_free(c, &d_off); dev_free(c, &d_ne);
		                       dev_free(c, &d_rb); dev_free(c, &d_edges); dev_free(c, &d_cursor); };
		if ((rc = dev_alloc(c, &d_of, S * m)) || (rc = dev_alloc(c, &d_rep, S * X)) || (rc = dev_alloc(c, &d_size, S * X)) || (rc = dev_alloc(c, &d_count, S)) ||
		    (rc = dev_alloc(c, &d_off, S)) || (rc = dev_alloc(c, &d_ne, S)) || (rc = dev_alloc(c, &d_rb, S)) || (rc = dev_alloc(c, &d_edges, cap_total)) ||
		    (rc = dev_alloc(c, &d_cursor, 1)))
		{
			cleanup();
			if (rc == FSEQ_E_OOM) { c->err.clear(); break; }       // no room for the device front: the host joiner needs none
			return rc;
		}
		std::vector<uint64_t> rbs(S);
		for (size_t i = 0; i < S; ++i) rbs[i] = c->segments[i].rb;
		hipError_t e = hipMemcpyAsync(d_rb, rbs.data(), S * 8, hipMemcpyHostToDevice, st);
		if (e == hipSuccess) e = hipMemsetAsync(d_cursor, 0, 8, st);
		if (e == hipSuccess) e = hipMemsetAsync(d_rep, 0, S * X * 4, st);
		size_t const lds = (size_t) X * X * 4;
		if (e == hipSuccess) e = allow_lds(k_join_edges, lds);
		if (e != hipSuccess) { cleanup(); return fail(c, FSEQ_E_HIP, "join preparation", e); }
		hipLaunchKernelGGL(k_join_classes, dim3((uint32_t) S), dim3(JP_T), 0, st, c->d_snap_a, c->d_snap_d, d_rb, (uint32_t) m, X, d_of, d_rep, d_size, d_count);
		if (S > 1)
			hipLaunchKernelGGL(k_join_edges, dim3((uint32_t) (S - 1)), dim3(JP_T), lds, st, d_of, d_count, (uint32_t) m, X, d_edges, cap_total, d_off, d_ne, d_cursor);
		std::vector<uint32_t> count(S), rep(S * X), size(S * X), off(S), ne(S);
		unsigned long long total = 0;
		e = hipMemcpyAsync(count.data(), d_count, S * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipMemcpyAsync(rep.data(), d_rep, S * X * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipMemcpyAsync(size.data(), d_size, S * X * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess && S > 1) e = hipMemcpyAsync(off.data(), d_off, (S - 1) * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess && S > 1) e = hipMemcpyAsync(ne.data(), d_ne, (S - 1) * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipMemcpyAsync(&total, d_cursor, 8, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		if (e == hipSuccess) e = hipGetLastError();
		bool sane = e == hipSuccess && total < cap_total && total <= 0xFFFFFFFFull;
		for (size_t i = 0; sane && i < S; ++i) sane = count[i] >= 1 && count[i] <= X;
		std::vector<uint32_t> edge_words(sane ? 2 * (size_t) total + 2 : 2);
		if (sane && total) e = hipMemcpy(edge_words.data(), d_edges, (size_t) total * 8, hipMemcpyDeviceToHost);
		cleanup();
		if (e != hipSuccess) return fail(c, FSEQ_E_HIP, "join preparation", e);
		if (!sane) break;                                         // (the host joiner builds its own tables from the boundary states)
		double const t1 = now_ms();
		JoinProfile prof;
		greedy_match_prepared(c->p.m, X, S, count.data(), rep.data(), size.data(), edge_words.data(), off.data(), ne.data(), permutations, &prof);
		c->jp = fseq_join_profile{t1 - t0, prof.ms_classes, prof.ms_edges, prof.ms_draw, now_ms() - t0,
		                          (uint64_t) S * (2ull * X + 3) * 4 + (uint64_t) total * 8};
		return FSEQ_OK;
	}
	std::vector<uint32_t> A(S * m), D(S * m);
	HIP_TRY(c, hipMemcpy(A.data(), c->d_snap_a, S * m * 4, hipMemcpyDeviceToHost));
	HIP_TRY(c, hipMemcpy(D.data(), c->d_snap_d, S * m * 4, hipMemcpyDeviceToHost));
	double const t1 = now_ms();
	std::vector<JoinSegment> segs(S);
	for (size_t i = 0; i < S; ++i) { segs[i].lb = c->segments[i].lb; segs[i].rb = c->segments[i].rb; }
	JoinProfile prof;
	greedy_match(c->p.m, c->res.max_segment_size, segs, A.data(), D.data(), permutations, &prof);
	c->jp = fseq_join_profile{t1 - t0, prof.ms_classes, prof.ms_edges, prof.ms_draw, now_ms() - t0, (uint64_t) S * m * 8ull};
	return FSEQ_OK;
}

// boundary states of all merged segments on the host (what join_context reads from the pbwt samples)
static int fetch_boundary_states(fseq_ctx *c, std::vector<uint32_t> &A, std::vector<uint32_t> &D, std::vector<JoinSegment> &segs)
{
	if (!c->have_result || c->res.short_path) return FSEQ_E_ARG;
	if (c->segments.empty()) return fail(c, FSEQ_E_ARG, "no segments to join (segmentation failed or was not run)");
	if (c->sh.on) return fail(c, FSEQ_E_UNSUPPORTED, "sharded run: collect the boundary states (fseq_boundary_state on their owners) and use the *_match_host entry points");
	(void) hipSetDevice(c->p.device);
	size_t const m = c->p.m, S = c->segments.size();
	double const t0 = now_ms();
	A.resize(S * m); D.resize(S * m);
	HIP_TRY(c, hipMemcpy(A.data(), c->d_snap_a, S * m * 4, hipMemcpyDeviceToHost));
	HIP_TRY(c, hipMemcpy(D.data(), c->d_snap_d, S * m * 4, hipMemcpyDeviceToHost));
	c->jp = fseq_join_profile{now_ms() - t0, 0, 0, 0, 0, (uint64_t) S * m * 8ull};
	segs.resize(S);
	for (size_t i = 0; i < S; ++i) { segs[i].lb = c->segments[i].lb; segs[i].rb = c->segments[i].rb; }
	return FSEQ_OK;
}

int fseq_join_bipartite(fseq_ctx *c, uint32_t *permutations)
{
	if (!c || !permutations) return FSEQ_E_ARG;
	if (!c->have_result || c->res.short_path) return FSEQ_E_ARG;
	if (c->segments.empty()) return fail(c, FSEQ_E_ARG, "no segments to join (segmentation failed or was not run)");
	size_t const m = c->p.m, S = c->segments.size();
	uint32_t const X = c->res.max_segment_size;
	// [r5] texts, intersection weights, the matchings and their chaining where the boundary states are (fseq_joinbip.hpp);
	// falls through to the host joiner when the tables do not fit a workgroup's LDS or cannot be allocated
	bool tiled = true;                                           // (lb of a segment = rb of the one in front: what k_join_classes goes by)
	for (size_t i = 0; tiled && i < S; ++i) tiled = c->segments[i].lb == (i ? c->segments[i - 1].rb : 0u);
	while (tiled && !c->sh.on && X >= 1 && X <= JP_MAX_CLASSES && m <= 0xFFFFFFFFull && S <= 0xFFFFFFFFull && !c->tune.join_host)
	{
		(void) hipSetDevice(c->p.device);
		double const t0 = now_ms();
		hipStream_t st = c->stream;
		uint16_t *d_of = nullptr, *d_tpos = nullptr, *d_src = nullptr, *d_match = nullptr;
		uint32_t *d_rep = nullptr, *d_size = nullptr, *d_count = nullptr, *d_min = nullptr, *d_reprow = nullptr, *d_perm = nullptr;
		uint64_t *d_rb = nullptr;
		int rc;
		auto cleanup = [&]() { dev_free(c, &d_of); dev_free(c, &d_tpos); dev_free(c, &d_src); dev_free(c, &d_match); dev_free(c, &d_rep); dev_free(c, &d_size);
		                       dev_free(c, &d_count); dev_free(c, &d_min); dev_free(c, &d_reprow); dev_free(c, &d_perm); dev_free(c, &d_rb); };
		if ((rc = dev_alloc(c, &d_of, S * m)) || (rc = dev_alloc(c, &d_tpos, S * X)) || (rc = dev_alloc(c, &d_src, S * X)) || (rc = dev_alloc(c, &d_match, S * X)) ||
		    (rc = dev_alloc(c, &d_rep, S * X)) || (rc = dev_alloc(c, &d_size, S * X)) || (rc = dev_alloc(c, &d_count, S)) || (rc = dev_alloc(c, &d_min, S * X)) ||
		    (rc = dev_alloc(c, &d_reprow, S * X)) || (rc = dev_alloc(c, &d_perm, S * X)) || (rc = dev_alloc(c, &d_rb, S)))
		{
			cleanup();
			if (rc == FSEQ_E_OOM) { c->err.clear(); break; }
			return rc;
		}
		std::vector<uint64_t> rbs(S);
		for (size_t i = 0; i < S; ++i) rbs[i] = c->segments[i].rb;
		hipError_t e = hipMemcpyAsync(d_rb, rbs.data(), S * 8, hipMemcpyHostToDevice, st);
		size_t const lds_match = bip_match_lds_bytes(X), lds_chain = bip_chain_lds_bytes(X);
		if (e == hipSuccess) e = allow_lds(k_bip_match, lds_match);
		if (e == hipSuccess) e = allow_lds(k_bip_chain, lds_chain);
		if (e != hipSuccess) { cleanup(); return fail(c, FSEQ_E_HIP, "bipartite join preparation", e); }
		// (a segment's classes: the rows that agree on [lb, rb), lb = the segment in front's rb -- the merged segments tile the columns)
		hipLaunchKernelGGL(k_join_classes, dim3((uint32_t) S), dim3(JP_T), 0, st, c->d_snap_a, c->d_snap_d, d_rb, (uint32_t) m, X, d_of, d_rep, d_size, d_count);
		hipLaunchKernelGGL(k_bip_minrow, dim3((uint32_t) S), dim3(JP_T), 0, st, d_of, (uint32_t) m, X, d_min);
		hipLaunchKernelGGL(k_bip_texts, dim3((uint32_t) S), dim3(64), 0, st, d_count, d_size, d_min, (uint32_t) m, X, d_tpos, d_src, d_reprow);
		if (S > 1)
			hipLaunchKernelGGL(k_bip_match, dim3((uint32_t) (S - 1)), dim3(64), lds_match, st, d_of, d_count, d_tpos, d_src, (uint32_t) m, X, d_match, (long long *) nullptr);
		hipLaunchKernelGGL(k_bip_chain, dim3(1), dim3(JB_CHAIN_T), lds_chain, st, d_match, d_reprow, (uint32_t) S, X, d_perm);
		std::vector<uint32_t> count(S);
		e = hipMemcpyAsync(count.data(), d_count, S * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipMemcpyAsync(permutations, d_perm, S * X * 4, hipMemcpyDeviceToHost, st);
		if (e == hipSuccess) e = hipStreamSynchronize(st);
		if (e == hipSuccess) e = hipGetLastError();
		cleanup();
		if (e != hipSuccess) return fail(c, FSEQ_E_HIP, "bipartite join", e);
		bool sane = true;
		for (size_t i = 0; sane && i < S; ++i) sane = count[i] >= 1 && count[i] <= X;
		if (!sane) break;                                         // (implausible class tables: the host joiner builds its own)
		double const t1 = now_ms();
		c->jp = fseq_join_profile{0.0, 0.0, 0.0, t1 - t0, t1 - t0, (uint64_t) S * X * 4ull + (uint64_t) S * 4ull};
		return FSEQ_OK;
	}
	std::vector<uint32_t> A, D;
	std::vector<JoinSegment> segs;
	int const rc = fetch_boundary_states(c, A, D, segs);
	if (rc) return rc;
	double const t0 = now_ms();
	bipartite_match(c->p.m, c->res.max_segment_size, segs, A.data(), D.data(), permutations);
	c->jp.ms_draw = now_ms() - t0; c->jp.ms_total = c->jp.ms_d2h + c->jp.ms_draw;
	return FSEQ_OK;
}

int fseq_join_random(fseq_ctx *c, uint32_t seed, uint32_t *permutations)
{
	if (!c || !permutations) return FSEQ_E_ARG;
	std::vector<uint32_t> A, D;
	std::vector<JoinSegment> segs;
	int const rc = fetch_boundary_states(c, A, D, segs);
	if (rc) return rc;
	double const t0 = now_ms();
	random_join(c->p.m, c->res.max_segment_size, segs, A.data(), D.data(), seed, permutations);
	c->jp.ms_draw = now_ms() - t0; c->jp.ms_total = c->jp.ms_d2h + c->jp.ms_draw;
	return FSEQ_OK;
}

int fseq_bipartite_match_host(uint32_t m, uint32_t max_segment_size, uint64_t n_segments, uint64_t const *lb, uint64_t const *rb,
                              uint32_t const *a, uint32_t const *d, uint32_t *permutations, int64_t *weights)
{
	if (!m || !max_segment_size || !lb || !rb || !a || !d || !permutations) return FSEQ_E_ARG;
	std::vector<JoinSegment> segs(n_segments);
	for (uint64_t i = 0; i < n_segments; ++i) { segs[i].lb = lb[i]; segs[i].rb = rb[i]; }
	std::vector<int64_t> w;
	bipartite_match(m, max_segment_size, segs, a, d, permutations, nullptr, &w);
	if (weights) std::copy(w.begin(), w.end(), weights);
	return FSEQ_OK;
}

int fseq_random_join_host(uint32_t m, uint32_t max_segment_size, uint64_t n_segments, uint64_t const *lb, uint64_t const *rb,
                          uint32_t const *a, uint32_t const *d, uint32_t seed, uint32_t *permutations)
{
	if (!m || !max_segment_size || !lb || !rb || !a || !d || !permutations) return FSEQ_E_ARG;
	std::vector<JoinSegment> segs(n_segments);
	for (uint64_t i = 0; i < n_segments; ++i) { segs[i].lb = lb[i]; segs[i].rb = rb[i]; }
	random_join(m, max_segment_size, segs, a, d, seed, permutations);
	return FSEQ_OK;
}

static int write_segments_impl(fseq_ctx *c, uint8_t const *const *rows, int joining, uint32_t const *A_, uint32_t const *D_,
                               std::vector<JoinSegment> const &segs, char const *path);

int fseq_write_segments(fseq_ctx *c, uint8_t const *const *rows, int joining, char const *path)
{
	if (!c || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	if (joining != FSEQ_JOIN_GREEDY && !rows) return FSEQ_E_ARG;
	std::vector<uint32_t> A, D;
	std::vector<JoinSegment> segs;
	if (joining != FSEQ_JOIN_GREEDY)
	{
		int const rc = fetch_boundary_states(c, A, D, segs);
		if (rc) return rc;
	}
	return write_segments_impl(c, rows, joining, A.data(), D.data(), segs, path);
}

// the same with the boundary states supplied by the caller (a sharded run: collected from their owners)
int fseq_write_segments_host(fseq_ctx *c, uint8_t const *const *rows, int joining, uint32_t const *a, uint32_t const *d, char const *path)
{
	if (!c || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	if (joining != FSEQ_JOIN_GREEDY && (!rows || !a || !d)) return FSEQ_E_ARG;
	std::vector<JoinSegment> segs;
	if (joining != FSEQ_JOIN_GREEDY)
	{
		segs.resize(c->segments.size());
		for (size_t i = 0; i < segs.size(); ++i) { segs[i].lb = c->segments[i].lb; segs[i].rb = c->segments[i].rb; }
	}
	return write_segments_impl(c, rows, joining, a, d, segs, path);
}

static int write_segments_impl(fseq_ctx *c, uint8_t const *const *rows, int joining, uint32_t const *A_, uint32_t const *D_,
                               std::vector<JoinSegment> const &segs, char const *path)
{
	FILE *f = (path && strcmp(path, "-") != 0) ? fopen(path, "wb") : stdout;
	if (!f) return fail(c, FSEQ_E_ARG, "cannot open the segments output file");
	size_t const m = c->p.m, S = segs.size();
	uint32_t const X = c->res.max_segment_size;
	if (FSEQ_JOIN_BIPARTITE == joining)
	{
		// segmentation_dp_arg.cc:59-104
		fputs("SEGMENT\tLB\tRB\tSIZE\tSUBSEQUENCE\tSEQUENCES\tCOPIED_FROM\n", f);
		for (size_t s = 0; s < S; ++s)
		{
			uint32_t const *a = A_ + s * m, *d = D_ + s * m;
			auto const texts = create_segment_texts((uint32_t) m, X, a, prepare_copy_numbers((uint32_t) m, X, segs[s].lb, a, d, true));
			for (size_t i = 0; i < texts.size(); ++i)
			{
				SegmentText const &tx = texts[i];
				fprintf(f, "%zu\t%llu\t%llu\t%u\t", s, (unsigned long long) segs[s].lb, (unsigned long long) segs[s].rb, c->segments[s].segment_size);
				uint32_t const rep = texts[tx.row_number(i)].sequence_indices.front();    // segment_text::write_text
				fwrite(rows[rep] + segs[s].lb, 1, segs[s].rb - segs[s].lb, f);
				fputc('\t', f);
				for (size_t k = 0; k < tx.sequence_indices.size(); ++k) fprintf(f, k ? ",%u" : "%u", tx.sequence_indices[k]);
				if (tx.is_copied()) fprintf(f, "\t%zu\n", tx.copied_from); else fputs("\t-\n", f);
			}
		}
	}
	else
	{
		// segmentation_dp_arg.cc:13-56; with greedy joining the copy-number matrix is empty
		fputs("SEGMENT\tLB\tRB\tSIZE\tSUBSEQUENCE_NUMBER\tCOPY_NUMBER\tSUBSEQUENCE\n", f);
		for (size_t s = 0; FSEQ_JOIN_RANDOM == joining && s < S; ++s)
		{
			uint32_t const *a = A_ + s * m, *d = D_ + s * m;
			auto const cn = prepare_copy_numbers((uint32_t) m, X, segs[s].lb, a, d, false);
			uint32_t prev = 0;
			for (auto const &x : cn)
			{
				fprintf(f, "%zu\t%llu\t%llu\t%u\t%u\t%u\t", s, (unsigned long long) segs[s].lb, (unsigned long long) segs[s].rb, c->segments[s].segment_size,
				        x.substring_idx, x.copy_number - prev);
				prev = x.copy_number;
				fwrite(rows[x.substring_idx] + segs[s].lb, 1, segs[s].rb - segs[s].lb, f);
				fputc('\n', f);
			}
		}
	}
	fflush(f);
	if (f != stdout) fclose(f);
	return FSEQ_OK;
}

int fseq_greedy_match_host(uint32_t m, uint32_t max_segment_size, uint64_t n_segments, uint64_t const *lb, uint64_t const *rb,
                           uint32_t const *a, uint32_t const *d, uint32_t *permutations)
{
	if (!m || !max_segment_size || !lb || !rb || !a || !d || !permutations) return FSEQ_E_ARG;
	std::vector<JoinSegment> segs(n_segments);
	for (uint64_t i = 0; i < n_segments; ++i) { segs[i].lb = lb[i]; segs[i].rb = rb[i]; }
	greedy_match(m, max_segment_size, segs, a, d, permutations);
	return FSEQ_OK;
}

int fseq_write_founders(fseq_ctx *c, uint8_t const *const *rows, uint32_t const *permutations, char const *path)
{
	if (!c || !rows || !permutations || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	FILE *f = (path && strcmp(path, "-") != 0) ? fopen(path, "wb") : stdout;
	if (!f) return fail(c, FSEQ_E_ARG, "cannot open the founders output file");
	size_t const X = c->res.max_segment_size, S = c->segments.size();
	// join_context.cc:341-356: line r = the segments' substrings of the rows permutations[s][r], one after the other.
	// The lines are put together in memory -- a batch of them at a time, on a few host threads: a line is S pieces of a
	// few hundred bytes from S different input rows -- and go out in one write per batch (one fwrite per piece was 41 of a
	// drop-in BASELINE C3 run's ~150 ms, profiles/r03_e2e_C3_greedy.json).
	size_t const line = (size_t) c->p.n + 1;
	size_t const batch = std::max<size_t>(1, std::min<size_t>(X, (size_t) (256u << 20) / line));
	std::vector<char> buf;
	try { buf.resize(batch * line); } catch (std::bad_alloc const &) { if (f != stdout) fclose(f); return fail(c, FSEQ_E_OOM, "founders output buffer"); }
	bool ok = true;
	for (size_t r0 = 0; r0 < X && ok; r0 += batch)
	{
		size_t const r1 = std::min(X, r0 + batch);
		unsigned const nth = (unsigned) std::max<size_t>(1, std::min<size_t>({(size_t) std::thread::hardware_concurrency(), (size_t) 8, r1 - r0}));
		auto work = [&](unsigned t) {
			for (size_t row = r0 + t; row < r1; row += nth)
			{
				char *out = buf.data() + (row - r0) * line;
				for (size_t s = 0; s < S; ++s)
				{
					fseq_segment const &sg = c->segments[s];
					memcpy(out + sg.lb, rows[permutations[s * X + row]] + sg.lb, sg.rb - sg.lb);
				}
				out[line - 1] = '\n';
			}
		};
		std::vector<std::thread> ths;
		for (unsigned t = 1; t < nth; ++t) ths.emplace_back(work, t);
		work(0);
		for (auto &th : ths) th.join();
		ok = fwrite(buf.data(), 1, (r1 - r0) * line, f) == (r1 - r0) * line;
	}
	fflush(f);
	if (f != stdout) fclose(f);
	if (!ok) return fail(c, FSEQ_E_ARG, "writing the founders output file failed");
	return FSEQ_OK;
}


// [r5] --output-founders straight from the alignment the device holds (k_founders, fseq_joinprep.hpp): no host rows needed.
// The lines are put together on the device, a batch of rows (<= 256 MB) at a time, and leave in one copy and one write per batch.
int fseq_write_founders_device(fseq_ctx *c, uint32_t const *permutations, char const *path)
{
	if (!c || !permutations || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	if (!c->have_input || !c->d_msa) return fail(c, FSEQ_E_ARG, "no alignment resident on the device");
	if (c->sh.on) return fail(c, FSEQ_E_UNSUPPORTED, "sharded run: a rank holds its own columns only (write the founders from host rows: fseq_write_founders)");
	(void) hipSetDevice(c->p.device);
	size_t const X = c->res.max_segment_size, S = c->segments.size();
	if (!X || !S) return fail(c, FSEQ_E_ARG, "no segments to write");
	FILE *f = (path && strcmp(path, "-") != 0) ? fopen(path, "wb") : stdout;
	if (!f) return fail(c, FSEQ_E_ARG, "cannot open the founders output file");
	hipStream_t st = c->stream;
	size_t const line = (size_t) c->p.n + 1;
	size_t const batch = std::max<size_t>(1, std::min<size_t>(X, (size_t) (256u << 20) / line));
	uint32_t *d_perm = nullptr;
	uint64_t *d_seg = nullptr;
	uint8_t *d_lut = nullptr, *d_out = nullptr, *h_out = nullptr;
	int rc = FSEQ_OK;
	auto cleanup = [&]() {
		dev_free(c, &d_perm); dev_free(c, &d_seg); dev_free(c, &d_lut); dev_free(c, &d_out);
		if (h_out) (void) hipHostFree(h_out);
		if (f != stdout) fclose(f);
	};
	if ((rc = dev_alloc(c, &d_perm, S * X)) || (rc = dev_alloc(c, &d_seg, 2 * S)) || (rc = dev_alloc(c, &d_lut, 256)) || (rc = dev_alloc(c, &d_out, batch * line))) { cleanup(); return rc; }
	if (hipHostMalloc(reinterpret_cast<void **>(&h_out), batch * line, hipHostMallocDefault) != hipSuccess) { h_out = nullptr; cleanup(); return fail(c, FSEQ_E_OOM, "founders output buffer"); }
	std::vector<uint64_t> seg(2 * S);
	for (size_t s = 0; s < S; ++s) { seg[s] = c->segments[s].lb; seg[S + s] = c->segments[s].rb; }
	bool ok = true;
	do {
		if (hipMemcpyAsync(d_perm, permutations, S * X * 4, hipMemcpyHostToDevice, st) != hipSuccess) { ok = false; break; }
		if (hipMemcpyAsync(d_seg, seg.data(), 2 * S * 8, hipMemcpyHostToDevice, st) != hipSuccess) { ok = false; break; }
		if (hipMemcpyAsync(d_lut, c->code_to_byte, 256, hipMemcpyHostToDevice, st) != hipSuccess) { ok = false; break; }
		for (size_t r0 = 0; r0 < X && ok; r0 += batch)
		{
			size_t const nr = std::min(batch, X - r0);
			uint32_t const rows_per_wg = 16;
			hipLaunchKernelGGL(k_founders, dim3((uint32_t) S, (uint32_t) ((nr + rows_per_wg - 1) / rows_per_wg)), dim3(256), 0, st, c->d_msa, c->ld, c->p.m, (uint64_t) c->p.n, c->bsh,
			                   d_perm, (uint32_t) X, d_seg, d_seg + S, (uint32_t) S, (uint32_t) r0, (uint32_t) nr, rows_per_wg, d_lut, d_out);
			if (hipMemcpyAsync(h_out, d_out, nr * line, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { ok = false; break; }
			ok = fwrite(h_out, 1, nr * line, f) == nr * line;
		}
	} while (false);
	if (ok && hipGetLastError() != hipSuccess) ok = false;
	fflush(f);
	cleanup();
	if (!ok) return fail(c, FSEQ_E_HIP, "writing the founders from the device failed");
	return FSEQ_OK;
}

} // extern "C"
