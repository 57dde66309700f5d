// fseq_join.hpp -- host-side segment joining: the greedy, bipartite-matching
// and random joiners and what the segments file needs from them.  Stays on the host, as in the reference (north_star: "segment joining ... stays on host
// until it shows in the profile").  Restated from founder-sequences/greedy_matcher.cc:31-465 and
// founder-sequences/join_context.cc:191-229,333-356; citations relative to /root/reference.
//
// Assumption carried from SURVEY.md Appendix B A8 (libbio is un-vendored): lb::radix_sort<true>::sort
// orders by descending key and is stable; lb::radix_sort<>::sort orders by ascending key.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <utility>
#include <vector>
#include <atomic>
#include <chrono>
#include <thread>
#include <algorithm>

namespace fseq {

struct JoinSegment {
	uint64_t lb, rb;
};

// ------------------------------------------------------------------------------------------------
// Greedy joiner (row N1).  Behaviour of greedy_matcher::match (founder-sequences/greedy_matcher.cc:204-465,
// entered through join_context.cc:211-229) in flat arrays: one class table per segment, copy numbers by a counting
// sort of the class sizes, the (lhs class, rhs class) co-occurrence edges as one array ordered by (count
// descending, pair ascending) that is compacted round after round, slot FIFOs as linked lists over one `next`
// array.  What must come out the same (oracle/greedy_oracle.py restates it independently; tests/test_join.py):
//   * a row starts a new class at pBWT position i iff seg_start < d[i] (:51); the class's representative is the
//     row at its first position (:54)
//   * copies (:81-105): every class one slot, the X - #classes spare slots handed out in rounds over the classes
//     in descending size (ties: earlier class first, A8), ceil(size / m * spare) at a time (double arithmetic)
//   * first segment (:128-160, :241-262): classes in pBWT order take consecutive slots
//   * later segments (:266-463): edges are tried in the order above; an edge is drawn while both of its classes
//     have copies left (the slot comes off the front of the lhs class's FIFO and goes to the back of the rhs
//     class's), an edge that cannot be drawn is dropped; rounds repeat while one drew anything (:353-408); what is
//     left is paired off by two ascending pointers (:412-439)
// ------------------------------------------------------------------------------------------------
namespace join_detail {

constexpr uint32_t NIL = 0xFFFFFFFFu;

// the distinct substrings of one segment, in pBWT order
struct ClassTable {
	uint32_t count = 0;
	std::vector<uint32_t> rep;        // [class] representative row
	std::vector<uint32_t> size;       // [class] rows in the class
	std::vector<uint32_t> of_row;     // [row] class
	std::vector<uint32_t> copies;     // [class] slots the class gets

	void build(uint32_t m, uint64_t seg_start, uint32_t const *a, uint32_t const *d, uint32_t X)
	{
		rep.clear(); size.clear();
		of_row.assign(m, NIL);
		for (uint32_t i = 0; i < m; ++i)
		{
			if (seg_start < d[i]) { rep.push_back(a[i]); size.push_back(0); }
			of_row[a[i]] = (uint32_t) rep.size() - 1u;
			++size.back();
		}
		count = (uint32_t) rep.size();
		// classes by descending size, earlier class first among equals: counting sort on the size
		std::vector<uint32_t> start(m + 2, 0), order(count);
		for (uint32_t c = 0; c < count; ++c) ++start[m - size[c] + 1];
		for (uint32_t v = 0; v <= m; ++v) start[v + 1] += start[v];
		for (uint32_t c = 0; c < count; ++c) order[start[m - size[c]]++] = c;
		copies.assign(count, 1);
		size_t const spare = X - count;
		size_t left = spare;
		while (left)
			for (uint32_t k = 0; k < count && left; ++k)
			{
				uint32_t const c = order[k];
				size_t const give = std::min(left, size_t(std::ceil(1.0 * size[c] / m * spare)));
				copies[c] += (uint32_t) give;
				left -= give;
			}
	}
};

// FIFO of founder slots per class: singly linked through `next` (a slot sits in one queue at a time)
struct SlotQueues {
	std::vector<uint32_t> head, tail;
	void reset(uint32_t classes) { head.assign(classes, NIL); tail.assign(classes, NIL); }
	void push(std::vector<uint32_t> &next, uint32_t c, uint32_t slot)
	{
		next[slot] = NIL;
		if (tail[c] == NIL) head[c] = slot; else next[tail[c]] = slot;
		tail[c] = slot;
	}
	uint32_t pop(std::vector<uint32_t> const &next, uint32_t c)
	{
		uint32_t const slot = head[c];
		head[c] = next[slot];
		if (head[c] == NIL) tail[c] = NIL;
		return slot;
	}
};

struct Edge { uint32_t l, r, rows; };

} // namespace join_detail

// where a joiner's host time goes (fseq_get_join_profile): the class tables (greedy_matcher.cc:31-68), the
// co-occurrence edges (:295-343), the rounds that draw them (:353-439)
struct JoinProfile {
	double ms_classes = 0, ms_edges = 0, ms_draw = 0;
	static double now()
	{
		using namespace std::chrono;
		return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
	}
};

// A, D: segment_count x m (input_permutation / input_divergence at each segment's rb).
// permutations: segment_count x max_segment_size, permutations[s][row] = input row whose substring
// [lb_s, rb_s) is placed in founder `row`.
inline void greedy_match(
	uint32_t const seq_count, uint32_t const max_segment_size, std::vector<JoinSegment> const &segs,
	uint32_t const *A, uint32_t const *D, uint32_t *permutations, JoinProfile *prof = nullptr)
{
	using namespace join_detail;
	uint32_t const m = seq_count, X = max_segment_size;
	if (segs.empty()) return;
	std::fill(permutations, permutations + segs.size() * (size_t) X, 0u);
	double t_mark = prof ? JoinProfile::now() : 0;
	auto lap = [&](double JoinProfile::*field) { if (prof) { double const t = JoinProfile::now(); prof->*field += t - t_mark; t_mark = t; } };

	ClassTable L, R;
	SlotQueues lq, rq;
	std::vector<uint32_t> next(X, NIL);
	std::vector<uint64_t> keys(m);
	std::vector<Edge> edges, ordered;
	std::vector<uint32_t> bucket;

	// first segment: classes in pBWT order take consecutive slots
	L.build(m, 0, A, D, X);
	lq.reset(L.count);
	{
		uint32_t slot = 0;
		for (uint32_t c = 0; c < L.count; ++c)
			for (uint32_t k = 0; k < L.copies[c]; ++k, ++slot)
			{
				permutations[slot] = L.rep[c];
				lq.push(next, c, slot);
			}
	}
	uint64_t seg_start = segs[0].rb;

	for (size_t s = 1; s < segs.size(); ++s)
	{
		uint32_t const *a = A + s * (size_t) m, *d = D + s * (size_t) m;
		uint32_t *perm = permutations + s * (size_t) X;
		lap(&JoinProfile::ms_draw);
		R.build(m, seg_start, a, d, X);
		rq.reset(R.count);
		lap(&JoinProfile::ms_classes);

		// co-occurrence edges: how many rows go from lhs class l to rhs class r
		for (uint32_t row = 0; row < m; ++row) keys[row] = ((uint64_t) L.of_row[row] << 32) | R.of_row[row];
		std::sort(keys.begin(), keys.end());
		edges.clear();
		for (uint32_t i = 0; i < m;)
		{
			uint32_t j = i + 1;
			while (j < m && keys[j] == keys[i]) ++j;
			edges.push_back(Edge{(uint32_t) (keys[i] >> 32), (uint32_t) keys[i], j - i});
			i = j;
		}
		// ... by descending row count, pairs ascending within a count (the edges are in ascending pair order now)
		bucket.assign(m + 2, 0);
		for (Edge const &e : edges) ++bucket[m - e.rows + 1];
		for (uint32_t v = 0; v <= m; ++v) bucket[v + 1] += bucket[v];
		ordered.resize(edges.size());
		for (Edge const &e : edges) ordered[bucket[m - e.rows]++] = e;
		lap(&JoinProfile::ms_edges);

		std::vector<uint32_t> &lhs_left = L.copies;            // copies of an lhs class not yet continued
		std::vector<uint32_t> rhs_left = R.copies;             // copies of an rhs class not yet placed
		auto connect = [&](uint32_t l, uint32_t r) {
			uint32_t const slot = lq.pop(next, l);
			rq.push(next, r, slot);
			perm[slot] = R.rep[r];
			--lhs_left[l];
			--rhs_left[r];
		};
		// rounds over the surviving edges
		size_t alive = ordered.size();
		bool drew = true;
		while (drew)
		{
			drew = false;
			size_t keep = 0;
			for (size_t i = 0; i < alive; ++i)
			{
				Edge const e = ordered[i];
				if (lhs_left[e.l] && rhs_left[e.r])
				{
					connect(e.l, e.r);
					ordered[keep++] = e;
					drew = true;
				}
			}
			alive = keep;
		}
		// leftovers: ascending classes on both sides
		for (uint32_t l = 0, r = 0;;)
		{
			while (l < L.count && 0 == lhs_left[l]) ++l;
			while (r < R.count && 0 == rhs_left[r]) ++r;
			if (l == L.count || r == R.count) break;
			connect(l, r);
		}

		std::swap(L, R);
		std::swap(lq, rq);
		seg_start = segs[s].rb;
	}
	lap(&JoinProfile::ms_draw);
}

// The same joiner on class tables and edge lists that were built elsewhere (on the device: fseq_joinprep.hpp).
// count[s] classes of segment s with rep / size at [s * X + c]; the edges of the pair (s - 1, s) at
// edges[offset[s - 1] .. + nedges[s - 1]) as {l << 16 | r, rows} in ascending (l, r) order.
inline void greedy_match_prepared(
	uint32_t const seq_count, uint32_t const max_segment_size, size_t const S,
	uint32_t const *count, uint32_t const *rep, uint32_t const *size, uint32_t const *edge_words /* 2 per edge */,
	uint32_t const *offset, uint32_t const *nedges, uint32_t *permutations, JoinProfile *prof = nullptr)
{
	using namespace join_detail;
	uint32_t const m = seq_count, X = max_segment_size;
	if (0 == S) return;
	std::fill(permutations, permutations + S * (size_t) X, 0u);
	double t_mark = prof ? JoinProfile::now() : 0;
	auto lap = [&](double JoinProfile::*field) { if (prof) { double const t = JoinProfile::now(); prof->*field += t - t_mark; t_mark = t; } };
	// copies of every class (ClassTable::build's second half): every class one slot, the spare ones in rounds over the
	// classes in descending size
	std::vector<uint32_t> start, order;
	auto copies_of = [&](size_t s, std::vector<uint32_t> &copies) {
		uint32_t const cnt = count[s];
		uint32_t const *sz = size + s * (size_t) X;
		start.assign(m + 2, 0); order.resize(cnt);
		for (uint32_t c = 0; c < cnt; ++c) ++start[m - sz[c] + 1];
		for (uint32_t v = 0; v <= m; ++v) start[v + 1] += start[v];
		for (uint32_t c = 0; c < cnt; ++c) order[start[m - sz[c]]++] = c;
		copies.assign(cnt, 1);
		size_t const spare = X - cnt;
		size_t left = spare;
		while (left)
			for (uint32_t k = 0; k < cnt && left; ++k)
			{
				uint32_t const c = order[k];
				size_t const give = std::min(left, size_t(std::ceil(1.0 * sz[c] / m * spare)));
				copies[c] += (uint32_t) give;
				left -= give;
			}
	};
	SlotQueues lq, rq;
	std::vector<uint32_t> next(X, NIL), lhs_left, rhs_copies, rhs_left, bucket;
	std::vector<Edge> ordered;
	copies_of(0, lhs_left);
	lq.reset(count[0]);
	{
		uint32_t slot = 0;
		for (uint32_t c = 0; c < count[0]; ++c)
			for (uint32_t k = 0; k < lhs_left[c]; ++k, ++slot)
			{
				permutations[slot] = rep[c];
				lq.push(next, c, slot);
			}
	}
	lap(&JoinProfile::ms_classes);
	for (size_t s = 1; s < S; ++s)
	{
		uint32_t *perm = permutations + s * (size_t) X;
		uint32_t const LC = count[s - 1], RC = count[s];
		uint32_t const *rrep = rep + s * (size_t) X;
		copies_of(s, rhs_copies);
		rq.reset(RC);
		lap(&JoinProfile::ms_classes);
		// edges by descending row count, pairs ascending within a count (they arrive in ascending pair order)
		uint32_t const ne = nedges[s - 1];
		uint32_t const *ew = edge_words + 2 * (size_t) offset[s - 1];
		bucket.assign(m + 2, 0);
		for (uint32_t i = 0; i < ne; ++i) ++bucket[m - ew[2 * i + 1] + 1];
		for (uint32_t v = 0; v <= m; ++v) bucket[v + 1] += bucket[v];
		ordered.resize(ne);
		for (uint32_t i = 0; i < ne; ++i) ordered[bucket[m - ew[2 * i + 1]]++] = Edge{ew[2 * i] >> 16, ew[2 * i] & 0xFFFFu, ew[2 * i + 1]};
		lap(&JoinProfile::ms_edges);
		rhs_left = rhs_copies;
		auto connect = [&](uint32_t l, uint32_t r) {
			uint32_t const slot = lq.pop(next, l);
			rq.push(next, r, slot);
			perm[slot] = rrep[r];
			--lhs_left[l];
			--rhs_left[r];
		};
		size_t alive = ordered.size();
		bool drew = true;
		while (drew)
		{
			drew = false;
			size_t keep = 0;
			for (size_t i = 0; i < alive; ++i)
			{
				Edge const e = ordered[i];
				if (lhs_left[e.l] && rhs_left[e.r])
				{
					connect(e.l, e.r);
					ordered[keep++] = e;
					drew = true;
				}
			}
			alive = keep;
		}
		for (uint32_t l = 0, r = 0;;)
		{
			while (l < LC && 0 == lhs_left[l]) ++l;
			while (r < RC && 0 == rhs_left[r]) ++r;
			if (l == LC || r == RC) break;
			connect(l, r);
		}
		lhs_left.swap(rhs_copies);                             // the rhs classes' copies are the next pair's lhs copies
		std::swap(lq, rq);
		lap(&JoinProfile::ms_draw);
	}
}

// ------------------------------------------------------------------------------------------------
// Non-greedy joiners (row N3).  Parity with the reference is unpinned twice over here: Lemon 1.3.1
// (MaxWeightedPerfectMatching on a FullBpGraph, merge_segments_task.cc:103-131) is not in the tree, so
// the matching below is our own Kuhn-Munkres -- the total weight of every matching is optimal, which
// of several optimal matchings Lemon would return is not reproduced; and the reference orders equal
// keys with std::sort / std::shuffle, whose results depend on the standard library it was built with.
// ------------------------------------------------------------------------------------------------

// substring_copy_number (substring_copy_number.hh:15-31)
struct SubstringCopyNumber {
	uint32_t substring_idx, copy_number, string_idx;
};

// pbwt_sample::unique_substring_count_idxs_lhs(lb, out) (libbio): one entry per
// run of rows that agree on [lb, rb), in pBWT order: (first row of the run, run length).
inline std::vector<SubstringCopyNumber> unique_substring_runs(uint32_t const m, uint64_t const lb, uint32_t const *a, uint32_t const *d)
{
	std::vector<SubstringCopyNumber> out;
	for (uint32_t i = 0; i < m; ++i)
	{
		if (0 == i || d[i] > lb) out.push_back(SubstringCopyNumber{a[i], 0u, 0u});
		++out.back().copy_number;
	}
	return out;
}

// join_context::join_segments_and_output, the copy-number preparation for the non-greedy joiners
// (join_context.cc:63-126).  Returns the runs with copy_number turned into a cumulative sum.
inline std::vector<SubstringCopyNumber> prepare_copy_numbers(
	uint32_t const m, uint32_t const max_segment_size, uint64_t const lb, uint32_t const *a, uint32_t const *d, bool const bipartite)
{
	std::vector<SubstringCopyNumber> cn = unique_substring_runs(m, lb, a, d);
	size_t const substring_count = cn.size();
	for (uint32_t i = 0; i < cn.size(); ++i) cn[i].string_idx = i;                 // :82-86
	if (!bipartite)
	{
		std::sort(cn.begin(), cn.end(), [](SubstringCopyNumber const &x, SubstringCopyNumber const &y) { return x.copy_number < y.copy_number; });   // :91
		size_t const empty_slots = max_segment_size - substring_count;             // :94
		size_t remaining = empty_slots;
		for (auto it = cn.rbegin(); it != cn.rend(); ++it)                          // :96-101
		{
			size_t const addition = std::min(remaining, size_t(std::ceil(1.0 * it->copy_number / substring_count * empty_slots)));
			it->copy_number = (uint32_t) (1 + addition);
			remaining -= addition;
		}
		while (remaining)                                                          // :104-114
			for (auto it = cn.rbegin(); it != cn.rend() && remaining; ++it) { ++it->copy_number; --remaining; }
	}
	uint32_t acc = 0;                                                              // make_cumulative_sum, :124
	for (auto &x : cn) { acc += x.copy_number; x.copy_number = acc; }
	return cn;
}

// join_context::join_random_order_and_output (join_context.cc:259-289): slots filled class by class, then
// one std::shuffle per segment from a single std::mt19937(seed).
inline void random_join(
	uint32_t const seq_count, uint32_t const max_segment_size, std::vector<JoinSegment> const &segs,
	uint32_t const *A, uint32_t const *D, uint32_t const seed, uint32_t *permutations,
	std::vector<std::vector<SubstringCopyNumber>> *copy_numbers_out = nullptr)
{
	size_t const X = max_segment_size;
	std::mt19937 urbg(seed);
	if (copy_numbers_out) copy_numbers_out->clear();
	for (size_t s = 0; s < segs.size(); ++s)
	{
		auto cn = prepare_copy_numbers(seq_count, max_segment_size, segs[s].lb, A + s * (size_t) seq_count, D + s * (size_t) seq_count, false);
		uint32_t *perm = permutations + s * X;
		size_t pos = 0;
		for (auto const &x : cn)
			for (; pos < x.copy_number; ++pos) perm[pos] = x.substring_idx;
		std::shuffle(perm, perm + X, urbg);
		if (copy_numbers_out) copy_numbers_out->push_back(std::move(cn));
	}
}

// segment_text (segment_text.hh:22-37)
struct SegmentText {
	std::vector<uint32_t> sequence_indices;
	size_t copied_from = SIZE_MAX;
	bool is_copied() const { return SIZE_MAX != copied_from; }
	size_t row_number(size_t const row) const { return is_copied() ? copied_from : row; }
};

// create_segment_texts_task::execute (create_segment_texts_task.cc:15-81)
inline std::vector<SegmentText> create_segment_texts(
	uint32_t const seq_count, uint32_t const max_segment_size, uint32_t const *a, std::vector<SubstringCopyNumber> const &cumulative)
{
	std::vector<SegmentText> texts(max_segment_size);
	size_t seg_idx = 0, string_idx = 0;
	for (auto const &cn : cumulative)                                              // :24-34
	{
		SegmentText seg;
		seg.sequence_indices.assign(a + string_idx, a + cn.copy_number);
		string_idx = cn.copy_number;
		std::sort(seg.sequence_indices.begin(), seg.sequence_indices.end());
		texts[seg_idx++] = std::move(seg);
	}
	if (seg_idx < max_segment_size)                                                // :37-80
	{
		size_t remaining = max_segment_size - seg_idx;
		// (the reference's std::sort leaves the order of equal sizes to the standard library it was built with; here they keep
		// the classes' pBWT order, which is also what the device form computes, fseq_joinbip.hpp)
		std::stable_sort(texts.begin(), texts.begin() + seg_idx, [](SegmentText const &x, SegmentText const &y) {
			return x.sequence_indices.size() > y.sequence_indices.size();
		});
		size_t const limit = seg_idx;
		size_t it = seg_idx;
		for (size_t i = 0; i < limit; ++i)                                         // :50-63
		{
			size_t const copy_number = std::min(remaining, size_t(std::ceil(1.0 * texts[i].sequence_indices.size() / seq_count * remaining)));
			for (size_t k = 0; k < copy_number; ++k) { texts[it] = SegmentText(); texts[it].copied_from = i; ++it; }
			remaining -= copy_number;
			if (0 == remaining) break;
		}
		while (remaining)                                                          // :66-78
			for (size_t i = 0; i < limit && remaining; ++i) { texts[it] = SegmentText(); texts[it].copied_from = i; ++it; --remaining; }
	}
	return texts;
}

// Maximum-weight perfect matching on the complete bipartite graph with weights w[l * X + r] >= 0
// (Kuhn-Munkres with potentials, O(X^3)); matching[l] = r.  Returns the total weight.
inline int64_t max_weight_perfect_matching(size_t const X, std::vector<int32_t> const &w, std::vector<uint32_t> &matching)
{
	// minimise cost = -weight; rows / columns are 1-based inside
	int64_t const INF = std::numeric_limits<int64_t>::max() / 4;
	std::vector<int64_t> u(X + 1, 0), v(X + 1, 0), minv(X + 1);
	std::vector<size_t> p(X + 1, 0), way(X + 1, 0);
	std::vector<char> used(X + 1);
	for (size_t i = 1; i <= X; ++i)
	{
		p[0] = i;
		size_t j0 = 0;
		std::fill(minv.begin(), minv.end(), INF);
		std::fill(used.begin(), used.end(), 0);
		do
		{
			used[j0] = 1;
			size_t const i0 = p[j0];
			size_t j1 = 0;
			int64_t delta = INF;
			for (size_t j = 1; j <= X; ++j)
			{
				if (used[j]) continue;
				int64_t const cur = -(int64_t) w[(i0 - 1) * X + (j - 1)] - u[i0] - v[j];
				if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
				if (minv[j] < delta) { delta = minv[j]; j1 = j; }
			}
			for (size_t j = 0; j <= X; ++j)
			{
				if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
				else minv[j] -= delta;
			}
			j0 = j1;
		} while (p[j0] != 0);
		do
		{
			size_t const j1 = way[j0];
			p[j0] = p[j1];
			j0 = j1;
		} while (j0);
	}
	matching.assign(X, 0);
	int64_t total = 0;
	for (size_t j = 1; j <= X; ++j)
	{
		matching[p[j] - 1] = (uint32_t) (j - 1);
		total += w[(p[j] - 1) * X + (j - 1)];
	}
	return total;
}

// merge_segments_task::execute with bipartite_set_scoring::INTERSECTION (main.cc:126;
// merge_segments_task.cc:133-195): weights |rows_l n rows_r| between the real texts, copied to the
// placeholder rows / columns from the text they copy.
inline std::vector<int32_t> intersection_weights(uint32_t const seq_count, std::vector<SegmentText> const &lhs, std::vector<SegmentText> const &rhs)
{
	size_t const X = lhs.size();
	std::vector<int32_t> w(X * X, 0);
	std::vector<uint32_t> cl(seq_count, 0), cr(seq_count, 0);
	for (size_t i = 0; i < X; ++i)
	{
		for (uint32_t const r : lhs[i].sequence_indices) cl[r] = (uint32_t) i;
		for (uint32_t const r : rhs[i].sequence_indices) cr[r] = (uint32_t) i;
	}
	for (uint32_t r = 0; r < seq_count; ++r) ++w[(size_t) cl[r] * X + cr[r]];       // every row is in exactly one text per side
	for (size_t i = 0; i < X; ++i)
		for (size_t j = 0; j < X; ++j)
			if (lhs[i].is_copied() || rhs[j].is_copied())
				w[i * X + j] = w[lhs[i].row_number(i) * X + rhs[j].row_number(j)];
	return w;
}

// bipartite_matcher::match ... create_permutations_and_notify (bipartite_matcher.cc:17-151).
// texts_out (optional): the segment texts per segment (the bipartite segments file prints them);
// weights_out (optional): total weight of the matching between segments s and s + 1.
inline void bipartite_match(
	uint32_t const seq_count, uint32_t const max_segment_size, std::vector<JoinSegment> const &segs,
	uint32_t const *A, uint32_t const *D, uint32_t *permutations,
	std::vector<std::vector<SegmentText>> *texts_out = nullptr, std::vector<int64_t> *weights_out = nullptr)
{
	size_t const X = max_segment_size, S = segs.size();
	if (0 == S) return;
	// The segment texts of every segment and the matching of every adjacent pair are independent of each other (the
	// reference runs them as concurrent tasks too: create_segment_texts_task / merge_segments_task on a dispatch queue);
	// only the chaining of the permutations is serial.  On BASELINE C3 the 6,150 Kuhn-Munkres solves were 1.27 s of
	// one host thread against 10 ms of segmentation.
	std::vector<std::vector<SegmentText>> texts(S);
	std::vector<std::vector<uint32_t>> matchings(S);
	std::vector<int64_t> totals(S, 0);
	size_t const workers = std::max<size_t>(1, std::min<size_t>({(size_t) std::thread::hardware_concurrency(), (size_t) 32, S}));
	auto parallel_for = [&](size_t lo, size_t hi, auto &&body) {
		std::atomic<size_t> next{lo};
		auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < hi;) body(i); };
		std::vector<std::thread> ths;
		for (size_t t = 1; t < workers; ++t) ths.emplace_back(work);
		work();
		for (auto &th : ths) th.join();
	};
	parallel_for(0, S, [&](size_t s) {
		uint32_t const *a = A + s * (size_t) seq_count, *d = D + s * (size_t) seq_count;
		texts[s] = create_segment_texts(seq_count, max_segment_size, a, prepare_copy_numbers(seq_count, max_segment_size, segs[s].lb, a, d, true));
	});
	parallel_for(1, S, [&](size_t s) {
		totals[s] = max_weight_perfect_matching(X, intersection_weights(seq_count, texts[s - 1], texts[s]), matchings[s]);
	});
	auto representative = [](std::vector<SegmentText> const &tx, size_t const i) {
		return tx[tx[i].row_number(i)].sequence_indices.front();                   // first_sequence_index of the non-copied text
	};
	// create_initial_permutation (:129-151)
	std::vector<uint32_t> order(X);
	std::iota(order.begin(), order.end(), 0u);
	for (size_t i = 0; i < X; ++i) permutations[i] = representative(texts[0], i);
	if (weights_out) weights_out->clear();
	for (size_t s = 1; s < S; ++s)                                                 // :95-118
	{
		std::vector<uint32_t> const &matching = matchings[s];
		if (weights_out) weights_out->push_back(totals[s]);
		uint32_t *perm = permutations + s * X;
		for (size_t i = 0; i < X; ++i)
		{
			uint32_t const matched = matching[order[i]];
			perm[i] = representative(texts[s], matched);
			order[i] = matched;
		}
	}
	if (texts_out) *texts_out = std::move(texts);
}

} // namespace fseq
