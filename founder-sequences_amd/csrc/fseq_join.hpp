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
#include <list>
#include <map>
#include <numeric>
#include <random>
#include <utility>
#include <vector>
#include <algorithm>

namespace fseq {

struct JoinSegment {
	uint64_t lb, rb;
};

namespace join_detail {

typedef uint32_t seq_index;
typedef std::vector<seq_index> seq_index_vector;
typedef std::vector<std::pair<seq_index, seq_index>> seq_occurrence_vector;
typedef std::vector<std::list<seq_index>> index_list_vector;

// greedy_matcher.cc:31-68
inline void update_string_mappings(
	size_t const seq_count, uint64_t const seg_start_pos, uint32_t const *permutation, uint32_t const *divergence,
	seq_index &distinct_substrings, seq_index_vector &seq_index_mapping, seq_index_vector &seq_inverse_mapping,
	seq_index_vector &run_lengths)
{
	size_t current_run_length = 0;
	distinct_substrings = 0;
	for (size_t i = 0; i < seq_count; ++i)
	{
		auto const string_idx = permutation[i];
		if (seg_start_pos < divergence[i])                           // :51
		{
			run_lengths[distinct_substrings] = (seq_index) current_run_length;   // run_lengths[0] stays 0
			seq_index_mapping[distinct_substrings++] = string_idx;
			current_run_length = 0;
		}
		seq_inverse_mapping[string_idx] = distinct_substrings - 1;  // :60
		++current_run_length;
	}
	run_lengths[distinct_substrings] = (seq_index) current_run_length;          // :67
}

// greedy_matcher.cc:71-78 followed by lb::radix_sort<true>::sort(..., pair.second) (:259,:290; A8)
inline void sorted_occurrences(seq_index const distinct_substrings, seq_index_vector const &run_lengths, seq_occurrence_vector &occ)
{
	occ.clear();
	for (size_t i = 0; i < distinct_substrings; ++i)
		occ.emplace_back((seq_index) i, run_lengths[1 + i]);
	std::stable_sort(occ.begin(), occ.end(), [](auto const &x, auto const &y) { return x.second > y.second; });
}

// greedy_matcher.cc:81-105
inline void update_copies(seq_occurrence_vector const &occ, size_t const max_segment_size, size_t const seq_count, seq_index_vector &cn)
{
	std::fill(cn.begin(), cn.end(), 1);
	size_t const to_fill = max_segment_size - occ.size();
	size_t rem_size = to_fill;
	while (true)
	{
		for (auto const &pr : occ)
		{
			size_t const copy_count = std::min(rem_size, size_t(std::ceil(1.0 * pr.second / seq_count * to_fill)));   // :97
			rem_size -= copy_count;
			cn[pr.first] += (seq_index) copy_count;
			if (0 == rem_size)
				return;
		}
	}
}

// greedy_matcher.cc:163-198
inline void draw_edge(seq_index const lhs_idx, seq_index const rhs_idx, seq_index_vector const &rhs_seq_mapping,
                      index_list_vector &lhs_slots, index_list_vector &rhs_slots, uint32_t *permutation)
{
	auto &slot_list = lhs_slots[lhs_idx];
	auto const slot = slot_list.front();
	slot_list.pop_front();
	rhs_slots[rhs_idx].push_back(slot);
	permutation[slot] = rhs_seq_mapping[rhs_idx];
}

} // namespace join_detail

// greedy_matcher::match, greedy_matcher.cc:204-465.
// A, D: segment_count x m (input_permutation / input_divergence at each segment's rb).
// permutations: segment_count x max_segment_size, permutations[s][row] = input row whose substring
// [lb_s, rb_s) is placed in founder `row`.
inline void greedy_match(
	uint32_t const seq_count, uint32_t const max_segment_size, std::vector<JoinSegment> const &segs,
	uint32_t const *A, uint32_t const *D, uint32_t *permutations)
{
	using namespace join_detail;
	size_t const X = max_segment_size;
	seq_index lhs_distinct = 0, rhs_distinct = 0;
	seq_index_vector lhs_seq_mapping(X, UINT32_MAX), rhs_seq_mapping(X, UINT32_MAX);
	seq_index_vector lhs_inverse(seq_count, UINT32_MAX), rhs_inverse(seq_count, UINT32_MAX);
	seq_index_vector lhs_rl(1 + X, 0), rhs_rl(1 + X, 0);
	seq_index_vector lhs_cn(X, 0), rhs_cn(X, 0), rhs_rc(X, 0);
	seq_occurrence_vector occ;
	std::vector<uint64_t> index_pairs(seq_count, 0);
	uint64_t seg_start_idx = 0;
	typedef std::list<std::pair<seq_index, seq_index>> index_pair_list;
	std::map<size_t, index_pair_list, std::greater<size_t>> index_pairs_by_count;
	index_list_vector lhs_slots(X), rhs_slots(X);

	if (segs.empty()) return;
	std::fill(permutations, permutations + segs.size() * X, 0u);

	// first segment (:241-262)
	{
		update_string_mappings(seq_count, seg_start_idx, A, D, lhs_distinct, lhs_seq_mapping, lhs_inverse, lhs_rl);
		seg_start_idx = segs[0].rb;                                              // :257, sequence_idx()
		sorted_occurrences(lhs_distinct, lhs_rl, occ);
		update_copies(occ, X, seq_count, lhs_cn);
		// update_initial_permutation (:128-160)
		size_t i = 0;
		for (size_t seq_idx = 0; seq_idx < lhs_distinct; ++seq_idx)
		{
			auto const copy_count = lhs_cn[seq_idx];
			lhs_slots[seq_idx].clear();
			for (size_t j = 0; j < copy_count; ++j)
				lhs_slots[seq_idx].emplace_back((seq_index) (j + i));
			std::fill(permutations + i, permutations + i + copy_count, lhs_seq_mapping[seq_idx]);
			i += copy_count;
		}
	}

	for (size_t target = 1; target < segs.size(); ++target)                      // :266
	{
		uint32_t const *permutation_in = A + target * (size_t) seq_count;
		uint32_t const *divergence_in = D + target * (size_t) seq_count;
		uint32_t *permutation = permutations + target * X;
		index_pairs_by_count.clear();

		update_string_mappings(seq_count, seg_start_idx, permutation_in, divergence_in, rhs_distinct, rhs_seq_mapping, rhs_inverse, rhs_rl);
		sorted_occurrences(rhs_distinct, rhs_rl, occ);
		update_copies(occ, X, seq_count, rhs_cn);

		// edges as (lhs class, rhs class) pairs, sorted ascending (:294-311)
		for (size_t i = 0; i < seq_count; ++i)
		{
			auto const seq_idx = permutation_in[i];
			index_pairs[i] = ((uint64_t) lhs_inverse[seq_idx] << 32) | rhs_inverse[seq_idx];
		}
		std::sort(index_pairs.begin(), index_pairs.end());

		// counts -> lists of unique edges in ascending pair order (:316-343)
		{
			uint64_t prev_item = index_pairs.front();
			size_t current_count = 1;
			for (size_t i = 1; i < seq_count; ++i)
			{
				auto const current_item = index_pairs[i];
				if (prev_item == current_item)
					++current_count;
				else
				{
					index_pairs_by_count[current_count].emplace_back((seq_index) (prev_item >> 32), (seq_index) (prev_item & 0xFFFFFFFFu));
					prev_item = current_item;
					current_count = 1;
				}
			}
			index_pairs_by_count[current_count].emplace_back((seq_index) (prev_item >> 32), (seq_index) (prev_item & 0xFFFFFFFFu));
		}

		// draw the edges (:346-408)
		rhs_rc = rhs_cn;
		bool did_draw_edge = true;
		while (did_draw_edge)
		{
			did_draw_edge = false;
			auto ip_it = index_pairs_by_count.begin();
			while (ip_it != index_pairs_by_count.end())
			{
				auto &list = ip_it->second;
				auto it = list.begin();
				while (it != list.end())
				{
					auto const lhs_idx = it->first, rhs_idx = it->second;
					if (lhs_cn[lhs_idx] && rhs_rc[rhs_idx])                         // :380
					{
						did_draw_edge = true;
						--lhs_cn[lhs_idx];
						--rhs_rc[rhs_idx];
						draw_edge(lhs_idx, rhs_idx, rhs_seq_mapping, lhs_slots, rhs_slots, permutation);
						++it;
					}
					else
						it = list.erase(it);
				}
				if (!list.empty())
					++ip_it;
				else
					ip_it = index_pairs_by_count.erase(ip_it);
			}
		}

		// the remaining edges (:412-439)
		{
			size_t i = 0, j = 0;
			while (true)
			{
				bool done = false;
				while (0 == lhs_cn[i])
				{
					++i;
					if (i == lhs_distinct) { done = true; break; }
				}
				if (done) break;
				while (0 == rhs_rc[j])
				{
					++j;
					if (j == rhs_distinct) { done = true; break; }
				}
				if (done) break;
				draw_edge((seq_index) i, (seq_index) j, rhs_seq_mapping, lhs_slots, rhs_slots, permutation);
				--lhs_cn[i];
				--rhs_rc[j];
			}
		}

		using std::swap;                                                         // :453-460
		swap(lhs_distinct, rhs_distinct);
		swap(lhs_seq_mapping, rhs_seq_mapping);
		swap(lhs_inverse, rhs_inverse);
		swap(lhs_rl, rhs_rl);
		swap(lhs_cn, rhs_cn);
		swap(lhs_slots, rhs_slots);
		seg_start_idx = segs[target].rb;
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
		std::sort(texts.begin(), texts.begin() + seg_idx, [](SegmentText const &x, SegmentText const &y) {
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
	std::vector<std::vector<SegmentText>> texts(S);
	for (size_t s = 0; s < S; ++s)
	{
		uint32_t const *a = A + s * (size_t) seq_count, *d = D + s * (size_t) seq_count;
		texts[s] = create_segment_texts(seq_count, max_segment_size, a, prepare_copy_numbers(seq_count, max_segment_size, segs[s].lb, a, d, true));
	}
	auto representative = [](std::vector<SegmentText> const &tx, size_t const i) {
		return tx[tx[i].row_number(i)].sequence_indices.front();                   // first_sequence_index of the non-copied text
	};
	// create_initial_permutation (:129-151)
	std::vector<uint32_t> order(X);
	std::iota(order.begin(), order.end(), 0u);
	for (size_t i = 0; i < X; ++i) permutations[i] = representative(texts[0], i);
	if (weights_out) weights_out->clear();
	std::vector<uint32_t> matching;
	for (size_t s = 1; s < S; ++s)                                                 // :95-118
	{
		int64_t const total = max_weight_perfect_matching(X, intersection_weights(seq_count, texts[s - 1], texts[s]), matching);
		if (weights_out) weights_out->push_back(total);
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
