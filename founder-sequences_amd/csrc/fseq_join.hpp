// fseq_join.hpp -- host-side segment joining: the greedy matcher and the founders
// writer.  Stays on the host, as in the reference (north_star: "segment joining ... stays on host
// until it shows in the profile").  Restated from founder-sequences/greedy_matcher.cc:31-465 and
// founder-sequences/join_context.cc:191-229,333-356; citations relative to /root/reference.
//
// Assumption carried from SURVEY.md Appendix B A8 (libbio is un-vendored): lb::radix_sort<true>::sort
// orders by descending key and is stable; lb::radix_sort<>::sort orders by ascending key.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <list>
#include <map>
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

} // namespace fseq
