// fseq_joinbip.hpp -- [r5] the bipartite-matching joiner on the device (row N3).
//
// replaces: bipartite_matcher::match (bipartite_matcher.cc:17-151) with create_segment_texts_task::execute
// (create_segment_texts_task.cc:15-81) and merge_segments_task::execute (merge_segments_task.cc:62-67,103-195), as
// fseq_join.hpp restates them on the host -- the same arithmetic in the same order, so the permutations are the host
// joiner's, entry for entry (tests/test_join.py compares them).
//
// End to end on BASELINE C3 -- the configuration that names this joiner -- the host form was 30 ms of boundary states over
// PCIe (123 MB) and ~100-170 ms of 6,150 Kuhn-Munkres solves on the host's threads beside a 5 ms segmentation.  The boundary
// states are in HBM, an assignment problem of max_segment_size <= 181 texts fits a workgroup's LDS, and the 6,150 of them are
// independent (the reference runs them as concurrent tasks, merge_segments_task): so
//   k_join_classes  (fseq_joinprep.hpp) classes of every row, their sizes and smallest rows, per merged segment
//   k_bip_texts     one wave per segment: the segment's texts -- classes ordered by size where there are fewer than
//                   max_segment_size of them, the placeholder texts and what they copy (create_segment_texts_task.cc:37-80)
//   k_bip_match     one wave per adjacent pair: the intersection weights of the real texts in an LDS matrix
//                   (merge_segments_task.cc:133-195), then Kuhn-Munkres with potentials over the wave's lanes (a lane owns
//                   three columns; the row loop and the augmenting path are the serial part, as on the host)
//   k_bip_chain     one workgroup: the permutations chained through the matchings (bipartite_matcher.cc:95-151), the
//                   matchings and representatives staged through LDS in tiles of segments
// and only the permutations (S x X words) go to the host.  Parity with the reference stays unpinned as for the host form:
// Lemon's choice among optimal matchings and std::sort's order of equal sizes are not reproduced (fseq_join.hpp).
#pragma once

#include "fseq_joinprep.hpp"

namespace fseq {

constexpr uint32_t JB_STRIPS = 3;                        // columns per lane in k_bip_match: 3 x 64 >= JP_MAX_CLASSES
constexpr uint32_t JB_CHAIN_T = 192;                     // threads of k_bip_chain (one per slot)
constexpr uint32_t JB_CHAIN_TILE = 32;                   // segments per LDS tile there
static_assert(JB_STRIPS * 64u >= JP_MAX_CLASSES && JB_CHAIN_T >= JP_MAX_CLASSES, "a lane / thread per text");

// smallest row of every class (the texts' first_sequence_index): k_join_classes' rep is the class's first row in pBWT order
static __global__ __launch_bounds__(JP_T) void k_bip_minrow(
	uint16_t const *__restrict__ of_row, uint32_t m, uint32_t X, uint32_t *__restrict__ minrow)
{
	__shared__ uint32_t mn[JP_MAX_CLASSES + 1];
	uint32_t const s = blockIdx.x, tid = threadIdx.x;
	for (uint32_t c = tid; c <= JP_MAX_CLASSES; c += JP_T) mn[c] = 0xFFFFFFFFu;
	__syncthreads();
	uint16_t const *of = of_row + (size_t) s * m;
	for (uint32_t row = tid; row < m; row += JP_T) atomicMin(&mn[min((uint32_t) of[row], JP_MAX_CLASSES)], row);
	__syncthreads();
	for (uint32_t c = tid; c < X; c += JP_T) minrow[(size_t) s * X + c] = c <= JP_MAX_CLASSES ? mn[c] : 0xFFFFFFFFu;
}

// tpos[s][class] = text of the class; src[s][text] = the real text a slot shows (itself, or what a placeholder copies);
// reprow[s][text] = the first row of that real text
static __global__ __launch_bounds__(64) void k_bip_texts(
	uint32_t const *__restrict__ count, uint32_t const *__restrict__ size, uint32_t const *__restrict__ minrow, uint32_t m, uint32_t X,
	uint16_t *__restrict__ tpos, uint16_t *__restrict__ src, uint32_t *__restrict__ reprow)
{
	__shared__ uint32_t t_size[JP_MAX_CLASSES], t_min[JP_MAX_CLASSES];
	__shared__ uint16_t t_src[JP_MAX_CLASSES];
	uint32_t const s = blockIdx.x, lane = threadIdx.x;
	uint32_t const nc = min(min(count[s], X), JP_MAX_CLASSES);
	uint32_t const *sz = size + (size_t) s * X;
	// fewer classes than slots: the texts in descending order of their sizes (create_segment_texts_task.cc:43-45; equal
	// sizes keep the classes' order -- fseq_join.hpp sorts stably), else the classes' own order
	bool const sorted = nc < X;
	for (uint32_t c = lane; c < nc; c += 64u)
	{
		uint32_t const mine = sz[c];
		uint32_t pos = c;
		if (sorted)
		{
			pos = 0;
			for (uint32_t c2 = 0; c2 < nc; ++c2)
			{
				uint32_t const other = sz[c2];
				pos += (other > mine || (other == mine && c2 < c)) ? 1u : 0u;
			}
		}
		tpos[(size_t) s * X + c] = (uint16_t) pos;
		t_size[pos] = mine;
		t_min[pos] = minrow[(size_t) s * X + c];
		t_src[pos] = (uint16_t) pos;
	}
	__syncthreads();
	if (lane == 0 && sorted && nc)
	{
		// the placeholders: copies in proportion to the texts' sizes, then one each in turn (:50-78)
		uint32_t remaining = X - nc, it = nc;
		for (uint32_t i = 0; i < nc && remaining; ++i)
		{
			double const share = ceil(1.0 * (double) t_size[i] / (double) m * (double) remaining);
			uint32_t const cn = share < (double) remaining ? (uint32_t) share : remaining;
			for (uint32_t k = 0; k < cn; ++k) t_src[it++] = (uint16_t) i;
			remaining -= cn;
		}
		while (remaining)
			for (uint32_t i = 0; i < nc && remaining; ++i) { t_src[it++] = (uint16_t) i; --remaining; }
	}
	__syncthreads();
	for (uint32_t i = lane; i < X; i += 64u)
	{
		uint32_t const r = nc ? t_src[i] : 0u;
		src[(size_t) s * X + i] = (uint16_t) r;
		reprow[(size_t) s * X + i] = nc ? t_min[r] : 0xFFFFFFFFu;
	}
}

__host__ __device__ inline size_t bip_match_lds_bytes(uint32_t X) { return ((size_t) X * X + 3u * ((size_t) X + 1u) + X) * 4u; }

// pair p = (segment p, segment p + 1): matching[p][l] = r of a maximum-weight perfect matching on the X x X intersection
// weights; weight[p] its total.  fseq_join.hpp's max_weight_perfect_matching, the columns j = 1 .. X over the lanes.
static __global__ __launch_bounds__(64) void k_bip_match(
	uint16_t const *__restrict__ of_row, uint32_t const *__restrict__ count, uint16_t const *__restrict__ tpos, uint16_t const *__restrict__ src,
	uint32_t m, uint32_t X, uint16_t *__restrict__ matching, long long *__restrict__ weight)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t jb_lds[];
	uint32_t *const wreal = jb_lds;                              // [nl][nr]: rows in real text l of the left and r of the right segment
	int32_t *const u_l = reinterpret_cast<int32_t *>(wreal + (size_t) X * X);       // [X + 1]
	uint32_t *const p_l = reinterpret_cast<uint32_t *>(u_l + X + 1u);                // [X + 1]: row matched to column j (0: none)
	uint32_t *const way_l = p_l + X + 1u;                        // [X + 1]
	uint32_t *const srcl = way_l + X + 1u;                       // [X]
	uint32_t const p = blockIdx.x, lane = threadIdx.x;
	uint32_t const nl = min(min(count[p], X), JP_MAX_CLASSES), nr = min(min(count[p + 1], X), JP_MAX_CLASSES);
	uint16_t const *ofl = of_row + (size_t) p * m, *ofr = of_row + (size_t) (p + 1) * m;
	uint16_t const *tl = tpos + (size_t) p * X, *tr = tpos + (size_t) (p + 1) * X;
	for (uint32_t i = lane; i < nl * nr; i += 64u) wreal[i] = 0;
	for (uint32_t i = lane; i <= X; i += 64u) { u_l[i] = 0; p_l[i] = 0; way_l[i] = 0; }
	for (uint32_t i = lane; i < X; i += 64u) srcl[i] = src[(size_t) p * X + i];
	__syncthreads();
	for (uint32_t row = lane; row < m; row += 64u)
	{
		uint32_t const cl = ofl[row], cr = ofr[row];
		if (cl < nl && cr < nr) atomicAdd(&wreal[(uint32_t) tl[cl] * nr + tr[cr]], 1u);
	}
	__syncthreads();
	constexpr int32_t INF = 0x3FFFFFFF;
	uint32_t sr[JB_STRIPS];                                      // the real text of the right segment behind column j
	int32_t v[JB_STRIPS], minv[JB_STRIPS];
#pragma unroll
	for (uint32_t k = 0; k < JB_STRIPS; ++k)
	{
		uint32_t const j = lane + 64u * k + 1u;
		sr[k] = j <= X ? src[(size_t) (p + 1) * X + j - 1u] : 0u;
		v[k] = 0;
	}
	for (uint32_t i = 1; i <= X; ++i)
	{
		if (lane == 0) p_l[0] = i;
		__syncthreads();
		uint32_t j0 = 0, used = 0;
#pragma unroll
		for (uint32_t k = 0; k < JB_STRIPS; ++k) minv[k] = INF;
		do
		{
			if (j0 && lane == ((j0 - 1u) & 63u)) used |= 1u << ((j0 - 1u) >> 6);
			uint32_t const i0 = p_l[j0];
			int32_t const ui0 = u_l[i0];
			uint32_t const rowbase = srcl[i0 - 1u] * nr;
			int32_t best = INF;
			uint32_t bestj = 0x7FFFFFFFu;
#pragma unroll
			for (uint32_t k = 0; k < JB_STRIPS; ++k)
			{
				uint32_t const j = lane + 64u * k + 1u;
				if (j <= X && !((used >> k) & 1u))
				{
					int32_t const cur = -(int32_t) wreal[rowbase + sr[k]] - ui0 - v[k];
					if (cur < minv[k]) { minv[k] = cur; way_l[j] = j0; }
					if (minv[k] < best) { best = minv[k]; bestj = j; }       // (the lowest column among equal ones: ascending j)
				}
			}
#pragma unroll
			for (int off = 32; off >= 1; off >>= 1)
			{
				int32_t const ob = __shfl_xor(best, off, WAVE);
				uint32_t const oj = (uint32_t) __shfl_xor((int) bestj, off, WAVE);
				if (ob < best || (ob == best && oj < bestj)) { best = ob; bestj = oj; }
			}
			int32_t const delta = best;
			__syncthreads();                                     // (u_l[i0] has been read by everybody)
#pragma unroll
			for (uint32_t k = 0; k < JB_STRIPS; ++k)
			{
				uint32_t const j = lane + 64u * k + 1u;
				if (j <= X)
				{
					if ((used >> k) & 1u) { u_l[p_l[j]] += delta; v[k] -= delta; }
					else minv[k] -= delta;
				}
			}
			if (lane == 0) u_l[p_l[0]] += delta;
			__syncthreads();
			j0 = bestj;
		} while (p_l[j0] != 0u);
		if (lane == 0)
		{
			do
			{
				uint32_t const j1 = way_l[j0];
				p_l[j0] = p_l[j1];
				j0 = j1;
			} while (j0);
		}
		__syncthreads();
	}
	long long total = 0;
#pragma unroll
	for (uint32_t k = 0; k < JB_STRIPS; ++k)
	{
		uint32_t const j = lane + 64u * k + 1u;
		if (j <= X)
		{
			uint32_t const l = p_l[j] - 1u;
			matching[(size_t) p * X + l] = (uint16_t) (j - 1u);
			total += wreal[srcl[l] * nr + sr[k]];
		}
	}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, WAVE);
	if (lane == 0 && weight) weight[p] = total;
}

__host__ __device__ inline size_t bip_chain_lds_bytes(uint32_t X) { return (size_t) JB_CHAIN_TILE * X * 6u + 16u; }

// create_initial_permutation and the chaining (bipartite_matcher.cc:95-151): slot i shows, in segment s, the text its text of
// segment s - 1 is matched to; perm[s][i] = that text's first row
static __global__ __launch_bounds__(JB_CHAIN_T) void k_bip_chain(
	uint16_t const *__restrict__ matching, uint32_t const *__restrict__ reprow, uint32_t S, uint32_t X, uint32_t *__restrict__ perm)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t jc_lds[];
	uint32_t *const rt = jc_lds;                                  // [tile][X] first rows
	uint16_t *const mt = reinterpret_cast<uint16_t *>(rt + (size_t) JB_CHAIN_TILE * X);      // [tile][X] matchings
	uint32_t const i = threadIdx.x;
	uint32_t order = i;
	if (i < X) perm[i] = reprow[i];
	for (uint32_t s0 = 1; s0 < S; s0 += JB_CHAIN_TILE)
	{
		uint32_t const g = min(JB_CHAIN_TILE, S - s0);
		__syncthreads();
		for (uint32_t e = i; e < g * X; e += JB_CHAIN_T)
		{
			mt[e] = matching[(size_t) (s0 - 1u) * X + e];
			rt[e] = reprow[(size_t) s0 * X + e];
		}
		__syncthreads();
		if (i < X)
			for (uint32_t t = 0; t < g; ++t)
			{
				uint32_t const matched = mt[t * X + order];
				perm[(size_t) (s0 + t) * X + i] = rt[t * X + matched];
				order = matched;
			}
	}
}

} // namespace fseq
