// fseq_joinprep.hpp -- the data-parallel front of the greedy joiner on the device (row N1).
//
// End to end (profiles/r03_e2e_*_before.json) the join was the wall time of the drop-in: on BASELINE C3 the
// segmentation takes 10.5 ms, then 123 MB of boundary states cross PCIe (26 ms) and the host spends 275 ms building
// the (lhs class, rhs class) co-occurrence edges -- a 64-bit sort of m keys per segment pair -- and 18 ms on the
// class tables; the serial part of greedy_matcher.cc, the edge drawing (:353-439), is 5 ms.  So the two data-parallel
// pieces run where the boundary states already are:
//   k_join_classes  one workgroup per merged segment: a row starts a new class at pBWT position i iff
//                   seg_start < d[i] (update_string_mappings, greedy_matcher.cc:31-68): class of every row, the
//                   representative (first row) and the size of every class
//   k_join_edges    one workgroup per adjacent pair: the co-occurrence counts |lhs class l n rhs class r| in an LDS
//                   matrix (the classes of a segment are at most max_segment_size, a few dozen), its non-zero entries
//                   in ascending (l, r) order = the edge list greedy_matcher.cc:295-343 builds by sorting pair keys
// and only the class tables (S' x X words) and the edges (a few per class) go to the host, which hands out the copies
// and draws the edges as before (fseq_join.hpp).
#pragma once

#include "fseq_core.hpp"

namespace fseq {

constexpr uint32_t JP_T = 256;
constexpr uint32_t JP_MAX_CLASSES = 181;                 // LDS matrix of 181 x 181 counters = 128 KiB

// of_row[s][row] (u16), rep[s][c], size[s][c] (stride X), count[s]
__global__ __launch_bounds__(JP_T) void k_join_classes(
	uint32_t const *__restrict__ snap_a, uint32_t const *__restrict__ snap_d, uint64_t const *__restrict__ seg_rb, uint32_t m, uint32_t X,
	uint16_t *__restrict__ of_row, uint32_t *__restrict__ rep, uint32_t *__restrict__ size, uint32_t *__restrict__ count)
{
	__shared__ uint32_t sz[JP_MAX_CLASSES + 1];
	__shared__ uint32_t scratch[JP_T / WAVE + 1];
	uint32_t const s = blockIdx.x, tid = threadIdx.x;
	uint64_t const seg_start = s ? seg_rb[s - 1] : 0;
	uint32_t const *a = snap_a + (size_t) s * m, *d = snap_d + (size_t) s * m;
	uint16_t *out = of_row + (size_t) s * m;
	for (uint32_t c = tid; c <= JP_MAX_CLASSES; c += JP_T) sz[c] = 0;
	__syncthreads();
	uint32_t running = 0;
	for (uint32_t base = 0; base < m; base += JP_T)
	{
		uint32_t const i = base + tid;
		uint32_t const ai = i < m ? a[i] : 0u;
		uint32_t const flag = (i < m && seg_start < (uint64_t) d[i]) ? 1u : 0u;
		uint32_t total;
		uint32_t const before = block_excl_add<JP_T>(flag, scratch, &total);
		if (i < m)
		{
			uint32_t const cls = min(running + before + flag - 1u, JP_MAX_CLASSES);      // (position 0 always starts a class: d[0] = rb > seg_start)
			out[ai] = (uint16_t) cls;
			atomicAdd(&sz[cls], 1u);
			if (flag && cls < X) rep[(size_t) s * X + cls] = ai;
		}
		running += total;
		__syncthreads();
	}
	for (uint32_t c = tid; c < X; c += JP_T) size[(size_t) s * X + c] = c <= JP_MAX_CLASSES ? sz[c] : 0u;
	if (tid == 0) count[s] = running;
}

// pair p = (segment p, segment p + 1): edges[offset[p] .. offset[p] + nedges[p]) = {l << 16 | r, rows}, ascending (l, r)
__global__ __launch_bounds__(JP_T) void k_join_edges(
	uint16_t const *__restrict__ of_row, uint32_t const *__restrict__ count, uint32_t m, uint32_t X,
	uint2 *__restrict__ edges, uint64_t cap_total, uint32_t *__restrict__ offset, uint32_t *__restrict__ nedges, unsigned long long *__restrict__ cursor)
{
	extern __shared__ __attribute__((aligned(16))) uint32_t cooc[];
	__shared__ uint32_t scratch[JP_T / WAVE + 1];
	__shared__ unsigned long long s_base;
	uint32_t const p = blockIdx.x, tid = threadIdx.x;
	// (the LDS matrix holds X x X counters: a class count above X -- which the host rejects afterwards -- must not index past it)
	uint32_t const LC = min(min(count[p], JP_MAX_CLASSES), X), RC = min(min(count[p + 1], JP_MAX_CLASSES), X);
	uint32_t const cells = LC * RC;
	uint16_t const *L = of_row + (size_t) p * m, *R = of_row + (size_t) (p + 1) * m;
	for (uint32_t i = tid; i < cells; i += JP_T) cooc[i] = 0;
	__syncthreads();
	for (uint32_t row = tid; row < m; row += JP_T)
	{
		uint32_t const l = L[row], r = R[row];
		if (l < LC && r < RC) atomicAdd(&cooc[l * RC + r], 1u);
	}
	__syncthreads();
	// compact the non-zero cells in index order: every thread takes a contiguous run of cells
	uint32_t const per = (cells + JP_T - 1) / JP_T, c0 = tid * per, c1 = min(cells, c0 + per);
	uint32_t nz = 0;
	for (uint32_t i = c0; i < c1; ++i) nz += cooc[i] ? 1u : 0u;
	uint32_t total;
	uint32_t k = block_excl_add<JP_T>(nz, scratch, &total);
	if (tid == 0)
	{
		s_base = atomicAdd(cursor, (unsigned long long) total);
		offset[p] = (uint32_t) s_base;
		nedges[p] = total;
	}
	__syncthreads();
	unsigned long long const ob = s_base;
	for (uint32_t i = c0; i < c1; ++i)
	{
		uint32_t const v = cooc[i];
		if (v)
		{
			if (ob + k < cap_total) edges[ob + k] = make_uint2(((i / RC) << 16) | (i % RC), v);
			++k;
		}
	}
}


// ------------------------------------------------------------------------------------------------
// [r5] --output-founders from the RESIDENT alignment (join_context::output_in_permutation_order, join_context.cc:333-356):
// line r is, segment after segment, the substring [lb, rb) of row permutations[s][r]; a slot without a row
// (permutations[s][r] >= m: bipartite_matcher's m_permutation_max) prints '-'.  The alignment is already in HBM, packed
// column-major; the lines leave in one copy instead of 418,000 host memcpy's of 160 bytes from 2.5 GB of raw rows (BASELINE C3).
// Workgroup (x, y): segment x, the lines [row0 + y * rows_per_wg, ...) of this batch; a wave writes one line's piece, its
// lanes along the columns (coalesced stores; the loads hit the segment's ~160 columns in L2).
// out: [rows of the batch][n + 1] bytes, '\n' at column n (written by the workgroups of the last segment).
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_founders(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t bsh, uint32_t const *__restrict__ perm, uint32_t X,
	uint64_t const *__restrict__ seg_lb, uint64_t const *__restrict__ seg_rb, uint32_t S, uint32_t row0, uint32_t nrows, uint32_t rows_per_wg,
	uint8_t const *__restrict__ code_to_byte, uint8_t *__restrict__ out)
{
	__shared__ uint8_t lut[256];
	lut[threadIdx.x] = code_to_byte[threadIdx.x];
	__syncthreads();
	uint32_t const s = blockIdx.x;
	uint64_t const lb = seg_lb[s], rb = seg_rb[s];
	uint32_t const bits = 8u >> bsh, cmask = (1u << bits) - 1u, wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
	uint32_t const r_lo = blockIdx.y * rows_per_wg, r_hi = min(nrows, r_lo + rows_per_wg);
	for (uint32_t r = r_lo + wv; r < r_hi; r += 4u)
	{
		uint32_t const src = perm[(size_t) s * X + row0 + r];
		uint8_t *const line = out + (size_t) r * (n + 1u);
		uint32_t const off = src >> bsh, sh = (src & ((1u << bsh) - 1u)) * bits;
		for (uint64_t k = lb + lane; k < rb; k += 64u)
			line[k] = src < m ? lut[(msa[k * ld + off] >> sh) & cmask] : (uint8_t) '-';
		if (s + 1u == S && lane == 0) line[n] = (uint8_t) '\n';
	}
}

} // namespace fseq
