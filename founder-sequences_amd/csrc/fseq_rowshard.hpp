// fseq_rowshard.hpp -- the pBWT column update with the POSITIONS of the order sharded over ranks: the partition
// BASELINE.json's north_star names ("rows shard across the GPUs with a per-column sigma-bucket-histogram all-reduce
// and a boundary exchange for the divergence scan").  This is the conformance
// path beside the column-block split of fseq_api.hip (which needs one exchange per PHASE, not per column): every
// column costs two all-reduces here, so the sweep runs at the latency of the collective, not at the speed of the
// arithmetic -- DESIGN.md section 6 holds the measured curve.
//
// Rank g of G owns the positions [m g / G, m (g + 1) / G) of the order (a_k, d_k) and the symbols of the ROWS whose
// packed words [cw g / G, cw (g + 1) / G) it was given.  Per column k (reference: the per-column update of
// libbio::pbwt_context as founder_sequences.hh:56-65 drives it -- SURVEY.md Appendix B):
//   X0  the column: every rank writes its rows' packed words into a zeroed buffer, all-reduce (sum) = all-gather
//   X1  the sigma-bucket histogram of the rank's positions      } one all-reduce of a 9-word summary per rank:
//   X2  the carry of the divergence scan over the rank boundary  } counts, running maxima since the last
//       occurrence of every symbol, the symbols seen (exactly the tile carry of the streamed kernels,
//       fseq_core.hpp partition_step<.., TILE>: a rank is a "tile" whose left neighbours live on other GPUs)
//   X3  the scatter: a rank's rows of bucket c land in one contiguous destination range; written into a zeroed
//       full-length (a, d), all-reduce (sum) -- folded into one collective with X0 of the next column
// sigma > 4: a column is ceil(log2 sigma / 2) partitions by 2-bit digits (as everywhere else), each with its own X1-X3.
#pragma once

#include "fseq_stream.hpp"

namespace fseq {

constexpr uint32_t RS_SLOT = 12;              // words per rank in the summary exchange: cnt[4] val[4] has pad[3]

// order = identity, divergences = 0 (the state in front of column 0)
__global__ void k_rs_init(uint32_t *__restrict__ a, uint32_t *__restrict__ d, uint32_t m)
{
	uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < m) { a[i] = i; d[i] = 0u; }
}

// X0: my rows' words of one packed column into the (zeroed) exchange column
__global__ void k_rs_contrib(uint32_t const *__restrict__ col, uint32_t w_lo, uint32_t w_hi, uint32_t *__restrict__ dst)
{
	uint32_t const w = w_lo + blockIdx.x * blockDim.x + threadIdx.x;
	if (w < w_hi) dst[w] = col[w];
}

// tiles of my ml positions through partition_step; APPLY = false: only the summary of the whole range (X1 + X2 out),
// APPLY = true: with the carry of the ranks to my left and the global bucket starts, rows to their destinations (X3 out)
template <bool APPLY>
__global__ __launch_bounds__(ST) void k_rs_sweep(
	uint8_t const *__restrict__ col, uint32_t const *__restrict__ a_src, uint32_t const *__restrict__ d_src, uint32_t ml,
	uint32_t bsh, uint32_t pass, uint32_t first_val, uint32_t *__restrict__ slots, uint32_t rank, uint32_t world,
	uint32_t *__restrict__ xa, uint32_t *__restrict__ xd)
{
	__shared__ StreamLds L;
	uint32_t const tid = threadIdx.x;
	TileCarry tc;
#pragma unroll
	for (int x = 0; x < 4; ++x) { tc.cnt[x] = 0; tc.val[x] = 0; tc.start[x] = 0; }
	tc.has = 0;
	if (APPLY)
	{
		// fold of the ranks to my left (combine(L, R): fseq_core.hpp), totals -> bucket starts
		uint32_t tot[4] = {0, 0, 0, 0};
		for (uint32_t h = 0; h < world; ++h)
		{
			uint32_t const *s = slots + (size_t) h * RS_SLOT;
#pragma unroll
			for (int x = 0; x < 4; ++x) tot[x] += s[x];
			if (h < rank)
			{
				uint32_t const has = s[8];
#pragma unroll
				for (int x = 0; x < 4; ++x)
				{
					tc.cnt[x] += s[x];
					tc.val[x] = ((has >> x) & 1u) ? s[4 + x] : max(tc.val[x], s[4 + x]);
				}
				tc.has |= has;
			}
		}
		uint32_t acc = 0;
#pragma unroll
		for (int x = 0; x < 4; ++x) { tc.start[x] = acc; acc += tot[x]; }
	}
	for (uint32_t base = 0; base < ml; base += SCAP)
	{
		uint32_t a[SE], d[SE], s[SE], dst[SE], dnew[SE];
#pragma unroll
		for (int e = 0; e < SE; ++e)
		{
			uint32_t const pos = base + tid * SE + e;
			bool const in = pos < ml;
			a[e] = in ? a_src[pos] : 0u;
			d[e] = in ? d_src[pos] : 0u;
			s[e] = in ? sym_digit(col, a[e], bsh, pass) : 4u;
		}
		partition_step<ST, SE, 4, true>(d, s, first_val, L.scr, dst, dnew, &tc);
		if (APPLY)
		{
#pragma unroll
			for (int e = 0; e < SE; ++e)
				if (base + tid * SE + e < ml) { xa[dst[e]] = a[e]; xd[dst[e]] = dnew[e]; }
		}
		__syncthreads();
	}
	if (!APPLY && tid == 0)
	{
		uint32_t *s = slots + (size_t) rank * RS_SLOT;
#pragma unroll
		for (int x = 0; x < 4; ++x) { s[x] = tc.cnt[x]; s[4 + x] = tc.val[x]; }
		s[8] = tc.has;
	}
}

} // namespace fseq
