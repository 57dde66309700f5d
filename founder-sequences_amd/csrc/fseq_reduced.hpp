// fseq_reduced.hpp -- [r5] phase C and pass 2 on the rows that differ.
//
// pbwt_context::process (libbio, un-vendored; call sites segmentation_lp_context.cc:52,74,115,156, update_pbwt_task.cc:19)
// costs m rows per column.  What segmentation_lp_context consumes of a column -- the counts of the divergence values,
// output_divergence_value_counts(), lp.cc:78,119 -- is the function  v -> #{ d >= v } = the number of distinct row
// substrings over [v - 1, k], and calculate_segmentation_lp_dp_arg (lp.cc:393-481) reads it from the top down only until
// the cumulative count passes its best value.  Two rows that agree on [vmin - 1, k1) (k1: end of the column block) count
// once in every one of those numbers for every v >= vmin and every column k of the block.  So a block is run on ONE
// representative per class of such rows:
//   * the classes are there already: in the exact state (a1, d1) behind the block (phase B) the rows of a class are
//     contiguous, and a position starts a class iff d1 >= vmin;
//   * the pBWT of a subset of the rows, started from the restriction of the exact state (a0, d0) in front of the block
//     (order kept; divergence = the maximum of d0 over the positions skipped since the last kept row), IS the pBWT of
//     the sub-alignment -- its counts of values >= v are the full run's for every v >= vmin;
//   * vmin is chosen from d0 so that the lists (entries below thr = k + 2 - L while their counts do not exceed X) end above
//     it; a list that takes an entry below vmin flags the block, which is then run on all rows (k_columns): exactness
//     never rests on the choice;
//   * vmin == 1: the rows left out are duplicates over ALL of [0, k1), their divergences are zeros: every list is exact.
// BASELINE C3: ~190 of 2,504 rows per block; C4: ~10,000 of 100,000; C5: ~550 of 10,000 (tests/proto_reduced.py is the
// numpy statement of this file, tests/test_proto_reduced.py its proof against the oracle).
//
// Pass 2 (update_pbwt_task::execute, update_pbwt_task.cc:13-35): the state at a column k inside a block is one CHAIN STEP
// from the block's boundary state -- stable sort of a0 by the class of the row's key prefix [k0, k), range maxima of d0
// inside a class, the divergence in front of a class for its first row -- and the classes and their divergences at k are
// what the reduced run knows at column k (k_columns_red with red.cls: the tables; k_chain_snap: the step).
#pragma once

#include "fseq_kernels.hpp"

namespace fseq {

constexpr uint32_t RED_W = 2048;        // values below the first column's threshold the search for vmin looks at
constexpr int RED_PREP_T = 256;


__host__ __device__ inline size_t reduce_prep_lds_bytes(uint32_t m)
{
	size_t const nw = (m + 31u) / 32u;
	size_t const maps = 2 * ((nw * 4 + 15) & ~size_t(15));
	size_t const hist = RED_W * 4;
	return (maps > hist ? maps : hist) + 256;
}

// one workgroup per block: vmin, the representatives (ascending row id), their block keys, the reduced start state
static __global__ __launch_bounds__(RED_PREP_T) void k_reduce_prep(RedPrepArgs const A)
{
	constexpr int T = RED_PREP_T;
	constexpr uint32_t NWV = T / WAVE;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint32_t const m = A.m;
	uint32_t const nw = (m + 31u) / 32u;
	uint32_t *const big = reinterpret_cast<uint32_t *>(smem);               // the histogram, then bitmap + word prefixes
	uint32_t *const bm = big;
	uint32_t *const wpre = big + ((nw + 3u) & ~3u);
	size_t const big_bytes = reduce_prep_lds_bytes(m) - 256;
	uint32_t *const sscr = reinterpret_cast<uint32_t *>(smem + big_bytes);  // [T / 64 + 1] scan scratch
	uint32_t *const wsum = sscr + 8;                                        // [NWV][3] per-wave {kept, has, tail}
	uint32_t *const misc = sscr + 32;                                       // [0]: bin found

	uint32_t const tid = threadIdx.x, lane = lane_id(), wv = wave_id();
	uint32_t const blk = A.blocks ? A.blocks[blockIdx.x] : A.block0 + blockIdx.x;
	if (tid == 0)
	{
		if (A.invalid) A.invalid[blk] = 0u;
		if (A.flags && blockIdx.x == 0) { A.flags[0] = 0u; A.flags[1] = 0u; }
	}
	uint64_t const k0 = (uint64_t) blk * A.B;
	uint32_t const *const a0 = A.bstate_a + (size_t) blk * m, *const d0 = A.bstate_d + (size_t) blk * m;
	uint32_t const *const a1 = a0 + m, *const d1 = d0 + m;

	// ---- vmin: the largest v with #{ v <= d0 < thr0 } > Xp (thr0: the threshold of the block's first column)
	uint32_t vmin;
	uint32_t const thr0 = (k0 + 2u > (uint64_t) A.L) ? (uint32_t) (k0 + 2u - A.L) : 0u;
	if (A.leaf_only) vmin = (uint32_t) k0 + 1u;
	else if (thr0 == 0u) vmin = 1u;
	else
	{
		uint32_t const lo = thr0 > RED_W ? thr0 - RED_W : 0u;
		for (uint32_t i = tid; i < RED_W; i += T) big[i] = 0u;
		if (tid == 0) misc[0] = 0xFFFFFFFFu;
		__syncthreads();
		for (uint32_t i = tid; i < m; i += T)
		{
			uint32_t const dv = d0[i];
			if (dv < thr0 && dv >= lo) atomicAdd(&big[thr0 - 1u - dv], 1u);
		}
		__syncthreads();
		constexpr uint32_t PER = RED_W / T;
		uint32_t sum = 0;
		for (uint32_t q = 0; q < PER; ++q) sum += big[tid * PER + q];
		uint32_t total;
		uint32_t run = block_excl_add<T>(sum, sscr, &total);
		if (run <= A.Xp && run + sum > A.Xp)
		{
			for (uint32_t q = 0; q < PER; ++q)
			{
				run += big[tid * PER + q];
				if (run > A.Xp) { misc[0] = tid * PER + q; break; }
			}
		}
		__syncthreads();
		uint32_t const found = misc[0];
		vmin = found != 0xFFFFFFFFu ? thr0 - 1u - found : lo;
		if (vmin == 0u) vmin = 1u;
		__syncthreads();
	}

	// ---- the representatives: rows whose position in (a1, d1) starts a class (d1 >= vmin)
	for (uint32_t i = tid; i < 2u * ((nw + 3u) & ~3u); i += T) big[i] = 0u;
	__syncthreads();
	for (uint32_t i = tid; i < m; i += T)
		if (d1[i] >= vmin) { uint32_t const r = a1[i]; atomicOr(&bm[r >> 5], 1u << (r & 31u)); }
	__syncthreads();
	uint32_t Lr;
	{
		uint32_t const per = (nw + T - 1u) / T;
		uint32_t const w0 = tid * per, w1 = min(nw, w0 + per);
		uint32_t sum = 0;
		for (uint32_t w = w0; w < w1; ++w) sum += (uint32_t) __popc(bm[w]);
		uint32_t run = block_excl_add<T>(sum, sscr, &Lr);
		for (uint32_t w = w0; w < w1; ++w) { wpre[w] = run; run += (uint32_t) __popc(bm[w]); }
	}
	__syncthreads();
	if (tid == 0) { A.cnt[blk] = Lr > A.cap ? RED_NONE : Lr; A.vmin[blk] = vmin; }
	if (Lr > A.cap) return;
	size_t const ob = (size_t) blk * A.cap;
	for (uint32_t w = tid; w < nw; w += T)
	{
		uint32_t bits = bm[w], idx = wpre[w];
		while (bits)
		{
			uint32_t const b = (uint32_t) __builtin_ctz(bits);
			bits &= bits - 1u;
			uint32_t const row = w * 32u + b;
			A.rows[ob + idx] = row;
			A.leaf[ob + idx] = A.rank[(size_t) blk * m + row];
			++idx;
		}
	}

	// ---- the start state restricted to them: a wave per contiguous range of positions, 64 at a time
	uint32_t const chunk = ((m + NWV - 1u) / NWV + 63u) & ~63u;
	uint32_t const r0 = wv * chunk, r1 = min(m, r0 + chunk);
	{
		uint32_t kept = 0, has = 0, tail = 0;
		for (uint32_t base = r0; base < r1; base += 64u)
		{
			uint32_t const i = base + lane;
			bool const in = i < r1;
			uint32_t const dv = in ? d0[i] : 0u;
			uint32_t const row = in ? a0[i] : 0u;
			bool const k = in && ((bm[row >> 5] >> (row & 31u)) & 1u);
			unsigned long long const mask = __ballot(k);
			if (mask)
			{
				int const last = 63 - (int) __builtin_clzll(mask);
				tail = readlane_u32(wave_incl_max((int) lane > last ? dv : 0u), 63);
				has = 1u;
				kept += (uint32_t) __popcll(mask);
			}
			else tail = max(tail, readlane_u32(wave_incl_max(dv), 63));
		}
		if (lane == 0) { wsum[3u * wv] = kept; wsum[3u * wv + 1u] = has; wsum[3u * wv + 2u] = tail; }
	}
	__syncthreads();
	{
		uint32_t base_idx = 0, carry = 0;
		for (uint32_t w = 0; w < wv; ++w) base_idx += wsum[3u * w];
		for (int w = (int) wv - 1; w >= 0; --w)
		{
			carry = max(carry, wsum[3u * (uint32_t) w + 2u]);
			if (wsum[3u * (uint32_t) w + 1u]) break;
		}
		for (uint32_t base = r0; base < r1; base += 64u)
		{
			uint32_t const i = base + lane;
			bool const in = i < r1;
			uint32_t const dv = in ? d0[i] : 0u;
			uint32_t const row = in ? a0[i] : 0u;
			uint32_t const word = bm[row >> 5];
			bool const k = in && ((word >> (row & 31u)) & 1u);
			unsigned long long const mask = __ballot(k);
			// maximum of d0 over (the last kept position in front of me, me]: a segmented inclusive max-scan, a segment
			// starts behind every kept lane; lanes with no kept lane in front of them in this group also take the carry
			uint32_t v = dv;
			uint32_t f = (lane > 0u && ((mask >> (lane - 1u)) & 1ull)) ? 1u : 0u;
#pragma unroll
			for (int delta = 1; delta < 64; delta <<= 1)
			{
				uint32_t const v2 = shfl_up_u32(v, delta), f2 = shfl_up_u32(f, delta);
				if ((int) lane >= delta && !f) { v = max(v, v2); f |= f2; }
			}
			unsigned long long const below = mask & ((1ull << lane) - 1ull);
			if (!below) v = max(v, carry);
			if (k)
			{
				uint32_t const idx = base_idx + (uint32_t) __popcll(below);
				A.a[ob + idx] = A.direct ? row : wpre[row >> 5] + (uint32_t) __popc(word & ((1u << (row & 31u)) - 1u));
				A.d[ob + idx] = v;
			}
			if (mask)
			{
				int const last = 63 - (int) __builtin_clzll(mask);
				carry = readlane_u32(wave_incl_max((int) lane > last ? dv : 0u), 63);
				base_idx += (uint32_t) __popcll(mask);
			}
			else carry = max(carry, readlane_u32(wave_incl_max(dv), 63));
		}
	}
}

// The plan of a run is the plan of the run before it on the same input (the counts have not changed: same input, same
// capacity): flags[1] = 1 if they have after all (the host then plans afresh and runs the attempt again).
static __global__ __launch_bounds__(256) void k_reduce_check(uint32_t const *__restrict__ cnt, uint32_t const *__restrict__ planned, uint32_t count, uint32_t *__restrict__ flags)
{
	uint32_t const i = blockIdx.x * 256u + threadIdx.x;
	if (i < count && cnt[i] != planned[i]) flags[1] = 1u;
}

// The packed columns of the representatives: column k of the reduced alignment at red + k * ldr, representative i at byte
// i >> bsh (same packing as the alignment).  Workgroup (x, y): block blocks[x], output bytes [64 y, 64 y + 64) -- a wave is 64
// consecutive output bytes of one column (the representatives ascend by row: the gathered bytes ascend too), the four
// waves take the block's columns in turn.
static __global__ __launch_bounds__(256) void k_reduce_msa(
	uint8_t const *__restrict__ msa, size_t ld, uint8_t *__restrict__ red, size_t ldr, uint32_t const *__restrict__ cnt,
	uint32_t const *__restrict__ rows, uint32_t cap, uint64_t n, uint32_t B, uint32_t bsh, uint32_t const *__restrict__ blocks)
{
	uint32_t const blk = blocks[blockIdx.x];
	uint32_t const Lr = cnt[blk];
	if (Lr == RED_NONE) return;
	uint32_t const rpb = 1u << bsh, bits = 8u >> bsh, cmask = (1u << bits) - 1u;
	uint32_t const nq = (Lr + rpb - 1u) >> bsh;
	uint32_t const q = blockIdx.y * 64u + lane_id();
	if (blockIdx.y * 64u >= nq) return;
	uint64_t const k0 = (uint64_t) blk * B;
	uint32_t const nb = (uint32_t) (((k0 + B < n) ? k0 + B : n) - k0);
	uint32_t off[4], sh[4];
#pragma unroll
	for (uint32_t i = 0; i < 4; ++i)
	{
		uint32_t const idx = q * rpb + i;
		uint32_t const row = (i < rpb && idx < Lr) ? rows[(size_t) blk * cap + idx] : 0u;
		off[i] = row >> bsh;
		sh[i] = (row & (rpb - 1u)) * bits;
	}
	if (q >= nq) return;
	// (four columns per iteration: their sixteen gathers in flight together -- a wave is otherwise one round trip to L2 / HBM
	// per column, and the kernel a third below what the alignment's 125 GB take to read on BASELINE C4)
	constexpr uint32_t U = 4;
	for (uint32_t j0 = wave_id(); j0 < nb; j0 += 4u * U)
	{
		uint32_t b[U][4];
#pragma unroll
		for (uint32_t u = 0; u < U; ++u)
		{
			uint32_t const j = min(j0 + 4u * u, nb - 1u);
			uint8_t const *const col = msa + (k0 + j) * ld;
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i) b[u][i] = i < rpb ? col[off[i]] : 0u;
		}
#pragma unroll
		for (uint32_t u = 0; u < U; ++u)
		{
			uint32_t const j = j0 + 4u * u;
			uint32_t v = 0;
#pragma unroll
			for (uint32_t i = 0; i < 4; ++i)
				if (i < rpb) v |= ((b[u][i] >> sh[i]) & cmask) << (i * bits);
			if (j < nb) red[(k0 + j) * ldr + q] = (uint8_t) v;
		}
	}
}

// The same with the alignment's column staged in LDS (a column of at most R x 16 KB: 131,072 rows of 2-bit symbols).  The
// form above gathers bytes at ~7,000 of BASELINE C4's 100,000 rows straight from memory -- every line of the column comes in
// for a few of its bytes, by way of some thirty workgroups per block on as many CUs, and the gathers' round trips bound it
// (3.1 TB/s over the alignment's 125 GB).  Here workgroup (x, y) takes a quarter of block blocks[x]'s columns: a column is
// R LDS-DMA pieces per lane, contiguous, one column ahead of the one being gathered (s_waitcnt vmcnt(R): the pieces of the
// column in front have landed, the next one's stay in flight -- loads return in order); the representatives' rows stay in
// registers (output byte q = thread + 1024 i, its 1 << BSH rows), the gathers are LDS byte reads.
template <int BSH, int R>
__global__ __launch_bounds__(1024) void k_reduce_msa_lds(
	uint8_t const *__restrict__ msa, size_t ld, uint8_t *__restrict__ red, size_t ldr, uint32_t const *__restrict__ cnt,
	uint32_t const *__restrict__ rows, uint32_t cap, uint64_t n, uint32_t B, uint32_t const *__restrict__ blocks, uint32_t colbytes)
{
	constexpr uint32_t RPB = 1u << BSH, BITS = 8u >> BSH, CMASK = (1u << BITS) - 1u;
	constexpr uint32_t NOUT = (12288u / RPB + 1023u) / 1024u;          // output bytes per thread: 12,288 representatives at most
	constexpr uint32_t BUF = (uint32_t) R * 16384u;
	extern __shared__ __attribute__((aligned(16))) char smem[];         // two columns of BUF bytes
	uint32_t const blk = blocks[blockIdx.x];
	uint32_t const Lr = cnt[blk];
	if (Lr == RED_NONE) return;
	uint32_t const tid = threadIdx.x;
	uint32_t const nq = (Lr + RPB - 1u) >> BSH;
	uint64_t const k0 = (uint64_t) blk * B;
	uint32_t const nb = (uint32_t) (((k0 + B < n) ? k0 + B : n) - k0);
	uint32_t const per = (nb + gridDim.y - 1u) / gridDim.y;
	uint32_t const j_lo = min(nb, blockIdx.y * per), j_hi = min(nb, j_lo + per);
	if (j_lo >= j_hi) return;
	uint32_t row[NOUT][RPB];
#pragma unroll
	for (uint32_t i = 0; i < NOUT; ++i)
#pragma unroll
		for (uint32_t r = 0; r < RPB; ++r)
		{
			uint32_t const idx = (tid + 1024u * i) * RPB + r;
			row[i][r] = idx < Lr ? rows[(size_t) blk * cap + idx] : 0u;
		}
	uint32_t const base = (uint32_t) (uintptr_t) smem;
	auto stage = [&](uint64_t k, uint32_t buf) {
		uint8_t const *const col = msa + k * ld;
#pragma unroll
		for (uint32_t r = 0; r < (uint32_t) R; ++r)
		{
			// (every wave issues R pieces -- the count s_waitcnt goes by; a piece past the column's bytes re-reads its first)
			uint32_t const off = r * 16384u + tid * 16u;
			lds_dma16(col + (off < colbytes ? off : 0u), __builtin_amdgcn_readfirstlane(base + buf * BUF + r * 16384u + wave_id() * 1024u));
		}
	};
	stage(k0 + j_lo, 0u);
	for (uint32_t j = j_lo; j < j_hi; ++j)
	{
		uint32_t const cur = (j - j_lo) & 1u;
		if (j + 1u < j_hi)
		{
			stage(k0 + j + 1u, cur ^ 1u);
			if constexpr (R == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
		}
		else
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		lds_only_barrier();
		uint8_t const *const colb = reinterpret_cast<uint8_t const *>(smem) + cur * BUF;
#pragma unroll
		for (uint32_t i = 0; i < NOUT; ++i)
		{
			uint32_t const q = tid + 1024u * i;
			uint32_t v = 0;
#pragma unroll
			for (uint32_t r = 0; r < RPB; ++r)
				v |= (((uint32_t) colb[row[i][r] >> BSH] >> ((row[i][r] & (RPB - 1u)) * BITS)) & CMASK) << (r * BITS);
			if (q < nq) red[(k0 + j) * ldr + q] = (uint8_t) v;
		}
		lds_only_barrier();                   // (everybody has read the column the piece after next lands on)
	}
}

// ------------------------------------------------------------------------------------------------
// Pass 2: one chain step per boundary (k_chain's step with the class tables of k_columns_red as the key block).
// Workgroup t: boundary task t in block task_blk[t]; ncls[t] == 0: the boundary IS the block's boundary state (a copy).
// ------------------------------------------------------------------------------------------------
template <int T, int E, bool PK>
__global__ __launch_bounds__(T) void k_chain_snap(
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d, uint32_t const *__restrict__ rank,
	uint32_t m, uint32_t const *__restrict__ task_blk, uint32_t const *__restrict__ cls, uint32_t const *__restrict__ headd,
	uint32_t const *__restrict__ ncls, uint32_t cap, uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d, uint32_t keyed)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	using AT = std::conditional_t<PK, uint16_t, uint32_t>;
	Carver cv{smem};
	AT *a_l = cv.take<AT>(CAP);
	AT *rk = cv.take<AT>(CAP);
	uint32_t *d_l = cv.take<uint32_t>(CAP);
	uint32_t *kd = cv.take<uint32_t>(CAP);
	StepScratch<T, 4> &scr = *cv.take<StepScratch<T, 4>>(1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint32_t const task = blockIdx.x;
	uint32_t const blk = task_blk[task];
	uint32_t const D = ncls[task];
	size_t const sb = (size_t) blk * m, tb = (size_t) task * cap, ob = (size_t) task * m;
	if (D == 0u)
	{
		for (uint32_t idx = tid; idx < m; idx += T) { snap_a[ob + idx] = bstate_a[sb + idx]; snap_d[ob + idx] = bstate_d[sb + idx]; }
		return;
	}
#pragma unroll
	for (int i = 0; i < E; ++i)
	{
		uint32_t idx = tid + i * T;
		asm volatile("" : "+v"(idx));
		bool const in = idx < m;
		a_l[idx] = (AT) (in ? bstate_a[sb + idx] : 0u);
		d_l[idx] = in ? bstate_d[sb + idx] : 0u;
		rk[idx] = (AT) (in ? cls[tb + rank[sb + idx]] : 0u);
		kd[idx] = idx < D ? headd[tb + idx] : 0u;
	}
	__syncthreads();
	uint32_t const nd = rank_digits(D);
	for (uint32_t p = 0; p < nd; ++p)
	{
		uint32_t a[E], d[E], s[E], dst[E], dnew[E];
		read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
		for (int e = 0; e < E; ++e)
			s[e] = (p0 + e < m) ? (((uint32_t) rk[a[e]] >> (2u * p)) & 3u) : 4u;
		if (keyed == 1u) partition_step<T, E, 4, false, false, scan_shift_for(T, E)>(d, s, 0u, scr, dst, dnew);
		else if (keyed == 2u) partition_step<T, E, 4, false, false, 25, false, false, true>(d, s, 0u, scr, dst, dnew);
		else partition_step<T, E, 4>(d, s, 0u, scr, dst, dnew);
#pragma unroll
		for (int e = 0; e < E; ++e)
			if (p0 + e < m) { a_l[dst[e]] = (AT) a[e]; d_l[dst[e]] = dnew[e]; }
		__syncthreads();
	}
	{
		uint32_t rprev = (p0 > 0 && p0 - 1 < m) ? rk[a_l[p0 - 1]] : PAD_KEY;
		uint32_t dn[E];
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			dn[e] = 0;
			if (pos < m)
			{
				uint32_t const r = rk[a_l[pos]];
				dn[e] = (pos == 0 || r != rprev) ? kd[r] : d_l[pos];
				rprev = r;
			}
		}
#pragma unroll
		for (int e = 0; e < E; ++e)
			if (p0 + e < m) d_l[p0 + e] = dn[e];
	}
	__syncthreads();
	for (uint32_t idx = tid; idx < m; idx += T) { snap_a[ob + idx] = a_l[idx]; snap_d[ob + idx] = d_l[idx]; }
}

template <int T, int E, bool PK>
__host__ __device__ inline size_t chain_snap_lds_bytes()
{
	constexpr size_t CAP = (size_t) T * E;
	return 2 * carve_bytes(CAP, PK ? 2 : 4) + 2 * carve_bytes(CAP, 4) + carve_bytes(1, sizeof(StepScratch<T, 4>));
}

} // namespace fseq
