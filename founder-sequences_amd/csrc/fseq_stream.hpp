// fseq_stream.hpp -- the column kernels for orders that do not fit LDS (m > 11,264 rows; BASELINE
// config C4 has m = 100,000).  Same algorithm and the same partition_step as fseq_kernels.hpp, but
// the order (a, d) of a block lives in a per-block HBM/L2 workspace (two buffers, ping-pong) and a
// partition pass streams it through the workgroup tile by tile (ST * SE = 4096 rows) with a running
// TileCarry.  A column pass = the bucket sizes counted off the staged column + one partition sweep (one
// read and one write of a and d: SURVEY.md's 17 B/cell as real HBM traffic); a rank-digit or key pass
// has a counting sweep over the order first.
// Only the current column (m bytes) is staged in LDS.
#pragma once

#include "fseq_kernels.hpp"

namespace fseq {

constexpr int ST = 1024, SE = 4;                   // measured on m = 100,000 with the keyed scan and the full-tile path: phase C 32.7 / 25.5 / 29.1 / 28.3 / 30.6 ms for SE = 3 / 4 / 5 / 6 / 7 (no spills at 4)
constexpr uint32_t SCAP = ST * SE;
constexpr uint32_t STREAM_MAX_COLBYTES = 147456;   // column staging: one packed column (sym_bytes(m, bsh)) in LDS

// Stride states of the streamed regime are packed when a row id and a divergence fit 40 bits together (ss_pack = bits
// of a row id, 0 = not packed): word = a | d << ss_pack in the first array, byte = d >> (32 - ss_pack) in the second
// (ss_high_stride(m) bytes per state) -- 5 instead of 8 bytes per row, so the states can lie closer together.
__host__ __device__ inline size_t ss_high_stride(uint32_t m) { return ((size_t) m + 15) & ~size_t(15); }

struct StreamLds {
	StepScratch<ST, 4> scr;
	uint32_t red[4 * (ST / WAVE) + 8];
	uint32_t sink[WAVE];                             // where stream_touch's loads land (never read)
	uint32_t dh[64];                                 // rank_digit_counts: bucket sizes of up to 16 digit passes
	uint32_t dhw[ST / WAVE][64];                     // ... per wave
};

// staged: plus the 2 x SCAP words in which stream_pass lays a tile out in output order
__host__ __device__ inline size_t stream_lds_bytes(uint32_t colbytes, bool staged)
{
	return carve_bytes((size_t) colbytes + 16, 1) + carve_bytes(1, sizeof(StreamLds)) + (staged ? carve_bytes(2 * (size_t) SCAP, 4) : 0);
}

struct DigitColumn {
	uint8_t const *sym; uint32_t bsh, pass;
	__device__ __forceinline__ uint32_t operator()(uint32_t a) const { return sym_digit(sym, a, bsh, pass); }
};
struct DigitRank {
	uint32_t const *rank; uint32_t shift;
	__device__ __forceinline__ uint32_t operator()(uint32_t a) const { return (rank[a] >> shift) & 3u; }
};
struct DigitKey {
	uint32_t shift;
	__device__ __forceinline__ uint32_t operator()(uint32_t a) const { return (a >> shift) & 3u; }
};
struct NoHook {
	__device__ __forceinline__ void operator()(uint32_t, uint32_t) const {}
};

// block-wide sum of four counters; every thread gets the totals
__device__ __forceinline__ void block_sum4(uint32_t (&c)[4], uint32_t *red)
{
	uint32_t const lane = lane_id(), wave = wave_id();
#pragma unroll
	for (int x = 0; x < 4; ++x) c[x] = readlane_u32(wave_incl_add(c[x]), 63);
	__syncthreads();
	if (lane == 0)
#pragma unroll
		for (int x = 0; x < 4; ++x) red[wave * 4 + x] = c[x];
	__syncthreads();
#pragma unroll
	for (int x = 0; x < 4; ++x) c[x] = 0;
	for (int w = 0; w < ST / WAVE; ++w)
#pragma unroll
		for (int x = 0; x < 4; ++x) c[x] += red[w * 4 + x];
}

// Bucket sizes of a column pass do not depend on the order: count the digit straight off the staged column -- a
// word of 16 / 8 / 4 packed symbols at a time: the two bits of the digit of every symbol as two masks, four popcounts.
__device__ __forceinline__ void column_digit_counts(uint8_t const *sym, uint32_t m, uint32_t bsh, uint32_t pass, uint32_t (&cnt)[4], uint32_t *red)
{
#pragma unroll
	for (int x = 0; x < 4; ++x) cnt[x] = 0;
	uint32_t const spw = 4u << bsh, bits = 8u >> bsh;          // symbols per 32-bit word, bits per symbol
	uint32_t const nwords = (sym_bytes(m, bsh) + 3u) / 4u;
	uint32_t const ones = bsh == 2 ? 0x55555555u : bsh == 1 ? 0x11111111u : 0x01010101u;      // bit 0 of every symbol
	for (uint32_t wi = threadIdx.x; wi < nwords; wi += ST)
	{
		uint32_t const w = *reinterpret_cast<uint32_t const *>(sym + wi * 4u);
		uint32_t const r0 = wi * spw;
		// symbols of rows behind m (the last word only) do not count
		uint32_t const nv = min(spw, m - r0);
		uint32_t const valid = nv == spw ? ones : (ones & ((1u << (nv * bits)) - 1u));
		uint32_t const lo = (w >> (2u * pass)) & valid, hi = (w >> (2u * pass + 1u)) & valid;
		cnt[0] += (uint32_t) __popc(valid & ~lo & ~hi);
		cnt[1] += (uint32_t) __popc(lo & ~hi);
		cnt[2] += (uint32_t) __popc(hi & ~lo);
		cnt[3] += (uint32_t) __popc(lo & hi);
	}
	block_sum4(cnt, red);
}

// Bucket sizes of ALL digit passes over a key block's ranks at once: they do not depend on the order, so one coalesced
// sweep over rank[0 .. m) replaces the counting sweep in front of every pass (nd gathers of rk[a[i]] over all rows --
// about 40 % of a pass).  Counts per (pass, digit) by ballot + popcount (uniform accumulators), L.dh[4 p + x] when done.
__device__ __forceinline__ void rank_digit_counts(uint32_t const *__restrict__ rk, uint32_t m, uint32_t nd, StreamLds &L)
{
	uint32_t const tid = threadIdx.x, wave = tid / WAVE, lane = tid % WAVE;
	uint32_t c[16][3];
#pragma unroll
	for (int p = 0; p < 16; ++p) { c[p][0] = 0; c[p][1] = 0; c[p][2] = 0; }
	uint32_t rows = 0;                                         // rows this wave has seen (digit 3 = the rest)
	for (uint32_t base = 0; base < m; base += ST)
	{
		uint32_t const r = base + tid;
		bool const in = r < m;
		uint32_t const v = in ? rk[r] : 0u;
		rows += (uint32_t) __popcll(__ballot(in));
#pragma unroll
		for (int p = 0; p < 16; ++p)
			if ((uint32_t) p < nd)
			{
				uint32_t const g = (v >> (2 * p)) & 3u;
#pragma unroll
				for (int x = 0; x < 3; ++x) c[p][x] += (uint32_t) __popcll(__ballot(in && g == (uint32_t) x));
			}
	}
	if (lane == 0)
	{
#pragma unroll
		for (int p = 0; p < 16; ++p)
		{
			L.dhw[wave][4 * p] = c[p][0]; L.dhw[wave][4 * p + 1] = c[p][1]; L.dhw[wave][4 * p + 2] = c[p][2];
			L.dhw[wave][4 * p + 3] = rows - c[p][0] - c[p][1] - c[p][2];
		}
	}
	__syncthreads();
	if (tid < 64u)
	{
		uint32_t t = 0;
		for (uint32_t w = 0; w < ST / WAVE; ++w) t += L.dhw[w][tid];
		L.dh[tid] = t;
	}
	__syncthreads();
}

// One stable 4-bucket partition pass over an order of m rows held in global memory:
// (a_src, d_src) -> (a_dst, d_dst).  KEYS: sort keys only (no divergences).  pre_cnt: bucket sizes if the
// caller already knows them (column passes), else a counting sweep comes first.  Ends with a barrier.
template <bool KEYS, int KS = 0, bool KO = false, typename DF, typename HOOK>
__device__ __forceinline__ void stream_pass(
	uint32_t m, uint32_t const *a_src, uint32_t const *d_src, uint32_t *a_dst, uint32_t *d_dst,
	uint32_t first_val, DF const digit, HOOK const hook, StreamLds &L, uint32_t const *pre_cnt = nullptr, uint32_t *stage = nullptr)
{
	uint32_t const tid = threadIdx.x;
	uint32_t cnt[4] = {0, 0, 0, 0};
	if (pre_cnt)
	{
#pragma unroll
		for (int x = 0; x < 4; ++x) cnt[x] = pre_cnt[x];
	}
	else
	{
		// ---- sweep 1: bucket sizes
		for (uint32_t base = 0; base < m; base += SCAP)
		{
#pragma unroll
			for (int e = 0; e < SE; ++e)
			{
				uint32_t const pos = base + tid * SE + e;
				if (pos < m)
				{
					uint32_t const g = digit(a_src[pos]);
#pragma unroll
					for (int x = 0; x < 4; ++x) cnt[x] += (g == (uint32_t) x) ? 1u : 0u;
				}
			}
		}
		block_sum4(cnt, L.red);
	}
	TileCarry tc;
	{
		uint32_t acc = 0;
#pragma unroll
		for (int x = 0; x < 4; ++x) { tc.cnt[x] = 0; tc.val[x] = 0; tc.start[x] = acc; acc += cnt[x]; }
		tc.has = 0;
	}
	// ---- sweep 2: the partition, tile by tile.  With a staging buffer (2 x SCAP words of LDS) the tile is
	// first laid out in LDS in output order -- its four bucket runs back to back -- and written out with
	// consecutive lanes on consecutive words: the direct scatter writes every 32-byte sector in several
	// partial pieces (measured: 2x the write traffic, 1.4x the time of phase A at m = 100,000).
	uint32_t const sink = __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) L.sink);
	// One tile.  FULL: all SCAP positions hold rows (every tile but the last): no bounds in the loads, the staging and
	// the write-out -- the seven "position < m" predicates alone were seven SGPR pairs in a kernel that spills them.
	auto tile = [&](uint32_t base, auto full_tag) {
		constexpr bool FULL = decltype(full_tag)::value;
		uint32_t a[SE], d[SE], s[SE], dst[SE], dnew[SE];
#pragma unroll
		for (int e = 0; e < SE; ++e)
		{
			uint32_t const pos = base + tid * SE + e;
			bool const in = FULL || pos < m;
			a[e] = in ? a_src[pos] : 0u;
			d[e] = (in && !KEYS) ? d_src[pos] : 0u;
			s[e] = in ? digit(a[e]) : 4u;
		}
		// The next tile's rows on their way from HBM into L2 while this tile is partitioned (one workgroup per CU: no
		// other wave would hide that latency): one touch per 64 bytes, the first half of the workgroup for a, the
		// second for d.
		if (FULL)
		{
			uint32_t const w = base + SCAP + (tid & (ST / 2 - 1)) * 16u;
			bool const second = tid >= ST / 2;
			if (w < m && (tid & (ST / 2 - 1)) < SCAP / 16u && !(second && KEYS))
				stream_touch((second ? d_src : a_src) + w, sink);
		}
		uint32_t gs[4];
#pragma unroll
		for (int x = 0; x < 4; ++x) gs[x] = tc.start[x] + tc.cnt[x];       // where this tile's rows of bucket x go
		partition_step<ST, SE, 4, true, false, KS, false, false, KO>(d, s, first_val, L.scr, dst, dnew, &tc);
		if (stage)
		{
			// tile-local start of every bucket run, and what turns a global destination into a tile-local one
			uint32_t lofs[4], shift[4];
			{
				uint32_t acc = 0;
#pragma unroll
				for (int x = 0; x < 4; ++x) { lofs[x] = acc; shift[x] = gs[x] - acc; acc += tc.start[x] + tc.cnt[x] - gs[x]; }
			}
#pragma unroll
			for (int e = 0; e < SE; ++e)
			{
				if (FULL || base + tid * SE + e < m)
				{
					// (a 4-way select as a tree over the two bits of the symbol)
					bool const b0 = s[e] & 1u, b1 = s[e] & 2u;
					uint32_t const lo2 = b0 ? shift[1] : shift[0], hi2 = b0 ? shift[3] : shift[2];
					uint32_t const lp = dst[e] - (b1 ? hi2 : lo2);
					stage[lp] = a[e];
					if (!KEYS) { stage[SCAP + lp] = dnew[e]; hook(d[e], dnew[e]); }
				}
			}
			__syncthreads();
			uint32_t const tile_n = FULL ? SCAP : m - base;
#pragma unroll
			for (int e = 0; e < SE; ++e)
			{
				uint32_t const j = (uint32_t) e * ST + tid;
				if (FULL || j < tile_n)
				{
					// bucket of output slot j: the runs lie back to back
					uint32_t sh = shift[0];
					sh = j >= lofs[1] ? shift[1] : sh;
					sh = j >= lofs[2] ? shift[2] : sh;
					sh = j >= lofs[3] ? shift[3] : sh;
					a_dst[j + sh] = stage[j];
					if (!KEYS) d_dst[j + sh] = stage[SCAP + j];
				}
			}
		}
		else
		{
#pragma unroll
			for (int e = 0; e < SE; ++e)
			{
				if (FULL || base + tid * SE + e < m)
				{
					a_dst[dst[e]] = a[e];
					if (!KEYS) { d_dst[dst[e]] = dnew[e]; hook(d[e], dnew[e]); }
				}
			}
		}
		__syncthreads();
	};
	uint32_t base = 0;
	for (; base + SCAP <= m; base += SCAP) tile(base, std::true_type{});
	if (base < m) tile(base, std::false_type{});
}

__device__ __forceinline__ void stage_column(uint8_t *sym, uint8_t const *col, uint32_t colbytes)
{
	for (uint32_t i = threadIdx.x * 16u; i < colbytes; i += ST * 16u)
		*reinterpret_cast<uint4 *>(sym + i) = *reinterpret_cast<uint4 const *>(col + i);
}

// rank / keyd / nkeys of the order in (a, d): a row starts a new key iff d > dlow (position 0 always).
__device__ __forceinline__ void stream_emit_ranks(
	uint32_t m, uint32_t const *a, uint32_t const *d, uint32_t dlow, uint32_t *rank_out, uint32_t *keyd_out, uint32_t *nkeys_out,
	StreamLds &L)
{
	uint32_t const tid = threadIdx.x;
	uint32_t running = 0;
	for (uint32_t base = 0; base < m; base += SCAP)
	{
		uint32_t av[SE], dv[SE], nf = 0;
#pragma unroll
		for (int e = 0; e < SE; ++e)
		{
			uint32_t const pos = base + tid * SE + e;
			av[e] = pos < m ? a[pos] : 0u;
			dv[e] = pos < m ? d[pos] : 0u;
			nf += (pos < m && (pos == 0 || dv[e] > dlow)) ? 1u : 0u;
		}
		uint32_t total;
		uint32_t r = running + block_excl_add<ST>(nf, L.red, &total);
#pragma unroll
		for (int e = 0; e < SE; ++e)
		{
			uint32_t const pos = base + tid * SE + e;
			if (pos < m)
			{
				bool const first = (pos == 0 || dv[e] > dlow);
				r += first ? 1u : 0u;
				rank_out[av[e]] = r - 1u;
				if (first) keyd_out[r - 1u] = dv[e];
			}
		}
		running += total;
		__syncthreads();
	}
	if (tid == 0) *nkeys_out = running;
}

// ------------------------------------------------------------------------------------------------
// phase A (MODE_RANK) and pass 2 (MODE_SNAP), streamed.  ws: [gridDim.x][4][m] words (a0, d0, a1, d1).
// ------------------------------------------------------------------------------------------------
// KO: every divergence is below 2^25 (n is): the partition steps scan occurrence keys (fseq_core.hpp) instead of {has, value}
// (pass 2 at 64 registers, 19 spilled: two of these workgroups share a CU -- the staged column and the tile staging
// are 58 KiB -- and hide each other's memory latency: BASELINE C4 pass 2 168 -> 157 ms; k_chain_stream the same way: 140 -> 150)
template <int MODE, bool KO = false>
__global__ __launch_bounds__(ST, MODE == MODE_SNAP ? 8 : 1) void k_colblock_stream(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblocks, uint32_t npass, uint32_t bsh, uint32_t *ws, uint32_t staged,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint64_t const *__restrict__ task_rb, uint2 const *__restrict__ task_grp,
	uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d,
	uint64_t const *__restrict__ task_src, uint32_t snap_stride, uint32_t const *__restrict__ ss_a, uint32_t const *__restrict__ ss_d,
	uint64_t col0, uint32_t ss_pack)
{
	// (MODE_RANK: bstate_a, if given, is a per-block filter as in k_colblock)
	if (MODE == MODE_RANK && bstate_a && bstate_a[blockIdx.x] == 0u) return;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	uint8_t *sym = cv.take<uint8_t>((size_t) sym_bytes(m, bsh) + 16);
	StreamLds &L = *cv.take<StreamLds>(1);
	uint32_t *const stage = staged ? cv.take<uint32_t>(2 * (size_t) SCAP) : nullptr;
	uint32_t const tid = threadIdx.x;
	uint32_t *buf[2][2];
	{
		uint32_t *w = ws + (size_t) blockIdx.x * 4u * m;
		buf[0][0] = w; buf[0][1] = w + m; buf[1][0] = w + 2u * (size_t) m; buf[1][1] = w + 3u * (size_t) m;
	}
	uint64_t k0, kend;
	uint32_t t_first = 0, t_count = 0, t_next = 0;
	if (MODE == MODE_RANK)
	{
		k0 = col0 + (uint64_t) blockIdx.x * B;
		kend = (k0 + B < n) ? k0 + B : n;
		for (uint32_t i = tid; i < m; i += ST) { buf[0][0][i] = i; buf[0][1][i] = (uint32_t) k0; }
	}
	else
	{
		uint2 const grp = task_grp[blockIdx.x];
		t_first = grp.x; t_count = grp.y;
		uint64_t const src = task_src[blockIdx.x];
		bool const from_stride = (src >> 63) != 0;
		uint64_t const sidx = src & ~(1ull << 63);
		k0 = from_stride ? sidx * snap_stride : sidx * B;
		kend = task_rb[t_first + t_count - 1u];
		uint32_t const *sa = (from_stride ? ss_a : bstate_a) + sidx * (size_t) m;
		if (from_stride && ss_pack)
		{
			// packed stride states (ss_unpack): 5 bytes per row
			uint8_t const *sh = reinterpret_cast<uint8_t const *>(ss_d) + sidx * (size_t) ss_high_stride(m);
			for (uint32_t i = tid; i < m; i += ST) { uint32_t const w = sa[i]; buf[0][0][i] = w & ((1u << ss_pack) - 1u); buf[0][1][i] = (w >> ss_pack) | ((uint32_t) sh[i] << (32u - ss_pack)); }
		}
		else
		{
			uint32_t const *sd = (from_stride ? ss_d : bstate_d) + sidx * (size_t) m;
			for (uint32_t i = tid; i < m; i += ST) { buf[0][0][i] = sa[i]; buf[0][1][i] = sd[i]; }
		}
	}
	__syncthreads();
	uint32_t cur = 0;
	auto snapshot_if_requested = [&](uint64_t k) {
		if (MODE == MODE_SNAP && t_next < t_count && task_rb[t_first + t_next] == k)
		{
			size_t const ob = (size_t) (t_first + t_next) * m;
			for (uint32_t i = tid; i < m; i += ST) { snap_a[ob + i] = buf[cur][0][i]; snap_d[ob + i] = buf[cur][1][i]; }
			++t_next;
		}
	};
	snapshot_if_requested(k0);
	for (uint64_t k = k0; k < kend; ++k)
	{
		stage_column(sym, msa + k * ld, sym_bytes(m, bsh));
		__syncthreads();
		for (uint32_t pass = 0; pass < npass; ++pass)
		{
			uint32_t cnt4[4];
			column_digit_counts(sym, m, bsh, pass, cnt4, L.red);
			stream_pass<false, KO ? 25 : 0, KO>(m, buf[cur][0], buf[cur][1], buf[cur ^ 1u][0], buf[cur ^ 1u][1], (uint32_t) (k + 1),
			                                    DigitColumn{sym, bsh, pass}, NoHook{}, L, cnt4, stage);
			cur ^= 1u;
		}
		snapshot_if_requested(k + 1);
	}
	if (MODE == MODE_RANK)
		stream_emit_ranks(m, buf[cur][0], buf[cur][1], (uint32_t) k0, rank + (size_t) blockIdx.x * m, keyd + (size_t) blockIdx.x * m,
		                  nkeys + blockIdx.x, L);
}

// ------------------------------------------------------------------------------------------------
// phase B, streamed (same contract as k_chain).  ws: [gridDim.x][4][m] words.
// ------------------------------------------------------------------------------------------------
template <bool KO = false>
__global__ __launch_bounds__(ST) void k_chain_stream(
	uint32_t const *__restrict__ rank, uint32_t const *__restrict__ keyd, uint32_t const *__restrict__ nkeys,
	uint32_t m, uint32_t nb_total, uint32_t G, uint64_t cols_per_block, uint32_t *ws, uint32_t staged,
	uint32_t const *__restrict__ start_a, uint32_t const *__restrict__ start_d,
	uint32_t *__restrict__ out_state_a, uint32_t *__restrict__ out_state_d,
	uint32_t *__restrict__ out_rank, uint32_t *__restrict__ out_keyd, uint32_t *__restrict__ out_nkeys, uint32_t grp0)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	StreamLds &L = *cv.take<StreamLds>(1);
	uint32_t *const stage = staged ? cv.take<uint32_t>(2 * (size_t) SCAP) : nullptr;
	uint32_t const tid = threadIdx.x;
	uint32_t *buf[2][2];
	{
		uint32_t *w = ws + (size_t) blockIdx.x * 4u * m;
		buf[0][0] = w; buf[0][1] = w + m; buf[1][0] = w + 2u * (size_t) m; buf[1][1] = w + 3u * (size_t) m;
	}
	uint32_t const grp = blockIdx.x + grp0;                     // chain index (workspaces stay per workgroup)
	uint32_t const b0 = grp * G;
	uint32_t const b1 = min(nb_total, b0 + G);
	uint32_t const kstart = (uint32_t) ((uint64_t) b0 * cols_per_block);
	for (uint32_t i = tid; i < m; i += ST)
	{
		buf[0][0][i] = start_a ? start_a[(size_t) grp * m + i] : i;
		buf[0][1][i] = start_d ? start_d[(size_t) grp * m + i] : kstart;
	}
	__syncthreads();
	uint32_t cur = 0;
	for (uint32_t b = b0; b < b1; ++b)
	{
		uint32_t const *rk = rank + (size_t) b * m, *kd = keyd + (size_t) b * m;
		if (out_state_a)
			for (uint32_t i = tid; i < m; i += ST) { out_state_a[(size_t) b * m + i] = buf[cur][0][i]; out_state_d[(size_t) b * m + i] = buf[cur][1][i]; }
		if (b + 1 == b1 && !out_rank && b1 != nb_total) break;       // (an expansion's last step: k_chain, fseq_kernels.hpp)
		uint32_t const nd = rank_digits(nkeys[b]);
		rank_digit_counts(rk, m, nd, L);
		for (uint32_t p = 0; p < nd; ++p)
		{
			stream_pass<false, KO ? 25 : 0, KO>(m, buf[cur][0], buf[cur][1], buf[cur ^ 1u][0], buf[cur ^ 1u][1], 0u, DigitRank{rk, 2u * p}, NoHook{}, L, &L.dh[4u * p], stage);
			cur ^= 1u;
		}
		// rows that start a new block key take the in-block divergence of that key
		for (uint32_t pos = tid; pos < m; pos += ST)
		{
			uint32_t const r = rk[buf[cur][0][pos]];
			uint32_t const rprev = pos ? rk[buf[cur][0][pos - 1u]] : PAD_KEY;
			if (r != rprev) buf[cur][1][pos] = kd[r];
		}
		__syncthreads();
	}
	if (out_state_a && b1 == nb_total)
		for (uint32_t i = tid; i < m; i += ST) { out_state_a[(size_t) nb_total * m + i] = buf[cur][0][i]; out_state_d[(size_t) nb_total * m + i] = buf[cur][1][i]; }
	if (out_rank)
		stream_emit_ranks(m, buf[cur][0], buf[cur][1], kstart, out_rank + (size_t) grp * m, out_keyd + (size_t) grp * m,
		                  out_nkeys + grp, L);
}

// ------------------------------------------------------------------------------------------------
// phase C, streamed.  Per block workspace (words): state 4m | keys 2m | V m | Vpos m | cnt m + B.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t columns_stream_ws_words(uint32_t m, uint32_t B) { return (size_t) 9 * m + B + 16; }

struct HistHook {
	uint32_t *cnt;
	__device__ __forceinline__ void operator()(uint32_t dold, uint32_t dnew) const
	{
		if (dold != dnew) { atomicSub(&cnt[dold], 1u); atomicAdd(&cnt[dnew], 1u); }
	}
};

// KS: key shift of the partition step's keyed scan (fseq_core.hpp): 19 when every value id is below 2^19
// (m + B < 524,288; a tile holds 4,096 < 2^13 rows), else 0 (the has-based scan)
template <int KS>
__global__ __launch_bounds__(ST) void k_columns_stream(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t npass, uint32_t bsh, uint32_t *ws, uint32_t staged,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint32_t Lseg, uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr,
	uint32_t snap_stride, uint32_t *__restrict__ ss_a, uint32_t *__restrict__ ss_d, uint32_t block0,
	uint32_t *done_host, uint32_t epoch, uint32_t ss_pack)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	uint8_t *sym = cv.take<uint8_t>((size_t) sym_bytes(m, bsh) + 16);
	StreamLds &L = *cv.take<StreamLds>(1);
	uint32_t *const stage = staged ? cv.take<uint32_t>(2 * (size_t) SCAP) : nullptr;
	uint32_t const tid = threadIdx.x;
	uint32_t const blk = blockIdx.x + block0;                  // phase C may be launched in several parts
	uint32_t *w = ws + (size_t) blk * columns_stream_ws_words(m, B);
	uint32_t *buf[2][2] = {{w, w + m}, {w + 2u * (size_t) m, w + 3u * (size_t) m}};
	uint32_t *keys[2] = {w + 4u * (size_t) m, w + 5u * (size_t) m};
	uint32_t *V = w + 6u * (size_t) m, *Vpos = w + 7u * (size_t) m, *cnt = w + 8u * (size_t) m;

	uint64_t const k0 = (uint64_t) blk * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const *sa = bstate_a + (size_t) blk * m, *sd = bstate_d + (size_t) blk * m;

	// ---- prologue: sort the boundary divergences (2-bit LSD passes), distinct values -> V, counts, ids
	for (uint32_t i = tid; i < m; i += ST) { buf[0][0][i] = sa[i]; keys[0][i] = sd[i]; }
	__syncthreads();
	uint32_t kc = 0;
	{
		uint32_t bits = 1;
		while (bits < 32u && (k0 >> bits) != 0) ++bits;          // divergences at the boundary are <= k0
		for (uint32_t sh = 0; sh < bits; sh += 2)
		{
			stream_pass<true>(m, keys[kc], nullptr, keys[kc ^ 1u], nullptr, 0u, DigitKey{sh}, NoHook{}, L, nullptr, stage);
			kc ^= 1u;
		}
	}
	uint32_t D0 = 0;
	{
		uint32_t const *sk = keys[kc];
		for (uint32_t base = 0; base < m; base += SCAP)
		{
			uint32_t kv[SE], nf = 0;
#pragma unroll
			for (int e = 0; e < SE; ++e)
			{
				uint32_t const pos = base + tid * SE + e;
				kv[e] = pos < m ? sk[pos] : 0u;
				nf += (pos < m && (pos == 0 || kv[e] != sk[pos - 1u])) ? 1u : 0u;
			}
			uint32_t total;
			uint32_t wv = D0 + block_excl_add<ST>(nf, L.red, &total);
#pragma unroll
			for (int e = 0; e < SE; ++e)
			{
				uint32_t const pos = base + tid * SE + e;
				if (pos < m && (pos == 0 || kv[e] != sk[pos - 1u])) { V[wv] = kv[e]; Vpos[wv] = pos; ++wv; }
			}
			D0 += total;
			__syncthreads();
		}
	}
	for (uint32_t i = tid; i < D0 + nb; i += ST)
		cnt[i] = i < D0 ? ((i + 1u < D0 ? Vpos[i + 1u] : m) - Vpos[i]) : 0u;
	for (uint32_t i = tid; i < m; i += ST)
	{
		uint32_t lo = 0, hi = D0;
		uint32_t const key = sd[i];
		while (lo < hi)
		{
			uint32_t const mid = (lo + hi) >> 1;
			if (V[mid] < key) lo = mid + 1; else hi = mid;
		}
		buf[0][1][i] = lo;
	}
	__syncthreads();
	bool const zero_present = (V[0] == 0u);
	uint32_t cur = 0;

	for (uint32_t j = 0; j < nb; ++j)
	{
		stage_column(sym, msa + (k0 + j) * ld, sym_bytes(m, bsh));
		__syncthreads();
		for (uint32_t pass = 0; pass < npass; ++pass)
		{
			uint32_t cnt4[4];
			column_digit_counts(sym, m, bsh, pass, cnt4, L.red);
			stream_pass<false, KS>(m, buf[cur][0], buf[cur][1], buf[cur ^ 1u][0], buf[cur ^ 1u][1], D0 + j, DigitColumn{sym, bsh, pass}, HistHook{cnt}, L, cnt4, stage);
			cur ^= 1u;
		}
		// ---- every snap_stride columns: drop the exact (a, d) for pass 2 (ids back to divergence values)
		if (ss_a && (k0 + j + 1) % snap_stride == 0)
		{
			size_t const q = (size_t) ((k0 + j + 1) / snap_stride), ob = q * m;
			uint8_t *sh = reinterpret_cast<uint8_t *>(ss_d) + q * ss_high_stride(m);
			for (uint32_t i = tid; i < m; i += ST)
			{
				uint32_t const vid = buf[cur][1][i];
				uint32_t const dv = vid < D0 ? V[vid] : (uint32_t) (k0 + (vid - D0) + 1u);
				if (ss_pack) { ss_a[ob + i] = buf[cur][0][i] | (dv << ss_pack); sh[i] = (uint8_t) (dv >> (32u - ss_pack)); }
				else { ss_a[ob + i] = buf[cur][0][i]; ss_d[ob + i] = dv; }
			}
		}
		// ---- emit the top of the histogram (same list format as k_columns); counters were updated
		// with device-scope atomics, read them past L1
		if (wave_id() == 0)
		{
			uint32_t const lane = lane_id();
			uint64_t const k = k0 + j;
			uint32_t const thr = (k + 2 > (uint64_t) Lseg) ? (uint32_t) (k + 2 - Lseg) : 0u;
			uint2 *out = ent + k * (size_t) stride;
			int32_t const top = (int32_t) (D0 + j);
			uint32_t cumN = 0, nent = 1, R = 0;
			for (int32_t base = top; base >= 0; base -= 64)
			{
				int32_t const i = base - (int32_t) lane;
				uint32_t const c = (i >= 0) ? __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
				uint32_t const v = (i < 0) ? 0u : (((uint32_t) i < D0) ? V[i] : (uint32_t) (k0 + ((uint32_t) i - D0) + 1u));
				bool const nz = c > 0;
				bool const rec = nz && v >= thr;
				uint32_t const inc = wave_incl_add(c);
				uint64_t const rmask = __ballot(rec);
				uint32_t const r_inc = rmask ? shfl_u32(inc, 63 - (int) __builtin_clzll(rmask)) : 0u;
				uint32_t const excN = cumN + (inc - r_inc) - c;
				bool const take = nz && !rec && excN <= X;
				uint64_t const omask = __ballot(nz && !rec);
				uint32_t const pos = nent + (uint32_t) __popcll(omask & ((1ull << lane) - 1ull));
				if (take) out[pos] = make_uint2(v, c);
				uint64_t const tmask = __ballot(take);
				nent += (uint32_t) __popcll(tmask);
				uint32_t const t_inc = tmask ? shfl_u32(inc, 63 - (int) __builtin_clzll(tmask)) : r_inc;
				cumN += t_inc - r_inc;
				R += r_inc;
				if (tmask != omask || cumN > X) break;
			}
			uint32_t const cum = R + cumN;
			if (lane == 0)
			{
				uint32_t const c0 = zero_present ? __hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
				out[0] = make_uint2((uint32_t) (k + 1), R);
				hdr[k] = make_uint4(nent, c0, cum == m ? 1u : 0u, cum);
			}
		}
		__syncthreads();
	}
	publish_block_done(done_host, blk, epoch);
}

} // namespace fseq
