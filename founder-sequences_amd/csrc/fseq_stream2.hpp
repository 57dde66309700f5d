// fseq_stream2.hpp -- phase C for orders that do not fit LDS (m > 11,264 rows), second form.
//
// Same algorithm and outputs as k_columns_stream (fseq_stream.hpp): the order of a block lives in a per-block
// HBM/L2 workspace and every column pass streams it through the workgroup tile by tile with a running TileCarry.
// What changed is the tile step, rebuilt around what profiles/r03_valu_rates.* and the round-2 counters showed (the
// old step issues ~490 vector instructions per thread and tile of 4 rows -- 122 per cell, ~70 % of the SIMDs' issue
// time at 4.2 cycles each -- among them 170 v_readlane / 70 v_writelane of spilled SGPRs, 145 64-bit address
// computations and flat_ (not global_) memory instructions):
//   * (a, d) of a row are ONE 8-byte pair in the workspace: one 16-byte load per two rows, one 8-byte LDS write
//     and one 8-byte store per row, one address each;
//   * T threads x E consecutive rows with E = 8 (fewer, longer threads: the scan and its second level over the
//     waves cost the same per thread whatever E is);
//   * loads and stores are buffer instructions on a descriptor of the workspace (uniform base + 32-bit offset:
//     no per-lane 64-bit address arithmetic; rows behind m of the last tile read as zero);
//   * the destination of a row is resolved in TILE-LOCAL coordinates: tile-local bucket start + rows of the
//     bucket in front (13 bits) and the prefix maximum (value ids < 2^KS) share a word, so one select tree per
//     row yields both, and the write-out takes the bucket of 64 consecutive output slots from scalar compares
//     (a wave's 64 slots straddle a bucket boundary in at most 3 of a tile's T * E / 64 stores);
//   * the next tile's rows are loaded into registers before the write-out of this one (the old form touched
//     them into L2 through an LDS-DMA sink); barriers inside a column wait for LDS only, two per tile;
//   * first_val (the id of the new column) is the largest id there is, so "first of its bucket" needs no select:
//     the prefix maximum of a bucket nobody has seen is first_val itself.
// The prologue of a block (sort of the boundary divergences -> value ids) stays the old code, as its own kernel.
#pragma once

#include "fseq_stream.hpp"

namespace fseq {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void const *p, uint32_t bytes)
{
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int) bytes, 0x00020000);
}

// barrier that waits for this wave's LDS traffic only (loads of the next tile stay in flight across it)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// key = (rows of the bucket up to and including me, inside the tile) << KS | running maximum: the count of a tile of
// `tile` rows must fit above the shift
constexpr int s2_key_shift(int tile)
{
	int bits = 0;
	while ((1ll << bits) <= (long long) tile) ++bits;
	return 32 - bits;
}

// -DFSEQ_S2_SKIP=bits: timing experiments (results are wrong): 1 no write-out stores, 2 no histogram atomics, 4 no list,
// 8 no tile loads (the first tile's rows are reused), 16 no LDS staging
#ifndef FSEQ_S2_SKIP
#define FSEQ_S2_SKIP 0
#endif
// -DFSEQ_S2_PW=0: the thread's own rows with four running maxima per row (the form before round 4's pairwise halves)
// -DFSEQ_S2_NARROW=0: the high bytes of the packed rows move with every block (by itself: only with blocks whose value ids need them)
#ifndef FSEQ_S2_NARROW
#define FSEQ_S2_NARROW 1
#endif
#ifndef FSEQ_S2_PW
#define FSEQ_S2_PW 1
#endif
// -DFSEQ_S2_STAMPS: per-wave cycle stamps of the tile loop (diagnostic build), printed for two blocks
#ifdef FSEQ_S2_STAMPS
#define S2_STAMP(i) do { long long const t_ = clock64(); s2_seg[i] += t_ - s2_last; s2_last = t_; } while (0)
#else
#define S2_STAMP(i) do { } while (0)
#endif

template <int T>
struct S2Lds {
	uint32_t cnt[T / WAVE][2];
	uint32_t val[T / WAVE][4];
	uint32_t red[4 * (T / WAVE) + 8];
	uint32_t sel[4][T];                     // per thread: the four buckets' {slot, prefix maximum} words, read back per row
};

// PACK: a row of the order is 5 bytes in the workspace instead of 8 -- a word a | d << abits (abits = bits of a row id)
// and the byte d >> (32 - abits) in a second array: value ids are < 2^KS <= 2^19 and row ids < m <= 2^KS, so 40 bits
// always hold both.  A CU moves ~10 bytes per cycle through its memory pipeline whatever the chip's HBM does, and the
// order crosses it twice per column: 16 bytes per row were ~75 % of a column's time on the C4 rows.
template <int T, int E, bool PACK>
__host__ __device__ inline size_t stream2_lds_bytes(uint32_t colbytes)
{
	return carve_bytes((size_t) colbytes + 16, 1) + carve_bytes(1, sizeof(S2Lds<T>))
	     + (PACK ? carve_bytes((size_t) T * E, 4) + carve_bytes((size_t) T * E, 1) : carve_bytes((size_t) T * E, 8));
}

// ------------------------------------------------------------------------------------------------
// Prologue of every block (the old kernel's, on its own): sorted distinct boundary divergences -> V, their
// counts -> cnt, the order as (a, value id) pairs -> pairs0, D0 -> w[9m + B].  ST threads.
// Workspace (words): pairs0 2m | pairs1 2m | keys 2m | V m | Vpos m | cnt m + B | D0
// pack_abits != 0 (PACK kernels): the first 4m words hold words0 m | words1 m | bytes0 m / 4 | .. | bytes1 m / 4 (at 3m) instead
// ------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(ST) void k_columns_stream2_prologue(
	uint32_t m, uint64_t n, uint32_t B, uint32_t *ws, uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d, uint32_t block0,
	uint32_t pack_abits, uint32_t *__restrict__ bs_w = nullptr, uint8_t *__restrict__ bs_h = nullptr, uint32_t const *__restrict__ blocklist = nullptr)
{
	// blocklist [r5]: workgroup i owns block blocklist[i] (the blocks the reduced phase C hands to the run on all rows)
	// bs_w / bs_h (packed rows only): a copy of the block's start state in id form -- what pass 2 replays from when no stride
	// state of the block lies in front of a boundary (k_columns_stream2<.., S2_SNAP>); block b at bs_w + b * m, bs_h + b * ss_high_stride(m)
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	StreamLds &L = *cv.take<StreamLds>(1);
	uint32_t *const stage = cv.take<uint32_t>(2 * (size_t) SCAP);
	uint32_t const tid = threadIdx.x;
	uint32_t const blk = blocklist ? blocklist[blockIdx.x] : blockIdx.x + block0;
	uint32_t *w = ws + (size_t) blk * columns_stream_ws_words(m, B);
	uint2 *pairs0 = reinterpret_cast<uint2 *>(w);
	uint32_t *keys[2] = {w + 4u * (size_t) m, w + 5u * (size_t) m};
	uint32_t *V = w + 6u * (size_t) m, *Vpos = w + 7u * (size_t) m, *cnt = w + 8u * (size_t) m;
	uint64_t const k0 = (uint64_t) blk * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const *sa = bstate_a + (size_t) blk * m, *sd = bstate_d + (size_t) blk * m;

	for (uint32_t i = tid; i < m; i += ST) keys[0][i] = sd[i];
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
		if (pack_abits)
		{
			uint32_t const pw = sa[i] | (lo << pack_abits);
			uint8_t const ph = (uint8_t) (lo >> (32u - pack_abits));
			w[i] = pw;
			reinterpret_cast<uint8_t *>(w + 2u * (size_t) m)[i] = ph;
			if (bs_w) { bs_w[(size_t) blk * m + i] = pw; bs_h[(size_t) blk * ss_high_stride(m) + i] = ph; }
		}
		else pairs0[i] = make_uint2(sa[i], lo);
	}
	if (tid == 0) w[9u * (size_t) m + B] = D0;
#ifdef FSEQ_S2_D0_PRINT
	if (tid == 0 && (blk & 127u) == 0u) printf("s2 prologue block %u: D0 %u (distinct divergence values in the start state), nb %u, pack_abits %u\n", blk, D0, nb, pack_abits);
#endif
}

// block-wide sum of four counters (T threads); every thread gets the totals.  Two barriers.
template <int T>
__device__ __forceinline__ void s2_block_sum4(uint32_t (&c)[4], uint32_t *red)
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
#pragma unroll
	for (int w = 0; w < T / WAVE; ++w)
	{
		uint4 const r = *reinterpret_cast<uint4 const *>(red + w * 4);
		c[0] += r.x; c[1] += r.y; c[2] += r.z; c[3] += r.w;
	}
}

// bucket sizes of a column pass straight off the staged packed column (as column_digit_counts, T threads)
template <int T>
__device__ __forceinline__ void s2_column_digit_counts(uint8_t const *sym, uint32_t m, uint32_t bsh, uint32_t pass, uint32_t (&cnt)[4], uint32_t *red)
{
#pragma unroll
	for (int x = 0; x < 4; ++x) cnt[x] = 0;
	uint32_t const spw = 4u << bsh, bits = 8u >> bsh;
	uint32_t const nwords = (sym_bytes(m, bsh) + 3u) / 4u;
	uint32_t const ones = bsh == 2 ? 0x55555555u : bsh == 1 ? 0x11111111u : 0x01010101u;
	for (uint32_t wi = threadIdx.x; wi < nwords; wi += T)
	{
		uint32_t const w = *reinterpret_cast<uint32_t const *>(sym + wi * 4u);
		uint32_t const r0 = wi * spw;
		uint32_t const nv = min(spw, m - r0);
		uint32_t const valid = nv == spw ? ones : (ones & ((1u << (nv * bits)) - 1u));
		uint32_t const lo = (w >> (2u * pass)) & valid, hi = (w >> (2u * pass + 1u)) & valid;
		cnt[0] += (uint32_t) __popc(valid & ~lo & ~hi);
		cnt[1] += (uint32_t) __popc(lo & ~hi);
		cnt[2] += (uint32_t) __popc(hi & ~lo);
		cnt[3] += (uint32_t) __popc(lo & hi);
	}
	s2_block_sum4<T>(cnt, red);
}

// ------------------------------------------------------------------------------------------------
// One tile: the stable 4-bucket partition of T * E consecutive rows of the order, in tile-local coordinates.
// In: d[e] (value ids < 2^KS), s[e] (s = 4: no row at this position -- last tile only; its d is 0).
// Out: lp[e] = slot of the row in the tile's output layout (the four bucket runs back to back), dnew[e];
// lofs[x] = first slot of bucket x, gsh[x] = (global position of the tile's run of bucket x) - lofs[x]: both uniform.
// first_val must be >= every value in the order (phase C: the id of the new column).  One LDS-only barrier.
// ------------------------------------------------------------------------------------------------
template <int T, int E, int KS, bool FULL>
__device__ __forceinline__ void s2_tile_step(
	uint32_t const (&d)[E], uint32_t const (&s)[E], uint32_t const first_val, S2Lds<T> &L, TileCarry &tc,
	uint32_t (&lp)[E], uint32_t (&dnew)[E], uint32_t (&lofs)[4], uint32_t (&gsh)[4]
#ifdef FSEQ_S2_STAMPS
	, long long (&s2_seg)[12], long long &s2_last
#endif
	)
{
	constexpr int NW = T / WAVE;
	constexpr uint32_t VMASK = (1u << KS) - 1u;
	static_assert(E <= 15, "local counts are nibble-packed");
	static_assert(NW >= 2 && NW <= 16, "wave totals fit one DPP row");
	static_assert((uint64_t) T * E < (1ull << (32 - KS)), "counts of one tile must fit the key");
	uint32_t const lane = lane_id();
	uint32_t const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

	// ---- the thread's own rows: running maximum per symbol since its last row, rank of a row among the thread's
	// rows of its symbol (nibble counters; position without a row: fifth nibble)
	uint32_t run[4] = {0u, 0u, 0u, 0u};
	uint32_t lcp = 0, pend = 0;
	uint32_t lidx[E];
#if FSEQ_S2_PW
	if constexpr (E == 8)
	{
		// [r4] The pairwise form in two halves of four rows (partition_step's PW / TWO, fseq_core.hpp): a row's value is the
		// maximum since the NEAREST earlier row of its half with its symbol -- six compares and selects per half over the chain
		// maxima -- instead of four running maxima per row (a max, a compare and two selects per row AND symbol: 128 of the
		// tile's ~520 vector instructions per thread); the tail maximum behind the last row of every symbol, which the scan
		// needs, goes through the thread's run slots (L.sel[x][thread]: conflict-free, rewritten by the resolve below).
		constexpr int E0 = 4;
		uint32_t const t_ = threadIdx.x;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const sh = s[e] * 4u;
			lidx[e] = (lcp >> sh) & 15u;
			pend |= ((FULL || s[e] < 4u) && lidx[e] == 0u) ? (1u << e) : 0u;
			lcp += 1u << sh;
		}
		uint32_t pre[E], suf[E];
		pre[0] = d[0];
#pragma unroll
		for (int e = 1; e < E; ++e) pre[e] = (e == E0) ? d[e] : max(pre[e - 1], d[e]);      // per half
		suf[E0 - 1] = 0u; suf[E - 1] = 0u;
#pragma unroll
		for (int j = E0 - 2; j >= 0; --j) suf[j] = max(suf[j + 1], d[j + 1]);
#pragma unroll
		for (int j = E - 2; j >= E0; --j) suf[j] = max(suf[j + 1], d[j + 1]);
		// tail maximum per symbol after the first half: the whole half if the symbol does not occur, else what follows its last row
#pragma unroll
		for (int x = 0; x < 4; ++x) L.sel[x][t_] = pre[E0 - 1];
#pragma unroll
		for (int e = 0; e < E0; ++e) if (FULL || s[e] < 4u) L.sel[s[e] & 3u][t_] = suf[e];      // (a position without a row, s = 4, writes nothing)
		uint32_t t0s[E - E0], t0x[4];
#pragma unroll
		for (int e = E0; e < E; ++e) t0s[e - E0] = L.sel[s[e] & 3u][t_];
#pragma unroll
		for (int x = 0; x < 4; ++x) t0x[x] = L.sel[x][t_];
		// ... and after the second: the first half's tail joined with the whole second half, or what follows the last row there
#pragma unroll
		for (int x = 0; x < 4; ++x) L.sel[x][t_] = max(t0x[x], pre[E - 1]);
#pragma unroll
		for (int e = E0; e < E; ++e) if (FULL || s[e] < 4u) L.sel[s[e] & 3u][t_] = suf[e];
		// the rows' own values: the nearest earlier row of the half with the same symbol, else the half's prefix (second half:
		// joined with the first half's tail of the symbol)
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			int const lo = (e >= E0) ? E0 : 0;
			uint32_t o = (e >= E0) ? max(t0s[e >= E0 ? e - E0 : 0], pre[e]) : pre[e];
			uint32_t ch[E];                                      // ch[j] = max d(j, e]
			if (e >= lo + 1)
			{
				ch[e - 1] = d[e];
#pragma unroll
				for (int j = e - 2; j >= lo; --j) ch[j] = max(ch[j + 1], d[j + 1]);
#pragma unroll
				for (int j = lo; j < e; ++j) o = (s[j] == s[e]) ? ch[j] : o;      // the nearest earlier one wins (ascending j)
			}
			dnew[e] = o;
			asm volatile("" : "+v"(dnew[e]));
		}
#pragma unroll
		for (int x = 0; x < 4; ++x) run[x] = L.sel[x][t_];
	}
	else
#endif
#pragma unroll
	for (int e = 0; e < E; ++e)
	{
		uint32_t const c = s[e];
		uint32_t const de = d[e];
		uint32_t o = 0;
#pragma unroll
		for (int x = 0; x < 4; ++x)
		{
			uint32_t const r = max(run[x], de);
			bool const is = (c == (uint32_t) x);
			o = is ? r : o;
			run[x] = is ? 0u : r;
		}
		dnew[e] = o;
		asm volatile("" : "+v"(dnew[e]));                       // (selected here, not behind the barrier: fseq_core.hpp)
		uint32_t const sh = c * 4u;
		lidx[e] = (lcp >> sh) & 15u;
		pend |= ((FULL || c < 4u) && lidx[e] == 0u) ? (1u << e) : 0u;
		lcp += 1u << sh;
	}

	// ---- bucket counts, inclusive over the lanes (two 16-bit counts per word), then the keys
	uint32_t ic[2];
	ic[0] = wave_incl_add((lcp & 15u) | (((lcp >> 4) & 15u) << 16));
	ic[1] = wave_incl_add(((lcp >> 8) & 15u) | (((lcp >> 12) & 15u) << 16));
	auto occ_key = [](uint32_t const (&c)[2], int x) -> uint32_t {
		return (x & 1) ? ((c[x >> 1] >> 16) << KS) : ((c[x >> 1] & 0xFFFFu) << KS);
	};
	uint32_t key[4];
#pragma unroll
	for (int x = 0; x < 4; ++x) key[x] = wave_incl_max(occ_key(ic, x) | run[x]);
	if (lane == 63)
	{
		L.cnt[wave][0] = ic[0]; L.cnt[wave][1] = ic[1];
#pragma unroll
		for (int x = 0; x < 4; ++x) L.val[wave][x] = key[x] & VMASK;
	}
	uint32_t ek[4];
#pragma unroll
	for (int x = 0; x < 4; ++x) ek[x] = dpp_mov<DPP_WAVE_SHR1, 0xF>(0u, key[x]);
	S2_STAMP(1);
	lds_barrier();
	S2_STAMP(2);

	// ---- second level over the NW wave totals (lanes 0 .. NW-1 of DPP row 0), every wave for itself
	uint32_t wc[2], wk[4];
#pragma unroll
	for (int i = 0; i < 2; ++i) wc[i] = lane < (uint32_t) NW ? L.cnt[lane][i] : 0u;
#pragma unroll
	for (int x = 0; x < 4; ++x) wk[x] = lane < (uint32_t) NW ? L.val[lane][x] : 0u;
#pragma unroll
	for (int i = 0; i < 2; ++i)
	{
		wc[i] += dpp_mov<DPP_ROW_SHR1, 0xF>(0u, wc[i]);
		if (NW > 2) wc[i] += dpp_mov<DPP_ROW_SHR2, 0xF>(0u, wc[i]);
		if (NW > 4) wc[i] += dpp_mov<DPP_ROW_SHR4, 0xF>(0u, wc[i]);
		if (NW > 8) wc[i] += dpp_mov<DPP_ROW_SHR8, 0xF>(0u, wc[i]);
	}
#pragma unroll
	for (int x = 0; x < 4; ++x)
	{
		uint32_t k = occ_key(wc, x) | wk[x];
		k = max(k, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, k));
		if (NW > 2) k = max(k, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, k));
		if (NW > 4) k = max(k, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, k));
		if (NW > 8) k = max(k, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, k));
		wk[x] = k;
	}
	uint32_t totc[2], totk[4], pc[2] = {0u, 0u}, pk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
	for (int i = 0; i < 2; ++i) totc[i] = readlane_u32(wc[i], NW - 1);
#pragma unroll
	for (int x = 0; x < 4; ++x) totk[x] = readlane_u32(wk[x], NW - 1);
	if (wave > 0)
	{
		int const src = (int) wave - 1;
#pragma unroll
		for (int i = 0; i < 2; ++i) pc[i] = readlane_u32(wc[i], src);
#pragma unroll
		for (int x = 0; x < 4; ++x) pk[x] = readlane_u32(wk[x], src);
	}

	// ---- per bucket: tile-local start, rows in front of this thread, prefix maximum -- one word
	uint32_t pv[4];
	{
		uint32_t acc = 0;
#pragma unroll
		for (int x = 0; x < 4; ++x)
		{
			uint32_t const tot = (totc[x >> 1] >> ((x & 1) * 16)) & 0xFFFFu;
			lofs[x] = acc;
			gsh[x] = tc.start[x] + tc.cnt[x] - acc;
			uint32_t const k = max(pk[x], ek[x] + occ_key(pc, x));
			// nobody in front of me inside the tile: the tiles to the left decide (and if they never saw the bucket
			// either, its first row takes first_val, which is >= every other value)
			uint32_t const left = ((tc.has >> x) & 1u) ? tc.val[x] : first_val;
			uint32_t const low = (k >> KS) ? 0u : left;
			pv[x] = max(k, (k & ~VMASK) | low) + (acc << KS);
			acc += tot;
			// the carry moves past this tile
			tc.cnt[x] += tot;
			tc.val[x] = (totk[x] >> KS) ? (totk[x] & VMASK) : max(tc.val[x], totk[x] & VMASK);
			tc.has |= (totk[x] >> KS) ? (1u << x) : 0u;
		}
	}
#pragma unroll
	for (int x = 0; x < 4; ++x) L.sel[x][threadIdx.x] = pv[x];
#pragma unroll
	for (int e = 0; e < E; ++e)
	{
		uint32_t const sel = L.sel[s[e] & 3u][threadIdx.x];     // (a position without a row, s = 4, reads entry 0; nothing uses it)
		lp[e] = (sel >> KS) + lidx[e];
		if ((pend >> e) & 1u) dnew[e] = max(sel & VMASK, dnew[e]);
	}
}

// ------------------------------------------------------------------------------------------------
// phase C, streamed, second form.  After k_columns_stream2_prologue on the same blocks.
// ------------------------------------------------------------------------------------------------
// (at least four waves per SIMD: two workgroups of 512 threads -- or three to four of 256 -- share a CU, and what one of
// them waits for at its barriers the others compute)
// MODE S2_SNAP: pass 2 with the same tile step -- update_pbwt_task::execute (update_pbwt_task.cc:13-35): the (a, d) at the merged
// boundaries.  Phase C leaves its stride states (and every block's start state, k_columns_stream2_prologue) in ID form,
// i.e. as the packed rows of its workspace, so a boundary is reached by replaying the columns from the nearest such state
// of its block in the block's id space (V and D0 are still in the block's workspace), with nothing but the partition: no
// histogram, no lists.  One workgroup per block that has boundaries; it takes the block's groups of boundaries (those that
// share a start state) one after the other in the block's own workspace.  BASELINE C4: pass 2 157 -> see DESIGN.md.
enum { S2_COLUMNS = 0, S2_SNAP = 1 };
constexpr uint32_t S2_SS_IDS = 0x80000000u;                   // ss_pack flag: the stride states hold value ids (packed rows), not divergences
struct S2SnapArgs {
	uint32_t const *wg_block;            // [grid] block of every workgroup
	uint2 const *wg_groups;              // [grid] {first group, groups}
	uint2 const *grp_tasks;              // [groups] {first boundary, boundaries}: the boundaries that share a start state, ascending
	uint64_t const *grp_src;             // [groups] bit 63 set: stride state q (the state at column q * snap_stride), else the block's start state
	uint64_t const *task_rb;             // [boundaries] column of every boundary
	uint32_t *snap_a, *snap_d;           // [boundaries][m]
	uint32_t const *bs_w;                // block start states in id form: block b at bs_w + b * m, bs_h + b * ss_high_stride(m)
	uint8_t const *bs_h;
};

template <int T, int E, bool PACK, int MODE = S2_COLUMNS>
__global__ __launch_bounds__(T, 4) void k_columns_stream2(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t npass, uint32_t bsh, uint32_t *ws,
	uint32_t Lseg, uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr,
	uint32_t snap_stride, uint32_t *__restrict__ ss_a, uint32_t *__restrict__ ss_d, uint32_t block0,
	uint32_t *done_host, uint32_t epoch, uint32_t ss_pack, S2SnapArgs const SN)
{
	static_assert(MODE == S2_COLUMNS || PACK, "pass 2 on the tile step replays packed rows");
	if (MODE == S2_COLUMNS) FSEQ_CLOCK_STAMP(blockIdx.x, 0);
	constexpr int KS = s2_key_shift(T * E);
	constexpr uint32_t TILE = (uint32_t) T * E;
	static_assert(E % 2 == 0, "a thread loads its rows as 16-byte pieces of two (a, d) pairs");
	static_assert(!PACK || E % 4 == 0, "packed rows: 16-byte pieces of four words, 4-byte pieces of four bytes");
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	uint8_t *sym = cv.take<uint8_t>((size_t) sym_bytes(m, bsh) + 16);
	S2Lds<T> &L = *cv.take<S2Lds<T>>(1);
	uint2 *const stage = PACK ? nullptr : cv.take<uint2>(TILE);
	uint32_t *const stage_w = PACK ? cv.take<uint32_t>(TILE) : nullptr;
	uint8_t *const stage_h = PACK ? cv.take<uint8_t>(TILE) : nullptr;
	uint32_t const tid = threadIdx.x;
	uint32_t const lane = lane_id();
	uint32_t const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	// (S2_COLUMNS with SN.wg_block set [r5]: the listed blocks -- what the reduced phase C hands to the run on all rows)
	uint32_t const blk = (MODE == S2_SNAP || SN.wg_block) ? SN.wg_block[blockIdx.x] : blockIdx.x + block0;
	uint32_t *w = ws + (size_t) blk * columns_stream_ws_words(m, B);
	uint2 *pairs[2] = {reinterpret_cast<uint2 *>(w), reinterpret_cast<uint2 *>(w + 2u * (size_t) m)};
	uint32_t *words[2] = {w, w + (size_t) m};                                 // PACK
	uint8_t *highs[2] = {reinterpret_cast<uint8_t *>(w + 2u * (size_t) m), reinterpret_cast<uint8_t *>(w + 3u * (size_t) m)};
	uint32_t abits = 1;
	while ((1u << abits) < m) ++abits;
	abits = __builtin_amdgcn_readfirstlane(abits);
	uint32_t const amask = (1u << abits) - 1u, hshift = 32u - abits;
	uint32_t const *V = w + 6u * (size_t) m;
	uint32_t *cnt = w + 8u * (size_t) m;
	uint32_t const D0 = __builtin_amdgcn_readfirstlane(w[9u * (size_t) m + B]);

	uint64_t const k0 = (uint64_t) blk * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	uint32_t const nb = (uint32_t) (kend - k0);
	// [r4] A block whose value ids all fit the word beside the row id (D0 boundary values + one id per column <= 2^hshift: every
	// block of BASELINE C4, D0 13,000 - 20,000 at 15 bits) never sets a high byte: they are neither loaded, nor staged, nor
	// stored, nor copied into the stride states -- 8 instead of 10 bytes per row and column through HBM.  (Uniform per block, and
	// the same in phase C and in pass 2: both read D0 from the block's workspace.)
	bool const narrow = PACK && FSEQ_S2_NARROW && D0 + nb <= (1u << hshift);
	bool const zero_present = (V[0] == 0u);
	uint32_t const colbytes = sym_bytes(m, bsh);
	uint32_t cur = 0;
#ifdef FSEQ_S2_STAMPS
	long long s2_seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s2_last = clock64();
#endif

	// phase C: the block's columns once.  Pass 2: the block's groups of boundaries, each from its start state to its last boundary
	uint2 const wg_groups = MODE == S2_SNAP ? SN.wg_groups[blockIdx.x] : make_uint2(0u, 1u);
	for (uint32_t gi = 0; gi < wg_groups.y; ++gi)
	{
	uint32_t j_begin = 0, j_end = nb;
	uint32_t t_first = 0, t_count = 0, t_next = 0;
	if (MODE == S2_SNAP)
	{
		uint2 const gt = SN.grp_tasks[wg_groups.x + gi];
		t_first = gt.x; t_count = gt.y;
		uint64_t const src = SN.grp_src[wg_groups.x + gi];
		uint64_t const sidx = src & ~(1ull << 63);
		uint32_t const *sw;
		uint8_t const *sh;
		if (src >> 63) { j_begin = (uint32_t) (sidx * snap_stride - k0); sw = ss_a + sidx * (size_t) m; sh = reinterpret_cast<uint8_t const *>(ss_d) + sidx * ss_high_stride(m); }
		else { j_begin = 0; sw = SN.bs_w + (size_t) blk * m; sh = SN.bs_h + (size_t) blk * ss_high_stride(m); }
		j_end = (uint32_t) (SN.task_rb[t_first + t_count - 1u] - k0);
		__syncthreads();                                          // (the group before may still be read: its last snapshot)
		cur = 0;
		if ((m & 3u) == 0u)
		{
			uint4 const *src = reinterpret_cast<uint4 const *>(sw);
			uint4 *dst = reinterpret_cast<uint4 *>(words[0]);
#pragma unroll 4
			for (uint32_t i = tid; i < m / 4u; i += T) dst[i] = src[i];
		}
		else
			for (uint32_t i = tid; i < m; i += T) words[0][i] = sw[i];
		if (!narrow)
		{
			uint32_t const *hs = reinterpret_cast<uint32_t const *>(sh);
			uint32_t *hd = reinterpret_cast<uint32_t *>(highs[0]);
#pragma unroll 4
			for (uint32_t i = tid; i < (m + 3u) / 4u; i += T) hd[i] = hs[i];
		}
		__syncthreads();
	}
	// a boundary at column k (the state in front of it): ids back to divergences, unpacked
	auto snapshot_if_requested = [&](uint64_t k) {
		if (MODE == S2_SNAP && t_next < t_count && SN.task_rb[t_first + t_next] == k)
		{
			size_t const ob = (size_t) (t_first + t_next) * m;
			uint32_t const *PW = words[cur];
			uint8_t const *PH = highs[cur];
			for (uint32_t i = tid; i < m; i += T)
			{
				uint32_t const pw = PW[i];
				uint32_t const id = (pw >> abits) | (narrow ? 0u : ((uint32_t) PH[i] << hshift));
				SN.snap_a[ob + i] = pw & amask;
				SN.snap_d[ob + i] = id < D0 ? V[id] : (uint32_t) (k0 + (id - D0) + 1u);
			}
			++t_next;
		}
	};
	snapshot_if_requested(k0 + j_begin);
	for (uint32_t j = j_begin; j < j_end; ++j)
	{
		{
			uint8_t const *col = msa + (k0 + j) * ld;
			for (uint32_t i = tid * 16u; i < colbytes; i += T * 16u)
				*reinterpret_cast<uint4 *>(sym + i) = *reinterpret_cast<uint4 const *>(col + i);
		}
		__syncthreads();
		S2_STAMP(11);
		for (uint32_t pass = 0; pass < npass; ++pass)
		{
			uint32_t cnt4[4];
			s2_column_digit_counts<T>(sym, m, bsh, pass, cnt4, L.red);
#pragma unroll
			for (int x = 0; x < 4; ++x) cnt4[x] = __builtin_amdgcn_readfirstlane(cnt4[x]);      // (sums read from LDS: uniform, but not to the compiler)
			TileCarry tc;
			{
				uint32_t acc = 0;
#pragma unroll
				for (int x = 0; x < 4; ++x) { tc.cnt[x] = 0; tc.val[x] = 0; tc.start[x] = acc; acc += cnt4[x]; }
				tc.has = 0;
			}
			__amdgpu_buffer_rsrc_t const rs = PACK ? make_rsrc(words[cur], m * 4u) : make_rsrc(pairs[cur], m * 8u),
			                             rd = PACK ? make_rsrc(words[cur ^ 1u], m * 4u) : make_rsrc(pairs[cur ^ 1u], m * 8u);
			// (bytes: the range rounded up to whole words -- a word that straddles the end of a descriptor's range reads as zero)
			__amdgpu_buffer_rsrc_t const rsh = make_rsrc(highs[cur], (m + 3u) & ~3u), rdh = make_rsrc(highs[cur ^ 1u], (m + 3u) & ~3u);      // PACK
			uint32_t const first_val = D0 + j;
			uint32_t const toff = tid * (uint32_t) (E * (PACK ? 4 : 8));          // byte offset of the thread's rows inside a tile
			// rows of the tile at `base` (rows behind m read as zero)
			uint32_t an[E], dn[E];
			auto load_tile = [&](uint32_t base) {
				if constexpr (PACK)
				{
					uint32_t hw[E / 4];
#pragma unroll
					for (int q = 0; q < E / 4; ++q)
					{
						u32x4 const v = __builtin_amdgcn_raw_buffer_load_b128(rs, toff + 16u * q, base * 4u, 0);
						an[4 * q] = v.x; an[4 * q + 1] = v.y; an[4 * q + 2] = v.z; an[4 * q + 3] = v.w;
						hw[q] = 0u;
						if (!narrow) hw[q] = __builtin_amdgcn_raw_buffer_load_b32(rsh, tid * (uint32_t) E + 4u * q, base, 0);
					}
					// (unpacked where they arrive: dn = value id, an = row id)
					if (narrow)
					{
#pragma unroll
						for (int e = 0; e < E; ++e) { dn[e] = an[e] >> abits; an[e] &= amask; }
					}
					else
					{
#pragma unroll
						for (int e = 0; e < E; ++e)
						{
							uint32_t const hb = (hw[e / 4] >> (8 * (e % 4))) & 255u;
							dn[e] = (an[e] >> abits) | (hb << hshift);
							an[e] &= amask;
						}
					}
				}
				else
				{
#pragma unroll
					for (int q = 0; q < E / 2; ++q)
					{
						u32x4 const v = __builtin_amdgcn_raw_buffer_load_b128(rs, toff + 16u * q, base * 8u, 0);
						an[2 * q] = v.x; dn[2 * q] = v.y; an[2 * q + 1] = v.z; dn[2 * q + 1] = v.w;
					}
				}
			};
			auto tile = [&](uint32_t base, auto full_tag) {
				constexpr bool FULL = decltype(full_tag)::value;
				uint32_t a[E], d[E], s[E], lp[E], dnew[E], lofs[4], gsh[4];
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					bool const in = FULL || base + tid * E + e < m;
					a[e] = an[e]; d[e] = in ? dn[e] : 0u;                  // (a position without a row must carry d = 0)
					s[e] = in ? sym_digit(sym, a[e], bsh, pass) : 4u;
				}
				// the next tile's rows on their way while this one is partitioned, staged and written out (issued last thing
				// before the write-out, they were 5 of 21 ms of the C4 prefix: all waves of the workgroup wait for them together)
				if (FULL && base + TILE < m && !(FSEQ_S2_SKIP & 8)) load_tile(base + TILE);
				S2_STAMP(0);
#ifdef FSEQ_S2_STAMPS
				s2_tile_step<T, E, KS, FULL>(d, s, first_val, L, tc, lp, dnew, lofs, gsh, s2_seg, s2_last);
#else
				s2_tile_step<T, E, KS, FULL>(d, s, first_val, L, tc, lp, dnew, lofs, gsh);
#endif
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					if (FULL || base + tid * E + e < m)
					{
						if constexpr (PACK)
						{
							stage_w[lp[e]] = a[e] | (dnew[e] << abits);
							if (!narrow) stage_h[lp[e]] = (uint8_t) (dnew[e] >> hshift);
						}
						else if (!(FSEQ_S2_SKIP & 16)) stage[lp[e]] = make_uint2(a[e], dnew[e]);
						if (MODE == S2_COLUMNS && !(FSEQ_S2_SKIP & 2) && d[e] != dnew[e])
						{
							(void) __hip_atomic_fetch_add(&cnt[d[e]], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
							(void) __hip_atomic_fetch_add(&cnt[dnew[e]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
						}
					}
				}
				S2_STAMP(3);
				lds_barrier();
				S2_STAMP(4);
				uint32_t const tile_n = FULL ? TILE : m - base;
				auto ge = [](uint32_t j, uint32_t l) -> uint32_t { return (l - 1u - j) >> 31; };      // j >= l as 0 / 1 (both < 2^31)
				if constexpr (PACK)
				{
					// words: groups of 256 output slots (16 bytes per lane); bytes: groups of 64 (one byte per lane -- a run of a
					// bucket starts at any byte of the array).  gb[x] = first group entirely behind the start of bucket x; a group
					// that holds a bucket start strictly inside it takes the per-slot path.
					uint32_t const gb1 = (lofs[1] + 255u) >> 8, gb2 = (lofs[2] + 255u) >> 8, gb3 = (lofs[3] + 255u) >> 8;
					uint32_t const sg1 = (lofs[1] & 255u) ? (lofs[1] >> 8) : 0xFFFFFFFFu, sg2 = (lofs[2] & 255u) ? (lofs[2] >> 8) : 0xFFFFFFFFu,
					               sg3 = (lofs[3] & 255u) ? (lofs[3] >> 8) : 0xFFFFFFFFu;
#pragma unroll
					for (int e = 0; e < E / 4; ++e)
					{
						uint32_t const g = (uint32_t) e * (T / WAVE) + wave;
						uint32_t const j0 = g * 256u;
						if (!FULL && j0 >= tile_n) break;
						uint4 const v = *reinterpret_cast<uint4 const *>(stage_w + j0 + 4u * lane);
						// the high bytes of the same 256 slots: four per lane, one 4-byte store at whatever byte the run stands at (the
						// memory pipeline takes unaligned dwords; one byte per lane and store instruction was 8 of a thread's 10 stores
						// per tile, each with its own LDS read and scalar bucket selection)
						uint32_t hv = 0u;
						if (!narrow) hv = *reinterpret_cast<uint32_t const *>(stage_h + j0 + 4u * lane);
						if (FULL && g != sg1 && g != sg2 && g != sg3)
						{
							uint32_t sh = gsh[0];
							sh = ge(g, gb1) ? gsh[1] : sh;
							sh = ge(g, gb2) ? gsh[2] : sh;
							sh = ge(g, gb3) ? gsh[3] : sh;
							u32x4 const vv = {v.x, v.y, v.z, v.w};
							__builtin_amdgcn_raw_buffer_store_b128(vv, rd, lane * 16u, (j0 + sh) * 4u, 0);
							if (!narrow) __builtin_amdgcn_raw_buffer_store_b32(hv, rdh, lane * 4u, j0 + sh, 0);
						}
						else
						{
							uint32_t const vq[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
							for (int h = 0; h < 4; ++h)
							{
								uint32_t const jj = j0 + 4u * lane + (uint32_t) h;
								uint32_t sh = gsh[0];
								sh = jj >= lofs[1] ? gsh[1] : sh;
								sh = jj >= lofs[2] ? gsh[2] : sh;
								sh = jj >= lofs[3] ? gsh[3] : sh;
								if (FULL || jj < tile_n)
								{
									__builtin_amdgcn_raw_buffer_store_b32(vq[h], rd, (jj + sh) * 4u, 0u, 0);
									if (!narrow) __builtin_amdgcn_raw_buffer_store_b8((uint8_t) (hv >> (8 * h)), rdh, jj + sh, 0u, 0);
								}
							}
						}
					}
				}
				else
				{
				// Output slots go out in groups of 128 (one 16-byte store per lane: the memory pipeline of a CU, ~10 bytes per
				// cycle, takes wide stores at twice the rate of 8-byte ones).  gb[x] = first group that lies entirely behind the
				// start of bucket x; a group that holds a bucket start strictly inside it takes the per-slot path.
				uint32_t const gb1 = (lofs[1] + 127u) >> 7, gb2 = (lofs[2] + 127u) >> 7, gb3 = (lofs[3] + 127u) >> 7;
				uint32_t const sg1 = (lofs[1] & 127u) ? (lofs[1] >> 7) : 0xFFFFFFFFu, sg2 = (lofs[2] & 127u) ? (lofs[2] >> 7) : 0xFFFFFFFFu,
				               sg3 = (lofs[3] & 127u) ? (lofs[3] >> 7) : 0xFFFFFFFFu;
#pragma unroll
				for (int e = 0; e < E / 2; ++e)
				{
					uint32_t const g = (uint32_t) e * (T / WAVE) + wave;      // this store's group of 128 output slots: uniform
					uint32_t const j0 = g * 128u;
					if (!FULL && j0 >= tile_n) break;
					uint4 const v = *reinterpret_cast<uint4 const *>(stage + j0 + 2u * lane);
					if (FULL && g != sg1 && g != sg2 && g != sg3)
					{
						uint32_t sh = gsh[0];
						sh = ge(g, gb1) ? gsh[1] : sh;
						sh = ge(g, gb2) ? gsh[2] : sh;
						sh = ge(g, gb3) ? gsh[3] : sh;
						u32x4 const vv = {v.x, v.y, v.z, v.w};
						if (!(FSEQ_S2_SKIP & 1))
						__builtin_amdgcn_raw_buffer_store_b128(vv, rd, lane * 16u, (j0 + sh) * 8u, 0);
					}
					else
					{
#pragma unroll
						for (int h = 0; h < 2; ++h)
						{
							uint32_t const jj = j0 + 2u * lane + (uint32_t) h;
							uint32_t sh = gsh[0];
							sh = jj >= lofs[1] ? gsh[1] : sh;
							sh = jj >= lofs[2] ? gsh[2] : sh;
							sh = jj >= lofs[3] ? gsh[3] : sh;
							u32x2 const vv = {h ? v.z : v.x, h ? v.w : v.y};
							if (FULL || jj < tile_n)
								__builtin_amdgcn_raw_buffer_store_b64(vv, rd, (jj + sh) * 8u, 0u, 0);
						}
					}
				}
				}
				S2_STAMP(5);
				// (no barrier here: the next tile's stage writes come behind its own barrier, which every wave reaches only
				// after its reads above; the scan scratch is rewritten only by waves that have passed the barrier above)
			};
			uint32_t base = 0;
			load_tile(0);
			S2_STAMP(6);
			for (; base + TILE <= m; base += TILE) tile(base, std::true_type{});
			if (base < m) tile(base, std::false_type{});
			cur ^= 1u;
			__syncthreads();
			S2_STAMP(7);
		}
		uint2 const *P = pairs[cur];
		uint32_t const *PW = words[cur];
		uint8_t const *PH = highs[cur];
		// ---- every snap_stride columns: drop the exact (a, d) for pass 2 (ids back to divergence values)
		if (MODE == S2_COLUMNS && ss_a && (k0 + j + 1) % snap_stride == 0 && (ss_pack & S2_SS_IDS))
		{
			// (id form: the packed rows as they are -- pass 2 replays them in this block's id space)
			size_t const q = (size_t) ((k0 + j + 1) / snap_stride), ob = q * m;
			uint8_t *sh = reinterpret_cast<uint8_t *>(ss_d) + q * ss_high_stride(m);
			if constexpr (PACK)
			{
				// (16 bytes per thread and step, four steps in flight: one word per step was a chain of ~200 dependent round trips,
				// 4.6 % of a column on the C4 rows)
				if ((m & 3u) == 0u)
				{
					uint4 const *src = reinterpret_cast<uint4 const *>(PW);
					uint4 *dst = reinterpret_cast<uint4 *>(ss_a + ob);
#pragma unroll 4
					for (uint32_t i = tid; i < m / 4u; i += T) dst[i] = src[i];
				}
				else
					for (uint32_t i = tid; i < m; i += T) ss_a[ob + i] = PW[i];
				if (!narrow)
				{
					// (whole words of four high bytes: both arrays are 4-byte aligned and padded past m)
					uint32_t const *hs = reinterpret_cast<uint32_t const *>(PH);
					uint32_t *hd = reinterpret_cast<uint32_t *>(sh);
#pragma unroll 4
					for (uint32_t i = tid; i < (m + 3u) / 4u; i += T) hd[i] = hs[i];
				}
			}
		}
		else if (MODE == S2_COLUMNS && ss_a && (k0 + j + 1) % snap_stride == 0)
		{
			size_t const q = (size_t) ((k0 + j + 1) / snap_stride), ob = q * m;
			uint8_t *sh = reinterpret_cast<uint8_t *>(ss_d) + q * ss_high_stride(m);
			for (uint32_t i = tid; i < m; i += T)
			{
				uint2 p;
				if constexpr (PACK) { uint32_t const pw = PW[i]; p = make_uint2(pw & amask, (pw >> abits) | (narrow ? 0u : ((uint32_t) PH[i] << hshift))); }
				else p = P[i];
				uint32_t const dv = p.y < D0 ? V[p.y] : (uint32_t) (k0 + (p.y - D0) + 1u);
				if (ss_pack) { ss_a[ob + i] = p.x | (dv << ss_pack); sh[i] = (uint8_t) (dv >> (32u - ss_pack)); }
				else { ss_a[ob + i] = p.x; ss_d[ob + i] = dv; }
			}
		}
		S2_STAMP(8);
		// ---- emit the top of the histogram (same list format as k_columns); counters were updated
		// with device-scope atomics, read them past L1
		if (MODE == S2_COLUMNS && wave == 0 && !(FSEQ_S2_SKIP & 4))
		{
			// LQ ids per lane and step (the id space is sparse -- one id per column and boundary value, most of them with
			// count 0 by now -- and every step is a round trip to L2: with one id per lane the list was ~20 % of a column)
			constexpr int LQ = 8;
			uint64_t const k = k0 + j;
			uint32_t const thr = (k + 2 > (uint64_t) Lseg) ? (uint32_t) (k + 2 - Lseg) : 0u;
			uint2 *out = ent + k * (size_t) stride;
			int32_t const top = (int32_t) (D0 + j);
			uint32_t cumN = 0, nent = 1, R = 0;       // cumN: count of the values below thr taken so far
			for (int32_t base = top; base >= 0; base -= 64 * LQ)
			{
				// lane l holds the ids base - LQ l - q: descending ids = descending values, lane-major
				uint32_t c[LQ], v[LQ];
				uint32_t lane_c = 0, lane_o = 0, lane_rc = 0, candm = 0;
#pragma unroll
				for (int q = 0; q < LQ; ++q)
				{
					int32_t const i = base - LQ * (int32_t) lane - q;
					c[q] = (i >= 0) ? __hip_atomic_load(&cnt[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
					v[q] = (i < 0) ? 0u : (((uint32_t) i < D0) ? V[i] : (uint32_t) (k0 + ((uint32_t) i - D0) + 1u));
				}
#pragma unroll
				for (int q = 0; q < LQ; ++q)
				{
					bool const nz = c[q] > 0;
					bool const rec = nz && v[q] >= thr;       // the values >= thr are a prefix of the non-zero entries
					candm |= (nz && !rec) ? (1u << q) : 0u;
					lane_c += c[q];
					lane_o += (nz && !rec) ? 1u : 0u;
					lane_rc += rec ? c[q] : 0u;
				}
				uint32_t const inc_c = wave_incl_add(lane_c), inc_o = wave_incl_add(lane_o);
				uint32_t const r_tot = readlane_u32(wave_incl_add(lane_rc), 63);
				uint32_t const tot_o = readlane_u32(inc_o, 63);
				uint32_t run_c = inc_c - lane_c;          // counts in front of this lane's entries
				uint32_t run_o = inc_o - lane_o;          // candidates in front of them
				uint32_t lastP = 0, ntk = 0;
#pragma unroll
				for (int q = 0; q < LQ; ++q)
				{
					// a candidate is taken while the below-thr counts in front of it do not exceed X
					uint32_t const excN = cumN + run_c - r_tot;
					bool const cand = (candm >> q) & 1u;
					bool const tk = cand && excN <= X;
					if (tk) { out[nent + run_o] = make_uint2(v[q], c[q]); lastP = excN + c[q]; ++ntk; }
					run_o += cand ? 1u : 0u;
					run_c += c[q];
				}
				uint32_t const taken = readlane_u32(wave_incl_add(ntk), 63);
				// the taken entries are a prefix of the candidates: the largest inclusive count among them is the new cumN
				uint32_t const mx = readlane_u32(wave_incl_max(lastP), 63);
				nent += taken;
				if (taken) cumN = mx;
				R += r_tot;
				if (taken != tot_o || cumN > X) break;
			}
			uint32_t const cum = R + cumN;
			if (lane == 0)
			{
				uint32_t const c0 = zero_present ? __hip_atomic_load(&cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
				out[0] = make_uint2((uint32_t) (k + 1), R);
				hdr[k] = make_uint4(nent, c0, cum == m ? 1u : 0u, cum);
			}
		}
		S2_STAMP(9);
		snapshot_if_requested(k0 + j + 1u);
		__syncthreads();
		S2_STAMP(10);
	}
	}
#ifdef FSEQ_S2_STAMPS
	if (lane == 0 && (blockIdx.x == 7 || blockIdx.x == 200) && nb)
		printf("s2 stamps block %u wave %2u, cycles per column: loads+symbols %lld | own rows+scan %lld | barrier A %lld | resolve+stage+hook %lld | barrier B %lld | "
		       "write-out %lld | counts + first load %lld | pass tail barrier %lld | snapshot %lld | list %lld | column end barrier %lld | column staged %lld\n", blockIdx.x, wave,
		       s2_seg[0] / nb, s2_seg[1] / nb, s2_seg[2] / nb, s2_seg[3] / nb, s2_seg[4] / nb, s2_seg[5] / nb, s2_seg[6] / nb, s2_seg[7] / nb, s2_seg[8] / nb, s2_seg[9] / nb, s2_seg[10] / nb, s2_seg[11] / nb);
#endif
	if (MODE == S2_COLUMNS) FSEQ_CLOCK_STAMP(blockIdx.x, 1);
	if (MODE == S2_COLUMNS) publish_block_done(done_host, blk, epoch);
}

} // namespace fseq
