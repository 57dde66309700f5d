// fseq_core.hpp -- device building blocks for the pBWT column update on gfx950 (wave64).
//
// The unit of work is one *stable sigma-bucket partition step* of an LDS-resident order
// (a[], d[]) of up to T*E rows, owned by one workgroup of T threads (E consecutive positions
// per thread):
//     dst(i)  = start[sym_i] + #{j < i : sym_j == sym_i}            (stable counting sort)
//     d'(dst) = first in its bucket ? first_val : max d(p, i]        (p = previous row with sym_i)
// This is libbio::pbwt::pbwt_context::process for one column (SURVEY.md Appendix A step 2;
// instantiation include/founder_sequences/founder_sequences.hh:56-65), and it is also the digit
// pass of the block-rank sort in phase B.  Both prefix problems are solved with ONE block-wide
// scan of a per-thread summary {counts[sigma], running-max[sigma], seen-mask}:
// wave-level Hillis-Steele over 64 lanes, then a fold of the wave totals through LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fseq {

constexpr int WAVE = 64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ uint32_t shfl_up_u32(uint32_t v, int delta) { return (uint32_t) __shfl_up((int) v, delta, WAVE); }
__device__ __forceinline__ uint32_t shfl_dn_u32(uint32_t v, int delta) { return (uint32_t) __shfl_down((int) v, delta, WAVE); }
__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, int src) { return (uint32_t) __shfl((int) v, src, WAVE); }

// ---- DPP cross-lane moves (VALU, no LDS traffic).  A lane whose source is outside its row, or
// whose row is masked out, keeps `old` -- pass the operator's identity there.
// wave64 inclusive scan = row_shr 1,2,4,8 inside the 16-lane rows, then row_bcast:15 into rows
// 1 and 3, then row_bcast:31 into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t src)
{
	return (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) src, CTRL, ROW_MASK, 0xF, false);
}

constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int lane) { return (uint32_t) __builtin_amdgcn_readlane((int) v, lane); }

// inclusive add-scan over the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_add(uint32_t v)
{
	v += dpp_mov<DPP_ROW_SHR1, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_SHR2, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_SHR4, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_SHR8, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, v);
	v += dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, v);
	return v;
}

// inclusive max-scan over the 64 lanes of a wave (identity 0)
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v)
{
	v = max(v, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, v));
	v = max(v, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, v));
	v = max(v, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, v));
	v = max(v, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, v));
	v = max(v, dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, v));
	v = max(v, dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, v));
	return v;
}

// minimum over the 64 lanes of a wave (returned in every lane)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
	v = min(v, dpp_mov<DPP_ROW_SHR1, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_SHR2, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_SHR4, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_SHR8, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_BCAST15, 0xA>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_BCAST31, 0xC>(0xFFFFFFFFu, v));
	return readlane_u32(v, 63);
}

// Block-wide exclusive add-scan of one uint32 per thread. scratch: T/64 + 1 words of LDS.
// Contains two __syncthreads().  Returns the exclusive prefix; *total = sum over the block.
template <int T>
__device__ __forceinline__ uint32_t block_excl_add(uint32_t v, uint32_t *scratch, uint32_t *total)
{
	constexpr int NW = T / WAVE;
	uint32_t const inc = wave_incl_add(v);
	if (NW == 1)
	{
		*total = readlane_u32(inc, 63);
		return inc - v;
	}
	__syncthreads();                       // protect scratch against the previous use
	if (lane_id() == 63) scratch[wave_id()] = inc;
	__syncthreads();
	uint32_t pre = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < NW; ++w)
	{
		uint32_t const x = scratch[w];
		if ((uint32_t) w < wave_id()) pre += x;
		tot += x;
	}
	*total = tot;
	return pre + inc - v;
}

// ---------------------------------------------------------------------------------------------
// Per-thread summary for the partition scan.
// cnt: two 16-bit counts per word (m <= 65535 for the LDS-resident kernels).
// val[x]: max d since the last row with symbol x inside the summarised range (whole range if none).
// has:  bit x set iff symbol x occurs in the range.
// combine(L, R) (L to the left of R): cnt = L.cnt + R.cnt ; val[x] = R.has[x] ? R.val[x] : max(L.val[x], R.val[x]).
// ---------------------------------------------------------------------------------------------
template <int SIGMA>
struct Summary {
	static constexpr int NC = SIGMA / 2;
	uint32_t cnt[NC];
	uint32_t val[SIGMA];
	uint32_t has;

	__device__ __forceinline__ void clear()
	{
#pragma unroll
		for (int i = 0; i < NC; ++i) cnt[i] = 0;
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) val[x] = 0;
		has = 0;
	}

	// *this = L (+) *this
	__device__ __forceinline__ void prepend(Summary const &L)
	{
#pragma unroll
		for (int i = 0; i < NC; ++i) cnt[i] += L.cnt[i];
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
		{
			uint32_t const mx = max(L.val[x], val[x]);
			val[x] = ((has >> x) & 1u) ? val[x] : mx;
		}
		has |= L.has;
	}

	// identity where the DPP source lane is invalid / the row is masked
	template <int CTRL, int ROW_MASK>
	__device__ __forceinline__ Summary dpp_shifted() const
	{
		Summary o;
#pragma unroll
		for (int i = 0; i < NC; ++i) o.cnt[i] = dpp_mov<CTRL, ROW_MASK>(0u, cnt[i]);
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) o.val[x] = dpp_mov<CTRL, ROW_MASK>(0u, val[x]);
		o.has = dpp_mov<CTRL, ROW_MASK>(0u, has);
		return o;
	}

	__device__ __forceinline__ Summary from_lane_uniform(int src) const
	{
		Summary o;
#pragma unroll
		for (int i = 0; i < NC; ++i) o.cnt[i] = readlane_u32(cnt[i], src);
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) o.val[x] = readlane_u32(val[x], src);
		o.has = readlane_u32(has, src);
		return o;
	}

	__device__ __forceinline__ Summary shifted_up(int delta) const
	{
		Summary o;
#pragma unroll
		for (int i = 0; i < NC; ++i) o.cnt[i] = shfl_up_u32(cnt[i], delta);
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) o.val[x] = shfl_up_u32(val[x], delta);
		o.has = shfl_up_u32(has, delta);
		return o;
	}

	__device__ __forceinline__ Summary from_lane(int src) const
	{
		Summary o;
#pragma unroll
		for (int i = 0; i < NC; ++i) o.cnt[i] = shfl_u32(cnt[i], src);
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) o.val[x] = shfl_u32(val[x], src);
		o.has = shfl_u32(has, src);
		return o;
	}

	__device__ __forceinline__ uint32_t count(int x) const { return (cnt[x >> 1] >> ((x & 1) * 16)) & 0xFFFFu; }
};

// largest key shift a step of T * E rows allows: its bucket counts must fit the 32 - KS bits above the value
constexpr int scan_shift_for(int T, int E)
{
	int bits = 0;
	while ((1ll << bits) <= (long long) T * E) ++bits;
	return 32 - bits > 25 ? 25 : 32 - bits;
}

// One dword per lane from `g` into LDS at lds_addr + 4 * lane (LDS-DMA: no destination register, so nothing waits for
// it and nothing can be clobbered by it): used to pull the NEXT tile's lines into L2 while this tile is computed.
__device__ __forceinline__ void stream_touch(void const *g, uint32_t lds_addr)
{
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// Sixteen bytes per lane from `g` into LDS at lds_addr + 16 * lane (LDS-DMA; lds_addr wave-uniform).  The compiler does not
// see the load: the caller waits for it (s_waitcnt vmcnt(0)) before the barrier in front of the first read of those bytes.
__device__ __forceinline__ void lds_dma16(void const *g, uint32_t lds_addr)
{
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// a barrier that waits for this wave's LDS operations only (__syncthreads() also drains its global stores and whatever LDS-DMA
// it has in flight)
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// N consecutive 32-bit LDS words at a 4N-byte aligned address, as one or two wide reads (ds_read_b64 / ds_read_b128): lanes
// that each take their own aligned group do not meet in a bank the way they do with N single reads N words apart
template <int N> __device__ __forceinline__ void lds_read_words(uint32_t const *p, uint32_t (&o)[N])
{
	static_assert(N == 2 || N == 4 || N == 8, "two, four or eight words");
	if constexpr (N == 2) { uint2 const w = *reinterpret_cast<uint2 const *>(p); o[0] = w.x; o[1] = w.y; }
	else
	{
#pragma unroll
		for (int h = 0; h < N / 4; ++h)
		{
			uint4 const w = *reinterpret_cast<uint4 const *>(p + 4 * h);
			o[4 * h] = w.x; o[4 * h + 1] = w.y; o[4 * h + 2] = w.z; o[4 * h + 3] = w.w;
		}
	}
}

// -DFSEQ_KC_STAMPS: cycle stamps of a column step per wave (k_columns prints them)
#ifdef FSEQ_KC_STAMPS
struct KcStamps { long long acc[8]; long long last; };
#define KC_T(i) do { if (kc) { long long const t_ = clock64(); kc->acc[i] += t_ - kc->last; kc->last = t_; } } while (0)
#else
#define KC_T(i) do {} while (0)
#endif

template <int T, int SIGMA>
struct StepScratch {
	static constexpr int NW = T / WAVE;
	static constexpr int NC = SIGMA / 2;
	uint32_t cnt[NW][NC];
	uint32_t val[NW][SIGMA];
	uint32_t has[NW];
	// IDLE0 with the keyed scan: the idle wave scans the wave totals for everybody; pre[w] = {counts in key position (4),
	// keys (4)} of the waves in front of wave w
	alignas(16) uint32_t pre[NW + 1][8];
	alignas(16) uint32_t glob[8];           // {bucket starts (4), (start << 16) + first_val (4)}
};

// Running state of a partition that is fed tile by tile (orders too long for LDS): what all the tiles
// to the left contributed (32-bit counts: the order may hold more than 65535 rows), and the bucket
// starts of the whole order.
struct TileCarry {
	uint32_t cnt[4], val[4], has, start[4];
};

// One partition step.  In: d[e], s[e] for the thread's E consecutive positions (s >= SIGMA marks an
// unused tail position; its d must be 0).  Out: dst[e], dnew[e].  Contains exactly one
// __syncthreads(); the caller must barrier again before the scratch is reused.
// TILE: the T*E positions are one tile of a longer order; *tc carries the prefix of the tiles
// before it (and is advanced past this tile), bucket starts come from tc->start.
// IDLE0: wave 0 owns no rows (phase C keeps it free for the per-column list, fseq_kernels.hpp): it contributes the
// identity summary and skips the arithmetic, but meets the step's barrier.
// KS (key shift, 16 .. 25; 0 = off): every d and first_val is below 2^KS (value ids of phase C, block-relative
// divergences) and one step holds fewer than 2^(32 - KS) rows.
// The scan of the running maxima -- combine(L, R).val[x] = R.has[x] ? R.val[x] : max(L.val[x], R.val[x]), four
// instructions per symbol and step -- then becomes a plain max-scan of keys (occurrences of x so far) << KS | val[x]:
// the bucket counts are scanned first (they are needed anyway), a thread's key carries its INCLUSIVE count of x, so
// the maximum over the threads to the left picks the threads behind the last x (equal counts) and among them the
// largest value; a thread that holds x itself restarts with its own tail maximum.  The key's upper half is at the
// same time the exclusive bucket count and the "seen before" test.  ~250 -> ~100 instructions per step.
// PW ("pairwise"; four symbols, KS != 0): the thread's own rows without the four running maxima.  A row's value is the
// maximum since the nearest earlier row of the thread with its symbol -- E (E - 1) / 2 compares and selects over the
// chain maxima max d(j, e] -- and the tail maximum behind the LAST row of every symbol, which the scan needs, is a
// dynamically indexed store: the rows write their suffix maxima into runs[symbol][thread] (LDS, [5][T] words,
// conflict-free: lanes differ in thread mod 64), later rows over earlier ones.  ~16 instead of ~22 vector instructions
// per row at E = 6 (every one of them 4.2 cycles of a SIMD: profiles/r03_valu_rates.txt).
// FM (runs: at least 4 * T words): first_val is >= every value of the order (phase C: the id of the new column), so a bucket nobody has
// seen needs no select -- its prefix maximum IS first_val --, and the per-row choice among the four buckets' {start,
// prefix maximum} words is a read of the run slots (the words written there once the tail maxima have been read).
// KO (KS = 25): the count part of a key counts OCCURRENCES instead of rows -- lanes of the wave that hold the bucket, then
// waves of the workgroup that do (at most 64 + 16: 7 bits) -- so the values may be anything below 2^25 whatever T * E is
// (phase B and pass 2 scan absolute column numbers: C5's (1024, 10) had 18 bits for them, the streamed rows none).  All a
// key's count has to do is grow wherever the bucket occurs; the rows in front of a thread come from the count scan.
template <int T, int E, int SIGMA, bool TILE = false, bool IDLE0 = false, int KS = 0, bool PW = false, bool FM = false, bool KO = false>
__device__ __forceinline__ void partition_step(
	uint32_t const (&d)[E], uint32_t const (&s)[E], uint32_t const first_val,
	StepScratch<T, SIGMA> &scr, uint32_t (&dst)[E], uint32_t (&dnew)[E], TileCarry *tc = nullptr, uint32_t *runs = nullptr
#ifdef FSEQ_KC_STAMPS
	, KcStamps *kc = nullptr
#endif
	)
{
	// the idle wave has nothing to do between the step's barrier and the caller's: it runs the second level of the scan
	// (the wave totals) and hands every wave its prefix through LDS -- ~35 instructions less in every row wave, one more barrier
	constexpr bool SCAN0 = IDLE0 && KS != 0 && SIGMA == 4 && !TILE;
	if constexpr (SCAN0)
	{
		if (wave_id() == 0)
		{
			constexpr int NW_ = T / WAVE;
			static_assert(NW_ <= 16, "wave totals fit one DPP row");
			uint32_t const l = lane_id();
#pragma unroll
			for (int e = 0; e < E; ++e) { dst[e] = 0; dnew[e] = 0; }
			KC_T(0);
			__syncthreads();
			KC_T(1);
			bool const in = l >= 1u && l < (uint32_t) NW_;       // lane l = wave l; wave 0 has no rows
			uint32_t wc[2], wk[4];
#pragma unroll
			for (int i = 0; i < 2; ++i) wc[i] = in ? scr.cnt[l & (NW_ - 1)][i] : 0u;
#pragma unroll
			for (int x = 0; x < 4; ++x) wk[x] = in ? scr.val[l & (NW_ - 1)][x] : 0u;
#pragma unroll
			for (int i = 0; i < 2; ++i)
			{
				wc[i] += dpp_mov<DPP_ROW_SHR1, 0xF>(0u, wc[i]);
				if (NW_ > 2) wc[i] += dpp_mov<DPP_ROW_SHR2, 0xF>(0u, wc[i]);
				if (NW_ > 4) wc[i] += dpp_mov<DPP_ROW_SHR4, 0xF>(0u, wc[i]);
				if (NW_ > 8) wc[i] += dpp_mov<DPP_ROW_SHR8, 0xF>(0u, wc[i]);
			}
#pragma unroll
			for (int x = 0; x < 4; ++x)
			{
				uint32_t k = ((x & 1) ? ((wc[x >> 1] >> 16) << KS) : ((wc[x >> 1] & 0xFFFFu) << KS)) | wk[x];
				k = max(k, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, k));
				if (NW_ > 2) k = max(k, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, k));
				if (NW_ > 4) k = max(k, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, k));
				if (NW_ > 8) k = max(k, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, k));
				wk[x] = k;
			}
			if (l < (uint32_t) NW_ - 1u)                           // inclusive over waves 0 .. l = exclusive of wave l + 1
			{
				*reinterpret_cast<uint4 *>(&scr.pre[l + 1][0]) = make_uint4((wc[0] & 0xFFFFu) << KS, (wc[0] >> 16) << KS, (wc[1] & 0xFFFFu) << KS, (wc[1] >> 16) << KS);
				*reinterpret_cast<uint4 *>(&scr.pre[l + 1][4]) = make_uint4(wk[0], wk[1], wk[2], wk[3]);
			}
			if (l == (uint32_t) NW_ - 1u)                          // the totals: bucket starts
			{
				uint32_t const s1 = wc[0] & 0xFFFFu, s2 = s1 + (wc[0] >> 16), s3 = s2 + (wc[1] & 0xFFFFu);
				*reinterpret_cast<uint4 *>(&scr.glob[0]) = make_uint4(0u, s1, s2, s3);
				*reinterpret_cast<uint4 *>(&scr.glob[4]) = make_uint4(first_val, (s1 << 16) + first_val, (s2 << 16) + first_val, (s3 << 16) + first_val);
			}
			KC_T(2);
			__syncthreads();
			KC_T(3);
			return;
		}
	}
	else if (IDLE0 && wave_id() == 0)
	{
		static_assert(!IDLE0 || (T / WAVE > 1 && !TILE), "an idle wave needs other waves");
		if (lane_id() == 63)
		{
#pragma unroll
			for (int i = 0; i < SIGMA / 2; ++i) scr.cnt[0][i] = 0;
#pragma unroll
			for (int x = 0; x < SIGMA; ++x) scr.val[0][x] = 0;
			scr.has[0] = 0;
		}
#pragma unroll
		for (int e = 0; e < E; ++e) { dst[e] = 0; dnew[e] = 0; }
		__syncthreads();
		return;
	}
	static_assert(!TILE || (SIGMA == 4 && T / WAVE > 4), "tile mode: 4 buckets, second-level scan path");
	static_assert(E <= 15, "local counts are nibble-packed");
	static_assert(SIGMA == 4 || SIGMA == 16, "sigma instantiations");
	constexpr int NW = T / WAVE;
	constexpr int NC = SIGMA / 2;
	uint32_t const lane = lane_id();
	uint32_t const wave = wave_id();

	// ---- local pass over the thread's E rows
	uint32_t run[SIGMA];
#pragma unroll
	for (int x = 0; x < SIGMA; ++x) run[x] = 0;
	uint32_t has = 0, pend = 0;
	// nibble x = rows with symbol x seen so far in this thread (four symbols: one 32-bit word, and an unused position,
	// s = 4, counts in a fifth nibble nobody reads -- no 64-bit shifts and nothing conditional in the bookkeeping)
	using LcpT = std::conditional_t<SIGMA == 4, uint32_t, uint64_t>;
	LcpT lcp = 0;
	uint32_t lidx[E];
	// pairwise local pass, the values: dnew[e] = max d(j, e] for the nearest earlier row j of this thread with the symbol of
	// row e, the whole prefix max d[0 .. e] when there is none.  Nothing in the scan needs them: with the idle wave running
	// the second level (SCAN0) the row waves compute them while they wait for it.
	// Six and more rows (TWO): the rows are taken as two halves, pairs only inside a half (E^2 / 4 instead of E^2 / 2 of
	// them); a row of the second half whose symbol has no earlier row THERE starts from the first half's tail maximum of
	// that symbol (t0s: read from the run slots) joined with the prefix maximum of the second half.  BASELINE C3 (E = 6)
	// phase C 5.09 -> 4.98 ms; C5 (E = 10, which the one-level form never paid for) 37.0 -> 36.4 ms: 10 % fewer vector
	// instructions there, but its LDS is nearly as busy as its SIMDs and the slots cost 20 % more LDS instructions.
#ifndef FSEQ_PW_TWO_MIN
#define FSEQ_PW_TWO_MIN 6
#endif
	constexpr bool TWO = PW && E >= FSEQ_PW_TWO_MIN;
	constexpr int E0 = TWO ? (E + 1) / 2 : E;
	uint32_t t0s[TWO ? E - E0 : 1];
	auto pw_chain = [&]() {
		uint32_t pre[E];
		pre[0] = d[0];
#pragma unroll
		for (int e = 1; e < E; ++e) pre[e] = (e == E0) ? d[e] : max(pre[e - 1], d[e]);      // (TWO: per half)
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			int const lo = (e >= E0) ? E0 : 0;                   // first row of e's half
			uint32_t o = (e >= E0) ? max(t0s[TWO ? e - E0 : 0], pre[e]) : pre[e];
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
	};
	if constexpr (PW)
	{
		static_assert(!PW || (SIGMA == 4 && KS != 0), "pairwise local pass: four symbols, keyed scan");
		// pre0 = max d[0 .. E0-1]; suf[j] = max d(j, last row of j's half]
		uint32_t pre0 = d[0];
#pragma unroll
		for (int e = 1; e < E0; ++e) pre0 = max(pre0, d[e]);
		uint32_t suf[E];
		suf[E0 - 1] = 0u;
#pragma unroll
		for (int j = E0 - 2; j >= 0; --j) suf[j] = max(suf[j + 1], d[j + 1]);
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const sh = s[e] * 4u;
			lidx[e] = (uint32_t) (lcp >> sh) & 15u;
			pend |= ((s[e] < 4u) && lidx[e] == 0u) ? (1u << e) : 0u;
			lcp += (LcpT) 1u << sh;
		}
		// tail maximum per symbol: the whole thread if the symbol does not occur, else what follows its last row
		uint32_t const t_ = threadIdx.x;
#pragma unroll
		for (int x = 0; x < 4; ++x) runs[x * T + t_] = pre0;
#pragma unroll
		for (int e = 0; e < E0; ++e) runs[s[e] * (uint32_t) T + t_] = suf[e];
		if constexpr (TWO)
		{
#pragma unroll
			for (int e = E0; e < E; ++e) t0s[e - E0] = runs[s[e] * (uint32_t) T + t_];
			uint32_t t0x[4];
#pragma unroll
			for (int x = 0; x < 4; ++x) t0x[x] = runs[x * T + t_];
			uint32_t pre1 = d[E0];
#pragma unroll
			for (int e = E0 + 1; e < E; ++e) pre1 = max(pre1, d[e]);
			suf[E - 1] = 0u;
#pragma unroll
			for (int j = E - 2; j >= E0; --j) suf[j] = max(suf[j + 1], d[j + 1]);
#pragma unroll
			for (int x = 0; x < 4; ++x) runs[x * T + t_] = max(t0x[x], pre1);
#pragma unroll
			for (int e = E0; e < E; ++e) runs[s[e] * (uint32_t) T + t_] = suf[e];
		}
		if constexpr (!SCAN0) pw_chain();
#pragma unroll
		for (int x = 0; x < 4; ++x) run[x] = runs[x * T + t_];
	}
	else
	{
#pragma unroll
	for (int e = 0; e < E; ++e)
	{
		uint32_t const c = s[e];
		bool const act = c < (uint32_t) SIGMA;
		uint32_t const de = d[e];
		uint32_t o = 0;
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
		{
			uint32_t const r = max(run[x], de);
			bool const is = (c == (uint32_t) x);
			o = is ? r : o;
			run[x] = is ? 0u : r;
		}
		dnew[e] = o;
		// (the value is needed only behind the barrier; left alone, the compiler sinks the four selects there and keeps
		// the four (c == x) conditions of every row alive in SGPR pairs instead -- in kernels that already spill them)
		asm volatile("" : "+v"(dnew[e]));
		if constexpr (SIGMA == 4)
		{
			uint32_t const sh = c * 4u;                        // (c <= 4)
			lidx[e] = (uint32_t) (lcp >> sh) & 15u;
			pend |= (act && lidx[e] == 0u) ? (1u << e) : 0u;     // first row of its symbol in this thread
			lcp += (LcpT) 1u << sh;
		}
		else
		{
			uint32_t const sh = (c & 15u) * 4u;
			lidx[e] = (uint32_t) (lcp >> sh) & 15u;
			pend |= (act && !((has >> (c & 15u)) & 1u)) ? (1u << e) : 0u;
			lcp += act ? ((LcpT) 1u << sh) : (LcpT) 0u;
			has |= act ? (1u << (c & 15u)) : 0u;
		}
	}
	}
	if constexpr (SIGMA == 4)
		has = ((lcp & 0xFu) ? 1u : 0u) | ((lcp & 0xF0u) ? 2u : 0u) | ((lcp & 0xF00u) ? 4u : 0u) | ((lcp & 0xF000u) ? 8u : 0u);

	if constexpr (KS != 0)
	{
		constexpr uint32_t VMASK = (1u << KS) - 1u;
		static_assert(KS >= 16 && KS <= 25, "key = count << KS | value: counts of one step below 2^(32 - KS)");
		static_assert(KO || (uint64_t) T * E < (1ull << (32 - KS)), "counts of one step must fit the key");
		static_assert(!KO || (SIGMA == 4 && KS == 25 && !IDLE0 && !PW && !FM), "occurrence keys: four buckets, 7 bits of count");
		// ---- bucket counts: inclusive over the lanes (two 16-bit counts per word: at most T * E <= 65535 rows)
		uint32_t ic[NC], own[NC];
#pragma unroll
		for (int i = 0; i < NC; ++i)
		{
			uint32_t const lo = (uint32_t) (lcp >> (8 * i)) & 15u;
			uint32_t const hi = (uint32_t) (lcp >> (8 * i + 4)) & 15u;
			own[i] = lo | (hi << 16);
			ic[i] = wave_incl_add(own[i]);
		}
		auto occ_key = [](uint32_t const (&c)[NC], int x) -> uint32_t {        // count of x, in key position
			return (x & 1) ? ((c[x >> 1] >> 16) << KS) : ((c[x >> 1] & 0xFFFFu) << KS);
		};
		auto field = [](uint32_t const (&c)[NC], int x) -> uint32_t { return (x & 1) ? (c[x >> 1] >> 16) : (c[x >> 1] & 0xFFFFu); };
		// KO: inclusive number of lanes (waves, in the second level) up to this one that hold the bucket, in key position
		auto occurrences = [](bool h) -> uint32_t {
			uint64_t const b = __ballot(h);
			return (__builtin_amdgcn_mbcnt_hi((uint32_t) (b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) b, 0u)) + (h ? 1u : 0u)) << KS;
		};
		// ---- keys, inclusive max-scan
		uint32_t key[SIGMA];
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
			key[x] = wave_incl_max((KO ? occurrences(((uint32_t) (lcp >> (4 * x)) & 15u) != 0u) : occ_key(ic, x)) | run[x]);
		if (NW > 1 && lane == 63)
		{
#pragma unroll
			for (int i = 0; i < NC; ++i) scr.cnt[wave][i] = ic[i];
#pragma unroll
			for (int x = 0; x < SIGMA; ++x) scr.val[wave][x] = key[x] & VMASK;
		}
		// exclusive within the wave (lane 0: the identity)
		uint32_t ek[SIGMA];
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) ek[x] = dpp_mov<DPP_WAVE_SHR1, 0xF>(0u, key[x]);
		uint32_t totc[NC], pc[NC], pk[SIGMA], totk[SIGMA];
#pragma unroll
		for (int i = 0; i < NC; ++i) pc[i] = 0;
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) pk[x] = 0;
		uint32_t base[SIGMA], cval[SIGMA], seen = 0;
		KC_T(0);
		__syncthreads();
		KC_T(1);
		if constexpr (SCAN0)
		{
			if constexpr (PW) pw_chain();
			KC_T(2);
			__syncthreads();
			KC_T(3);
			uint4 const q0 = *reinterpret_cast<uint4 const *>(&scr.pre[wave][0]);
			uint4 const q1 = *reinterpret_cast<uint4 const *>(&scr.pre[wave][4]);
			uint4 const g0 = *reinterpret_cast<uint4 const *>(&scr.glob[0]);
			uint32_t const pck[4] = {q0.x, q0.y, q0.z, q0.w}, pkk[4] = {q1.x, q1.y, q1.z, q1.w}, st[4] = {g0.x, g0.y, g0.z, g0.w};
			if constexpr (KS == 16 && FM)
			{
				// {bucket start + rows in front, prefix maximum} in one word per bucket, chosen per row by a read of the run slots
				uint4 const g1 = *reinterpret_cast<uint4 const *>(&scr.glob[4]);
				uint32_t const alt[4] = {g1.x, g1.y, g1.z, g1.w};
				uint32_t const t_ = threadIdx.x;
#pragma unroll
				for (int x = 0; x < 4; ++x)
				{
					uint32_t const k = max(pkk[x], ek[x] + pck[x]);
					runs[x * T + t_] = (k >> 16) ? (st[x] << 16) + k : alt[x];
				}
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					uint32_t const sel = runs[s[e] * (uint32_t) T + t_];
					if ((pend >> e) & 1u) dnew[e] = max(sel & 0xFFFFu, dnew[e]);
					dst[e] = (sel >> 16) + lidx[e];
				}
				return;
			}
			else
			{
#pragma unroll
				for (int x = 0; x < SIGMA; ++x)
				{
					uint32_t const k = max(pkk[x & 3], ek[x] + pck[x & 3]);
					base[x] = st[x & 3] + (k >> KS);
					cval[x] = k & VMASK;
					seen |= (k >> KS) ? (1u << x) : 0u;
				}
			}
		}
		else if (NW == 1)
		{
#pragma unroll
			for (int i = 0; i < NC; ++i) totc[i] = readlane_u32(ic[i], 63);
#pragma unroll
			for (int x = 0; x < SIGMA; ++x) totk[x] = TILE ? readlane_u32(key[x], 63) : 0u;
		}
		else
		{
			// second level over the NW <= 16 wave totals (lanes 0 .. NW-1 of DPP row 0), every wave for itself
			static_assert(NW <= 16, "wave totals fit one DPP row");
			uint32_t wc[NC], wk[SIGMA];
#pragma unroll
			for (int i = 0; i < NC; ++i) wc[i] = lane < (uint32_t) NW ? scr.cnt[lane][i] : 0u;
#pragma unroll
			for (int x = 0; x < SIGMA; ++x) wk[x] = lane < (uint32_t) NW ? scr.val[lane][x] : 0u;
			uint32_t wocc[SIGMA];                                  // (KO) waves up to this lane's that hold the bucket
#pragma unroll
			for (int x = 0; x < SIGMA; ++x) wocc[x] = KO ? occurrences(field(wc, x) != 0u) : 0u;
#pragma unroll
			for (int i = 0; i < NC; ++i)
			{
				wc[i] += dpp_mov<DPP_ROW_SHR1, 0xF>(0u, wc[i]);
				if (NW > 2) wc[i] += dpp_mov<DPP_ROW_SHR2, 0xF>(0u, wc[i]);
				if (NW > 4) wc[i] += dpp_mov<DPP_ROW_SHR4, 0xF>(0u, wc[i]);
				if (NW > 8) wc[i] += dpp_mov<DPP_ROW_SHR8, 0xF>(0u, wc[i]);
			}
#pragma unroll
			for (int x = 0; x < SIGMA; ++x)
			{
				uint32_t k = (KO ? wocc[x] : occ_key(wc, x)) | wk[x];
				k = max(k, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, k));
				if (NW > 2) k = max(k, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, k));
				if (NW > 4) k = max(k, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, k));
				if (NW > 8) k = max(k, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, k));
				wk[x] = k;
			}
#pragma unroll
			for (int i = 0; i < NC; ++i) totc[i] = readlane_u32(wc[i], NW - 1);
#pragma unroll
			for (int x = 0; x < SIGMA; ++x) totk[x] = TILE ? readlane_u32(wk[x], NW - 1) : 0u;
			if (wave > 0)
			{
				int const src = (int) __builtin_amdgcn_readfirstlane(wave) - 1;
#pragma unroll
				for (int i = 0; i < NC; ++i) pc[i] = readlane_u32(wc[i], src);
#pragma unroll
				for (int x = 0; x < SIGMA; ++x) pk[x] = readlane_u32(wk[x], src);
			}
		}
		// ---- this thread's prefix: key of the waves to the left against the lanes to the left (whose counts are
		// wave-local: lifted by the count of the waves to the left); its upper part = rows of bucket x in front
		if constexpr (!SCAN0)
		{
			uint32_t acc = 0;
#pragma unroll
			for (int x = 0; x < SIGMA; ++x)
			{
				// (KO: the count part of pk[x] IS what the waves to the left add to a wave-local count)
				uint32_t const k = max(pk[x], ek[x] + (KO ? (pk[x] & ~VMASK) : occ_key(pc, x)));
				uint32_t const tot = (totc[x >> 1] >> ((x & 1) * 16)) & 0xFFFFu;
				// rows of the bucket in front of this thread (inside the tile)
				uint32_t front = k >> KS;
				if constexpr (KO)
				{
					uint32_t ex[NC];
#pragma unroll
					for (int i = 0; i < NC; ++i) ex[i] = ic[i] - own[i] + pc[i];      // (16-bit fields: no borrow, no carry)
					front = field(ex, x);
				}
				if (TILE)
				{
					// (tiles to the left) (+) (threads to the left in this tile); then the carry moves past this tile
					base[x] = tc->start[x & 3] + tc->cnt[x & 3] + front;
					cval[x] = (k >> KS) ? (k & VMASK) : max(tc->val[x & 3], k & VMASK);
					seen |= ((k >> KS) || ((tc->has >> x) & 1u)) ? (1u << x) : 0u;
				}
				else
				{
					base[x] = acc + front;
					cval[x] = k & VMASK;
					seen |= (k >> KS) ? (1u << x) : 0u;
					acc += tot;
				}
			}
			if (TILE)
			{
#pragma unroll
				for (int x = 0; x < SIGMA; ++x)
				{
					uint32_t const tot = (totc[x >> 1] >> ((x & 1) * 16)) & 0xFFFFu;
					tc->cnt[x & 3] += tot;
					tc->val[x & 3] = (totk[x] >> KS) ? (totk[x] & VMASK) : max(tc->val[x & 3], totk[x] & VMASK);
					tc->has |= (totk[x] >> KS) ? (1u << x) : 0u;
				}
			}
		}
		if constexpr (SIGMA == 4 && KS == 16 && !TILE)
		{
			// bucket start (< 65536 rows) and prefix maximum (< 65536) of a symbol share a word: one 4-way select per row,
			// as a tree over the two bits of the symbol (an unused position, s = 4, reads entry 0; nothing uses it)
			if constexpr (FM)
			{
				uint32_t const t_ = threadIdx.x;
#pragma unroll
				for (int x = 0; x < 4; ++x) runs[x * T + t_] = (base[x] << 16) | (((seen >> x) & 1u) ? cval[x] : first_val);
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					uint32_t const sel = runs[s[e] * (uint32_t) T + t_];
					if ((pend >> e) & 1u) dnew[e] = max(sel & 0xFFFFu, dnew[e]);
					dst[e] = (sel >> 16) + lidx[e];
				}
				return;
			}
			uint32_t pv[4];
#pragma unroll
			for (int x = 0; x < 4; ++x) pv[x] = (base[x] << 16) | cval[x];
#pragma unroll
			for (int e = 0; e < E; ++e)
			{
				uint32_t const c = s[e];
				bool const b0 = c & 1u, b1 = c & 2u;
				uint32_t const lo2 = b0 ? pv[1] : pv[0], hi2 = b0 ? pv[3] : pv[2];
				uint32_t const sel = b1 ? hi2 : lo2;
				if ((pend >> e) & 1u)
					dnew[e] = ((seen >> (c & 15u)) & 1u) ? max(sel & 0xFFFFu, dnew[e]) : first_val;
				dst[e] = (sel >> 16) + lidx[e];
			}
			return;
		}
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const c = s[e];
			uint32_t b = 0, cv = 0;
#pragma unroll
			for (int x = 0; x < SIGMA; ++x)
			{
				bool const is = (c == (uint32_t) x);
				b = is ? base[x] : b;
				cv = is ? cval[x] : cv;
			}
			if ((pend >> e) & 1u)
				dnew[e] = ((seen >> (c & 15u)) & 1u) ? max(cv, dnew[e]) : first_val;
			dst[e] = b + lidx[e];
		}
		return;
	}

	Summary<SIGMA> S;
#pragma unroll
	for (int i = 0; i < NC; ++i)
	{
		uint32_t const lo = (uint32_t) (lcp >> (8 * i)) & 15u;
		uint32_t const hi = (uint32_t) (lcp >> (8 * i + 4)) & 15u;
		S.cnt[i] = lo | (hi << 16);
	}
#pragma unroll
	for (int x = 0; x < SIGMA; ++x) S.val[x] = run[x];
	S.has = has;

	// ---- wave-level inclusive scan (DPP; the identity summary is neutral for prepend)
	S.prepend(S.template dpp_shifted<DPP_ROW_SHR1, 0xF>());
	S.prepend(S.template dpp_shifted<DPP_ROW_SHR2, 0xF>());
	S.prepend(S.template dpp_shifted<DPP_ROW_SHR4, 0xF>());
	S.prepend(S.template dpp_shifted<DPP_ROW_SHR8, 0xF>());
	S.prepend(S.template dpp_shifted<DPP_ROW_BCAST15, 0xA>());
	S.prepend(S.template dpp_shifted<DPP_ROW_BCAST31, 0xC>());
	if (NW > 1 && lane == 63)
	{
#pragma unroll
		for (int i = 0; i < NC; ++i) scr.cnt[wave][i] = S.cnt[i];
#pragma unroll
		for (int x = 0; x < SIGMA; ++x) scr.val[wave][x] = S.val[x];
		scr.has[wave] = S.has;
	}
	// exclusive within the wave (wave_shr:1; lane 0 keeps the identity)
	Summary<SIGMA> C = S.template dpp_shifted<DPP_WAVE_SHR1, 0xF>();

	Summary<SIGMA> TOT;                     // whole-block totals (counts only are used)
	if (NW == 1)
	{
		TOT = S.from_lane_uniform(63);
		__syncthreads();
	}
	else
	{
		__syncthreads();
		if (NW <= 4)
		{
			Summary<SIGMA> P;               // fold of the waves to the left of this one
			P.clear();
			TOT.clear();
#pragma unroll
			for (int w = 0; w < NW; ++w)
			{
				Summary<SIGMA> W;
#pragma unroll
				for (int i = 0; i < NC; ++i) W.cnt[i] = scr.cnt[w][i];
#pragma unroll
				for (int x = 0; x < SIGMA; ++x) W.val[x] = scr.val[w][x];
				W.has = scr.has[w];
#pragma unroll
				for (int i = 0; i < NC; ++i) TOT.cnt[i] += W.cnt[i];
				if ((uint32_t) w < wave) { W.prepend(P); P = W; }
			}
			C.prepend(P);
		}
		else
		{
			// second-level scan over the NW wave totals, done redundantly by every wave
			Summary<SIGMA> W;
			W.clear();
			if (lane < (uint32_t) NW)
			{
#pragma unroll
				for (int i = 0; i < NC; ++i) W.cnt[i] = scr.cnt[lane][i];
#pragma unroll
				for (int x = 0; x < SIGMA; ++x) W.val[x] = scr.val[lane][x];
				W.has = scr.has[lane];
			}
			// NW <= 16: the totals sit in lanes 0..NW-1 of DPP row 0; lanes >= NW hold the identity
			W.prepend(W.template dpp_shifted<DPP_ROW_SHR1, 0xF>());
			if (NW > 2) W.prepend(W.template dpp_shifted<DPP_ROW_SHR2, 0xF>());
			if (NW > 4) W.prepend(W.template dpp_shifted<DPP_ROW_SHR4, 0xF>());
			if (NW > 8) W.prepend(W.template dpp_shifted<DPP_ROW_SHR8, 0xF>());
			TOT = W.from_lane_uniform(NW - 1);
			Summary<SIGMA> P = W.from_lane_uniform(wave == 0 ? 0 : (int) __builtin_amdgcn_readfirstlane(wave) - 1);
			if (wave == 0) P.clear();
			C.prepend(P);
		}
	}

	// ---- bucket bases and resolution of the pending (first-in-thread) rows
	uint32_t base[SIGMA];
	if (TILE)
	{
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
		{
			base[x] = tc->start[x & 3] + tc->cnt[x & 3] + C.count(x);
			// prefix seen by this thread = (tiles to the left) (+) (threads to the left in this tile)
			C.val[x] = ((C.has >> x) & 1u) ? C.val[x] : max(tc->val[x & 3], C.val[x]);
		}
		C.has |= tc->has;
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
		{
			tc->cnt[x & 3] += TOT.count(x);
			tc->val[x & 3] = ((TOT.has >> x) & 1u) ? TOT.val[x] : max(tc->val[x & 3], TOT.val[x]);
		}
		tc->has |= TOT.has;
	}
	else
	{
		uint32_t acc = 0;
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
		{
			base[x] = acc + C.count(x);
			acc += TOT.count(x);
		}
	}
#pragma unroll
	for (int e = 0; e < E; ++e)
	{
		uint32_t const c = s[e];
		uint32_t b = 0, cv = 0;
#pragma unroll
		for (int x = 0; x < SIGMA; ++x)
		{
			bool const is = (c == (uint32_t) x);
			b = is ? base[x] : b;
			cv = is ? C.val[x] : cv;
		}
		bool const seen_before = (C.has >> (c & 15u)) & 1u;
		if ((pend >> e) & 1u)
			dnew[e] = seen_before ? max(cv, dnew[e]) : first_val;
		dst[e] = b + lidx[e];
	}
}

// A workgroup publishes "all my global stores are done" to the HOST: every storing wave drains its stores,
// the workgroup meets, one lane releases at system scope (the XCD L2's dirty lines are written back) and
// stores the flag into host-coherent memory.  The consumer is never a running kernel: the host polls the
// flag and only then launches the kernel that reads the data, whose dispatch carries the acquire
// (MI355X_MICROARCH.md, inter-workgroup visibility: producer form; HSA memory model for the rest).
// -DFSEQ_CLOCK_STAMPS (diagnostic build only; the product kernels execute no stamp): the in-kernel clock of phase C's
// kernels, MI355X_MICROARCH.md "DVFS give-back" item 6 -- every workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) once at its start and once at its end; clock = d(memtime) / d(memrealtime) x 100 MHz, the median
// over the workgroups (fseq_debug_clock).  The stamps go to a buffer of their own that nothing else reads.
#ifdef FSEQ_CLOCK_STAMPS
constexpr uint32_t FSEQ_CLOCK_SLOTS = 8192;
__device__ unsigned long long g_clock_stamps[FSEQ_CLOCK_SLOTS][4];
__device__ __forceinline__ void clock_stamp(uint32_t slot, int which)
{
	if (threadIdx.x == 0 && slot < FSEQ_CLOCK_SLOTS)
	{
		g_clock_stamps[slot][2 * which] = __builtin_amdgcn_s_memtime();
		g_clock_stamps[slot][2 * which + 1] = __builtin_amdgcn_s_memrealtime();
	}
}
#define FSEQ_CLOCK_STAMP(slot, which) clock_stamp((slot), (which))
#else
#define FSEQ_CLOCK_STAMP(slot, which) do { } while (0)
#endif

__device__ __forceinline__ void publish_block_done(uint32_t *done_host, uint32_t index, uint32_t epoch)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (done_host && threadIdx.x == 0)
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_store(done_host + index, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

} // namespace fseq
