// fseq_blockkeys.hpp -- phase A without the per-column sweep: the block keys ranked in key space.
//
// Phase A needs, per block of B columns, the dense co-lex rank of every row's block key and the divergence in
// front of every distinct key (what k_colblock<MODE_RANK> gets by running the pBWT over the block from the
// identity: B stable partitions of all m rows).  Neither needs a row ORDER in between -- only which keys are
// equal and how distinct keys compare -- so the block is ranked as a tree over its columns instead:
//   leaf     16 bits of one row = 8 / 4 / 2 consecutive columns at 2 / 4 / 8 bits per symbol, later columns in the
//            higher bits; dense rank of the word = number of distinct smaller words: a 65536-bit presence bitmap
//            in LDS, one prefix popcount over it, one lookup per row
//   merge    two neighbouring ranges (ids lo < D_lo over the earlier columns, hi < D_hi over the later ones): the
//            rank of (hi, lo) the same way on a D_hi x D_lo bitmap -- ids are dense ranks in co-lex order, so the
//            pair order IS the key order (libbio's pbwt sorts by reversed prefixes: the later column is the more
//            significant one)
//   shape    groups of 2^GL leaves are merged as balanced trees, the groups then one after the other into the
//            running prefix (D_prefix x D_group bits), ~2 B / 8 merges of ~25 instructions per row instead of B
//            partition steps of ~110
//   keyd     for neighbouring distinct keys j - 1, j: two representative rows, the last group in which their ids
//            differ (group ids are kept in an HBM scratch), then the last leaf word / column in which they differ
// Exact whatever the data; a merge whose bitmap would not fit the LDS budget (very diverse blocks: D_prefix x
// D_group > cap) is done in slices of whole hi values.
// Two forms: the id arrays in LDS as 16-bit words (m <= 11,264, the LDS-resident kernel configurations; bk_rank8 /
// blockkeys_tree_lds), or in a per-workgroup HBM / L2 workspace (the streamed regime: BASELINE C4's m = 100,000;
// bk_merge / blockkeys_tree_stream) -- as halfwords too while no range has more than 65,536 distinct keys, else as
// 32-bit words (the tree is bound by that id traffic).
//
// LDS-resident form, shaped by what the stamps of the first version showed (245 rank operations of ~5,900 cycles and
// 123 word builds of ~5,500 per C3 block: LDS instruction issue and barriers, not arithmetic):
//   * a thread owns 8 CONSECUTIVE rows: one 16-byte LDS access per id array and step instead of eight 2-byte ones,
//     and -- the same thread builds, ranks and rewrites a row throughout -- no barrier around the word build
//   * the presence map is {bits, prefix} pairs: one 8-byte read per look-up
//   * the prefix over the map words is wave-local; the look-up adds the offset of the word's wave with a
//     ds_bpermute from the scanned wave totals -> two barriers per rank operation: [set bits] | [popcount, local
//     prefix, clear the other map] | [look-up]
//   * the leaf's word build sets the bits itself (the set phase of a leaf rank is the build)
#pragma once

#include "fseq_kernels.hpp"

namespace fseq {

constexpr uint32_t BK_GL = 3;                 // 2^3 leaves per group (64 columns at 2 bits per symbol)
constexpr uint32_t BK_LEAF_BITS = 65536;      // one bit per 16-bit leaf word
// A rank operation whose map exceeds the LDS runs in slices (exact for any data), each slice one more pass over the rows:
// fine for the odd diverse block, ruinous when every merge of every block needs dozens of them (rows that are all
// distinct: BASELINE C5's shape ran phase A in 1.1 s instead of 7 ms, profiles/r04_diversity_sweep.txt).  So a block has a
// budget of extra slices (about what its rank operations cost without any); a rank operation that would pass it returns
// BK_ABORT, the tree gives the block up, and the column sweep (k_colblock<MODE_RANK>) does it instead.  0 = no limit.
constexpr uint32_t BK_ABORT = 0xFFFFFFFEu;

#ifdef FSEQ_BK_STAMPS
#define BK_T(S, i) do { long long const t_ = clock64(); (S).acc_t[i] += t_ - (S).last_t; (S).last_t = t_; } while (0)
#else
#define BK_T(S, i) do {} while (0)
#endif

// rows of a u16 array in the HBM scratch: whole 8-row pieces (16-byte stores)
__host__ __device__ inline size_t bk_m8(uint32_t m) { return ((size_t) m + 7) & ~size_t(7); }

// threads of the LDS-resident kernel: one thread per 8 rows where that fits
__host__ __device__ inline uint32_t blockkeys_threads(uint32_t m)
{
	uint32_t const t = (m + 7u) / 8u;
	return t <= 256u ? 256u : t <= 320u ? 320u : t <= 512u ? 512u : t <= 768u ? 768u : 1024u;
}

// LDS of the LDS-resident tree: prefix ids + (GL + 1) stack arrays of m u16, two maps of cap_words {bits, prefix}
// pairs, the wave totals, the staged columns of one leaf (8 >> (2 - bsh) columns of ld bytes)
__host__ __device__ inline size_t blockkeys_lds_bytes(uint32_t m, uint32_t cap_words, int T, size_t ld = 0, uint32_t bsh = 2)
{
	return carve_bytes(m, 2) * (BK_GL + 2) + carve_bytes(2 * (size_t) cap_words, 8) + carve_bytes((size_t) T / WAVE + 1, 4)
	     + carve_bytes((size_t) (8u >> (2u - bsh)) * ld + 16, 1);
}
// ... of the streamed tree: bitmap + 32-bit prefix counts + scan scratch
__host__ __device__ inline size_t blockkeys_stream_lds_bytes(uint32_t cap_words, int T)
{
	return carve_bytes(cap_words, 4) * 3 + carve_bytes(2 * ((size_t) T / WAVE + 1), 4);
}

// halfwords per block of the HBM scratch of the LDS-resident tree: leaf words [nleaf][m8] and group ids [ngrp][m8], u16
__host__ __device__ inline size_t blockkeys_scratch_halfwords(uint32_t m, uint32_t B, uint32_t bsh)
{
	uint32_t const cl = 8u >> (2u - bsh);                     // columns per leaf
	uint32_t const nleaf = (B + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	return ((size_t) nleaf + ngrp) * bk_m8(m);
}
// words per WORKGROUP of the streamed tree's workspace: the id arrays (GL + 2) x m and the group ids [ngrp][m], u32
__host__ __device__ inline size_t blockkeys_stream_ws_words(uint32_t m, uint32_t B, uint32_t bsh)
{
	uint32_t const cl = 8u >> (2u - bsh);
	uint32_t const nleaf = (B + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	return ((size_t) (BK_GL + 2) + ngrp) * m + 64;
}

// =================================================================================================================
// LDS-resident form
// =================================================================================================================
struct BkLds {
	// (no arrays in here: a dynamically indexed member would move the whole struct to scratch memory)
	uint16_t *acc;                 // ids over the columns merged so far
	uint16_t *stk0;                // DFS stack of id arrays inside a group: level i at stk0 + i * stk_stride
	uint32_t stk_stride;           // halfwords, a multiple of 8
	uint2 *map0;                   // two maps of cap_words {bits, wave-local prefix}, used alternately: the idle one is cleared while the other is summed
	uint32_t *scr;                 // popcount totals of the waves
	uint32_t cap_words;
	uint32_t turn;                 // rank operations done so far (selects the map)
	uint32_t used_a, used_b;       // entries of map 0 / 1 that may be non-zero
	uint32_t budget;               // extra slices the block may take in all (0: any number)
#ifdef FSEQ_BK_STAMPS
	long long acc_t[8], last_t;
#endif
	__device__ __forceinline__ uint16_t *stk(uint32_t i) const { return stk0 + (size_t) i * stk_stride; }
	__device__ __forceinline__ uint2 *cur() const { return map0 + (size_t) (turn & 1u) * cap_words; }
};

// Barrier over the LDS traffic only: __syncthreads() also waits for the wave's outstanding HBM accesses (vmcnt(0)) --
// here that would be the prefetch of the next leaf's columns and the stores of the leaf words, both of which have
// all the time in the world.
__device__ __forceinline__ void bk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// halfword u (0 .. 7) of a 16-byte piece
__device__ __forceinline__ uint32_t bk_half(uint4 const &q, uint32_t u)
{
	uint32_t const w = u < 2 ? q.x : u < 4 ? q.y : u < 6 ? q.z : q.w;
	return (u & 1u) ? w >> 16 : w & 0xFFFFu;
}
__device__ __forceinline__ uint4 bk_pack(uint32_t const (&v)[8])
{
	return make_uint4(v[0] | v[1] << 16, v[2] | v[3] << 16, v[4] | v[5] << 16, v[6] | v[7] << 16);
}

// Set bit k of a map.  Most rows repeat a key another row has already marked (BASELINE C5: ~1,000 distinct keys among
// 10,000 rows), and atomics on one word serialise: a plain read first (same-address reads broadcast) leaves the atomic to
// the rows that find their bit clear -- a stale read only repeats an atomic.  The set phase and the barrier behind it
// were 43 % of a C5 block's time.
__device__ __forceinline__ void bk_set(uint2 *map, uint32_t k)
{
	uint32_t const bit = 1u << (k & 31u);
	if (!(map[k >> 5].x & bit)) atomicOr(&map[k >> 5].x, bit);
}

// Dense rank of (hi[r], lo[r]) over the rows, hi the more significant (later columns); hi == nullptr: of lo[r] alone
// (a leaf: Dlo = 65536 possible words, its bits already set by the word build: PRESET).  The Dhi x Dlo map is
// processed in slices of whole hi values when it exceeds the LDS budget (very diverse blocks): a slice ranks the
// rows whose hi falls into it, earlier slices hold the smaller keys.  out may alias lo.  A row is read and written
// by the thread that owns it in every pass (rows 8 t .. 8 t + 7, then + 8 T), so consecutive rank operations need
// no barrier between them.  mid() runs between the two barriers of the first slice.
template <int T, bool PRESET, typename F>
__device__ __forceinline__ uint32_t bk_rank8(BkLds &S, uint32_t m, uint32_t Dlo, uint32_t Dhi, uint16_t const *lo, uint16_t const *hi,
                                             uint16_t *out, uint32_t *sliced, F &&mid)
{
	constexpr uint32_t NW = T / WAVE;
	uint32_t const tid = threadIdx.x;
	uint32_t const cap_bits = S.cap_words * 32u;
	uint32_t const hps = ((uint64_t) Dlo * Dhi <= cap_bits) ? Dhi : max(1u, cap_bits / Dlo);     // hi values per slice (Dlo <= cap_bits)
	if (hps < Dhi)
	{
		uint32_t const extra = (Dhi + hps - 1u) / hps - 1u;
		if (S.budget && *sliced + extra > S.budget) return BK_ABORT;       // (uniform: every thread sees the same sizes)
		*sliced += extra;
	}
	uint32_t base = 0;
	for (uint32_t h0 = 0; h0 < Dhi; h0 += hps)
	{
		uint32_t const h1 = min(Dhi, h0 + hps);
		bool const whole = (h0 == 0u && h1 == Dhi);
		uint32_t const W = ((h1 - h0) * Dlo + 31u) >> 5;
		uint32_t const par = S.turn & 1u;
		uint2 *const map = S.map0 + (size_t) par * S.cap_words, *const other = S.map0 + (size_t) (par ^ 1u) * S.cap_words;
		uint32_t const other_used = par ? S.used_a : S.used_b;
		if (par) { S.used_a = 0u; S.used_b = W; } else { S.used_b = 0u; S.used_a = W; }
		++S.turn;
		if (!PRESET)
		{
			if (whole)
				// (the common case: no slice test, nothing between the eight atomics of a thread)
				for (uint32_t r0 = tid * 8u; r0 < m; r0 += T * 8u)
				{
					uint4 const lq = *reinterpret_cast<uint4 const *>(lo + r0);
					uint4 const hq = *reinterpret_cast<uint4 const *>(hi + r0);
					uint32_t k[8];
#pragma unroll
					for (uint32_t u = 0; u < 8; ++u) k[u] = __umul24(bk_half(hq, u), Dlo) + bk_half(lq, u);
#pragma unroll
					for (uint32_t u = 0; u < 8; ++u)
					{
						uint32_t const kk = r0 + u < m ? k[u] : k[0];   // (a row behind m sets the bit of row r0 once more)
						bk_set(map, kk);
					}
				}
			else
				for (uint32_t r0 = tid * 8u; r0 < m; r0 += T * 8u)
				{
					uint4 const lq = *reinterpret_cast<uint4 const *>(lo + r0);
					uint4 const hq = *reinterpret_cast<uint4 const *>(hi + r0);
#pragma unroll
					for (uint32_t u = 0; u < 8; ++u)
					{
						uint32_t const h = bk_half(hq, u);
						if (r0 + u < m && h >= h0 && h < h1)
						{
							uint32_t const k = __umul24(h - h0, Dlo) + bk_half(lq, u);
							bk_set(map, k);
						}
					}
				}
		}
#ifdef FSEQ_BK_STAMPS
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		BK_T(S, 0);
		bk_barrier();
		BK_T(S, 1);
#else
		bk_barrier();
#endif
		if (h0 == 0u) mid();
#ifdef FSEQ_BK_STAMPS
		asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
		BK_T(S, 2);
#endif
		// Popcounts of the map words and their wave-local prefix; the wave totals go to the scratch.  A wave takes the
		// 64 << lp entries [wave << (6 + lp), ...): rows of 128, lane i the entries 2 i and 2 i + 1 of every row -- the
		// 16-byte accesses of consecutive lanes are consecutive (no bank conflicts; 64-byte pieces per lane ran at a
		// quarter of the LDS rate), one wave scan per row.
		uint32_t lp = 0;
		while (((uint32_t) T << lp) < W) ++lp;
		if (lp == 0)
		{
			uint32_t const s = tid < W ? (uint32_t) __popc(map[tid].x) : 0u;
			uint32_t const inc = wave_incl_add(s);
			if (lane_id() == 63) S.scr[wave_id()] = inc;
			if (tid < W) map[tid].y = inc - s;
		}
		else
		{
			uint32_t const cb = wave_id() << (6u + lp);
			uint32_t carry = 0;
			for (uint32_t j = 0; j < (1u << (lp - 1u)); ++j)
			{
				uint32_t const w = cb + (j << 7) + 2u * lane_id();
				uint4 e = make_uint4(0, 0, 0, 0);
				if (w < W) e = *reinterpret_cast<uint4 const *>(map + w);       // (entry W of an odd W is zero and inside the map)
				uint32_t const c0 = (uint32_t) __popc(e.x), s = c0 + (uint32_t) __popc(e.z);
				uint32_t const inc = wave_incl_add(s);
				e.y = carry + inc - s;
				e.w = e.y + c0;
				if (w < W) *reinterpret_cast<uint4 *>(map + w) = e;
				carry += readlane_u32(inc, 63);
			}
			if (lane_id() == 63) S.scr[wave_id()] = carry;
		}
		// the other map has been read for the last time before this call's first barrier: clear what its last use
		// may have set, for the next rank operation
#ifdef FSEQ_BK_STAMPS
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		BK_T(S, 5);
#endif
		for (uint32_t w = tid * 2u; w < other_used; w += T * 2u) *reinterpret_cast<uint4 *>(other + w) = make_uint4(0u, 0u, 0u, 0u);
#ifdef FSEQ_BK_STAMPS
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		BK_T(S, 6);
#endif
		bk_barrier();
		BK_T(S, 7);
		// exclusive scan of the wave totals, one per lane; the look-up fetches the entry of its word's wave with a
		// ds_bpermute -- which reads 0 from an inactive lane: the loop runs with every lane of the wave, rows or not
		uint32_t const wt = lane_id() < NW ? S.scr[lane_id()] : 0u;
		uint32_t const wi = wave_incl_add(wt);
		uint32_t const total = readlane_u32(wi, 63);
		int const wex = (int) (wi - wt);
		uint32_t const wsh = 6u + lp;
		uint32_t const rlast = (m - 1u) & ~7u;
		if (whole)
			// (the common case: the eight map reads and permutes of a thread in flight together, one 16-byte store)
			for (uint32_t rb = 0; rb < m; rb += T * 8u)
			{
				uint32_t const r0 = rb + tid * 8u, rl = min(r0, rlast);
				uint4 const lq = *reinterpret_cast<uint4 const *>(lo + rl);
				uint4 hq = make_uint4(0, 0, 0, 0);
				if (!PRESET) hq = *reinterpret_cast<uint4 const *>(hi + rl);
				uint32_t k[8], id[8];
				uint2 e[8];
#pragma unroll
				for (uint32_t u = 0; u < 8; ++u) k[u] = PRESET ? bk_half(lq, u) : __umul24(bk_half(hq, u), Dlo) + bk_half(lq, u);
#pragma unroll
				for (uint32_t u = 0; u < 8; ++u) e[u] = map[k[u] >> 5];
#pragma unroll
				for (uint32_t u = 0; u < 8; ++u) id[u] = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((k[u] >> 5 >> wsh) << 2), wex);
#pragma unroll
				for (uint32_t u = 0; u < 8; ++u) id[u] += base + e[u].y + (uint32_t) __popc(e[u].x & ((1u << (k[u] & 31u)) - 1u));
				if (r0 < m) *reinterpret_cast<uint4 *>(out + r0) = bk_pack(id);
			}
		else
			for (uint32_t rb = 0; rb < m; rb += T * 8u)
			{
				uint32_t const r0 = rb + tid * 8u, rl = min(r0, rlast);
				uint4 const lq = *reinterpret_cast<uint4 const *>(lo + rl);
				uint4 const hq = *reinterpret_cast<uint4 const *>(hi + rl);
#pragma unroll
				for (uint32_t u = 0; u < 8; ++u)
				{
					uint32_t const h = bk_half(hq, u);
					bool const in = r0 + u < m && h >= h0 && h < h1;
					uint32_t const k = in ? __umul24(h - h0, Dlo) + bk_half(lq, u) : 0u;
					uint2 const e = map[k >> 5];
					uint32_t const wpre = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((k >> 5 >> wsh) << 2), wex);
					if (in) out[r0 + u] = (uint16_t) (base + wpre + e.y + (uint32_t) __popc(e.x & ((1u << (k & 31u)) - 1u)));
				}
			}
		base += total;
		BK_T(S, 3);
	}
	return base;
}

// dst[r] = src[r] (and dst2[r], if given) for the rows this thread owns; dst in HBM, rows padded to 8
template <int T>
__device__ __forceinline__ void bk_copy8(uint32_t m, uint16_t const *src, uint16_t *__restrict__ dst, uint16_t *dst2 = nullptr)
{
	for (uint32_t r0 = threadIdx.x * 8u; r0 < m; r0 += T * 8u)
	{
		uint4 const v = *reinterpret_cast<uint4 const *>(src + r0);
		*reinterpret_cast<uint4 *>(dst + r0) = v;
		if (dst2) *reinterpret_cast<uint4 *>(dst2 + r0) = v;
	}
}

// cw[c] = the 16 rows' symbols of column c (row j at bits 2j) -> out[i] = word(row 2i) | word(row 2i + 1) << 16,
// word(row) = column c at bits 2c
__device__ __forceinline__ void bk_words8(uint32_t const (&cw)[8], uint32_t (&out)[8])
{
#pragma unroll
	for (uint32_t b = 0; b < 4; ++b)
	{
		uint32_t const sel = b | ((4u + b) << 8) | 0x0C0C0000u;                       // byte b of S1, byte b of S0, 0, 0
		uint32_t const p01 = __builtin_amdgcn_perm(cw[1], cw[0], sel), p23 = __builtin_amdgcn_perm(cw[3], cw[2], sel);
		uint32_t const p45 = __builtin_amdgcn_perm(cw[5], cw[4], sel), p67 = __builtin_amdgcn_perm(cw[7], cw[6], sel);
		uint32_t x = __builtin_amdgcn_perm(p23, p01, 0x05040100u);                    // byte c = rows 4b .. 4b+3 of column c
		uint32_t y = __builtin_amdgcn_perm(p67, p45, 0x05040100u);                    // ... of column 4 + c
		// 4 x 4 transpose of 2-bit elements: (c, r) at bit 8c + 2r -> (r, c) at bit 8r + 2c
		uint32_t t;
		t = ((x >> 6) ^ x) & 0x00CC00CCu; x ^= t ^ (t << 6);
		t = ((x >> 12) ^ x) & 0x0000F0F0u; x ^= t ^ (t << 12);
		t = ((y >> 6) ^ y) & 0x00CC00CCu; y ^= t ^ (t << 6);
		t = ((y >> 12) ^ y) & 0x0000F0F0u; y ^= t ^ (t << 12);
		out[2 * b] = __builtin_amdgcn_perm(y, x, 0x05010400u);                        // rows 4b, 4b + 1: {x0, y0, x1, y1}
		out[2 * b + 1] = __builtin_amdgcn_perm(y, x, 0x07030602u);                    // rows 4b + 2, 4b + 3
	}
}

// The 16-bit words of one leaf (nc <= 16 / bits columns staged at sym, ld bytes apart) for the rows this thread owns:
// to `top` (LDS), to `leafw` (HBM) and as set bits of `map`.  BSH: log2 of the rows per byte (2 / 1 / 0).
template <int T, int BSH>
__device__ __forceinline__ void bk_build_leaf(uint8_t const *sym, size_t ld, uint32_t nc, uint32_t m, uint16_t *top, uint16_t *__restrict__ leafw, uint2 *map)
{
	constexpr uint32_t BITS = 8u >> BSH, CL = 16u / BITS, SMASK = (1u << BITS) - 1u;
	for (uint32_t r0 = threadIdx.x * 8u; r0 < m; r0 += T * 8u)
	{
		// the 8 rows' symbols of every column: 2 / 4 / 8 bytes each, all loads in flight together (a column behind
		// the leaf's last reads as column 0 and counts as zeros)
		uint32_t x0[CL], x1[CL];
#pragma unroll
		for (uint32_t c = 0; c < CL; ++c)
		{
			uint8_t const *const at = sym + (size_t) (c < nc ? c : 0u) * ld + (r0 >> BSH);
			x1[c] = 0;
			if (BSH == 2) x0[c] = *reinterpret_cast<uint16_t const *>(at);
			else if (BSH == 1) x0[c] = *reinterpret_cast<uint32_t const *>(at);
			else { uint2 const t = *reinterpret_cast<uint2 const *>(at); x0[c] = t.x; x1[c] = t.y; }
		}
		uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		uint4 q;
		if constexpr (BSH == 2)
		{
			// 8 columns x 8 rows of 2-bit symbols: the register transpose of the pair leaves (bk_words8; 6 instructions per
			// row instead of 24 shift / mask / or -- the word build was a quarter of a C3 block's time)
			uint32_t cw[8], o[8];
#pragma unroll
			for (uint32_t c = 0; c < 8; ++c) cw[c] = c < nc ? x0[c] : 0u;
			bk_words8(cw, o);
			q = make_uint4(o[0], o[1], o[2], o[3]);
#pragma unroll
			for (uint32_t u = 0; u < 8; ++u) w[u] = (u & 1u) ? o[u >> 1] >> 16 : o[u >> 1] & 0xFFFFu;
		}
		else
		{
#pragma unroll
			for (uint32_t c = 0; c < CL; ++c)
			{
				uint32_t const a = c < nc ? x0[c] : 0u, b = c < nc ? x1[c] : 0u;
#pragma unroll
				for (uint32_t u = 0; u < 8; ++u)
				{
					uint32_t const sh = u * BITS;
					uint32_t const v = (sh < 32u ? a >> sh : b >> (sh - 32u)) & SMASK;
					w[u] |= v << (BITS * c);
				}
			}
			q = bk_pack(w);
		}
		*reinterpret_cast<uint4 *>(top + r0) = q;
		*reinterpret_cast<uint4 *>(leafw + r0) = q;
		// (a row behind m sets the bit of row r0 once more)
#pragma unroll
		for (uint32_t u = 0; u < 8; ++u)
		{
			uint32_t const k = r0 + u < m ? w[u] : w[0];
			bk_set(map, k);
		}
	}
}

// The tree over the block [k0, kend): smem holds everything, scratch = this block's leaf words and group ids (u16).
// Returns the number of merges that had to be sliced (diagnostic only).
template <int T>
__device__ __forceinline__ uint32_t blockkeys_tree_lds(
	char *smem, uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t k0, uint64_t kend, uint32_t bsh, uint32_t cap_words,
	uint16_t *__restrict__ scratch, uint32_t *__restrict__ rank_out, uint32_t *__restrict__ keyd_out, uint32_t *__restrict__ nkeys_out, bool limited)
{
	uint32_t const tid = threadIdx.x;
	uint32_t const bits = 8u >> bsh, cl = 16u / bits;
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const nleaf = (nb + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	size_t const m8 = bk_m8(m);
	Carver cv{smem};
	BkLds S;
	S.acc = cv.take<uint16_t>(m);
	S.stk_stride = (uint32_t) (carve_bytes(m, 2) / 2);
	S.stk0 = cv.take<uint16_t>((size_t) S.stk_stride * (BK_GL + 1));
	S.map0 = cv.take<uint2>(2 * (size_t) cap_words);
	S.scr = cv.take<uint32_t>(T / WAVE + 1);
	uint8_t *const sym = cv.take<uint8_t>((size_t) cl * ld + 16);     // the columns of the current leaf, staged
	uint16_t *const leafw = scratch;                          // [nleaf][m8]
	uint16_t *const grpid = leafw + (size_t) nleaf * m8;      // [ngrp][m8]
	S.cap_words = cap_words;
	S.turn = 0;
	S.used_a = S.used_b = 0;
	S.budget = limited ? 2u * nleaf + ngrp : 0u;              // about one extra pass per rank operation of the block
	for (uint32_t w = tid; w < 2u * cap_words; w += T) S.map0[w] = make_uint2(0u, 0u);
	uint32_t Dacc = 0, sliced = 0;
#ifdef FSEQ_BK_STAMPS
	for (int i = 0; i < 8; ++i) S.acc_t[i] = 0;
	S.last_t = clock64();
#endif

	// The columns of a leaf are one contiguous piece of the column-major alignment (cl * ld bytes): 16 bytes per
	// thread and piece; the next leaf's columns are fetched into registers before this leaf's word build and land in
	// the staging buffer behind the first barrier of its rank (every thread is done reading the buffer there, and the
	// second barrier stands before the next build).
	constexpr uint32_t NPF = 2;                               // 16-byte pieces per thread: cl * ld <= T * 32 (host checks)
	uint4 pf[NPF];
	auto fetch = [&](uint32_t l) {
		uint64_t const kc = k0 + (uint64_t) l * cl;
		uint32_t const bytes = (l < nleaf) ? (uint32_t) (min<uint64_t>(cl, kend - kc) * ld) : 0u;
#pragma unroll
		for (uint32_t q = 0; q < NPF; ++q)
		{
			uint32_t const off = (tid + q * T) * 16u;
			pf[q] = (off < bytes) ? *reinterpret_cast<uint4 const *>(msa + kc * ld + off) : make_uint4(0, 0, 0, 0);
		}
	};
	auto land = [&]() {
#pragma unroll
		for (uint32_t q = 0; q < NPF; ++q)
		{
			uint32_t const off = (tid + q * T) * 16u;
			if (off < cl * (uint32_t) ld) *reinterpret_cast<uint4 *>(sym + off) = pf[q];
		}
	};
	auto nothing = []() {};
	fetch(0);
	land();
	__syncthreads();                                          // maps cleared, leaf 0 staged

	for (uint32_t g = 0; g < ngrp; ++g)
	{
		// ---- one group: leaves left to right, equal-sized neighbours merged at once (a balanced tree, left child
		// = the largest power of two below the size), the rest merged at the end of the group
		uint32_t const l0 = g << BK_GL, l1 = min(nleaf, l0 + (1u << BK_GL));
		uint32_t sp = 0, sz[BK_GL + 1], D[BK_GL + 1];
		for (uint32_t l = l0; l < l1; ++l)
		{
			uint64_t const kc = k0 + (uint64_t) l * cl;
			uint32_t const nc = (uint32_t) min<uint64_t>(cl, kend - kc);
			uint16_t *const top = S.stk(sp);
			fetch(l + 1);
			if (bsh == 2) bk_build_leaf<T, 2>(sym, ld, nc, m, top, leafw + (size_t) l * m8, S.cur());
			else if (bsh == 1) bk_build_leaf<T, 1>(sym, ld, nc, m, top, leafw + (size_t) l * m8, S.cur());
			else bk_build_leaf<T, 0>(sym, ld, nc, m, top, leafw + (size_t) l * m8, S.cur());
			BK_T(S, 4);
			D[sp] = bk_rank8<T, true>(S, m, BK_LEAF_BITS, 1u, top, nullptr, top, &sliced, land);
			sz[sp] = 1;
			++sp;
			while (sp >= 2 && (sz[sp - 2] == sz[sp - 1] || l + 1 == l1))
			{
				D[sp - 2] = bk_rank8<T, false>(S, m, D[sp - 2], D[sp - 1], S.stk(sp - 2), S.stk(sp - 1), S.stk(sp - 2), &sliced, nothing);
				if (D[sp - 2] == BK_ABORT) return BK_ABORT;
				sz[sp - 2] += sz[sp - 1];
				--sp;
			}
		}
		// ---- the group joins the prefix
		uint16_t *const gi = S.stk(0);
		if (g == 0)
		{
			bk_copy8<T>(m, gi, grpid, S.acc);
			Dacc = D[0];
		}
		else
		{
			bk_copy8<T>(m, gi, grpid + (size_t) g * m8);
			Dacc = bk_rank8<T, false>(S, m, Dacc, D[0], S.acc, gi, S.acc, &sliced, nothing);
			if (Dacc == BK_ABORT) return BK_ABORT;
		}
	}

	// ---- outputs: rank of every row, one representative row per distinct key, the divergence in front of each key
	uint16_t *const rep = S.stk(1);
	__syncthreads();                                          // rep is written by key, not by row
	for (uint32_t r0 = tid * 8u; r0 < m; r0 += T * 8u)
	{
		uint4 const q = *reinterpret_cast<uint4 const *>(S.acc + r0);
#pragma unroll
		for (uint32_t u = 0; u < 8; ++u)
			if (r0 + u < m) { uint32_t const v = bk_half(q, u); rank_out[r0 + u] = v; rep[v] = (uint16_t) (r0 + u); }
	}
	__syncthreads();
	for (uint32_t j = tid; j < Dacc; j += T)
	{
		uint32_t d = (uint32_t) kend;                        // key 0: first of its bucket in every column
		if (j > 0)
		{
			uint32_t const ra = rep[j - 1], rb = rep[j];
			uint32_t g = ngrp - 1;
			while (g > 0 && grpid[(size_t) g * m8 + ra] == grpid[(size_t) g * m8 + rb]) --g;
			uint32_t const l0 = g << BK_GL, l1 = min(nleaf, l0 + (1u << BK_GL));
			uint32_t l = l1 - 1u;
			uint32_t x = (uint32_t) leafw[(size_t) l * m8 + ra] ^ (uint32_t) leafw[(size_t) l * m8 + rb];
			while (x == 0u && l > l0) { --l; x = (uint32_t) leafw[(size_t) l * m8 + ra] ^ (uint32_t) leafw[(size_t) l * m8 + rb]; }
			// highest differing symbol of the word = the last column in which the two keys differ; the common
			// suffix starts one column behind it
			uint32_t const p = (31u - (uint32_t) __builtin_clz(x | 1u)) / bits;
			d = (uint32_t) (k0 + (uint64_t) l * cl + p + 1u);
		}
		keyd_out[j] = d;
	}
	if (tid == 0) *nkeys_out = Dacc;
#ifdef FSEQ_BK_STAMPS
	if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 700))
		printf("bk stamps block %u: set %lld B1 %lld mid %lld count+scan %lld clear %lld B2 %lld lookup %lld build %lld | Dacc %u turns %u\n", blockIdx.x,
		       S.acc_t[0], S.acc_t[1], S.acc_t[2], S.acc_t[5], S.acc_t[6], S.acc_t[7], S.acc_t[3], S.acc_t[4], Dacc, S.turn);
#endif
	return sliced;
}

// =================================================================================================================
// streamed form
// =================================================================================================================
template <typename IdT, typename PrefT>
struct BkState {
	// (no arrays in here: a dynamically indexed member would move the whole struct to scratch memory)
	IdT *acc;                      // ids over the columns merged so far
	IdT *stk0;                     // DFS stack of id arrays inside a group: level i at stk0 + i * stk_stride
	size_t stk_stride;
	uint32_t *bm0;                 // two bitmaps of cap_words, used alternately: the idle one is cleared while the other is summed
	PrefT *pref;
	uint32_t *sscr;                // two scan scratch areas of T / 64 + 1 words, alternating as well
	uint32_t cap_words;
	uint32_t turn;                 // rank operations done so far (selects the bitmap)
	uint32_t used_a, used_b;       // words of bitmap 0 / 1 that may be non-zero
	uint32_t budget;               // extra slices the block may take in all (0: any number)
	__device__ __forceinline__ IdT *stk(uint32_t i) const { return stk0 + (size_t) i * stk_stride; }
};

// Dense rank of (hi[r], lo[r]) over the rows, hi the more significant (later columns); hi == nullptr: of lo[r] alone
// (a leaf: Dlo = 65536 possible words).  The Dhi x Dlo bitmap is processed in slices of whole hi values when it
// exceeds the LDS budget (very diverse blocks): a slice ranks the rows whose hi falls into it, earlier slices hold
// the smaller keys.  out may alias lo.  Every row is read and written by the same thread in every pass.
// Three barriers per slice: [set bits] | [popcount my words; clear the OTHER bitmap for the next rank] | [prefix
// over the threads -> per-word prefix counts] | [look the rows up].  The bitmap of this call was cleared during the
// call before (both are cleared once at kernel start); nothing here waits for the look-ups of the call before --
// they only read the other bitmap and the prefix counts, which are rewritten behind this call's second barrier.
template <int T, typename IdT, typename PrefT>
__device__ __forceinline__ uint32_t bk_merge(BkState<IdT, PrefT> &S, uint32_t m, uint32_t Dlo, uint32_t Dhi, IdT const *lo, IdT const *hi,
                                             IdT *out, uint32_t *sliced)
{
	// rows per thread and step: the 32-bit ids of the streamed regime live in HBM / L2 -- eight independent loads in
	// flight per thread instead of one round trip per row
	constexpr uint32_t U = 8u;
	constexpr uint32_t NW = T / WAVE;
	uint32_t const tid = threadIdx.x;
	uint32_t const cap_bits = S.cap_words * 32u;
	uint32_t const hps = ((uint64_t) Dlo * Dhi <= cap_bits) ? Dhi : max(1u, cap_bits / Dlo);     // hi values per slice (Dlo <= cap_bits)
	if (hps < Dhi)
	{
		uint32_t const extra = (Dhi + hps - 1u) / hps - 1u;
		if (S.budget && *sliced + extra > S.budget) return BK_ABORT;       // (uniform: every thread sees the same sizes)
		*sliced += extra;
	}
	uint32_t base = 0;
	for (uint32_t h0 = 0; h0 < Dhi; h0 += hps)
	{
		uint32_t const h1 = min(Dhi, h0 + hps);
		uint32_t const W = ((h1 - h0) * Dlo + 31u) >> 5;
		uint32_t const par = S.turn & 1u;
		uint32_t *const bm = S.bm0 + par * S.cap_words, *const other = S.bm0 + (par ^ 1u) * S.cap_words;
		uint32_t *const scr = S.sscr + par * (NW + 1u);
		uint32_t const other_used = par ? S.used_a : S.used_b;
		if (par) { S.used_a = 0u; S.used_b = W; } else { S.used_b = 0u; S.used_a = W; }
		++S.turn;
		for (uint32_t r0 = tid; r0 < m; r0 += T * U)
		{
			uint32_t lv[U], hv[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				uint32_t const r = r0 + u * T;
				lv[u] = r < m ? (uint32_t) lo[r] : 0u;
				hv[u] = (hi && r < m) ? (uint32_t) hi[r] : (r < m ? 0u : 0xFFFFFFFFu);
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
				if (hv[u] >= h0 && hv[u] < h1)
				{
					uint32_t const k = (hv[u] - h0) * Dlo + lv[u];
					uint32_t const bit = 1u << (k & 31u);
					if (!(bm[k >> 5] & bit)) atomicOr(&bm[k >> 5], bit);   // rows of one founder share keys: most find their bit set
				}
		}
		__syncthreads();
		// my words' popcounts; the wave totals go to the scan scratch
		uint32_t const per = (W + T - 1) / T, w0 = tid * per;
		uint32_t s = 0;
		for (uint32_t q = 0; q < per; ++q)
			if (w0 + q < W) s += (uint32_t) __popc(bm[w0 + q]);
		uint32_t const inc = wave_incl_add(s);
		if (lane_id() == 63) scr[wave_id()] = inc;
		// the other bitmap has been read for the last time before this call's first barrier: clear what its last use
		// may have set, for the next rank operation
		for (uint32_t w = tid; w < other_used; w += T) other[w] = 0u;
		__syncthreads();
		uint32_t pre = 0, total = 0;
#pragma unroll
		for (uint32_t w = 0; w < NW; ++w)
		{
			uint32_t const x = scr[w];
			if (w < wave_id()) pre += x;
			total += x;
		}
		uint32_t run = pre + inc - s;
		for (uint32_t q = 0; q < per; ++q)
			if (w0 + q < W) { S.pref[w0 + q] = (PrefT) run; run += (uint32_t) __popc(bm[w0 + q]); }
		__syncthreads();
		for (uint32_t r0 = tid; r0 < m; r0 += T * U)
		{
			uint32_t lv[U], hv[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				uint32_t const r = r0 + u * T;
				lv[u] = r < m ? (uint32_t) lo[r] : 0u;
				hv[u] = (hi && r < m) ? (uint32_t) hi[r] : (r < m ? 0u : 0xFFFFFFFFu);
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
				if (hv[u] >= h0 && hv[u] < h1)
				{
					uint32_t const k = (hv[u] - h0) * Dlo + lv[u];
					out[r0 + u * T] = (IdT) (base + (uint32_t) S.pref[k >> 5] + (uint32_t) __popc(bm[k >> 5] & ((1u << (k & 31u)) - 1u)));
				}
		}
		base += total;
	}
	return base;
}

// ------------------------------------------------------------------------------------------------------------------
// Pair leaf (2-bit symbols): the dense rank of the 32-bit key of SIXTEEN columns in one go.  A leaf (8 columns) used to
// cost a word build (2 B per row written), its rank (read, read, write) and half a merge with its neighbour (two reads of
// two id arrays, one write): ~15 B per row and leaf through L2 / HBM, which is what bounds the streamed tree.  Here the
// two leaf words of a row are rebuilt from the packed columns in every pass (a 16 x 8 transpose of 2-bit symbols in
// registers: byte permutes + two delta swaps per 4 x 4 tile) instead of being kept in memory:
//   pass 1  both words -> one bit each in two 65,536-bit maps (first bitmap);             prefix popcounts of both
//   pass 2  both ranks (id_lo < D_lo, id_hi < D_hi) -> bit id_hi * D_lo + id_lo (second bitmap);   prefix popcounts
//   pass 3  the rank of that bit -> out[row]
// 3 x 4 B of columns + 2 B of ids per row and pair = 7 B per row and leaf.  Returns the number of distinct keys, or
// BK_PAIR_NO (uniformly, bitmaps clean) when D_lo * D_hi does not fit the part of the maps that is left -- the caller
// then takes the two leaves one by one.
// ------------------------------------------------------------------------------------------------------------------
constexpr uint32_t BK_PAIR_NO = 0xFFFFFFFFu;

// exclusive prefix popcounts of the words bm[0 .. nW) into pref[0 .. nW); returns the total.  Three barriers; the last
// one makes pref visible.
template <int T, typename PrefT>
__device__ __forceinline__ uint32_t bk_prefix_words(uint32_t const *bm, PrefT *pref, uint32_t nW, uint32_t *scr)
{
	constexpr uint32_t NW = T / WAVE;
	uint32_t const tid = threadIdx.x;
	uint32_t const per = (nW + T - 1) / T, w0 = tid * per;
	uint32_t s = 0;
	for (uint32_t q = 0; q < per; ++q)
		if (w0 + q < nW) s += (uint32_t) __popc(bm[w0 + q]);
	uint32_t const inc = wave_incl_add(s);
	__syncthreads();                                           // (the scratch may still be read by the call before)
	if (lane_id() == 63) scr[wave_id()] = inc;
	__syncthreads();
	uint32_t pre = 0, total = 0;
#pragma unroll
	for (uint32_t w = 0; w < NW; ++w)
	{
		uint32_t const x = scr[w];
		if (w < wave_id()) pre += x;
		total += x;
	}
	uint32_t run = pre + inc - s;
	for (uint32_t q = 0; q < per; ++q)
		if (w0 + q < nW) { pref[w0 + q] = (PrefT) run; run += (uint32_t) __popc(bm[w0 + q]); }
	__syncthreads();
	return total;
}

template <int T, typename IdT, typename PrefT>
__device__ __forceinline__ uint32_t bk_pair_leaf(BkState<IdT, PrefT> &S, uint32_t m, uint8_t const *__restrict__ msa, size_t ld, uint64_t kc, uint32_t nc,
                                                 IdT *__restrict__ out)
{
	constexpr uint32_t LW = BK_LEAF_BITS / 32u;                // words of one leaf map
	uint32_t const tid = threadIdx.x;
	uint32_t *const bmA = S.bm0, *const bmB = S.bm0 + S.cap_words;
	// both bitmaps clean (a rank operation leaves its own bitmap to the next one to clear)
	for (uint32_t w = tid; w < S.used_a; w += T) bmA[w] = 0u;
	for (uint32_t w = tid; w < S.used_b; w += T) bmB[w] = 0u;
	S.used_a = S.used_b = 0u;
	__syncthreads();
	uint32_t const nq = (m + 15u) / 16u;                       // row groups: 16 rows share a 32-bit word of a packed column
	auto load_words = [&](uint32_t q, uint32_t (&lo)[8], uint32_t (&hi)[8]) {
		uint32_t cl_[8], ch_[8];
#pragma unroll
		for (uint32_t c = 0; c < 8; ++c)
		{
			cl_[c] = (c < nc) ? *reinterpret_cast<uint32_t const *>(msa + (kc + c) * ld + (size_t) q * 4u) : 0u;
			ch_[c] = (8u + c < nc) ? *reinterpret_cast<uint32_t const *>(msa + (kc + 8u + c) * ld + (size_t) q * 4u) : 0u;
		}
		bk_words8(cl_, lo);
		bk_words8(ch_, hi);
	};
	// (the two leaf maps and their prefix counts take 2 LW words: a bitmap smaller than that -- FSEQ_BLOCKKEYS_CAP in the tests --
	// has no room for pass 1, so the capacity is tested before anything is written)
	if (S.cap_words < 2u * LW + 64u) return BK_PAIR_NO;
	// ---- pass 1
	for (uint32_t q = tid; q < nq; q += T)
	{
		uint32_t lo[8], hi[8];
		load_words(q, lo, hi);
#pragma unroll
		for (uint32_t j = 0; j < 16; ++j)
			if (q * 16u + j < m)
			{
				uint32_t const wl = (lo[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu, wh = (hi[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu;
				uint32_t const bl = 1u << (wl & 31u), bh = 1u << (wh & 31u);
				if (!(bmA[wl >> 5] & bl)) atomicOr(&bmA[wl >> 5], bl);
				if (!(bmA[LW + (wh >> 5)] & bh)) atomicOr(&bmA[LW + (wh >> 5)], bh);
			}
	}
	__syncthreads();
	uint32_t const both = bk_prefix_words<T, PrefT>(bmA, S.pref, 2u * LW, S.sscr);
	uint32_t const Dlo = (uint32_t) S.pref[LW], Dhi = both - Dlo;
	uint32_t const W = (uint32_t) (((uint64_t) Dlo * Dhi + 31u) >> 5);
	if ((uint64_t) Dlo * Dhi > (uint64_t) (S.cap_words - 2u * LW) * 32u)
	{
		for (uint32_t w = tid; w < 2u * LW; w += T) bmA[w] = 0u;
		__syncthreads();
		return BK_PAIR_NO;
	}
	PrefT *const prefM = S.pref + 2u * LW;
	auto ids_of = [&](uint32_t wl, uint32_t wh) -> uint32_t {
		uint32_t const il = (uint32_t) S.pref[wl >> 5] + (uint32_t) __popc(bmA[wl >> 5] & ((1u << (wl & 31u)) - 1u));
		uint32_t const ih = (uint32_t) S.pref[LW + (wh >> 5)] + (uint32_t) __popc(bmA[LW + (wh >> 5)] & ((1u << (wh & 31u)) - 1u)) - Dlo;
		return ih * Dlo + il;
	};
	// ---- pass 2
	for (uint32_t q = tid; q < nq; q += T)
	{
		uint32_t lo[8], hi[8];
		load_words(q, lo, hi);
#pragma unroll
		for (uint32_t j = 0; j < 16; ++j)
			if (q * 16u + j < m)
			{
				uint32_t const k = ids_of((lo[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu, (hi[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu);
				uint32_t const bit = 1u << (k & 31u);
				if (!(bmB[k >> 5] & bit)) atomicOr(&bmB[k >> 5], bit);
			}
	}
	__syncthreads();
	uint32_t const D = bk_prefix_words<T, PrefT>(bmB, prefM, W, S.sscr);
	// ---- pass 3
	for (uint32_t q = tid; q < nq; q += T)
	{
		uint32_t lo[8], hi[8];
		load_words(q, lo, hi);
#pragma unroll
		for (uint32_t j = 0; j < 16; ++j)
			if (q * 16u + j < m)
			{
				uint32_t const k = ids_of((lo[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu, (hi[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu);
				out[q * 16u + j] = (IdT) ((uint32_t) prefM[k >> 5] + (uint32_t) __popc(bmB[k >> 5] & ((1u << (k & 31u)) - 1u)));
			}
	}
	__syncthreads();
	for (uint32_t w = tid; w < 2u * LW; w += T) bmA[w] = 0u;
	for (uint32_t w = tid; w < W; w += T) bmB[w] = 0u;
	__syncthreads();
	return D;
}

// dst[r] = src[r] (and dst2[r], if given) for the rows of this thread, several loads in flight
template <int T, typename IdT>
__device__ __forceinline__ void bk_copy(uint32_t m, IdT const *__restrict__ src, IdT *__restrict__ dst, IdT *__restrict__ dst2 = nullptr)
{
	constexpr uint32_t U = 8u;
	for (uint32_t r0 = threadIdx.x; r0 < m; r0 += T * U)
	{
		IdT v[U];
#pragma unroll
		for (uint32_t u = 0; u < U; ++u) v[u] = (r0 + u * T < m) ? src[r0 + u * T] : (IdT) 0;
#pragma unroll
		for (uint32_t u = 0; u < U; ++u)
			if (r0 + u * T < m) { dst[r0 + u * T] = v[u]; if (dst2) dst2[r0 + u * T] = v[u]; }
	}
}

__device__ __forceinline__ uint32_t bk_symbol(uint8_t const *msa, size_t ld, uint64_t k, uint32_t r, uint32_t bsh)
{
	uint32_t const bits = 8u >> bsh;
	return ((uint32_t) msa[k * ld + (r >> bsh)] >> ((r & ((1u << bsh) - 1u)) * bits)) & ((1u << bits) - 1u);
}

// The tree over the block [k0, kend), streamed rows: smem holds the bitmaps, ws = this workgroup's id arrays and
// group ids (u32).  Returns the number of merges that had to be sliced (diagnostic only).
// IdT = uint16_t: the ids of the ranges as halfwords (this tree moves ~20 bytes of id traffic per row and rank
// operation through L2 / HBM and is bound by it; half of that with 16-bit ids) -- as long as no range has more than
// 65,535 distinct keys: returns BK_WIDE (uniformly, nothing written) at the first one that has, and the caller runs
// the block again with IdT = uint32_t.
constexpr uint32_t BK_WIDE = 0xFFFFFFFFu;
// (measured, BASELINE C4: as a function of its own -- __noinline__ -- the 32-bit form halves the spilled SGPRs of the kernel,
// 435 -> 224, and phase A gets SLOWER, 340 -> 350 ms: the spills were not what bounds it)
#ifndef FSEQ_BK_WIDE_INLINE
#define FSEQ_BK_WIDE_INLINE __forceinline__
#endif
template <typename IdT>
__device__ __forceinline__ uint32_t blockkeys_tree_stream(
	char *smem, uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t k0, uint64_t kend, uint32_t bsh, uint32_t cap_words,
	uint32_t *__restrict__ ws_words, uint32_t *__restrict__ rank_out, uint32_t *__restrict__ keyd_out, uint32_t *__restrict__ nkeys_out,
	bool pair_leaves, bool limited)
{
	constexpr int T = 1024;
	constexpr bool NARROW = sizeof(IdT) == 2;
	using PrefT = uint32_t;
	IdT *const ws = reinterpret_cast<IdT *>(ws_words);
	uint32_t const tid = threadIdx.x;
	uint32_t const bits = 8u >> bsh, cl = 16u / bits, smask = (1u << bits) - 1u;
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const nleaf = (nb + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	Carver cv{smem};
	BkState<IdT, PrefT> S;
	S.bm0 = cv.take<uint32_t>(2 * (size_t) cap_words);
	S.pref = cv.take<PrefT>(cap_words);
	S.sscr = cv.take<uint32_t>(2 * (T / WAVE + 1));
	IdT *grpid;                                               // [ngrp][m]
	{
		IdT *w = ws;
		S.acc = w; w += m;
		S.stk0 = w; S.stk_stride = m; w += (size_t) m * (BK_GL + 1);
		grpid = w;
	}
	S.cap_words = cap_words;
	S.turn = 0;
	S.used_a = S.used_b = 0;
	S.budget = limited ? 2u * nleaf + ngrp : 0u;              // about one extra pass per rank operation of the block
	for (uint32_t w = tid; w < 2u * cap_words; w += T) S.bm0[w] = 0u;
	__syncthreads();
	uint32_t Dacc = 0, sliced = 0;
	bool const pairs = bsh == 2u && pair_leaves;
#ifdef FSEQ_BK_STAMPS
	long long bks[5] = {0, 0, 0, 0, 0}, bkl = clock64();
#define BKS_T(i) do { long long const t_ = clock64(); bks[i] += t_ - bkl; bkl = t_; } while (0)
#else
#define BKS_T(i) do {} while (0)
#endif

	for (uint32_t g = 0; g < ngrp; ++g)
	{
		// ---- one group: leaves left to right, equal-sized neighbours merged at once (a balanced tree, left child
		// = the largest power of two below the size), the rest merged at the end of the group
		uint32_t const l0 = g << BK_GL, l1 = min(nleaf, l0 + (1u << BK_GL));
		uint32_t sp = 0, sz[BK_GL + 1], D[BK_GL + 1];
		for (uint32_t l = l0; l < l1; ++l)
		{
			uint64_t const kc = k0 + (uint64_t) l * cl;
			uint32_t const nc = (uint32_t) min<uint64_t>(cl, kend - kc);
			IdT *const top = S.stk(sp);
			// two leaves at once where the symbols are 2 bits wide and the group has both (bk_pair_leaf)
			if (pairs && ((l - l0) & 1u) == 0u && l + 1 < l1)
			{
				uint32_t const nc2 = (uint32_t) min<uint64_t>(2u * cl, kend - kc);
				__syncthreads();                                      // (`top` may still be read by the look-ups of the merge before)
				uint32_t const Dp = bk_pair_leaf<T, IdT, PrefT>(S, m, msa, ld, kc, nc2, top);
				BKS_T(0);
				if (Dp != BK_PAIR_NO)
				{
					if (NARROW && Dp > 65536u) { __syncthreads(); return BK_WIDE; }
					D[sp] = Dp; sz[sp] = 2;
					++sp; ++l;
					while (sp >= 2 && (sz[sp - 2] == sz[sp - 1] || l + 1 == l1))
					{
						D[sp - 2] = bk_merge<T, IdT, PrefT>(S, m, D[sp - 2], D[sp - 1], S.stk(sp - 2), S.stk(sp - 1), S.stk(sp - 2), &sliced);
						if (D[sp - 2] == BK_ABORT) { __syncthreads(); return BK_ABORT; }
						if (NARROW && D[sp - 2] > 65536u) { __syncthreads(); return BK_WIDE; }
						sz[sp - 2] += sz[sp - 1];
						--sp;
					}
					BKS_T(1);
					continue;
				}
			}
			// The word build writes `top` by row GROUPS (the rows that share a word of the packed column), the rank
			// operations read and write their arrays by single rows: a thread that is done with the last look-up must not
			// overwrite a row another thread has still to read there (bk_merge does not end with a barrier).
			__syncthreads();
			{
				// a thread takes the rows that share a 32-bit word of the packed column (16 / 8 / 4 rows): one coalesced
				// word per column, all of a leaf's columns in flight together
				uint32_t const rpw = 4u << bsh, nw = (m + rpw - 1u) / rpw;
				for (uint32_t q = tid; q < nw; q += T)
				{
					uint32_t cw[8];
#pragma unroll
					for (uint32_t c = 0; c < 8; ++c)
						cw[c] = (c < nc) ? *reinterpret_cast<uint32_t const *>(msa + (kc + c) * ld + (size_t) q * 4u) : 0u;
					for (uint32_t j = 0; j < rpw; ++j)
					{
						uint32_t const r = q * rpw + j;
						uint32_t w = 0;
#pragma unroll
						for (uint32_t c = 0; c < 8; ++c)
							if (c < cl) w |= ((cw[c] >> (j * bits)) & smask) << (bits * c);
						if (r < m) top[r] = (IdT) w;
					}
				}
			}
			__syncthreads();                                      // the words were written by row groups, the rank reads them by rows
			D[sp] = bk_merge<T, IdT, PrefT>(S, m, BK_LEAF_BITS, 1u, top, nullptr, top, &sliced);
			if (D[sp] == BK_ABORT) { __syncthreads(); return BK_ABORT; }
			sz[sp] = 1;
			++sp;
			while (sp >= 2 && (sz[sp - 2] == sz[sp - 1] || l + 1 == l1))
			{
				D[sp - 2] = bk_merge<T, IdT, PrefT>(S, m, D[sp - 2], D[sp - 1], S.stk(sp - 2), S.stk(sp - 1), S.stk(sp - 2), &sliced);
				if (D[sp - 2] == BK_ABORT) { __syncthreads(); return BK_ABORT; }
				if (NARROW && D[sp - 2] > 65536u) { __syncthreads(); return BK_WIDE; }
				sz[sp - 2] += sz[sp - 1];
				--sp;
			}
		}
		// ---- the group joins the prefix
		BKS_T(2);
		IdT *const gi = S.stk(0);
		if (g == 0)
		{
			bk_copy<T, IdT>(m, gi, grpid, S.acc);
			Dacc = D[0];
			__syncthreads();
		}
		else
		{
			bk_copy<T, IdT>(m, gi, grpid + (size_t) g * m);
			Dacc = bk_merge<T, IdT, PrefT>(S, m, Dacc, D[0], S.acc, gi, S.acc, &sliced);
			if (Dacc == BK_ABORT) { __syncthreads(); return BK_ABORT; }
			if (NARROW && Dacc > 65536u) { __syncthreads(); return BK_WIDE; }
		}
		BKS_T(3);
	}

	// ---- outputs: rank of every row, one representative row per distinct key, the divergence in front of each key
	// (representatives are ROW numbers, 32 bits whatever the ids: with halfword ids the upper half of the workspace is free)
	uint32_t *const rep = NARROW ? ws_words + ((size_t) (BK_GL + 2 + ngrp) * m + 3) / 2 : reinterpret_cast<uint32_t *>(S.stk(1));
	__syncthreads();
	{
		constexpr uint32_t U = 8u;
		for (uint32_t r0 = tid; r0 < m; r0 += T * U)
		{
			uint32_t v[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) v[u] = (r0 + u * T < m) ? (uint32_t) S.acc[r0 + u * T] : 0u;
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
				if (r0 + u * T < m) { rank_out[r0 + u * T] = v[u]; rep[v[u]] = r0 + u * T; }
		}
	}
	__syncthreads();
	for (uint32_t j = tid; j < Dacc; j += T)
	{
		uint32_t d = (uint32_t) kend;                        // key 0: first of its bucket in every column
		if (j > 0)
		{
			uint32_t const ra = rep[j - 1], rb = rep[j];
			uint32_t g = ngrp - 1;
			while (g > 0 && grpid[(size_t) g * m + ra] == grpid[(size_t) g * m + rb]) --g;
			uint32_t const l0 = g << BK_GL, l1 = min(nleaf, l0 + (1u << BK_GL));
			// the keys differ inside group g: the last column of the group in which the two rows differ
			uint64_t k = min<uint64_t>(kend, k0 + (uint64_t) l1 * cl);
			uint64_t const klo = k0 + (uint64_t) l0 * cl;
			while (k > klo + 1u && bk_symbol(msa, ld, k - 1u, ra, bsh) == bk_symbol(msa, ld, k - 1u, rb, bsh)) --k;
			d = (uint32_t) k;
		}
		keyd_out[j] = d;
	}
#ifdef FSEQ_BK_STAMPS
	BKS_T(4);
	if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == 100))
		printf("bk stream stamps block %u: pair leaves %lld | merges in groups %lld | single leaves etc %lld | group -> prefix %lld | outputs %lld | Dacc %u sliced %u turns %u\n",
		       blockIdx.x, bks[0], bks[1], bks[2], bks[3], bks[4], Dacc, sliced, S.turn);
#endif
	if (tid == 0) *nkeys_out = Dacc;
	__syncthreads();                                          // the workspace is reused for the next block
	return sliced;
}

// The same with 32-bit ids: a block takes it only when a range has more than 65,535 distinct keys (rows that are nearly all
// distinct).  FSEQ_BK_WIDE_INLINE decides whether it is a function of its own (experiment above)
__device__ FSEQ_BK_WIDE_INLINE uint32_t blockkeys_tree_stream_wide(
	char *smem, uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t k0, uint64_t kend, uint32_t bsh, uint32_t cap_words,
	uint32_t *__restrict__ ws_words, uint32_t *__restrict__ rank_out, uint32_t *__restrict__ keyd_out, uint32_t *__restrict__ nkeys_out,
	bool pair_leaves, bool limited)
{
	return blockkeys_tree_stream<uint32_t>(smem, msa, ld, m, k0, kend, bsh, cap_words, ws_words, rank_out, keyd_out, nkeys_out, pair_leaves, limited);
}

// Phase A, LDS-resident rows: workgroup i ranks the block of columns starting at col0 + i * B (outputs indexed by i,
// as k_colblock<MODE_RANK>).  *sliced counts the blocks in which some merge exceeded the LDS bitmap (diagnostic).
template <int T>
__global__ __launch_bounds__(T) void k_blockkeys(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t bsh,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys, uint64_t col0,
	uint16_t *__restrict__ scratch, size_t scratch_per_block, uint32_t cap_words, uint32_t *__restrict__ sliced, uint32_t *__restrict__ todo,
	uint32_t const *__restrict__ only = nullptr)
{
	// only: the blocks with a non-zero entry (what the trie of fseq_blocktrie.hpp gave up)
	if (only && only[blockIdx.x] == 0u) return;
	// todo (per block, zeroed by the host): set to 1 for a block the tree gave up on (BK_ABORT) -- the column sweep that
	// follows does those; sliced[0] counts the blocks that sliced some merge, sliced[1] the blocks given up.  todo ==
	// nullptr: no limit on the slices.
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint64_t const k0 = col0 + (uint64_t) blockIdx.x * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	size_t const ob = (size_t) blockIdx.x * m;
	uint32_t const ns = blockkeys_tree_lds<T>(smem, msa, ld, m, k0, kend, bsh, cap_words, scratch + (size_t) blockIdx.x * scratch_per_block,
	                                          rank + ob, keyd + ob, nkeys + blockIdx.x, todo != nullptr);
	if (ns == BK_ABORT) { if (threadIdx.x == 0) { todo[blockIdx.x] = 1u; if (sliced) atomicAdd(sliced + 1, 1u); } return; }
	if (ns && threadIdx.x == 0 && sliced) atomicAdd(sliced, 1u);
}

// Phase A, streamed rows (m > 11,264): gridDim.x workgroups, each with its own workspace, take the blocks
// i, i + gridDim.x, ... of the launch's nblk blocks.
static __global__ __launch_bounds__(1024) void k_blockkeys_stream(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t bsh, uint32_t nblk,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys, uint64_t col0,
	uint32_t *__restrict__ ws, size_t ws_per_group, uint32_t cap_words, uint32_t *__restrict__ sliced, uint32_t wide, uint32_t *__restrict__ todo,
	uint32_t const *__restrict__ only = nullptr)
{
	// (todo, sliced[1]: as in k_blockkeys); only: the blocks with a non-zero entry (what the trie of fseq_blocktrie.hpp gave up)
	// wide bit 0: 32-bit ids from the start (tests; else a block is tried with halfword ids first); bit 1: leaves one by one
	bool const pair_leaves = !(wide & 2u);
	wide &= 1u;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x)
	{
		if (only && only[b] == 0u) continue;
		uint64_t const k0 = col0 + (uint64_t) b * B;
		uint64_t const kend = (k0 + B < n) ? k0 + B : n;
		size_t const ob = (size_t) b * m;
		uint32_t ns = wide ? BK_WIDE : blockkeys_tree_stream<uint16_t>(smem, msa, ld, m, k0, kend, bsh, cap_words, ws + (size_t) blockIdx.x * ws_per_group,
		                                                               rank + ob, keyd + ob, nkeys + b, pair_leaves, todo != nullptr);
		if (ns == BK_WIDE)
			ns = blockkeys_tree_stream_wide(smem, msa, ld, m, k0, kend, bsh, cap_words, ws + (size_t) blockIdx.x * ws_per_group,
			                                rank + ob, keyd + ob, nkeys + b, pair_leaves, todo != nullptr);
		if (ns == BK_ABORT) { if (threadIdx.x == 0) { todo[b] = 1u; if (sliced) atomicAdd(sliced + 1, 1u); } continue; }
		if (ns && threadIdx.x == 0 && sliced) atomicAdd(sliced, 1u);
	}
}

} // namespace fseq
