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
// Two instantiations: the id arrays in LDS as 16-bit words (m <= 11,264, the LDS-resident kernel configurations),
// or in a per-workgroup HBM / L2 workspace as 32-bit words (the streamed regime: BASELINE C4's m = 100,000).
#pragma once

#include "fseq_kernels.hpp"

namespace fseq {

constexpr uint32_t BK_GL = 3;                 // 2^3 leaves per group (64 columns at 2 bits per symbol)
constexpr uint32_t BK_LEAF_BITS = 65536;      // one bit per 16-bit leaf word

// LDS of the LDS-resident tree: prefix ids + (GL + 1) stack arrays of m u16, two bitmaps + one u16 prefix count per
// word, the block scan scratch, the staged columns of one leaf (8 >> (2 - bsh) columns of ld bytes)
__host__ __device__ inline size_t blockkeys_lds_bytes(uint32_t m, uint32_t cap_words, int T, size_t ld = 0, uint32_t bsh = 2)
{
	return carve_bytes(m, 2) * (BK_GL + 2) + 2 * carve_bytes(cap_words, 4) + carve_bytes(cap_words, 2) + carve_bytes(2 * ((size_t) T / WAVE + 1), 4)
	     + carve_bytes((size_t) (8u >> (2u - bsh)) * ld + 16, 1);
}
// ... of the streamed tree: bitmap + 32-bit prefix counts + scan scratch
__host__ __device__ inline size_t blockkeys_stream_lds_bytes(uint32_t cap_words, int T)
{
	return carve_bytes(cap_words, 4) * 3 + carve_bytes(2 * ((size_t) T / WAVE + 1), 4);
}

// halfwords per block of the HBM scratch of the LDS-resident tree: leaf words [nleaf][m] and group ids [ngrp][m], u16
__host__ __device__ inline size_t blockkeys_scratch_halfwords(uint32_t m, uint32_t B, uint32_t bsh)
{
	uint32_t const cl = 8u >> (2u - bsh);                     // columns per leaf
	uint32_t const nleaf = (B + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	return ((size_t) nleaf + ngrp) * m;
}
// words per WORKGROUP of the streamed tree's workspace: the id arrays (GL + 2) x m and the group ids [ngrp][m], u32
__host__ __device__ inline size_t blockkeys_stream_ws_words(uint32_t m, uint32_t B, uint32_t bsh)
{
	uint32_t const cl = 8u >> (2u - bsh);
	uint32_t const nleaf = (B + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	return ((size_t) (BK_GL + 2) + ngrp) * m + 64;
}

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
	constexpr uint32_t U = sizeof(IdT) == 4 ? 8u : 4u;
	constexpr uint32_t NW = T / WAVE;
	uint32_t const tid = threadIdx.x;
	uint32_t const cap_bits = S.cap_words * 32u;
	uint32_t const hps = ((uint64_t) Dlo * Dhi <= cap_bits) ? Dhi : max(1u, cap_bits / Dlo);     // hi values per slice (Dlo <= cap_bits)
	if (hps < Dhi) ++*sliced;
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

// dst[r] = src[r] (and dst2[r], if given) for the rows of this thread, several loads in flight
template <int T, typename IdT>
__device__ __forceinline__ void bk_copy(uint32_t m, IdT const *__restrict__ src, IdT *__restrict__ dst, IdT *__restrict__ dst2 = nullptr)
{
	constexpr uint32_t U = sizeof(IdT) == 4 ? 8u : 4u;
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

// The tree over the block [k0, kend).  STREAM = false: smem holds everything, scratch = this block's leaf words and
// group ids (u16).  STREAM = true: smem holds the bitmap, ws = this workgroup's id arrays and group ids (u32).
// Returns the number of merges that had to be sliced (diagnostic only).
template <int T, bool STREAM>
__device__ __forceinline__ uint32_t blockkeys_tree(
	char *smem, uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t k0, uint64_t kend, uint32_t bsh, uint32_t cap_words,
	void *__restrict__ scratch_or_ws, uint32_t *__restrict__ rank_out, uint32_t *__restrict__ keyd_out, uint32_t *__restrict__ nkeys_out)
{
	using IdT = std::conditional_t<STREAM, uint32_t, uint16_t>;
	using PrefT = std::conditional_t<STREAM, uint32_t, uint16_t>;
	uint32_t const tid = threadIdx.x;
	uint32_t const bits = 8u >> bsh, cl = 16u / bits, smask = (1u << bits) - 1u;
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const nleaf = (nb + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	Carver cv{smem};
	BkState<IdT, PrefT> S;
	uint8_t *sym = nullptr;                                   // !STREAM: the columns of the current leaf, staged
	uint16_t *leafw = nullptr;                                // !STREAM: [nleaf][m]
	IdT *grpid;                                               // [ngrp][m]
	if (!STREAM)
	{
		S.acc = cv.take<IdT>(m);
		S.stk_stride = carve_bytes(m, sizeof(IdT)) / sizeof(IdT);
		S.stk0 = cv.take<IdT>(S.stk_stride * (BK_GL + 1));
		S.bm0 = cv.take<uint32_t>(2 * (size_t) cap_words);
		S.pref = cv.take<PrefT>(cap_words);
		S.sscr = cv.take<uint32_t>(2 * (T / WAVE + 1));
		sym = cv.take<uint8_t>((size_t) cl * ld + 16);
		leafw = static_cast<uint16_t *>(scratch_or_ws);
		grpid = reinterpret_cast<IdT *>(leafw + (size_t) nleaf * m);
	}
	else
	{
		S.bm0 = cv.take<uint32_t>(2 * (size_t) cap_words);
		S.pref = cv.take<PrefT>(cap_words);
		S.sscr = cv.take<uint32_t>(2 * (T / WAVE + 1));
		IdT *w = static_cast<IdT *>(scratch_or_ws);
		S.acc = w; w += m;
		S.stk0 = w; S.stk_stride = m; w += (size_t) m * (BK_GL + 1);
		grpid = w;
	}
	S.cap_words = cap_words;
	S.turn = 0;
	S.used_a = S.used_b = 0;
	for (uint32_t w = tid; w < 2u * cap_words; w += T) S.bm0[w] = 0u;
	__syncthreads();
	uint32_t Dacc = 0, sliced = 0;

	// !STREAM: the columns of a leaf are one contiguous piece of the column-major alignment (cl * ld bytes): 16 bytes
	// per thread and piece; the next leaf's columns are fetched into registers while this leaf is ranked (the HBM
	// latency hides behind the ~10 barriers of a leaf and its merges).
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
	if (!STREAM)
	{
		fetch(0);
		land();
		__syncthreads();
	}
	uint32_t const rpb = 1u << bsh, ngr = (m + rpb - 1u) >> bsh;    // rows per byte; row groups = bytes per column

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
			// The word build writes `top` by row GROUPS (the rows that share a byte / word of the packed column), the rank
			// operations read and write their arrays by single rows: a thread that is done with the last look-up must not
			// overwrite a row another thread has still to read there (bk_merge does not end with a barrier).
			__syncthreads();
			if (!STREAM)
			{
				fetch(l + 1);
				// a thread takes the rows that share a byte: one LDS byte per column gives all of their symbols
				for (uint32_t q = tid; q < ngr; q += T)
				{
					uint32_t w[4] = {0, 0, 0, 0};
					for (uint32_t c = 0; c < nc; ++c)
					{
						uint32_t const b = sym[(size_t) c * ld + q];
#pragma unroll
						for (uint32_t j = 0; j < 4; ++j)
							if (j < rpb) w[j] |= ((b >> (j * bits)) & smask) << (bits * c);
					}
#pragma unroll
					for (uint32_t j = 0; j < 4; ++j)
					{
						uint32_t const r = q * rpb + j;
						if (j < rpb && r < m) { top[r] = (IdT) w[j]; leafw[(size_t) l * m + r] = (uint16_t) w[j]; }
					}
				}
			}
			else
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
			// every thread is past this leaf's word build: the next leaf's columns may land in the staging buffer (the
			// barriers of the rank below stand between this and the next word build)
			if (!STREAM) land();
			D[sp] = bk_merge<T, IdT, PrefT>(S, m, BK_LEAF_BITS, 1u, top, nullptr, top, &sliced);
			sz[sp] = 1;
			++sp;
			while (sp >= 2 && (sz[sp - 2] == sz[sp - 1] || l + 1 == l1))
			{
				D[sp - 2] = bk_merge<T, IdT, PrefT>(S, m, D[sp - 2], D[sp - 1], S.stk(sp - 2), S.stk(sp - 1), S.stk(sp - 2), &sliced);
				sz[sp - 2] += sz[sp - 1];
				--sp;
			}
		}
		// ---- the group joins the prefix
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
		}
	}

	// ---- outputs: rank of every row, one representative row per distinct key, the divergence in front of each key
	IdT *const rep = S.stk(1);
	{
		constexpr uint32_t U = STREAM ? 8u : 1u;
		for (uint32_t r0 = tid; r0 < m; r0 += T * U)
		{
			uint32_t v[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) v[u] = (r0 + u * T < m) ? (uint32_t) S.acc[r0 + u * T] : 0u;
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
				if (r0 + u * T < m) { rank_out[r0 + u * T] = v[u]; rep[v[u]] = (IdT) (r0 + u * T); }
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
			if (!STREAM)
			{
				uint32_t l = l1 - 1u;
				uint32_t x = (uint32_t) leafw[(size_t) l * m + ra] ^ (uint32_t) leafw[(size_t) l * m + rb];
				while (x == 0u && l > l0) { --l; x = (uint32_t) leafw[(size_t) l * m + ra] ^ (uint32_t) leafw[(size_t) l * m + rb]; }
				// highest differing symbol of the word = the last column in which the two keys differ; the common
				// suffix starts one column behind it
				uint32_t const p = (31u - (uint32_t) __builtin_clz(x | 1u)) / bits;
				d = (uint32_t) (k0 + (uint64_t) l * cl + p + 1u);
			}
			else
			{
				// the keys differ inside group g: the last column of the group in which the two rows differ
				uint64_t k = min<uint64_t>(kend, k0 + (uint64_t) l1 * cl);
				uint64_t const klo = k0 + (uint64_t) l0 * cl;
				while (k > klo + 1u && bk_symbol(msa, ld, k - 1u, ra, bsh) == bk_symbol(msa, ld, k - 1u, rb, bsh)) --k;
				d = (uint32_t) k;
			}
		}
		keyd_out[j] = d;
	}
	if (tid == 0) *nkeys_out = Dacc;
	__syncthreads();                                          // (STREAM: the workspace is reused for the next block)
	return sliced;
}

// Phase A, LDS-resident rows: workgroup i ranks the block of columns starting at col0 + i * B (outputs indexed by i,
// as k_colblock<MODE_RANK>).  *sliced counts the blocks in which some merge exceeded the LDS bitmap (diagnostic).
template <int T>
__global__ __launch_bounds__(T) void k_blockkeys(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t bsh,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys, uint64_t col0,
	uint16_t *__restrict__ scratch, size_t scratch_per_block, uint32_t cap_words, uint32_t *__restrict__ sliced)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint64_t const k0 = col0 + (uint64_t) blockIdx.x * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	size_t const ob = (size_t) blockIdx.x * m;
	uint32_t const ns = blockkeys_tree<T, false>(smem, msa, ld, m, k0, kend, bsh, cap_words, scratch + (size_t) blockIdx.x * scratch_per_block,
	                                             rank + ob, keyd + ob, nkeys + blockIdx.x);
	if (ns && threadIdx.x == 0 && sliced) atomicAdd(sliced, 1u);
}

// Phase A, streamed rows (m > 11,264): gridDim.x workgroups, each with its own workspace, take the blocks
// i, i + gridDim.x, ... of the launch's nblk blocks.
__global__ __launch_bounds__(1024) void k_blockkeys_stream(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t bsh, uint32_t nblk,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys, uint64_t col0,
	uint32_t *__restrict__ ws, size_t ws_per_group, uint32_t cap_words, uint32_t *__restrict__ sliced)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x)
	{
		uint64_t const k0 = col0 + (uint64_t) b * B;
		uint64_t const kend = (k0 + B < n) ? k0 + B : n;
		size_t const ob = (size_t) b * m;
		uint32_t const ns = blockkeys_tree<1024, true>(smem, msa, ld, m, k0, kend, bsh, cap_words, ws + (size_t) blockIdx.x * ws_per_group,
		                                               rank + ob, keyd + ob, nkeys + b);
		if (ns && threadIdx.x == 0 && sliced) atomicAdd(sliced, 1u);
	}
}

} // namespace fseq
