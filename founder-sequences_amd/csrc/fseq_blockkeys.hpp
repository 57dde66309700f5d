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
//   shape    groups of 2^GL leaves are merged as balanced trees (all in LDS), the groups then one after the other
//            into the running prefix (D_prefix x D_group bits: never large), ~2 B / 16 merges of ~25 instructions per
//            row instead of B partition steps of ~110
//   keyd     for neighbouring distinct keys j - 1, j: two representative rows, the last group / leaf in which their
//            ids / words differ (kept in a per-block HBM scratch), the highest differing symbol of that word
// Exact whatever the data; a merge whose bitmap would not fit the LDS budget (very diverse blocks: D_prefix x
// D_group > cap) is done in slices of whole hi values.
#pragma once

#include "fseq_kernels.hpp"

namespace fseq {

constexpr uint32_t BK_GL = 3;                 // 2^3 leaves per group (64 columns at 2 bits per symbol)
constexpr uint32_t BK_LEAF_BITS = 65536;      // one bit per 16-bit leaf word

// LDS of the tree: prefix ids + (GL + 1) stack arrays of m u16, bitmap words + one u16 prefix count per word,
// representative rows (u16, m < 65536), the block scan scratch
__host__ __device__ inline size_t blockkeys_lds_bytes(uint32_t m, uint32_t cap_words, int T, size_t ld = 0, uint32_t bsh = 2)
{
	// + the staged columns of one leaf (8 >> (2 - bsh) columns of ld bytes)
	return carve_bytes(m, 2) * (BK_GL + 2) + carve_bytes(cap_words, 4) + carve_bytes(cap_words, 2) + carve_bytes((size_t) T / WAVE + 1, 4)
	     + carve_bytes((size_t) (8u >> (2u - bsh)) * ld + 16, 1);
}

// words per block of the HBM scratch: leaf words [nleaf][m] and group ids [ngrp][m], both u16
__host__ __device__ inline size_t blockkeys_scratch_halfwords(uint32_t m, uint32_t B, uint32_t bsh)
{
	uint32_t const cl = 8u >> (2u - bsh);                     // columns per leaf
	uint32_t const nleaf = (B + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	return ((size_t) nleaf + ngrp) * m;
}

struct BkLds {
	uint16_t *acc;                 // ids over the columns merged so far
	uint16_t *stk[BK_GL + 1];      // DFS stack of id arrays inside a group
	uint32_t *bm;
	uint16_t *pref;
	uint32_t *sscr;
	uint32_t cap_words;
};

// dense rank of key(r) < nbits over the rows; key(r) is evaluated twice.  ids may alias an input of key(r) that
// only row r reads.  Returns the number of distinct keys.
template <int T, typename KeyFn>
__device__ __forceinline__ uint32_t bk_rank(BkLds const &S, uint32_t m, uint32_t nbits, KeyFn key, uint16_t *ids)
{
	uint32_t const tid = threadIdx.x;
	uint32_t const W = (nbits + 31u) >> 5;
	for (uint32_t w = tid; w < W; w += T) S.bm[w] = 0u;
	__syncthreads();
	for (uint32_t r = tid; r < m; r += T)
	{
		uint32_t const k = key(r);
		uint32_t const bit = 1u << (k & 31u);
		if (!(S.bm[k >> 5] & bit)) atomicOr(&S.bm[k >> 5], bit);   // rows of one founder share keys: most find their bit set
	}
	__syncthreads();
	uint32_t const per = (W + T - 1) / T, w0 = tid * per;
	uint32_t s = 0;
	for (uint32_t q = 0; q < per; ++q)
		if (w0 + q < W) s += (uint32_t) __popc(S.bm[w0 + q]);
	uint32_t total;
	uint32_t run = block_excl_add<T>(s, S.sscr, &total);
	for (uint32_t q = 0; q < per; ++q)
		if (w0 + q < W) { S.pref[w0 + q] = (uint16_t) run; run += (uint32_t) __popc(S.bm[w0 + q]); }
	__syncthreads();
	for (uint32_t r = tid; r < m; r += T)
	{
		uint32_t const k = key(r);
		ids[r] = (uint16_t) (S.pref[k >> 5] + (uint32_t) __popc(S.bm[k >> 5] & ((1u << (k & 31u)) - 1u)));
	}
	__syncthreads();
	return total;
}

// Merge of two neighbouring column ranges: ids[r] = dense rank of (hi[r], lo[r]), hi the more significant (later
// columns).  The Dhi x Dlo bitmap is processed in slices of whole hi values when it exceeds the LDS budget (very
// diverse blocks): a slice ranks the rows whose hi falls into it, earlier slices hold the smaller keys.  out may
// alias lo.  *sliced is counted up when more than one slice was needed.
template <int T>
__device__ __forceinline__ uint32_t bk_merge(BkLds const &S, uint32_t m, uint32_t Dlo, uint32_t Dhi, uint16_t const *lo, uint16_t const *hi,
                                             uint16_t *out, uint32_t *sliced)
{
	uint32_t const cap_bits = S.cap_words * 32u;
	if ((uint64_t) Dlo * Dhi <= cap_bits)
		return bk_rank<T>(S, m, Dlo * Dhi, [&](uint32_t r) { return (uint32_t) hi[r] * Dlo + lo[r]; }, out);
	++*sliced;
	uint32_t const hps = max(1u, cap_bits / Dlo);            // hi values per slice (Dlo <= m < cap_bits)
	uint32_t const tid = threadIdx.x;
	uint32_t base = 0;
	for (uint32_t h0 = 0; h0 < Dhi; h0 += hps)
	{
		uint32_t const h1 = min(Dhi, h0 + hps);
		uint32_t const W = ((h1 - h0) * Dlo + 31u) >> 5;
		for (uint32_t w = tid; w < W; w += T) S.bm[w] = 0u;
		__syncthreads();
		for (uint32_t r = tid; r < m; r += T)
		{
			uint32_t const h = hi[r];
			if (h >= h0 && h < h1)
			{
				uint32_t const k = (h - h0) * Dlo + lo[r];
				uint32_t const bit = 1u << (k & 31u);
				if (!(S.bm[k >> 5] & bit)) atomicOr(&S.bm[k >> 5], bit);
			}
		}
		__syncthreads();
		uint32_t const per = (W + T - 1) / T, w0 = tid * per;
		uint32_t s = 0;
		for (uint32_t q = 0; q < per; ++q)
			if (w0 + q < W) s += (uint32_t) __popc(S.bm[w0 + q]);
		uint32_t total;
		uint32_t run = block_excl_add<T>(s, S.sscr, &total);
		for (uint32_t q = 0; q < per; ++q)
			if (w0 + q < W) { S.pref[w0 + q] = (uint16_t) run; run += (uint32_t) __popc(S.bm[w0 + q]); }
		__syncthreads();
		for (uint32_t r = tid; r < m; r += T)
		{
			uint32_t const h = hi[r];
			if (h >= h0 && h < h1)
			{
				uint32_t const k = (h - h0) * Dlo + lo[r];
				out[r] = (uint16_t) (base + S.pref[k >> 5] + (uint32_t) __popc(S.bm[k >> 5] & ((1u << (k & 31u)) - 1u)));
			}
		}
		__syncthreads();
		base += total;
	}
	return base;
}

// The tree over the block [k0, kend).  Returns false when some merge had to be sliced (diagnostic only).
template <int T>
__device__ __forceinline__ bool blockkeys_tree(
	char *smem, uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t k0, uint64_t kend, uint32_t bsh, uint32_t cap_words,
	uint16_t *__restrict__ scratch, uint32_t *__restrict__ rank_out, uint32_t *__restrict__ keyd_out, uint32_t *__restrict__ nkeys_out)
{
	uint32_t const tid = threadIdx.x;
	Carver cv{smem};
	BkLds S;
	S.acc = cv.take<uint16_t>(m);
	for (uint32_t i = 0; i <= BK_GL; ++i) S.stk[i] = cv.take<uint16_t>(m);
	S.bm = cv.take<uint32_t>(cap_words);
	S.pref = cv.take<uint16_t>(cap_words);
	S.sscr = cv.take<uint32_t>(T / WAVE + 1);
	S.cap_words = cap_words;

	uint32_t const bits = 8u >> bsh, cl = 16u / bits, smask = (1u << bits) - 1u;
	uint8_t *const sym = cv.take<uint8_t>((size_t) cl * ld + 16);     // the columns of the current leaf, staged
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const nleaf = (nb + cl - 1) / cl, ngrp = (nleaf + (1u << BK_GL) - 1) >> BK_GL;
	uint16_t *const leafw = scratch;                          // [nleaf][m]
	uint16_t *const grpid = scratch + (size_t) nleaf * m;     // [ngrp][m]
	uint32_t Dacc = 0, sliced = 0;

	// The columns of a leaf are one contiguous piece of the column-major alignment (cl * ld bytes): 16 bytes per
	// thread and piece; the next leaf's columns are fetched into registers while this leaf is ranked (the HBM
	// latency hides behind the ~10 barriers of a leaf and its merges).
	constexpr uint32_t NPF = 2;                               // 16-byte pieces per thread: cl * ld <= T * 32 (host checks)
	uint4 pf[NPF];
	auto fetch = [&](uint32_t l) {
		uint64_t const kc = k0 + (uint64_t) l * cl;
		uint32_t const bytes = (uint32_t) (min<uint64_t>(cl, kend - kc) * ld);
#pragma unroll
		for (uint32_t q = 0; q < NPF; ++q)
		{
			uint32_t const off = (tid + q * T) * 16u;
			pf[q] = (l < nleaf && off < bytes) ? *reinterpret_cast<uint4 const *>(msa + kc * ld + off) : make_uint4(0, 0, 0, 0);
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
	fetch(0);
	land();
	__syncthreads();
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
			uint16_t *const top = S.stk[sp];
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
					if (j < rpb && r < m) { top[r] = (uint16_t) w[j]; leafw[(size_t) l * m + r] = (uint16_t) w[j]; }
				}
			}
			D[sp] = bk_rank<T>(S, m, BK_LEAF_BITS, [&](uint32_t r) { return (uint32_t) top[r]; }, top);
			land();                                               // every thread is past the word build (barriers in bk_rank)
			sz[sp] = 1;
			++sp;
			while (sp >= 2 && (sz[sp - 2] == sz[sp - 1] || l + 1 == l1))
			{
				uint16_t *const lo = S.stk[sp - 2], *const hi = S.stk[sp - 1];
				D[sp - 2] = bk_merge<T>(S, m, D[sp - 2], D[sp - 1], lo, hi, lo, &sliced);
				sz[sp - 2] += sz[sp - 1];
				--sp;
			}
		}
		// ---- the group joins the prefix
		uint16_t *const gi = S.stk[0];
		for (uint32_t r = tid; r < m; r += T) grpid[(size_t) g * m + r] = gi[r];
		if (g == 0)
		{
			for (uint32_t r = tid; r < m; r += T) S.acc[r] = gi[r];
			Dacc = D[0];
			__syncthreads();
		}
		else
		{
			Dacc = bk_merge<T>(S, m, Dacc, D[0], S.acc, gi, S.acc, &sliced);
		}
	}

	// ---- outputs: rank of every row, one representative row per distinct key, the divergence in front of each key
	uint16_t *const rep = S.stk[1];
	for (uint32_t r = tid; r < m; r += T) { rank_out[r] = S.acc[r]; rep[S.acc[r]] = (uint16_t) r; }
	__syncthreads();
	__threadfence_block();
	for (uint32_t j = tid; j < Dacc; j += T)
	{
		uint32_t d = (uint32_t) kend;                        // key 0: first of its bucket in every column
		if (j > 0)
		{
			uint32_t const ra = rep[j - 1], rb = rep[j];
			uint32_t g = ngrp - 1;
			while (g > 0 && grpid[(size_t) g * m + ra] == grpid[(size_t) g * m + rb]) --g;
			uint32_t const l0 = g << BK_GL;
			uint32_t l = min(nleaf, l0 + (1u << BK_GL)) - 1u;
			uint32_t x = (uint32_t) leafw[(size_t) l * m + ra] ^ (uint32_t) leafw[(size_t) l * m + rb];
			while (x == 0u && l > l0) { --l; x = (uint32_t) leafw[(size_t) l * m + ra] ^ (uint32_t) leafw[(size_t) l * m + rb]; }
			// highest differing symbol of the word = the last column in which the two keys differ; the common suffix
			// starts one column behind it
			uint32_t const p = (31u - (uint32_t) __builtin_clz(x | 1u)) / bits;
			d = (uint32_t) (k0 + (uint64_t) l * cl + p + 1u);
		}
		keyd_out[j] = d;
	}
	if (tid == 0) *nkeys_out = Dacc;
	return sliced == 0;
}

// Phase A: workgroup i ranks the block of columns starting at col0 + i * B (outputs indexed by i, as
// k_colblock<MODE_RANK>).  *sliced counts the blocks in which some merge exceeded the LDS bitmap (diagnostic).
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
	bool const whole = blockkeys_tree<T>(smem, msa, ld, m, k0, kend, bsh, cap_words, scratch + (size_t) blockIdx.x * scratch_per_block,
	                                     rank + ob, keyd + ob, nkeys + blockIdx.x);
	if (!whole && threadIdx.x == 0 && sliced) atomicAdd(sliced, 1u);
}

} // namespace fseq
