// fseq_kernels.hpp -- HIP kernels of the segmentation path for gfx950.
//
// Column-block decomposition of segmentation_lp_context's two hot loops
// (founder-sequences/segmentation_lp_context.cc:26-188 pass 1, update_pbwt_task.cc:13-35 pass 2):
//   phase A  k_colblock<MODE_RANK>  per block of B columns: pBWT from the identity -> dense co-lex rank of
//                                   every row's block key + divergence in front of each distinct key
//   phase B  k_chain (x3)           stable sort of the running order by block rank (4-bit LSD digit passes of
//                                   the same partition step), two-level: compose groups of G blocks into
//                                   super-blocks (parallel), chain the super-blocks (one workgroup), expand
//                                   each super-block (parallel) -> exact (a_k, d_k) at every block boundary
//   phase C  k_columns              per block from its boundary state: the true per-column update; per
//                                   column the top of the divergence-value histogram (what
//                                   calculate_segmentation_lp_dp_arg consumes, lp.cc:393-481)
//   phase D  k_dp                   the DP over columns incl. the exact rmq.hh semantics
//   pass 2   k_colblock<MODE_SNAP>  (a,d) at the merged segment boundaries
// All state of a block lives in LDS; HBM traffic is the 1 B/cell column stream plus the outputs.
#pragma once

#include "fseq_core.hpp"

#include <type_traits>

namespace fseq {

constexpr uint32_t PAD_KEY = 0xFFFFFFFFu;

struct Carver {
	char *p;
	template <typename U> __device__ __host__ U *take(size_t n)
	{
		U *r = reinterpret_cast<U *>(p);
		p += (n * sizeof(U) + 15) & ~size_t(15);
		return r;
	}
};

__host__ __device__ inline size_t carve_bytes(size_t n, size_t elem) { return (n * elem + 15) & ~size_t(15); }

// ------------------------------------------------------------------------------------------------
// Synthetic founder-mosaic generator -- same arithmetic as oracle/fseq_oracle.c
// ------------------------------------------------------------------------------------------------
struct SynthArgs {
	uint64_t seed;
	uint32_t n_founders;
	uint32_t block_len;
	uint64_t mut_threshold;
	uint32_t kind;
	uint32_t sigma;
	uint8_t  code_of_sym[16];     // alphabet index -> dense code (rank of the byte)
};

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
	x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
	x ^= x >> 27; x *= 0x94d049bb133111ebULL;
	x ^= x >> 31;
	return x;
}

__device__ __forceinline__ uint64_t synth_h(uint64_t seed, uint64_t tag, uint64_t r, uint64_t c)
{
	return mix64(seed ^ mix64(tag) ^ mix64(r * 0x9E3779B97F4A7C15ULL + c));
}

__device__ __forceinline__ uint32_t synth_pick(uint32_t kind, uint64_t h)
{
	if (0 == kind) return (uint32_t) (h & 3u);
	uint32_t const u = (uint32_t) (h % 1000u);
	uint64_t const hi = h >> 32;
	if (u < 900u) return (uint32_t) (hi & 3u);
	return 4u + (uint32_t) (hi % 12u);
}

// Alignment storage in HBM: column-major, 8 >> bsh bits per dense symbol code (bsh = 2 when sigma <= 4,
// 1 when sigma <= 16, else 0); row r of a column sits in byte r >> bsh at bit (r & (2^bsh - 1)) * (8 >> bsh).
__host__ __device__ inline uint32_t sym_bytes(uint32_t m, uint32_t bsh) { return (m + (1u << bsh) - 1u) >> bsh; }

// 2-bit digit `pass` of the code of row `a` in a (staged) packed column
__device__ __forceinline__ uint32_t sym_digit(uint8_t const *col, uint32_t a, uint32_t bsh, uint32_t pass)
{
	uint32_t const byte = col[a >> bsh];
	uint32_t const sh = (a & ((1u << bsh) - 1u)) * (8u >> bsh) + 2u * pass;
	return (byte >> sh) & 3u;
}

// one thread per 32-bit word of one column = 4 << bsh consecutive rows (coalesced stores down the column);
// blockIdx.y strides over the columns (a launch holds fewer than 2^32 threads per dimension)
__global__ __launch_bounds__(256) void k_synth(SynthArgs const A, uint8_t *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t bsh)
{
	uint32_t const words_per_col = (uint32_t) (ld / 4);
	uint32_t const w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= words_per_col) return;
	for (uint64_t c = blockIdx.y; c < n; c += gridDim.y)
	{
		uint32_t const spw = 4u << bsh, bits = 8u >> bsh;
		uint32_t const r0 = w * spw;
		uint64_t const b = c / A.block_len;
		uint32_t packed = 0;
		for (uint32_t q = 0; q < spw; ++q)
		{
			uint32_t const r = r0 + q;
			uint32_t code = 0;
			if (r < m)
			{
				uint64_t const f = synth_h(A.seed, 2, r, b) % A.n_founders;
				uint32_t sym = synth_pick(A.kind, synth_h(A.seed, 1, f, c));
				if (synth_h(A.seed, 3, r, c) < A.mut_threshold)
					sym = (sym + 1u + (uint32_t) (synth_h(A.seed, 4, r, c) % (A.sigma - 1u))) % A.sigma;
				code = A.code_of_sym[sym];
			}
			packed |= code << (bits * q);
		}
		*reinterpret_cast<uint32_t *>(msa + c * ld + (size_t) w * 4u) = packed;
	}
}

// ------------------------------------------------------------------------------------------------
// Input side: raw sequence bytes -> dense codes, row-major -> column-major.
// consecutive_alphabet_as_builder (generate_context.cc:135-147): which byte values occur ...
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_presence(uint8_t const *__restrict__ raw, size_t total, uint32_t *__restrict__ present /* 8 words */)
{
	__shared__ uint32_t bm[8];
	if (threadIdx.x < 8) bm[threadIdx.x] = 0;
	__syncthreads();
	uint32_t loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	size_t const stride = (size_t) gridDim.x * blockDim.x * 16;
	for (size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 16; i < total; i += stride)
	{
		if (i + 16 <= total)
		{
			uint4 const v = *reinterpret_cast<uint4 const *>(raw + i);
			uint32_t const w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
			for (int q = 0; q < 4; ++q)
#pragma unroll
				for (int b = 0; b < 4; ++b)
				{
					uint32_t const c = (w[q] >> (8 * b)) & 255u;
#pragma unroll
					for (int k = 0; k < 8; ++k) loc[k] |= ((c >> 5) == (uint32_t) k) ? (1u << (c & 31u)) : 0u;
				}
		}
		else
			for (size_t j = i; j < total; ++j)
			{
				uint32_t const c = raw[j];
#pragma unroll
				for (int k = 0; k < 8; ++k) loc[k] |= ((c >> 5) == (uint32_t) k) ? (1u << (c & 31u)) : 0u;
			}
	}
#pragma unroll
	for (int k = 0; k < 8; ++k)
		if (loc[k]) atomicOr(&bm[k], loc[k]);
	__syncthreads();
	if (threadIdx.x < 8 && bm[threadIdx.x]) atomicOr(&present[threadIdx.x], bm[threadIdx.x]);
}

struct CodeTable { uint8_t code_of[256]; };

// ... and the encode + transpose + pack: raw[r * n + c] -> code_of[byte] at row r of packed column c; 64 x 64
// tiles through LDS, reads coalesced along a row, writes down a column (64 >> bsh bytes per tile column).
__global__ __launch_bounds__(256) void k_encode_transpose(
	uint8_t const *__restrict__ raw, CodeTable const tab, uint32_t m, uint64_t n, uint8_t *__restrict__ msa, size_t ld, uint32_t bsh)
{
	__shared__ uint8_t tile[64][65];
	uint64_t const c0 = (uint64_t) blockIdx.x * 64;
	uint32_t const r0 = blockIdx.y * 64;
	uint32_t const tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
	for (uint32_t rr = ty; rr < 64; rr += 4)
	{
		uint32_t const r = r0 + rr;
		uint64_t const c = c0 + tx;
		tile[rr][tx] = (r < m && c < n) ? tab.code_of[raw[(size_t) r * n + c]] : (uint8_t) 0;
	}
	__syncthreads();
	uint32_t const spb = 1u << bsh, bits = 8u >> bsh;
	for (uint32_t cc = ty; cc < 64; cc += 4)
	{
		uint64_t const c = c0 + cc;
		size_t const byte = (size_t) (r0 >> bsh) + tx;
		if (c < n && tx < (64u >> bsh) && byte < ld)
		{
			uint32_t v = 0;
			for (uint32_t q = 0; q < spb; ++q) v |= (uint32_t) tile[tx * spb + q][cc] << (bits * q);
			msa[c * ld + byte] = (uint8_t) v;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Shared pieces of the column kernels
// ------------------------------------------------------------------------------------------------
// PK ("packed") kernels keep row ids -- and, where the values allow it, divergences -- as 16-bit
// LDS words: m <= 65535 rows then fit the 160 KiB of a CU (BASELINE config C5: m = 10,000).
template <int T, int E, typename PA, typename PD>
__device__ __forceinline__ void read_chunk(PA const *a_l, PD const *d_l, uint32_t (&a)[E], uint32_t (&d)[E])
{
	uint32_t const p0 = threadIdx.x * E;
#pragma unroll
	for (int e = 0; e < E; ++e) { a[e] = a_l[p0 + e]; d[e] = d_l[p0 + e]; }
}

// MODE_RANK: start from the identity at column k0 = blockIdx.x * B, emit block ranks (phase A).
// MODE_SNAP: workgroup j starts from a stored exact state -- a block boundary state of phase B or one of the
//            states phase C drops every snap_stride columns (task_src[j]: bit 63 = the latter) -- sweeps
//            forward and emits (a,d) at every requested column task_rb[first .. first+count) on the way
//            (task_grp[j] = {first, count}, task_rb ascending).
enum { MODE_RANK = 0, MODE_SNAP = 1 };

template <int T, int E, int SIGMA, int MODE, bool PK>
__host__ __device__ inline size_t colblock_lds_bytes()
{
	constexpr size_t CAP = (size_t) T * E;
	return carve_bytes(CAP, PK ? 2 : 4) + carve_bytes(CAP, (PK && MODE == 0) ? 2 : 4) + 2 * carve_bytes(CAP, 1)
	     + carve_bytes(1, sizeof(StepScratch<T, SIGMA>)) + carve_bytes(T / WAVE + 1, 4);
}

// A column with alphabet size sigma is npass = ceil(log2(sigma) / 2) stable partitions by 2-bit digits,
// least significant first, every pass with the same first-in-bucket value k+1: rows whose
// predecessor differs in a lower digit already carry k+1 (the largest value there is) through the
// range maximum of the later passes, so the result equals the single sigma-bucket partition.
template <int T, int E, int SIGMA, int MODE, bool PK>
__global__ __launch_bounds__(T) void k_colblock(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblocks, uint32_t npass, uint32_t bsh,
	// MODE_RANK outputs
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys,
	// MODE_SNAP inputs / outputs
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint64_t const *__restrict__ task_rb, uint2 const *__restrict__ task_grp,
	uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d,
	uint64_t const *__restrict__ task_src, uint32_t snap_stride, uint32_t const *__restrict__ ss_a, uint32_t const *__restrict__ ss_d)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	// MODE_RANK keeps divergences relative to the block start (0..B: 16 bits when PK)
	using AT = std::conditional_t<PK, uint16_t, uint32_t>;
	using DT = std::conditional_t<(PK && MODE == MODE_RANK), uint16_t, uint32_t>;
	Carver cv{smem};
	AT *a_l = cv.take<AT>(CAP);
	DT *d_l = cv.take<DT>(CAP);
	uint8_t *sym0 = cv.take<uint8_t>(CAP);
	uint8_t *sym1 = cv.take<uint8_t>(CAP);
	StepScratch<T, SIGMA> &scr = *cv.take<StepScratch<T, SIGMA>>(1);
	uint32_t *sscr = cv.take<uint32_t>(T / WAVE + 1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint64_t k0, kend;
	uint32_t t_first = 0, t_count = 0, t_next = 0;
	if (MODE == MODE_RANK)
	{
		k0 = (uint64_t) blockIdx.x * B;
		kend = (k0 + B < n) ? k0 + B : n;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			a_l[pos] = (AT) (pos < m ? pos : 0u);
			d_l[pos] = 0;
		}
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
		uint32_t const *sd = (from_stride ? ss_d : bstate_d) + sidx * (size_t) m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			a_l[idx] = (AT) (idx < m ? sa[idx] : 0u);
			d_l[idx] = (DT) (idx < m ? sd[idx] : 0u);
		}
	}
	uint32_t const nb = (uint32_t) (kend - k0);
	uint32_t const dbase = (MODE == MODE_RANK) ? (uint32_t) k0 : 0u;

	bool const has_chunk = tid * 16u < sym_bytes(m, bsh);
	uint4 nxt = make_uint4(0, 0, 0, 0);
	if (nb && has_chunk)
	{
		nxt = *reinterpret_cast<uint4 const *>(msa + k0 * ld + tid * 16u);
		*reinterpret_cast<uint4 *>(sym0 + tid * 16u) = nxt;
	}
	__syncthreads();

	// MODE_SNAP: the state in LDS is (a_k, d_k) for k = k0 + (columns done); copy it out when k is requested
	auto snapshot_if_requested = [&](uint64_t k) {
		if (MODE == MODE_SNAP && t_next < t_count && task_rb[t_first + t_next] == k)
		{
			size_t const ob = (size_t) (t_first + t_next) * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t const idx = tid + i * T;
				if (idx < m) { snap_a[ob + idx] = a_l[idx]; snap_d[ob + idx] = d_l[idx]; }
			}
			++t_next;
		}
	};
	snapshot_if_requested(k0);

	for (uint32_t j = 0; j < nb; ++j)
	{
		uint8_t const *symc = (j & 1u) ? sym1 : sym0;
		uint8_t *symn = (j & 1u) ? sym0 : sym1;
		bool const more = j + 1 < nb;
		if (more && has_chunk)
			nxt = *reinterpret_cast<uint4 const *>(msa + (k0 + j + 1) * ld + tid * 16u);

		for (uint32_t pass = 0; pass < npass; ++pass)
		{
			uint32_t a[E], d[E], s[E], dst[E], dnew[E];
			read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
			for (int e = 0; e < E; ++e) s[e] = (p0 + e < m) ? sym_digit(symc, a[e], bsh, pass) : (uint32_t) SIGMA;

			partition_step<T, E, SIGMA>(d, s, (uint32_t) (k0 + j + 1) - dbase, scr, dst, dnew);

#pragma unroll
			for (int e = 0; e < E; ++e)
				if (p0 + e < m) { a_l[dst[e]] = (AT) a[e]; d_l[dst[e]] = (DT) dnew[e]; }
			if (pass + 1 == npass && more && has_chunk)
				*reinterpret_cast<uint4 *>(symn + tid * 16u) = nxt;
			__syncthreads();
		}
		snapshot_if_requested(k0 + j + 1);
	}

	if (MODE == MODE_RANK)
	{
		// rank of a row = number of bucket starts at or before its position, minus one; a position
		// starts a new block key iff its in-block divergence is > k0 (position 0 always is).
		uint32_t a[E], d[E];
		read_chunk<T, E>(a_l, d_l, a, d);
		uint32_t nf = 0;
#pragma unroll
		for (int e = 0; e < E; ++e) nf += (p0 + e < m && (p0 + e == 0 || d[e] > 0u)) ? 1u : 0u;
		uint32_t total;
		uint32_t r = block_excl_add<T>(nf, sscr, &total);
		size_t const ob = (size_t) blockIdx.x * m;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			if (pos < m)
			{
				bool const first = (pos == 0 || d[e] > 0u);
				r += first ? 1u : 0u;
				rank[ob + a[e]] = r - 1u;
				if (first) keyd[ob + r - 1u] = d[e] + dbase;
			}
		}
		if (tid == 0) nkeys[blockIdx.x] = total;
	}
}

// ------------------------------------------------------------------------------------------------
// Phase B: the serial chain over blocks (one workgroup).
// ------------------------------------------------------------------------------------------------
template <int T, int E, bool PK>
__host__ __device__ inline size_t chain_lds_bytes()
{
	constexpr size_t CAP = (size_t) T * E;
	return 2 * carve_bytes(CAP, PK ? 2 : 4) + 2 * carve_bytes(CAP, 4) + carve_bytes(1, sizeof(StepScratch<T, 4>)) + carve_bytes(T / WAVE + 1, 4);
}

__device__ __forceinline__ uint32_t rank_digits(uint32_t D)
{
	if (D <= 1) return 0;
	uint32_t const bits = 32u - (uint32_t) __builtin_clz(D - 1u);
	return (bits + 1u) / 2u;
}

// One workgroup = one chain over the consecutive key blocks [b0, b1), b0 = blockIdx.x * G.
// A "key block" is described by rank[b][row] (dense co-lex rank of the row's key), keyd[b][r]
// (divergence in front of key r) and nkeys[b].  The chain applies, block after block, the stable
// sort of the running order by block rank (2-bit LSD digit passes of partition_step; divergences
// ride along; rows that start a new key take keyd).
//   start:  start_a/start_d == nullptr : identity order, d = b0 * cols_per_block   (composition from scratch)
//           else                        : state [blockIdx.x] of start_a/start_d     (exact state at the chain's first column)
//   out_state_*: if set, the exact state in front of every block b (and behind the last block of
//           the whole sequence, index nb_total)
//   out_rank/out_keyd/out_nkeys: if set, the chain's composite key block [blockIdx.x]
// Used three times: compose super-blocks (parallel), chain the super-blocks (one
// workgroup), expand every super-block back to block boundaries (parallel).
template <int T, int E, bool PK>
__global__ __launch_bounds__(T) void k_chain(
	uint32_t const *__restrict__ rank, uint32_t const *__restrict__ keyd, uint32_t const *__restrict__ nkeys,
	uint32_t m, uint32_t nb_total, uint32_t G, uint64_t cols_per_block,
	uint32_t const *__restrict__ start_a, uint32_t const *__restrict__ start_d,
	uint32_t *__restrict__ out_state_a, uint32_t *__restrict__ out_state_d,
	uint32_t *__restrict__ out_rank, uint32_t *__restrict__ out_keyd, uint32_t *__restrict__ out_nkeys)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	using AT = std::conditional_t<PK, uint16_t, uint32_t>;     // row ids and ranks are < m
	Carver cv{smem};
	AT *a_l = cv.take<AT>(CAP);
	AT *rk = cv.take<AT>(CAP);
	uint32_t *d_l = cv.take<uint32_t>(CAP);
	uint32_t *kd = cv.take<uint32_t>(CAP);
	StepScratch<T, 4> &scr = *cv.take<StepScratch<T, 4>>(1);
	uint32_t *sscr = cv.take<uint32_t>(T / WAVE + 1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint32_t const b0 = blockIdx.x * G;
	uint32_t const b1 = min(nb_total, b0 + G);
	uint32_t const kstart = (uint32_t) ((uint64_t) b0 * cols_per_block);
	{
		size_t const sb = (size_t) blockIdx.x * m, rb = (size_t) b0 * m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			bool const in = idx < m;
			a_l[idx] = (AT) (in ? (start_a ? start_a[sb + idx] : idx) : 0u);
			d_l[idx] = in ? (start_d ? start_d[sb + idx] : kstart) : 0u;
			rk[idx] = (AT) (in ? rank[rb + idx] : 0u);
			kd[idx] = in ? keyd[rb + idx] : 0u;
		}
	}
	uint32_t D_cur = nkeys[b0];
	__syncthreads();

	for (uint32_t b = b0; b < b1; ++b)
	{
		// (1) publish the exact state in front of block b
		if (out_state_a)
		{
			size_t const ob = (size_t) b * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t const idx = tid + i * T;
				if (idx < m) { out_state_a[ob + idx] = a_l[idx]; out_state_d[ob + idx] = d_l[idx]; }
			}
		}
		// (2) prefetch the next block's rank / keyd into registers
		uint32_t pr[E], pk[E], D_next = 0;
		bool const more = b + 1 < b1;
		if (more)
		{
			size_t const nbase = (size_t) (b + 1) * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t const idx = tid + i * T;
				pr[i] = idx < m ? rank[nbase + idx] : 0u;
				pk[i] = idx < m ? keyd[nbase + idx] : 0u;
			}
			D_next = nkeys[b + 1];
		}
		__syncthreads();

		// (3) LSD 2-bit digit passes over the block rank
		uint32_t const nd = rank_digits(D_cur);
		for (uint32_t p = 0; p < nd; ++p)
		{
			uint32_t a[E], d[E], s[E], dst[E], dnew[E];
			read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
			for (int e = 0; e < E; ++e)
				s[e] = (p0 + e < m) ? (((uint32_t) rk[a[e]] >> (2u * p)) & 3u) : 4u;
			partition_step<T, E, 4>(d, s, 0u, scr, dst, dnew);
#pragma unroll
			for (int e = 0; e < E; ++e)
				if (p0 + e < m) { a_l[dst[e]] = (AT) a[e]; d_l[dst[e]] = dnew[e]; }
			__syncthreads();
		}

		// (4) rows that start a new block key take the in-block divergence of that key
		{
			uint32_t rprev = (p0 > 0 && p0 - 1 < m) ? rk[a_l[p0 - 1]] : PAD_KEY;
#pragma unroll
			for (int e = 0; e < E; ++e)
			{
				uint32_t const pos = p0 + e;
				if (pos < m)
				{
					uint32_t const r = rk[a_l[pos]];
					if (pos == 0 || r != rprev) d_l[pos] = kd[r];
					rprev = r;
				}
			}
		}
		// (5) land the prefetch (rank / keyd of block b are dead once every thread is past (4))
		__syncthreads();
		if (more)
		{
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t const idx = tid + i * T;
				rk[idx] = (AT) pr[i];
				kd[idx] = pk[i];
			}
		}
		D_cur = D_next;
		__syncthreads();
	}

	if (out_state_a && b1 == nb_total)
	{
		size_t const ob = (size_t) nb_total * m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			if (idx < m) { out_state_a[ob + idx] = a_l[idx]; out_state_d[ob + idx] = d_l[idx]; }
		}
	}
	if (out_rank)
	{
		// composite key block of the chain: a row starts a new composite key iff its divergence is
		// inside the chain's column range (> kstart); same epilogue as k_colblock<MODE_RANK>
		uint32_t a[E], d[E];
		read_chunk<T, E>(a_l, d_l, a, d);
		uint32_t nf = 0;
#pragma unroll
		for (int e = 0; e < E; ++e) nf += (p0 + e < m && (p0 + e == 0 || d[e] > kstart)) ? 1u : 0u;
		uint32_t total;
		uint32_t r = block_excl_add<T>(nf, sscr, &total);
		size_t const ob = (size_t) blockIdx.x * m;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			if (pos < m)
			{
				bool const first = (pos == 0 || d[e] > kstart);
				r += first ? 1u : 0u;
				out_rank[ob + a[e]] = r - 1u;
				if (first) out_keyd[ob + r - 1u] = d[e];
			}
		}
		if (tid == 0) out_nkeys[blockIdx.x] = total;
	}
}

// ------------------------------------------------------------------------------------------------
// Phase C: true per-column updates inside a block + per-column top of the divergence histogram.
// Divergence values are replaced by order-preserving ids (sorted distinct boundary values first,
// then one new id per column), so the histogram is a dense LDS table updated incrementally.
// ------------------------------------------------------------------------------------------------
template <int T, int E, int SIGMA, bool PK>
__host__ __device__ inline size_t columns_lds_bytes(uint32_t B)
{
	constexpr size_t CAP = (size_t) T * E;
	return 2 * carve_bytes(CAP, PK ? 2 : 4) + 2 * carve_bytes(CAP, 1) + carve_bytes(PK ? (CAP + B + 1) / 2 : CAP + B, 4) + carve_bytes(CAP, 4)
	     + carve_bytes(1, sizeof(StepScratch<T, SIGMA>)) + carve_bytes(T / WAVE + 1, 4);
}

// divergence-value histogram: one counter per id; PK: two 16-bit counters per word, updated with
// 32-bit LDS atomics (word = hi * 65536 + lo holds exactly once every update of a column has landed:
// the counts themselves are in [0, m], m <= 65535)
template <bool PK> __device__ __forceinline__ uint32_t cnt_get(uint32_t const *cnt_l, uint32_t i)
{
	return PK ? ((cnt_l[i >> 1] >> ((i & 1u) * 16u)) & 0xFFFFu) : cnt_l[i];
}
template <bool PK> __device__ __forceinline__ void cnt_inc(uint32_t *cnt_l, uint32_t i)
{
	if (PK) atomicAdd(&cnt_l[i >> 1], 1u << ((i & 1u) * 16u)); else atomicAdd(&cnt_l[i], 1u);
}
template <bool PK> __device__ __forceinline__ void cnt_dec(uint32_t *cnt_l, uint32_t i)
{
	if (PK) atomicSub(&cnt_l[i >> 1], 1u << ((i & 1u) * 16u)); else atomicSub(&cnt_l[i], 1u);
}

template <int T>
__device__ __forceinline__ void bitonic_sort_lds(uint32_t *sb, uint32_t N2)
{
	for (uint32_t k = 2; k <= N2; k <<= 1)
	{
		for (uint32_t j = k >> 1; j > 0; j >>= 1)
		{
			for (uint32_t i = threadIdx.x; i < N2 / 2; i += T)
			{
				uint32_t const lo = ((i / j) * 2u * j) + (i % j);
				uint32_t const hi = lo + j;
				bool const asc = (lo & k) == 0;
				uint32_t const x = sb[lo], y = sb[hi];
				if ((x > y) == asc) { sb[lo] = y; sb[hi] = x; }
			}
			__syncthreads();
		}
	}
}

// header of a per-column list: {n_entries, cnt0, complete, cum}
template <int T, int E, int SIGMA, bool PK>
__global__ __launch_bounds__(T) void k_columns(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t N2,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint32_t L, uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr, uint32_t npass, uint32_t bsh,
	uint32_t snap_stride, uint32_t *__restrict__ ss_a, uint32_t *__restrict__ ss_d)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	// [a_l][d_l][sym0][sym1][cnt_l] are contiguous: the prologue's sort buffer (N2 < 2m words) overlays them
	using AT = std::conditional_t<PK, uint16_t, uint32_t>;     // row ids < m, value ids < m + B
	Carver cv{smem};
	AT *a_l = cv.take<AT>(CAP);
	AT *d_l = cv.take<AT>(CAP);
	uint8_t *sym0 = cv.take<uint8_t>(CAP);
	uint8_t *sym1 = cv.take<uint8_t>(CAP);
	uint32_t *cnt_l = cv.take<uint32_t>(PK ? (CAP + B + 1) / 2 : CAP + B);
	uint32_t *V_l = cv.take<uint32_t>(CAP);
	StepScratch<T, SIGMA> &scr = *cv.take<StepScratch<T, SIGMA>>(1);
	uint32_t *sscr = cv.take<uint32_t>(T / WAVE + 1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint64_t const k0 = (uint64_t) blockIdx.x * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	uint32_t const nb = (uint32_t) (kend - k0);

	// ---- prologue
	// boundary state straight into registers (chunk ownership: positions tid*E .. tid*E+E-1)
	uint32_t a[E], d[E];
	{
		size_t const ob = (size_t) blockIdx.x * m;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			a[e] = pos < m ? bstate_a[ob + pos] : 0u;
			d[e] = pos < m ? bstate_d[ob + pos] : 0u;
		}
	}

	// ---- sorted distinct divergence values -> V_l[0..D0)
	uint32_t *sb = reinterpret_cast<uint32_t *>(a_l);   // N2 < 2m words, overlays a_l .. cnt_l
	for (uint32_t i = tid; i < N2; i += T) sb[i] = PAD_KEY;
	__syncthreads();
#pragma unroll
	for (int e = 0; e < E; ++e)
		if (p0 + e < m) sb[p0 + e] = d[e];
	__syncthreads();
	bitonic_sort_lds<T>(sb, N2);
	uint32_t D0;
	{
		uint32_t const per = (N2 + T - 1) / T;
		uint32_t const start = tid * per;
		uint32_t nflag = 0;
		for (uint32_t q = 0; q < per; ++q)
		{
			uint32_t const i = start + q;
			if (i < N2)
			{
				uint32_t const v = sb[i];
				nflag += (v != PAD_KEY && (i == 0 || v != sb[i - 1])) ? 1u : 0u;
			}
		}
		uint32_t w = block_excl_add<T>(nflag, sscr, &D0);
		for (uint32_t q = 0; q < per; ++q)
		{
			uint32_t const i = start + q;
			if (i < N2)
			{
				uint32_t const v = sb[i];
				if (v != PAD_KEY && (i == 0 || v != sb[i - 1])) V_l[w++] = v;
			}
		}
	}
	__syncthreads();                          // the sort buffer (overlaying cnt_l) is dead from here
	for (uint32_t i = tid; i < (PK ? (D0 + nb + 1u) / 2u : D0 + nb); i += T) cnt_l[i] = 0;
	__syncthreads();

	// ---- ids + initial histogram
	uint32_t id[E];
#pragma unroll
	for (int e = 0; e < E; ++e)
	{
		id[e] = 0;
		if (p0 + e < m)
		{
			uint32_t lo = 0, hi = D0;
			uint32_t const key = d[e];
			while (lo < hi)
			{
				uint32_t const mid = (lo + hi) >> 1;
				if (V_l[mid] < key) lo = mid + 1; else hi = mid;
			}
			id[e] = lo;
			cnt_inc<PK>(cnt_l, lo);
		}
	}
#pragma unroll
	for (int e = 0; e < E; ++e) { a_l[p0 + e] = (AT) a[e]; d_l[p0 + e] = (AT) id[e]; }

	bool const has_chunk = tid * 16u < sym_bytes(m, bsh);
	uint4 nxt = make_uint4(0, 0, 0, 0);
	if (nb && has_chunk)
	{
		nxt = *reinterpret_cast<uint4 const *>(msa + k0 * ld + tid * 16u);
		*reinterpret_cast<uint4 *>(sym0 + tid * 16u) = nxt;
	}
	__syncthreads();

	bool const zero_present = (V_l[0] == 0u);

	for (uint32_t j = 0; j < nb; ++j)
	{
		uint8_t const *symc = (j & 1u) ? sym1 : sym0;
		uint8_t *symn = (j & 1u) ? sym0 : sym1;
		bool const more = j + 1 < nb;
		if (more && has_chunk)
			nxt = *reinterpret_cast<uint4 const *>(msa + (k0 + j + 1) * ld + tid * 16u);

		for (uint32_t pass = 0; pass < npass; ++pass)
		{
			uint32_t s[E], dst[E], dnew[E];
			read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
			for (int e = 0; e < E; ++e) s[e] = (p0 + e < m) ? sym_digit(symc, a[e], bsh, pass) : (uint32_t) SIGMA;

			partition_step<T, E, SIGMA>(d, s, D0 + j, scr, dst, dnew);

#pragma unroll
			for (int e = 0; e < E; ++e)
			{
				if (p0 + e < m)
				{
					a_l[dst[e]] = (AT) a[e];
					d_l[dst[e]] = (AT) dnew[e];
					if (dnew[e] != d[e])
					{
						cnt_dec<PK>(cnt_l, d[e]);
						cnt_inc<PK>(cnt_l, dnew[e]);
					}
				}
			}
			if (pass + 1 == npass && more && has_chunk)
				*reinterpret_cast<uint4 *>(symn + tid * 16u) = nxt;
			__syncthreads();
		}

		// ---- every snap_stride columns: drop the exact (a, d) for pass 2 (ids back to divergence values)
		if (ss_a && (k0 + j + 1) % snap_stride == 0)
		{
			size_t const ob = (size_t) ((k0 + j + 1) / snap_stride) * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t const idx = tid + i * T;
				if (idx < m)
				{
					uint32_t const vid = d_l[idx];
					ss_a[ob + idx] = a_l[idx];
					ss_d[ob + idx] = vid < D0 ? V_l[vid] : (uint32_t) (k0 + (vid - D0) + 1u);
				}
			}
		}

		// ---- emit the top of the histogram for column k0+j (wave 0; the others run ahead into
		// the next column and meet it again at the partition step's barrier).
		// Entry 0 lumps every value >= thr = end+1-L (end = k+1): the DP clips all of them to the same
		// cut bound (lp.cc:444-445), so only their total count matters.  Then the distinct values
		// below thr, descending, until their cumulative count exceeds X (every DP cell is >= the lump's
		// count, so the list always reaches X counts past the smallest value the cell can take).
		if (wave_id() == 0)
		{
			uint32_t const lane = lane_id();
			uint64_t const k = k0 + j;
			uint32_t const thr = (k + 2 > (uint64_t) L) ? (uint32_t) (k + 2 - L) : 0u;
			uint2 *out = ent + k * (size_t) stride;
			int32_t const top = (int32_t) (D0 + j);
			uint32_t cumN = 0, nent = 1, R = 0;       // cumN: count of the values below thr taken so far
			for (int32_t base = top; base >= 0; base -= 64)
			{
				int32_t const i = base - (int32_t) lane;
				uint32_t const c = (i >= 0) ? cnt_get<PK>(cnt_l, (uint32_t) i) : 0u;
				uint32_t const v = (i < 0) ? 0u : (((uint32_t) i < D0) ? V_l[i] : (uint32_t) (k0 + ((uint32_t) i - D0) + 1u));
				bool const nz = c > 0;
				bool const rec = nz && v >= thr;
				uint32_t const inc = wave_incl_add(c);
				uint64_t const rmask = __ballot(rec);       // values descend with the lane: a prefix of the non-zero lanes
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
				out[0] = make_uint2((uint32_t) (k + 1), R);
				hdr[k] = make_uint4(nent, zero_present ? cnt_get<PK>(cnt_l, 0u) : 0u, cum == m ? 1u : 0u, cum);
			}
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Phase D: the DP.  One workgroup; waves take columns; rounds of <= L columns are mutually
// independent (the step at column end reads DP entries <= end-2L and writes end-L:
// segmentation_lp_context.cc:444-445,463-465,135-137).
// rmq.hh semantics (block 64) via P (first-min of the block prefix), S (first-min of the block
// suffix) and the sparse table Tb with the smp1 == smp2 quirk (rmq.hh:76-79).
// ------------------------------------------------------------------------------------------------
// K[t]: 64-bit mask over the 64-block of t; bit p (p <= t mod 64) is set iff M[p] <= min(M[p+1..t])
// (the monotonic stack after scanning the block up to t, popping only strictly greater keys).
// The first minimum of [b, t] inside one block (std::min_element, rmq.hh:116) is the lowest set
// bit of K[t] at or above b -- every partial-block scan of rmq.hh in O(1).
// Tb[p][j] / Tbv[p][j]: sparse-table sample (index / key) of rmq.hh's m_precalc[p][j].
struct DpArrays {
	uint32_t *M, *LB, *SZ, *Tb, *Tbv;
	unsigned long long *K;
	uint32_t tstride;
};

// One workgroup of 16 waves, specialised: 14 compute waves that touch LDS only, one loader wave
// that streams the per-column lists (and the old sparse-table samples the update needs) into LDS
// two rounds ahead with LDS-DMA, one writer wave that flushes finished rounds to HBM.  A global
// memory round trip costs ~1.5 us here, a round must cost about that in total, so no wave that
// the round barriers wait for may ever wait on HBM.
constexpr uint32_t DPW = 4096;            // DP entries mirrored in LDS (ring, slot = index mod DPW)
constexpr uint32_t DP_RL = 56;            // cells per round (<= L)
constexpr uint32_t DP_NWC = 14;           // compute waves
constexpr int      DP_G = 4;              // cells per compute wave per round (14 * 4 = 56)
constexpr uint32_t DP_STG = 512;          // staging ring for LB / SZ
constexpr uint32_t DP_TRN = 64;           // sparse-table ring: last 64 samples of every level
constexpr uint32_t DP_LEVELS = 32;
constexpr uint32_t DP_HPMIN = 7;          // levels >= 7: input sample is older than the ring -> mailbox
constexpr uint32_t DP_SLOTS = 3;          // list slots: rounds r, r+1, r+2
constexpr uint32_t DP_MBSLOTS = 4;        // mailbox slots (read one round later in pipelined mode)
constexpr uint32_t DP_Q = DP_RL / 2 + 1 + 2;   // LDS-DMA instructions the loader issues per round
constexpr uint32_t DP_LOADER = 14, DP_WRITER = 15;

// LDS pointers carry their address space so that a choice between an LDS and an HBM source stays
// two different loads (ds_read vs global_load) instead of one flat load through a selected pointer
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) uint16_t lds_u16;

struct DpLds {
	lds_u64 *Kr;
	lds_u32 *Mr, *LBr, *SZr, *Tr, *Trv;
	lds_u32 *LS;                          // [DP_SLOTS][DP_RL][64] x {value, count}
	lds_u32 *H;                           // [DP_SLOTS][64] x {n_entries, cnt0, complete, cum}
	lds_u32 *MBi, *MBv;                   // [DP_SLOTS][64]
};

__host__ __device__ inline size_t dp_lds_bytes()
{
	return carve_bytes(DPW, 8) + carve_bytes(DPW, 4) + 2 * carve_bytes(DP_STG, 4) + 2 * carve_bytes((size_t) DP_LEVELS * DP_TRN, 4)
	     + carve_bytes((size_t) DP_SLOTS * DP_RL * 64, 8) + carve_bytes((size_t) DP_SLOTS * 64, 16) + 2 * carve_bytes((size_t) DP_MBSLOTS * 64, 4)
	     + carve_bytes(4, 4);
}

// where a query may read: entries >= safe_lo and samples produced after block cb - DP_TRN live in LDS
struct DpView {
	uint32_t safe_lo;     // first DP entry guaranteed to be in the LDS ring during this round
	uint32_t cb;          // complete (indexed) blocks at the start of this round
};

__device__ __forceinline__ uint32_t dp_key(DpArrays const &A, DpLds const &D, DpView const &V, uint32_t i)
{
	return i >= V.safe_lo ? D.Mr[i & (DPW - 1u)] : A.M[i];
}

__device__ __forceinline__ unsigned long long dp_mask(DpArrays const &A, DpLds const &D, DpView const &V, uint32_t i)
{
	return i >= V.safe_lo ? D.Kr[i & (DPW - 1u)] : A.K[i];
}

// sample (p, j) was produced when block j + 2^p - 1 completed
__device__ __forceinline__ void dp_sample(DpArrays const &A, DpLds const &D, DpView const &V, uint32_t p, uint32_t j, uint32_t *idx, uint32_t *val)
{
	if (j + (1u << p) + DP_TRN > V.cb + 2u)
	{
		*idx = D.Tr[p * DP_TRN + (j & (DP_TRN - 1u))];
		*val = D.Trv[p * DP_TRN + (j & (DP_TRN - 1u))];
	}
	else
	{
		*idx = A.Tb[(size_t) p * A.tstride + j];
		*val = A.Tbv[(size_t) p * A.tstride + j];
	}
}

// rmq.hh:85-105 (operator()), every query of the DP has end <= number of indexed entries.
// Returns the index; *val = its key.
__device__ __forceinline__ uint32_t rmq_query(DpArrays const &A, DpLds const &D, DpView const &V, uint32_t beg, uint32_t end, uint32_t *val)
{
	uint32_t const bb = beg >> 6, eb = (end - 1u) >> 6;
	uint32_t const beg_block = bb + 1u, end_block = end >> 6;
	uint32_t const ie = end - 1u;
	unsigned long long const ke = dp_mask(A, D, V, ie);
	if (bb == eb)
	{
		// beg_block >= end_block, one block: naive_min (rmq.hh:90-91)
		uint32_t const idx = beg + (uint32_t) __builtin_ctzll(ke >> (beg & 63u));
		*val = dp_key(A, D, V, idx);
		return idx;
	}
	unsigned long long const kb = dp_mask(A, D, V, (bb << 6) + 63u);
	uint32_t const il = beg + (uint32_t) __builtin_ctzll(kb >> (beg & 63u));   // naive_min(beg, beg_block*64)
	uint32_t const ir = (ie & ~63u) + (uint32_t) __builtin_ctzll(ke);           // naive_min(end_block*64, end)
	uint32_t const m_il = dp_key(A, D, V, il), m_ir = dp_key(A, D, V, ir);
	uint32_t idx, v;
	if (beg_block < end_block)
	{
		uint32_t const pow2 = 31u - (uint32_t) __builtin_clz(end_block - beg_block);
		uint32_t t1, t2, m_t1, m_t2;
		dp_sample(A, D, V, pow2, beg_block, &t1, &m_t1);
		dp_sample(A, D, V, pow2, end_block - (1u << pow2), &t2, &m_t2);
		idx = t1; v = m_t1;
		if (m_t2 < v) { idx = t2; v = m_t2; }                   // rmq.hh:96
		if (m_il < v) { idx = il; v = m_il; }                   // rmq.hh:97-98
		if ((end & 63u) != 0 && m_ir < v) { idx = ir; v = m_ir; }   // rmq.hh:100-104
	}
	else
	{
		// two adjacent blocks: naive_min over [beg, block end) ++ [block start, end), first minimum
		idx = il; v = m_il;
		if (m_ir < v) { idx = ir; v = m_ir; }
	}
	*val = v;
	return idx;
}

// The same query when the whole range (and therefore every sample it needs) is inside the LDS
// ring: beg >= safe_lo implies beg_block >= cb - DP_TRN + 3, so dp_sample would pick LDS anyway.
// 32-bit address math only; this is the path practically every candidate takes.
__device__ __forceinline__ uint32_t rmq_query_lds(DpLds const &D, uint32_t beg, uint32_t end, uint32_t *val)
{
	uint32_t const bb = beg >> 6, eb = (end - 1u) >> 6;
	uint32_t const ie = end - 1u;
	unsigned long long const ke = D.Kr[ie & (DPW - 1u)];
	unsigned long long const kb = D.Kr[((bb << 6) + 63u) & (DPW - 1u)];
	uint32_t const sh = beg & 63u;
	// one block: first min of [beg, end) from the mask of the right end; else the two partial blocks
	uint32_t const il = beg + (uint32_t) __builtin_ctzll((bb == eb ? ke : kb) >> sh);
	uint32_t const ir = (ie & ~63u) + (uint32_t) __builtin_ctzll(ke);
	uint32_t idx = il, v = D.Mr[il & (DPW - 1u)];
	if (bb != eb)
	{
		uint32_t const m_ir = D.Mr[ir & (DPW - 1u)];
		uint32_t const beg_block = bb + 1u, end_block = end >> 6;
		if (beg_block < end_block)
		{
			uint32_t const pow2 = 31u - (uint32_t) __builtin_clz(end_block - beg_block);
			uint32_t const s1 = pow2 * DP_TRN + (beg_block & (DP_TRN - 1u));
			uint32_t const s2 = pow2 * DP_TRN + ((end_block - (1u << pow2)) & (DP_TRN - 1u));
			uint32_t const t1 = D.Tr[s1], m_t1 = D.Trv[s1], t2 = D.Tr[s2], m_t2 = D.Trv[s2];
			uint32_t const m_il = v;
			idx = t1; v = m_t1;
			if (m_t2 < v) { idx = t2; v = m_t2; }                   // rmq.hh:96
			if (m_il < v) { idx = il; v = m_il; }                   // rmq.hh:97-98
			if ((end & 63u) != 0 && m_ir < v) { idx = ir; v = m_ir; }   // rmq.hh:100-104
		}
		else if (m_ir < v) { idx = ir; v = m_ir; }                  // two adjacent blocks, first minimum
	}
	*val = v;
	return idx;
}

// State of one DP cell while a wave evaluates it (calculate_segmentation_lp_dp_arg, lp.cc:393-481,
// candidate ranges visited in descending divergence order and pruned exactly).
struct CellState {
	uint32_t best_v, best_lb, best_sz, cum_base;
};

// One strip of 64 list entries starting at entry s0; lanes < ncand are candidates (the lane's
// range needs the next entry's value).  Returns true when the cell is decided (pruned).
__device__ __forceinline__ bool dp_strip(
	DpArrays const &A, DpLds const &D, DpView const &V, uint2 en_in, uint32_t vnext_in, uint32_t nent, uint32_t s0,
	uint32_t ncand, uint32_t L, uint32_t end, CellState &st)
{
	uint32_t const lane = lane_id();
	uint32_t const i = s0 + lane;
	bool const valid = i < nent;
	bool const have_next = i + 1 < nent && lane < ncand;
	uint2 const en = valid ? en_in : make_uint2(0u, 0u);
	uint32_t const vnext = have_next ? vnext_in : 0u;
	bool const is0 = valid && en.x == 0u;
	uint32_t const cc = (valid && !is0) ? en.y : 0u;
	uint32_t const cum = st.cum_base + wave_incl_add(cc);
	bool ok = valid && !is0 && have_next && vnext != 0u;
	uint32_t lo = vnext;
	uint32_t const c = min(en.x, end + 1u - L);              // lp.cc:444-445 (text_pos + 2 - L)
	if (lo < L)                                              // lp.cc:449-455 (lb == 0)
	{
		if (L < c) lo = L; else ok = false;
	}
	ok = ok && lo < c;                                       // lp.cc:458
	uint32_t val = 0xFFFFFFFFu, idx = 0;
	uint32_t const qb = lo - L, qe = c - L;
	if (__ballot(ok && qb < V.safe_lo) == 0)
	{
		// every candidate of the strip lies inside the LDS ring
		if (ok)
		{
			uint32_t mv;
			idx = rmq_query_lds(D, qb, qe, &mv);                 // lp.cc:465
			val = max(mv, cum);                                  // lp.cc:468-471
		}
	}
	else if (ok)
	{
		uint32_t mv;
		idx = rmq_query(A, D, V, qb, qe, &mv);
		val = max(mv, cum);
	}
	// minimum value; among equal values the candidate the reference visits first = the largest i
	// (lowest divergence value) = the highest lane of the strip
	uint32_t const vmin = wave_min_u32(val);
	if (vmin != 0xFFFFFFFFu && vmin <= st.best_v)
	{
		uint64_t const wmask = __ballot(ok && val == vmin);
		int const src = 63 - (int) __builtin_clzll(wmask);       // wave-uniform: v_readlane, no LDS round trip
		st.best_v = vmin;
		st.best_lb = readlane_u32(idx, src) + L;
		st.best_sz = readlane_u32(cum, src);
	}
	bool const last = s0 + 64u >= nent;                       // strip reaches the end of the list
	st.cum_base = readlane_u32(cum, 63);
	if (!last && ncand < 64u) st.cum_base = readlane_u32(cum, 62);
	return st.best_v != 0xFFFFFFFFu && st.cum_base > st.best_v;
}

// A whole cell, strip after strip (the general path: lists longer than one strip, or ranges that
// reach behind the LDS ring).  Returns the finished state; sets the overflow flag when the list is
// too short to prove the result.
__device__ __forceinline__ CellState dp_cell_sequential(
	DpArrays const &A, DpLds const &D, DpView const &V, uint2 const *__restrict__ list, uint2 en0, uint32_t vnext0,
	uint4 const h, uint32_t m, uint32_t L, uint32_t end, uint32_t *flags)
{
	uint32_t const lane = lane_id();
	CellState st;
	st.best_v = 0xFFFFFFFFu; st.best_lb = 0; st.best_sz = 0; st.cum_base = 0;
	uint32_t const nent = h.x;
	bool done = dp_strip(A, D, V, en0, vnext0, nent, 0, 63, L, end, st);
	for (uint32_t s0 = 63; !done && s0 < nent; s0 += 64)     // continue from HBM (rare; only this wave waits)
	{
		uint2 const e2 = list[s0 + lane];
		uint32_t const v2 = list[s0 + lane + 1u].x;
		done = dp_strip(A, D, V, e2, v2, nent, s0, 64, L, end, st);
	}
	uint32_t const cnt0 = h.y, complete = h.z;
	bool const stopped = st.best_v != 0xFFFFFFFFu && st.cum_base > st.best_v;
	if (!complete && !stopped && lane == 0) atomicOr(flags, 1u);   // list too short to prove the result
	if (complete && cnt0 > 0)
	{
		uint32_t const w = m - cnt0;                         // lp.cc:416-421, visited first by the reference
		if (w <= st.best_v) { st.best_v = w; st.best_lb = 0; st.best_sz = w; }
	}
	if (m <= st.best_v) { st.best_v = m; st.best_lb = 0; st.best_sz = m; }   // initial min_arg, lp.cc:123
	return st;
}

struct DpRound {
	uint32_t e0, len, t0, t1;
	bool final_round;
};

// The two 32-lane halves of a wave as independent scans / reductions
__device__ __forceinline__ uint32_t half_incl_add(uint32_t v)
{
	v += dpp_mov<DPP_ROW_SHR1, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_SHR2, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_SHR4, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_SHR8, 0xF>(0u, v);
	v += dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, v);
	return v;
}

__device__ __forceinline__ uint32_t half_min_u32(uint32_t v)
{
	v = min(v, dpp_mov<DPP_ROW_SHR1, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_SHR2, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_SHR4, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_SHR8, 0xF>(0xFFFFFFFFu, v));
	v = min(v, dpp_mov<DPP_ROW_BCAST15, 0xA>(0xFFFFFFFFu, v));
	uint32_t const a = readlane_u32(v, 31), b = readlane_u32(v, 63);
	return lane_id() < 32u ? a : b;
}

// Two cells at once, one per 32-lane half (cells i0 and i0 + 1 of the round): the arithmetic of
// dp_strip on the first 32 list entries plus the tail of dp_cell_sequential.  Nearly every cell is
// decided there (the lumped entry plus a few dozen distinct values reach the pruning bound), so
// a cell costs half the instructions of a full-wave strip.  A cell that is neither decided nor at the
// end of its list after 32 entries is left unwritten and reported in the returned mask (bit 0 /
// bit 32); the caller runs the general path for it.
__device__ __forceinline__ uint64_t dp_cell_pair(
	DpArrays const &A, DpLds const &D, DpView const &V, uint32_t slot, uint32_t i0, DpRound const &R,
	uint32_t m, uint32_t L, uint32_t *flags)
{
	uint32_t const lane = lane_id(), half = lane >> 5, sub = lane & 31u;
	bool const has = i0 + half < R.len;
	uint32_t const ci = has ? i0 + half : i0;
	uint32_t const end = R.e0 + ci, t = end - L;
	lds_u32 const *hp = D.H + (slot * 64u + ci) * 4u;
	uint32_t const nent = hp[0], cnt0 = hp[1], complete = hp[2];
	lds_u32 const *ls = D.LS + (slot * DP_RL + ci) * 128u;
	bool const valid = sub < nent;
	bool const have_next = sub + 1u < nent;
	uint2 const en = valid ? make_uint2(ls[2u * sub], ls[2u * sub + 1u]) : make_uint2(0u, 0u);
	uint32_t const vnext = have_next ? ls[2u * sub + 2u] : 0u;
	bool const is0 = valid && en.x == 0u;
	uint32_t const cc = (valid && !is0) ? en.y : 0u;
	uint32_t const cum = half_incl_add(cc);
	bool ok = valid && !is0 && have_next && vnext != 0u;
	uint32_t lo = vnext;
	uint32_t const c = min(en.x, end + 1u - L);              // lp.cc:444-445
	if (lo < L)                                              // lp.cc:449-455
	{
		if (L < c) lo = L; else ok = false;
	}
	ok = ok && lo < c;                                       // lp.cc:458
	uint32_t val = 0xFFFFFFFFu, idx = 0;
	uint32_t const qb = lo - L, qe = c - L;
	if (__ballot(ok && qb < V.safe_lo) == 0)
	{
		if (ok)
		{
			uint32_t mv;
			idx = rmq_query_lds(D, qb, qe, &mv);                 // lp.cc:465
			val = max(mv, cum);                                  // lp.cc:468-471
		}
	}
	else if (ok)
	{
		uint32_t mv;
		idx = rmq_query(A, D, V, qb, qe, &mv);
		val = max(mv, cum);
	}
	// minimum per half; among equal values the highest lane of the half (dp_strip)
	uint32_t const vmin = half_min_u32(val);
	uint64_t const wmask = __ballot(ok && val == vmin);
	uint32_t const mA = (uint32_t) wmask, mB = (uint32_t) (wmask >> 32);
	int const srcA = mA ? 31 - (int) __builtin_clz(mA) : 0, srcB = mB ? 63 - (int) __builtin_clz(mB) : 32;
	uint32_t const lbA = readlane_u32(idx, srcA), lbB = readlane_u32(idx, srcB);
	uint32_t const szA = readlane_u32(cum, srcA), szB = readlane_u32(cum, srcB);
	uint32_t const cumA = readlane_u32(cum, 31), cumB = readlane_u32(cum, 63);
	uint32_t best_v = vmin;
	uint32_t best_lb = vmin != 0xFFFFFFFFu ? (half ? lbB : lbA) + L : 0u;
	uint32_t best_sz = vmin != 0xFFFFFFFFu ? (half ? szB : szA) : 0u;
	uint32_t const cum_base = half ? cumB : cumA;
	bool const decided = best_v != 0xFFFFFFFFu && cum_base > best_v;
	bool const more = !decided && nent > 32u;
#ifdef FSEQ_DP_STATS
	{
		// diagnostic build: histogram of the list entries a cell needed before the pruning bound was
		// reached, in flags[128 + n] (n = 33: more than the 32 of a half wave)
		uint64_t const need = __ballot(valid && (cum <= best_v || best_v == 0xFFFFFFFFu));
		uint32_t const nh = (uint32_t) __popc((uint32_t) (need >> (half * 32u)));
		if (sub == 0u && has) atomicAdd(flags + 128 + (more ? 33u : nh), 1u);
	}
#endif
	if (!more)
	{
		if (!complete && !decided && has && sub == 0u) atomicOr(flags, 1u);   // list too short to prove the result
		if (complete && cnt0 > 0)
		{
			uint32_t const w = m - cnt0;                         // lp.cc:416-421
			if (w <= best_v) { best_v = w; best_lb = 0; best_sz = w; }
		}
		if (m <= best_v) { best_v = m; best_lb = 0; best_sz = m; }             // lp.cc:123
		if (has && sub == 0u)
		{
			D.Mr[t & (DPW - 1u)] = best_v;
			D.LBr[t & (DP_STG - 1u)] = best_lb;
			D.SZr[t & (DP_STG - 1u)] = best_sz;
		}
	}
	return __ballot(has && more && sub == 0u);
}

// LDS-DMA: every lane names its own 16 (or 4) global bytes; they land at LDS address lds + lane * size.
// Inline asm on purpose: the loader wave counts these itself (s_waitcnt vmcnt(DP_Q) = "the round
// before the one just issued has landed"); issued through the builtin, hipcc would drain them
// with vmcnt(0) in front of every barrier and LDS read that follows.  M0 carries the LDS base
// and is restored (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma16(void const *g, uint32_t lds_addr)
{
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
// the same without the save / restore, for a run of loads bracketed by m0_save() / m0_restore() with
// nothing else in between (the loader's list loop: two scalar instructions fewer per load)
__device__ __forceinline__ uint32_t m0_save()
{
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0" : "=s"(keep) : : "memory");
	return keep;
}
__device__ __forceinline__ void m0_restore(uint32_t keep)
{
	asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
}
__device__ __forceinline__ void dma16_m0(void const *g, uint32_t lds_addr)
{
	asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4(void const *g, uint32_t lds_addr)
{
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// s_waitcnt vmcnt(N) takes an immediate: wait until at most N (4..31) of this wave's LDS-DMA loads are in flight
__device__ __forceinline__ void dp_wait_all_but(uint32_t n)
{
#define FSEQ_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
	switch (n)
	{
		FSEQ_W(4) FSEQ_W(5) FSEQ_W(6) FSEQ_W(7) FSEQ_W(8) FSEQ_W(9) FSEQ_W(10) FSEQ_W(11) FSEQ_W(12) FSEQ_W(13) FSEQ_W(14)
		FSEQ_W(15) FSEQ_W(16) FSEQ_W(17) FSEQ_W(18) FSEQ_W(19) FSEQ_W(20) FSEQ_W(21) FSEQ_W(22) FSEQ_W(23) FSEQ_W(24)
		FSEQ_W(25) FSEQ_W(26) FSEQ_W(27) FSEQ_W(28) FSEQ_W(29) FSEQ_W(30) FSEQ_W(31)
		default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
	}
#undef FSEQ_W
}

// Round schedule.  nreg regular rounds of <= RL cells (end = L + r*RL + i), then -- pipelined mode
// only -- one empty drain round (the update of the last regular round), then the final cell at
// rb = n (lp.cc:165-183).
struct DpSchedule {
	uint32_t L, n, RL, nreg, nrounds;
	bool pipe;
};

__device__ __forceinline__ DpRound dp_round(DpSchedule const &S, uint32_t r)
{
	DpRound R;
	uint32_t const last_end = S.n - S.L;
	R.final_round = (r + 1u == S.nrounds);
	bool const regular = r < S.nreg;
	R.e0 = R.final_round ? S.n : (regular ? S.L + r * S.RL : last_end + 1u);
	R.len = R.final_round ? 1u : (regular ? min(S.RL, last_end - R.e0 + 1u) : 0u);
	R.t0 = R.e0 - S.L;
	R.t1 = R.t0 + R.len;
	return R;
}

__device__ __forceinline__ void dp_barrier()
{
	// LDS traffic only; outstanding HBM loads / stores of the loader and writer waves stay in flight
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_s_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// rmq.update, part 1 (rmq.hh:61-68 made O(1) to query): one 16-bit slice (positions [16c, 16c+16)
// of the block) of the stack masks of the fresh entries of block blk.
__device__ __forceinline__ void dp_mask_slice(DpLds const &D, DpRound const &R, uint32_t blk, uint32_t chunk)
{
	uint32_t const lane = lane_id();
	uint32_t const base = blk * 64u, idx = base + lane;
	bool const fresh = idx >= R.t0 && idx < R.t1;
	uint32_t const mine = D.Mr[idx & (DPW - 1u)];
	uint32_t const plo = 16u * chunk, phi = plo + 16u;
	// running minimum of the keys to the right of the slice, up to this lane: inclusive prefix-min
	// over the lanes >= phi (the lane's own key included)
	uint32_t pm = lane >= phi ? mine : 0xFFFFFFFFu;
	pm = min(pm, dpp_mov<DPP_ROW_SHR1, 0xF>(0xFFFFFFFFu, pm));
	pm = min(pm, dpp_mov<DPP_ROW_SHR2, 0xF>(0xFFFFFFFFu, pm));
	pm = min(pm, dpp_mov<DPP_ROW_SHR4, 0xF>(0xFFFFFFFFu, pm));
	pm = min(pm, dpp_mov<DPP_ROW_SHR8, 0xF>(0xFFFFFFFFu, pm));
	pm = min(pm, dpp_mov<DPP_ROW_BCAST15, 0xA>(0xFFFFFFFFu, pm));
	pm = min(pm, dpp_mov<DPP_ROW_BCAST31, 0xC>(0xFFFFFFFFu, pm));
	uint32_t runmin = lane >= phi ? pm : mine;
	uint32_t bits = (lane >= plo && lane < phi) ? (1u << (lane - plo)) : 0u;
#pragma unroll
	for (int pp = 15; pp >= 0; --pp)
	{
		uint32_t const p = plo + (uint32_t) pp;
		uint32_t const x = readlane_u32(mine, (int) p);
		bool const in = lane > p;
		bits |= (in && x <= runmin) ? (1u << pp) : 0u;
		runmin = in ? min(runmin, x) : runmin;
	}
	if (fresh) ((lds_u16 *) D.Kr)[(idx & (DPW - 1u)) * 4u + chunk] = (uint16_t) bits;
}

// All four slices of one block by one wave: four independent dependency chains that interleave
// (pipelined schedule, where only two waves do the update).
__device__ __forceinline__ void dp_mask_block(DpLds const &D, DpRound const &R, uint32_t blk)
{
	uint32_t const lane = lane_id();
	uint32_t const base = blk * 64u, idx = base + lane;
	bool const fresh = idx >= R.t0 && idx < R.t1;
	uint32_t const mine = D.Mr[idx & (DPW - 1u)];
	uint32_t runmin[4], bits[4];
#pragma unroll
	for (int c = 0; c < 4; ++c)
	{
		uint32_t const plo = 16u * c, phi = plo + 16u;
		uint32_t pm = lane >= phi ? mine : 0xFFFFFFFFu;
		pm = min(pm, dpp_mov<DPP_ROW_SHR1, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_SHR2, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_SHR4, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_SHR8, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_BCAST15, 0xA>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_BCAST31, 0xC>(0xFFFFFFFFu, pm));
		runmin[c] = lane >= phi ? pm : mine;
		bits[c] = (lane >= plo && lane < phi) ? (1u << (lane - plo)) : 0u;
	}
#pragma unroll
	for (int pp = 15; pp >= 0; --pp)
	{
#pragma unroll
		for (int c = 0; c < 4; ++c)
		{
			uint32_t const p = 16u * c + (uint32_t) pp;
			uint32_t const x = readlane_u32(mine, (int) p);
			bool const in = lane > p;
			bits[c] |= (in && x <= runmin[c]) ? (1u << pp) : 0u;
			runmin[c] = in ? min(runmin[c], x) : runmin[c];
		}
	}
	if (fresh)
		D.Kr[idx & (DPW - 1u)] = (unsigned long long) (bits[0] | (bits[1] << 16)) | ((unsigned long long) (bits[2] | (bits[3] << 16)) << 32);
}

// rmq.update, part 2 (rmq.hh:66-80): block blk is complete, push its samples on every level.
// mb: mailbox slot holding the old level-(p-1) samples for the levels whose input is older than the ring.
__device__ __forceinline__ void dp_push_samples(DpLds const &D, uint32_t blk, uint32_t mb)
{
	uint32_t const lane = lane_id();
	uint32_t const base = blk * 64u;
	uint32_t const mine = D.Mr[(base + lane) & (DPW - 1u)];
	uint32_t const new_val = wave_min_u32(mine);
	uint32_t const new_smp = base + (uint32_t) __builtin_ctzll(__ballot(mine == new_val));   // first minimum of the block
	uint32_t const bnum = blk + 1u;
	if (lane < DP_LEVELS && (1u << lane) <= bnum)
	{
		uint32_t const j = bnum - (1u << lane);
		uint32_t res = new_smp, resv = new_val;
		if (lane >= 1)
		{
			uint32_t smp, sval;
			if (lane < DP_HPMIN)
			{
				smp = D.Tr[(lane - 1u) * DP_TRN + (j & (DP_TRN - 1u))];
				sval = D.Trv[(lane - 1u) * DP_TRN + (j & (DP_TRN - 1u))];
			}
			else
			{
				smp = D.MBi[mb * 64u + lane];
				sval = D.MBv[mb * 64u + lane];
			}
			if (!(new_val < sval)) { res = smp; resv = sval; }    // rmq.hh:76-79 (smp1 == smp2)
		}
		D.Tr[lane * DP_TRN + (j & (DP_TRN - 1u))] = res;
		D.Trv[lane * DP_TRN + (j & (DP_TRN - 1u))] = resv;
	}
}

__global__ __launch_bounds__(1024) void k_dp(
	DpArrays const A, uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint32_t stride,
	uint32_t m, uint32_t n, uint32_t L, uint32_t *flags)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	lds_char *const lds0 = (lds_char *) smem;
	uint32_t const lds0_addr = __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) lds0);   // LDS byte address of the carve base
	uint32_t off = 0;
	auto take = [&](size_t bytes) { uint32_t const o = off; off += (uint32_t) ((bytes + 15) & ~size_t(15)); return o; };
	DpLds D;
	D.Kr = (lds_u64 *) (lds0 + take((size_t) DPW * 8));
	D.Mr = (lds_u32 *) (lds0 + take((size_t) DPW * 4));
	D.LBr = (lds_u32 *) (lds0 + take((size_t) DP_STG * 4));
	D.SZr = (lds_u32 *) (lds0 + take((size_t) DP_STG * 4));
	D.Tr = (lds_u32 *) (lds0 + take((size_t) DP_LEVELS * DP_TRN * 4));
	D.Trv = (lds_u32 *) (lds0 + take((size_t) DP_LEVELS * DP_TRN * 4));
	uint32_t const off_LS = take((size_t) DP_SLOTS * DP_RL * 64 * 8);
	uint32_t const off_H = take((size_t) DP_SLOTS * 64 * 16);
	uint32_t const off_MBi = take((size_t) DP_MBSLOTS * 64 * 4);
	uint32_t const off_MBv = take((size_t) DP_MBSLOTS * 64 * 4);
	D.LS = (lds_u32 *) (lds0 + off_LS);
	D.H = (lds_u32 *) (lds0 + off_H);
	D.MBi = (lds_u32 *) (lds0 + off_MBi);
	D.MBv = (lds_u32 *) (lds0 + off_MBv);
	lds_u32 *const fbcnt = (lds_u32 *) (lds0 + take(16));     // cells per round that needed the general path (ring of 4 rounds)

	uint32_t const wave = __builtin_amdgcn_readfirstlane(wave_id());
	uint32_t const lane = lane_id();
	uint32_t const p2lim = min(2u * L, n - L) - 1u;          // lp.cc:72
	uint32_t const last_end = n - L;                          // lp.cc:113

	// Two schedules.  Classic: rounds of <= min(L, 56) cells, the rmq.update of a round between two
	// barriers.  Pipelined (L >= 96): rounds of 48 cells -- a round then never reads
	// what the previous round wrote (a cell reads entries <= end - 2L), so two dedicated waves do
	// the update of round r-1 while the compute waves are already in round r: one barrier a round.
	DpSchedule S;
	S.L = L; S.n = n;
	S.pipe = L >= 96u;                                        // measured: pays only with 4 cells per compute wave
	S.RL = S.pipe ? (min(L / 2u, 48u) / 12u) * 12u : min(L, DP_RL);   // pipelined: whole cells per compute wave
	S.nreg = (last_end - L) / S.RL + 1u;
	S.nrounds = S.nreg + (S.pipe ? 2u : 1u);
	uint32_t const NWC = S.pipe ? 12u : DP_NWC;               // compute waves
	uint32_t const nrounds = S.nrounds, RL = S.RL;
	uint32_t const npairs = (RL + 1u) / 2u;                    // list DMA instructions per round (+3: headers, mailbox)

	// loader: all LDS-DMA of round R (exactly DP_Q instructions, so that vmcnt counts rounds)
	auto load_round = [&](uint32_t r) {
		DpRound const R = dp_round(S, r);
		uint32_t const slot = r % DP_SLOTS;
		// (RL + 1) / 2 list loads (two cells each) whatever the round's own length, so that every
		// round issues the same number of instructions and vmcnt counts rounds
		// The loader's ~30 issues a round compete with three compute waves of its SIMD (measured: the round
		// waits for it), so the loop is kept to five instructions a load: M0 saved once, no address select
		// while both cells of a load exist.
		uint2 const *src = ent + (size_t) (R.e0 - 1u + (lane >> 5)) * stride + (lane & 31u) * 2u;
		uint32_t dst = __builtin_amdgcn_readfirstlane(lds0_addr + off_LS + slot * DP_RL * 512u);
		uint32_t const nfull = R.len / 2u;
		uint32_t q = 0;
		uint32_t const m0_keep = m0_save();
#pragma nounroll
		for (; q < nfull; ++q)
		{
			dma16_m0(src, dst);
			src += 2u * (size_t) stride;
			dst += 1024u;
		}
#pragma nounroll
		for (; q < npairs; ++q)
		{
			uint32_t const i = 2u * q + (lane >> 5);
			dma16_m0(i < R.len ? (void const *) src : (void const *) ent, dst);
			src += 2u * (size_t) stride;
			dst += 1024u;
		}
		m0_restore(m0_keep);
		{
			uint32_t const k = (lane < R.len) ? R.e0 + lane - 1u : 0u;
			dma16(hdr + k, __builtin_amdgcn_readfirstlane(lds0_addr + off_H + slot * 1024u));
		}
		{
			// old sparse-table samples for the block that completes in round R (at most one: RL < 64)
			uint32_t const blk = R.t0 >> 6;
			uint32_t const bnum = blk + 1u;
			bool const completes = R.len > 0 && !R.final_round && bnum * 64u <= R.t1;
			size_t o = 0;
			if (completes && lane >= DP_HPMIN && lane < DP_LEVELS && (1u << lane) <= bnum)
				o = (size_t) (lane - 1u) * A.tstride + (bnum - (1u << lane));
			uint32_t const mb = r % DP_MBSLOTS;
			dma4(A.Tb + o, __builtin_amdgcn_readfirstlane(lds0_addr + off_MBi + mb * 256u));
			dma4(A.Tbv + o, __builtin_amdgcn_readfirstlane(lds0_addr + off_MBv + mb * 256u));
		}
	};
	// writer: round P to HBM (stores only, never waited for)
	auto flush_round = [&](uint32_t r) {
		DpRound const P = dp_round(S, r);
		if (lane < P.len)
		{
			uint32_t const t = P.t0 + lane;
			A.M[t] = D.Mr[t & (DPW - 1u)];
			A.LB[t] = D.LBr[t & (DP_STG - 1u)];
			A.SZ[t] = D.SZr[t & (DP_STG - 1u)];
			if (!P.final_round) A.K[t] = D.Kr[t & (DPW - 1u)];
		}
		uint32_t const bnum = (P.t0 >> 6) + 1u;
		if (P.len > 0 && !P.final_round && bnum * 64u <= P.t1 && lane < DP_LEVELS && (1u << lane) <= bnum)
		{
			uint32_t const j = bnum - (1u << lane);
			A.Tb[(size_t) lane * A.tstride + j] = D.Tr[lane * DP_TRN + (j & (DP_TRN - 1u))];
			A.Tbv[(size_t) lane * A.tstride + j] = D.Trv[lane * DP_TRN + (j & (DP_TRN - 1u))];
		}
	};

	if (wave == DP_LOADER)
	{
		load_round(0);
		if (nrounds > 1) load_round(1);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	if (threadIdx.x < 4) fbcnt[threadIdx.x] = 0;
	dp_barrier();

#ifdef FSEQ_DP_STAMPS
	unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
#define DP_STAMP(x) unsigned long long x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define DP_STAMP(x)
#endif
	uint32_t pair_cool = 0;                                   // rounds left before this wave tries cell pairs again
	for (uint32_t r = 0; r < nrounds; ++r)
	{
		DP_STAMP(ts0);
		DpRound const R = dp_round(S, r);
		uint32_t const slot = r % DP_SLOTS;
		// Entries < filled are indexed (masks + samples).  The ring keeps the last DPW entries and
		// this round's (pipelined: this and the previous round's) results land above `filled`, so
		// anything >= filled + (writes in flight) - DPW (+ margin) is safe to read from LDS.
		uint32_t filled;
		if (R.final_round) filled = n - 2u * L + 1u;
		else if (!S.pipe) filled = R.t0;
		else filled = (r == 0) ? 0u : dp_round(S, r - 1u).t0;
		uint32_t const inflight = S.pipe ? 2u * RL : RL;
		DpView V;
		V.safe_lo = (filled + inflight + 128u > DPW) ? (filled + inflight + 128u - DPW) : 0u;
		V.cb = filled >> 6;

		if (wave < NWC)
		{
			// ---- compute (one CU's VALU issue rate bounds this phase: a stage-interleaved, branch-free
			// variant was measured slower)
			auto single_cell = [&](uint32_t i) {
				uint32_t const end = R.e0 + i;
				uint32_t const t = end - L;
				lds_u32 const *hp = D.H + (slot * 64u + i) * 4u;
				uint4 const h = make_uint4(hp[0], hp[1], hp[2], hp[3]);
				CellState st;
				if (!R.final_round && end <= p2lim)
				{
					// part 2, lp.cc:85-93
					st.best_v = m - h.y; st.best_lb = 0; st.best_sz = st.best_v; st.cum_base = 0;
				}
				else
				{
					lds_u32 const *ls = D.LS + (slot * DP_RL + i) * 128u;
					uint2 const en = make_uint2(ls[2u * lane], ls[2u * lane + 1u]);
					uint32_t const vnext = ls[2u * ((lane + 1u) & 63u)];
					st = dp_cell_sequential(A, D, V, ent + (size_t) (end - 1u) * stride, en, vnext, h, m, L, end, flags);
				}
				if (lane == 0)
				{
					D.Mr[t & (DPW - 1u)] = st.best_v;
					D.LBr[t & (DP_STG - 1u)] = st.best_lb;
					D.SZr[t & (DP_STG - 1u)] = st.best_sz;
				}
			};
			// the choice between cell pairs and full-wave cells must be the same in every wave (it decides
			// which wave owns which cell): all waves read the same counter of the previous round
			if (r >= 1u && 2u * fbcnt[(r - 1u) & 3u] > RL) pair_cool = 16u;   // long lists here: full-wave cells for a while
			if (wave == 0 && lane == 0) fbcnt[(r + 2u) & 3u] = 0;
			// one copy of each cell routine in the instruction stream (the kernel must stay well inside the
			// instruction cache): cells for the general path are collected in a wave-uniform bit mask
			unsigned long long todo = 0;
			if (!R.final_round && R.e0 > p2lim && pair_cool == 0u)
			{
				// two cells per step, one per half wave: pairs wave, wave + NWC
#pragma nounroll
				for (uint32_t i0 = 2u * wave; i0 < R.len; i0 += 2u * NWC)
				{
					uint64_t const fb = dp_cell_pair(A, D, V, slot, i0, R, m, L, flags);
					if (fb & 1ull) todo |= 1ull << i0;
					if (fb >> 32) todo |= 2ull << i0;
				}
				if (todo && lane == 0) atomicAdd((uint32_t *) (fbcnt + (r & 3u)), (uint32_t) __popcll(todo));
			}
			else
			{
				if (pair_cool) --pair_cool;
				for (uint32_t i = wave; i < R.len; i += NWC) todo |= 1ull << i;
			}
#pragma nounroll
			while (todo)
			{
				uint32_t const i = (uint32_t) __builtin_ctzll(todo);
				todo &= todo - 1ull;
				single_cell(i);
			}
		}
		else if (wave == DP_LOADER)
		{
			// ---- loader: lists of round r+2; then make sure round r+1 has landed
			if (r + 2u < nrounds)
			{
				load_round(r + 2u);
				dp_wait_all_but(npairs + 3u);                  // = everything but the round just issued
			}
			else
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		else if (wave == DP_WRITER)
		{
			// ---- writer: a finished *and indexed* round to HBM
			uint32_t const lag = S.pipe ? 2u : 1u;
			if (r >= lag) flush_round(r - lag);
		}
		else
		{
			// ---- pipelined mode, waves 12 and 13: rmq.update of round r-1 (rmq.hh:61-81)
			if (r >= 1u)
			{
				DpRound const P = dp_round(S, r - 1u);
				if (P.len > 0 && !P.final_round)
				{
					uint32_t const blkA = P.t0 >> 6, blkB = (P.t1 - 1u) >> 6;
					if (wave == 12u)
					{
						dp_mask_block(D, P, blkA);
						if ((blkA + 1u) * 64u <= P.t1) dp_push_samples(D, blkA, (r - 1u) % DP_MBSLOTS);
					}
					else if (blkB != blkA)
						dp_mask_block(D, P, blkB);
				}
			}
		}
		DP_STAMP(ts1);
		dp_barrier();
		DP_STAMP(ts2);
		if (R.final_round) break;                             // no rmq.update after the last cell
		if (S.pipe)
		{
#ifdef FSEQ_DP_STAMPS
			acc[0] += ts1 - ts0; acc[1] += ts2 - ts1; acc[4] += 1;
#endif
			continue;
		}

		// ---- classic mode: rmq.update for the new entries [t0, t1) (rmq.hh:61-81).  At most two
		// 64-blocks are touched and at most one completes (RL < 64).  Eight waves build the stack
		// masks, each a 16-bit slice of one block; a ninth pushes the sparse-table samples.
		{
			uint32_t const blkA = R.t0 >> 6, blkB = (R.t1 - 1u) >> 6;
			uint32_t const nslices = 4u * (blkB - blkA + 1u);
			if (wave < nslices)
				dp_mask_slice(D, R, blkA + (wave >> 2), wave & 3u);
			else if (wave == 8 && (blkA + 1u) * 64u <= R.t1)
				dp_push_samples(D, blkA, r % DP_MBSLOTS);
		}
		DP_STAMP(ts3);
		dp_barrier();
#ifdef FSEQ_DP_STAMPS
		{
			DP_STAMP(ts4);
			acc[0] += ts1 - ts0; acc[1] += ts2 - ts1; acc[2] += ts3 - ts2; acc[3] += ts4 - ts3; acc[4] += 1;
		}
#endif
	}
#ifdef FSEQ_DP_STAMPS
	if (lane == 0)
	{
		unsigned long long *o = reinterpret_cast<unsigned long long *>(flags + 8) + wave * 3u;
		o[0] = acc[0]; o[1] = acc[1] + acc[2] + acc[3]; o[2] = acc[4];
	}
#endif

	// the writer flushes what is still only in LDS
	if (wave == DP_WRITER)
	{
		uint32_t const lag = S.pipe ? 2u : 1u;
		for (uint32_t r = (nrounds >= lag ? nrounds - lag : 0u); r < nrounds; ++r) flush_round(r);
	}
}

// List-capacity estimate: the length-L segment ending at block boundary k has #{i : d_k[i] > k - L} distinct
// rows; the median over the boundaries sizes the per-column lists before phase C runs (a wrong guess only
// costs a retry, never the result).
__global__ __launch_bounds__(256) void k_boundary_recent(
	uint32_t const *__restrict__ bstate_d, uint32_t m, uint64_t n, uint32_t B, uint32_t L, uint32_t *__restrict__ out)
{
	uint64_t k = (uint64_t) blockIdx.x * B;
	if (k > n) k = n;
	if (k < L) { if (threadIdx.x == 0) out[blockIdx.x] = 0xFFFFFFFFu; return; }
	uint32_t const thr = (uint32_t) (k - L);
	uint32_t const *d = bstate_d + (size_t) blockIdx.x * m;
	uint32_t cnt = 0;
	for (uint32_t i = threadIdx.x; i < m; i += 256) cnt += d[i] > thr ? 1u : 0u;
	__shared__ uint32_t red[4];
	cnt = readlane_u32(wave_incl_add(cnt), 63);
	if (lane_id() == 0) red[wave_id()] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// copy the lists of selected columns into a compact buffer (for the host-side merge)
__global__ __launch_bounds__(64) void k_gather_lists(
	uint64_t const *__restrict__ cols, uint32_t stride, uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr,
	uint2 *__restrict__ out_ent, uint4 *__restrict__ out_hdr)
{
	uint64_t const k = cols[blockIdx.x];
	uint4 const h = hdr[k];
	if (threadIdx.x == 0) out_hdr[blockIdx.x] = h;
	for (uint32_t i = threadIdx.x; i < h.x; i += 64)
		out_ent[(size_t) blockIdx.x * stride + i] = ent[k * (size_t) stride + i];
}

} // namespace fseq
