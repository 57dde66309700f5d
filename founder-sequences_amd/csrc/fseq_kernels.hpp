// fseq_kernels.hpp -- HIP kernels of the segmentation path for gfx950.
//
// Column-block decomposition of segmentation_lp_context's two hot loops
// (founder-sequences/segmentation_lp_context.cc:26-188 pass 1, update_pbwt_task.cc:13-35 pass 2):
//   phase A  k_colblock<MODE_RANK>  per block of B columns: pBWT from the identity -> dense co-lex rank of
//                                   every row's block key + divergence in front of each distinct key
//   phase B  k_chain                serial over blocks, one workgroup: stable sort of the running order by
//                                   block rank (4-bit LSD digit passes of the same partition step); emits
//                                   the exact (a_k, d_k) at every block boundary
//   phase C  k_columns              per block from its boundary state: the true per-column update; per
//                                   column the top of the divergence-value histogram (what
//                                   calculate_segmentation_lp_dp_arg consumes, lp.cc:393-481)
//   phase D  k_dp                   the DP over columns incl. the exact rmq.hh semantics
//   pass 2   k_colblock<MODE_SNAP>  (a,d) at the merged segment boundaries
// All state of a block lives in LDS; HBM traffic is the 1 B/cell column stream plus the outputs.
#pragma once

#include "fseq_core.hpp"

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

// one thread per 4 consecutive rows of one column (coalesced 4-byte stores down the column)
__global__ __launch_bounds__(256) void k_synth(SynthArgs const A, uint8_t *__restrict__ msa, size_t ld, uint32_t m, uint64_t n)
{
	uint64_t const quads_per_col = ld / 4;
	uint64_t const gid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if (gid >= quads_per_col * n) return;
	uint64_t const c = gid / quads_per_col;
	uint32_t const r0 = (uint32_t) (gid % quads_per_col) * 4u;
	uint64_t const b = c / A.block_len;
	uint32_t packed = 0;
#pragma unroll
	for (int q = 0; q < 4; ++q)
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
		packed |= code << (8 * q);
	}
	*reinterpret_cast<uint32_t *>(msa + c * ld + r0) = packed;
}

// ------------------------------------------------------------------------------------------------
// Shared pieces of the column kernels
// ------------------------------------------------------------------------------------------------
template <int T, int E>
__device__ __forceinline__ void read_chunk(uint32_t const *a_l, uint32_t const *d_l, uint32_t (&a)[E], uint32_t (&d)[E])
{
	uint32_t const p0 = threadIdx.x * E;
#pragma unroll
	for (int e = 0; e < E; ++e) { a[e] = a_l[p0 + e]; d[e] = d_l[p0 + e]; }
}

// MODE_RANK: start from the identity at column k0 = blockIdx.x * B, emit block ranks (phase A).
// MODE_SNAP: start from the block boundary state below task_rb[blockIdx.x], emit (a,d) at that column.
enum { MODE_RANK = 0, MODE_SNAP = 1 };

template <int T, int E, int SIGMA>
__host__ __device__ inline size_t colblock_lds_bytes()
{
	constexpr size_t CAP = (size_t) T * E;
	return 2 * carve_bytes(CAP, 4) + 2 * carve_bytes(CAP, 1) + carve_bytes(1, sizeof(StepScratch<T, SIGMA>)) + carve_bytes(T / WAVE + 1, 4);
}

template <int T, int E, int SIGMA, int MODE>
__global__ __launch_bounds__(T) void k_colblock(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblocks,
	// MODE_RANK outputs
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys,
	// MODE_SNAP inputs / outputs
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint64_t const *__restrict__ task_rb, uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	uint32_t *a_l = cv.take<uint32_t>(CAP);
	uint32_t *d_l = cv.take<uint32_t>(CAP);
	uint8_t *sym0 = cv.take<uint8_t>(CAP);
	uint8_t *sym1 = cv.take<uint8_t>(CAP);
	StepScratch<T, SIGMA> &scr = *cv.take<StepScratch<T, SIGMA>>(1);
	uint32_t *sscr = cv.take<uint32_t>(T / WAVE + 1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint64_t k0, kend;
	if (MODE == MODE_RANK)
	{
		k0 = (uint64_t) blockIdx.x * B;
		kend = (k0 + B < n) ? k0 + B : n;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			a_l[pos] = pos < m ? pos : 0u;
			d_l[pos] = pos < m ? (uint32_t) k0 : 0u;
		}
	}
	else
	{
		uint64_t const rb = task_rb[blockIdx.x];
		uint64_t blk = rb / B;
		if (blk > nblocks) blk = nblocks;
		k0 = blk * B;
		kend = rb;
		uint32_t const *sa = bstate_a + blk * (size_t) m;
		uint32_t const *sd = bstate_d + blk * (size_t) m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			a_l[idx] = idx < m ? sa[idx] : 0u;
			d_l[idx] = idx < m ? sd[idx] : 0u;
		}
	}
	uint32_t const nb = (uint32_t) (kend - k0);

	bool const has_chunk = tid * 16u < m;
	uint4 nxt = make_uint4(0, 0, 0, 0);
	if (nb && has_chunk)
	{
		nxt = *reinterpret_cast<uint4 const *>(msa + k0 * ld + tid * 16u);
		*reinterpret_cast<uint4 *>(sym0 + tid * 16u) = nxt;
	}
	__syncthreads();

	for (uint32_t j = 0; j < nb; ++j)
	{
		uint8_t const *symc = (j & 1u) ? sym1 : sym0;
		uint8_t *symn = (j & 1u) ? sym0 : sym1;
		bool const more = j + 1 < nb;
		if (more && has_chunk)
			nxt = *reinterpret_cast<uint4 const *>(msa + (k0 + j + 1) * ld + tid * 16u);

		uint32_t a[E], d[E], s[E], dst[E], dnew[E];
		read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
		for (int e = 0; e < E; ++e) s[e] = (p0 + e < m) ? (uint32_t) symc[a[e]] : (uint32_t) SIGMA;

		partition_step<T, E, SIGMA>(d, s, (uint32_t) (k0 + j + 1), scr, dst, dnew);

#pragma unroll
		for (int e = 0; e < E; ++e)
			if (p0 + e < m) { a_l[dst[e]] = a[e]; d_l[dst[e]] = dnew[e]; }
		if (more && has_chunk)
			*reinterpret_cast<uint4 *>(symn + tid * 16u) = nxt;
		__syncthreads();
	}

	if (MODE == MODE_RANK)
	{
		// rank of a row = number of bucket starts at or before its position, minus one; a position
		// starts a new block key iff its in-block divergence is > k0 (position 0 always is).
		uint32_t a[E], d[E];
		read_chunk<T, E>(a_l, d_l, a, d);
		uint32_t nf = 0;
#pragma unroll
		for (int e = 0; e < E; ++e) nf += (p0 + e < m && (p0 + e == 0 || d[e] > (uint32_t) k0)) ? 1u : 0u;
		uint32_t total;
		uint32_t r = block_excl_add<T>(nf, sscr, &total);
		size_t const ob = (size_t) blockIdx.x * m;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			if (pos < m)
			{
				bool const first = (pos == 0 || d[e] > (uint32_t) k0);
				r += first ? 1u : 0u;
				rank[ob + a[e]] = r - 1u;
				if (first) keyd[ob + r - 1u] = d[e];
			}
		}
		if (tid == 0) nkeys[blockIdx.x] = total;
	}
	else
	{
		size_t const ob = (size_t) blockIdx.x * m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			if (idx < m) { snap_a[ob + idx] = a_l[idx]; snap_d[ob + idx] = d_l[idx]; }
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Phase B: the serial chain over blocks (one workgroup).
// ------------------------------------------------------------------------------------------------
template <int T, int E>
__host__ __device__ inline size_t chain_lds_bytes()
{
	constexpr size_t CAP = (size_t) T * E;
	return 4 * carve_bytes(CAP, 4) + carve_bytes(1, sizeof(StepScratch<T, 16>));
}

__device__ __forceinline__ uint32_t rank_digits(uint32_t D)
{
	if (D <= 1) return 0;
	uint32_t const bits = 32u - (uint32_t) __builtin_clz(D - 1u);
	return (bits + 3u) / 4u;
}

template <int T, int E>
__global__ __launch_bounds__(T) void k_chain(
	uint32_t const *__restrict__ rank, uint32_t const *__restrict__ keyd, uint32_t const *__restrict__ nkeys,
	uint32_t m, uint32_t nblocks, uint32_t *__restrict__ bstate_a, uint32_t *__restrict__ bstate_d)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	uint32_t *a_l = cv.take<uint32_t>(CAP);
	uint32_t *d_l = cv.take<uint32_t>(CAP);
	uint32_t *rk = cv.take<uint32_t>(CAP);
	uint32_t *kd = cv.take<uint32_t>(CAP);
	StepScratch<T, 16> &scr = *cv.take<StepScratch<T, 16>>(1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
#pragma unroll
	for (int i = 0; i < E; ++i)
	{
		uint32_t const idx = tid + i * T;
		a_l[idx] = idx < m ? idx : 0u;
		d_l[idx] = 0u;
		rk[idx] = idx < m ? rank[idx] : 0u;
		kd[idx] = idx < m ? keyd[idx] : 0u;
	}
	uint32_t D_cur = nkeys[0];
	__syncthreads();

	for (uint32_t b = 0; b < nblocks; ++b)
	{
		// (1) publish the exact state at column b*B
		{
			size_t const ob = (size_t) b * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t const idx = tid + i * T;
				if (idx < m) { bstate_a[ob + idx] = a_l[idx]; bstate_d[ob + idx] = d_l[idx]; }
			}
		}
		// (2) prefetch the next block's rank / keyd into registers
		uint32_t pr[E], pk[E], D_next = 0;
		bool const more = b + 1 < nblocks;
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

		// (3) LSD digit passes over the block rank
		uint32_t const nd = rank_digits(D_cur);
		for (uint32_t p = 0; p < nd; ++p)
		{
			uint32_t a[E], d[E], s[E], dst[E], dnew[E];
			read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
			for (int e = 0; e < E; ++e)
				s[e] = (p0 + e < m) ? ((rk[a[e]] >> (4u * p)) & 15u) : 16u;
			partition_step<T, E, 16>(d, s, 0u, scr, dst, dnew);
#pragma unroll
			for (int e = 0; e < E; ++e)
				if (p0 + e < m) { a_l[dst[e]] = a[e]; d_l[dst[e]] = dnew[e]; }
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
				rk[idx] = pr[i];
				kd[idx] = pk[i];
			}
		}
		D_cur = D_next;
		__syncthreads();
	}

	{
		size_t const ob = (size_t) nblocks * m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			if (idx < m) { bstate_a[ob + idx] = a_l[idx]; bstate_d[ob + idx] = d_l[idx]; }
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Phase C: true per-column updates inside a block + per-column top of the divergence histogram.
// Divergence values are replaced by order-preserving ids (sorted distinct boundary values first,
// then one new id per column), so the histogram is a dense LDS table updated incrementally.
// ------------------------------------------------------------------------------------------------
template <int T, int E, int SIGMA>
__host__ __device__ inline size_t columns_lds_bytes(uint32_t B)
{
	constexpr size_t CAP = (size_t) T * E;
	return carve_bytes(2 * CAP, 4) + 2 * carve_bytes(CAP, 1) + carve_bytes(CAP + B, 4) + carve_bytes(CAP, 4)
	     + carve_bytes(1, sizeof(StepScratch<T, SIGMA>)) + carve_bytes(T / WAVE + 1, 4);
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
template <int T, int E, int SIGMA>
__global__ __launch_bounds__(T) void k_columns(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t N2,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr)
{
	constexpr uint32_t CAP = T * E;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	uint32_t *a_l = cv.take<uint32_t>(2 * CAP);
	uint32_t *d_l = a_l + CAP;
	uint8_t *sym0 = cv.take<uint8_t>(CAP);
	uint8_t *sym1 = cv.take<uint8_t>(CAP);
	uint32_t *cnt_l = cv.take<uint32_t>(CAP + B);
	uint32_t *V_l = cv.take<uint32_t>(CAP);
	StepScratch<T, SIGMA> &scr = *cv.take<StepScratch<T, SIGMA>>(1);
	uint32_t *sscr = cv.take<uint32_t>(T / WAVE + 1);

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint64_t const k0 = (uint64_t) blockIdx.x * B;
	uint64_t const kend = (k0 + B < n) ? k0 + B : n;
	uint32_t const nb = (uint32_t) (kend - k0);

	// ---- prologue: boundary state -> registers
	{
		size_t const ob = (size_t) blockIdx.x * m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t const idx = tid + i * T;
			a_l[idx] = idx < m ? bstate_a[ob + idx] : 0u;
			d_l[idx] = idx < m ? bstate_d[ob + idx] : 0u;
		}
	}
	__syncthreads();
	uint32_t a[E], d[E];
	read_chunk<T, E>(a_l, d_l, a, d);
	__syncthreads();

	// ---- sorted distinct divergence values -> V_l[0..D0)
	uint32_t *sb = a_l;                       // N2 <= 2*CAP words
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
	for (uint32_t i = tid; i < D0 + nb; i += T) cnt_l[i] = 0;
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
			atomicAdd(&cnt_l[lo], 1u);
		}
	}
	__syncthreads();                          // everyone is done with sb (aliases a_l/d_l)
#pragma unroll
	for (int e = 0; e < E; ++e) { a_l[p0 + e] = a[e]; d_l[p0 + e] = id[e]; }

	bool const has_chunk = tid * 16u < m;
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

		uint32_t s[E], dst[E], dnew[E];
		read_chunk<T, E>(a_l, d_l, a, d);
#pragma unroll
		for (int e = 0; e < E; ++e) s[e] = (p0 + e < m) ? (uint32_t) symc[a[e]] : (uint32_t) SIGMA;

		partition_step<T, E, SIGMA>(d, s, D0 + j, scr, dst, dnew);

#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			if (p0 + e < m)
			{
				a_l[dst[e]] = a[e];
				d_l[dst[e]] = dnew[e];
				if (dnew[e] != d[e])
				{
					atomicSub(&cnt_l[d[e]], 1u);
					atomicAdd(&cnt_l[dnew[e]], 1u);
				}
			}
		}
		if (more && has_chunk)
			*reinterpret_cast<uint4 *>(symn + tid * 16u) = nxt;
		__syncthreads();

		// ---- emit the top of the histogram for column k0+j (wave 0; the others run ahead into
		// the next column and meet it again at the partition step's barrier)
		if (wave_id() == 0)
		{
			uint32_t const lane = lane_id();
			uint64_t const k = k0 + j;
			uint2 *out = ent + k * (size_t) stride;
			int32_t const top = (int32_t) (D0 + j);
			uint32_t cum = 0, nent = 0;
			for (int32_t base = top; base >= 0; base -= 64)
			{
				int32_t const i = base - (int32_t) lane;
				uint32_t const c = (i >= 0) ? cnt_l[i] : 0u;
				bool const nz = c > 0;
				uint64_t const mask = __ballot(nz);
				uint32_t const inc = wave_incl_add(c);
				uint32_t const exc = cum + inc - c;
				bool const take = nz && exc <= X;
				uint32_t const pos = nent + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
				if (take)
				{
					uint32_t const v = ((uint32_t) i < D0) ? V_l[i] : (uint32_t) (k0 + ((uint32_t) i - D0) + 1u);
					out[pos] = make_uint2(v, c);
				}
				uint64_t const tmask = __ballot(take);
				nent += (uint32_t) __popcll(tmask);
				// counts of the taken lanes: they are a prefix of the non-zero lanes
				uint32_t const taken_inc = shfl_u32(inc, tmask ? 63 - (int) __builtin_clzll(tmask) : 0);
				cum += tmask ? taken_inc : 0u;
				if (tmask != mask || cum > X) break;
			}
			if (lane == 0)
				hdr[k] = make_uint4(nent, zero_present ? cnt_l[0] : 0u, cum == m ? 1u : 0u, cum);
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
struct DpArrays {
	uint32_t *M, *LB, *SZ, *P, *S, *Tb;
	uint32_t tstride;
};

__device__ __forceinline__ uint32_t rmq_naive(DpArrays const &A, uint32_t b, uint32_t e)
{
	uint32_t const bb = b >> 6, eb = (e - 1u) >> 6;
	if (bb == eb)
	{
		if ((b & 63u) == 0) return A.P[e - 1u];
		if ((e & 63u) == 0) return A.S[b];
		uint32_t best = b, bv = A.M[b];
		for (uint32_t u = b + 1; u < e; ++u)
		{
			uint32_t const v = A.M[u];
			if (v < bv) { bv = v; best = u; }
		}
		return best;
	}
	uint32_t const left = A.S[b], right = A.P[e - 1u];
	return (A.M[right] < A.M[left]) ? right : left;
}

// rmq.hh:85-105
__device__ __forceinline__ uint32_t rmq_query(DpArrays const &A, uint32_t beg, uint32_t end)
{
	uint32_t const beg_block = (beg >> 6) + 1u;
	uint32_t const end_block = end >> 6;
	if (beg_block >= end_block)
		return rmq_naive(A, beg, end);
	uint32_t const pow2 = 31u - (uint32_t) __builtin_clz(end_block - beg_block);
	uint32_t const smp1 = A.Tb[(size_t) pow2 * A.tstride + beg_block];
	uint32_t const smp2 = A.Tb[(size_t) pow2 * A.tstride + end_block - (1u << pow2)];
	uint32_t smp = (A.M[smp2] < A.M[smp1]) ? smp2 : smp1;
	uint32_t const left = A.S[beg];
	smp = (A.M[left] < A.M[smp]) ? left : smp;
	if (end == end_block * 64u)
		return smp;
	uint32_t const right = A.P[end - 1u];
	return (A.M[right] < A.M[smp]) ? right : smp;
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
	{
		uint32_t const lo = (uint32_t) __shfl_xor((int) (uint32_t) v, off, WAVE);
		uint32_t const hi = (uint32_t) __shfl_xor((int) (uint32_t) (v >> 32), off, WAVE);
		uint64_t const o = ((uint64_t) hi << 32) | lo;
		v = o < v ? o : v;
	}
	return v;
}

// One DP cell, evaluated by a whole wave (calculate_segmentation_lp_dp_arg, lp.cc:393-481, with the
// candidate ranges visited in descending divergence order and pruned exactly).
__device__ __forceinline__ void dp_cell(
	DpArrays const &A, uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint32_t stride,
	uint32_t m, uint32_t L, uint32_t end, bool part2, uint32_t *flags)
{
	uint32_t const lane = lane_id();
	uint32_t const k = end - 1u;
	uint4 const h = hdr[k];
	uint32_t const nent = h.x, cnt0 = h.y, complete = h.z;
	uint32_t const t = end - L;
	if (part2)
	{
		// lp.cc:85-93
		if (lane == 0) { uint32_t const sz = m - cnt0; A.M[t] = sz; A.LB[t] = 0; A.SZ[t] = sz; }
		return;
	}
	uint2 const *list = ent + (size_t) k * stride;
	uint32_t best_v = 0xFFFFFFFFu, best_lb = 0, best_sz = 0;
	uint32_t cum_base = 0;
	for (uint32_t s0 = 0; s0 < nent; s0 += 64)
	{
		uint32_t const i = s0 + lane;
		bool const valid = i < nent;
		uint2 const en = valid ? list[i] : make_uint2(0u, 0u);
		uint32_t vnext = shfl_dn_u32(en.x, 1);
		if (lane == 63) vnext = (s0 + 64 < nent) ? list[s0 + 64].x : 0u;
		bool const have_next = i + 1 < nent;
		bool const is0 = valid && en.x == 0u;
		uint32_t const cc = (valid && !is0) ? en.y : 0u;
		uint32_t const cum = cum_base + wave_incl_add(cc);

		bool ok = valid && !is0 && have_next && vnext != 0u;
		uint32_t lo = vnext;
		uint32_t const c = min(en.x, end + 1u - L);          // lp.cc:444-445 (text_pos + 2 - L)
		if (lo < L)                                          // lp.cc:449-455 (lb == 0)
		{
			if (L < c) lo = L; else ok = false;
		}
		ok = ok && lo < c;                                   // lp.cc:458
		uint32_t val = 0xFFFFFFFFu, idx = 0;
		if (ok)
		{
			idx = rmq_query(A, lo - L, c - L);               // lp.cc:465
			val = max(A.M[idx], cum);                        // lp.cc:468-471
		}
		uint64_t const key = ok ? (((uint64_t) val << 32) | (uint64_t) (0xFFFFFFFFu - i)) : ~0ull;
		uint64_t const kmin = wave_min_u64(key);
		if (kmin != ~0ull)
		{
			uint32_t const v = (uint32_t) (kmin >> 32);
			if (v <= best_v)
			{
				uint64_t const wmask = __ballot(key == kmin);
				int const src = (int) __builtin_ctzll(wmask);
				best_v = v;
				best_lb = shfl_u32(idx, src) + L;
				best_sz = shfl_u32(cum, src);
			}
		}
		cum_base = shfl_u32(cum, 63);
		if (best_v != 0xFFFFFFFFu && cum_base > best_v) break;
	}
	bool const stopped = best_v != 0xFFFFFFFFu && cum_base > best_v;
	if (!complete && !stopped)
	{
		if (lane == 0) atomicOr(flags, 1u);                  // list too short to prove the result
	}
	if (complete && cnt0 > 0)
	{
		uint32_t const w = m - cnt0;                         // lp.cc:416-421, visited first by the reference
		if (w <= best_v) { best_v = w; best_lb = 0; best_sz = w; }
	}
	if (m <= best_v) { best_v = m; best_lb = 0; best_sz = m; }   // initial min_arg, lp.cc:123
	if (lane == 0) { A.M[t] = best_v; A.LB[t] = best_lb; A.SZ[t] = best_sz; }
}

template <int T>
__global__ __launch_bounds__(T) void k_dp(
	DpArrays const A, uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint32_t stride,
	uint32_t m, uint32_t n, uint32_t L, uint32_t *flags)
{
	constexpr uint32_t NW = T / WAVE;
	uint32_t const wave = wave_id();
	uint32_t const lane = lane_id();
	uint32_t const p2lim = min(2u * L, n - L) - 1u;          // lp.cc:72
	uint32_t const last_end = n - L;                          // lp.cc:113

	for (uint32_t e0 = L; e0 <= last_end; e0 += L)
	{
		uint32_t const len = min(L, last_end - e0 + 1u);
		for (uint32_t i = wave; i < len; i += NW)
		{
			uint32_t const end = e0 + i;
			dp_cell(A, ent, hdr, stride, m, L, end, end <= p2lim, flags);
		}
		__syncthreads();
		if (wave == 0)
		{
			// rmq.update for the new entries [t0, t1)  (rmq.hh:61-81)
			uint32_t const t0 = e0 - L, t1 = t0 + len;
			for (uint32_t blk = t0 >> 6; blk <= (t1 - 1u) >> 6; ++blk)
			{
				uint32_t const idx = blk * 64u + lane;
				bool const valid = idx < t1;
				uint32_t const v = valid ? A.M[idx] : 0xFFFFFFFFu;
				uint32_t pv = v, pi = idx;
#pragma unroll
				for (int delta = 1; delta < WAVE; delta <<= 1)
				{
					uint32_t const ov = shfl_up_u32(pv, delta), oi = shfl_up_u32(pi, delta);
					if (lane >= (uint32_t) delta && !(pv < ov)) { pv = ov; pi = oi; }
				}
				if (valid) A.P[idx] = pi;
				if (t1 >= blk * 64u + 64u)
				{
					uint32_t sv = v, si = idx;
#pragma unroll
					for (int delta = 1; delta < WAVE; delta <<= 1)
					{
						uint32_t const ov = shfl_dn_u32(sv, delta), oi = shfl_dn_u32(si, delta);
						if (lane + (uint32_t) delta < 64u && !(sv <= ov)) { sv = ov; si = oi; }
					}
					A.S[idx] = si;
					uint32_t const new_smp = shfl_u32(si, 0);
					uint32_t const new_val = shfl_u32(sv, 0);
					uint32_t const bnum = blk + 1u;
					if (lane == 0) A.Tb[blk] = new_smp;
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
					if (lane >= 1 && lane < 32 && (1u << lane) <= bnum)
					{
						uint32_t const j = bnum - (1u << lane);
						uint32_t const smp = A.Tb[(size_t) (lane - 1u) * A.tstride + j];
						A.Tb[(size_t) lane * A.tstride + j] = (new_val < A.M[smp]) ? new_smp : smp;
					}
					__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
				}
			}
		}
		__syncthreads();
	}
	// final cell at rb = n (lp.cc:165-183); no rmq.update afterwards
	if (wave == 0)
		dp_cell(A, ent, hdr, stride, m, L, n, false, flags);
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
