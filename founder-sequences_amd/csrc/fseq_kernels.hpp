// fseq_kernels.hpp -- HIP kernels of the segmentation path for gfx950.
//
// Column-block decomposition of segmentation_lp_context's two hot loops
// (founder-sequences/segmentation_lp_context.cc:26-188 pass 1, update_pbwt_task.cc:13-35 pass 2):
//   phase A  k_colblock<MODE_RANK>  per block of B columns: pBWT from the identity -> dense co-lex rank of
//                                   every row's block key + divergence in front of each distinct key
//   phase B  k_chain (x3 or x5)     stable sort of the running order by block rank (2-bit LSD digit passes of
//                                   the same partition step), two or three levels: compose groups of G blocks
//                                   into super-blocks (parallel), chain the top level (one workgroup), expand
//                                   each group (parallel) -> exact (a_k, d_k) at every block boundary
//   phase C  k_columns              per block from its boundary state: the true per-column update; per
//                                   column the top of the divergence-value histogram (what
//                                   calculate_segmentation_lp_dp_arg consumes, lp.cc:393-481)
//   phase D  k_dp (fseq_dp.hpp)     the DP over columns incl. the exact rmq.hh semantics
//   pass 2   k_colblock<MODE_SNAP>  (a,d) at the merged segment boundaries
// All state of a block lives in LDS (fseq_stream.hpp: in an HBM / L2 workspace for m > 11,264); HBM traffic
// is the packed column stream (2 / 4 / 8 bits per cell) plus the outputs.
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
	uint32_t const sh = ((a & ((1u << bsh) - 1u)) << (3u - bsh)) + 2u * pass;     // (8 >> bsh) bits per row
	return (byte >> sh) & 3u;
}

// one thread per 32-bit word of one column = 4 << bsh consecutive rows (coalesced stores down the column);
// blockIdx.y strides over the columns (a launch holds fewer than 2^32 threads per dimension)
// columns [c_begin, c_end) (a rank of a sharded run generates its own share), column c at msa + c * ld
static __global__ __launch_bounds__(256) void k_synth(SynthArgs const A, uint8_t *__restrict__ msa, size_t ld, uint32_t m, uint64_t c_begin, uint64_t c_end, uint32_t bsh)
{
	uint32_t const words_per_col = (uint32_t) (ld / 4);
	uint32_t const w = blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= words_per_col) return;
	for (uint64_t c = c_begin + blockIdx.y; c < c_end; c += gridDim.y)
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
static __global__ __launch_bounds__(256) void k_presence(uint8_t const *__restrict__ raw, size_t total, uint32_t *__restrict__ present /* 8 words */)
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

// Largest symbol code in borrowed device columns (fseq_set_device_columns*: the caller promises codes < sigma; a
// larger code would silently lose its high digits in the 2-bit digit passes).  bits per code = 8 >> bsh.
static __global__ __launch_bounds__(256) void k_max_code(uint8_t const *__restrict__ msa, size_t ld, uint32_t col_bytes, uint64_t ncols, uint32_t bsh,
                                                  uint32_t tail_rows, uint32_t *__restrict__ out)
{
	uint32_t const bits = 8u >> bsh, smask = (1u << bits) - 1u;
	uint32_t mx = 0;
	for (uint64_t c = blockIdx.x; c < ncols; c += gridDim.x)
		for (uint32_t b = threadIdx.x; b < col_bytes; b += 256u)
		{
			uint32_t const v = msa[c * ld + b];
			// the last byte of a column may hold fewer rows than it has room for (the rest is padding)
			uint32_t const rows = (b + 1u == col_bytes && tail_rows) ? tail_rows : (1u << bsh);
			for (uint32_t j = 0; j < rows; ++j) mx = max(mx, (v >> (j * bits)) & smask);
		}
	mx = max(mx, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, mx));
	mx = max(mx, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, mx));
	mx = max(mx, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, mx));
	mx = max(mx, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, mx));
	mx = max(mx, dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, mx));
	mx = max(mx, dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, mx));
	if (lane_id() == 63) atomicMax(out, mx);
}

// The codes present in every column of 4-bit symbols (bsh = 1): bit c of out[k] = some row carries code c in column k.  A
// property of the input, computed once per input (not per run).  [r4] k_columns takes a column with at most four present
// codes in ONE digit pass over the codes' ranks among the present ones (the remap keeps their order, so the stable sort -- and
// every divergence -- is the one the two 2-bit passes produce): rare on the survey's generator, common in real gapped
// alignments.  A padding nibble behind row m - 1 counts as code 0: a superset of the present codes is as good.
static __global__ __launch_bounds__(256) void k_column_presence(uint8_t const *__restrict__ msa, size_t ld, uint32_t col_bytes, uint64_t c_lo, uint64_t c_hi,
                                                         uint32_t *__restrict__ out, uint32_t *__restrict__ n_dense /* += columns with <= 4 codes */)
{
	__shared__ uint32_t part[4];
	for (uint64_t c = c_lo + blockIdx.x; c < c_hi; c += gridDim.x)
	{
		uint32_t pm = 0;
		for (uint32_t b = threadIdx.x * 4u; b < col_bytes; b += 1024u)
		{
			uint32_t const w = *reinterpret_cast<uint32_t const *>(msa + c * ld + b);      // (columns are padded to 16 bytes)
			uint32_t const nv = min(4u, col_bytes - b);
#pragma unroll
			for (uint32_t j = 0; j < 8; ++j)
				if (j < 2u * nv) pm |= 1u << ((w >> (4u * j)) & 15u);
		}
		pm |= dpp_mov<DPP_ROW_SHR1, 0xF>(0u, pm);
		pm |= dpp_mov<DPP_ROW_SHR2, 0xF>(0u, pm);
		pm |= dpp_mov<DPP_ROW_SHR4, 0xF>(0u, pm);
		pm |= dpp_mov<DPP_ROW_SHR8, 0xF>(0u, pm);
		pm |= dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, pm);
		pm |= dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, pm);
		if (lane_id() == 63) part[wave_id()] = pm;
		__syncthreads();
		if (threadIdx.x == 0)
		{
			uint32_t const all = part[0] | part[1] | part[2] | part[3];
			out[c] = all;
			if (__popc(all) <= 4) atomicAdd(n_dense, 1u);
		}
		__syncthreads();
	}
}

struct CodeTable { uint8_t code_of[256]; };

// ... and the encode + transpose + pack: raw[r * n + c] -> code_of[byte] at row r of packed column c; 64 x 64
// tiles through LDS, reads coalesced along a row, writes down a column (64 >> bsh bytes per tile column).
static __global__ __launch_bounds__(256) void k_encode_transpose(
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
__device__ __forceinline__ void read_chunk(PA const *a_l, PD const *d_l, uint32_t (&a)[E], uint32_t (&d)[E], uint32_t p0 = 0xFFFFFFFFu)
{
	if (p0 == 0xFFFFFFFFu) p0 = threadIdx.x * E;
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
__device__ __forceinline__ void colblock_body(char *smem,
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblocks, uint32_t npass, uint32_t bsh,
	// MODE_RANK outputs
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys,
	// MODE_SNAP inputs / outputs
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint64_t const *__restrict__ task_rb, uint2 const *__restrict__ task_grp,
	uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d,
	uint64_t const *__restrict__ task_src, uint32_t snap_stride, uint32_t const *__restrict__ ss_a, uint32_t const *__restrict__ ss_d,
	uint64_t col0, uint32_t keyed)
{
	// keyed != 0: every divergence this launch can see is below 2^scan_shift_for(T, E) (the host knows: n, or B for a
	// rank block) -- the partition step scans keys (fseq_core.hpp)
	// MODE_RANK: workgroup i owns the block of columns starting at col0 + i * B (col0: first column of this
	// launch -- a rank of a sharded run owns a contiguous block range); rank / keyd / nkeys are indexed by i
	constexpr uint32_t CAP = T * E;
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
		k0 = col0 + (uint64_t) blockIdx.x * B;
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
			uint32_t idx = tid + i * T;
			asm volatile("" : "+v"(idx));              // (opaque: no addresses kept -- and spilled -- across the loops)
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
				uint32_t idx = tid + i * T;
				asm volatile("" : "+v"(idx));              // (opaque: no store addresses kept -- and spilled -- across the loops)
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

			// keyed: 1 = row counts in the keys (values below 2^scan_shift_for(T, E)), 2 = occurrence counts (below 2^25), 0 = has-based scan
			if (keyed == 1u) partition_step<T, E, SIGMA, false, false, scan_shift_for(T, E)>(d, s, (uint32_t) (k0 + j + 1) - dbase, scr, dst, dnew);
			else if (SIGMA == 4 && keyed == 2u) partition_step<T, E, 4, false, false, 25, false, false, true>(d, s, (uint32_t) (k0 + j + 1) - dbase, scr, dst, dnew);
			else partition_step<T, E, SIGMA>(d, s, (uint32_t) (k0 + j + 1) - dbase, scr, dst, dnew);

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

// (pass 2 on 512 threads at 80 registers -- 81 otherwise --: three workgroups of this latency-bound kernel share a CU,
// BASELINE C3 0.35 -> 0.29 ms; the same bound on k_chain cost more in spills than it brought, 0.71 -> 0.77 ms)
template <int T, int E, int SIGMA, int MODE, bool PK>
__global__ __launch_bounds__(T, (T == 512 && MODE == MODE_SNAP) ? 6 : 1) void k_colblock(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblocks, uint32_t npass, uint32_t bsh,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint64_t const *__restrict__ task_rb, uint2 const *__restrict__ task_grp,
	uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d,
	uint64_t const *__restrict__ task_src, uint32_t snap_stride, uint32_t const *__restrict__ ss_a, uint32_t const *__restrict__ ss_d,
	uint64_t col0, uint32_t keyed)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	// MODE_RANK has no start state: bstate_a, if given, is a per-block filter -- the workgroups of blocks whose word is zero
	// have nothing to do (phase A: the blocks the key-space tree gave up on, fseq_blockkeys.hpp)
	if (MODE == MODE_RANK && bstate_a && bstate_a[blockIdx.x] == 0u) return;
	colblock_body<T, E, SIGMA, MODE, PK>(smem, msa, ld, m, n, B, nblocks, npass, bsh, rank, keyd, nkeys, bstate_a, bstate_d, task_rb, task_grp,
	                                     snap_a, snap_d, task_src, snap_stride, ss_a, ss_d, col0, keyed);
}

// ------------------------------------------------------------------------------------------------
// Phase B: the serial chain over blocks (one workgroup).
// ------------------------------------------------------------------------------------------------
template <int T, int E, bool PK>
__host__ __device__ inline size_t chain_lds_bytes()
{
	constexpr size_t CAP = (size_t) T * E;
	return 2 * carve_bytes(CAP, PK ? 2 : 4) + 2 * carve_bytes(CAP, 4) + carve_bytes(1, sizeof(StepScratch<T, 4>)) + carve_bytes(T / WAVE + 1, 4) + carve_bytes(WAVE, 4);
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
	uint32_t *__restrict__ out_rank, uint32_t *__restrict__ out_keyd, uint32_t *__restrict__ out_nkeys, uint32_t grp0, uint32_t keyed)
{
	// keyed: 1 = all divergences are below 2^scan_shift_for(T, E) (n is), 2 = below 2^25: the partition steps scan keys
	// workgroup i of the launch is chain grp = grp0 + i (a rank of a sharded run owns a contiguous range of chains)
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
	uint32_t *sink = cv.take<uint32_t>(WAVE);                  // where the L2-warming loads land (never read)

	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = tid * E;
	uint32_t const grp = blockIdx.x + grp0;
	uint32_t const b0 = grp * G;
	uint32_t const b1 = min(nb_total, b0 + G);
	uint32_t const kstart = (uint32_t) ((uint64_t) b0 * cols_per_block);
	{
		size_t const sb = (size_t) grp * m, rb = (size_t) b0 * m;
#pragma unroll
		for (int i = 0; i < E; ++i)
		{
			uint32_t idx = tid + i * T;
			asm volatile("" : "+v"(idx));              // (opaque: no addresses kept -- and spilled -- across the loops)
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
				uint32_t idx = tid + i * T;
				asm volatile("" : "+v"(idx));              // (opaque: no store addresses kept -- and spilled -- across the loops)
				if (idx < m) { out_state_a[ob + idx] = a_l[idx]; out_state_d[ob + idx] = d_l[idx]; }
			}
		}
		// [r5] the state behind a chain's last block is the start state of the next chain, which the level above has given it:
		// an expansion (states wanted, no composite keys) does not step through its last block unless that is the last of all
		if (b + 1 == b1 && !out_rank && b1 != nb_total) break;
		// (2) prefetch the next block's rank / keyd into registers
		// (16-bit configurations, 9+ rows per thread at 128 registers: the 2 E prefetched words were spilled to scratch
		// across the digit passes -- there the loads only warm L2 now and are issued again in (5))
		constexpr bool PREF = !PK;
		uint32_t pr[E], pk[E], D_next = 0;
		bool const more = b + 1 < b1;
		if (more)
		{
			size_t const nbase = (size_t) (b + 1) * m;
			if constexpr (PREF)
			{
#pragma unroll
				for (int i = 0; i < E; ++i)
				{
					uint32_t idx = tid + i * T;
					asm volatile("" : "+v"(idx));              // (opaque: no addresses kept -- and spilled -- across the loops)
					pr[i] = idx < m ? rank[nbase + idx] : 0u;
					pk[i] = idx < m ? keyd[nbase + idx] : 0u;
				}
			}
			else
			{
				uint32_t const sink_addr = __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) sink);
#pragma unroll
				for (int i = 0; i < E; ++i)
				{
					uint32_t const idx = min(tid + i * T, m - 1u);
					stream_touch(rank + nbase + idx, sink_addr);
					stream_touch(keyd + nbase + idx, sink_addr);
				}
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
			if (keyed == 1u) partition_step<T, E, 4, false, false, scan_shift_for(T, E)>(d, s, 0u, scr, dst, dnew);
			else if (keyed == 2u) partition_step<T, E, 4, false, false, 25, false, false, true>(d, s, 0u, scr, dst, dnew);
			else partition_step<T, E, 4>(d, s, 0u, scr, dst, dnew);
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
			size_t const nbase = (size_t) (b + 1) * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t idx = tid + i * T;
				asm volatile("" : "+v"(idx));              // (opaque: no store addresses kept -- and spilled -- across the loops)
				if constexpr (!PREF)
				{
					pr[i] = idx < m ? rank[nbase + idx] : 0u;
					pk[i] = idx < m ? keyd[nbase + idx] : 0u;
				}
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
			uint32_t idx = tid + i * T;
			asm volatile("" : "+v"(idx));              // (opaque: no store addresses kept -- and spilled -- across the loops)
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
		size_t const ob = (size_t) grp * m;
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
		if (tid == 0) out_nkeys[grp] = total;
	}
}

// ------------------------------------------------------------------------------------------------
// Phase C: true per-column updates inside a block + per-column top of the divergence histogram.
// Divergence values are replaced by order-preserving ids (sorted distinct boundary values first,
// then one new id per column), so the histogram is a dense LDS table updated incrementally.
// ------------------------------------------------------------------------------------------------
// phase C's partition step with the pairwise local pass (fseq_core.hpp, PW): four symbols, and few enough rows per
// thread that E (E - 1) / 2 compares stay below the 4 E of the running maxima with room to spare
#ifndef FSEQ_PW_MAX_E
#define FSEQ_PW_MAX_E 8
#endif
// (not the 1024 x 7 configuration: its 20 KiB of run slots would push the largest block lengths past the 160 KiB of a CU)
// (nine and ten rows of the 16-bit configurations: the two-level form, fseq_core.hpp)
__host__ __device__ constexpr bool columns_pairwise(int T, int E, int SIGMA, bool PK = false)
{
	return SIGMA == 4 && ((E >= 2 && E <= FSEQ_PW_MAX_E && !(T >= 1024 && E >= 7 && !PK)) || (PK && E >= 9 && T * E <= 10240));
}

// the resolve of a step as a read of per-thread run slots (partition_step's FM): wherever 4 * T more words of LDS are to be had
__host__ __device__ constexpr bool columns_lookup(int T, int E, int SIGMA, bool PK) { return columns_pairwise(T, E, SIGMA, PK) || (SIGMA == 4 && PK && T * E <= 10240); }
__host__ __device__ constexpr size_t columns_run_words(int T, int E, int SIGMA, bool PK) { return columns_pairwise(T, E, SIGMA, PK) ? 5 * (size_t) T : columns_lookup(T, E, SIGMA, PK) ? 4 * (size_t) T : 0; }

template <int T, int E, int SIGMA, bool PK>
__host__ __device__ inline size_t columns_lds_bytes(uint32_t B)
{
	constexpr size_t CAP = (size_t) T * E;
	return 2 * carve_bytes(CAP, PK ? 2 : 4) + 2 * carve_bytes(CAP, 1) + carve_bytes(PK ? (CAP + B + 9) / 2 : CAP + B + 8, 4) + carve_bytes(CAP, 4)
	     + carve_bytes(1, sizeof(StepScratch<T, SIGMA>)) + carve_bytes(T / WAVE + 1, 4) + (columns_run_words(T, E, SIGMA, PK) ? carve_bytes(columns_run_words(T, E, SIGMA, PK), 4) : 0);
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
		for (uint32_t j = k >> 1, lj = (uint32_t) __builtin_ctz(k) - 1u; j > 0; j >>= 1, --lj)
		{
			for (uint32_t i = threadIdx.x; i < N2 / 2; i += T)
			{
				uint32_t const lo = ((i >> lj) << (lj + 1u)) | (i & (j - 1u));      // (j = 2^lj: no division by a run-time j)
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
// EW ("emitter wave"): wave 0 owns no rows -- thread t >= 64 owns the positions (t - 64) * E .. -- and spends its
// time on the per-column list alone.  Measured on BASELINE C3: with the list on a wave that also carries 1/8 of the
// rows, the list's serial chain (~150 instructions under 8-way issue contention) stretched EVERY column's critical
// path: 2.9 of 10.8 ms.  Needs m <= (T - 64) * E.
// DENSE (4-bit symbols): a column with at most four present codes takes ONE digit pass (colmask: k_column_presence).  A kernel of
// its own -- compiled into the one every input runs, the three uniform branches cost BASELINE C5 2.6 ms and C3 0.14 ms --
// that the host launches when at least one column in twenty is that dense.
// RED [r5] (fseq_reduced.hpp): the block's REPRESENTATIVE rows only -- one row of every class of rows that agree from column
// vmin - 1 to the end of the block, the start state restricted to them (k_reduce_prep), the packed columns of those rows
// (k_reduce_msa: msa / ld are the reduced alignment's).  Every count of values >= v is then the full run's for every
// v >= vmin, so the lists are the full run's as long as no entry below vmin is taken; a list that takes one flags the block
// (red.invalid) and the block is run again on all rows.  vmin == 1: the rows left out are duplicates over all of [0, k1) --
// their divergences are zeros, added where the zeros are counted -- and every list is exact.  With red.cls set (pass 2) the
// workgroup emits, instead of lists, the class tables at its task columns: the class of every block key (rank among the
// distinct key prefixes up to that column) and the divergence in front of every class -- what one chain step from the
// block's boundary state needs (k_chain_snap).
template <int T, int E, int SIGMA, bool PK, bool EW, bool DENSE, bool RED>
__device__ __forceinline__ void columns_body(char *smem,
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t N2,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint32_t L, uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr, uint32_t npass, uint32_t bsh,
	uint32_t snap_stride, uint32_t *__restrict__ ss_a, uint32_t *__restrict__ ss_d, uint32_t block0,
	uint32_t *done_host, uint32_t epoch, uint32_t const *__restrict__ colmask, RedArgs const &red, uint32_t const *__restrict__ blocklist = nullptr)
{
	// blocklist (not RED) [r5]: workgroup i owns block blocklist[i] -- the blocks the reduced phase C hands to the run on all rows
	// colmask (4-bit symbols, or nullptr): the codes present in every column (k_column_presence)
	// workgroup i of the launch owns column block block0 + i (phase C may be launched in several parts)
	FSEQ_CLOCK_STAMP(blockIdx.x, 0);
	constexpr uint32_t CAP = T * E;
	uint32_t const blk = RED ? red.blocks[blockIdx.x] : (blocklist ? blocklist[blockIdx.x] : blockIdx.x + block0);
	uint32_t red_vmin = 0, red_deficit = 0, t_first = 0, t_count = 0, t_next = 0;
	bool const red_snap = RED && red.cls != nullptr;
	if constexpr (RED)
	{
		m = red.cnt[blk];
		if (m == RED_NONE || m > (EW ? (uint32_t) (T - 64) * E : CAP)) return;      // (never listed: a block the host runs on all rows)
		red_vmin = red.vmin[blk];
		red_deficit = red.m_true - m;
		N2 = 2; while (N2 < m + 1u) N2 <<= 1;
		if (red_snap) { t_first = red.wg_tasks[3u * blockIdx.x]; t_count = red.wg_tasks[3u * blockIdx.x + 1u]; }
	}
	bool const red_exact = RED && red_vmin <= 1u;
	// [a_l][d_l][sym0][sym1][cnt_l] are contiguous: the prologue's sort buffer (N2 < 2m words) overlays them
	using AT = std::conditional_t<PK, uint16_t, uint32_t>;     // row ids < m, value ids < m + B
	Carver cv{smem};
	AT *a_l = cv.take<AT>(CAP);
	AT *d_l = cv.take<AT>(CAP);
	uint8_t *sym0 = cv.take<uint8_t>(RED ? red.symcap : CAP);
	uint8_t *sym1 = cv.take<uint8_t>(RED ? red.symcap : CAP);
	uint32_t *cnt_l = cv.take<uint32_t>(PK ? (CAP + B + 9) / 2 : CAP + B + 8);
	uint32_t *V_l = cv.take<uint32_t>(CAP);
	StepScratch<T, SIGMA> &scr = *cv.take<StepScratch<T, SIGMA>>(1);
	uint32_t *sscr = cv.take<uint32_t>(T / WAVE + 1);
	constexpr bool PW = columns_pairwise(T, E, SIGMA, PK);
	constexpr bool LU = columns_lookup(T, E, SIGMA, PK);
	uint32_t *runs = LU ? cv.take<uint32_t>(columns_run_words(T, E, SIGMA, PK)) : nullptr;
	uint32_t const tid = threadIdx.x;
	uint32_t const p0 = EW ? (tid >= 64u ? (tid - 64u) * E : 0x7FFF0000u) : tid * E;     // 0x7FFF0000: owns nothing (every p0 + e >= m)
	bool const rows = !EW || tid >= 64u;
	// (pass 2 on the representatives: the sweep starts at the block's first column or at one of the states phase C dropped)
	uint64_t const kblk = (uint64_t) blk * B;
	uint64_t const k0 = red_snap ? (uint64_t) red.wg_tasks[3u * blockIdx.x + 2u] : kblk;
	uint64_t const kend = red_snap ? (uint64_t) red.task_rb[t_first + t_count - 1u] : ((k0 + B < n) ? k0 + B : n);
	uint32_t const nb = (uint32_t) (kend - k0);

	// ---- prologue
	// boundary state straight into registers (chunk ownership: positions tid*E .. tid*E+E-1)
	uint32_t a[E], d[E];
	{
		bool const from_ss = RED && k0 != kblk;
		size_t const ob = from_ss ? (size_t) (k0 / red.ss_stride) * red.ss_cap : RED ? (size_t) blk * red.cap : (size_t) blk * m;
		uint32_t const *const sa = from_ss ? red.ss_a : RED ? red.a : bstate_a, *const sd = from_ss ? red.ss_d : RED ? red.d : bstate_d;
#pragma unroll
		for (int e = 0; e < E; ++e)
		{
			uint32_t const pos = p0 + e;
			a[e] = pos < m ? sa[ob + pos] : 0u;
			d[e] = pos < m ? sd[ob + pos] : 0u;
		}
	}

	// ---- sorted distinct divergence values -> V_l[0..D0)
	uint32_t *sb = reinterpret_cast<uint32_t *>(a_l);   // N2 < 2m words, overlays a_l .. cnt_l
	uint32_t D0 = 0;
	bool have_values = false;
	if constexpr (RED)
	{
		// [r5] the distinct values first, by hashing, then the sort of THOSE: a block's representatives hold a few thousand rows
		// and far fewer values (BASELINE C4: ~1,750 among ~6,600), and the bitonic sort of all rows' values -- 91 steps over 8,192
		// words, ~150,000 cycles -- was nearly half of what a class-table task of pass 2 costs (its ~16 columns are ~170,000).
		// sb is the table (up to N2 slots, open addressing; a value that finds no slot within 64 probes, or more distinct
		// values than V_l sorts in place, takes the sort of all values below).
		uint32_t &hash_fail = scr.has[0];                       // (the step's scratch is idle in the prologue)
		// (the table ends in front of V_l, which the distinct values are compacted into while the table is read: the sort buffer
		// of N2 words may reach into V_l in the 16-bit configurations -- harmless there, its tail is padding by then)
		uint32_t const avail = (uint32_t) ((reinterpret_cast<char *>(V_l) - reinterpret_cast<char *>(sb)) / 4);
		uint32_t NT = N2;
		while (NT > avail) NT >>= 1;
		uint32_t const hshift = 32u - (uint32_t) __builtin_ctz(NT);
		for (uint32_t i = tid; i < NT; i += T) sb[i] = PAD_KEY;
		if (tid == 0) hash_fail = 0u;
		__syncthreads();
		auto insert = [&](uint32_t v) {
			uint32_t h = (v * 2654435761u) >> hshift;
			for (uint32_t probes = 0; probes < 64u; ++probes)
			{
				uint32_t const old = atomicCAS(&sb[h], PAD_KEY, v);
				if (old == PAD_KEY || old == v) return;
				h = (h + 1u) & (NT - 1u);
			}
			hash_fail = 1u;
		};
#pragma unroll
		for (int e = 0; e < E; ++e)
			if (p0 + e < m) insert(d[e]);
		if (red_exact && tid == 0) insert(0u);                   // the value of the rows left out always has an id: 0
		__syncthreads();
		if (hash_fail == 0u)
		{
			uint32_t const per = (NT + T - 1) / T, start = tid * per;
			uint32_t mine = 0;
			for (uint32_t q = 0; q < per; ++q) mine += (start + q < NT && sb[start + q] != PAD_KEY) ? 1u : 0u;
			uint32_t w = block_excl_add<T>(mine, sscr, &D0);
			uint32_t N2s = 2;
			while (N2s < D0) N2s <<= 1;
			if (N2s <= CAP)
			{
				for (uint32_t q = 0; q < per; ++q)
					if (start + q < NT && sb[start + q] != PAD_KEY) V_l[w++] = sb[start + q];
				for (uint32_t i = D0 + tid; i < N2s; i += T) V_l[i] = PAD_KEY;
				__syncthreads();
				bitonic_sort_lds<T>(V_l, N2s);
				have_values = true;
			}
		}
		__syncthreads();
	}
	if (!have_values)
	{
	for (uint32_t i = tid; i < N2; i += T) sb[i] = PAD_KEY;
	__syncthreads();
#pragma unroll
	for (int e = 0; e < E; ++e)
		if (p0 + e < m) sb[p0 + e] = d[e];
	if (RED && red_exact && tid == 0) sb[m] = 0u;               // the value of the rows left out always has an id: 0
	__syncthreads();
	bitonic_sort_lds<T>(sb, N2);
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
	}
	__syncthreads();                          // the sort buffer (overlaying cnt_l) is dead from here
	for (uint32_t i = tid; i < (PK ? (D0 + nb + 9u) / 2u : D0 + nb + 8u); i += T) cnt_l[i] = 0;     // (+8: the list reads aligned groups of ids)
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
	for (int e = 0; e < E; ++e)
		if (rows) { a_l[p0 + e] = (AT) a[e]; d_l[p0 + e] = (AT) id[e]; }

	bool const has_chunk = !RED && tid * 16u < sym_bytes(m, bsh);
	uint4 nxt = make_uint4(0, 0, 0, 0);
	if (nb && has_chunk)
	{
		nxt = *reinterpret_cast<uint4 const *>(msa + k0 * ld + tid * 16u);
		*reinterpret_cast<uint4 *>(sym0 + tid * 16u) = nxt;
	}
	// RED: the packed column straight into LDS (LDS-DMA, no registers: a column of the alignment itself may be several
	// 16-byte pieces per thread); every lane loads -- a lane past the column's bytes re-reads its first piece into the
	// buffer's padding
	uint32_t const red_colbytes = RED ? (red.direct ? red.colbytes : sym_bytes(m, bsh)) : 0u;
	auto red_stage = [&](uint64_t k, uint8_t *dstb) {
		uint8_t const *const col = msa + k * ld;
		for (uint32_t cb = 0; cb < red_colbytes; cb += (uint32_t) T * 16u)
		{
			uint32_t const off = cb + tid * 16u;
			if (cb + wave_id() * 1024u < red_colbytes)               // (wave-uniform: the buffers end with the last wave's kilobyte that holds bytes)
				lds_dma16(col + (off < red_colbytes ? off : 0u), __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) (dstb + cb + wave_id() * 1024u)));
		}
	};
	if (RED && nb) { red_stage(k0, sym0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
	__syncthreads();

	bool const zero_present = (V_l[0] == 0u);
	// (see the pass loop; compiled into the nine-to-eleven-row kernels only -- the 512-thread 16-bit kernels have no register to spare)
	constexpr bool CARRY_OK = PK && E >= 9;
	bool const carry = CARRY_OK && npass == 2u && bsh == 1u && ((RED && red.direct) ? red.m_true : m) <= 16384u;
#ifdef FSEQ_KC_STAMPS
	long long kc_work = 0, kc_wait = 0, kc_last = clock64();
	KcStamps kcs{{0, 0, 0, 0, 0, 0, 0, 0}, clock64()};
#endif

	for (uint32_t j = 0; j < nb; ++j)
	{
		uint8_t const *symc = (j & 1u) ? sym1 : sym0;
		uint8_t *symn = (j & 1u) ? sym0 : sym1;
		bool const more = j + 1 < nb;
		if (more && has_chunk)
			nxt = *reinterpret_cast<uint4 const *>(msa + (k0 + j + 1) * ld + tid * 16u);
		if (RED && more) red_stage(k0 + j + 1, symn);
		// [r4] at most four codes present in a column of 4-bit symbols: ONE pass over their ranks among the present codes
		// (remap: two bits per code)
		bool dense = false;
		uint32_t remap = 0;
		if (DENSE && colmask && npass == 2u && bsh == 1u)
		{
			uint32_t const pm = __builtin_amdgcn_readfirstlane(colmask[k0 + j]);
			if (__popc(pm) <= 4)
			{
				dense = true;
				uint32_t idx = 0;
#pragma unroll
				for (uint32_t cde = 0; cde < 16; ++cde) { remap |= (idx & 3u) << (2u * cde); idx += (pm >> cde) & 1u; }
			}
		}

		for (uint32_t pass = 0; pass < npass; ++pass)
		{
			bool const last_pass = (DENSE && dense) || pass + 1 == npass;
			uint32_t s[E], dst[E], dnew[E], hi[CARRY_OK ? E : 1];
			if (rows) read_chunk<T, E>(a_l, d_l, a, d, p0);
			// 16-bit row ids of at most 14 bits and a 4-bit alphabet (two digit passes): the first pass fetches the whole symbol
			// and the row carries its second digit in the two spare bits of its id through the scatter, so the second pass
			// reads its digit where it reads the row -- half of the byte gathers at random rows (and their bank conflicts) gone
			if (carry && pass == 0)
			{
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					uint32_t const sy = (symc[a[e] >> 1] >> ((a[e] & 1u) * 4u)) & 15u;
					s[e] = (p0 + e < m) ? ((DENSE && dense) ? ((remap >> (2u * sy)) & 3u) : (sy & 3u)) : (uint32_t) SIGMA;
					hi[e] = (DENSE && dense) ? 0u : (sy >> 2);
				}
			}
			else if (DENSE && dense)
			{
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					uint32_t const sy = (symc[a[e] >> 1] >> ((a[e] & 1u) * 4u)) & 15u;
					s[e] = (p0 + e < m) ? ((remap >> (2u * sy)) & 3u) : (uint32_t) SIGMA;
					if (CARRY_OK) hi[e] = 0;
				}
			}
			else if (carry)
			{
#pragma unroll
				for (int e = 0; e < E; ++e)
				{
					s[e] = (p0 + e < m) ? (a[e] >> 14) : (uint32_t) SIGMA;
					a[e] &= 0x3FFFu;
					hi[e] = 0;
				}
			}
			else
			{
#pragma unroll
				for (int e = 0; e < E; ++e) { s[e] = (p0 + e < m) ? sym_digit(symc, a[e], bsh, pass) : (uint32_t) SIGMA; if (CARRY_OK) hi[e] = 0; }
			}

			// (value ids: D0 + nb <= m + B < 65536 -- the LDS of the id histogram bounds B long before -- so the keyed scan)
#ifdef FSEQ_KC_STAMPS
			kcs.last = clock64();
			partition_step<T, E, SIGMA, false, EW, 16, PW, LU>(d, s, D0 + j, scr, dst, dnew, nullptr, runs, &kcs);
#else
			partition_step<T, E, SIGMA, false, EW, 16, PW, LU>(d, s, D0 + j, scr, dst, dnew, nullptr, runs);
#endif

#pragma unroll
			for (int e = 0; e < E; ++e)
			{
				if (p0 + e < m)
				{
					a_l[dst[e]] = (AT) (CARRY_OK ? (a[e] | (hi[CARRY_OK ? e : 0] << 14)) : a[e]);
					d_l[dst[e]] = (AT) dnew[e];
					if (dnew[e] != d[e])
					{
						cnt_dec<PK>(cnt_l, d[e]);
						cnt_inc<PK>(cnt_l, dnew[e]);
					}
				}
			}
			if (last_pass && more && has_chunk)
				*reinterpret_cast<uint4 *>(symn + tid * 16u) = nxt;
			if (RED && last_pass && more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next column has landed
#ifdef FSEQ_KC_STAMPS
			{ long long const t_ = clock64(); kc_work += t_ - kc_last; kc_last = t_; }
			__syncthreads();
			{ long long const t_ = clock64(); kc_wait += t_ - kc_last; kc_last = t_; }
#else
			__syncthreads();
#endif
			if (DENSE && dense) break;
		}

		// ---- every snap_stride columns: drop the exact (a, d) for pass 2 (ids back to divergence values)
		if (ss_a && (k0 + j + 1) % snap_stride == 0)
		{
			size_t const ob = (size_t) ((k0 + j + 1) / snap_stride) * m;
#pragma unroll
			for (int i = 0; i < E; ++i)
			{
				uint32_t idx = tid + i * T;
				// (opaque: left alone, the compiler keeps the 2 E store addresses of this rare path in registers across the whole
				// column loop -- and spills them: 6 GB of scratch traffic per C5 step at ten rows per thread)
				asm volatile("" : "+v"(idx));
				if (idx < m)
				{
					uint32_t const vid = d_l[idx];
					ss_a[ob + idx] = a_l[idx];
					ss_d[ob + idx] = vid < D0 ? V_l[vid] : (uint32_t) (k0 + (vid - D0) + 1u);
				}
			}
		}

		// ---- [r5] the reduced state every ss_stride columns (the columns strictly inside the block): pass 2's sweeps start there
		if (RED && !red_snap && red.ss_a && (k0 + j + 1u) % red.ss_stride == 0u && j + 1u < nb)
		{
			size_t const ob = (size_t) ((k0 + j + 1u) / red.ss_stride) * red.ss_cap;
			for (uint32_t idx = tid; idx < m; idx += T)
			{
				uint32_t const vid = d_l[idx];
				red.ss_a[ob + idx] = a_l[idx];
				red.ss_d[ob + idx] = vid < D0 ? V_l[vid] : (uint32_t) (k0 + (vid - D0) + 1u);
			}
		}
		// ---- [r5] pass 2 on the representatives: the class tables at a task column.  A position starts a class iff its
		// divergence lies inside the block (> the block's first column); the class of a block key is the class of any
		// representative that carries it.
		if (RED && red_snap)
		{
			if (t_next < t_count && red.task_rb[t_first + t_next] == k0 + j + 1u)
			{
				uint32_t const task = t_first + t_next;
				// (ids of this sweep's own columns are inside the block; of its start values those that are -- V_l ascends -- from id_in on)
				uint32_t id_in = D0;
				if (k0 != kblk)
				{
					uint32_t lo = 0, hi = D0;
					while (lo < hi) { uint32_t const mid = (lo + hi) >> 1; if (V_l[mid] <= (uint32_t) kblk) lo = mid + 1; else hi = mid; }
					id_in = lo;
				}
				uint32_t nf = 0;
				if (rows)
				{
#pragma unroll
					for (int e = 0; e < E; ++e) nf += (p0 + e < m && (uint32_t) d_l[p0 + e] >= id_in) ? 1u : 0u;
				}
				uint32_t total;
				uint32_t r = block_excl_add<T>(nf, sscr, &total);
				size_t const ot = (size_t) task * red.cap, ol = (size_t) blk * red.cap;
				if (rows)
				{
#pragma unroll
					for (int e = 0; e < E; ++e)
					{
						uint32_t const pos = p0 + e;
						if (pos < m)
						{
							uint32_t const vid = d_l[pos];
							bool const first = vid >= id_in;
							r += first ? 1u : 0u;
							uint32_t const lf = red.direct ? red.rank[(size_t) blk * red.m_true + a_l[pos]] : red.leaf[ol + a_l[pos]];
							red.cls[ot + lf] = r - 1u;
							if (first) red.headd[ot + r - 1u] = vid < D0 ? V_l[vid] : (uint32_t) (k0 + (vid - D0) + 1u);
						}
					}
				}
				if (tid == 0) red.ncls[task] = total;
				++t_next;
			}
			continue;
		}

		// ---- emit the top of the histogram for column k0+j (wave 0; the others run ahead into
		// the next column and meet it again at the partition step's barrier).
		// Entry 0 lumps every value >= thr = end+1-L (end = k+1): the DP clips all of them to the same
		// cut bound (lp.cc:444-445), so only their total count matters.  Then the distinct values
		// below thr, descending, until their cumulative count exceeds X (every DP cell is >= the lump's
		// count, so the list always reaches X counts past the smallest value the cell can take).
		if (wave_id() == 0)
		{
			// The list is one wave's serial chain (LDS reads -> scan -> ballots -> stores) on every column's critical
			// path: issue priority over the other waves of the SIMD, and four ids per lane (256 per step) -- the id
			// space is sparse (one id per column and boundary value, most of them with count 0 by now), so a step of
			// 64 ids needed 5-8 rounds of that chain per column, 256 need 1-2.
			__builtin_amdgcn_s_setprio(3);
#ifdef FSEQ_KC_STAMPS
			long long const kc_l0 = clock64();
#endif
			uint32_t const lane = lane_id();
			uint64_t const k = k0 + j;
			uint32_t const thr = (k + 2 > (uint64_t) L) ? (uint32_t) (k + 2 - L) : 0u;
			uint2 *out = ent + k * (size_t) stride;
			constexpr int IPL = (RED && T >= 256) ? 8 : 4;
			int32_t const top = (int32_t) (D0 + j);
			uint32_t cumN = 0, nent = 1, R = 0;       // cumN: count of the values below thr taken so far
			// RED: exact -- id 0 (value 0, with the rows left out) follows behind the loop; else the ids below red_idmin are values
			// below vmin, which the representatives cannot vouch for
			int32_t const id_lo = (RED && red_exact) ? 1 : 0;
			bool red_stopped = false, red_bad = false;
			// [r5] IPL ids per lane and round: eight for the reduced configurations of 256 threads and more -- their lists are hundreds
			// of entries long (BASELINE C4: ~280 of ~500 ids visited per column), and the fixed part of a round (two scans, the ballots, the
			// maximum) was paid twice a column by the one wave every row wave then waits for
			// ids in aligned groups of IPL per lane, read as 8- and 16-byte LDS words: with one 4-byte read per id the lanes'
			// addresses were IPL words apart -- an IPL-way bank conflict on every read of the one wave each column waits for
			// (BASELINE C4: 3,900 of the list's 6,100 cycles per column).  The ids above top (at most IPL - 1) have count 0:
			// cnt_l is zeroed that far.
			for (int32_t base = top | (IPL - 1); base >= id_lo; base -= 64 * IPL)
			{
				// lane l holds the ids base - IPL l - q, q = 0 .. IPL - 1: descending ids = descending values, lane-major
				uint32_t c[IPL], v[IPL];
				uint32_t lane_c = 0, lane_o = 0, lane_rc = 0;
				bool cand[IPL];
#ifdef FSEQ_KC_STAMPS
				kcs.acc[4] += 1;
#endif
				int32_t const hi = base - IPL * (int32_t) lane;     // = IPL - 1 (mod IPL): the group is [lo, hi], all of it >= 0 or none
				int32_t const lo = hi - (IPL - 1);
				uint32_t cw[IPL], vw[IPL];
#pragma unroll
				for (int q = 0; q < IPL; ++q) { cw[q] = 0; vw[q] = 0; }
				if (hi >= 0)
				{
					if constexpr (PK)
					{
						uint32_t w[IPL / 2];
						lds_read_words<IPL / 2>(cnt_l + ((uint32_t) lo >> 1), w);
#pragma unroll
						for (int q = 0; q < IPL; ++q) cw[q] = (q & 1) ? (w[q >> 1] >> 16) : (w[q >> 1] & 0xFFFFu);
					}
					else
						lds_read_words<IPL>(cnt_l + lo, cw);
					if ((uint32_t) lo < D0) lds_read_words<IPL>(V_l + lo, vw);     // (past D0: words nobody reads)
				}
#pragma unroll
				for (int q = 0; q < IPL; ++q)
				{
					int32_t const i = hi - q;
					c[q] = (i >= id_lo) ? cw[IPL - 1 - q] : 0u;
					v[q] = (i < 0) ? 0u : (((uint32_t) i < D0) ? vw[IPL - 1 - q] : (uint32_t) (k0 + ((uint32_t) i - D0) + 1u));
					bool const nz = c[q] > 0;
					bool const rec = nz && v[q] >= thr;       // the values >= thr are a prefix of the non-zero entries
					cand[q] = nz && !rec;
					lane_c += c[q];
					lane_o += cand[q] ? 1u : 0u;
					lane_rc += rec ? c[q] : 0u;
				}
#ifdef FSEQ_KC_STAMPS
				long long const kc_e1 = clock64();
#endif
				// one scan for the counts (< 2^16 in all: m <= 65535) and the candidate numbers, one for the lump
				uint32_t const inc = wave_incl_add(lane_c | (lane_o << 16));
				uint32_t const r_tot = readlane_u32(wave_incl_add(lane_rc), 63);
				uint32_t const tot_o = readlane_u32(inc, 63) >> 16;
				uint32_t run_c = (inc & 0xFFFFu) - lane_c;   // counts in front of this lane's entries
				uint32_t run_o = (inc >> 16) - lane_o;       // candidates in front of them
				uint32_t lastP = 0;
				bool tk[IPL];
#ifdef FSEQ_KC_STAMPS
				long long const kc_e2 = clock64();
#endif
#pragma unroll
				for (int q = 0; q < IPL; ++q)
				{
					// a candidate is taken while the below-thr counts in front of it do not exceed X
					uint32_t const excN = cumN + run_c - r_tot;
					tk[q] = cand[q] && excN <= X;
					if (tk[q]) { out[nent + run_o] = make_uint2(v[q], c[q]); lastP = excN + c[q]; }
					if (RED && !red_exact) red_bad = red_bad || (tk[q] && v[q] < red_vmin);
					run_o += cand[q] ? 1u : 0u;
					run_c += c[q];
				}
#ifdef FSEQ_KC_STAMPS
				long long const kc_e3 = clock64();
				kcs.acc[0] += kc_e1 - kc_l0; kcs.acc[1] += kc_e2 - kc_e1; kcs.acc[2] += kc_e3 - kc_e2;
#endif
				uint32_t taken = 0;
#pragma unroll
				for (int q = 0; q < IPL; ++q) taken += (uint32_t) __popcll(__ballot(tk[q]));
				// the taken entries are a prefix of the candidates: the largest inclusive count among them is the new cumN
				uint32_t mx = lastP;
				mx = max(mx, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, mx));
				mx = max(mx, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, mx));
				mx = max(mx, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, mx));
				mx = max(mx, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, mx));
				mx = max(mx, dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, mx));
				mx = max(mx, dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, mx));
				nent += taken;
				if (taken) cumN = readlane_u32(mx, 63);
				R += r_tot;
				if (taken != tot_o || cumN > X) { red_stopped = true; break; }
			}
			uint32_t cnt0 = zero_present ? cnt_get<PK>(cnt_l, 0u) : 0u;
			if constexpr (RED)
			{
				if (red_exact)
				{
					// the zeros: those of the representatives and the rows left out (duplicates of a representative over [0, k1))
					cnt0 += red_deficit;
					if (!red_stopped && cnt0)
					{
						if (thr == 0u) R += cnt0;
						else { if (lane == 0) out[nent] = make_uint2(0u, cnt0); ++nent; cumN += cnt0; }
					}
				}
				else
				{
					cnt0 = 0;                                            // (unknown -- and unused: the list is not complete)
					// a taken entry below vmin, or every value taken while rows are left out: the run on all rows would go on
					if (__ballot(red_bad) != 0ull || (!red_stopped && red_deficit != 0u)) { if (lane == 0) { red.invalid[blk] = 1u; red.any_invalid[0] = 1u; } }
				}
			}
			uint32_t const cum = R + cumN;
			if (lane == 0)
			{
				out[0] = make_uint2((uint32_t) (k + 1), R);
				hdr[k] = make_uint4(nent, cnt0, cum == (RED ? red.m_true : m) ? 1u : 0u, cum);
			}
#ifdef FSEQ_KC_STAMPS
			kcs.acc[5] += clock64() - kc_l0;
			kcs.acc[6] += nent;
#endif
			__builtin_amdgcn_s_setprio(0);
		}
	}
#ifdef FSEQ_KC_STAMPS
	if (lane_id() == 0 && (blockIdx.x == 100 || blockIdx.x == 3000))
		printf("kc stamps block %u wave %u: work %lld wait %lld cycles per column (%u columns); step: to barrier 1 %lld, wait %lld, between %lld, wait 1b %lld\n",
		       blockIdx.x, wave_id(), kc_work / nb, kc_wait / nb, nb, kcs.acc[0] / nb, kcs.acc[1] / nb, kcs.acc[2] / nb, kcs.acc[3] / nb);
	if (lane_id() == 0 && wave_id() == 0 && (blockIdx.x == 100 || blockIdx.x == 3000))
		printf("kc stamps block %u list: %lld rounds x 100 per column, %lld cycles per column, %lld entries x 100 per column, X %u D0 %u; (last round, from the list's start) reads %lld scans %lld stores %lld\n", blockIdx.x, kcs.acc[4] * 100 / nb, kcs.acc[5] / nb, kcs.acc[6] * 100 / nb, X, D0, kcs.acc[0] / nb, kcs.acc[1] / nb, kcs.acc[2] / nb);
#endif
	FSEQ_CLOCK_STAMP(blockIdx.x, 1);
	// done_host: tell the host that this block's lists and stride states are in memory (fseq_core.hpp)
	if constexpr (!RED) publish_block_done(done_host, blk, epoch);
}

template <int T, int E, int SIGMA, bool PK, bool EW = false, bool DENSE = false>
__global__ __launch_bounds__(T, (PK && T == 512) ? 6 : 4) void k_columns(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t N2,
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d,
	uint32_t L, uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr, uint32_t npass, uint32_t bsh,
	uint32_t snap_stride, uint32_t *__restrict__ ss_a, uint32_t *__restrict__ ss_d, uint32_t block0,
	uint32_t *done_host, uint32_t epoch, uint32_t const *__restrict__ colmask = nullptr, uint32_t const *__restrict__ blocklist = nullptr)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	columns_body<T, E, SIGMA, PK, EW, DENSE, false>(smem, msa, ld, m, n, B, N2, bstate_a, bstate_d, L, X, stride, ent, hdr, npass, bsh, snap_stride, ss_a, ss_d, block0,
	                                                 done_host, epoch, colmask, RedArgs{}, blocklist);
}

// [r5] the same on a block's representative rows (msa / ld: the reduced alignment; red: fseq_types.hpp)
template <int T, int E, int SIGMA, bool PK, bool EW = false>
__global__ __launch_bounds__(T, (PK && T == 512) ? 6 : 4) void k_columns_red(
	uint8_t const *__restrict__ msa, size_t ld, uint64_t n, uint32_t B,
	uint32_t L, uint32_t X, uint32_t stride, uint2 *__restrict__ ent, uint4 *__restrict__ hdr, uint32_t npass, uint32_t bsh, RedArgs const red)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	columns_body<T, E, SIGMA, PK, EW, false, true>(smem, msa, ld, 0u, n, B, 0u, nullptr, nullptr, L, X, stride, ent, hdr, npass, bsh, 0u, nullptr, nullptr, 0u,
	                                                nullptr, 0u, nullptr, red);
}

// List-capacity estimate: the length-L segment ending at block boundary k has #{i : d_k[i] > k - L} distinct
// rows; the median over the boundaries sizes the per-column lists before phase C runs (a wrong guess only
// costs a retry, never the result).
static __global__ __launch_bounds__(256) void k_boundary_recent(
	uint32_t const *__restrict__ bstate_d, uint32_t m, uint64_t n, uint32_t B, uint32_t L, uint32_t *__restrict__ out, uint32_t block0)
{
	// boundary block0 + blockIdx.x (a rank of a sharded run looks at its own boundaries)
	uint64_t k = (uint64_t) (blockIdx.x + block0) * B;
	if (k > n) k = n;
	if (k < L) { if (threadIdx.x == 0) out[blockIdx.x] = 0xFFFFFFFFu; return; }
	uint32_t const thr = (uint32_t) (k - L);
	uint32_t const *d = bstate_d + (size_t) (blockIdx.x + block0) * m;
	uint32_t cnt = 0;
	for (uint32_t i = threadIdx.x; i < m; i += 256) cnt += d[i] > thr ? 1u : 0u;
	__shared__ uint32_t red[4];
	cnt = readlane_u32(wave_incl_add(cnt), 63);
	if (lane_id() == 0) red[wave_id()] = cnt;
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// out[0] = 1 iff some word of v[0 .. count) is not zero
static __global__ __launch_bounds__(256) void k_any_nonzero(uint32_t const *__restrict__ v, uint32_t count, uint32_t *__restrict__ out)
{
	uint32_t any = 0;
	for (uint32_t i = threadIdx.x; i < count; i += 256u) any |= v[i];
	__shared__ uint32_t red[4];
	bool const some = __ballot(any != 0u) != 0ull;
	if (lane_id() == 0) red[wave_id()] = some ? 1u : 0u;
	__syncthreads();
	if (threadIdx.x == 0) out[0] = red[0] | red[1] | red[2] | red[3];
}

// copy the lists of selected columns into a compact buffer (for the host-side merge)
static __global__ __launch_bounds__(64) void k_gather_lists(
	uint64_t const *__restrict__ cols, uint32_t stride, uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr,
	uint2 *__restrict__ out_ent, uint4 *__restrict__ out_hdr)
{
	uint64_t const k = cols[blockIdx.x];
	uint4 const h = hdr[k];
	if (threadIdx.x == 0) out_hdr[blockIdx.x] = h;
	for (uint32_t i = threadIdx.x; i < h.x; i += 64)
		out_ent[(size_t) blockIdx.x * stride + i] = ent[k * (size_t) stride + i];
}

// ------------------------------------------------------------------------------------------------
// follow_traceback (segmentation_lp_context.cc:191-224) on the device: from DP[n - L] follow arg_idx = lb - L
// until lb == 0.  A chain of S dependent reads (6156 on BASELINE C3: 0.75 ms as host cache misses after a 4 MB
// copy, no better as one device thread), so it is cut into windows of TB_WIN entries:
//   k_tb_windows  every window: pointer doubling inside LDS -> for EVERY entry the first entry below the window
//                 its chain reaches and the number of entries visited on the way (the chain only descends)
//   k_tb_chain    one thread hops from window to window along the real chain (one dependent read per window
//                 instead of one per segment) and hands every visited window its entry point and output offset
//   k_tb_emit     every visited window walks its few hops in LDS and writes {entry, lb, key, size}
// out[] lists the last segment first; count[0] = S, count[1] = 1 iff the chain ended in an entry with lb == 0.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t TB_WIN = 8192;
constexpr uint32_t TB_TERM = 0xFFFFFFFFu;

// successor of entry t, or TB_TERM when the chain ends at t (lb == 0) or t does not hold a valid lb (entries no cell
// writes; they are never on the chain from n - L)
__device__ __forceinline__ uint32_t tb_next(uint32_t lb, uint32_t t, uint32_t L) { return (lb >= L && lb - L < t) ? lb - L : TB_TERM; }

// (a rank of a sharded run holds valid lb's for the entries [vlo, vhi) it computed: the chain leaves its part at the
// first entry below vlo, windows outside the part have nothing to do; vlo = 0, vhi = dp_size: the whole array)
static __global__ __launch_bounds__(256) void k_tb_windows(uint32_t const *__restrict__ LB, uint32_t dp_size, uint32_t L,
                                                    uint32_t *__restrict__ exit_next, uint32_t *__restrict__ exit_cnt,
                                                    uint32_t vlo = 0, uint32_t vhi = 0xFFFFFFFFu)
{
	__shared__ uint32_t nx[2][TB_WIN];
	__shared__ uint16_t cn[2][TB_WIN];
	uint32_t const wbase = blockIdx.x * TB_WIN;
	if (wbase >= vhi || wbase + TB_WIN <= vlo) return;
	uint32_t const base = wbase;
	for (uint32_t i = threadIdx.x; i < TB_WIN; i += 256u)
	{
		uint32_t const t = base + i;
		nx[0][i] = t < dp_size ? tb_next(LB[t], t, L) : TB_TERM;
		cn[0][i] = 1;
	}
	__syncthreads();
	// after round k an entry has jumped over 2^k entries of its chain inside the window (at most TB_WIN / max(L, 1) + 1 of them)
	uint32_t const hops = TB_WIN / (L ? L : 1u) + 2u;
	uint32_t cur = 0;
	for (uint32_t span = 1; span < hops; span <<= 1)
	{
		for (uint32_t i = threadIdx.x; i < TB_WIN; i += 256u)
		{
			uint32_t j = nx[cur][i];
			uint32_t c = cn[cur][i];
			if (j != TB_TERM && j >= base && j >= vlo) { c += cn[cur][j - base]; j = nx[cur][j - base]; }
			nx[cur ^ 1u][i] = j;
			cn[cur ^ 1u][i] = (uint16_t) c;
		}
		cur ^= 1u;
		__syncthreads();
	}
	for (uint32_t i = threadIdx.x; i < TB_WIN; i += 256u)
		if (base + i < dp_size) { exit_next[base + i] = nx[cur][i]; exit_cnt[base + i] = cn[cur][i]; }
}

// head[w] = {entry point, output offset} of the w-th visited window; count[0] = segments, count[1] = ok, count[2] = windows
static __global__ __launch_bounds__(64) void k_tb_chain(uint32_t const *__restrict__ exit_next, uint32_t const *__restrict__ exit_cnt, uint32_t dp_size,
                                                 uint2 *__restrict__ head, uint32_t max_windows, uint32_t *__restrict__ count)
{
	if (threadIdx.x != 0) return;
	uint32_t cur = dp_size - 1u, off = 0, nw = 0;
	while (nw < max_windows)
	{
		head[nw++] = make_uint2(cur, off);
		off += exit_cnt[cur];
		uint32_t const e = exit_next[cur];
		if (e == TB_TERM) break;
		cur = e;
	}
	count[0] = off; count[2] = nw;
}

// The chain of one rank of a sharded run: from entry `start` (inside the rank's part [vlo, ..)) window by window until the
// chain ends or leaves the part; the rank's output begins at offset off0 of the whole traceback.
// count[0] = entries of this rank, count[2] = its windows; word[0] = 1 + the entry the chain continues at (0: it ended
// here), word[1] = entries of this rank -- what the other ranks need to carry on (fseq_api.hip, follow_traceback_sharded)
static __global__ __launch_bounds__(64) void k_tb_chain_part(uint32_t const *__restrict__ exit_next, uint32_t const *__restrict__ exit_cnt, uint32_t start, uint32_t off0,
                                                      uint32_t vlo, uint2 *__restrict__ head, uint32_t max_windows, uint32_t *__restrict__ count, uint32_t *__restrict__ word)
{
	if (threadIdx.x != 0) return;
	uint32_t cur = start, off = off0, nw = 0, next = 0;
	while (nw < max_windows)
	{
		head[nw++] = make_uint2(cur, off);
		off += exit_cnt[cur];
		uint32_t const e = exit_next[cur];
		if (e == TB_TERM) break;
		if (e < vlo) { next = e + 1u; break; }
		cur = e;
	}
	count[0] = off - off0; count[2] = nw;
	word[0] = next; word[1] = off - off0;
}

static __global__ __launch_bounds__(256) void k_tb_emit(uint32_t const *__restrict__ LB, uint32_t const *__restrict__ M, uint32_t const *__restrict__ SZ,
                                                 uint32_t dp_size, uint32_t L, uint2 const *__restrict__ head, uint32_t *__restrict__ count,
                                                 uint4 *__restrict__ out, uint32_t cap, uint32_t vlo = 0)
{
	if (blockIdx.x >= count[2]) return;
	__shared__ uint32_t lbw[TB_WIN];
	__shared__ uint32_t s_n;
	uint2 const h = head[blockIdx.x];
	uint32_t const base = (h.x / TB_WIN) * TB_WIN;
	for (uint32_t i = threadIdx.x; i < TB_WIN; i += 256u) lbw[i] = base + i < dp_size ? LB[base + i] : 0u;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		uint32_t cur = h.x, n = 0;
		while (true)
		{
			uint32_t const lb = lbw[cur - base];
			if (h.y + n < cap) { out[h.y + n].x = cur; out[h.y + n].y = lb; }
			++n;
			uint32_t const nx = tb_next(lb, cur, L);
			if (nx == TB_TERM) { if (lb == 0u) count[1] = 1u; break; }       // (only the last window of the chain ends like this)
			if (nx < base || nx < vlo) break;
			cur = nx;
		}
		s_n = n;
	}
	__syncthreads();
	for (uint32_t j = threadIdx.x; j < s_n; j += 256u)
		if (h.y + j < cap)
		{
			uint32_t const t = out[h.y + j].x;
			out[h.y + j].z = M[t];
			out[h.y + j].w = SZ[t];
		}
}

// out[j] = {a[idx[j]], b[idx[j]]}: the DP entries on the traceback path (the host walks the lb chain, the keys and
// sizes of the visited entries follow in one small copy)
static __global__ __launch_bounds__(256) void k_gather_pairs(uint32_t const *__restrict__ idx, uint32_t count, uint32_t const *__restrict__ a,
                                                      uint32_t const *__restrict__ b, uint2 *__restrict__ out)
{
	uint32_t const j = blockIdx.x * 256u + threadIdx.x;
	if (j < count) out[j] = make_uint2(a[idx[j]], b[idx[j]]);
}

// find_segments_greedy (segmentation_lp_context.cc:335-390) asks, traceback boundary after boundary, whether
// unique_substring_count_lhs(current_lb) = #{d_rb > current_lb} <= max_segment_size (:363-364).  That holds exactly
// for current_lb >= tau, tau = the value of the first list entry at which the cumulative count exceeds
// max_segment_size -- one number per boundary, computed where the list lives (one wave per boundary; a rank of a
// sharded run answers for the columns [col_lo, col_hi) it owns and leaves zeros elsewhere).
//   out[j] = {tau, kind}: kind 0: merge iff current_lb >= tau (tau = 0: always);  1: the same, but the list ended
//   (incomplete) before the count was exceeded: current_lb < tau is undecided (lists too short);  2: never (the
//   values within L of the boundary alone exceed it).
enum { SEG_TAU_EXACT = 0, SEG_TAU_OPEN = 1, SEG_TAU_NEVER = 2 };
static __global__ __launch_bounds__(64) void k_seg_tau(
	uint64_t const *__restrict__ cols, uint64_t col_lo, uint64_t col_hi, uint32_t max_seg, uint32_t stride,
	uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint2 *__restrict__ out)
{
	uint64_t const k = cols[blockIdx.x];
	uint32_t const lane = lane_id();
	if (k < col_lo || k >= col_hi) { if (lane == 0) out[blockIdx.x] = make_uint2(0u, 0u); return; }
	uint4 const h = hdr[k];
	uint2 const *list = ent + k * (size_t) stride;
	uint32_t cum_base = 0;
	for (uint32_t s0 = 0; s0 < h.x; s0 += 64u)
	{
		uint32_t const i = s0 + lane;
		uint2 const e = i < h.x ? list[i] : make_uint2(0u, 0u);
		uint32_t const inc = cum_base + wave_incl_add(e.y);
		uint64_t const over = __ballot(i < h.x && inc > max_seg);
		if (over)
		{
			uint32_t const first = (uint32_t) __builtin_ctzll(over);
			uint32_t const v = shfl_u32(e.x, (int) first);
			if (lane == 0) out[blockIdx.x] = (s0 + first == 0u) ? make_uint2(0xFFFFFFFFu, SEG_TAU_NEVER) : make_uint2(v, SEG_TAU_EXACT);
			return;
		}
		cum_base = readlane_u32(inc, 63);
	}
	if (lane == 0)
		out[blockIdx.x] = h.z ? make_uint2(0u, SEG_TAU_EXACT) : make_uint2(list[h.x - 1u].x, SEG_TAU_OPEN);
}

// The same straight off the device traceback (k_tb_emit's output: entry i = {end - L, lb, max, size}, the last segment
// first; count[0] = S entries), so that the thresholds travel to the host together with the traceback: workgroup i
// answers for the boundary of entry i and writes out[S - 1 - i] (the host's order); max_seg = entry 0's maximum.
static __global__ __launch_bounds__(64) void k_seg_tau_tb(
	uint4 const *__restrict__ tb, uint32_t const *__restrict__ count, uint32_t L, uint32_t stride,
	uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint2 *__restrict__ out)
{
	uint32_t const S = count[0];
	if (blockIdx.x >= S) return;
	uint32_t const max_seg = tb[0].z;
	uint64_t const k = (uint64_t) tb[blockIdx.x].x + L - 1u;          // column rb - 1
	uint32_t const j = S - 1u - blockIdx.x;
	uint32_t const lane = lane_id();
	uint4 const h = hdr[k];
	uint2 const *list = ent + k * (size_t) stride;
	uint32_t cum_base = 0;
	for (uint32_t s0 = 0; s0 < h.x; s0 += 64u)
	{
		uint32_t const i = s0 + lane;
		uint2 const e = i < h.x ? list[i] : make_uint2(0u, 0u);
		uint32_t const inc = cum_base + wave_incl_add(e.y);
		uint64_t const over = __ballot(i < h.x && inc > max_seg);
		if (over)
		{
			uint32_t const first = (uint32_t) __builtin_ctzll(over);
			uint32_t const v = shfl_u32(e.x, (int) first);
			if (lane == 0) out[j] = (s0 + first == 0u) ? make_uint2(0xFFFFFFFFu, SEG_TAU_NEVER) : make_uint2(v, SEG_TAU_EXACT);
			return;
		}
		cum_base = readlane_u32(inc, 63);
	}
	if (lane == 0)
		out[j] = h.z ? make_uint2(0u, SEG_TAU_EXACT) : make_uint2(list[h.x - 1u].x, SEG_TAU_OPEN);
}

// ... and the size of a merged segment: #{d_col+1 > lb} for (col, lb) pairs (lp.cc:363,366), same ownership rule
static __global__ __launch_bounds__(64) void k_seg_count(
	uint64_t const *__restrict__ cols, uint64_t const *__restrict__ lbs, uint64_t col_lo, uint64_t col_hi, uint32_t stride,
	uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint32_t *__restrict__ out)
{
	uint64_t const k = cols[blockIdx.x], lb = lbs[blockIdx.x];
	uint32_t const lane = lane_id();
	if (k < col_lo || k >= col_hi) { if (lane == 0) out[blockIdx.x] = 0u; return; }
	uint4 const h = hdr[k];
	uint2 const *list = ent + k * (size_t) stride;
	uint32_t cnt = 0;
	for (uint32_t s0 = 0; s0 < h.x; s0 += 64u)
	{
		uint32_t const i = s0 + lane;
		uint2 const e = i < h.x ? list[i] : make_uint2(0u, 0u);
		bool const in = i < h.x && (uint64_t) e.x > lb;
		cnt += in ? e.y : 0u;
		if (__ballot(i < h.x && !in)) break;           // values descend: nothing behind this strip counts
	}
	cnt = readlane_u32(wave_incl_add(cnt), 63);
	if (lane == 0) out[blockIdx.x] = cnt;
}

} // namespace fseq
