// fseq_chainsort.hpp -- phase B for orders that do not fit LDS (m > 11,264 rows): one chain step as a stable radix
// sort by the block rank plus range maxima, instead of ceil(log2(nkeys) / 2) two-bit partition passes.
//
// A chain step (k_chain / k_chain_stream, fseq_kernels.hpp) takes the order (a, d) in front of a key block
// {rank[row], keyd[key], nkeys} to the order behind it: a' = the stable sort of a by rank[a]; the row at new position p
// whose predecessor there has ANOTHER rank starts a block key and takes d' = keyd[rank]; a row whose predecessor has the
// SAME rank was, in the old order, the next row of that rank behind it, at positions q < r, and takes
// d' = max d(q, r] -- the column update of libbio::pbwt::pbwt_context (founder_sequences.hh:56-65, SURVEY.md Appendix A
// step 2) with nkeys buckets.  The two-bit digit passes of the partition step carry that maximum through every pass,
// which is what makes them exact -- and, for 100,000 rows and 17-bit ranks, nine passes of the whole tile machinery by
// ONE workgroup: 2.5 ms a step, and phase B is a chain of ~40 such steps whatever the chip (and whatever the rank count of
// a sharded run: its Amdahl floor).  Here:
//   1. pairs (rank[a[i]], i), stably sorted by rank with an LSD radix sort of <= 9-bit digits (two passes for ranks below
//      2^18).  A wave owns a contiguous chunk of the array: per-wave digit histograms in LDS, one prefix over (digit,
//      wave), then every wave scatters its chunk in order -- the rows of a 64-row group that share a digit found with
//      one ballot per digit bit (their rank within the group = a popcount below the lane) -- no barrier inside a sweep;
//   2. prefix and suffix maxima of d inside 64-blocks (one wave scan each) and a sparse table over the block maxima;
//   3. every new position: its row, and keyd or the range maximum between the old positions of the two rows (two
//      block-end look-ups and two table entries, or a scan of at most 63 values inside one block).
// Everything but the histograms lives in the workgroup's workspace (L2-resident: a few MB).
#pragma once

#include <type_traits>

#include "fseq_stream.hpp"

namespace fseq {

constexpr uint32_t CS_MAX_DIGIT_BITS = 9;
constexpr uint32_t CS_BINS = 1u << CS_MAX_DIGIT_BITS;
constexpr uint32_t CS_LEVELS = 16;               // sparse table over at most 2^16 blocks of 64 rows

// workspace words of one workgroup: a0 d0 | a1 d1 (= pairs A during a sort) | pairs B | prefix max | suffix max | table
__host__ __device__ inline size_t chainsort_ws_words(uint32_t m)
{
	size_t const nblk = ((size_t) m + 63) / 64;
	return 8 * (size_t) m + CS_LEVELS * nblk + 64;
}

struct ChainSortLds {
	uint32_t hist[ST / WAVE][CS_BINS];           // per wave: digit counts, then the wave's write offsets
	uint32_t total[CS_BINS];
	uint32_t scan[ST / WAVE + 1];
};

__host__ __device__ inline size_t chainsort_lds_bytes() { return carve_bytes(1, sizeof(StreamLds)) + carve_bytes(1, sizeof(ChainSortLds)); }

// lanes of the wave whose digit equals mine (in = this lane holds a row), nbits digit bits
__device__ __forceinline__ uint64_t cs_match(uint32_t dg, bool in, uint32_t nbits)
{
	uint64_t mask = __ballot(in);
#pragma unroll
	for (uint32_t b = 0; b < CS_MAX_DIGIT_BITS; ++b)
		if (b < nbits)
		{
			uint64_t const bal = __ballot(in && ((dg >> b) & 1u));
			mask &= ((dg >> b) & 1u) ? bal : ~bal;
		}
	return mask;
}

__device__ __forceinline__ uint32_t cs_below(uint64_t mask)
{
	return (uint32_t) __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// One step: (a0, d0) -> (a1, d1) through the key block (rk, kd, nkeys).  w: the workgroup's workspace with a0 d0 at
// w + off0 .. and a1 d1 at w + off1 .. (each 2m words, off in {0, 2m}); ends with a barrier.
// cls (pass 2 behind the reduced phase C, k_chain_snap_stream): the key of a row is cls[rk[row]] -- the class of its block
// key at the boundary's column -- instead of rk[row]
template <bool P4 = false>
__device__ __forceinline__ void chain_step_sorted(
	uint32_t m, uint32_t const *__restrict__ rk, uint32_t const *__restrict__ kd, uint32_t nkeys,
	uint32_t *w, uint32_t cur, ChainSortLds &S, StreamLds &L, uint32_t const *__restrict__ cls = nullptr,
	uint32_t const *__restrict__ src_a = nullptr, uint32_t const *__restrict__ src_d = nullptr, uint32_t *__restrict__ dst_a = nullptr, uint32_t *__restrict__ dst_d = nullptr)
{
	// src_* / dst_* (pass 2): the order in front of the step is read where it lies, the order behind it written where it is wanted
	// (the workspace then only holds the pairs and the range maxima)
	uint32_t const tid = threadIdx.x, lane = lane_id();
	uint32_t const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	constexpr uint32_t NW = ST / WAVE;
	uint32_t const *a0 = src_a ? src_a : w + (size_t) cur * 2u * m, *d0 = src_d ? src_d : w + (size_t) cur * 2u * m + m;
	uint32_t *a1 = w + (size_t) (cur ^ 1u) * 2u * m, *d1 = a1 + m;
	// P4 [r5]: a pair is ONE word, key << pb | position (pb = bits of m - 1; the caller has checked that the key's bits fit
	// beside them) -- the sweeps of the sort and the new order move half the bytes
	using PairT = std::conditional_t<P4, uint32_t, uint2>;
	uint32_t pb = 1;
	while (pb < 32u && ((m - 1u) >> pb) != 0u) ++pb;
	uint32_t const pmask = pb < 32u ? (1u << pb) - 1u : 0xFFFFFFFFu;
	auto mk = [&](uint32_t key, uint32_t pos) -> PairT { if constexpr (P4) return (key << pb) | pos; else return make_uint2(key, pos); };
	auto key_of = [&](PairT pr) -> uint32_t { if constexpr (P4) return pr >> pb; else return pr.x; };
	auto pos_of = [&](PairT pr) -> uint32_t { if constexpr (P4) return pr & pmask; else return pr.y; };
	PairT *const pairA = reinterpret_cast<PairT *>(a1);                 // (the output buffers are free until step 3)
	if (dst_a) { a1 = dst_a; d1 = dst_d; }
	PairT *const pairB = reinterpret_cast<PairT *>(w + 4u * (size_t) m);
	uint32_t *const pm = w + 6u * (size_t) m, *const sm = w + 7u * (size_t) m, *const tab = w + 8u * (size_t) m;
	uint32_t const nblk = (m + 63u) / 64u;

	// ---- 1. the sort.  bits of a rank, passes of at most 9 bits, the last pass lands in pairB
	uint32_t bits = 1;
	while (bits < 32u && ((nkeys - 1u) >> bits) != 0u) ++bits;
	uint32_t const npass = (bits + CS_MAX_DIGIT_BITS - 1u) / CS_MAX_DIGIT_BITS;
	uint32_t const db = (bits + npass - 1u) / npass, nbins = 1u << db;
	// a wave's chunk: whole groups of 64 positions
	uint32_t const per = ((m + NW - 1u) / NW + 63u) & ~63u;
	uint32_t const c_lo = min(m, wave * per), c_hi = min(m, c_lo + per);
	for (uint32_t p = 0; p < npass; ++p)
	{
		uint32_t const shift = p * db;
		bool const first = p == 0;
		PairT const *src = ((npass - p) & 1u) ? pairA : pairB;          // (unused in the first pass)
		PairT *dst = ((npass - p) & 1u) ? pairB : pairA;
		// first pass: the pairs are made on the way (rank of the row at position i, i) -- a chain of dependent gathers
		// (position -> row -> rank [-> class]), followed once: the counting sweep leaves the key in the suffix-maxima buffer
		// (free until step 2), the scatter sweep reads it there
		auto load_count = [&](uint32_t i) -> PairT {
			if (!first) return src[i];
			uint32_t const key = cls ? cls[rk[a0[i]]] : rk[a0[i]];
			sm[i] = key;
			return mk(key, i);
		};
		auto load = [&](uint32_t i) -> PairT { return first ? mk(sm[i], i) : src[i]; };
		for (uint32_t b = lane; b < nbins; b += 64u) S.hist[wave][b] = 0;
		// (a wave's histogram row is its own: no barrier between clearing and counting; LDS operations of a wave stay in order)
		// U groups of 64 positions per iteration: their (dependent: position -> row -> rank) loads in flight together -- one
		// workgroup has a CU to itself here, and a sweep of ~100 round trips to L2 per wave would be all latency
		constexpr uint32_t U = 4;
		for (uint32_t i0 = c_lo; i0 < c_hi; i0 += 64u * U)
		{
			PairT pr[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) { uint32_t const i = i0 + u * 64u + lane; pr[u] = i < c_hi ? load_count(i) : mk(0u, 0u); }
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
				if (i0 + u * 64u + lane < c_hi) atomicAdd(&S.hist[wave][(key_of(pr[u]) >> shift) & (nbins - 1u)], 1u);
		}
		__syncthreads();
		// offsets: bins ascending, inside a bin the waves ascending (= the array order: the sort is stable)
		uint32_t tot = 0;
		if (tid < nbins)
			for (uint32_t v = 0; v < NW; ++v) { uint32_t const c = S.hist[v][tid]; S.hist[v][tid] = tot; tot += c; }
		uint32_t all;
		uint32_t const start = block_excl_add<ST>(tid < nbins ? tot : 0u, S.scan, &all);
		if (tid < nbins) S.total[tid] = start;
		__syncthreads();
		for (uint32_t b = lane; b < nbins; b += 64u) S.hist[wave][b] += S.total[b];
		for (uint32_t i0 = c_lo; i0 < c_hi; i0 += 64u * U)
		{
			PairT pr[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) { uint32_t const i = i0 + u * 64u + lane; pr[u] = i < c_hi ? load(i) : mk(0u, 0u); }
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				bool const in = i0 + u * 64u + lane < c_hi;
				uint32_t const dg = (key_of(pr[u]) >> shift) & (nbins - 1u);
				uint64_t const same = cs_match(dg, in, db);
				uint32_t const below = cs_below(same);
				uint32_t const base = S.hist[wave][dg];
				if (in) dst[base + below] = pr[u];
				// (the read above and this write are LDS operations of one wave: they execute in order)
				if (in && below == 0u) S.hist[wave][dg] = base + (uint32_t) __popcll(same);
			}
		}
		__syncthreads();
	}
	PairT const *perm = pairB;

	// ---- 2. range maxima of the old d: prefix / suffix maxima inside 64-blocks, sparse table over the block maxima
	for (uint32_t blk = wave; blk < nblk; blk += NW)
	{
		uint32_t const i = blk * 64u + lane;
		uint32_t const v = i < m ? d0[i] : 0u;
		uint32_t const pre = wave_incl_max(v);
		uint32_t const rev = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((63u - lane) << 2), (int) v);
		uint32_t const sufr = wave_incl_max(rev);
		uint32_t const suf = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((63u - lane) << 2), (int) sufr);
		if (i < m) { pm[i] = pre; sm[i] = suf; }
		if (lane == 63u) tab[blk] = pre;
	}
	__syncthreads();
	for (uint32_t k = 1; k < CS_LEVELS && (1u << k) <= nblk; ++k)
	{
		uint32_t const *lo = tab + (size_t) (k - 1u) * nblk;
		uint32_t *hi = tab + (size_t) k * nblk;
		for (uint32_t j = tid; j + (1u << k) <= nblk; j += ST) hi[j] = max(lo[j], lo[j + (1u << (k - 1u))]);
		__syncthreads();
	}

	// ---- 3. the new order (U positions per thread and iteration, their loads side by side; the branches are selects on
	// clamped look-ups, but for the rare range that lies inside one 64-block)
	{
		constexpr uint32_t U = 4;
		for (uint32_t p0 = tid; p0 < m; p0 += ST * U)
		{
			uint2 me[U], pv[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				uint32_t const p = min(p0 + u * ST, m - 1u);
				PairT const pm_ = perm[p], pp_ = perm[p ? p - 1u : 0u];
				me[u] = make_uint2(key_of(pm_), pos_of(pm_));
				pv[u] = make_uint2(key_of(pp_), pos_of(pp_));
			}
			uint32_t row[U], kdv[U], sv[U], pmv[U], t0[U], t1[U], dlast[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				uint32_t const lo = min(pv[u].y + 1u, m - 1u), hi = me[u].y;
				uint32_t const bl = lo >> 6, bh = hi >> 6;
				uint32_t const cnt = bh > bl + 1u ? bh - bl - 1u : 1u;
				uint32_t const k = 31u - (uint32_t) __builtin_clz(cnt);
				uint32_t const *t = tab + (size_t) k * nblk;
				row[u] = a0[hi];
				kdv[u] = kd[me[u].x];
				sv[u] = sm[lo];
				pmv[u] = pm[hi];
				dlast[u] = d0[hi];
				t0[u] = t[min(bl + 1u, nblk - 1u)];
				t1[u] = t[bh >= (1u << k) ? bh - (1u << k) : 0u];
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				uint32_t const p = p0 + u * ST;
				if (p < m)
				{
					bool const first = p == 0u || pv[u].x != me[u].x;
					uint32_t const lo = pv[u].y + 1u, hi = me[u].y;      // (same rank:) max of d0[lo .. hi], lo <= hi
					uint32_t const bl = lo >> 6, bh = hi >> 6;
					uint32_t dv = max(sv[u], pmv[u]);
					if (bh > bl + 1u) dv = max(dv, max(t0[u], t1[u]));
					if (!first && bl == bh)
					{
						dv = dlast[u];
						for (uint32_t i = lo; i < hi; ++i) dv = max(dv, d0[i]);
					}
					a1[p] = row[u];
					d1[p] = first ? kdv[u] : dv;
				}
			}
		}
	}
	(void) L;
	__syncthreads();
}

// ------------------------------------------------------------------------------------------------
// phase B, streamed, sorted form (same contract as k_chain_stream).  ws: [gridDim.x][chainsort_ws_words(m)] words.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ST) void k_chain_stream_sort(
	uint32_t const *__restrict__ rank, uint32_t const *__restrict__ keyd, uint32_t const *__restrict__ nkeys,
	uint32_t m, uint32_t nb_total, uint32_t G, uint64_t cols_per_block, uint32_t *ws,
	uint32_t const *__restrict__ start_a, uint32_t const *__restrict__ start_d,
	uint32_t *__restrict__ out_state_a, uint32_t *__restrict__ out_state_d,
	uint32_t *__restrict__ out_rank, uint32_t *__restrict__ out_keyd, uint32_t *__restrict__ out_nkeys, uint32_t grp0)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	StreamLds &L = *cv.take<StreamLds>(1);
	ChainSortLds &S = *cv.take<ChainSortLds>(1);
	uint32_t const tid = threadIdx.x;
	uint32_t *const w = ws + (size_t) blockIdx.x * chainsort_ws_words(m);
	uint32_t const grp = blockIdx.x + grp0;                     // chain index (workspaces stay per workgroup)
	uint32_t const b0 = grp * G;
	uint32_t const b1 = min(nb_total, b0 + G);
	uint32_t const kstart = (uint32_t) ((uint64_t) b0 * cols_per_block);
	for (uint32_t i = tid; i < m; i += ST)
	{
		w[i] = start_a ? start_a[(size_t) grp * m + i] : i;
		w[(size_t) m + i] = start_d ? start_d[(size_t) grp * m + i] : kstart;
	}
	__syncthreads();
	uint32_t cur = 0;
	for (uint32_t b = b0; b < b1; ++b)
	{
		uint32_t const *a = w + (size_t) cur * 2u * m, *d = a + m;
		if (out_state_a)
			for (uint32_t i = tid; i < m; i += ST) { out_state_a[(size_t) b * m + i] = a[i]; out_state_d[(size_t) b * m + i] = d[i]; }
		if (b + 1 == b1 && !out_rank && b1 != nb_total) break;       // (an expansion's last step: k_chain, fseq_kernels.hpp)
		{
			uint32_t const nk = nkeys[b];
			uint32_t kb = 1, pbits = 1;
			while (kb < 32u && ((nk - 1u) >> kb) != 0u) ++kb;
			while (pbits < 32u && ((m - 1u) >> pbits) != 0u) ++pbits;
			if (kb + pbits <= 32u) chain_step_sorted<true>(m, rank + (size_t) b * m, keyd + (size_t) b * m, nk, w, cur, S, L);
			else chain_step_sorted<false>(m, rank + (size_t) b * m, keyd + (size_t) b * m, nk, w, cur, S, L);
		}
		cur ^= 1u;
	}
	uint32_t const *a = w + (size_t) cur * 2u * m, *d = a + m;
	if (out_state_a && b1 == nb_total)
		for (uint32_t i = tid; i < m; i += ST) { out_state_a[(size_t) nb_total * m + i] = a[i]; out_state_d[(size_t) nb_total * m + i] = d[i]; }
	if (out_rank)
		stream_emit_ranks(m, a, d, kstart, out_rank + (size_t) grp * m, out_keyd + (size_t) grp * m, out_nkeys + grp, L);
}

// ------------------------------------------------------------------------------------------------
// [r5] Pass 2 behind the reduced phase C, streamed rows (fseq_reduced.hpp): a boundary inside a block is ONE such step from
// the block's boundary state, keyed by the classes the block's representatives form at the boundary's column (the tables of
// k_columns_red).  Workgroups take the tasks in turn, each in its own workspace.  ncls[t] == 0: the boundary is the block's
// border (a copy); 0xFFFFFFFF: not this kernel's.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ST) void k_chain_snap_stream(
	uint32_t const *__restrict__ bstate_a, uint32_t const *__restrict__ bstate_d, uint32_t const *__restrict__ rank, uint32_t m,
	uint32_t const *__restrict__ task_blk, uint32_t const *__restrict__ cls, uint32_t const *__restrict__ headd, uint32_t const *__restrict__ ncls,
	uint32_t cap, uint32_t ntasks, uint32_t *__restrict__ snap_a, uint32_t *__restrict__ snap_d, uint32_t *ws)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	StreamLds &L = *cv.take<StreamLds>(1);
	ChainSortLds &S = *cv.take<ChainSortLds>(1);
	uint32_t const tid = threadIdx.x;
	uint32_t *const w = ws + (size_t) blockIdx.x * chainsort_ws_words(m);
	for (uint32_t task = blockIdx.x; task < ntasks; task += gridDim.x)
	{
		uint32_t const D = ncls[task];
		if (D == 0xFFFFFFFFu) continue;
		size_t const sb = (size_t) task_blk[task] * m, ob = (size_t) task * m;
		if (D == 0u)
		{
			for (uint32_t i = tid; i < m; i += ST) { snap_a[ob + i] = bstate_a[sb + i]; snap_d[ob + i] = bstate_d[sb + i]; }
			continue;
		}
		// (the classes of a block's representatives and the positions share a word -- at most 12,288 classes, 2^18 rows)
		uint32_t kb = 1, pbits = 1;
		while (kb < 32u && ((D - 1u) >> kb) != 0u) ++kb;
		while (pbits < 32u && ((m - 1u) >> pbits) != 0u) ++pbits;
		if (kb + pbits <= 32u)
			chain_step_sorted<true>(m, rank + sb, headd + (size_t) task * cap, D, w, 0u, S, L, cls + (size_t) task * cap, bstate_a + sb, bstate_d + sb, snap_a + ob, snap_d + ob);
		else
			chain_step_sorted<false>(m, rank + sb, headd + (size_t) task * cap, D, w, 0u, S, L, cls + (size_t) task * cap, bstate_a + sb, bstate_d + sb, snap_a + ob, snap_d + ob);
	}
}

// ================================================================================================
// The same step spread over the chip.  One workgroup per chain leaves 255 CUs idle on the levels of phase B's recursion
// that matter for its length -- the top ones, a handful of chains of four steps each -- and a step is bound by what ONE
// CU's texture path takes to gather and scatter 100,000 rows a dozen times (1.6 ms a step: BASELINE C4 phase B 95 -> 61 ms
// with the single-workgroup form above).  Here every sweep of a step is a launch over (parts of 1,024 positions) x
// (chains of the level): per radix pass  count  ->  offsets  ->  scatter, then the new order; a chain of G blocks is G
// such rounds, the chains of a level side by side.  A wave owns a part (16 groups of 64 positions in order); the digit
// histograms of the parts sit in a table of their own ([chain][part][bin]), whose prefix over (bin, part) one workgroup
// per chain takes -- it also builds the sparse table of the block maxima the count sweep of the first pass left.
// The number of passes a block needs (bits of nkeys) is device data: the host always queues the passes m needs and the
// kernels of a pass the block does not need return at once (the ping-pong of the pair buffers is by the block's own count).
// Pairs are (rank | row low 12 bits << 20, position | row high bits << 20): rows, positions and ranks are below 2^20.
// ================================================================================================
constexpr uint32_t CM_PART = 1024;               // positions per part (one wave: 16 groups of 64)
constexpr uint32_t CM_WG = 256;                  // threads per workgroup of the part kernels: four parts

struct ChainMultiArgs {
	uint32_t const *rank, *keyd, *nkeys;         // the key blocks of the level below
	uint32_t m, nb_total, G;
	uint64_t cols_per_block;
	uint32_t *ws;                                // [chains of the launch][chainsort_ws_words(m)]
	uint32_t *hist;                              // [chains of the launch][parts][CS_BINS]
	uint32_t const *start_a, *start_d;
	uint32_t *out_state_a, *out_state_d, *out_rank, *out_keyd, *out_nkeys;
	uint32_t grp0;
	uint32_t step;                               // block b0 + step of every chain
	uint32_t pass;                               // radix pass of the sweep kernels
	uint32_t nchains;                            // chains of the launch (the grid's y is rounded up to the XCDs)
	uint32_t xcd_map;                            // the workgroups of a chain on ONE XCD (cm_wg)
};

// [r5] Which (part group, chain) a workgroup of the part / row kernels takes.  The hardware hands consecutive workgroups to
// the eight XCDs in turn, so with (x, y) = (part group, chain) taken as they come the twenty-five workgroups of a chain land
// on all eight -- and each XCD's L2 fetches the chain's tables for an eighth of its gathers: rank[row] of the first count
// sweep, the pair buffers of the scatter sweeps, d0 / the block maxima of the output sweep are all addressed at random inside
// arrays of m words that belong to ONE chain (400 KB each at m = 100,000; an L2 is 4 MB).  Here workgroup L goes to XCD
// L mod 8 and takes chain 8 (slot / nx) + (L mod 8), slot = L / 8: the workgroups of a chain are consecutive slots of one XCD.
struct CmWg { uint32_t x, chain; bool ok; };
__device__ __forceinline__ CmWg cm_wg(ChainMultiArgs const &A)
{
	CmWg w;
	if (!A.xcd_map) { w.x = blockIdx.x; w.chain = blockIdx.y; w.ok = blockIdx.y < A.nchains; return w; }
	uint32_t const nx = gridDim.x, L = blockIdx.y * nx + blockIdx.x;
	uint32_t const xcd = L & 7u, slot = L >> 3;
	w.chain = (slot / nx) * 8u + xcd;
	w.x = slot % nx;
	w.ok = w.chain < A.nchains;
	return w;
}

__host__ __device__ inline uint32_t chainmulti_parts(uint32_t m) { return (m + CM_PART - 1u) / CM_PART; }
__host__ __device__ inline uint32_t chainmulti_passes(uint32_t m)
{
	uint32_t bits = 1;
	while (bits < 32u && ((m - 1u) >> bits) != 0u) ++bits;
	return (bits + CS_MAX_DIGIT_BITS - 1u) / CS_MAX_DIGIT_BITS;
}

struct ChainMultiGeom {
	uint32_t b, npass, db, nbins;
	uint32_t pb;                                 // bits of a position (m - 1)
	bool p4;                                     // [r5] a pair is one word, key << pb | position (the key's bits fit beside the position's)
	bool active;
	uint32_t *w;
	uint32_t const *a0, *d0;
	uint32_t *a1, *d1;
	uint2 *pairA, *pairB;
	uint32_t *pm, *sm, *tab;
	uint32_t *hist;
};

__device__ __forceinline__ ChainMultiGeom chainmulti_geom(ChainMultiArgs const &A, uint32_t chain)
{
	ChainMultiGeom g;
	uint32_t const grp = chain + A.grp0;
	uint32_t const b0 = grp * A.G, b1 = min(A.nb_total, b0 + A.G);
	g.b = b0 + A.step;
	// (an expansion -- states wanted, no composite keys -- does not step through a chain's last block unless that is the last
	// of all: the state behind it is the next chain's start state, k_chain in fseq_kernels.hpp)
	g.active = g.b < b1 && (A.out_rank != nullptr || g.b + 1u < b1 || g.b + 1u == A.nb_total);
	uint32_t const nk = g.active ? A.nkeys[g.b] : 1u;
	uint32_t bits = 1;
	while (bits < 32u && ((nk - 1u) >> bits) != 0u) ++bits;
	g.npass = (bits + CS_MAX_DIGIT_BITS - 1u) / CS_MAX_DIGIT_BITS;
	g.db = (bits + g.npass - 1u) / g.npass;
	g.nbins = 1u << g.db;
	uint32_t const cur = A.step & 1u, m = A.m;
	g.pb = 1;
	while (g.pb < 32u && ((m - 1u) >> g.pb) != 0u) ++g.pb;
	g.p4 = bits + g.pb <= 32u;
	g.w = A.ws + (size_t) chain * chainsort_ws_words(m);
	g.a0 = g.w + (size_t) cur * 2u * m; g.d0 = g.a0 + m;
	g.a1 = g.w + (size_t) (cur ^ 1u) * 2u * m; g.d1 = g.a1 + m;
	g.pairA = reinterpret_cast<uint2 *>(g.a1);
	g.pairB = reinterpret_cast<uint2 *>(g.w + 4u * (size_t) m);
	g.pm = g.w + 6u * (size_t) m; g.sm = g.w + 7u * (size_t) m; g.tab = g.w + 8u * (size_t) m;
	g.hist = A.hist + (size_t) chain * chainmulti_parts(m) * CS_BINS;
	return g;
}

__device__ __forceinline__ uint2 cm_pack(uint32_t key, uint32_t pos, uint32_t row) { return make_uint2(key | (row << 20), pos | ((row >> 12) << 20)); }
__device__ __forceinline__ uint32_t cm_key(uint2 p) { return p.x & 0xFFFFFu; }
__device__ __forceinline__ uint32_t cm_pos(uint2 p) { return p.y & 0xFFFFFu; }
__device__ __forceinline__ uint32_t cm_row(uint2 p) { return (p.x >> 20) | ((p.y >> 20) << 12); }

// [r5] P4: the pairs of a chain whose keys fit beside a position in ONE word (the key blocks of the alignment's own blocks:
// at most 12 x 1,024 keys; BASELINE C4: eleven of every twelve steps) are that word, key << pb | position, and the new order
// looks its row up at the old position -- half the bytes through the four sweeps of the sort for one gather more in the last
template <bool P4>
struct CmPair {
	using T = std::conditional_t<P4, uint32_t, uint2>;
	static __device__ __forceinline__ T pack(ChainMultiGeom const &g, uint32_t key, uint32_t pos, uint32_t row)
	{
		if constexpr (P4) { (void) row; return (key << g.pb) | pos; } else { (void) g; return cm_pack(key, pos, row); }
	}
	static __device__ __forceinline__ T none() { if constexpr (P4) return 0u; else return make_uint2(0u, 0u); }
	static __device__ __forceinline__ uint32_t key(ChainMultiGeom const &g, T p) { if constexpr (P4) return p >> g.pb; else { (void) g; return cm_key(p); } }
	static __device__ __forceinline__ uint32_t pos(ChainMultiGeom const &g, T p) { if constexpr (P4) return p & ((1u << g.pb) - 1u); else { (void) g; return cm_pos(p); } }
	static __device__ __forceinline__ uint32_t row(ChainMultiGeom const &g, T p) { if constexpr (P4) return g.a0[pos(g, p)]; else { (void) g; return cm_row(p); } }
	static __device__ __forceinline__ T *bufA(ChainMultiGeom const &g) { return reinterpret_cast<T *>(g.pairA); }
	static __device__ __forceinline__ T *bufB(ChainMultiGeom const &g) { return reinterpret_cast<T *>(g.pairB); }
	static __device__ __forceinline__ T *stage(ChainMultiGeom const &g) { return (g.npass & 1u) ? bufA(g) : bufB(g); }
	// pair of position i for the sweep kernels of pass p: made on the way in the first pass, else from the pass before
	static __device__ __forceinline__ T load(ChainMultiGeom const &g, uint32_t const *rk, uint32_t pass, uint32_t i)
	{
		if (pass == 0u) { uint32_t const r = g.a0[i]; return pack(g, rk[r], i, r); }
		T const *src = ((g.npass - pass) & 1u) ? bufA(g) : bufB(g);
		return src[i];
	}
};

// [r5] the first pass's pairs are made ONCE, by its count sweep, which leaves them in the pair buffer the pass does not
// write (dead until the second pass overwrites it): rank[row] is a gather at a random row of a table of m words per block --
// a 64-byte line from memory for 4 bytes, thousands of chains side by side -- and the scatter sweep used to make it again
// (BASELINE C4, 2,048 chains: 3.5 ms of every step's 9)
// (CmPair::stage)

// start state of every chain of the launch into its workspace (and out_state in front of its first block)
__global__ __launch_bounds__(CM_WG) void k_cm_init(ChainMultiArgs const A)
{
	CmWg const wg = cm_wg(A);
	if (!wg.ok) return;
	uint32_t const chain = wg.chain, grp = chain + A.grp0, m = A.m;
	uint32_t const b0 = grp * A.G;
	if (b0 >= A.nb_total) return;
	uint32_t const kstart = (uint32_t) ((uint64_t) b0 * A.cols_per_block);
	uint32_t *w = A.ws + (size_t) chain * chainsort_ws_words(m);
	uint32_t const i = wg.x * CM_WG + threadIdx.x;
	if (i >= m) return;
	uint32_t const av = A.start_a ? A.start_a[(size_t) grp * m + i] : i;
	uint32_t const dv = A.start_d ? A.start_d[(size_t) grp * m + i] : kstart;
	w[i] = av; w[(size_t) m + i] = dv;
	if (A.out_state_a) { A.out_state_a[(size_t) b0 * m + i] = av; A.out_state_d[(size_t) b0 * m + i] = dv; }
}

// count sweep of pass A.pass: digit histogram of every part; first pass: also prefix / suffix maxima of d inside the
// 64-blocks of the part and the block maxima (level 0 of the table)
__global__ __launch_bounds__(CM_WG) void k_cm_count(ChainMultiArgs const A)
{
	__shared__ uint32_t hist[CM_WG / WAVE][CS_BINS];
	CmWg const wg = cm_wg(A);
	if (!wg.ok) return;
	ChainMultiGeom const g = chainmulti_geom(A, wg.chain);
	if (!g.active || A.pass >= g.npass) return;
	uint32_t const lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint32_t const part = wg.x * (CM_WG / WAVE) + wave, m = A.m;
	uint32_t const lo = part * CM_PART;
	if (lo >= m) return;
	uint32_t const hi = min(m, lo + CM_PART);
	uint32_t const *rk = A.rank + (size_t) g.b * m;
	uint32_t const shift = A.pass * g.db;
	for (uint32_t b = lane; b < g.nbins; b += 64u) hist[wave][b] = 0;
	auto sweep = [&](auto p4_) {
		using P = CmPair<decltype(p4_)::value>;
		constexpr uint32_t U = 4;
		for (uint32_t i0 = lo; i0 < hi; i0 += 64u * U)
		{
			typename P::T pr[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u) { uint32_t const i = i0 + u * 64u + lane; pr[u] = i < hi ? P::load(g, rk, A.pass, i) : P::none(); }
			if (A.pass == 0u)
			{
				typename P::T *const stage = P::stage(g);
#pragma unroll
				for (uint32_t u = 0; u < U; ++u) { uint32_t const i = i0 + u * 64u + lane; if (i < hi) stage[i] = pr[u]; }
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
				if (i0 + u * 64u + lane < hi) atomicAdd(&hist[wave][(P::key(g, pr[u]) >> shift) & (g.nbins - 1u)], 1u);
		}
	};
	if (g.p4) sweep(std::true_type{}); else sweep(std::false_type{});
	if (A.pass == 0u)
	{
		uint32_t const nblk = (m + 63u) / 64u;
		for (uint32_t i0 = lo; i0 < hi; i0 += 64u)
		{
			uint32_t const i = i0 + lane;
			uint32_t const v = i < m ? g.d0[i] : 0u;
			uint32_t const pre = wave_incl_max(v);
			uint32_t const rev = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((63u - lane) << 2), (int) v);
			uint32_t const sufr = wave_incl_max(rev);
			uint32_t const suf = (uint32_t) __builtin_amdgcn_ds_bpermute((int) ((63u - lane) << 2), (int) sufr);
			if (i < m) { g.pm[i] = pre; g.sm[i] = suf; }
			if (lane == 63u && (i0 >> 6) < nblk) g.tab[i0 >> 6] = pre;
		}
	}
	uint32_t *out = g.hist + (size_t) part * CS_BINS;
	for (uint32_t b = lane; b < g.nbins; b += 64u) out[b] = hist[wave][b];
}

// one workgroup per chain: the parts' histograms -> their write offsets (bins ascending, inside a bin the parts ascending);
// first pass: the levels of the sparse table over the block maxima
__global__ __launch_bounds__(ST) void k_cm_offsets(ChainMultiArgs const A)
{
	__shared__ uint32_t scan[ST / WAVE + 1];
	__shared__ uint32_t start_of[CS_BINS];
	ChainMultiGeom const g = chainmulti_geom(A, blockIdx.x);
	if (!g.active || A.pass >= g.npass) return;
	uint32_t const tid = threadIdx.x, m = A.m;
	uint32_t const nparts = chainmulti_parts(m);
	// (a thread per bin walks the parts in batches of independent loads: one load per round trip to L2 made this kernel --
	// a single workgroup per chain -- the longest of a step)
	constexpr uint32_t PB = 16;
	uint32_t tot = 0;
	if (tid < g.nbins)
		for (uint32_t p0 = 0; p0 < nparts; p0 += PB)
		{
			uint32_t c[PB];
#pragma unroll
			for (uint32_t q = 0; q < PB; ++q) c[q] = p0 + q < nparts ? g.hist[(size_t) (p0 + q) * CS_BINS + tid] : 0u;
#pragma unroll
			for (uint32_t q = 0; q < PB; ++q) tot += c[q];
		}
	uint32_t all;
	uint32_t const start = block_excl_add<ST>(tid < g.nbins ? tot : 0u, scan, &all);
	(void) start_of;
	if (tid < g.nbins)
	{
		uint32_t run = start;
		for (uint32_t p0 = 0; p0 < nparts; p0 += PB)
		{
			uint32_t c[PB];
#pragma unroll
			for (uint32_t q = 0; q < PB; ++q) c[q] = p0 + q < nparts ? g.hist[(size_t) (p0 + q) * CS_BINS + tid] : 0u;
#pragma unroll
			for (uint32_t q = 0; q < PB; ++q)
			{
				if (p0 + q < nparts) g.hist[(size_t) (p0 + q) * CS_BINS + tid] = run;
				run += c[q];
			}
		}
	}
	if (A.pass == 0u)
	{
		uint32_t const nblk = (m + 63u) / 64u;
		for (uint32_t k = 1; k < CS_LEVELS && (1u << k) <= nblk; ++k)
		{
			__syncthreads();
			uint32_t const *lo = g.tab + (size_t) (k - 1u) * nblk;
			uint32_t *hi = g.tab + (size_t) k * nblk;
			for (uint32_t j = tid; j + (1u << k) <= nblk; j += ST) hi[j] = max(lo[j], lo[j + (1u << (k - 1u))]);
		}
	}
}

// scatter sweep of pass A.pass: every part in order, the rows of a 64-row group that share a digit found by ballots
__global__ __launch_bounds__(CM_WG) void k_cm_scatter(ChainMultiArgs const A)
{
	__shared__ uint32_t offs[CM_WG / WAVE][CS_BINS];
	CmWg const wg = cm_wg(A);
	if (!wg.ok) return;
	ChainMultiGeom const g = chainmulti_geom(A, wg.chain);
	if (!g.active || A.pass >= g.npass) return;
	uint32_t const lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint32_t const part = wg.x * (CM_WG / WAVE) + wave, m = A.m;
	uint32_t const lo = part * CM_PART;
	if (lo >= m) return;
	uint32_t const hi = min(m, lo + CM_PART);
	uint32_t const *rk = A.rank + (size_t) g.b * m;
	uint32_t const shift = A.pass * g.db;
	uint32_t const *in_offs = g.hist + (size_t) part * CS_BINS;
	for (uint32_t b = lane; b < g.nbins; b += 64u) offs[wave][b] = in_offs[b];
	auto sweep = [&](auto p4_) {
		using P = CmPair<decltype(p4_)::value>;
		typename P::T *dst = ((g.npass - A.pass) & 1u) ? P::bufB(g) : P::bufA(g);
		constexpr uint32_t U = 4;
		for (uint32_t i0 = lo; i0 < hi; i0 += 64u * U)
		{
			typename P::T pr[U];
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				uint32_t const i = i0 + u * 64u + lane;
				pr[u] = i < hi ? (A.pass == 0u ? P::stage(g)[i] : P::load(g, rk, A.pass, i)) : P::none();
			}
#pragma unroll
			for (uint32_t u = 0; u < U; ++u)
			{
				bool const in = i0 + u * 64u + lane < hi;
				uint32_t const dg = (P::key(g, pr[u]) >> shift) & (g.nbins - 1u);
				uint64_t const same = cs_match(dg, in, g.db);
				uint32_t const below = cs_below(same);
				uint32_t const base = offs[wave][dg];
				if (in) dst[base + below] = pr[u];
				// (the read above and this write are LDS operations of one wave: they execute in order)
				if (in && below == 0u) offs[wave][dg] = base + (uint32_t) __popcll(same);
			}
		}
	};
	if (g.p4) sweep(std::true_type{}); else sweep(std::false_type{});
}

// the new order of every chain (and out_state behind the block, where the contract asks for it)
__global__ __launch_bounds__(CM_WG) void k_cm_output(ChainMultiArgs const A)
{
	CmWg const wg = cm_wg(A);
	if (!wg.ok) return;
	ChainMultiGeom const g = chainmulti_geom(A, wg.chain);
	if (!g.active) return;
	uint32_t const m = A.m, p = wg.x * CM_WG + threadIdx.x;
	if (p >= m) return;
	uint32_t const nblk = (m + 63u) / 64u;
	uint32_t const *kd = A.keyd + (size_t) g.b * m;
	uint32_t key_me, key_pv, pos_me, pos_pv, row;
	if (g.p4)
	{
		using P = CmPair<true>;
		P::T const me = P::bufB(g)[p], pv = P::bufB(g)[p ? p - 1u : 0u];
		key_me = P::key(g, me); key_pv = P::key(g, pv); pos_me = P::pos(g, me); pos_pv = P::pos(g, pv); row = P::row(g, me);
	}
	else
	{
		using P = CmPair<false>;
		P::T const me = P::bufB(g)[p], pv = P::bufB(g)[p ? p - 1u : 0u];
		key_me = P::key(g, me); key_pv = P::key(g, pv); pos_me = P::pos(g, me); pos_pv = P::pos(g, pv); row = P::row(g, me);
	}
	bool const first = p == 0u || key_pv != key_me;
	uint32_t dv;
	if (first) dv = kd[key_me];
	else
	{
		uint32_t const lo = pos_pv + 1u, hi = pos_me;                     // max of d0[lo .. hi], lo <= hi
		uint32_t const bl = lo >> 6, bh = hi >> 6;
		if (bl == bh)
		{
			dv = g.d0[hi];
			for (uint32_t i = lo; i < hi; ++i) dv = max(dv, g.d0[i]);
		}
		else
		{
			dv = max(g.sm[lo], g.pm[hi]);
			if (bh > bl + 1u)
			{
				uint32_t const k = 31u - (uint32_t) __builtin_clz(bh - bl - 1u);
				uint32_t const *t = g.tab + (size_t) k * nblk;
				dv = max(dv, max(t[bl + 1u], t[bh - (1u << k)]));
			}
		}
	}
	g.a1[p] = row; g.d1[p] = dv;
	uint32_t const grp = wg.chain + A.grp0;
	uint32_t const b1 = min(A.nb_total, grp * A.G + A.G);
	// out_state: the state in front of every block of the chain, and behind the last block of the whole sequence
	if (A.out_state_a && (g.b + 1u < b1 || g.b + 1u == A.nb_total))
	{
		A.out_state_a[(size_t) (g.b + 1u) * m + p] = row; A.out_state_d[(size_t) (g.b + 1u) * m + p] = dv;
	}
}

// the chains' composite key blocks (rank / keyd / nkeys of the order behind the last block of every chain)
__global__ __launch_bounds__(ST) void k_cm_emit(ChainMultiArgs const A, uint32_t steps_done)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	Carver cv{smem};
	StreamLds &L = *cv.take<StreamLds>(1);
	uint32_t const chain = blockIdx.x, grp = chain + A.grp0, m = A.m;
	uint32_t const b0 = grp * A.G;
	if (b0 >= A.nb_total) return;
	uint32_t const b1 = min(A.nb_total, b0 + A.G);
	uint32_t const cur = (b1 - b0) & 1u;                          // the chain made b1 - b0 steps
	(void) steps_done;
	uint32_t const *w = A.ws + (size_t) chain * chainsort_ws_words(m);
	uint32_t const *a = w + (size_t) cur * 2u * m, *d = a + m;
	uint32_t const kstart = (uint32_t) ((uint64_t) b0 * A.cols_per_block);
	stream_emit_ranks(m, a, d, kstart, A.out_rank + (size_t) grp * m, A.out_keyd + (size_t) grp * m, A.out_nkeys + grp, L);
}

} // namespace fseq
