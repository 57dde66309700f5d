// fseq_dpspec.hpp -- the DP of phase D spread over the chip: chunk-speculative sweeps.
//
// The DP (generate_traceback parts 2-4 + calculate_segmentation_lp_dp_arg, segmentation_lp_context.cc:26-188,
// 393-481) is a chain: cell t reads entries <= t - L.  k_dp<DP_WHOLE> walks it on one CU.  Here the regular
// rounds are cut into chunks, one workgroup (CU) each, and the chain is solved as a fixed point:
//   sweep 1   every chunk runs from "nothing known in front of me" (keys there count as 0): chunk 0 is exact,
//             the others get M_fresh = the best segmentation whose last segments lie inside the chunk.
//   lift      the recurrence is linear over (min, max): raising every key in front of a chunk to at least G
//             raises the chunk's results to max(G, .).  G_c = the largest tail-window minimum of the chunks in
//             front of c (the level M has settled at there; the window skips the L-wide spikes behind forced
//             cuts).  Dirty chunks are lifted to max(G_c, M) when G_c rose -- a guess, never trusted.
//   sweep k   every chunk behind the first one that changed runs again from the arrays as they stand
//             (masks and sparse table rebuilt from M in closed form in between).
//   stop      a sweep that changes no key: every cell was then computed from exactly the keys that are in the
//             arrays, i.e. the arrays are THE solution of the recurrence (cell t depends on entries < t only, so
//             the solution is unique) -- bit-identical to the serial walk, lb / size included.
// After sweep k the chunks 0..k-1 are exact whatever the guesses were, so the iteration ends after at most
// nchunks sweeps; measured: 3 (BASELINE C2, C3, C5), up to 7 with chunks as short as the tail window.  The host
// falls back to the serial kernel from the first dirty chunk if it is not done after a bounded number of sweeps.
#pragma once

#include "fseq_dp.hpp"

namespace fseq {

struct SpecCtl {
	uint32_t done;                // no key changed in the last sweep
	uint32_t first_changed;       // chunks <= first_changed are exact and frozen
	uint32_t sweeps;              // sweeps compared so far
	uint32_t overflow;            // OR of the chunks' "list too short" flags (valid when done)
};

struct SpecGeom {
	uint32_t const *chunk_r0;     // [nchunks + 1] first round of every chunk (device memory); [nchunks] = regular rounds
	uint32_t RL;                  // entries per round
	uint32_t nchunks;
	uint32_t NR;                  // regular entries: t = 0 .. n - 2L
	uint32_t t_final;             // n - L, the entry of the cell at rb = n
	uint32_t win;                 // tail window
};

__device__ __forceinline__ uint32_t spec_chunk_lo(SpecGeom const &G, uint32_t c) { return G.chunk_r0[c] * G.RL; }
__device__ __forceinline__ uint32_t spec_chunk_hi(SpecGeom const &G, uint32_t c)
{
	return (c + 1u == G.nchunks) ? G.NR : G.chunk_r0[c + 1u] * G.RL;
}
// chunk of a regular entry: the last chunk whose first round is <= t / RL
__device__ __forceinline__ uint32_t spec_chunk_of(SpecGeom const &G, uint32_t t)
{
	uint32_t const r = t / G.RL;
	uint32_t lo = 0, hi = G.nchunks;          // chunk_r0[lo] <= r < chunk_r0[hi]
	while (hi - lo > 1u)
	{
		uint32_t const mid = (lo + hi) >> 1;
		if (G.chunk_r0[mid] <= r) lo = mid; else hi = mid;
	}
	return lo;
}

// A: one workgroup per chunk: did the sweep change a key of the chunk, and the minimum over its tail window
// (sharded: a rank scans its own chunks chunk0 .. chunk0 + grid - 1 -- changed / tailmin of the others arrive by exchange --
// and reports in *below whether a query of one of its active chunks read an entry below valid_lo, the start of the
// window of foreign entries it holds; ovf2 = nullptr: not asked)
__global__ __launch_bounds__(256) void k_spec_scan(
	uint32_t const *__restrict__ M, uint32_t const *__restrict__ Mprev, SpecGeom const G,
	uint32_t const *__restrict__ active, uint32_t *__restrict__ changed, uint32_t *__restrict__ tailmin, SpecCtl const *ctl,
	uint32_t chunk0 = 0, uint32_t const *__restrict__ ovf2 = nullptr, uint32_t valid_lo = 0, uint32_t *__restrict__ below = nullptr)
{
	if (ctl->done) return;
	uint32_t const c = blockIdx.x + chunk0;
	if (ovf2 && threadIdx.x == 0 && active[c] != 0u && ovf2[2u * c + 1u] < valid_lo) atomicOr(below, 1u);
	uint32_t const lo = spec_chunk_lo(G, c), hi = spec_chunk_hi(G, c);
	uint32_t const w0 = (hi - lo > G.win) ? hi - G.win : lo;
	uint32_t ch = 0, mn = 0xFFFFFFFFu;
	bool const act = active[c] != 0u;
	for (uint32_t t = (act ? lo : w0) + threadIdx.x; t < hi; t += 256u)
	{
		uint32_t const v = M[t];
		if (act && v != Mprev[t]) ch = 1u;
		if (t >= w0) mn = min(mn, v);
	}
	if (act && c + 1u == G.nchunks && threadIdx.x == 0 && M[G.t_final] != Mprev[G.t_final]) ch = 1u;
	__shared__ uint32_t s_ch, s_mn;
	if (threadIdx.x == 0) { s_ch = 0; s_mn = 0xFFFFFFFFu; }
	__syncthreads();
	mn = wave_min_u32(mn);
	if (__ballot(ch != 0u) && lane_id() == 0) atomicOr(&s_ch, 1u);
	if (lane_id() == 0) atomicMin(&s_mn, mn);
	__syncthreads();
	if (threadIdx.x == 0) { changed[c] = s_ch; tailmin[c] = s_mn; }
}

// B: one wave: first changed chunk -> termination, the active set of the next sweep, the lifts
__global__ __launch_bounds__(64) void k_spec_decide(
	uint32_t nchunks, uint32_t first_sweep, uint32_t const *__restrict__ changed, uint32_t const *__restrict__ tailmin,
	uint32_t *__restrict__ floor_, uint32_t *__restrict__ lift, uint32_t *__restrict__ active, uint32_t const *__restrict__ ovf, SpecCtl *ctl)
{
	if (ctl->done) return;
	uint32_t const lane = lane_id();
	uint32_t first = 0xFFFFFFFFu, any_ovf = 0;
	for (uint32_t c0 = 0; c0 < nchunks; c0 += 64u)
	{
		uint32_t const c = c0 + lane;
		uint64_t const chm = __ballot(c < nchunks && changed[c] != 0u);
		if (chm && first == 0xFFFFFFFFu) first = c0 + (uint32_t) __builtin_ctzll(chm);
		any_ovf |= __ballot(c < nchunks && ovf[2u * c] != 0u) ? 1u : 0u;
	}
	if (first_sweep) first = 0;           // sweep 1: chunk 0 is exact, everything behind it is a guess
	if (first == 0xFFFFFFFFu)
	{
		if (lane == 0) { ctl->done = 1u; ctl->overflow = any_ovf; ctl->sweeps += 1u; }
		return;
	}
	uint32_t G = 0;                       // largest tail minimum of the chunks in front of c
	for (uint32_t c0 = 0; c0 < nchunks; c0 += 64u)
	{
		uint32_t const c = c0 + lane;
		uint32_t const tm = c < nchunks ? tailmin[c] : 0u;
		// inclusive prefix max over the lanes
		uint32_t inc = tm;
		inc = max(inc, dpp_mov<DPP_ROW_SHR1, 0xF>(0u, inc));
		inc = max(inc, dpp_mov<DPP_ROW_SHR2, 0xF>(0u, inc));
		inc = max(inc, dpp_mov<DPP_ROW_SHR4, 0xF>(0u, inc));
		inc = max(inc, dpp_mov<DPP_ROW_SHR8, 0xF>(0u, inc));
		inc = max(inc, dpp_mov<DPP_ROW_BCAST15, 0xA>(0u, inc));
		inc = max(inc, dpp_mov<DPP_ROW_BCAST31, 0xC>(0u, inc));
		uint32_t const exc = max(G, dpp_mov<DPP_WAVE_SHR1, 0xF>(0u, inc));
		if (c < nchunks)
		{
			bool const dirty = c > first;
			uint32_t const fl = floor_[c];
			bool const up = dirty && exc > fl;
			lift[c] = up ? exc : 0u;
			if (up) floor_[c] = exc;
			active[c] = dirty ? 1u : 0u;
		}
		G = max(G, readlane_u32(inc, 63));
	}
	if (lane == 0) { ctl->first_changed = first; ctl->sweeps += 1u; }
}

// C: one wave per 64-block of entries: apply the lift, keep the iterate (Mprev), rebuild the stack masks
// (rmq.hh:61-68 as K, fseq_dp.hpp) and the block minimum = level 0 of the sparse table
__global__ __launch_bounds__(256) void k_spec_rebuild(
	DpArrays const A, uint32_t *__restrict__ Mprev, SpecGeom const G, uint32_t const *__restrict__ lift, SpecCtl const *ctl)
{
	if (ctl->done) return;
	uint32_t const b = blockIdx.x * 4u + wave_id();
	uint32_t const lane = lane_id();
	uint32_t const t = b * 64u + lane;
	if (b * 64u >= G.NR)
	{
		// behind the regular entries: only the final cell's key is carried along
		if (t == G.t_final) Mprev[t] = A.M[t];
		return;
	}
	bool const valid = t < G.NR;
	uint32_t v = 0xFFFFFFFFu;
	if (valid)
	{
		v = max(A.M[t], lift[spec_chunk_of(G, t)]);
		A.M[t] = v;
		Mprev[t] = v;
	}
	else if (t == G.t_final) Mprev[t] = A.M[t];
	unsigned long long const k = stack_mask64(v);
	if (valid) A.K[t] = k;
	if (b * 64u + 63u < G.NR)
	{
		uint32_t const mn = wave_min_u32(v);
		uint32_t const smp = b * 64u + (uint32_t) __builtin_ctzll(__ballot(v == mn));   // first minimum of the block
		if (lane == 0) { A.Tb[b] = smp; A.Tbv[b] = mn; }
	}
}

// D: levels >= 1 of the sparse table in closed form.  rmq.hh:70-80 reads smp1 and smp2 from the same slot, so
// level p at j is the first strict minimum over the block minima of blocks j + 2^q - 1, q = 0..p.
__global__ __launch_bounds__(256) void k_spec_table(DpArrays const A, uint32_t ncomplete, SpecCtl const *ctl)
{
	if (ctl->done) return;
	uint32_t const j = blockIdx.x * 256u + threadIdx.x;
	if (j >= ncomplete) return;
	uint32_t idx = A.Tb[j], val = A.Tbv[j];
	for (uint32_t p = 1; p < DP_LEVELS && j + (1u << p) <= ncomplete; ++p)
	{
		uint32_t const bl = j + (1u << p) - 1u;
		uint32_t const nv = A.Tbv[bl];
		if (nv < val) { val = nv; idx = A.Tb[bl]; }             // rmq.hh:79: the new block only if strictly smaller
		A.Tb[(size_t) p * A.tstride + j] = idx;
		A.Tbv[(size_t) p * A.tstride + j] = val;
	}
}

// ---- test hook (fseq_debug_rmq): the restated rmq.hh on caller-supplied keys --------------------------------
// One workgroup.  The keys' masks and sparse table must have been built (k_spec_rebuild + k_spec_table, i.e. the
// closed forms the sweeps rely on).  Every query [beg, end) is answered twice: by rmq_query with everything read
// from HBM, and -- when the whole array fits the LDS rings (count <= DPW entries, <= DP_TRN blocks) -- by
// rmq_query_lds, the path practically every DP candidate takes.  out[q] = {index by the HBM path, index by the LDS
// path or 0xFFFFFFFF}.
__global__ __launch_bounds__(1024) void k_debug_rmq(DpArrays const A, uint32_t count, uint2 const *__restrict__ queries, uint32_t nq, uint2 *__restrict__ out)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	lds_char *const lds0 = (lds_char *) smem;
	DpLds D{};
	uint32_t off = 0;
	auto take = [&](size_t bytes) { uint32_t const o = off; off += (uint32_t) ((bytes + 15) & ~size_t(15)); return o; };
	D.Kr = (lds_u64 *) (lds0 + take((size_t) DPW * 8));
	D.Mr = (lds_u32 *) (lds0 + take((size_t) DPW * 4));
	D.Tr = (lds_u32 *) (lds0 + take((size_t) DP_LEVELS * DP_TRN * 4));
	D.Trv = (lds_u32 *) (lds0 + take((size_t) DP_LEVELS * DP_TRN * 4));
	bool const fits = count <= DPW && (count >> 6) <= DP_TRN;
	if (fits)
	{
		for (uint32_t t = threadIdx.x; t < count; t += 1024u) { D.Mr[t] = A.M[t]; D.Kr[t] = A.K[t]; }
		uint32_t const nb = count >> 6;
		for (uint32_t idx = threadIdx.x; idx < DP_LEVELS * DP_TRN; idx += 1024u)
		{
			uint32_t const p = idx / DP_TRN, j = idx % DP_TRN;
			if (p < 31u && j + (1u << p) <= nb) { D.Tr[idx] = A.Tb[(size_t) p * A.tstride + j]; D.Trv[idx] = A.Tbv[(size_t) p * A.tstride + j]; }
		}
	}
	__syncthreads();
	DpView V;
	V.safe_lo = 0xFFFFFFFFu;              // nothing is taken from the rings ...
	V.cb = 0xFFFFFF00u;                   // ... nor any sample
	V.fresh_lo = 0;
	for (uint32_t q = threadIdx.x; q < nq; q += 1024u)
	{
		uint32_t v;
		uint2 const qq = queries[q];
		uint32_t const a = rmq_query(A, D, V, qq.x, qq.y, &v);
		uint32_t const b = fits ? rmq_query_lds(D, qq.x, qq.y, &v) : 0xFFFFFFFFu;
		out[q] = make_uint2(a, b);
	}
}

} // namespace fseq
