// fseq_dp.hpp -- phase D of the segmentation path: the minimum-segmentation DP over the columns
// (calculate_segmentation_lp_dp_arg, segmentation_lp_context.cc:393-481, driven as generate_traceback
// parts 2-4 do, :26-188) with the exact semantics of rmq.hh, as ONE workgroup of sixteen specialised
// waves on one CU.  Consumes the per-column lists k_columns / k_columns_stream emit.
#pragma once

#include "fseq_kernels.hpp"
#include "fseq_types.hpp"

namespace fseq {

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
// (struct DpArrays {M, LB, SZ, Tb, Tbv, K, tstride}: fseq_types.hpp)

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
	uint32_t fresh_lo;    // speculative first sweep of a chunk: entries below this index count as 0 (else 0)
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
	uint32_t ncand, uint32_t L, uint32_t end, CellState &st, uint32_t &reach)
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
	bool const open_lhs = ok && qb < V.fresh_lo;             // range reaches in front of a speculative chunk: a key 0 is in it
	bool const okq = ok && !open_lhs;
	reach = min(reach, okq ? qb : 0xFFFFFFFFu);               // the lowest entry any query of this lane has read (k_dp, sharded sweeps)
	if (__ballot(okq && qb < V.safe_lo) == 0)
	{
		// every candidate of the strip lies inside the LDS ring
		if (okq)
		{
			uint32_t mv;
			idx = rmq_query_lds(D, qb, qe, &mv);                 // lp.cc:465
			val = max(mv, cum);                                  // lp.cc:468-471
		}
	}
	else if (okq)
	{
		uint32_t mv;
		idx = rmq_query(A, D, V, qb, qe, &mv);
		val = max(mv, cum);
	}
	if (open_lhs) { idx = qb; val = cum; }
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
	uint4 const h, uint32_t m, uint32_t L, uint32_t end, uint32_t *flags, uint32_t &reach)
{
	uint32_t const lane = lane_id();
	CellState st;
	st.best_v = 0xFFFFFFFFu; st.best_lb = 0; st.best_sz = 0; st.cum_base = 0;
	uint32_t const nent = h.x;
	bool done = dp_strip(A, D, V, en0, vnext0, nent, 0, 63, L, end, st, reach);
	for (uint32_t s0 = 63; !done && s0 < nent; s0 += 64)     // continue from HBM (rare; only this wave waits)
	{
		uint2 const e2 = list[s0 + lane];
		uint32_t const v2 = list[s0 + lane + 1u].x;
		done = dp_strip(A, D, V, e2, v2, nent, s0, 64, L, end, st, reach);
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
	uint32_t m, uint32_t L, uint32_t *flags, uint32_t &reach)
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
	bool const open_lhs = ok && qb < V.fresh_lo;             // see dp_strip
	bool const okq = ok && !open_lhs;
	reach = min(reach, okq ? qb : 0xFFFFFFFFu);
	if (__ballot(okq && qb < V.safe_lo) == 0)
	{
		if (okq)
		{
			uint32_t mv;
			idx = rmq_query_lds(D, qb, qe, &mv);                 // lp.cc:465
			val = max(mv, cum);                                  // lp.cc:468-471
		}
	}
	else if (okq)
	{
		uint32_t mv;
		idx = rmq_query(A, D, V, qb, qe, &mv);
		val = max(mv, cum);
	}
	if (open_lhs) { idx = qb; val = cum; }
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

__host__ __device__ inline DpRound dp_round(DpSchedule const &S, uint32_t r)
{
	DpRound R;
	uint32_t const last_end = S.n - S.L;
	R.final_round = (r + 1u == S.nrounds);
	bool const regular = r < S.nreg;
	R.e0 = R.final_round ? S.n : (regular ? S.L + r * S.RL : last_end + 1u);
	uint32_t const rest = last_end - R.e0 + 1u;
	R.len = R.final_round ? 1u : (regular ? (S.RL < rest ? S.RL : rest) : 0u);
	R.t0 = R.e0 - S.L;
	R.t1 = R.t0 + R.len;
	return R;
}

// Two schedules.  Classic: rounds of <= min(L, DP_RL) cells, the rmq.update of a round between two
// barriers.  Pipelined (L >= 96): rounds of 48 cells -- a round then never reads what the previous
// round wrote (a cell reads entries <= end - 2L), so two dedicated waves do the update of round r-1
// while the compute waves are already in round r: one barrier a round.
__host__ __device__ inline DpSchedule dp_schedule(uint32_t L, uint32_t n)
{
	DpSchedule S;
	S.L = L; S.n = n;
#ifndef FSEQ_DP_PIPE_MIN_L
#define FSEQ_DP_PIPE_MIN_L 96u
#endif
	S.pipe = L >= FSEQ_DP_PIPE_MIN_L;                         // measured: pays only with 4 cells per compute wave
	uint32_t const half = L / 2u < 48u ? L / 2u : 48u;
	S.RL = S.pipe ? (half / 12u) * 12u : (L < DP_RL ? L : DP_RL);   // pipelined: whole cells per compute wave
	S.nreg = ((n - L) - L) / S.RL + 1u;
	S.nrounds = S.nreg + (S.pipe ? 2u : 1u);
	return S;
}

// Rounds of the schedule whose cells only need the lists of columns < col_hi (a cell `end` reads the list
// of column end - 1): the DP of a column prefix can run while later columns are still being produced.
__host__ __device__ inline uint32_t dp_rounds_within(DpSchedule const &S, uint64_t col_hi)
{
	if (col_hi >= S.n) return S.nrounds;
	// the regular rounds need ascending columns (the drain round needs none, the final cell needs column n - 1):
	// first regular round that needs a column >= col_hi
	uint32_t lo = 0, hi = S.nreg;
	while (lo < hi)
	{
		uint32_t const mid = (lo + hi) / 2u;
		DpRound const R = dp_round(S, mid);
		bool const needs = (uint64_t) R.e0 + R.len - 2u >= col_hi;
		if (needs) hi = mid; else lo = mid + 1u;
	}
	return lo < S.nreg ? lo : S.nrounds - 1u;                  // all regular rounds (and the drain): everything but the final cell
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

// Two slices at once (any two (block, chunk) pairs): their dependency chains interleave -- one slice alone is a chain
// of 16 x (readlane, compare, select, min) that a wave sharing its SIMD with three others issues at ~60 cycles a link.
__device__ __forceinline__ void dp_mask_slice2(DpLds const &D, DpRound const &R, uint32_t blk0, uint32_t chunk0, uint32_t blk1, uint32_t chunk1)
{
	uint32_t const lane = lane_id();
	uint32_t idx[2] = {blk0 * 64u + lane, blk1 * 64u + lane}, plo[2] = {16u * chunk0, 16u * chunk1};
	uint32_t mine[2], runmin[2], bits[2];
#pragma unroll
	for (int c = 0; c < 2; ++c)
	{
		mine[c] = D.Mr[idx[c] & (DPW - 1u)];
		uint32_t const phi = plo[c] + 16u;
		uint32_t pm = lane >= phi ? mine[c] : 0xFFFFFFFFu;
		pm = min(pm, dpp_mov<DPP_ROW_SHR1, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_SHR2, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_SHR4, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_SHR8, 0xF>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_BCAST15, 0xA>(0xFFFFFFFFu, pm));
		pm = min(pm, dpp_mov<DPP_ROW_BCAST31, 0xC>(0xFFFFFFFFu, pm));
		runmin[c] = lane >= phi ? pm : mine[c];
		bits[c] = (lane >= plo[c] && lane < phi) ? (1u << (lane - plo[c])) : 0u;
	}
#pragma unroll
	for (int pp = 15; pp >= 0; --pp)
	{
#pragma unroll
		for (int c = 0; c < 2; ++c)
		{
			uint32_t const p = plo[c] + (uint32_t) pp;
			uint32_t const x = readlane_u32(mine[c], (int) p);
			bool const in = lane > p;
			bits[c] |= (in && x <= runmin[c]) ? (1u << pp) : 0u;
			runmin[c] = in ? min(runmin[c], x) : runmin[c];
		}
	}
	if (idx[0] >= R.t0 && idx[0] < R.t1) ((lds_u16 *) D.Kr)[(idx[0] & (DPW - 1u)) * 4u + chunk0] = (uint16_t) bits[0];
	if (idx[1] >= R.t0 && idx[1] < R.t1) ((lds_u16 *) D.Kr)[(idx[1] & (DPW - 1u)) * 4u + chunk1] = (uint16_t) bits[1];
}

// Stack masks of a whole 64-block held one key per lane: lane t gets the mask of entry t (bit p <= t set iff
// key[p] <= min key(p..t]).  Four independent dependency chains (one per 16-bit slice) that interleave.
__device__ __forceinline__ unsigned long long stack_mask64(uint32_t mine)
{
	uint32_t const lane = lane_id();
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
	return (unsigned long long) (bits[0] | (bits[1] << 16)) | ((unsigned long long) (bits[2] | (bits[3] << 16)) << 32);
}

// All four slices of one block by one wave (pipelined schedule, where only two waves do the update).
__device__ __forceinline__ void dp_mask_block(DpLds const &D, DpRound const &R, uint32_t blk)
{
	uint32_t const idx = blk * 64u + lane_id();
	bool const fresh = idx >= R.t0 && idx < R.t1;
	unsigned long long const k = stack_mask64(D.Mr[idx & (DPW - 1u)]);
	if (fresh) D.Kr[idx & (DPW - 1u)] = k;
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

// Chunk-speculative sweeps (fseq_dpspec.hpp): the regular rounds are cut into chunks of rounds_per_chunk rounds,
// workgroup c of the launch runs chunk c from whatever the arrays hold in front of it.
struct DpSpecArgs {
	uint32_t const *chunk_r0;             // [nchunks + 1] first round of every chunk; chunk_r0[nchunks] = number of regular rounds
	uint32_t nchunks;
	uint32_t chunk0;                      // workgroup i of the launch runs chunk chunk0 + i (a rank of a sharded run owns a chunk range)
	uint32_t fresh;                       // first sweep: nothing is known in front of a chunk (keys there count as 0)
	uint32_t const *active;               // [nchunks] chunks to run in this sweep
	uint32_t *ovf;                        // [nchunks][2] {"list too short" flag, lowest entry read} of the sweep that last ran the chunk
	uint32_t const *ctl;                  // ctl[0] != 0: the iteration has converged, nothing to do
};

// MODE 0: the whole schedule in one launch (r_begin_arg / r_end_arg ignored: the common case keeps its registers).
// MODE 1: rounds [r_begin_arg, r_end_arg).  MODE 2: workgroup = chunk of the speculative iteration.
enum { DP_WHOLE = 0, DP_PARTIAL = 1, DP_SPEC = 2 };
template <int MODE>
__global__ __launch_bounds__(1024) void k_dp(
	DpArrays const A, uint2 const *__restrict__ ent, uint4 const *__restrict__ hdr, uint32_t stride,
	uint32_t m, uint32_t n, uint32_t L, uint32_t *flags, uint32_t r_begin_arg, uint32_t r_end_arg, DpSpecArgs const SP)
{
	constexpr bool PARTIAL = MODE != DP_WHOLE;
	// Rounds [r_begin, r_end) of the schedule.  A launch that does not start at round 0 restores the LDS
	// rings from the arrays the launch before it flushed; one that stops early completes the rmq.update
	// of its last round before it flushes (pipelined schedule: one more iteration without cells).
	if (MODE == DP_SPEC)
	{
		if (SP.ctl[0] != 0u || SP.active[blockIdx.x + SP.chunk0] == 0u) return;
		flags = SP.ovf + 2u * (blockIdx.x + SP.chunk0);           // {list too short, lowest entry a query of the chunk read}
	}
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

	DpSchedule const S = dp_schedule(L, n);
	uint32_t const NWC = S.pipe ? 12u : DP_NWC;               // compute waves
	uint32_t const nrounds = S.nrounds, RL = S.RL;
	uint32_t r_begin = PARTIAL ? r_begin_arg : 0u, r_end = PARTIAL ? r_end_arg : nrounds;
	if (MODE == DP_SPEC)
	{
		// chunks tile the regular rounds; the last one also takes the drain round and the final cell
		uint32_t const ch = blockIdx.x + SP.chunk0;
		r_begin = SP.chunk_r0[ch];
		r_end = (ch + 1u == SP.nchunks) ? nrounds : SP.chunk_r0[ch + 1u];
	}
	bool const fresh = MODE == DP_SPEC && SP.fresh != 0u && r_begin > 0u;
	uint32_t const fresh_lo = fresh ? dp_round(S, r_begin).t0 : 0u;
	bool const stops_early = PARTIAL && r_end < nrounds;
	uint32_t const r_stop = (S.pipe && stops_early) ? r_end + 1u : r_end;   // pipelined: + the drain iteration
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
		// cells the round does not have load the round's first list again (any address that is surely mapped: a rank
		// of a sharded run holds the lists of its own columns only)
		uint2 const *pad = ent + (size_t) (R.e0 - 1u) * stride + (lane & 31u) * 2u;
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
			dma16_m0(i < R.len ? (void const *) src : (void const *) pad, dst);
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

	if (MODE == DP_SPEC && threadIdx.x == 0) { flags[0] = 0u; flags[1] = 0xFFFFFFFFu; }
	// lowest DP entry a query of this lane reads: a rank of a sharded run holds the entries of the other ranks only for a
	// window in front of its own (fseq_api.hip, run_dp_spec) and must know when a sweep looked below it
	uint32_t reach = 0xFFFFFFFFu;
	if (PARTIAL && r_begin > 0u && !fresh)
	{
		// resume: entries < T0 are computed, indexed and in HBM
		uint32_t const tid = threadIdx.x;
		// (the final cell sits L - 1 entries behind the last regular one: nothing is computed in between)
		DpRound const R0 = dp_round(S, r_begin);
		uint32_t const T0 = R0.final_round ? n - 2u * L + 1u : R0.t0;
		for (uint32_t t = (T0 > DPW ? T0 - DPW : 0u) + tid; t < T0; t += 1024u)
		{
			D.Mr[t & (DPW - 1u)] = A.M[t];
			D.Kr[t & (DPW - 1u)] = A.K[t];
		}
		uint32_t const cb = T0 >> 6;
		for (uint32_t idx = tid; idx < DP_LEVELS * DP_TRN; idx += 1024u)
		{
			uint32_t const p = idx / DP_TRN, q = idx % DP_TRN;
			if (p < 31u && (1u << p) <= cb)
			{
				uint32_t const jmax = cb - (1u << p);             // newest sample of level p
				uint32_t const back = (jmax - q) & (DP_TRN - 1u);  // the one sample j = q (mod DP_TRN) among the last DP_TRN
				if (back <= jmax)
				{
					uint32_t const j = jmax - back;
					D.Tr[idx] = A.Tb[(size_t) p * A.tstride + j];
					D.Trv[idx] = A.Tbv[(size_t) p * A.tstride + j];
				}
			}
		}
	}
	if (wave == DP_LOADER)
	{
		load_round(r_begin);
		if (r_begin + 1u < r_end) load_round(r_begin + 1u);
	}
	if (threadIdx.x < 4) fbcnt[threadIdx.x] = 0;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	dp_barrier();

#ifdef FSEQ_DP_STAMPS
	unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
#define DP_STAMP(x) unsigned long long x = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#else
#define DP_STAMP(x)
#endif
	uint32_t pair_cool = 0;                                   // rounds left before this wave tries cell pairs again
	for (uint32_t r = r_begin; r < r_stop; ++r)
	{
		DP_STAMP(ts0);
		bool const drain = PARTIAL && (r == r_end);               // no cells: only the update of round r - 1
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
		V.fresh_lo = fresh_lo;

		if (wave < NWC && drain)
		{
		}
		else if (wave < NWC)
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
					st = dp_cell_sequential(A, D, V, ent + (size_t) (end - 1u) * stride, en, vnext, h, m, L, end, flags, reach);
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
					uint64_t const fb = dp_cell_pair(A, D, V, slot, i0, R, m, L, flags, reach);
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
		else
		{
			// ---- the four waves that own no cells.  Pipelined mode: they share rmq.update of round r-1 (rmq.hh:61-81; the
			// launch before this one has done it for the round in front of r_begin) slice by slice -- one wave building
			// all four 16-bit slices of a block (~420 instructions, on a SIMD it shares with three compute waves) was
			// the critical path of every round (stamps: 6,900 of 7,000 cycles busy against ~5,000 on the compute
			// waves); the loader and the writer have instructions to spare, and a wave per SIMD evens the four out.
			auto update_share = [&]() {
				if (!S.pipe || r <= r_begin) return;
				DpRound const P = dp_round(S, r - 1u);
				if (P.len == 0 || P.final_round) return;
				// Two slices a call (interleaved chains).  Wave 12: slices 0, 1 of the first block; wave 13: its slices 2, 3;
				// the writer: the samples of a completed block and the slices of a second block that hold fresh lanes (the others cannot have a
				// bit set -- the fresh lanes lie below them -- and are zeroed).  The loader has its ~27 LDS-DMA issues a
				// round (measured ~6,400 cycles with M0 to set for each): no share.
				uint32_t const blkA = P.t0 >> 6, blkB = (P.t1 - 1u) >> 6;
				if (wave == 12u)
					dp_mask_slice2(D, P, blkA, 0u, blkA, 1u);
				else if (wave == 13u)
					dp_mask_slice2(D, P, blkA, 2u, blkA, 3u);
				else if (wave == DP_WRITER && (blkA + 1u) * 64u <= P.t1)
					dp_push_samples(D, blkA, (r - 1u) % DP_MBSLOTS);
				if (wave == DP_WRITER && blkB != blkA)
				{
					uint32_t const nfresh = P.t1 - blkB * 64u;              // fresh lanes 0 .. nfresh - 1 of the second block
					if (nfresh > 16u) dp_mask_slice2(D, P, blkB, 0u, blkB, 1u); else dp_mask_slice(D, P, blkB, 0u);
					if (nfresh > 32u) dp_mask_slice(D, P, blkB, 2u);
					uint32_t const idx = blkB * 64u + lane;
					if (lane < nfresh)
						for (uint32_t ch = (nfresh + 15u) / 16u; ch < 4u; ++ch) ((lds_u16 *) D.Kr)[(idx & (DPW - 1u)) * 4u + ch] = 0;
				}
			};
			if (wave == DP_LOADER)
			{
				// ---- loader: lists of round r+2; (its share of the update while they fly;) then make sure round r+1 has landed
				if (r + 2u < r_end)
				{
					load_round(r + 2u);
					update_share();
					dp_wait_all_but(npairs + 3u);                  // = everything but the round just issued
				}
				else
				{
					update_share();
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				}
			}
			else if (wave == DP_WRITER)
			{
				// ---- writer: a finished *and indexed* round to HBM
				uint32_t const lag = S.pipe ? 2u : 1u;
				if (r >= r_begin + lag) flush_round(r - lag);
				update_share();
			}
			else
				update_share();                                    // (waves 12 and 13 exist for this in pipelined mode)
		}
		DP_STAMP(ts1);
		dp_barrier();
		DP_STAMP(ts2);
		if (R.final_round && !drain) break;                   // no rmq.update after the last cell
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
		unsigned long long *q = reinterpret_cast<unsigned long long *>(flags + 8) + 48 + wave * 3u;   // classic schedule: barrier 1, update, barrier 2
		q[0] = acc[1]; q[1] = acc[2]; q[2] = acc[3];
	}
#endif

	if (MODE == DP_SPEC && wave < NWC)
	{
		uint32_t const lo = wave_min_u32(reach);
		if (lane == 0 && lo != 0xFFFFFFFFu) atomicMin(flags + 1, lo);
	}
	// the writer flushes what is still only in LDS
	if (wave == DP_WRITER)
	{
		uint32_t const lag = S.pipe ? 2u : 1u;
		for (uint32_t r = (r_stop >= r_begin + lag ? r_stop - lag : r_begin); r < r_end; ++r) flush_round(r);
	}
}

} // namespace fseq
