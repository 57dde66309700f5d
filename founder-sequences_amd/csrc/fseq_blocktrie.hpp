// fseq_blocktrie.hpp -- phase A for the streamed regime when a block has few distinct keys: the rows are sorted into
// their key classes by a trie over 16-column words, and only the trie is ranked.
//
// What phase A needs per block (fseq_blockkeys.hpp): the dense co-lex rank of every row's block key, the divergence in
// front of every distinct key, the number of distinct keys.  The key-space tree gets them with ~2 B / 8 rank operations
// over ALL m rows (BASELINE C4: 100,000 rows, 7,800 distinct keys in a block of 1,628 columns -- 215 lane-instructions
// per row and 16 columns).  Here:
//   phase 1  the block's columns are read ONCE, sixteen at a time from the LAST group to the first (the later column is
//            the more significant one: libbio's pbwt sorts by reversed prefixes).  A thread holds the class ids of its
//            16 * SL rows in registers; per group it transposes 16 packed words (16 rows x 16 columns of 2 bits) into the
//            rows' 32-bit group words.  A class keeps its id while it does not split: `direct[id]` holds the word of the
//            first row of the class that got there in this group (one compare-and-swap); a row with that word keeps its id
//            -- one LDS read and one compare, no hashing, no collisions: nearly every row, a class splits only where a row
//            first differs.  A row with another word looks (class, word) up in a hash table that holds the FULL pair --
//            equal pairs meet in one slot, different pairs never do, so the classes are exact, no fingerprints to verify
//            -- and whoever creates the entry draws a new id from a counter.  After a group the live ids are the nodes
//            of that trie level, {id, parent id, word}; they go to a per-workgroup workspace (L2 / HBM).
//   phase 2  the trie is ranked level by level in LDS: the nodes of a level in the order of (rank of the parent, word).
//            Nearly every parent has one child: count the children per parent rank, prefix sum, single children take
//            their parent's place and divergence; the nodes with siblings are collected in a list and ranked among them by
//            comparing words; a sibling's divergence is the last column in which it differs from the next smaller one.
//   output   rank of a row = rank of its last class (registers -> 16 words per thread and slot), keyd, nkeys: exactly
//            what blockkeys_tree_stream writes.
// Whatever does not fit (more than BT_NK_MAX classes, more than BT_OVF new ones in one group: diverse data) is given up:
// todo[b] = 1, and the key-space tree does that block.  One workgroup of 1024 threads per CU, blocks round-robin; 2 bits
// per symbol (sigma <= 4) and m <= 16 * 1024 * SL rows.
#pragma once

#include "fseq_blockkeys.hpp"

namespace fseq {

constexpr uint32_t BT_SENT = 0xFFFFFFFFu;         // direct[]: nobody here yet (a row whose word is this value goes to the pair table)
constexpr uint32_t BT_FLAG = 0x8000u;             // a row's id until the end of the group: FLAG | slot of its pair (the creator of the entry writes the id)
constexpr int BT_PER = 12;                        // entries of the id- / rank-indexed arrays per thread

// T threads (256, 512, 1024): 12 T classes (ids) per block, 8 T slots in the pair table (the classes one group can add), the
// nodes with siblings in pieces of 2 T.  LDS (bytes): phase 1 = direct[] + the pair table; phase 2 = rank by id (two),
// divergence by rank (two), children per rank, the list -- 34 KB at T = 256 (four workgroups per CU), 136 KB at T = 1024.
template <int T>
struct BtGeom {
	static constexpr uint32_t NK = (uint32_t) BT_PER * T, OVF = 8u * T, LIST = 2u * T;
	static constexpr size_t OFF_DIRECT = 0, OFF_PAIRS = 4 * (size_t) NK;
	static constexpr size_t OFF_RP = 0, OFF_RC = 2 * (size_t) NK, OFF_DP = 4 * (size_t) NK, OFF_DC = 6 * (size_t) NK,
	                        OFF_CNT = 8 * (size_t) NK, OFF_LIST = 10 * (size_t) NK, OFF_VARS = OFF_LIST + 8 * (size_t) LIST;
	static constexpr size_t LDS_BYTES = OFF_VARS + 256;
	static_assert(OFF_VARS >= OFF_PAIRS + 8 * (size_t) OVF, "the small variables lie behind the tables");
	static_assert(NK <= 0x3FFFu && OVF <= 0x2000u, "ids are 14 bits, pair slots 13");
};

// words per WORKGROUP of the workspace: the rows' ids (two to a word), node counts per level, the nodes (two words each), the nodes
// with siblings of one level.  bits per symbol 2 / 4 / 8: a packed word holds N = 32 / bits rows, a group is N columns.
__host__ __device__ inline uint32_t blocktrie_levels(uint32_t B, uint32_t bits) { uint32_t const N = 32u / bits; return (B + N - 1u) / N; }
__host__ __device__ inline size_t blocktrie_ws_words(uint32_t m, uint32_t B, uint32_t bits, uint32_t T)
{
	uint32_t const N = 32u / bits;
	size_t const nslots = ((m + N - 1u) / N + T - 1u) / T;
	size_t const NK = (size_t) BT_PER * T;
	return (N / 2u) * nslots * T + (size_t) blocktrie_levels(B, bits) * (1u + 2u * NK) + 2u * NK + 64;
}

// N x N matrix of BITS-bit elements (N = 32 / BITS): x[c] element r  ->  x[r] element c  (the stages written out: as a loop over
// the stage the compiler indexed the registers at run time)
template <int BITS, int J>
__device__ __forceinline__ void bt_transpose_stage(uint32_t (&x)[32 / BITS])
{
	constexpr int N = 32 / BITS;
	// the elements whose position has bit J clear
	constexpr uint32_t mask = BITS * J == 16 ? 0x0000FFFFu : BITS * J == 8 ? 0x00FF00FFu : BITS * J == 4 ? 0x0F0F0F0Fu : 0x33333333u;
#pragma unroll
	for (int i = 0; i < N; ++i)
		if (!(i & J))
		{
			uint32_t const t = ((x[i] >> (BITS * J)) ^ x[i + J]) & mask;
			x[i + J] ^= t;
			x[i] ^= t << (BITS * J);
		}
}
template <int BITS>
__device__ __forceinline__ void bt_transpose(uint32_t (&x)[32 / BITS])
{
	constexpr int N = 32 / BITS;
	if constexpr (N >= 16) bt_transpose_stage<BITS, 8>(x);
	if constexpr (N >= 8) bt_transpose_stage<BITS, 4>(x);
	bt_transpose_stage<BITS, 2>(x);
	bt_transpose_stage<BITS, 1>(x);
}

// a slot out of (class, word): 24-bit multiplies (full rate; a 32-bit multiply is a quarter-rate instruction here)
template <uint32_t OVF>
__device__ __forceinline__ uint32_t bt_hash(uint32_t parent, uint32_t word)
{
	uint32_t const x = __umul24(word, 0x9E3779u) ^ __umul24(word >> 8, 0x85EBCBu) ^ __umul24(parent, 0xC2B2AFu);
	return (x >> 12) & (OVF - 1u);
}

// A row whose word is not what direct[] holds for its class: the class's first row of this group registers its word there and
// keeps the id; every other word of the class goes through the pair table -- entry = {word, class | id << 14 | ready}: whoever
// creates it draws the id (returned); who finds it gets BT_FLAG | slot and reads the id behind the group's barrier.
// vars[5]: the next id; *full: no room (ids or slots).
template <int T>
__device__ __forceinline__ uint32_t bt_other_word(uint32_t *direct, unsigned long long *pairs, uint32_t *vars, uint32_t p, uint32_t w, uint32_t e, uint32_t &full)
{
	if (e == BT_SENT && w != BT_SENT)
	{
		uint32_t const old = atomicCAS(direct + p, BT_SENT, w);
		if (old == BT_SENT || old == w) return p;
	}
	unsigned long long const fresh = ((unsigned long long) p << 32) | w;
	constexpr uint32_t OVF = BtGeom<T>::OVF;
	uint32_t h = bt_hash<OVF>(p, w);
#pragma unroll 1
	for (uint32_t probe = 0; probe < OVF; ++probe)
	{
		unsigned long long cur = pairs[h];
		if (cur == ~0ull)
		{
			cur = atomicCAS(pairs + h, ~0ull, fresh);
			if (cur == ~0ull)
			{
				uint32_t id = atomicAdd(vars + 5, 1u);
				if (id >= BtGeom<T>::NK) { full = 1u; id = 0u; }
				reinterpret_cast<uint32_t *>(pairs + h)[1] = p | (id << 14) | (1u << 28);
				return id;
			}
		}
		if ((uint32_t) cur == w && ((uint32_t) (cur >> 32) & 0x3FFFu) == p) return BT_FLAG | h;
		h = (h + 1u) & (OVF - 1u);
	}
	full = 1u;
	return 0u;
}

#ifdef FSEQ_BT_STAMPS
#define BT_STAMP(i) do { long long const t_ = clock64(); bt_acc[i] += t_ - bt_last; bt_last = t_; } while (0)
#else
#define BT_STAMP(i) do {} while (0)
#endif

// the N packed words of a slot's group (zero behind the block's last column) and the ids of its N rows
template <int BITS>
__device__ __forceinline__ void bt_fetch(uint8_t const *__restrict__ msa, size_t ld, uint64_t kc, uint64_t kend, uint32_t wi, uint32_t nwords,
                                         uint32_t const *__restrict__ idw, bool with_ids, uint32_t (&x)[32 / BITS], uint32_t (&cur)[16 / BITS])
{
	constexpr int N = 32 / BITS, IW = N / 2;
	if (wi < nwords)
	{
#pragma unroll
		for (int c = 0; c < N; ++c)
			x[c] = (kc + (uint64_t) c < kend) ? *reinterpret_cast<uint32_t const *>(msa + (kc + (uint64_t) c) * ld + 4ull * wi) : 0u;
		if (with_ids)
		{
			if constexpr (IW >= 4)
			{
#pragma unroll
				for (int q = 0; q < IW / 4; ++q)
				{
					uint4 const a = reinterpret_cast<uint4 const *>(idw + (size_t) IW * wi)[q];
					cur[4 * q] = a.x; cur[4 * q + 1] = a.y; cur[4 * q + 2] = a.z; cur[4 * q + 3] = a.w;
				}
			}
			else
			{
				uint2 const a = *reinterpret_cast<uint2 const *>(idw + (size_t) IW * wi);
				cur[0] = a.x; cur[1] = a.y;
			}
		}
	}
}
template <int IW>
__device__ __forceinline__ void bt_store_ids(uint32_t *__restrict__ idw, size_t wi, uint32_t const (&cur)[IW])
{
	if constexpr (IW >= 4)
	{
#pragma unroll
		for (int q = 0; q < IW / 4; ++q)
			reinterpret_cast<uint4 *>(idw + (size_t) IW * wi)[q] = make_uint4(cur[4 * q], cur[4 * q + 1], cur[4 * q + 2], cur[4 * q + 3]);
	}
	else *reinterpret_cast<uint2 *>(idw + (size_t) IW * wi) = make_uint2(cur[0], cur[1]);
}

template <int BITS, int BT_T>
__global__ __launch_bounds__(BT_T) void k_blocktrie(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblk,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys, uint64_t col0,
	uint32_t *__restrict__ ws, size_t ws_per_group, uint32_t *__restrict__ given_up /* [0]: blocks given up */, uint32_t *__restrict__ todo)
{
	using G = BtGeom<BT_T>;
	constexpr int N = 32 / BITS, IW = N / 2;                               // rows per packed word = columns per group; id words per packed word
	constexpr uint32_t BT_NK_MAX = G::NK, BT_OVF = G::OVF, BT_LIST = G::LIST;
	constexpr size_t BT_OFF_DIRECT = G::OFF_DIRECT, BT_OFF_PAIRS = G::OFF_PAIRS, BT_OFF_RP = G::OFF_RP, BT_OFF_RC = G::OFF_RC, BT_OFF_DP = G::OFF_DP,
	                 BT_OFF_DC = G::OFF_DC, BT_OFF_CNT = G::OFF_CNT, BT_OFF_LIST = G::OFF_LIST, BT_OFF_VARS = G::OFF_VARS;
	extern __shared__ __attribute__((aligned(16))) char smem[];
	uint32_t *const direct = reinterpret_cast<uint32_t *>(smem + BT_OFF_DIRECT);
	unsigned long long *const pairs = reinterpret_cast<unsigned long long *>(smem + BT_OFF_PAIRS);
	uint16_t *const RA = reinterpret_cast<uint16_t *>(smem + BT_OFF_RP);
	uint16_t *const RB = reinterpret_cast<uint16_t *>(smem + BT_OFF_RC);
	uint16_t *const DA = reinterpret_cast<uint16_t *>(smem + BT_OFF_DP);
	uint16_t *const DB = reinterpret_cast<uint16_t *>(smem + BT_OFF_DC);
	uint16_t *const cnt = reinterpret_cast<uint16_t *>(smem + BT_OFF_CNT);
	uint32_t *const cnt32 = reinterpret_cast<uint32_t *>(smem + BT_OFF_CNT);
	uint2 *const list = reinterpret_cast<uint2 *>(smem + BT_OFF_LIST);
	// vars: [0 .. 2]: node counters of the levels (in turn); [3]: no room; [4]: list entries; [5]: the next id; [8 ..]: scan scratch
	uint32_t *const vars = reinterpret_cast<uint32_t *>(smem + BT_OFF_VARS);
	uint32_t const tid = threadIdx.x, lane = lane_id();
	uint32_t const nwords = (m + (uint32_t) N - 1u) / (uint32_t) N;
	uint32_t const nslots = (nwords + BT_T - 1u) / BT_T;                  // <= 32 (the host checks m)
	uint32_t *const wsg = ws + (size_t) blockIdx.x * ws_per_group;
	uint32_t *const idw = wsg;                                             // ids of the rows, two to a word: IW words per packed word of N rows
	uint32_t *const wst = wsg + (size_t) IW * nslots * BT_T;               // the trie: node counts per level, nodes, the list

	for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x)
	{
		uint64_t const k0 = col0 + (uint64_t) b * B;
		uint64_t const kend = (k0 + B < n) ? k0 + B : n;
		uint32_t const levels = ((uint32_t) (kend - k0) + (uint32_t) N - 1u) / (uint32_t) N;
		uint32_t *const hdr = wst;
		uint2 *const nodes = reinterpret_cast<uint2 *>(wst + ((levels + 3u) & ~3u));
		uint2 *const glist = nodes + (size_t) levels * BT_NK_MAX;      // the nodes with siblings of the level at hand
		size_t const ob = (size_t) b * m;

#ifdef FSEQ_BT_STAMPS
		long long bt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt_last = clock64();
#endif
		// ---------------- phase 1: the classes, group by group from the last one
		if (tid == 0) { vars[0] = 0u; vars[1] = 0u; vars[2] = 0u; vars[3] = 0u; vars[5] = 1u; }      // (id 0: every row, before the first group)
		uint32_t level_base = 0;
		bool bad = false;
		for (uint32_t t = 0; t < levels; ++t)
		{
			uint32_t const g = levels - 1u - t;
			uint64_t const kc = k0 + (uint64_t) N * g;
			// (the first slot's words are on their way while the tables are cleared)
			uint32_t xn[N], curn[IW];
#pragma unroll
			for (int q = 0; q < IW; ++q) curn[q] = 0u;
			bt_fetch<BITS>(msa, ld, kc, kend, tid, nwords, idw, t != 0u, xn, curn);
			{
				uint4 *const t4 = reinterpret_cast<uint4 *>(smem);
				uint4 const ones = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
				for (int i = 0; i < (int) ((4 * BT_NK_MAX + 8 * BT_OVF) / 16 / BT_T); ++i) t4[tid + i * BT_T] = ones;
			}
			if (tid == 0) vars[(t + 1u) % 3u] = 0u;            // (the counter of the level before is still being read)
			__syncthreads();
			BT_STAMP(0);
			uint32_t full = 0;
			uint32_t flagged = 0;                              // the slots in which one of my rows holds FLAG | slot of a pair
#pragma unroll 1
			for (uint32_t s = 0; s < nslots; ++s)
			{
				uint32_t const wi = s * BT_T + tid;
				uint32_t x[N], cur[IW];
#pragma unroll
				for (int c = 0; c < N; ++c) x[c] = xn[c];
#pragma unroll
				for (int q = 0; q < IW; ++q) cur[q] = curn[q];
				// the next slot's words and ids are loaded while this one is looked up
				if (s + 1u < nslots) bt_fetch<BITS>(msa, ld, kc, kend, wi + BT_T, nwords, idw, t != 0u, xn, curn);
				if (wi < nwords)
				{
					bt_transpose<BITS>(x);
					BT_STAMP(6);
					if (wi * (uint32_t) N + (uint32_t) N > m)
					{
						// the last word of a column: the positions behind row m - 1 follow its first row (same class throughout)
						uint32_t const valid = m - wi * (uint32_t) N;
#pragma unroll
						for (int r = 1; r < N; ++r) x[r] = ((uint32_t) r < valid) ? x[r] : x[0];
					}
					bool changed = t == 0u;
					uint32_t fl = 0;
#pragma unroll
					for (int q4 = 0; q4 < N / 4; ++q4)
					{
						// four rows at a time; no branch, and no new ids, while every row carries the word registered for its class
						uint32_t par[4], e[4];
#pragma unroll
						for (int u = 0; u < 4; ++u)
						{
							int const r = q4 * 4 + u;
							par[u] = (r & 1) ? (cur[r >> 1] >> 16) : (cur[r >> 1] & 0xFFFFu);
							e[u] = direct[par[u]];
						}
						bool miss = false;
#pragma unroll
						for (int u = 0; u < 4; ++u) miss |= e[u] != x[q4 * 4 + u] || e[u] == BT_SENT;
						if (miss)
						{
#pragma unroll
							for (int u = 0; u < 4; ++u)
							{
								int const r = q4 * 4 + u;
								if (e[u] != x[r] || e[u] == BT_SENT)
								{
									uint32_t const id = bt_other_word<BT_T>(direct, pairs, vars, par[u], x[r], e[u], full);
									changed |= id != par[u];
									par[u] = id;
									fl |= id & BT_FLAG;
								}
							}
							cur[2 * q4] = par[0] | (par[1] << 16);
							cur[2 * q4 + 1] = par[2] | (par[3] << 16);
						}
					}
					if (changed) bt_store_ids<IW>(idw, wi, cur);
					flagged |= fl ? (1u << s) : 0u;
					BT_STAMP(1);
				}
			}
			if (full) vars[3] = 1u;
			BT_STAMP(1);
			__syncthreads();
			BT_STAMP(2);
			// the rows that found a pair somebody else created: its id is in the table by now
			while (flagged)
			{
				uint32_t const s = (uint32_t) __ffs((int) flagged) - 1u;
				flagged &= flagged - 1u;
				uint32_t const wi = s * BT_T + tid;
				uint32_t xd[N], cur[IW];
				bt_fetch<BITS>(msa, ld, kend, kend, wi, nwords, idw, true, xd, cur);      // (the ids only: no column in front of kend)
#pragma unroll
				for (int q = 0; q < IW; ++q)
				{
					uint32_t v = cur[q];
					if (v & BT_FLAG) v = (v & 0xFFFF0000u) | ((reinterpret_cast<uint32_t const *>(pairs + (v & (BT_OVF - 1u)))[1] >> 14) & 0x3FFFu);
					if (v & (BT_FLAG << 16)) v = (v & 0x0000FFFFu) | (((reinterpret_cast<uint32_t const *>(pairs + ((v >> 16) & (BT_OVF - 1u)))[1] >> 14) & 0x3FFFu) << 16);
					cur[q] = v;
				}
				bt_store_ids<IW>(idw, wi, cur);
			}
			BT_STAMP(7);
			// the live ids are the nodes of this level: who kept its id (direct[]), who got a new one (the pair table)
			{
				uint32_t mine = 0;
#pragma unroll
				for (int i = 0; i < BT_PER; ++i) mine += direct[tid + i * BT_T] != BT_SENT ? 1u : 0u;      // (NK = BT_PER * T)
#pragma unroll
				for (int i = 0; i < (int) (BT_OVF / BT_T); ++i) mine += pairs[tid + i * BT_T] != ~0ull ? 1u : 0u;
				uint32_t const inc = wave_incl_add(mine);
				uint32_t base = 0;
				if (lane == 63) base = atomicAdd(vars + t % 3u, inc);
				base = readlane_u32(base, 63) + inc - mine;
				if (level_base + base + mine <= (size_t) levels * BT_NK_MAX)
				{
#pragma unroll 4
					for (int i = 0; i < BT_PER; ++i)
					{
						uint32_t const id = tid + i * BT_T;
						uint32_t const w = direct[id];
						if (w != BT_SENT) { nodes[level_base + base] = make_uint2(id | (id << 16), w); ++base; }
					}
#pragma unroll 4
					for (int i = 0; i < (int) (BT_OVF / BT_T); ++i)
					{
						unsigned long long const e = pairs[tid + i * BT_T];
						if (e != ~0ull)
						{
							uint32_t const hi = (uint32_t) (e >> 32);
							nodes[level_base + base] = make_uint2(((hi >> 14) & 0x3FFFu) | ((hi & 0x3FFFu) << 16), (uint32_t) e);
							++base;
						}
					}
				}
			}
			__syncthreads();
			BT_STAMP(3);
			uint32_t const nt = vars[t % 3u];
			if (nt > BT_NK_MAX || vars[3] != 0u) { bad = true; break; }
			if (tid == 0) hdr[t] = nt;
			level_base += nt;
		}
		if (bad)
		{
			if (tid == 0) { todo[b] = 1u; atomicAdd(given_up, 1u); }
			__syncthreads();
			continue;
		}

		// ---------------- phase 2: the trie ranked level by level
		uint16_t *Rp = RA, *Rc = RB, *Dp = DA, *Dc = DB;
		__syncthreads();                                       // (the tables are history: their LDS is the arrays of phase 2 now)
		if (tid == 0) { Rp[0] = 0; Dp[0] = (uint16_t) (kend - k0); }
		uint32_t nprev = 1;
		level_base = 0;
		for (uint32_t t = 0; t < levels; ++t)
		{
			uint32_t const g = levels - 1u - t;
			uint32_t const nt = hdr[t];
			uint2 const *const nd = nodes + level_base;
			// children per parent rank
			for (uint32_t i = tid; i < (nprev + 1u) / 2u; i += BT_T) cnt32[i] = 0u;
			if (tid == 0) vars[4] = 0u;
			__syncthreads();
			// (the thread's nodes of the level in registers: one trip to the workspace, not one per node and pass)
			uint2 mynd[BT_PER];
			uint32_t myr[BT_PER];
#pragma unroll
			for (int k = 0; k < BT_PER; ++k) mynd[k] = (tid + k * BT_T < nt) ? nd[tid + k * BT_T] : make_uint2(0u, 0u);
#pragma unroll
			for (int k = 0; k < BT_PER; ++k)
				if (tid + k * BT_T < nt)
				{
					myr[k] = Rp[(mynd[k].x >> 16) & 0x3FFFu];
					atomicAdd(cnt32 + (myr[k] >> 1), 1u << (16u * (myr[k] & 1u)));
				}
			__syncthreads();
			// exclusive prefix sum over the parent ranks, in place: first place of the children | 0x8000 when there are several
			{
				uint32_t c[BT_PER], sum = 0;
				uint32_t const j0 = tid * BT_PER;
#pragma unroll
				for (int k = 0; k < BT_PER; ++k) { c[k] = (j0 + k < nprev) ? cnt[j0 + k] : 0u; sum += c[k]; }
				uint32_t total;
				uint32_t run = block_excl_add<BT_T>(sum, vars + 8, &total);
#pragma unroll
				for (int k = 0; k < BT_PER; ++k)
					if (j0 + k < nprev) { cnt[j0 + k] = (uint16_t) (run | (c[k] > 1u ? 0x8000u : 0u)); run += c[k]; }
			}
			__syncthreads();
#pragma unroll
			for (int k = 0; k < BT_PER; ++k)
				if (tid + k * BT_T < nt)
				{
					uint32_t const id = mynd[k].x & 0x3FFFu;
					uint32_t const r = myr[k];
					uint32_t const e = cnt[r];
					if (!(e & 0x8000u)) { Rc[id] = (uint16_t) e; Dc[e] = Dp[r]; }
					else
					{
						uint32_t const at = atomicAdd(vars + 4, 1u);
						uint2 const v = make_uint2(r | (id << 16), mynd[k].y);
						if (at < BT_LIST) list[at] = v;                   // (the first piece is in place already)
						glist[at] = v;
					}
				}
			__syncthreads();
			// the nodes with siblings: place among the children of their parent by comparing words, pair by pair -- the list
			// is a hundred entries but for a level in which many rows change (every row, where a recombination block of the
			// founders ends) -- through LDS in pieces of BT_LIST
			uint32_t const nl = vars[4];
#pragma unroll 1
			for (uint32_t k0_ = 0; k0_ < nl; k0_ += BT_T)
			{
				uint32_t const i = k0_ + tid;
				// (a list longer than one piece is staged over list[]: read from the workspace then)
				uint2 const me = (i < nl) ? (nl <= BT_LIST ? list[i] : glist[i]) : make_uint2(0xFFFFu, 0u);      // (else: no such parent rank)
				uint32_t const r = me.x & 0xFFFFu, w = me.y;
				uint32_t pos = 0, pred = 0;
#pragma unroll 1
				for (uint32_t c0 = 0; c0 < nl; c0 += BT_LIST)
				{
					uint32_t const cn = min(BT_LIST, nl - c0);
					if (nl > BT_LIST)
					{
						// (the list in pieces; the first one is in place for k0_ = 0 only)
						__syncthreads();
						for (uint32_t j = tid; j < cn; j += BT_T) list[j] = glist[c0 + j];
						__syncthreads();
					}
					// [r5] (the waves without an entry of the list do not walk it: a list of a few hundred entries is five waves' worth,
					// and the walk of all sixteen was a fifth of the phase)
					if (k0_ + (tid & ~63u) < nl)
					{
#pragma unroll 4
						for (uint32_t j = 0; j < cn; ++j)
						{
							uint2 const o = list[j];
							bool const below = (o.x & 0xFFFFu) == r && o.y < w;
							pos += below ? 1u : 0u;
							pred = (below && o.y >= pred) ? o.y : pred;
						}
					}
				}
				if (i < nl)
				{
					uint32_t const at = (cnt[r] & 0x7FFFu) + pos;
					Rc[me.x >> 16] = (uint16_t) at;
					// the last column in which this word differs from the next smaller sibling (later columns in the higher bits)
					Dc[at] = pos == 0u ? Dp[r] : (uint16_t) ((uint32_t) N * g + (31u - (uint32_t) __clz((int) (w ^ pred))) / (uint32_t) BITS + 1u);
				}
			}
			__syncthreads();
			{ uint16_t *x_ = Rp; Rp = Rc; Rc = x_; x_ = Dp; Dp = Dc; Dc = x_; }
			nprev = nt;
			level_base += nt;
		}

		BT_STAMP(4);
		// ---------------- outputs
#pragma unroll 1
		for (uint32_t wi = tid; wi < nwords; wi += BT_T)
		{
			uint32_t xd[N], cur[IW];
			bt_fetch<BITS>(msa, ld, kend, kend, wi, nwords, idw, true, xd, cur);
			uint32_t const r0 = wi * (uint32_t) N;
#pragma unroll
			for (int r = 0; r < N; ++r)
				if (r0 + (uint32_t) r < m) rank[ob + r0 + r] = Rp[(cur[r >> 1] >> (16 * (r & 1))) & 0x3FFFu];
		}
		for (uint32_t j = tid; j < nprev; j += BT_T) keyd[ob + j] = (uint32_t) (k0 + Dp[j]);
		if (tid == 0) nkeys[b] = nprev;
		__syncthreads();                                       // (LDS and the workspace are the next block's)
#ifdef FSEQ_BT_STAMPS
		BT_STAMP(5);
		if ((tid == 0 || tid == 1000) && (b == 0 || b == 100 || b == 6))
			printf("bt stamps block %u thread %u: clear %lld | loads + transpose %lld | look-ups %lld | wait %lld | ids of found pairs %lld | dump %lld | phase 2 %lld | outputs %lld | keys %u\n",
			       b, tid, bt_acc[0], bt_acc[6], bt_acc[1], bt_acc[2], bt_acc[7], bt_acc[3], bt_acc[4], bt_acc[5], nprev);
#endif
	}
}

} // namespace fseq
