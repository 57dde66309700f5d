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
//            rows' 32-bit group words and looks (class so far, word) up in a hash table in LDS that holds the FULL 46-bit
//            pair -- equal pairs meet in one slot, different pairs never do, so the classes are exact, no fingerprints
//            to verify -- the slot number is the row's new class id.  After a group the occupied slots are the nodes of
//            that trie level, {slot, parent slot, word}; they go to a per-workgroup workspace (L2 / HBM).
//   phase 2  the trie is ranked level by level in LDS: the nodes of a level in the order of (rank of the parent, word).
//            Nearly every parent has one child (a new class appears only where a row first differs): count the children
//            per parent rank, prefix sum, single children take their parent's place and divergence; the few nodes with
//            siblings are collected in a list and ranked among them by comparing words; a sibling's divergence is the
//            last column in which it differs from the next smaller one.
//   output   rank of a row = rank of its last class (registers -> 16 words per thread and slot), keyd, nkeys: exactly
//            what blockkeys_tree_stream writes.
// Whatever does not fit (a level with more than BT_NK_MAX nodes, more than BT_LIST nodes with siblings in one level:
// diverse data) is given up: todo[b] = 1, and the key-space tree does that block.  One workgroup of 1024 threads per
// CU, blocks round-robin; 2 bits per symbol (sigma <= 4) and m <= 16 * 1024 * SL rows.
#pragma once

#include "fseq_blockkeys.hpp"

namespace fseq {

constexpr uint32_t BT_SLOTS = 16384;              // hash table: 16,384 pairs of 8 bytes = 128 KB of LDS
constexpr uint32_t BT_NK_MAX = 12288;             // nodes per trie level (load <= 0.75)
constexpr uint32_t BT_LIST = 2048;                // nodes with siblings per level
constexpr uint32_t BT_MAXPROBE = BT_SLOTS;        // (a full turn: whether a level fits never depends on the order of the inserts)
constexpr int BT_T = 1024;
constexpr int BT_PER = BT_NK_MAX / BT_T;          // entries of the rank-indexed arrays per thread (12)

// LDS (bytes): phase 1 = the table; phase 2 = rank by slot (two), divergence by rank (two), children per rank, the list
constexpr size_t BT_OFF_RP = 0, BT_OFF_RC = 32768, BT_OFF_DP = 65536, BT_OFF_DC = BT_OFF_DP + 2 * BT_NK_MAX,
                 BT_OFF_CNT = BT_OFF_DC + 2 * BT_NK_MAX, BT_OFF_LIST = BT_OFF_CNT + 2 * BT_NK_MAX,
                 BT_OFF_VARS = BT_OFF_LIST + 8 * BT_LIST;
static_assert(BT_OFF_VARS >= 8 * (size_t) BT_SLOTS, "the small variables lie behind the table");
constexpr size_t BT_LDS_BYTES = BT_OFF_VARS + 256;

// words per WORKGROUP of the workspace: node counts per level, the nodes (two words each), the nodes with siblings of one level
__host__ __device__ inline uint32_t blocktrie_levels(uint32_t B) { return (B + 15u) / 16u; }
__host__ __device__ inline size_t blocktrie_ws_words(uint32_t B)
{
	return (size_t) blocktrie_levels(B) * (1u + 2u * (size_t) BT_NK_MAX) + 2u * (size_t) BT_NK_MAX + 64;
}

// 16 x 16 matrix of 2-bit elements: x[c] bits [2r, 2r + 1]  ->  x[r] bits [2c, 2c + 1]  (four stages, written out: as a loop
// over the stage the compiler indexed the registers at run time)
template <int J>
__device__ __forceinline__ void bt_transpose_stage(uint32_t (&x)[16])
{
	constexpr uint32_t mask = J == 8 ? 0x0000FFFFu : J == 4 ? 0x00FF00FFu : J == 2 ? 0x0F0F0F0Fu : 0x33333333u;
#pragma unroll
	for (int i = 0; i < 16; ++i)
		if (!(i & J))
		{
			uint32_t const t = ((x[i] >> (2 * J)) ^ x[i + J]) & mask;
			x[i + J] ^= t;
			x[i] ^= t << (2 * J);
		}
}
__device__ __forceinline__ void bt_transpose16(uint32_t (&x)[16])
{
	bt_transpose_stage<8>(x);
	bt_transpose_stage<4>(x);
	bt_transpose_stage<2>(x);
	bt_transpose_stage<1>(x);
}

// 14 bits out of (class so far, word): 24-bit multiplies (full rate; a 32-bit multiply is a quarter-rate instruction here)
__device__ __forceinline__ uint32_t bt_hash(uint32_t parent, uint32_t word)
{
	uint32_t const x = __umul24(word, 0x9E3779u) ^ __umul24(word >> 8, 0x85EBCBu) ^ __umul24(parent, 0xC2B2AFu);
	return (x >> 12) & (BT_SLOTS - 1u);
}

// slot of the pair in the table (inserted if new); BT_SLOTS: the table is full
__device__ __forceinline__ uint32_t bt_find(unsigned long long *tab, uint32_t parent, uint32_t word, uint32_t h, unsigned long long first)
{
	unsigned long long const key = ((unsigned long long) parent << 32) | word;
	unsigned long long cur = first;
#pragma unroll 1
	for (uint32_t probe = 0; probe < BT_MAXPROBE; ++probe)
	{
		if (cur == key) return h;
		if (cur == ~0ull)
		{
			unsigned long long const old = atomicCAS(tab + h, ~0ull, key);
			if (old == ~0ull || old == key) return h;
		}
		h = (h + 1u) & (BT_SLOTS - 1u);
		cur = tab[h];
	}
	return BT_SLOTS;
}

#ifdef FSEQ_BT_STAMPS
#define BT_STAMP(i) do { long long const t_ = clock64(); bt_acc[i] += t_ - bt_last; bt_last = t_; } while (0)
#else
#define BT_STAMP(i) do {} while (0)
#endif

template <int SL>
__global__ __launch_bounds__(BT_T) void k_blocktrie(
	uint8_t const *__restrict__ msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t nblk,
	uint32_t *__restrict__ rank, uint32_t *__restrict__ keyd, uint32_t *__restrict__ nkeys, uint64_t col0,
	uint32_t *__restrict__ ws, size_t ws_per_group, uint32_t *__restrict__ given_up /* [0]: blocks given up */, uint32_t *__restrict__ todo)
{
	extern __shared__ __attribute__((aligned(16))) char smem[];
	unsigned long long *const tab = reinterpret_cast<unsigned long long *>(smem);
	uint16_t *const RA = reinterpret_cast<uint16_t *>(smem + BT_OFF_RP);
	uint16_t *const RB = reinterpret_cast<uint16_t *>(smem + BT_OFF_RC);
	uint16_t *const DA = reinterpret_cast<uint16_t *>(smem + BT_OFF_DP);
	uint16_t *const DB = reinterpret_cast<uint16_t *>(smem + BT_OFF_DC);
	uint16_t *const cnt = reinterpret_cast<uint16_t *>(smem + BT_OFF_CNT);
	uint32_t *const cnt32 = reinterpret_cast<uint32_t *>(smem + BT_OFF_CNT);
	uint2 *const list = reinterpret_cast<uint2 *>(smem + BT_OFF_LIST);
	uint32_t *const vars = reinterpret_cast<uint32_t *>(smem + BT_OFF_VARS);      // [0 .. 2]: node counters of the levels (in turn); [3]: no room; [4]: list entries; [8 ..]: scan scratch
	uint32_t const tid = threadIdx.x, lane = lane_id(), wave = wave_id();
	uint32_t const nwords = (m + 15u) / 16u;
	uint32_t *const wsg = ws + (size_t) blockIdx.x * ws_per_group;

	for (uint32_t b = blockIdx.x; b < nblk; b += gridDim.x)
	{
		uint64_t const k0 = col0 + (uint64_t) b * B;
		uint64_t const kend = (k0 + B < n) ? k0 + B : n;
		uint32_t const levels = ((uint32_t) (kend - k0) + 15u) / 16u;
		uint32_t *const hdr = wsg;
		uint2 *const nodes = reinterpret_cast<uint2 *>(wsg + ((levels + 1u) & ~1u));
		uint2 *const glist = nodes + (size_t) levels * BT_NK_MAX;      // the nodes with siblings of the level at hand
		size_t const ob = (size_t) b * m;

#ifdef FSEQ_BT_STAMPS
		long long bt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bt_last = clock64();
#endif
		// ---------------- phase 1: the classes, group by group from the last one
		uint32_t ids[SL][8];
#pragma unroll
		for (int s = 0; s < SL; ++s)
#pragma unroll
			for (int q = 0; q < 8; ++q) ids[s][q] = 0u;
		if (tid == 0) { vars[0] = 0u; vars[1] = 0u; vars[2] = 0u; vars[3] = 0u; }
		uint32_t level_base = 0;
		bool bad = false;
		for (uint32_t t = 0; t < levels; ++t)
		{
			uint32_t const g = levels - 1u - t;
			uint64_t const kc = k0 + 16ull * g;
			{
				uint4 *const t4 = reinterpret_cast<uint4 *>(smem);
				uint4 const ones = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
				for (int i = 0; i < (int) (BT_SLOTS * 8 / 16 / BT_T); ++i) t4[tid + i * BT_T] = ones;
			}
			if (tid == 0) vars[(t + 1u) % 3u] = 0u;            // (the counter of the level before is still being read)
			__syncthreads();
			BT_STAMP(0);
			uint32_t full = 0;
			// (the slots in a loop that stays a loop -- unrolled, the kernel is 280 KB of code and lives in the instruction
			// cache's misses: the ids of the slot at hand are picked out of the register array and put back with selects)
#pragma unroll 1
			for (int s = 0; s < SL; ++s)
			{
				uint32_t const wi = (uint32_t) s * BT_T + tid;
				if (wi < nwords)
				{
					uint32_t cur[8];
#pragma unroll
					for (int s2 = 0; s2 < SL; ++s2)
						if (s == s2)
						{
							asm volatile("" ::: "memory");          // (a branch on the uniform s, not eight selects per slot)
#pragma unroll
							for (int q = 0; q < 8; ++q) cur[q] = ids[s2][q];
						}
					uint32_t x[16];
#pragma unroll
					for (int c = 0; c < 16; ++c)
						x[c] = (kc + (uint64_t) c < kend) ? *reinterpret_cast<uint32_t const *>(msa + (kc + (uint64_t) c) * ld + 4ull * wi) : 0u;
					bt_transpose16(x);
					if (wi * 16u + 16u > m)
					{
						// the last word of a column: the positions behind row m - 1 follow its first row (same class throughout)
						uint32_t const valid = m - wi * 16u;
#pragma unroll
						for (int r = 1; r < 16; ++r) x[r] = ((uint32_t) r < valid) ? x[r] : x[0];
					}
#pragma unroll
					for (int q4 = 0; q4 < 4; ++q4)
					{
						// four rows at a time, no branch while every look-up finds its pair where the hash points
						uint32_t par[4], hh[4];
						unsigned long long val[4];
#pragma unroll
						for (int u = 0; u < 4; ++u)
						{
							int const r = q4 * 4 + u;
							par[u] = (r & 1) ? (cur[r >> 1] >> 16) : (cur[r >> 1] & 0xFFFFu);
							hh[u] = bt_hash(par[u], x[r]);
							val[u] = tab[hh[u]];
						}
						bool miss = false;
#pragma unroll
						for (int u = 0; u < 4; ++u) miss |= val[u] != (((unsigned long long) par[u] << 32) | x[q4 * 4 + u]);
						if (miss)
						{
#pragma unroll
							for (int u = 0; u < 4; ++u)
							{
								hh[u] = bt_find(tab, par[u], x[q4 * 4 + u], hh[u], val[u]);
								if (hh[u] == BT_SLOTS) { full = 1u; hh[u] = 0u; }
							}
						}
						cur[2 * q4] = hh[0] | (hh[1] << 16);
						cur[2 * q4 + 1] = hh[2] | (hh[3] << 16);
					}
#pragma unroll
					for (int s2 = 0; s2 < SL; ++s2)
						if (s == s2)
						{
							asm volatile("" ::: "memory");
#pragma unroll
							for (int q = 0; q < 8; ++q) ids[s2][q] = cur[q];
						}
				}
			}
			if (full) vars[3] = 1u;
			BT_STAMP(1);
			__syncthreads();
			BT_STAMP(2);
			// the occupied slots are the nodes of this level
			{
				uint32_t mine = 0;
#pragma unroll
				for (int i = 0; i < (int) (BT_SLOTS / BT_T); ++i) mine += tab[tid + i * BT_T] != ~0ull ? 1u : 0u;
				uint32_t const inc = wave_incl_add(mine);
				uint32_t base = 0;
				if (lane == 63) base = atomicAdd(vars + t % 3u, inc);
				base = readlane_u32(base, 63) + inc - mine;
				if (level_base + base + mine <= (size_t) levels * BT_NK_MAX)
				{
#pragma unroll 4
					for (int i = 0; i < (int) (BT_SLOTS / BT_T); ++i)
					{
						unsigned long long const e = tab[tid + i * BT_T];
						if (e != ~0ull)
						{
							uint32_t const slot = tid + i * BT_T;
							nodes[level_base + base] = make_uint2(slot | ((uint32_t) (e >> 32) << 16), (uint32_t) e);
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
		__syncthreads();                                       // (the table is history: its LDS is the arrays of phase 2 now)
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
					uint32_t const slot = mynd[k].x & 0x3FFFu;
					uint32_t const r = myr[k];
					uint32_t const e = cnt[r];
					if (!(e & 0x8000u)) { Rc[slot] = (uint16_t) e; Dc[e] = Dp[r]; }
					else
					{
						uint32_t const at = atomicAdd(vars + 4, 1u);
						uint2 const v = make_uint2(r | (slot << 16), mynd[k].y);
						if (at < BT_LIST) list[at] = v;                   // (the first piece is in place already)
						glist[at] = v;
					}
				}
			__syncthreads();
			// the nodes with siblings: place among the children of their parent by comparing words, pair by pair -- the list
			// is a hundred entries but for a level in which many rows change (every row, where a recombination block of the
			// founders ends) -- through LDS in pieces of BT_LIST
			uint32_t const nl = vars[4];
			if (nl)
			{
				uint2 me[BT_PER];
				uint32_t pos[BT_PER], pred[BT_PER];
#pragma unroll
				for (int k = 0; k < BT_PER; ++k)
				{
					me[k] = (tid + k * BT_T < nl) ? (tid + k * BT_T < BT_LIST ? list[tid + k * BT_T] : glist[tid + k * BT_T]) : make_uint2(0xFFFFu, 0u);      // (no such parent rank)
					pos[k] = 0u; pred[k] = 0u;
				}
				for (uint32_t c0 = 0; c0 < nl; c0 += BT_LIST)
				{
					uint32_t const cn = min(BT_LIST, nl - c0);
					if (c0)
					{
						__syncthreads();
						for (uint32_t j = tid; j < cn; j += BT_T) list[j] = glist[c0 + j];
						__syncthreads();
					}
#pragma unroll
					for (int k = 0; k < BT_PER; ++k)
					{
						if ((uint32_t) k * BT_T >= nl) break;                 // (uniform)
						uint32_t const r = me[k].x & 0xFFFFu, w = me[k].y;
						uint32_t p_ = pos[k], q_ = pred[k];
						for (uint32_t j = 0; j < cn; ++j)
						{
							uint2 const o = list[j];
							bool const below = (o.x & 0xFFFFu) == r && o.y < w;
							p_ += below ? 1u : 0u;
							q_ = (below && o.y >= q_) ? o.y : q_;
						}
						pos[k] = p_; pred[k] = q_;
					}
				}
#pragma unroll
				for (int k = 0; k < BT_PER; ++k)
					if (tid + k * BT_T < nl)
					{
						uint32_t const r = me[k].x & 0xFFFFu;
						uint32_t const at = (cnt[r] & 0x7FFFu) + pos[k];
						Rc[me[k].x >> 16] = (uint16_t) at;
						// the last column in which this word differs from the next smaller sibling (later columns in the higher bits)
						Dc[at] = pos[k] == 0u ? Dp[r] : (uint16_t) (16u * g + ((31u - (uint32_t) __clz((int) (me[k].y ^ pred[k]))) >> 1) + 1u);
					}
			}
			__syncthreads();
			{ uint16_t *x_ = Rp; Rp = Rc; Rc = x_; x_ = Dp; Dp = Dc; Dc = x_; }
			nprev = nt;
			level_base += nt;
		}
		if (bad)
		{
			if (tid == 0) { todo[b] = 1u; atomicAdd(given_up, 1u); }
			__syncthreads();
			continue;
		}

		BT_STAMP(4);
		// ---------------- outputs
#pragma unroll 1
		for (int s = 0; s < SL; ++s)
		{
			uint32_t const wi = (uint32_t) s * BT_T + tid;
			if (wi < nwords)
			{
				uint32_t cur[8];
#pragma unroll
				for (int s2 = 0; s2 < SL; ++s2)
					if (s == s2)
					{
						asm volatile("" ::: "memory");
#pragma unroll
						for (int q = 0; q < 8; ++q) cur[q] = ids[s2][q];
					}
				uint32_t const r0 = wi * 16u;
#pragma unroll
				for (int r = 0; r < 16; ++r)
					if (r0 + (uint32_t) r < m) rank[ob + r0 + r] = Rp[(cur[r >> 1] >> (16 * (r & 1))) & 0x3FFFu];
			}
		}
		for (uint32_t j = tid; j < nprev; j += BT_T) keyd[ob + j] = (uint32_t) (k0 + Dp[j]);
		if (tid == 0) nkeys[b] = nprev;
		__syncthreads();
#ifdef FSEQ_BT_STAMPS
		BT_STAMP(5);
		if ((tid == 0 || tid == 1000) && (b == 0 || b == 100 || b == 6))
			printf("bt stamps block %u thread %u: clear %lld | slots %lld | wait %lld | dump %lld | phase 2 %lld | outputs %lld | keys %u\n",
			       b, tid, bt_acc[0], bt_acc[1], bt_acc[2], bt_acc[3], bt_acc[4], bt_acc[5], nprev);
#endif                                       // (LDS and the workspace are the next block's)
	}
}

} // namespace fseq
