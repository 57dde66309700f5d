// fseq_api.hip -- C ABI (include/fseq.h) + host orchestration of the HIP segmentation path.
//
// Host side of what segmentation_lp_context drives through libdispatch queues
// (founder-sequences/segmentation_lp_context.cc:26-390): here one HIP stream, kernels per phase,
// and only the O(S) pieces (traceback walk lp.cc:191-224, segment merge lp.cc:335-390) on the CPU.
// There is deliberately NO CPU fallback for the column work: if the device or a kernel shape is
// unavailable the call fails with an error code.
// The context and the helpers shared with the other translation unit are in fseq_ctx.hpp; the host joiners, their device
// front and the output writers are csrc/fseq_api_join.hip.
#include "fseq_ctx.hpp"
#include "fseq_kernels.hpp"
#include "fseq_dp.hpp"
#include "fseq_dpspec.hpp"
#include "fseq_stream.hpp"
#include "fseq_stream2.hpp"
#include "fseq_chainsort.hpp"
#include "fseq_blockkeys.hpp"
#include "fseq_blocktrie.hpp"
#include "fseq_rowshard.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

// One range per phase (rocprofv3 --marker-trace shows them when the library is built against roctx).  The pushes and pops
// are counted, so a test can tell that the ranges are there and balanced without a profiler (fseq_debug_ranges).
static std::atomic<uint64_t> g_range_pushes{0}, g_range_pops{0};
#ifdef FSEQ_WITH_ROCTX
#include <rocprofiler-sdk-roctx/roctx.h>
#define FSEQ_RANGE_PUSH(name) do { g_range_pushes.fetch_add(1, std::memory_order_relaxed); (void) roctxRangePushA(name); } while (0)
#define FSEQ_RANGE_POP() do { g_range_pops.fetch_add(1, std::memory_order_relaxed); (void) roctxRangePop(); } while (0)
#else
#define FSEQ_RANGE_PUSH(name) do { g_range_pushes.fetch_add(1, std::memory_order_relaxed); } while (0)
#define FSEQ_RANGE_POP() do { g_range_pops.fetch_add(1, std::memory_order_relaxed); } while (0)
#endif

// a phase's range: popped where the phase ends -- or where the function leaves early (a retry with a larger list capacity, an
// attempt that runs again, an error), so that pushes and pops stay balanced on every path
struct RangeScope {
	bool open = true;
	explicit RangeScope(char const *name) { FSEQ_RANGE_PUSH(name); (void) name; }
	void end() { if (open) { FSEQ_RANGE_POP(); open = false; } }
	~RangeScope() { end(); }
	RangeScope(RangeScope const &) = delete;
	RangeScope &operator=(RangeScope const &) = delete;
};

using namespace fseq;

namespace {

constexpr size_t LDS_LIMIT = 160 * 1024;
constexpr uint64_t STREAM_BLOCK_TARGET_ALL_ROWS = 1600;   // columns per block the streamed regime aims for (prepare_geometry / block_geometry) ...
// [r5] ... and when phase C runs on the blocks' representatives: a block of ~800 columns of BASELINE C4 has ~6,600 of them, and
// the ~10,700 of a block in which the founders recombine still fit the largest configuration (11,264)
constexpr uint64_t STREAM_BLOCK_TARGET_REDUCED = 800;
#define STREAM_BLOCK_TARGET (c->tune.no_reduced ? STREAM_BLOCK_TARGET_ALL_ROWS : (c->tune.stream_block ? (uint64_t) c->tune.stream_block : STREAM_BLOCK_TARGET_REDUCED))
#ifndef FSEQ_X_FLOOR_VALUE
#define FSEQ_X_FLOOR_VALUE 63u
#endif
constexpr uint32_t FSEQ_X_FLOOR = FSEQ_X_FLOOR_VALUE;   // smallest per-column list capacity tried (the estimate and the retries raise it)

} // namespace

namespace {

// Pinned host staging.  A copy between the device and pageable host memory is staged by the runtime -- one blocking round
// trip of 20-50 microseconds each, and a step had a dozen of them (flags, counts, the traceback, thresholds: a fifth of a
// BASELINE C2 step).  pin_reserve(bytes) opens a stage (what the previous one handed out is dead), pin_take carves it.
int pin_reserve(fseq_ctx *c, size_t bytes)
{
	c->pin_used = 0;
	if (c->pin_cap >= bytes) return FSEQ_OK;
	if (c->h_pin) (void) hipHostFree(c->h_pin);
	c->h_pin = nullptr; c->pin_cap = 0;
	size_t const cap = std::max<size_t>((bytes + 4095) & ~size_t(4095), size_t(1) << 20);
	hipError_t const e = hipHostMalloc(reinterpret_cast<void **>(&c->h_pin), cap, hipHostMallocDefault);
	if (e != hipSuccess) { c->h_pin = nullptr; return fail(c, FSEQ_E_OOM, "hipHostMalloc of the staging buffer", e); }
	c->pin_cap = cap;
	return FSEQ_OK;
}
template <typename U>
U *pin_take(fseq_ctx *c, size_t count)
{
	size_t const at = (c->pin_used + 15) & ~size_t(15);
	c->pin_used = at + count * sizeof(U);
	return c->pin_used <= c->pin_cap ? reinterpret_cast<U *>(c->h_pin + at) : nullptr;      // (nullptr: the stage was reserved too small -- a bug)
}

void free_msa(fseq_ctx *c)
{
	if (c->own_msa) dev_free(c, &c->d_msa_alloc);
	c->d_msa_alloc = nullptr;
	c->d_msa = nullptr;
	c->own_msa = false;
	c->have_input = false;
}

// columns this context holds: all of them, or the rank's share of a sharded run
uint64_t held_lo(fseq_ctx const *c) { return c->sh.on ? c->sh.c_lo : 0; }
uint64_t held_hi(fseq_ctx const *c) { return c->sh.on ? c->sh.c_end : c->p.n; }

int alloc_msa(fseq_ctx *c)
{
	free_msa(c);
	c->bsh = c->sigma <= 4 ? 2u : c->sigma <= 16 ? 1u : 0u;
	c->ld = ((size_t) sym_bytes(c->p.m, c->bsh) + 15) & ~size_t(15);
	int rc = dev_alloc(c, &c->d_msa_alloc, c->ld * (held_hi(c) - held_lo(c)) + 16);
	if (rc) return rc;
	c->d_msa = c->d_msa_alloc - held_lo(c) * c->ld;          // column k at d_msa + k * ld for the held columns
	c->own_msa = true;
	return FSEQ_OK;
}

// Block structure of phases A-C.  Sharded: every rank is one hyper-block of phase B (shard_q groups of
// chain_fan^shard_k blocks), so the only exchange of phase B is the W composite key blocks of the ranks.
void block_geometry(fseq_ctx *c)
{
	fseq_params const &p = c->p;
	bool const streamed = p.m > 11264u;
	Shard &sh = c->sh;
	if (p.block_len) c->B = p.block_len;
	else if (c->auto_B) c->B = c->auto_B;                    // (prepare_geometry's second look, below)
	else if (!sh.on)
	{
		// LDS-resident kernels: ~1024 blocks (2-4 workgroups per CU).  Streamed kernels stage a whole column
		// in LDS (one workgroup per CU) and pay the phase-B chain per block and per row: ~256 blocks.
		// (32-bit LDS state and long inputs: ~4096 blocks -- measured on BASELINE C3: phase C 7.7 -> 7.3 ms with the finer
		// grain, phase B 0.50 -> 0.74 ms with its two more levels; blocks of fewer than ~200 columns lose more to the
		// per-block prologues and to phase B than they gain)
		uint64_t target = streamed ? 256u : 1024u;
		if (!streamed && p.m <= 7168u && p.n >= 4096u * 200u) target = 4096u;
		uint64_t b = (p.n + target - 1) / target;
		if (b < 16) b = 16;
		if (b > 4096) b = 4096;
		c->B = (uint32_t) b;
	}
	else
	{
		// sharded: the same per rank, and -- streamed kernels are one workgroup per CU -- a whole number of waves of
		// workgroups per rank (256 CUs x k blocks of <= 4096 columns), so that no rank ends on a nearly empty wave
		int ncu = 0;
		(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p.device);
		if (ncu < 1) ncu = 256;
		uint64_t const cols = (p.n + sh.world - 1) / sh.world;
		// (streamed rows: blocks of ~1,600 columns, see prepare_geometry)
		uint64_t const per_cu_cols = streamed ? STREAM_BLOCK_TARGET : 4096u;
		uint64_t k = std::max<uint64_t>(1, (cols + (uint64_t) ncu * per_cu_cols / 2) / ((uint64_t) ncu * per_cu_cols));
		// (the second form of the streamed phase C runs -- and was tuned for -- two workgroups per CU: a rank's blocks are whole
		// rounds of 2 x CUs workgroups, also when one workgroup per CU would hold its columns: BASELINE C4 on 8 ranks is 512
		// blocks of 1,221 columns per rank, not 256 of 2,442 with every CU's second slot empty)
		if ((k & 1u) && (uint64_t) p.m + 4096u < (1ull << 19) && !c->tune.stream_plain_scan && c->tune.stream2 != "0") ++k;
		uint64_t per = streamed ? (uint64_t) ncu * k : 1024u;
		uint64_t b = (p.n + per * sh.world - 1) / (per * sh.world);
		if (b < 16) b = 16;
		c->B = (uint32_t) b;
	}
	// sharded: the DP round that starts on a rank's last column reads the lists of the RL - 1 columns behind it; the
	// rank produces them itself by running into the next rank's first block -- which must hold them: B >= RL
	uint32_t halo = 0;
	if (sh.on && p.n >= 2 * p.segment_length)
	{
		halo = dp_schedule((uint32_t) p.segment_length, (uint32_t) p.n).RL;
		if (c->B < halo) c->B = halo;
	}
	if (c->B > p.n) c->B = (uint32_t) p.n;
	c->nblocks = (uint32_t) ((p.n + c->B - 1) / c->B);
	if (sh.on)
	{
		// A rank is one hyper-block of phase B: q groups of F^k blocks, composed level by level with fan F (k launches of
		// <= F serial steps up, the q composites into the hyper key block, and the same down again).  q F^k >= the
		// blocks a rank needs; (k, q) with the fewest serial steps among those that keep every rank busy.
		uint32_t const per = (c->nblocks + sh.world - 1) / sh.world;
		uint32_t F = 4;
		if (c->tune.chain_fan) F = (uint32_t) c->tune.chain_fan;
		uint32_t best_k = 0, best_q = std::max(1u, per), best_cost = ~0u;
		{
			uint64_t pw = 1;
			for (uint32_t k = 0; pw <= per; ++k, pw *= F)
			{
				uint32_t const q = (uint32_t) ((per + pw - 1) / pw);
				uint64_t const bpr = (uint64_t) q * pw;
				bool const all_busy = bpr == per || bpr * (sh.world - 1) < c->nblocks;      // the last rank still owns blocks
				uint32_t const cost = F * k + q;
				if ((all_busy || k == 0) && cost < best_cost) { best_cost = cost; best_k = k; best_q = q; }
			}
		}
		c->chain_fan = F; c->shard_k = best_k; c->shard_q = best_q;
		{
			uint64_t pw = 1;
			for (uint32_t i = 0; i < best_k; ++i) pw *= F;
			sh.bpr = (uint32_t) (best_q * pw);
			c->chain_G = (uint32_t) pw; c->chain_G2 = best_q;          // (diagnostics: a rank = chain_G2 groups of chain_G blocks)
		}
		c->n_super = (c->nblocks + c->chain_G - 1) / c->chain_G;
		c->n_hyper = (c->nblocks + sh.bpr - 1) / sh.bpr;
		sh.active = c->n_hyper;                                 // <= world
		sh.b_lo = std::min<uint64_t>(c->nblocks, (uint64_t) sh.rank * sh.bpr);
		sh.b_hi = std::min<uint64_t>(c->nblocks, (uint64_t) (sh.rank + 1) * sh.bpr);
		sh.c_lo = std::min<uint64_t>(p.n, (uint64_t) sh.b_lo * c->B);
		sh.c_hi = std::min<uint64_t>(p.n, (uint64_t) sh.b_hi * c->B);
		sh.c_end = (sh.b_hi > sh.b_lo) ? std::min<uint64_t>(p.n, sh.c_hi + halo) : sh.c_hi;
		return;
	}
	{
		// Phase B is serial over key blocks, so it is applied recursively: compose groups of G blocks from the identity
		// (parallel), groups of G of those, ... until at most G are left, chain them, expand level by level.  Serial depth
		// = G steps per launch, 2 levels - 1 launches (+ about half a step of launch gap each): G = 4 for 100..10^4
		// blocks (1024 blocks: 9 launches of <= 4 steps instead of the 5 x 11 of a three-level chain).
		uint32_t best_g = c->nblocks, best_cost = ~0u;
		for (uint32_t g = 2; g <= 64 && g < std::max(3u, c->nblocks); ++g)
		{
			uint32_t lv = 1;
			for (uint64_t cap = g; cap < c->nblocks; cap *= g) ++lv;
			uint32_t const cost = (2u * lv - 1u) * (2u * g + 1u);
			if (cost < best_cost) { best_cost = cost; best_g = g; }
		}
		// [r5] streamed rows: a step is a launch sequence over all the chains of a level (fseq_chainsort.hpp), bound by what it
		// moves, not by its depth -- so the fan weighs the steps in all (the blocks of every level once on the way up, all but
		// every chain's last on the way down: ~N (2g - 1) / (g - 1)) against the rounds of launches, (2g - 1) per level.
		// BASELINE C4 (6,143 blocks, 100,000 rows), phase B: fan 3: 69.6 ms, 4: 61.6, 6: 56.5, 8: 54.1, 12: 50.9, 16: 53.7, 32: 57.5
		KernelSet probe;
		if (!select_kernels(p.m, c->sigma, &probe, c->tune.no_emitter_wave) && c->nblocks > 8)
		{
			double best = 1e300;
			for (uint32_t g = 2; g <= 64 && g < c->nblocks; ++g)
			{
				double steps = 0, rounds = 0;
				uint64_t cnt = c->nblocks;
				while (cnt > g)
				{
					steps += (double) cnt * (2.0 * g - 1.0) / g;       // up: every item; down: all but the last of every group
					rounds += 2.0 * g - 1.0;
					cnt = (cnt + g - 1) / g;
				}
				steps += (double) cnt; rounds += (double) cnt;           // the top chain
				double const cost = steps * 3.5e-3 * ((double) p.m / 1e5) + rounds * 0.05;
				if (cost < best) { best = cost; best_g = g; }
			}
		}
		if (c->nblocks <= 8) best_g = std::max(1u, c->nblocks);        // one chain
		if (c->tune.two_level_chain) best_g = std::max(2u, (uint32_t) std::ceil(std::sqrt((double) c->nblocks)));
		if (c->tune.chain_fan) best_g = (uint32_t) c->tune.chain_fan;
		c->chain_fan = best_g;
		c->chain_G = best_g; c->n_super = (c->nblocks + best_g - 1) / best_g;      // (diagnostics)
		c->chain_G2 = 0; c->n_hyper = 0;
	}
}

int prepare_geometry(fseq_ctx *c)
{
	fseq_params const &p = c->p;
	c->auto_B = 0;
	block_geometry(c);
	uint32_t n2 = 1;
	while (n2 < p.m) n2 <<= 1;
	if (n2 < 2) n2 = 2;
	c->N2 = n2;
	{
		uint32_t bits = 1;
		while ((1u << bits) < c->sigma) ++bits;
		c->npass = (bits + 1) / 2;               // 2-bit digit passes per column
	}
	if (c->sigma > 256) return fail(c, FSEQ_E_UNSUPPORTED, "alphabet larger than 256 symbols");
	c->use_stream = !select_kernels(p.m, c->sigma, &c->ks, c->tune.no_emitter_wave);
	if (c->use_stream)
	{
		// rows beyond the LDS-resident configurations: the order streams through HBM / L2
		if (sym_bytes(p.m, c->bsh) > STREAM_MAX_COLBYTES)
			return fail(c, FSEQ_E_UNSUPPORTED, "more rows than this build handles (one packed column must fit LDS: 147456 bytes)");
		// the tile staging buffer of stream_pass (64 KiB) when the staged column leaves room for it
		c->stream_staged = stream_lds_bytes(sym_bytes(p.m, c->bsh), true) <= LDS_LIMIT;
		size_t const lds = stream_lds_bytes(sym_bytes(p.m, c->bsh), c->stream_staged);
		HIP_TRY(c, allow_lds(k_colblock_stream<MODE_RANK>, lds));
		HIP_TRY(c, allow_lds(k_colblock_stream<MODE_SNAP>, lds));
		HIP_TRY(c, allow_lds(k_colblock_stream<MODE_RANK, true>, lds));
		HIP_TRY(c, allow_lds(k_colblock_stream<MODE_SNAP, true>, lds));
		HIP_TRY(c, allow_lds(k_columns_stream<19>, lds));
		HIP_TRY(c, allow_lds(k_columns_stream<0>, lds));
		HIP_TRY(c, allow_lds(k_chain_stream<false>, stream_lds_bytes(0, true)));
		HIP_TRY(c, allow_lds(k_chain_stream<true>, stream_lds_bytes(0, true)));
		HIP_TRY(c, allow_lds(k_chain_stream_sort, chainsort_lds_bytes()));
		HIP_TRY(c, allow_lds(k_chain_snap_stream, chainsort_lds_bytes()));
		HIP_TRY(c, allow_lds(k_cm_emit, stream_lds_bytes(0, false)));
		// phase C in its second form (fseq_stream2.hpp) while every value id (< m + B) fits the key shift of its tile
		// (FSEQ_STREAM2=T,E[,0] picks another configuration [8-byte rows], FSEQ_STREAM2=0 and FSEQ_STREAM_PLAIN_SCAN keep the first form)
		c->s2 = Stream2Config{};
		{
			uint32_t T2 = 512, E2 = 8, P2 = 1;
			bool off = c->tune.stream_plain_scan;
			if (!c->tune.stream2.empty() && sscanf(c->tune.stream2.c_str(), "%u,%u,%u", &T2, &E2, &P2) < 2) off = true;
			Stream2Config cfg;
			if (!off && select_stream2(T2, E2, P2, &cfg) && (uint64_t) p.m + c->B < (1ull << cfg.key_shift) && c->stream_staged)
			{
				size_t const bytes = cfg.lds(sym_bytes(p.m, c->bsh));
				if (bytes <= LDS_LIMIT)
				{
					HIP_TRY(c, cfg.prepare(bytes));
					HIP_TRY(c, allow_lds(k_columns_stream2_prologue, stream_lds_bytes(0, true)));
					c->s2 = cfg; c->s2_lds = bytes;
				}
			}
		}
		// Long inputs (the block length was clamped to 4,096 columns): whole rounds of phase C's workgroups.  BASELINE C4 had
		// 1,221 blocks on 512 slots -- 2.4 rounds, the last one 38 % full: 2.33 s; 1,536 blocks of 3,256 columns: 2.22 s
		// (2,048 and 3,072 blocks the same: phase A gains what phase B loses).
		// [r4] ... and blocks of ~1,600 columns: the streamed key-space tree is cheaper per column in shorter blocks (its merges with
		// the running prefix see fewer distinct keys), and phase B no longer pays for more blocks what it did (fseq_chainsort.hpp).
		// BASELINE C4, blocks x columns: 1,536 x 3,256: A 340 + B 32 = 1,823 ms the step; 2,048 x 2,442: 317 + 41 = 1,823;
		// 3,072 x 1,628: 280 + 50 = 1,791; 4,096 x 1,221: 263 + 64 = 1,793.
		if (!p.block_len && !c->sh.on && !c->auto_B && c->B > STREAM_BLOCK_TARGET && c->B < p.n)
		{
			int ncu = 0;
			(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p.device);
			uint64_t const slots = (uint64_t) std::max(ncu, 1) * (c->s2.T ? c->s2.resident(c->s2_lds) : 1u);
			uint64_t const rounds = std::max<uint64_t>(1, (p.n + slots * STREAM_BLOCK_TARGET / 2) / (slots * STREAM_BLOCK_TARGET));
			uint64_t const b = (p.n + rounds * slots - 1) / (rounds * slots);
			if (c->tune.debug) fprintf(stderr, "[fseq] streamed phase C: %u blocks on %llu workgroup slots -> %llu rounds of %llu columns\n", c->nblocks, (unsigned long long) slots,
			                           (unsigned long long) rounds, (unsigned long long) b);
			if (b >= 256 && b < c->B)
			{
				c->auto_B = (uint32_t) b;
				block_geometry(c);
			}
		}
		// phase A in key space, streamed rows: the bitmap (and its 32-bit prefix counts) take the LDS
		c->bk_cap_words = 12288;                               // two bitmaps + 32-bit prefix counts: 12 B per word
		if (c->tune.blockkeys_cap) c->bk_cap_words = (uint32_t) c->tune.blockkeys_cap;
		c->bk_lds = blockkeys_stream_lds_bytes(c->bk_cap_words, 1024);
		// (bk_merge slices a merge by whole `hi` values: one hi value's Dlo <= m keys must fit the bitmap -- with more rows
		// than bitmap bits a diverse block could overrun it, so such inputs take the column sweep k_colblock_stream<MODE_RANK>)
		if (c->bk_lds <= LDS_LIMIT && (uint64_t) p.m <= (uint64_t) c->bk_cap_words * 32u) HIP_TRY(c, allow_lds(k_blockkeys_stream, c->bk_lds));
		else c->bk_cap_words = 0;
	}
	else
	{
		c->lds_columns = c->ks.columns_lds(c->B);
		if (c->lds_columns > LDS_LIMIT || c->ks.lds_chain > LDS_LIMIT || c->ks.lds_colblock > LDS_LIMIT || c->ks.lds_snap > LDS_LIMIT)
			return fail(c, FSEQ_E_UNSUPPORTED, "block state does not fit the 160 KiB LDS of one CU");
		// phase C works on value ids < m + B in 16-bit keys (partition_step<.., KEY16>); the LDS check above implies it
		if ((uint64_t) p.m + c->B > 65535u) return fail(c, FSEQ_E_UNSUPPORTED, "block length too large for the 16-bit value ids of phase C");
		HIP_TRY(c, c->ks.prepare(c->lds_columns));
		HIP_TRY(c, c->ks.prepare_columns(c->lds_columns));
		// Short inputs: phase C is a few rounds of (CUs x workgroups per CU) blocks, and a last round that is a third full
		// costs a whole one (BASELINE C2: 1,021 blocks on 768 slots; 764 blocks of 131 columns: phase C 0.65 -> 0.59 ms).
		// Below three rounds the block length is refitted to whole rounds (long inputs measured no better for it).
		if (!p.block_len && !c->sh.on && !c->auto_B)
		{
			int ncu = 0;
			(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p.device);
			uint64_t const slots = (uint64_t) std::max(ncu, 1) * c->ks.columns_resident(c->lds_columns);
			if (c->tune.debug) fprintf(stderr, "[fseq] phase C: %u blocks of %u columns on %llu workgroup slots (%zu bytes of LDS each)\n", c->nblocks, c->B, (unsigned long long) slots, c->lds_columns);
			if (c->nblocks > slots && c->nblocks < 3 * slots)
			{
				uint64_t const rounds = (c->nblocks + slots / 2) / slots;
				uint64_t const b = (p.n + rounds * slots - 1) / (rounds * slots);
				if (b >= 16 && b <= 4096 && b != c->B && p.m + b <= 65535u)
				{
					c->auto_B = (uint32_t) b;
					block_geometry(c);
					c->lds_columns = c->ks.columns_lds(c->B);
					// (rounds is rounded down, so the refit can RAISE the block length by up to ~1.5x: when the longer block no
					// longer fits the LDS, or holds another number of workgroups per CU than it was fitted for, keep the first one)
					if (c->lds_columns > LDS_LIMIT || (uint64_t) std::max(ncu, 1) * c->ks.columns_resident(c->lds_columns) != slots)
					{
						c->auto_B = 0;
						block_geometry(c);
						c->lds_columns = c->ks.columns_lds(c->B);
					}
					HIP_TRY(c, c->ks.prepare_columns(c->lds_columns));
				}
			}
		}
		// phase A in key space: the id arrays are (GL + 2) x m halfwords; the two maps take what is left of ~76 KiB
		// (two workgroups per CU) when that holds the leaf map with a quarter to spare, else of the whole CU
		{
			c->bk_T = blockkeys_threads(p.m);
			size_t const arrays = blockkeys_lds_bytes(p.m, 0, (int) c->bk_T, c->ld, c->bsh);
			size_t budget = 76 * 1024;
			if (arrays + 16 * 2560 > budget) budget = LDS_LIMIT - 1024;
			size_t cap = budget > arrays ? (budget - arrays) / 16 : 0;     // two maps of 8-byte {bits, prefix} entries
			cap = std::min<size_t>(cap & ~size_t(63), 32768);
			if (c->tune.blockkeys_cap) cap = (size_t) c->tune.blockkeys_cap;     // tests: force the sliced merges
			c->bk_cap_words = (uint32_t) cap;
			c->bk_lds = blockkeys_lds_bytes(p.m, c->bk_cap_words, (int) c->bk_T, c->ld, c->bsh);
			// (a leaf's columns are staged with two 16-byte pieces per thread)
			if (cap >= 2048 && c->bk_lds <= LDS_LIMIT && (size_t) (8u >> (2u - c->bsh)) * c->ld <= (size_t) c->bk_T * 32) HIP_TRY(c, prepare_blockkeys(c->bk_T, c->bk_lds, c->tune.debug));
			else c->bk_cap_words = 0;
		}
	}
	HIP_TRY(c, allow_lds(k_dp<DP_WHOLE>, dp_lds_bytes()));
	HIP_TRY(c, allow_lds(k_dp<DP_PARTIAL>, dp_lds_bytes()));
	HIP_TRY(c, allow_lds(k_dp<DP_SPEC>, dp_lds_bytes()));
	c->kernels_ready = true;
	return FSEQ_OK;
}

int ensure_work_buffers(fseq_ctx *c, uint32_t X, bool want_ss = true)
{
	fseq_params const &p = c->p;
	size_t const m = p.m;
	int rc;
	if (!c->d_rank)
	{
		// my blocks [bl, bh) (all of them when not sharded); boundary states also behind my last block
		size_t const bl = c->sh.on ? c->sh.b_lo : 0, bh = c->sh.on ? std::max(c->sh.b_hi, c->sh.b_lo) : c->nblocks;
		if ((rc = dev_alloc_range(c, &c->d_rank_alloc, &c->d_rank, bl, bh, m))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_keyd_alloc, &c->d_keyd, bl, bh, m))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_nkeys_alloc, &c->d_nkeys, bl, bh, 1))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_bstate_a_alloc, &c->d_bstate_a, bl, bh + 1, m))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_bstate_d_alloc, &c->d_bstate_d, bl, bh + 1, m))) return rc;
		{
			// the composites of phase B, level by level: until at most chain_fan are left, or -- sharded -- shard_k levels below
			// the hyper key blocks (indexed like the blocks: by their place in the whole alignment; a rank holds its own range)
			uint32_t cnt = c->nblocks;
			uint64_t cols = c->B;
			size_t lo = bl, hi = bh;
			for (uint32_t i = 0; c->sh.on ? i < c->shard_k : cnt > c->chain_fan; ++i)
			{
				fseq_ctx::ChainLevel lv;
				lv.count = (cnt + c->chain_fan - 1) / c->chain_fan;
				lv.cols = cols * c->chain_fan;
				lo = lo / c->chain_fan; hi = (hi + c->chain_fan - 1) / c->chain_fan;
				c->levels.push_back(lv);                               // (pushed at once: free_work releases what is there)
				fseq_ctx::ChainLevel &L = c->levels.back();
				if ((rc = dev_alloc_range(c, &L.rank_alloc, &L.rank, lo, hi, m))) return rc;
				if ((rc = dev_alloc_range(c, &L.keyd_alloc, &L.keyd, lo, hi, m))) return rc;
				if ((rc = dev_alloc_range(c, &L.nkeys_alloc, &L.nkeys, lo, hi, 1))) return rc;
				if ((rc = dev_alloc_range(c, &L.state_a_alloc, &L.state_a, lo, hi + 1, m))) return rc;
				if ((rc = dev_alloc_range(c, &L.state_d_alloc, &L.state_d, lo, hi + 1, m))) return rc;
				cnt = L.count; cols = L.cols;
			}
		}
		if (c->sh.on && c->n_hyper)
		{
			if ((rc = dev_alloc(c, &c->d_hrank, (size_t) c->n_hyper * m))) return rc;
			if ((rc = dev_alloc(c, &c->d_hkeyd, (size_t) c->n_hyper * m))) return rc;
			if ((rc = dev_alloc(c, &c->d_hnkeys, c->n_hyper))) return rc;
			if ((rc = dev_alloc(c, &c->d_hstate_a, ((size_t) c->n_hyper + 1) * m))) return rc;
			if ((rc = dev_alloc(c, &c->d_hstate_d, ((size_t) c->n_hyper + 1) * m))) return rc;
		}
		if ((rc = dev_alloc(c, &c->d_hdr, p.n))) return rc;
		if ((rc = dev_alloc(c, &c->d_flags, 256))) return rc;
		if ((rc = dev_alloc(c, &c->d_recent, c->nblocks + 1))) return rc;
		if (p.n >= 2 * p.segment_length)
		{
			c->dp_size = p.n - p.segment_length + 1;
			c->dp.tstride = (uint32_t) (c->dp_size / 64 + 2);
			if ((rc = dev_alloc(c, &c->dp.M, c->dp_size))) return rc;
			if ((rc = dev_alloc(c, &c->dp.LB, c->dp_size))) return rc;
			if ((rc = dev_alloc(c, &c->dp.SZ, c->dp_size))) return rc;
			if ((rc = dev_alloc(c, &c->dp.K, c->dp_size + 64))) return rc;
			if ((rc = dev_alloc(c, &c->dp.Tb, (size_t) 32 * c->dp.tstride))) return rc;
			if ((rc = dev_alloc(c, &c->dp.Tbv, (size_t) 32 * c->dp.tstride))) return rc;
			if ((rc = dev_alloc(c, &c->d_Mprev, c->dp_size))) return rc;
		}
	}
	if (c->use_stream && !c->d_ws)
	{
		// one workspace per block of phase C (sharded: my blocks and the halo block behind them); phase A's column sweep, phase B
		// and pass 2 index the same memory by workgroup (4m words each)
		size_t const per_block = std::max<size_t>(columns_stream_ws_words(p.m, c->B), (size_t) 4 * m);
		size_t const bl = c->sh.on ? c->sh.b_lo : 0, bh = c->sh.on ? std::min<size_t>(c->nblocks, (size_t) std::max(c->sh.b_hi, c->sh.b_lo) + 1) : c->nblocks;
		c->ws_words = per_block * std::max<size_t>(bh - bl, 1);
		if ((rc = dev_alloc(c, &c->d_ws, c->ws_words))) return rc;
		c->d_ws_c = c->d_ws - bl * columns_stream_ws_words(p.m, c->B);
		// phase B spread over the chip: the digit histograms of every part of every chain of a launch (the widest launch of the
		// recursion has a chain per chain_fan blocks; a sharded rank's own range the same)
		if (!c->tune.chain_stream_passes && !c->tune.chain_stream_single && p.m < (1u << 20))
		{
			size_t const chains = std::max<size_t>(1, (bh - bl + std::max(2u, c->chain_fan) - 1) / std::max(2u, c->chain_fan) + 1);
			c->cshist_words = chains * chainmulti_parts(p.m) * CS_BINS;
			if ((rc = dev_alloc(c, &c->d_cshist, c->cshist_words))) return rc;
		}
	}
	uint64_t const k_lo = held_lo(c), k_cnt = held_hi(c) - k_lo;      // sharded: lists and stride states of my columns only
	if (X && (!c->d_ent || c->X != X))
	{
		c->X = X;
		c->stride = (X + 3) & ~1u;                // lump + up to X+1 entries, even
		dev_free(c, &c->d_ent_alloc); c->d_ent = nullptr;
		rc = dev_alloc(c, &c->d_ent_alloc, (size_t) k_cnt * c->stride + 256);   // padded: the DP loads strips unconditionally
		if (rc == FSEQ_E_OOM && c->d_ss_a)
		{
			// the stride states were sized before the lists grew: give their memory back and size them again below
			dev_free(c, &c->d_ss_a_alloc); dev_free(c, &c->d_ss_d_alloc); c->d_ss_a = c->d_ss_d = nullptr;
			rc = dev_alloc(c, &c->d_ent_alloc, (size_t) k_cnt * c->stride + 256);
		}
		if (rc) return rc;
		c->d_ent = c->d_ent_alloc - (size_t) k_lo * c->stride;   // list of column k at d_ent + k * stride
	}
	if (X && want_ss && !c->d_ss_a && p.n >= 2 * p.segment_length)
	{
		// stride states for pass 2: one (a, d) pair of m words each every snap_stride columns.  Sized after the lists:
		// what is free now, minus the boundary snapshots pass 2 will need at most (one per L columns) and a margin,
		// within [4 GiB, 160 GiB]; the smallest stride >= 16 (8: below) that fits.  (FSEQ_DEBUG prints the choice.)
		// streamed rows: 5 bytes per row when a row id and a column number fit 40 bits together (fseq_stream.hpp)
		c->ss_pack = 0;
		c->ss_ids = false;
		if (c->use_stream && !c->tune.ss_unpacked)
		{
			uint32_t abits = 1, dbits = 1;
			while ((1ull << abits) < m) ++abits;
			while ((1ull << dbits) <= p.n) ++dbits;
			if (abits + dbits <= 40 && abits < 32) c->ss_pack = abits;
			// second form of the streamed phase C on packed rows: the states in id form (its packed rows as they are: a row id and
			// a value id below 2^19 always fit 40 bits), pass 2 on the same tile step
			if (c->s2.T && c->s2.pack && c->s2.launch_snap && !c->tune.ss_absolute) { c->ss_pack = abits; c->ss_ids = true; }
		}
		if (c->ss_ids && !c->d_bs_w)
		{
			// every block's start state in id form (written by the prologue of phase C): my blocks and the halo block
			size_t const bl = c->sh.on ? c->sh.b_lo : 0, bh = c->sh.on ? std::min<size_t>(c->nblocks, (size_t) std::max(c->sh.b_hi, c->sh.b_lo) + 1) : c->nblocks;
			if ((rc = dev_alloc_range(c, &c->d_bs_w_alloc, &c->d_bs_w, bl, std::max(bh, bl + 1), m))) return rc;
			if ((rc = dev_alloc_range(c, &c->d_bs_h_alloc, &c->d_bs_h, bl, std::max(bh, bl + 1), ss_high_stride(p.m)))) return rc;
		}
		uint64_t budget = 4ull << 30;
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
			{
				// (everything else of any size is allocated by now: the margin covers the traceback / task arrays of the
				// tail, a few MB, and fragmentation -- BASELINE C4 on one GPU sits within 1 GiB of the 64-column stride)
				uint64_t const reserve = (k_cnt / p.segment_length + 1) * (uint64_t) m * 8ull + (2ull << 30);
				// (a context with a memory budget -- ranks that share a card -- plans inside what is left of it)
				uint64_t mine = free_b;
				if (c->mem_budget) mine = std::min<uint64_t>(mine, c->mem_budget > c->alloc_total ? c->mem_budget - c->alloc_total : 0);
				uint64_t const avail = mine > reserve ? mine - reserve : 0;
				budget = std::max<uint64_t>(budget, std::min<uint64_t>(avail, 160ull << 30));
			}
		}
		uint64_t const state_bytes = c->ss_pack ? (uint64_t) m * 4ull + ss_high_stride(p.m) : (uint64_t) m * 8ull;
		// first stride tried: 16 columns; 8 where a column is two digit passes (pass 2 replays stride / 2 columns per boundary at
		// twice the price there, a state costs phase C the same: BASELINE C5 pass 2 4.3 -> 2.3 ms, phase C 36.7 -> 36.9;
		// sigma <= 4: BASELINE C3 8.7 / 8.6 / 8.6 / 8.7 ms for 8 / 12 / 16 / 24)
		uint64_t st_ = (c->npass >= 2 && !c->use_stream) ? 8 : 16;
		if (c->tune.snap_stride) st_ = (uint64_t) c->tune.snap_stride;     // (experiments: first stride tried)
		// the smallest stride >= 16 whose states fit (any number, not a power of two: pass 2 costs ~stride / 2 columns per boundary)
		if ((k_cnt / st_ + 2) * state_bytes > budget) st_ = std::max<uint64_t>(st_, (k_cnt * state_bytes + budget - 1) / std::max<uint64_t>(1, budget - 2 * state_bytes));
		while ((k_cnt / st_ + 2) * state_bytes > budget) ++st_;
		c->snap_stride = (uint32_t) st_;
		if (c->tune.debug) fprintf(stderr, "[fseq] stride states every %llu columns (budget %.1f GiB, %llu bytes per state)\n", (unsigned long long) st_, budget / 1073741824.0, (unsigned long long) state_bytes);
		uint64_t const q_lo = k_lo / st_, q_hi = held_hi(c) / st_;
		if ((rc = dev_alloc(c, &c->d_ss_a_alloc, (size_t) (q_hi - q_lo + 1) * m))) return rc;
		c->d_ss_a = c->d_ss_a_alloc - (size_t) q_lo * m;         // state at column q * snap_stride at d_ss_* + q * m
		if (c->ss_pack)
		{
			size_t const hs = ss_high_stride(p.m);
			if ((rc = dev_alloc(c, &c->d_ss_d_alloc, ((size_t) (q_hi - q_lo + 1) * hs + 3) / 4))) return rc;
			c->d_ss_d = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(c->d_ss_d_alloc) - (size_t) q_lo * hs);
		}
		else
		{
			if ((rc = dev_alloc(c, &c->d_ss_d_alloc, (size_t) (q_hi - q_lo + 1) * m))) return rc;
			c->d_ss_d = c->d_ss_d_alloc - (size_t) q_lo * m;
		}
	}
	return FSEQ_OK;
}

void free_work(fseq_ctx *c)
{
	dev_free(c, &c->d_rank_alloc); dev_free(c, &c->d_keyd_alloc); dev_free(c, &c->d_nkeys_alloc);
	dev_free(c, &c->d_bstate_a_alloc); dev_free(c, &c->d_bstate_d_alloc);
	c->d_rank = c->d_keyd = c->d_nkeys = c->d_bstate_a = c->d_bstate_d = nullptr;
	dev_free(c, &c->d_hrank); dev_free(c, &c->d_hkeyd); dev_free(c, &c->d_hnkeys); dev_free(c, &c->d_hstate_a); dev_free(c, &c->d_hstate_d);
	for (auto &lv : c->levels) { dev_free(c, &lv.rank_alloc); dev_free(c, &lv.keyd_alloc); dev_free(c, &lv.nkeys_alloc); dev_free(c, &lv.state_a_alloc); dev_free(c, &lv.state_d_alloc); }
	c->levels.clear();
	dev_free(c, &c->d_ent_alloc); c->d_ent = nullptr; dev_free(c, &c->d_hdr); dev_free(c, &c->d_flags); dev_free(c, &c->d_recent);
	dev_free(c, &c->d_chunk_r0); c->chunk_cap = 0; dev_free(c, &c->d_tau); c->tau_cap = 0;
	dev_free(c, &c->d_bk); c->bk_blocks = 0; dev_free(c, &c->d_bkws); c->bkws_words = 0; dev_free(c, &c->d_todo); c->todo_cap = 0;
	dev_free(c, &c->d_colmask_alloc); c->d_colmask = nullptr; c->colmask_ready = false;
	dev_free(c, &c->d_btws); c->btws_words = 0; dev_free(c, &c->d_only); c->only_cap = 0;
	dev_free(c, &c->d_tb); c->tb_cap = 0; c->tb_win = 0;
	dev_free(c, &c->dp.M); dev_free(c, &c->dp.LB); dev_free(c, &c->dp.SZ); dev_free(c, &c->dp.K); dev_free(c, &c->dp.Tb); dev_free(c, &c->dp.Tbv);
	dev_free(c, &c->d_Mprev); dev_free(c, &c->d_spec); c->spec_cap = 0;
	dev_free(c, &c->d_cols); dev_free(c, &c->d_grp); dev_free(c, &c->d_src); dev_free(c, &c->d_ss_a_alloc); dev_free(c, &c->d_ss_d_alloc); c->d_ss_a = c->d_ss_d = nullptr;
	dev_free(c, &c->d_bs_w_alloc); dev_free(c, &c->d_bs_h_alloc); c->d_bs_w = nullptr; c->d_bs_h = nullptr; dev_free(c, &c->d_wgblk); dev_free(c, &c->d_wggrp); c->wg_cap = 0;
	dev_free(c, &c->d_gent); dev_free(c, &c->d_ghdr);
	dev_free(c, &c->d_snap_a); dev_free(c, &c->d_snap_d); dev_free(c, &c->d_ws); c->d_ws_c = nullptr; dev_free(c, &c->d_cshist); c->cshist_words = 0;
	c->cols_cap = c->gather_cap = c->snap_cap = c->grp_cap = c->src_cap = 0;
	dev_free(c, &c->d_red_cnt); dev_free(c, &c->d_red_cnt_plan); c->red_plan_valid = false; c->red_declined = false; dev_free(c, &c->d_red_vmin); dev_free(c, &c->d_red_rows_alloc); dev_free(c, &c->d_red_leaf_alloc); dev_free(c, &c->d_red_a_alloc); dev_free(c, &c->d_red_d_alloc); c->d_red_rows = c->d_red_leaf = c->d_red_a = c->d_red_d = nullptr;
	dev_free(c, &c->d_red_invalid); dev_free(c, &c->d_red_blocks); dev_free(c, &c->d_red_msa_alloc); c->d_red_msa = nullptr; c->red_cap = 0; c->red_blocks_cap = 0; c->red_ld = 0; c->red_msa_bytes = 0;
	dev_free(c, &c->d_red_ss_a_alloc); dev_free(c, &c->d_red_ss_d_alloc); c->d_red_ss_a = c->d_red_ss_d = nullptr; c->red_ss_words = 0;
	dev_free(c, &c->d_red_cls); dev_free(c, &c->d_red_headd); dev_free(c, &c->d_red_ncls); dev_free(c, &c->d_red_taskblk); dev_free(c, &c->d_red_wgtasks); c->red_task_cap = 0;
	c->red_active = false;
}

// Device-side input path (row N2): rows go up as they are (one copy per row), the alphabet scan
// (consecutive_alphabet_as_builder, generate_context.cc:135-147: dense codes in ascending byte order,
// Appendix B A2) and the row-major -> column-major transpose run on the GPU.
int shard_exchange(fseq_ctx *c, uint64_t words, int op);
void shard_post_failure(fseq_ctx *c, int code);

int upload_rows_device_impl(fseq_ctx *c, uint8_t const *const *rows);

// (sharded: the upload contains an exchange -- the alphabet of the whole alignment -- so a rank that fails here, before or
// behind it, says so in the exchange the others make next instead of leaving them in a collective)
int upload_rows_device(fseq_ctx *c, uint8_t const *const *rows)
{
	int const rc = upload_rows_device_impl(c, rows);
	shard_post_failure(c, rc);
	return rc;
}

int upload_rows_device_impl(fseq_ctx *c, uint8_t const *const *rows)
{
	fseq_params const &p = c->p;
	uint64_t const k_lo = held_lo(c), nloc = held_hi(c) - k_lo;      // sharded: this rank's columns only
	size_t const total = (size_t) p.m * nloc;
	uint8_t *d_raw = nullptr;
	uint32_t *d_present = nullptr;
	int rc;
	if ((rc = dev_alloc(c, &d_raw, total + 16))) return rc;
	if ((rc = dev_alloc(c, &d_present, 8))) { dev_free(c, &d_raw); return rc; }
	auto cleanup = [&]() { dev_free(c, &d_raw); dev_free(c, &d_present); };
	// (one copy per row from the caller's pageable memory: the runtime stages them at ~32 GB/s.  Measured and dropped in round 4:
	// eight host threads filling pinned staging buffers of their own, each with its stream -- BASELINE C3's 2.5 GB in 77 - 86 ms
	// against 78, C2's 250 MB in 39 against 30: the host copies into the pinned buffers are no faster than the runtime's own
	// staging, and the buffers cost ~10 ms to pin)
	for (uint32_t r = 0; r < p.m && nloc; ++r)
	{
		hipError_t const e = hipMemcpyAsync(d_raw + (size_t) r * nloc, rows[r] + k_lo, nloc, hipMemcpyHostToDevice, c->stream);
		if (e != hipSuccess) { cleanup(); return fail(c, FSEQ_E_HIP, "row upload", e); }
	}
	(void) hipMemsetAsync(d_present, 0, 32, c->stream);
	if (total) hipLaunchKernelGGL(k_presence, dim3(1024), dim3(256), 0, c->stream, d_raw, total, d_present);
	uint32_t present[8];
	hipError_t e = hipMemcpyAsync(present, d_present, 32, hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) { cleanup(); return fail(c, FSEQ_E_HIP, "alphabet scan", e); }
	if (c->sh.on)
	{
		// the alphabet is that of the whole alignment: one presence word per byte value, max over the ranks
		uint32_t pw[256];
		for (int b = 0; b < 256; ++b) pw[b] = (present[b >> 5] >> (b & 31)) & 1u;
		e = hipMemcpy(c->sh.xbuf, pw, sizeof(pw), hipMemcpyHostToDevice);
		if (e != hipSuccess) { cleanup(); return fail(c, FSEQ_E_HIP, "alphabet exchange", e); }
		if ((rc = shard_exchange(c, 256, 1))) { cleanup(); return rc; }
		e = hipMemcpy(pw, c->sh.xbuf, sizeof(pw), hipMemcpyDeviceToHost);
		if (e != hipSuccess) { cleanup(); return fail(c, FSEQ_E_HIP, "alphabet exchange", e); }
		memset(present, 0, sizeof(present));
		for (int b = 0; b < 256; ++b) if (pw[b]) present[b >> 5] |= 1u << (b & 31);
	}
	CodeTable tab;
	memset(&tab, 0, sizeof(tab));
	uint32_t sigma = 0;
	for (int b = 0; b < 256; ++b)
		if ((present[b >> 5] >> (b & 31)) & 1u) { tab.code_of[b] = (uint8_t) sigma; c->code_to_byte[sigma] = (uint8_t) b; ++sigma; }
	c->sigma = sigma;
	if ((rc = alloc_msa(c))) { cleanup(); return rc; }
	if (nloc)
	{
		dim3 const grid((uint32_t) ((nloc + 63) / 64), (uint32_t) ((p.m + 63) / 64));
		hipLaunchKernelGGL(k_encode_transpose, grid, dim3(256), 0, c->stream, d_raw, tab, p.m, nloc, c->d_msa_alloc, c->ld, c->bsh);
	}
	e = hipGetLastError();
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	cleanup();
	if (e != hipSuccess) return fail(c, FSEQ_E_HIP, "encode + transpose", e);
	c->have_input = true;
	c->have_result = false;
	c->kernels_ready = false;
	c->X_hint = 0;
	c->bk_given_up = -1; c->bt_given_up = -1; c->colmask_ready = false; c->shard_dp_full_sticky = false; c->red_force_full.clear(); c->red_plan_valid = false; c->red_declined = false;
	return FSEQ_OK;
}

int set_alphabet_and_upload(fseq_ctx *c, uint8_t const *base, size_t rs, size_t cs)
{
	fseq_params const &p = c->p;
	bool present[256] = {false};
	for (uint32_t r = 0; r < p.m; ++r)
	{
		uint8_t const *row = base + (size_t) r * rs;
		for (uint64_t col = 0; col < p.n; ++col) present[row[col * cs]] = true;
	}
	// consecutive_alphabet_as_builder: dense codes in ascending byte order (generate_context.cc:135-147, A2)
	uint8_t code_of[256] = {0};
	uint32_t sigma = 0;
	for (int b = 0; b < 256; ++b)
		if (present[b]) { code_of[b] = (uint8_t) sigma; c->code_to_byte[sigma] = (uint8_t) b; ++sigma; }
	c->sigma = sigma;
	int rc = alloc_msa(c);
	if (rc) return rc;
	// encode + transpose on the host in column tiles, then one copy per tile
	size_t const tile = std::max<size_t>(1, (size_t) (8u << 20) / c->ld);
	std::vector<uint8_t> buf(tile * c->ld);
	uint32_t const bsh = c->bsh, smask = (1u << bsh) - 1u, bits = 8u >> bsh;
	for (uint64_t c0 = held_lo(c); c0 < held_hi(c); c0 += tile)
	{
		uint64_t const c1 = std::min<uint64_t>(held_hi(c), c0 + tile);
		std::fill(buf.begin(), buf.end(), 0);
		for (uint32_t r = 0; r < p.m; ++r)
		{
			uint8_t const *row = base + (size_t) r * rs;
			for (uint64_t col = c0; col < c1; ++col)
				buf[(col - c0) * c->ld + (r >> bsh)] |= (uint8_t) (code_of[row[col * cs]] << ((r & smask) * bits));
		}
		HIP_TRY(c, hipMemcpy(c->d_msa + c0 * c->ld, buf.data(), (c1 - c0) * c->ld, hipMemcpyHostToDevice));
	}
	c->have_input = true;
	c->have_result = false;
	c->kernels_ready = false;
	c->X_hint = 0;
	c->bk_given_up = -1; c->bt_given_up = -1; c->colmask_ready = false; c->shard_dp_full_sticky = false; c->red_force_full.clear(); c->red_plan_valid = false; c->red_declined = false;
	return FSEQ_OK;
}

// follow_traceback (segmentation_lp_context.cc:191-224): the lb chain is followed on the device, window by window
// (k_tb_windows / k_tb_chain / k_tb_emit, fseq_kernels.hpp); the S visited entries come back in one small copy.
// Scratch: Mprev (exit pointers) and the first words of K (hop counts) -- both are free once the DP is done.
int shard_exchange(fseq_ctx *c, uint64_t words, int op);

// The same when every rank of a sharded run holds the lb's of its own entries only (run_dp_spec, windows): the chain is
// followed rank by rank -- the owner of the entry it stands at walks its part and tells the others where it left and
// how many entries it visited (two words) --, every rank emits its entries at their place in the whole list, and the
// S x 16 bytes are gathered: one small exchange per rank the chain passes through instead of the lb and size arrays.
int follow_traceback_sharded(fseq_ctx *c, hipStream_t st)
{
	Shard const &sh = c->sh;
	uint32_t const L = (uint32_t) c->p.segment_length, dp_size = (uint32_t) c->dp_size;
	size_t const cap = (size_t) (c->p.n / L + 2);
	uint32_t const nwin = (dp_size + TB_WIN - 1u) / TB_WIN;
	int rc;
	if (c->tb_cap < cap || c->tb_win < nwin)
	{
		if ((rc = dev_alloc(c, &c->d_tb, cap + nwin / 2 + 2))) return rc;
		c->tb_cap = cap; c->tb_win = nwin;
	}
	uint2 *d_head = reinterpret_cast<uint2 *>(c->d_tb + cap);
	uint32_t *d_count = reinterpret_cast<uint32_t *>(d_head + nwin);
	uint32_t *d_exit_next = c->d_Mprev, *d_exit_cnt = reinterpret_cast<uint32_t *>(c->dp.K);
	// my part: my entries, and the final cell's (the last entry of the array) on the last active rank
	bool const have = sh.rank < sh.active && c->own_hi[sh.rank] > c->own_lo[sh.rank];
	uint32_t const vlo = have ? c->own_lo[sh.rank] : 0u, vhi = have ? (sh.rank + 1u == sh.active ? dp_size : c->own_hi[sh.rank]) : 0u;
	HIP_TRY(c, hipMemsetAsync(d_count, 0, 16, st));
	if (have) hipLaunchKernelGGL(k_tb_windows, dim3(nwin), dim3(256), 0, st, c->dp.LB, dp_size, L, d_exit_next, d_exit_cnt, vlo, vhi);
	auto owner_of = [&](uint32_t t) {
		uint32_t g = sh.active - 1u;
		while (g > 0 && t < c->own_lo[g]) --g;
		return g;
	};
	uint32_t cur = dp_size - 1u, off = 0, my_cnt = 0, my_off = 0, hops = 0;
	while (true)
	{
		uint32_t const g = owner_of(cur);
		HIP_TRY(c, hipMemsetAsync(sh.xbuf, 0, 16, st));
		if (g == sh.rank) hipLaunchKernelGGL(k_tb_chain_part, dim3(1), dim3(64), 0, st, d_exit_next, d_exit_cnt, cur, off, vlo, d_head, nwin, d_count, sh.xbuf);
		if ((rc = shard_exchange(c, 4, 0))) return rc;
		uint32_t w[4];
		HIP_TRY(c, hipMemcpy(w, sh.xbuf, 16, hipMemcpyDeviceToHost));
		if (g == sh.rank) { my_cnt = w[1]; my_off = off; }
		off += w[1];
		if (w[1] == 0 || off > cap || ++hops > sh.active) return fail(c, FSEQ_E_HIP, "internal: the sharded traceback chain does not descend");
		if (w[0] == 0) break;                                       // the chain ended on rank g
		if (w[0] - 1u >= c->own_lo[g]) return fail(c, FSEQ_E_HIP, "internal: the sharded traceback chain left a rank upwards");
		cur = w[0] - 1u;
	}
	size_t const S = off;
	if (my_cnt) hipLaunchKernelGGL(k_tb_emit, dim3(nwin), dim3(256), 0, st, c->dp.LB, c->dp.M, c->dp.SZ, dp_size, L, d_head, d_count, c->d_tb, (uint32_t) cap, vlo);
	// gather: every rank's entries sit at their final offsets of its own d_tb; word 4 S: "the chain ended in lb == 0"
	if (4 * S + 2 > sh.xwords) return fail(c, FSEQ_E_ARG, "exchange buffer too small (fseq_shard_xbuf_words)");
	HIP_TRY(c, hipMemsetAsync(sh.xbuf, 0, (4 * S + 1) * 4, st));
	if (my_cnt)
	{
		// (the chain visits a rank once, so my entries are one range of the list: d_tb[my_off .. my_off + my_cnt))
		HIP_TRY(c, hipMemcpyAsync(sh.xbuf + 4 * (size_t) my_off, c->d_tb + my_off, (size_t) my_cnt * sizeof(uint4), hipMemcpyDeviceToDevice, st));
		HIP_TRY(c, hipMemcpyAsync(sh.xbuf + 4 * S, d_count + 1, 4, hipMemcpyDeviceToDevice, st));
	}
	if ((rc = shard_exchange(c, 4 * S + 1, 0))) return rc;
	std::vector<uint4> h(S);
	uint32_t ok = 0;
	HIP_TRY(c, hipMemcpyAsync(h.data(), sh.xbuf, S * sizeof(uint4), hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipMemcpyAsync(&ok, sh.xbuf + 4 * S, 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipStreamSynchronize(st));
	HIP_TRY(c, hipGetLastError());
	if (ok != 1u || S == 0) return fail(c, FSEQ_E_HIP, "internal: the traceback chain does not descend to lb == 0");
	c->tau_host.clear();
	c->tb_guess = S;
	c->traceback.resize(S);
	for (size_t j = 0; j < S; ++j)
	{
		uint4 const e = h[S - 1 - j];                        // the kernels list the last segment first
		c->traceback[j] = fseq_dp_arg{e.y, (uint64_t) e.x + L, e.z, e.w};
	}
	return FSEQ_OK;
}

int follow_traceback(fseq_ctx *c, hipStream_t st)
{
	if (c->sh.on && c->dp_window_mode) return follow_traceback_sharded(c, st);
	uint32_t const L = (uint32_t) c->p.segment_length, dp_size = (uint32_t) c->dp_size;
	size_t const cap = (size_t) (c->p.n / L + 2);           // a segment is at least L columns long
	uint32_t const nwin = (dp_size + TB_WIN - 1u) / TB_WIN;
	int rc;
	if (c->tb_cap < cap || c->tb_win < nwin)
	{
		if ((rc = dev_alloc(c, &c->d_tb, cap + nwin / 2 + 2))) return rc;   // out[cap] | head[nwin] (uint2) | count[4]
		c->tb_cap = cap; c->tb_win = nwin;
	}
	uint2 *d_head = reinterpret_cast<uint2 *>(c->d_tb + cap);
	uint32_t *d_count = reinterpret_cast<uint32_t *>(d_head + nwin);
	uint32_t *d_exit_next = c->d_Mprev, *d_exit_cnt = reinterpret_cast<uint32_t *>(c->dp.K);
	HIP_TRY(c, hipMemsetAsync(d_count, 0, 16, st));
	hipLaunchKernelGGL(k_tb_windows, dim3(nwin), dim3(256), 0, st, c->dp.LB, dp_size, L, d_exit_next, d_exit_cnt);
	hipLaunchKernelGGL(k_tb_chain, dim3(1), dim3(64), 0, st, d_exit_next, d_exit_cnt, dp_size, d_head, nwin, d_count);
	hipLaunchKernelGGL(k_tb_emit, dim3(nwin), dim3(256), 0, st, c->dp.LB, c->dp.M, c->dp.SZ, dp_size, L, d_head, d_count, c->d_tb, (uint32_t) cap);
	// the count and -- in the same round trip -- as many entries as the last run of this context had (a second copy
	// only when there are more this time)
	size_t const guess = std::min(cap, c->tb_guess ? c->tb_guess + 16 : (size_t) 4096);
	if ((rc = pin_reserve(c, guess * (sizeof(uint4) + sizeof(uint2)) + 256))) return rc;
	uint32_t *const cnt = pin_take<uint32_t>(c, 4);
	uint4 *const hp = pin_take<uint4>(c, guess);
	uint2 *const taup = pin_take<uint2>(c, guess);
	std::vector<uint4> h;
	// not sharded: the merge thresholds of the traceback boundaries (k_seg_tau_tb) ride along -- one workgroup per
	// POSSIBLE entry, those behind the count return at once
	c->tau_host.clear();
	if (!c->sh.on)
	{
		if (c->tau_cap < cap) { if ((rc = dev_alloc(c, &c->d_tau, cap))) return rc; c->tau_cap = cap; }
		hipLaunchKernelGGL(k_seg_tau_tb, dim3((uint32_t) cap), dim3(64), 0, st, reinterpret_cast<uint4 const *>(c->d_tb), d_count, L, c->stride, c->d_ent, c->d_hdr, c->d_tau);
		HIP_TRY(c, hipMemcpyAsync(taup, c->d_tau, guess * sizeof(uint2), hipMemcpyDeviceToHost, st));
	}
	HIP_TRY(c, hipMemcpyAsync(cnt, d_count, 16, hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipMemcpyAsync(hp, c->d_tb, guess * sizeof(uint4), hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipStreamSynchronize(st));
	HIP_TRY(c, hipGetLastError());
	h.assign(hp, hp + std::min<size_t>(guess, cnt[0]));
	if (!c->sh.on) c->tau_host.assign(taup, taup + std::min<size_t>(guess, cnt[0]));
	if (cnt[1] != 1u || cnt[0] == 0 || cnt[0] > cap) return fail(c, FSEQ_E_HIP, "internal: the traceback chain does not descend to lb == 0");
	size_t const S = cnt[0];
	if (S > guess)
	{
		h.resize(S);
		HIP_TRY(c, hipMemcpy(h.data() + guess, c->d_tb + guess, (S - guess) * sizeof(uint4), hipMemcpyDeviceToHost));
		if (!c->tau_host.empty())
		{
			c->tau_host.resize(S);
			HIP_TRY(c, hipMemcpy(c->tau_host.data() + guess, c->d_tau + guess, (S - guess) * sizeof(uint2), hipMemcpyDeviceToHost));
		}
	}
	if (!c->tau_host.empty()) c->tau_host.resize(S);
	c->tb_guess = S;
	c->traceback.resize(S);
	for (size_t j = 0; j < S; ++j)
	{
		uint4 const e = h[S - 1 - j];                        // the kernels list the last segment first
		c->traceback[j] = fseq_dp_arg{e.y, (uint64_t) e.x + L, e.z, e.w};
	}
	return FSEQ_OK;
}

// ---- launches: LDS-resident kernels, or their HBM-streamed counterparts for large m
// grid workgroups = the blocks starting at column col0, col0 + B, ...; rank / keyd / nkeys point at the first of them
// phase B and pass 2 work on absolute divergences (column numbers <= n): their partition steps scan keys while n fits
// the configuration's key shift (FSEQ_PLAIN_SCAN: never)
uint32_t scan_keyed(fseq_ctx const *c)
{
	if (c->use_stream || c->tune.plain_scan) return 0u;
	return (c->p.n < (1ull << c->ks.scan_shift) && !c->tune.occurrence_keys) ? 1u : c->p.n < (1ull << 25) ? 2u : 0u;       // row-count keys, occurrence keys, has-based scan
}

// streamed rows: occurrence keys while every column number fits 25 bits (FSEQ_PLAIN_SCAN: the has-based scan)
bool stream_keyed(fseq_ctx const *c) { return c->p.n < (1ull << 25) && !c->tune.plain_scan; }

// only: per-block filter (blocks whose word is zero are skipped), or nullptr
void launch_rank(fseq_ctx *c, uint32_t grid, uint32_t B, uint32_t nblocks, uint32_t *rank, uint32_t *keyd, uint32_t *nkeys, uint64_t col0 = 0, uint32_t const *only = nullptr)
{
	fseq_params const &p = c->p;
	if (!grid) return;
	if (c->use_stream)
		hipLaunchKernelGGL((stream_keyed(c) ? k_colblock_stream<MODE_RANK, true> : k_colblock_stream<MODE_RANK, false>), dim3(grid), dim3(ST), stream_lds_bytes(sym_bytes(p.m, c->bsh), c->stream_staged), c->stream, c->d_msa, c->ld, p.m, p.n, B, nblocks,
		                   c->npass, c->bsh, c->d_ws, (uint32_t) c->stream_staged, rank, keyd, nkeys, only, (uint32_t const *) nullptr, (uint64_t const *) nullptr,
		                   (uint2 const *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, (uint64_t const *) nullptr, 0u,
		                   (uint32_t const *) nullptr, (uint32_t const *) nullptr, col0, 0u);
	else
		c->ks.rank(c->stream, grid, c->ks.lds_colblock, c->d_msa, c->ld, p.m, p.n, B, nblocks, c->npass, c->bsh, rank, keyd, nkeys, col0, only);
}

// grid chains grp0 .. grp0 + grid - 1, chain g over the key blocks [g * G, min(nb_total, (g + 1) * G))
void launch_chain(fseq_ctx *c, uint32_t grid, uint32_t const *rank, uint32_t const *keyd, uint32_t const *nkeys, uint32_t nb_total, uint32_t G,
                  uint64_t cols_per_block, uint32_t const *start_a, uint32_t const *start_d, uint32_t *out_a, uint32_t *out_d,
                  uint32_t *out_rank, uint32_t *out_keyd, uint32_t *out_nkeys, uint32_t grp0 = 0)
{
	if (!grid) return;
	// streamed rows: a chain step as a radix sort by rank + range maxima (fseq_chainsort.hpp) where the workspace holds its
	// arrays for every workgroup of the launch; else (and with FSEQ_CHAIN_STREAM_PASSES) the two-bit digit passes
	if (c->use_stream && !c->tune.chain_stream_passes && c->d_cshist && (size_t) grid * chainsort_ws_words(c->p.m) <= c->ws_words &&
	    (size_t) grid * chainmulti_parts(c->p.m) * CS_BINS <= c->cshist_words)
	{
		// ... every sweep of a step a launch over (parts) x (chains): a chain of G blocks is G rounds of them
		uint32_t const m = c->p.m, nparts = chainmulti_parts(m), npass = chainmulti_passes(m);
		ChainMultiArgs A;
		A.rank = rank; A.keyd = keyd; A.nkeys = nkeys; A.m = m; A.nb_total = nb_total; A.G = G; A.cols_per_block = cols_per_block;
		A.ws = c->d_ws; A.hist = c->d_cshist; A.start_a = start_a; A.start_d = start_d; A.out_state_a = out_a; A.out_state_d = out_d;
		A.out_rank = out_rank; A.out_keyd = out_keyd; A.out_nkeys = out_nkeys; A.grp0 = grp0; A.step = 0; A.pass = 0;
		A.nchains = grid; A.xcd_map = c->tune.chain_no_xcd_map ? 0u : 1u;
		uint32_t const grid_y = A.xcd_map ? (grid + 7u) & ~7u : grid;       // (cm_wg: the workgroups of a chain on one XCD)
		dim3 const by_row((m + CM_WG - 1u) / CM_WG, grid_y), by_part((nparts + CM_WG / WAVE - 1u) / (CM_WG / WAVE), grid_y);
		hipLaunchKernelGGL(k_cm_init, by_row, dim3(CM_WG), 0, c->stream, A);
		for (uint32_t s_ = 0; s_ < G; ++s_)
		{
			A.step = s_;
			for (uint32_t ps = 0; ps < npass; ++ps)
			{
				A.pass = ps;
				hipLaunchKernelGGL(k_cm_count, by_part, dim3(CM_WG), 0, c->stream, A);
				hipLaunchKernelGGL(k_cm_offsets, dim3(grid), dim3(ST), 0, c->stream, A);
				hipLaunchKernelGGL(k_cm_scatter, by_part, dim3(CM_WG), 0, c->stream, A);
			}
			hipLaunchKernelGGL(k_cm_output, by_row, dim3(CM_WG), 0, c->stream, A);
		}
		if (out_rank) hipLaunchKernelGGL(k_cm_emit, dim3(grid), dim3(ST), stream_lds_bytes(0, false), c->stream, A, G);
	}
	else if (c->use_stream && !c->tune.chain_stream_passes && (size_t) grid * chainsort_ws_words(c->p.m) <= c->ws_words)
		hipLaunchKernelGGL(k_chain_stream_sort, dim3(grid), dim3(ST), chainsort_lds_bytes(), c->stream, rank, keyd, nkeys, c->p.m, nb_total, G,
		                   cols_per_block, c->d_ws, start_a, start_d, out_a, out_d, out_rank, out_keyd, out_nkeys, grp0);
	else if (c->use_stream)
		hipLaunchKernelGGL((stream_keyed(c) ? k_chain_stream<true> : k_chain_stream<false>), dim3(grid), dim3(ST), stream_lds_bytes(0, true), c->stream, rank, keyd, nkeys, c->p.m, nb_total, G,
		                   cols_per_block, c->d_ws, 1u, start_a, start_d, out_a, out_d, out_rank, out_keyd, out_nkeys, grp0);
	else
		c->ks.chain(c->stream, grid, c->ks.lds_chain, rank, keyd, nkeys, c->p.m, nb_total, G, cols_per_block, start_a, start_d, out_a, out_d,
		            out_rank, out_keyd, out_nkeys, grp0, scan_keyed(c));
}

// ---- sharded runs: the one exchange primitive (include/fseq.h, fseq_set_shard) -------------------------------
// all-reduce of xbuf[0 .. words) over the ranks through the caller's function; the data must already be queued
// into xbuf on c->stream.  Not sharded: nothing to do.
// Every exchange starts with a one-word maximum of the ranks' status words (the last word of the buffer): a rank that
// has failed (out of memory, a HIP error) posts its error code there ONCE, in the exchange the others make next, and
// every rank leaves with FSEQ_E_PEER instead of waiting in a collective for a rank that will never arrive.
int shard_status(fseq_ctx *c, uint32_t mine)
{
	Shard &sh = c->sh;
	uint64_t const slot = sh.xwords - 1;
	HIP_TRY(c, hipMemcpyAsync(sh.xbuf + slot, &mine, 4, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if (sh.fn(sh.user, slot, 1, 1) != 0) return fail(c, FSEQ_E_HIP, "the caller's all-reduce failed");
	uint32_t got = 0;
	HIP_TRY(c, hipMemcpy(&got, sh.xbuf + slot, 4, hipMemcpyDeviceToHost));
	if (got && !mine)
	{
		char what[96];
		snprintf(what, sizeof(what), "another rank of the sharded run failed (its error code: %u)", got);
		return fail(c, FSEQ_E_PEER, what);
	}
	return FSEQ_OK;
}

int shard_exchange(fseq_ctx *c, uint64_t words, int op)
{
	if (!c->sh.on) return FSEQ_OK;
	if (words + 1 > c->sh.xwords) return fail(c, FSEQ_E_ARG, "exchange buffer too small (fseq_shard_xbuf_words)");
	int rc = shard_status(c, 0);
	if (rc) return rc;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if (c->sh.fn(c->sh.user, 0, words, op) != 0) return fail(c, FSEQ_E_HIP, "the caller's all-reduce failed");
	return FSEQ_OK;
}

// a rank that failed on its own tells the others (best effort: its device may be what failed)
void shard_post_failure(fseq_ctx *c, int code)
{
	if (!c->sh.on || code == FSEQ_OK || code == FSEQ_E_NO_REDUCTION || code == FSEQ_E_PEER) return;
	// once per context (the status exchange is a collective: a second post would have no partner), and not behind the
	// last exchange of a run (the other ranks have left)
	if (c->sh.posted || c->sh.closed) return;
	c->sh.posted = true;
	std::string const keep = c->err;
	(void) shard_status(c, (uint32_t) code);
	c->err = keep;
}

// "every rank contributes its own slice": zero the buffer, copy my words [lo, hi) of src in, all-reduce (sum), copy
// everything back over dst -- an all-gather of unequal slices through the one primitive
int shard_gather_u32(fseq_ctx *c, uint32_t *d_array, uint64_t total, uint64_t lo, uint64_t hi, uint64_t extra = ~0ull)
{
	if (!c->sh.on) return FSEQ_OK;
	hipStream_t st = c->stream;
	HIP_TRY(c, hipMemsetAsync(c->sh.xbuf, 0, total * 4, st));
	if (hi > lo) HIP_TRY(c, hipMemcpyAsync(c->sh.xbuf + lo, d_array + lo, (hi - lo) * 4, hipMemcpyDeviceToDevice, st));
	if (extra != ~0ull) HIP_TRY(c, hipMemcpyAsync(c->sh.xbuf + extra, d_array + extra, 4, hipMemcpyDeviceToDevice, st));
	int rc = shard_exchange(c, total, 0);
	if (rc) return rc;
	HIP_TRY(c, hipMemcpyAsync(d_array, c->sh.xbuf, total * 4, hipMemcpyDeviceToDevice, st));
	return FSEQ_OK;
}

// The chunk plan of the speculative DP (fseq_dpspec.hpp): chunk k runs the rounds [r0[k], r0[k + 1]) (the last one
// also the drain round and the final cell); no chunks = use the serial kernel.  Sharded: a round belongs to the
// rank that owns its first column; every rank cuts its own rounds into chunks and every rank computes the same table.
struct SpecPlan {
	std::vector<uint32_t> r0;                // nchunks + 1 entries
	uint32_t mine_lo = 0, mine_hi = 0;       // my chunks
	std::vector<uint32_t> rank_lo;           // sharded: rank g runs the chunks [rank_lo[g], rank_lo[g + 1]) (active + 1 entries)
	uint32_t nchunks() const { return r0.empty() ? 0u : (uint32_t) r0.size() - 1u; }
};

SpecPlan spec_plan(fseq_ctx *c, DpSchedule const &S)
{
	SpecPlan P;
#if defined(FSEQ_DP_STAMPS) || defined(FSEQ_DP_STATS)
	if (!c->sh.on) return P;                 // the diagnostic builds instrument the serial kernel
#endif
	if (c->tune.dp_serial && !c->sh.on) return P;
	int ncu = 0;
	(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->p.device);
	if (ncu < 1) ncu = 1;
	// one chunk per CU, but chunks of at least max(400, 8L) entries (the sweeps of shorter chunks are cheaper but more of
	// them are needed; BASELINE C2, L = 50: 96 chunks of 21 rounds 0.32 ms in 3 sweeps, 250 of 8 rounds 0.20 ms in 4,
	// 334 of 6 rounds 0.27 ms)
	uint32_t const min_entries = std::max<uint32_t>(400u, 8u * S.L);
	uint32_t forced = 0;
	if (c->tune.dp_spec_rounds) forced = (uint32_t) c->tune.dp_spec_rounds;   // tests: any chunk length
	auto cut = [&](uint32_t lo, uint32_t hi) {
		// rounds [lo, hi) of one rank into chunks
		if (hi <= lo) return;
		uint32_t rpc = std::max<uint32_t>((hi - lo + (uint32_t) ncu - 1u) / (uint32_t) ncu, (min_entries + S.RL - 1u) / S.RL);
		if (forced) rpc = forced;
		for (uint32_t r = lo; r < hi; r += rpc) P.r0.push_back(r);
	};
	if (!c->sh.on)
	{
		cut(0, S.nreg);
		P.r0.push_back(S.nreg);
		uint32_t const nch = P.nchunks();
		if ((nch < 3u && !forced) || nch < 2u || nch > 65535u) { P.r0.clear(); return P; }
		P.mine_lo = 0; P.mine_hi = nch;
		return P;
	}
	Shard const &sh = c->sh;
	uint32_t prev = 0;
	for (uint32_t g = 0; g < sh.active; ++g)
	{
		// rounds whose first column (L + r RL - 1) lies in rank g's columns; the last active rank takes the rest
		uint64_t const hi_col = std::min<uint64_t>(c->p.n, (uint64_t) (g + 1) * sh.bpr * c->B);
		uint32_t r_hi = S.nreg;
		if (g + 1 < sh.active)
		{
			uint64_t const need = hi_col + 1 > S.L ? hi_col + 1 - S.L : 0;      // first round with L + r RL - 1 >= hi_col
			r_hi = (uint32_t) std::min<uint64_t>(S.nreg, (need + S.RL - 1) / S.RL);
		}
		if (r_hi < prev) r_hi = prev;
		if (g == sh.rank) P.mine_lo = (uint32_t) P.r0.size();
		P.rank_lo.push_back((uint32_t) P.r0.size());
		cut(prev, r_hi);
		if (g == sh.rank) P.mine_hi = (uint32_t) P.r0.size();
		prev = r_hi;
	}
	P.rank_lo.push_back((uint32_t) P.r0.size());
	P.r0.push_back(S.nreg);
	if (sh.rank >= sh.active) P.mine_lo = P.mine_hi = P.nchunks();
	return P;
}

// The arrays a run of the speculative DP starts from (nothing here depends on phases A-C: run_long_path queues it on
// the second stream while phase C runs)
int dp_spec_reset(fseq_ctx *c, SpecPlan const &P, hipStream_t s)
{
	uint32_t const nch = P.nchunks();
	int rc;
	if (c->spec_cap < nch)
	{
		if ((rc = dev_alloc(c, &c->d_spec, (size_t) 7 * nch + 16))) return rc;
		c->spec_cap = nch;
	}
	if (c->chunk_cap < nch + 1u)
	{
		if ((rc = dev_alloc(c, &c->d_chunk_r0, nch + 1u))) return rc;
		c->chunk_cap = nch + 1u;
	}
	HIP_TRY(c, hipMemcpyAsync(c->d_chunk_r0, P.r0.data(), (size_t) (nch + 1u) * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(c, hipMemsetAsync(c->dp.M, 0, c->dp_size * 4, s));
	HIP_TRY(c, hipMemsetAsync(c->d_Mprev, 0, c->dp_size * 4, s));
	HIP_TRY(c, hipMemsetAsync(c->d_spec, 0, ((size_t) 7 * nch + 16) * 4, s));
	HIP_TRY(c, hipMemsetAsync(c->d_spec, 0x01, (size_t) nch * 4, s));         // d_active != 0: every chunk runs in sweep 1
	if (c->sh.on)
	{
		// LB / SZ are gathered over the ranks at the end: what nobody writes must be 0 everywhere
		HIP_TRY(c, hipMemsetAsync(c->dp.LB, 0, c->dp_size * 4, s));
		HIP_TRY(c, hipMemsetAsync(c->dp.SZ, 0, c->dp_size * 4, s));
	}
	return FSEQ_OK;
}

uint32_t Wx_for_debug(fseq_ctx const *c, uint32_t L) { return c->tune.shard_dp_window ? (uint32_t) c->tune.shard_dp_window : std::max<uint32_t>(2u * DPW, 16u * L); }

// Phase D as chunk-speculative sweeps on the whole chip (fseq_dpspec.hpp).  Leaves M / LB / SZ exactly as
// k_dp<DP_WHOLE> would (on every rank of a sharded run); *overflow = some cell's list was too short.
// reset_done: dp_spec_reset has been queued (on any stream `st` already waits for).
int run_dp_spec(fseq_ctx *c, DpSchedule const &S, SpecPlan const &P, hipStream_t st, uint32_t *overflow, uint32_t *sweeps_out, bool reset_done = false)
{
	fseq_params const &p = c->p;
	uint32_t const m = p.m, n = (uint32_t) p.n, L = (uint32_t) p.segment_length;
	uint32_t const nch = P.nchunks();
	bool const sharded = c->sh.on;
	int rc;
	if (!reset_done && (rc = dp_spec_reset(c, P, st))) return rc;
	uint32_t *d_active = c->d_spec, *d_changed = d_active + nch, *d_tailmin = d_changed + nch, *d_floor = d_tailmin + nch,
	         *d_lift = d_floor + nch, *d_ovf = d_lift + nch;
	SpecCtl *d_ctl = reinterpret_cast<SpecCtl *>(d_ovf + 2 * (size_t) nch);      // (ovf: {list too short, lowest entry read} per chunk)
	SpecGeom G;
	G.chunk_r0 = c->d_chunk_r0;
	G.RL = S.RL;
	G.nchunks = nch;
	G.NR = n - 2u * L + 1u;
	G.t_final = n - L;
	G.win = std::max<uint32_t>(256u, 4u * L);
	if (c->tune.dp_spec_win) G.win = (uint32_t) c->tune.dp_spec_win;
	uint32_t const ncomplete = G.NR / 64u;
	uint32_t const grid_c = (uint32_t) ((c->dp_size + 255) / 256);          // 4 blocks of 64 entries per workgroup, incl. the final cell's

	DpSpecArgs SP;
	SP.chunk_r0 = c->d_chunk_r0; SP.nchunks = nch; SP.chunk0 = P.mine_lo; SP.active = d_active; SP.ovf = d_ovf;
	SP.ctl = reinterpret_cast<uint32_t const *>(d_ctl);
	uint32_t const mine = P.mine_hi - P.mine_lo;
	// my entries: the chunks [mine_lo, mine_hi) are consecutive rounds
	uint64_t const t_lo = mine ? (uint64_t) P.r0[P.mine_lo] * S.RL : 0, t_hi = mine ? (P.mine_hi == nch ? G.NR : (uint64_t) P.r0[P.mine_hi] * S.RL) : 0;
	uint64_t const t_extra = (mine && P.mine_hi == nch) ? G.t_final : ~0ull;
	auto sweep = [&](bool fresh) {
		SP.fresh = fresh ? 1u : 0u;
		if (mine)
			hipLaunchKernelGGL(k_dp<DP_SPEC>, dim3(mine), dim3(1024), dp_lds_bytes(), st, c->dp, c->d_ent, c->d_hdr, c->stride, m, n, L,
			                   c->d_flags, 0u, 0u, SP);
	};
	auto compare = [&](bool first) {
		hipLaunchKernelGGL(k_spec_scan, dim3(nch), dim3(256), 0, st, c->dp.M, c->d_Mprev, G, d_active, d_changed, d_tailmin, d_ctl);
		hipLaunchKernelGGL(k_spec_decide, dim3(1), dim3(64), 0, st, nch, first ? 1u : 0u, d_changed, d_tailmin, d_floor, d_lift, d_active, d_ovf, d_ctl);
	};
	auto rebuild = [&]() {
		hipLaunchKernelGGL(k_spec_rebuild, dim3(grid_c), dim3(256), 0, st, c->dp, c->d_Mprev, G, d_lift, d_ctl);
		hipLaunchKernelGGL(k_spec_table, dim3((ncomplete + 255u) / 256u), dim3(256), 0, st, c->dp, ncomplete, d_ctl);
	};
	uint32_t max_sweeps = 12;
	if (c->tune.dp_spec_max_sweeps) max_sweeps = (uint32_t) c->tune.dp_spec_max_sweeps;
	SpecCtl h{};
	// "list too short" of my chunks lo .. hi - 1 (ovf words are {flag, lowest entry read} pairs)
	auto own_overflow = [&](std::vector<uint32_t> const &ovf2, uint32_t lo, uint32_t hi) {
		uint32_t o = 0;
		for (uint32_t k = lo; k < hi; ++k) o |= ovf2[2 * (size_t) k] ? 1u : 0u;
		return o;
	};
	// ---- sharded: who owns which entries, and whether a rank keeps windows or whole arrays
	c->dp_window_mode = false;
	c->dp_exchange_words = 0;
	std::vector<uint32_t> win_lo, win_off;                    // window in front of rank g: entries [win_lo[g], own_lo[g]) at xbuf + win_off[g]
	uint64_t win_total = 0;
	if (sharded)
	{
		Shard const &sh = c->sh;
		c->own_lo.assign(sh.world, 0); c->own_hi.assign(sh.world, 0);
		for (uint32_t g = 0; g < sh.active; ++g)
		{
			uint32_t const c_lo = P.rank_lo[g], c_hi = P.rank_lo[g + 1];
			c->own_lo[g] = P.r0[c_lo] * S.RL;
			c->own_hi[g] = c_hi == nch ? G.NR : P.r0[c_hi] * S.RL;
			if (c_hi <= c_lo) c->own_hi[g] = c->own_lo[g];
		}
		// A chunk reads entries in front of it through its LDS ring (the DPW entries in front of its first cell) and, rarely,
		// straight from memory: both stay within a few thousand entries on every input measured (the candidates of a cell end
		// where the cumulative count of its list passes the cell's value).  So a rank keeps, of the other ranks' keys, a WINDOW
		// in front of its own entries, the sweeps report the lowest entry they read (k_dp: ovf words), and a sweep that looked
		// below the window makes the run start again with whole-array exchanges (exactness never rests on the window).
		uint32_t Wx = std::max<uint32_t>(2u * DPW, 16u * L);
		if (c->tune.shard_dp_window) Wx = (uint32_t) c->tune.shard_dp_window;
		win_lo.assign(sh.world, 0); win_off.assign(sh.world, 0);
		uint64_t off = 2ull * nch + 2;                          // [changed nch][tailmin nch][below][pad]
		for (uint32_t g = 1; g < sh.active; ++g)
		{
			uint32_t const th = c->own_lo[g];
			win_lo[g] = th > Wx ? ((th - Wx) & ~63u) : 0u;       // (whole 64-blocks: the block minima of the window are then right too)
			win_off[g] = (uint32_t) off;
			off += th - win_lo[g];
		}
		win_total = off;
		c->dp_window_mode = !c->tune.shard_dp_full && !c->shard_dp_full_sticky && 2 * win_total < c->dp_size && win_total + 1 <= c->sh.xwords;
	}
	sweep(true);
	uint32_t done_sweeps = 1;
	std::vector<uint32_t> ovf_early;
	if (!sharded)
	{
		// every kernel returns at once when the iteration has converged, so sweeps are queued ahead of the
		// host's look at the control word: three further sweeps first (the measured common case needs three in
		// all), then one at a time
		uint32_t batch = 3;
		while (true)
		{
			for (uint32_t i = 0; i < batch && done_sweeps < max_sweeps; ++i)
			{
				compare(done_sweeps == 1);
				rebuild();
				sweep(false);
				++done_sweeps;
			}
			compare(done_sweeps == 1);
			if ((rc = pin_reserve(c, sizeof(h) + (size_t) nch * 8 + 64))) return rc;
			auto *const hpin = pin_take<std::remove_reference_t<decltype(h)>>(c, 1);
			uint32_t *const opin = pin_take<uint32_t>(c, 2 * (size_t) nch);
			HIP_TRY(c, hipMemcpyAsync(hpin, d_ctl, sizeof(h), hipMemcpyDeviceToHost, st));
			// (the chunks' "list too short" words in the same round trip: final if the iteration has converged)
			HIP_TRY(c, hipMemcpyAsync(opin, d_ovf, (size_t) nch * 8, hipMemcpyDeviceToHost, st));
			HIP_TRY(c, hipStreamSynchronize(st));
			HIP_TRY(c, hipGetLastError());
			h = *hpin;
			ovf_early.assign(opin, opin + 2 * (size_t) nch);
			if (h.done || done_sweeps >= max_sweeps) break;
			// the compare just queued has already chosen the next sweep's active set and lifts
			rebuild();
			sweep(false);
			++done_sweeps;
			batch = 1;
		}
	}
	else if (!c->dp_window_mode)
	{
		// sharded, whole arrays: after every sweep the ranks exchange the keys of their chunks; compare / lift / rebuild then run
		// on the whole arrays on every rank (same inputs, same results), the next sweep again on the rank's own chunks
		while (true)
		{
			if ((rc = shard_gather_u32(c, c->dp.M, c->dp_size, t_lo, t_hi, t_extra))) return rc;
			c->dp_exchange_words += c->dp_size;
			compare(done_sweeps == 1);
			HIP_TRY(c, hipMemcpyAsync(&h, d_ctl, sizeof(h), hipMemcpyDeviceToHost, st));
			HIP_TRY(c, hipStreamSynchronize(st));
			HIP_TRY(c, hipGetLastError());
			if (h.done) break;                   // no serial fallback here: after sweep k the chunks 0..k-1 are exact, so this ends
			if (done_sweeps > nch + 2u) return fail(c, FSEQ_E_HIP, "internal: speculative DP did not converge");
			rebuild();
			sweep(false);
			++done_sweeps;
		}
	}
	else
	{
		// sharded, windows: after every sweep ONE exchange carries what the others need of a rank -- "changed" and the tail
		// minimum of each of its chunks (the lifts follow from those on every rank alike), whether one of its chunks read
		// below its window, and the keys in the window in front of every other rank's entries (a rank contributes the part
		// of each window it owns).  compare (of its own chunks) / decide / lift / rebuild run on every rank; what a rank holds
		// outside its entries and its window is never read.
		Shard const &sh = c->sh;
		uint32_t *const xb = sh.xbuf;
		uint32_t const my_valid_lo = sh.rank < sh.active ? win_lo[sh.rank] : 0u;
		while (true)
		{
			HIP_TRY(c, hipMemsetAsync(xb, 0, (size_t) win_total * 4, st));
			if (mine)
				hipLaunchKernelGGL(k_spec_scan, dim3(mine), dim3(256), 0, st, c->dp.M, c->d_Mprev, G, d_active, xb, xb + nch, d_ctl,
				                   P.mine_lo, (uint32_t const *) d_ovf, my_valid_lo, xb + 2 * (size_t) nch);
			for (uint32_t g = 1; mine && g < sh.active; ++g)
			{
				uint64_t const a = std::max<uint64_t>(win_lo[g], t_lo), b = std::min<uint64_t>(c->own_lo[g], t_hi);
				if (b > a) HIP_TRY(c, hipMemcpyAsync(xb + win_off[g] + (a - win_lo[g]), c->dp.M + a, (b - a) * 4, hipMemcpyDeviceToDevice, st));
			}
			if ((rc = shard_exchange(c, win_total, 0))) return rc;
			c->dp_exchange_words += win_total;
			HIP_TRY(c, hipMemcpyAsync(d_changed, xb, (size_t) 2 * nch * 4, hipMemcpyDeviceToDevice, st));      // changed | tailmin are adjacent
			if (sh.rank >= 1 && sh.rank < sh.active && c->own_lo[sh.rank] > win_lo[sh.rank])
				HIP_TRY(c, hipMemcpyAsync(c->dp.M + win_lo[sh.rank], xb + win_off[sh.rank], (size_t) (c->own_lo[sh.rank] - win_lo[sh.rank]) * 4, hipMemcpyDeviceToDevice, st));
			hipLaunchKernelGGL(k_spec_decide, dim3(1), dim3(64), 0, st, nch, done_sweeps == 1 ? 1u : 0u, d_changed, d_tailmin, d_floor, d_lift, d_active, d_ovf, d_ctl);
			uint32_t below = 0;
			HIP_TRY(c, hipMemcpyAsync(&h, d_ctl, sizeof(h), hipMemcpyDeviceToHost, st));
			HIP_TRY(c, hipMemcpyAsync(&below, xb + 2 * (size_t) nch, 4, hipMemcpyDeviceToHost, st));
			HIP_TRY(c, hipStreamSynchronize(st));
			HIP_TRY(c, hipGetLastError());
			if (below)
			{
				// some rank's sweep read a key it does not hold: nothing of this run is trusted; whole arrays from here on
				// (every rank sees the same word, so every rank takes this way)
				if (c->tune.debug) fprintf(stderr, "[fseq] sharded DP: a sweep read below its window (%u entries): again with whole-array exchanges\n", Wx_for_debug(c, L));
				c->shard_dp_full_sticky = true;
				return run_dp_spec(c, S, P, st, overflow, sweeps_out, false);
			}
			if (h.done) break;
			if (done_sweeps > nch + 2u) return fail(c, FSEQ_E_HIP, "internal: speculative DP did not converge");
			rebuild();
			sweep(false);
			++done_sweeps;
		}
	}
	if (!h.done)
	{
		// bounded (one GPU only): finish serially behind the last chunk known to be exact (its masks and samples are rebuilt first)
		rebuild();
		uint32_t const first_dirty = std::min(h.first_changed + 1u, nch);
		uint32_t const r0 = first_dirty < nch ? P.r0[first_dirty] : S.nreg;
		HIP_TRY(c, hipMemsetAsync(c->d_flags, 0, 16, st));
		hipLaunchKernelGGL(k_dp<DP_PARTIAL>, dim3(1), dim3(1024), dp_lds_bytes(), st, c->dp, c->d_ent, c->d_hdr, c->stride, m, n, L,
		                   c->d_flags, r0, S.nrounds, DpSpecArgs{});
		// overflow: the serial part reports through d_flags, the frozen chunks through their own words
		std::vector<uint32_t> ovf(2 * (size_t) nch);
		uint32_t fl[4] = {0, 0, 0, 0};
		HIP_TRY(c, hipMemcpyAsync(ovf.data(), d_ovf, (size_t) nch * 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(c, hipMemcpyAsync(fl, c->d_flags, 16, hipMemcpyDeviceToHost, st));
		HIP_TRY(c, hipStreamSynchronize(st));
		HIP_TRY(c, hipGetLastError());
		*overflow = (fl[0] & 1u) | own_overflow(ovf, P.mine_lo, std::min(P.mine_hi, h.first_changed + 1u));
		if (c->tune.debug) fprintf(stderr, "[fseq] speculative DP: not converged after %u sweeps, serial from round %u\n", done_sweeps, r0);
	}
	else
	{
		// the chunks' "list too short" words are written by their owners only
		std::vector<uint32_t> ovf(2 * (size_t) nch);
		if (ovf_early.size() == 2 * (size_t) nch) ovf = ovf_early;           // (read together with the control word that said "done")
		else
		{
			HIP_TRY(c, hipMemcpyAsync(ovf.data(), d_ovf, (size_t) nch * 8, hipMemcpyDeviceToHost, st));
			HIP_TRY(c, hipStreamSynchronize(st));
		}
		*overflow = own_overflow(ovf, P.mine_lo, P.mine_hi);
	}
	if (sharded)
	{
		if (!c->dp_window_mode)
		{
			// lb and size of every entry from the rank that computed it (frozen chunks: from the sweep that last ran them)
			if ((rc = shard_gather_u32(c, c->dp.LB, c->dp_size, t_lo, t_hi, t_extra))) return rc;
			if ((rc = shard_gather_u32(c, c->dp.SZ, c->dp_size, t_lo, t_hi, t_extra))) return rc;
			c->dp_exchange_words += 2 * c->dp_size;
		}
		uint32_t o = *overflow;
		HIP_TRY(c, hipMemcpyAsync(c->sh.xbuf, &o, 4, hipMemcpyHostToDevice, st));
		if ((rc = shard_exchange(c, 1, 1))) return rc;
		HIP_TRY(c, hipMemcpy(&o, c->sh.xbuf, 4, hipMemcpyDeviceToHost));
		*overflow = o;
	}
	if (sweeps_out) *sweeps_out = h.done ? h.sweeps : done_sweeps + 1000u;
	if (c->tune.debug)
		fprintf(stderr, "[fseq] speculative DP: %u chunks (mine %u..%u), %u sweeps compared, done=%u%s, %.2f MB exchanged by the sweeps\n", nch, P.mine_lo, P.mine_hi, h.sweeps, h.done,
		        sharded ? (c->dp_window_mode ? ", windows" : ", whole arrays") : "", c->dp_exchange_words * 4 / 1e6);
	return FSEQ_OK;
}

// what the phases of one long-path run share (run_long_path)
struct LongRun {
	uint32_t X = 0, retries = 0;
	double ms_c = 0, ms_dp = 0, ms_host = 0, ms_p2 = 0;
	uint64_t pass2_cells = 0;
	bool keyspace = false;
	bool tree_ran = false;                   // phase A ran the key-space tree at all (else: the column sweep did every block, as last time)
	bool tree_alone = false;                 // phase A ran the key-space tree without the column sweep behind it (no block was given up last time)
	bool trie_ran = false, trie_alone = false;   // ... the trie over 16-column words (streamed rows); ... without the key-space tree behind it
	bool redo = false;                       // [r5] lists of some blocks could not be proven on their representatives: the attempt runs again, those blocks on all rows
	uint32_t redone = 0;
	bool range_ab_open = false;              // the roctx range of phases A + B spans two functions
};

// the aliases every phase uses
#define FSEQ_LONG_LOCALS(c)                                                                           \
	fseq_params const &p = (c)->p;                                                                    \
	uint32_t const m = p.m;                                                                           \
	uint64_t const n = p.n;                                                                           \
	uint64_t const L = p.segment_length;                                                              \
	hipStream_t st = (c)->stream;                                                                     \
	KernelSet const &ks = (c)->ks;                                                                    \
	Shard const &sh = (c)->sh;                                                                        \
	bool const sharded = sh.on;                                                                       \
	uint32_t const b_lo = sharded ? sh.b_lo : 0u, b_hi = sharded ? sh.b_hi : (c)->nblocks;            \
	uint32_t const my_blocks = b_hi - b_lo;                                                           \
	int rc = FSEQ_OK;                                                                                 \
	(void) m; (void) n; (void) L; (void) st; (void) ks; (void) sharded; (void) b_lo; (void) my_blocks; (void) rc

// diagnostic ("ABC" in FSEQ_SYNC_PHASES): synchronise behind a phase, so that a fault shows up at the phase that caused it
bool sync_at(fseq_ctx const *c, char ph) { return c->tune.sync_phases.find(ph) != std::string::npos; }

// ---- phase A: the key blocks of my column blocks (independent of the list capacity)
int long_phase_a(fseq_ctx *c, LongRun &R)
{
	FSEQ_LONG_LOCALS(c);
	// 4-bit symbols, LDS-resident rows: the codes present in every column I hold, once per input (k_columns takes a column with at
	// most four of them in one digit pass)
	if (c->bsh == 1u && c->npass == 2u && !c->use_stream && !c->tune.no_dense_columns && !c->colmask_ready && held_hi(c) > held_lo(c)
	    && c->d_msa_alloc && (c->ld & 3u) == 0)                  // (own columns: padded past their last byte, whole words can be read)
	{
		uint64_t const lo = held_lo(c), hi = held_hi(c);
		if ((rc = dev_alloc_range(c, &c->d_colmask_alloc, &c->d_colmask, (size_t) lo, (size_t) hi, 1))) return rc;
		HIP_TRY(c, hipMemsetAsync(c->d_flags + 67, 0, 4, st));
		hipLaunchKernelGGL(k_column_presence, dim3((uint32_t) std::min<uint64_t>(hi - lo, 8192)), dim3(256), 0, st, c->d_msa, c->ld, sym_bytes(m, c->bsh), lo, hi, c->d_colmask,
		                   c->d_flags + 67);
		// (once per input: one column in twenty with at most four codes, and phase C is the kernel with the one-pass branch)
		uint32_t n_dense = 0;
		HIP_TRY(c, hipMemcpyAsync(&n_dense, c->d_flags + 67, 4, hipMemcpyDeviceToHost, st));
		HIP_TRY(c, hipStreamSynchronize(st));
		c->colmask_use = (uint64_t) n_dense * 20u >= hi - lo;
		c->colmask_ready = true;
	}
	HIP_TRY(c, hipEventRecord(c->ev[0], st));
	progress(c, FSEQ_STAGE_TRACEBACK, 0, n);
	FSEQ_RANGE_PUSH("fseq pass 1: phases A + B (block keys, boundary states)");
	R.range_ab_open = true;                  // (popped in long_phase_b; run_long_path pops it when a phase fails in between)
	bool const keyspace = R.keyspace = c->bk_cap_words && my_blocks && !c->tune.phase_a_classic;
	// The key-space tree hands the blocks whose merges would slice past their budget to the column sweep (fseq_blockkeys.hpp,
	// BK_ABORT): per-block flags, the sweep launched over my blocks with the flags as its filter.  What the last run on this
	// input saw decides what is launched now (the input has not changed, so neither has the outcome): no block given up ->
	// the tree alone; most of them -> the sweep alone; else both.  FSEQ_BLOCKKEYS_CAP (tests of the slices) and
	// FSEQ_BLOCKKEYS_NO_LIMIT: the tree slices as often as it takes.
	bool const limited = keyspace && !c->tune.blockkeys_no_limit && !c->tune.blockkeys_cap;
	bool const tree = keyspace && !(limited && c->bk_given_up >= 0 && 2u * (uint32_t) c->bk_given_up > my_blocks);
	bool const sweep_after = limited && !(tree && c->bk_given_up == 0);
	R.tree_alone = tree && limited && !sweep_after;
	R.tree_ran = tree;
	uint32_t *todo = nullptr;
	if (limited)
	{
		if (c->todo_cap < my_blocks) { if ((rc = dev_alloc(c, &c->d_todo, my_blocks))) return rc; c->todo_cap = my_blocks; }
		todo = c->d_todo;
		HIP_TRY(c, hipMemsetAsync(todo, tree ? 0 : 0x01, (size_t) my_blocks * 4, st));     // (no tree: every block is the sweep's)
	}
	if (keyspace) HIP_TRY(c, hipMemsetAsync(c->d_flags + 64, 0, 12, st));
	// The trie over 32-bit group words first (fseq_blocktrie.hpp) -- it reads the block once and ranks only its distinct keys --
	// and the key-space tree for the blocks it gives up (too many distinct keys for its tables).  As with the tree and the
	// sweep, what the last run on this input saw decides what is launched: nothing given up -> the trie alone; most blocks ->
	// no trie.  The tests of the tree's slices (FSEQ_BLOCKKEYS_CAP, _NO_LIMIT) keep the tree.
	uint32_t const bt_bits = 8u >> c->bsh, bt_T = blocktrie_threads(m, c->use_stream);
	// (LDS-resident rows: from 6,145 rows on -- BASELINE C5's 10,000: phase A 7.3 -> 5.9 ms; on C3's 2,504 rows a level of the trie is
	// a dozen barriers for 157 busy threads and the tree is as fast, 1.31 against 1.36 ms; FSEQ_BLOCKTRIE_ALWAYS: tests)
	bool const trie = tree && limited && (uint64_t) m <= (uint64_t) (32u / bt_bits) * bt_T * 32u && c->B < 65536u && !c->tune.no_blocktrie
	                  && (c->use_stream || m > 12u * 512u || c->tune.blocktrie_always)
	                  && (c->ld & 3u) == 0 && (reinterpret_cast<uintptr_t>(c->d_msa) & 3u) == 0
	                  && !(c->bt_given_up >= 0 && 2u * (uint32_t) c->bt_given_up > my_blocks);
	bool const tree_after = tree && !(trie && c->bt_given_up == 0);
	R.trie_ran = trie;
	R.trie_alone = trie && !tree_after;
	uint32_t const *only = nullptr;
	if (trie)
	{
		int ncu = 0;
		(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->p.device);
		uint32_t const per_cu = (uint32_t) std::max<size_t>(1, std::min<size_t>(2048u / bt_T, (160u * 1024u) / blocktrie_lds(bt_T)));
		uint32_t const groups = std::min<uint32_t>(my_blocks, (uint32_t) std::max(1, ncu) * per_cu);
		size_t const per = (blocktrie_ws_words(m, c->B, bt_bits, bt_T) + 15) & ~size_t(15);
		if (c->btws_words < per * groups)
		{
			if ((rc = dev_alloc(c, &c->d_btws, per * groups))) return rc;
			c->btws_words = per * groups;
		}
		if (c->only_cap < my_blocks) { if ((rc = dev_alloc(c, &c->d_only, my_blocks))) return rc; c->only_cap = my_blocks; }
		HIP_TRY(c, hipMemsetAsync(c->d_only, 0, (size_t) my_blocks * 4, st));
		HIP_TRY(c, launch_blocktrie(bt_bits, bt_T, st, groups, c->d_msa, c->ld, m, n, c->B, my_blocks,
		                            c->d_rank + (size_t) b_lo * m, c->d_keyd + (size_t) b_lo * m, c->d_nkeys + b_lo, (uint64_t) b_lo * c->B,
		                            c->d_btws, per, c->d_flags + 66, c->d_only));
		only = c->d_only;
	}
	if (tree_after && c->use_stream)
	{
		// phase A in key space, streamed rows: one workgroup per CU with its own workspace, blocks round-robin
		int ncu = 0;
		(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->p.device);
		uint32_t const groups = std::min<uint32_t>(my_blocks, (uint32_t) std::max(1, ncu));
		size_t const per = (blockkeys_stream_ws_words(m, c->B, c->bsh) + 15) & ~size_t(15);
		if (c->bkws_words < per * groups)
		{
			if ((rc = dev_alloc(c, &c->d_bkws, per * groups))) return rc;
			c->bkws_words = per * groups;
		}
		hipLaunchKernelGGL(k_blockkeys_stream, dim3(groups), dim3(1024), c->bk_lds, st, c->d_msa, c->ld, m, n, c->B, c->bsh, my_blocks,
		                   c->d_rank + (size_t) b_lo * m, c->d_keyd + (size_t) b_lo * m, c->d_nkeys + b_lo, (uint64_t) b_lo * c->B,
		                   c->d_bkws, per, c->bk_cap_words, c->d_flags + 64, (c->tune.blockkeys_wide ? 1u : 0u) | (c->tune.blockkeys_single ? 2u : 0u), todo, only);
	}
	else if (tree_after)
	{
		// phase A in key space (fseq_blockkeys.hpp)
		size_t const per = (blockkeys_scratch_halfwords(m, c->B, c->bsh) + 7) & ~size_t(7);
		if (c->bk_per_block != per || c->bk_blocks < my_blocks)
		{
			if ((rc = dev_alloc(c, &c->d_bk, per * my_blocks))) return rc;
			c->bk_per_block = per; c->bk_blocks = my_blocks;
		}
		launch_blockkeys(c->bk_T, st, my_blocks, c->bk_lds, c->d_msa, c->ld, m, n, c->B, c->bsh, c->d_rank + (size_t) b_lo * m,
		                 c->d_keyd + (size_t) b_lo * m, c->d_nkeys + b_lo, (uint64_t) b_lo * c->B, c->d_bk, per, c->bk_cap_words, c->d_flags + 64, todo, only);
	}
	if (!keyspace || sweep_after)
		launch_rank(c, my_blocks, c->B, c->nblocks, c->d_rank + (size_t) b_lo * m, c->d_keyd + (size_t) b_lo * m, c->d_nkeys + b_lo, (uint64_t) b_lo * c->B,
		            keyspace && tree ? todo : nullptr);
	HIP_TRY(c, hipEventRecord(c->ev[1], st));
	if (sharded && c->tune.inject_failure_rank >= 0 && (uint32_t) c->tune.inject_failure_rank == sh.rank)
		return fail(c, FSEQ_E_OOM, "injected failure (FSEQ_INJECT_FAILURE_RANK)");
	if (sync_at(c, 'A')) { fprintf(stderr, "[fseq] phase A queued\n"); HIP_TRY(c, hipStreamSynchronize(st)); fprintf(stderr, "[fseq] phase A done\n"); }
	if (c->tune.check_phase_a)
	{
		// diagnostic: the key blocks must be well-formed before anything indexes with them (ranks < nkeys <= m, the
		// divergence in front of a key inside the block's columns)
		HIP_TRY(c, hipStreamSynchronize(st));
		std::vector<uint32_t> rk(m), kd(m);
		for (uint32_t b = b_lo; b < b_hi; ++b)
		{
			uint32_t nk = 0;
			HIP_TRY(c, hipMemcpy(&nk, c->d_nkeys + b, 4, hipMemcpyDeviceToHost));
			HIP_TRY(c, hipMemcpy(rk.data(), c->d_rank + (size_t) b * m, (size_t) m * 4, hipMemcpyDeviceToHost));
			HIP_TRY(c, hipMemcpy(kd.data(), c->d_keyd + (size_t) b * m, (size_t) m * 4, hipMemcpyDeviceToHost));
			uint64_t const k0 = (uint64_t) b * c->B, k1 = std::min<uint64_t>(n, k0 + c->B);
			uint32_t bad_r = 0, bad_k = 0;
			for (uint32_t i = 0; i < m; ++i) if (rk[i] >= nk) ++bad_r;
			for (uint32_t j = 0; j < nk && j < m; ++j) if (kd[j] <= k0 || kd[j] > k1) ++bad_k;
			if (nk == 0 || nk > m || bad_r || bad_k)
			{
				char what[200];
				snprintf(what, sizeof(what), "phase A check: block %u has %u keys (m = %u), %u ranks out of range, %u key divergences outside (%llu, %llu]",
				         b, nk, m, bad_r, bad_k, (unsigned long long) k0, (unsigned long long) k1);
				return fail(c, FSEQ_E_HIP, what);
			}
		}
	}
	return FSEQ_OK;
}

// ---- phase B: the exact boundary state of every block
int long_phase_b(fseq_ctx *c, LongRun &R)
{
	FSEQ_LONG_LOCALS(c);
	(void) R;
	if (!sharded)
	{
		// phase B: up the levels -- compose groups of G key blocks of a level into one key block of the next
		// (parallel, from the identity) --, chain the few key blocks of the top level (one workgroup), down the levels --
		// expand every group from the boundary state the level above gave it (parallel)
		uint32_t const G = c->chain_fan;
		size_t const top = c->levels.size();
		auto rank_of = [&](size_t i) { return i ? c->levels[i - 1].rank : c->d_rank; };
		auto keyd_of = [&](size_t i) { return i ? c->levels[i - 1].keyd : c->d_keyd; };
		auto nkeys_of = [&](size_t i) { return i ? c->levels[i - 1].nkeys : c->d_nkeys; };
		auto sa_of = [&](size_t i) { return i ? c->levels[i - 1].state_a : c->d_bstate_a; };
		auto sd_of = [&](size_t i) { return i ? c->levels[i - 1].state_d : c->d_bstate_d; };
		auto count_of = [&](size_t i) { return i ? c->levels[i - 1].count : c->nblocks; };
		auto cols_of = [&](size_t i) { return i ? c->levels[i - 1].cols : (uint64_t) c->B; };
		for (size_t i = 1; i <= top; ++i)
			launch_chain(c, count_of(i), rank_of(i - 1), keyd_of(i - 1), nkeys_of(i - 1), count_of(i - 1), G, cols_of(i - 1), nullptr, nullptr,
			             nullptr, nullptr, rank_of(i), keyd_of(i), nkeys_of(i));
		launch_chain(c, 1, rank_of(top), keyd_of(top), nkeys_of(top), count_of(top), count_of(top), cols_of(top), nullptr, nullptr,
		             sa_of(top), sd_of(top), nullptr, nullptr, nullptr);
		for (size_t i = top; i-- > 0;)
			launch_chain(c, count_of(i + 1), rank_of(i), keyd_of(i), nkeys_of(i), count_of(i), G, cols_of(i), sa_of(i + 1), sd_of(i + 1),
			             sa_of(i), sd_of(i), nullptr, nullptr, nullptr);
	}
	else
	{
		// Sharded phase B: rank r is hyper-block r.  Up the levels over my own block range (fan F, as on one GPU), my
		// last composites into my hyper key block, exchange the W hyper key blocks (the one collective of pass 1's column
		// work), chain them (every rank, same result), then down again from the state in front of my hyper-block.
		uint32_t const F = c->chain_fan, NH = c->n_hyper, K = c->shard_k, Q = c->shard_q;
		bool const have = sh.rank < NH;
		auto rank_of = [&](size_t i) { return i ? c->levels[i - 1].rank : c->d_rank; };
		auto keyd_of = [&](size_t i) { return i ? c->levels[i - 1].keyd : c->d_keyd; };
		auto nkeys_of = [&](size_t i) { return i ? c->levels[i - 1].nkeys : c->d_nkeys; };
		auto sa_of = [&](size_t i) { return i ? c->levels[i - 1].state_a : c->d_bstate_a; };
		auto sd_of = [&](size_t i) { return i ? c->levels[i - 1].state_d : c->d_bstate_d; };
		auto count_of = [&](size_t i) { return i ? c->levels[i - 1].count : c->nblocks; };
		auto cols_of = [&](size_t i) { return i ? c->levels[i - 1].cols : (uint64_t) c->B; };
		// my items of level i: [lo_i, hi_i) (rank boundaries are multiples of F^K blocks)
		std::vector<uint32_t> lo(K + 1), hi(K + 1);
		lo[0] = b_lo; hi[0] = b_hi;
		for (uint32_t i = 1; i <= K; ++i) { lo[i] = lo[i - 1] / F; hi[i] = (hi[i - 1] + F - 1) / F; }
		for (uint32_t i = 1; have && i <= K; ++i)
			launch_chain(c, hi[i] - lo[i], rank_of(i - 1), keyd_of(i - 1), nkeys_of(i - 1), count_of(i - 1), F, cols_of(i - 1), nullptr, nullptr,
			             nullptr, nullptr, rank_of(i), keyd_of(i), nkeys_of(i), lo[i]);
		if (have)
			launch_chain(c, 1, rank_of(K), keyd_of(K), nkeys_of(K), count_of(K), Q, cols_of(K), nullptr, nullptr,
			             nullptr, nullptr, c->d_hrank, c->d_hkeyd, c->d_hnkeys, sh.rank);
		{
			// xbuf: [hrank NH x m][hkeyd NH x m][hnkeys NH]
			size_t const w = (size_t) NH * m;
			HIP_TRY(c, hipMemsetAsync(sh.xbuf, 0, (2 * w + NH) * 4, st));
			if (have)
			{
				HIP_TRY(c, hipMemcpyAsync(sh.xbuf + (size_t) sh.rank * m, c->d_hrank + (size_t) sh.rank * m, (size_t) m * 4, hipMemcpyDeviceToDevice, st));
				HIP_TRY(c, hipMemcpyAsync(sh.xbuf + w + (size_t) sh.rank * m, c->d_hkeyd + (size_t) sh.rank * m, (size_t) m * 4, hipMemcpyDeviceToDevice, st));
				HIP_TRY(c, hipMemcpyAsync(sh.xbuf + 2 * w + sh.rank, c->d_hnkeys + sh.rank, 4, hipMemcpyDeviceToDevice, st));
			}
			if ((rc = shard_exchange(c, 2 * w + NH, 0))) return rc;
			HIP_TRY(c, hipMemcpyAsync(c->d_hrank, sh.xbuf, w * 4, hipMemcpyDeviceToDevice, st));
			HIP_TRY(c, hipMemcpyAsync(c->d_hkeyd, sh.xbuf + w, w * 4, hipMemcpyDeviceToDevice, st));
			HIP_TRY(c, hipMemcpyAsync(c->d_hnkeys, sh.xbuf + 2 * w, (size_t) NH * 4, hipMemcpyDeviceToDevice, st));
		}
		launch_chain(c, 1, c->d_hrank, c->d_hkeyd, c->d_hnkeys, NH, NH, (uint64_t) sh.bpr * c->B, nullptr, nullptr,
		             c->d_hstate_a, c->d_hstate_d, nullptr, nullptr, nullptr);
		if (have)
		{
			launch_chain(c, 1, rank_of(K), keyd_of(K), nkeys_of(K), count_of(K), Q, cols_of(K), c->d_hstate_a, c->d_hstate_d,
			             sa_of(K), sd_of(K), nullptr, nullptr, nullptr, sh.rank);
			for (uint32_t i = K; i >= 1; --i)
				launch_chain(c, hi[i] - lo[i], rank_of(i - 1), keyd_of(i - 1), nkeys_of(i - 1), count_of(i - 1), F, cols_of(i - 1), sa_of(i), sd_of(i),
				             sa_of(i - 1), sd_of(i - 1), nullptr, nullptr, nullptr, lo[i]);
			// the state behind my last block = in front of the next rank's hyper-block (or behind the whole alignment,
			// which the expansion has written itself): my halo block starts from it
			if (b_hi < c->nblocks)
			{
				HIP_TRY(c, hipMemcpyAsync(c->d_bstate_a + (size_t) b_hi * m, c->d_hstate_a + (size_t) (sh.rank + 1u) * m, (size_t) m * 4, hipMemcpyDeviceToDevice, st));
				HIP_TRY(c, hipMemcpyAsync(c->d_bstate_d + (size_t) b_hi * m, c->d_hstate_d + (size_t) (sh.rank + 1u) * m, (size_t) m * 4, hipMemcpyDeviceToDevice, st));
			}
		}
	}
	HIP_TRY(c, hipEventRecord(c->ev[2], st));
	HIP_TRY(c, hipGetLastError());
	FSEQ_RANGE_POP();
	R.range_ab_open = false;
	progress(c, FSEQ_STAGE_TRACEBACK, n / 5, n);                  // (phases A and B queued: about a fifth of pass 1)
	if (sync_at(c, 'B')) { fprintf(stderr, "[fseq] phase B queued\n"); HIP_TRY(c, hipStreamSynchronize(st)); fprintf(stderr, "[fseq] phase B done\n"); }
	return FSEQ_OK;
}

// ---- the list capacity X: what the caller asked for, what worked last time, or an estimate from the boundary states
int long_list_capacity(fseq_ctx *c, LongRun &R)
{
	FSEQ_LONG_LOCALS(c);
	uint32_t &X = R.X;
	if (!p.list_cap && !c->X_hint)
	{
		// first run on this input: size the lists from the block boundary states (k_boundary_recent)
		// (sharded: every rank looks at its own boundaries, the ranks then agree on the largest estimate)
		std::vector<uint32_t> recent(my_blocks ? my_blocks + 1 : 0);
		if (my_blocks)
		{
			hipLaunchKernelGGL(k_boundary_recent, dim3(my_blocks + 1), dim3(256), 0, st, c->d_bstate_d, m, n, c->B, (uint32_t) L, c->d_recent, b_lo);
			HIP_TRY(c, hipMemcpyAsync(recent.data(), c->d_recent, recent.size() * 4, hipMemcpyDeviceToHost, st));
		}
		HIP_TRY(c, hipStreamSynchronize(st));
		recent.erase(std::remove(recent.begin(), recent.end(), 0xFFFFFFFFu), recent.end());
		if (!recent.empty())
		{
			std::nth_element(recent.begin(), recent.begin() + recent.size() / 2, recent.end());
			uint64_t const med = recent[recent.size() / 2];
			// (a quarter above the median, to the next multiple of 64 -- not the next 2^k - 1: the lists of BASELINE C4 are
			// 5,000,000 x (X + 3) x 8 bytes, and what they do not take goes to the stride states of pass 2)
			uint64_t const want = med + med / 4;
			if (X < want) X = (uint32_t) (((want + 63) & ~63ull) - 1);
			if (c->tune.debug)
				fprintf(stderr, "[fseq] list capacity estimate: %zu boundaries, median recent count %llu -> X = %u\n",
				        recent.size(), (unsigned long long) med, X);
		}
	}
	if (sharded)
	{
		HIP_TRY(c, hipMemcpyAsync(sh.xbuf, &X, 4, hipMemcpyHostToDevice, st));
		if ((rc = shard_exchange(c, 1, 1))) return rc;
		HIP_TRY(c, hipMemcpy(&X, sh.xbuf, 4, hipMemcpyDeviceToHost));
	}
	if (X >= m) X = m;

	return FSEQ_OK;
}

// ---- one attempt with list capacity R.X: phase C (column updates + lists), phase D (the DP), traceback, merge walk.
// *overflow_out: some DP cell or merge threshold needed more of a list than X entries hold (the caller retries).
// ---- follow_traceback and find_segments_greedy for one attempt (the lists held their own so far): the traceback on
// the device, the merge walk over one threshold per traceback boundary on the host.  *overflow: a threshold or a merged
// size needed more of a list than it holds.
int long_traceback_and_merge(fseq_ctx *c, LongRun &R, double th0, bool *overflow_out)
{
	FSEQ_LONG_LOCALS(c);
	(void) R;
	bool overflow = false;
	if (!overflow)
	{
		if ((rc = follow_traceback(c, st))) return rc;
		if (c->tune.debug) fprintf(stderr, "[fseq] host: traceback walk + gather %.3f ms\n", now_ms() - th0);
		uint32_t const max_seg = c->traceback.back().segment_max_size;
		c->res.max_segment_size = max_seg;
		c->res.dp_segment_count = c->traceback.size();
		c->res.short_path = 0;
		size_t const S = c->traceback.size();

		// ---- find_segments_greedy (lp.cc:335-390).  Its test #{d_rb > current_lb} <= max_segment_size (:363-364)
		// holds exactly for current_lb >= tau_rb; tau comes from the list of column rb - 1 where that list lives
		// (k_seg_tau: one number per traceback boundary instead of the lists; sharded: every rank for its columns).
		c->segments.clear();
		if (max_seg < m)
		{
			uint64_t const own_lo = held_lo(c), own_hi = sharded ? sh.c_hi : n;      // columns whose lists I answer for
			std::vector<uint2> tau(S);
			if (S > 1 && c->tau_host.size() == S)
				tau = c->tau_host;                                          // came back with the traceback
			else if (S > 1)
			{
				if (c->cols_cap < 2 * S) { if ((rc = dev_alloc(c, &c->d_cols, 2 * S))) return rc; c->cols_cap = 2 * S; }
				if (c->tau_cap < S) { if ((rc = dev_alloc(c, &c->d_tau, S))) return rc; c->tau_cap = S; }
				std::vector<uint64_t> cols(S);
				for (size_t j = 0; j < S; ++j) cols[j] = c->traceback[j].rb - 1;
				HIP_TRY(c, hipMemcpyAsync(c->d_cols, cols.data(), S * 8, hipMemcpyHostToDevice, st));
				hipLaunchKernelGGL(k_seg_tau, dim3((uint32_t) S), dim3(64), 0, st, c->d_cols, own_lo, own_hi, max_seg, c->stride, c->d_ent, c->d_hdr, c->d_tau);
				if (sharded)
				{
					HIP_TRY(c, hipMemcpyAsync(sh.xbuf, c->d_tau, S * 8, hipMemcpyDeviceToDevice, st));
					if ((rc = shard_exchange(c, 2 * S, 0))) return rc;
					HIP_TRY(c, hipMemcpyAsync(tau.data(), sh.xbuf, S * 8, hipMemcpyDeviceToHost, st));
				}
				else
					HIP_TRY(c, hipMemcpyAsync(tau.data(), c->d_tau, S * 8, hipMemcpyDeviceToHost, st));
				HIP_TRY(c, hipStreamSynchronize(st));
				HIP_TRY(c, hipGetLastError());
			}
			// the walk itself; a merged segment's size is the count at its last boundary (:366), asked for afterwards
			struct Pending { size_t seg; uint64_t col, lb; };
			std::vector<Pending> ask;
			uint64_t current_lb = 0;
			uint64_t prev_size = c->traceback[0].segment_size;
			bool prev_size_pending = false;
			size_t prev = 0;
			auto emit = [&]() {
				fseq_segment sg{current_lb, c->traceback[prev].rb, (uint32_t) prev_size, 0};
				if (prev_size_pending) ask.push_back(Pending{c->segments.size(), c->traceback[prev].rb - 1, current_lb});
				c->segments.push_back(sg);
			};
			for (size_t j = 1; j < S && !overflow; ++j)
			{
				uint2 const t = tau[j];
				bool const fits = t.y != SEG_TAU_NEVER && current_lb >= t.x;
				if (!fits && t.y == SEG_TAU_OPEN) { overflow = true; break; }     // the list ended before it could tell
				if (fits)
					prev_size_pending = true;                                       // prev_size = the count at boundary j (:366)
				else
				{
					emit();
					prev_size = c->traceback[j].segment_size;
					prev_size_pending = false;
					current_lb = c->traceback[prev].rb;
				}
				prev = j;
			}
			if (!overflow)
			{
				emit();
				if (!ask.empty())
				{
					size_t const Q = ask.size();
					if ((rc = pin_reserve(c, Q * 20 + 64))) return rc;
					uint64_t *const qc = pin_take<uint64_t>(c, 2 * Q);
					for (size_t i = 0; i < Q; ++i) { qc[i] = ask[i].col; qc[Q + i] = ask[i].lb; }
					uint32_t *const cnt = pin_take<uint32_t>(c, Q);
					if (c->cols_cap < 2 * Q) { if ((rc = dev_alloc(c, &c->d_cols, 2 * Q))) return rc; c->cols_cap = 2 * Q; }
					if (c->tau_cap < Q) { if ((rc = dev_alloc(c, &c->d_tau, Q))) return rc; c->tau_cap = Q; }
					HIP_TRY(c, hipMemcpyAsync(c->d_cols, qc, 2 * Q * 8, hipMemcpyHostToDevice, st));
					uint32_t *d_cnt = reinterpret_cast<uint32_t *>(c->d_tau);
					hipLaunchKernelGGL(k_seg_count, dim3((uint32_t) Q), dim3(64), 0, st, c->d_cols, c->d_cols + Q, own_lo, own_hi, c->stride, c->d_ent, c->d_hdr, d_cnt);
					if (sharded)
					{
						HIP_TRY(c, hipMemcpyAsync(sh.xbuf, d_cnt, Q * 4, hipMemcpyDeviceToDevice, st));
						if ((rc = shard_exchange(c, Q, 0))) return rc;
						HIP_TRY(c, hipMemcpyAsync(cnt, sh.xbuf, Q * 4, hipMemcpyDeviceToHost, st));
					}
					else
						HIP_TRY(c, hipMemcpyAsync(cnt, d_cnt, Q * 4, hipMemcpyDeviceToHost, st));
					HIP_TRY(c, hipStreamSynchronize(st));
					HIP_TRY(c, hipGetLastError());
					for (size_t i = 0; i < Q; ++i) c->segments[ask[i].seg].segment_size = cnt[i];
				}
			}
		}
	}
	*overflow_out = overflow;
	return FSEQ_OK;
}


// ---- [r5] phase C on representative rows (fseq_reduced.hpp): the plan of one attempt.
// k_reduce_prep leaves, per block, the representatives and the reduced start state; the host reads the counts back and
// sorts the blocks into the configurations that hold them (a launch per configuration in use) and the blocks that run on all
// rows: more representatives than any configuration holds (or than pay: > 70 % of the rows), or lists an earlier run on this
// input could not prove on the representatives.  *use: false when more than a quarter of the blocks would run on all rows
// anyway -- the attempt then takes the run on all rows with its stride states (diverse inputs).
// bytes of each staged-column buffer of a reduced configuration
uint32_t red_symcap(fseq_ctx const *c, ReducedSet const &rs, bool direct)
{
	uint32_t const bytes = direct ? sym_bytes(c->p.m, c->bsh) : sym_bytes(rs.rows, c->bsh);
	(void) rs;
	return (bytes + 1023u) & ~1023u;                           // whole kilobytes: a wave stages sixteen bytes per lane
}

bool columns_fit_reduced(fseq_ctx const *c, ReducedSet const &rs, bool direct)
{
	// (value ids of a block -- its boundary values and one per column -- are 16-bit keys of the partition step)
	return rs.lds(c->B, red_symcap(c, rs, direct)) <= LDS_LIMIT && (uint64_t) rs.rows + c->B + 1u <= 65535u;
}

void red_fill_args(fseq_ctx *c, RedArgs &RA)
{
	RA.cnt = c->d_red_cnt; RA.vmin = c->d_red_vmin; RA.a = c->d_red_a; RA.d = c->d_red_d; RA.leaf = c->d_red_leaf;
	RA.invalid = c->d_red_invalid; RA.any_invalid = c->d_red_invalid + c->nblocks; RA.cap = c->red_cap; RA.m_true = c->p.m;
	RA.direct = c->red_direct ? 1u : 0u; RA.colbytes = sym_bytes(c->p.m, c->bsh); RA.rank = c->d_rank;
	RA.ss_a = c->d_red_ss_a; RA.ss_d = c->d_red_ss_d; RA.ss_stride = c->red_ss_stride; RA.ss_cap = c->red_ss_cap;
}

// ---- [r5] phase C on representative rows (fseq_reduced.hpp): the plan of one attempt.
// k_reduce_prep leaves, per block, the representatives and the reduced start state.  The blocks are sorted into the
// configurations that hold them (a launch per configuration in use, side by side on their own streams) and the blocks
// that run on all rows: more representatives than any configuration holds (or than pay: > 70 % of the rows), or lists an
// earlier run on this input could not prove on the representatives.  The first run on an input (or at a new capacity) reads
// the counts back and plans; later runs launch by the same plan without waiting and have the device check that the counts
// are the ones the plan was made from (flags word 1; the attempt is repeated with a fresh plan if not).
// *use: false when more than a quarter of the blocks would run on all rows anyway -- the attempt then takes the run on
// all rows with its stride states (diverse inputs).
int red_plan(fseq_ctx *c, uint32_t X, bool *use)
{
	FSEQ_LONG_LOCALS(c);
	*use = false;
	uint32_t cap = std::min<uint32_t>(m, 11264u);
	if (c->tune.reduced_cap) cap = std::min<uint32_t>(cap, (uint32_t) c->tune.reduced_cap);
	uint32_t const nbk = c->nblocks;
	if (!c->d_red_cnt || c->red_cap != cap)
	{
		// (the small per-block words for every block of the alignment; the per-block rows for my blocks only -- a rank of a
		// sharded run --, addressed by the block's place in the whole alignment like the key blocks and boundary states)
		if ((rc = dev_alloc(c, &c->d_red_cnt, nbk))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_cnt_plan, nbk))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_vmin, nbk))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_invalid, nbk + 2))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_blocks, 3 * (size_t) nbk))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_red_rows_alloc, &c->d_red_rows, b_lo, b_hi, cap))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_red_leaf_alloc, &c->d_red_leaf, b_lo, b_hi, cap))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_red_a_alloc, &c->d_red_a, b_lo, b_hi, cap))) return rc;
		if ((rc = dev_alloc_range(c, &c->d_red_d_alloc, &c->d_red_d, b_lo, b_hi, cap))) return rc;
		c->red_cap = cap;
		c->red_plan_valid = false;
	}
	if (!my_blocks) return FSEQ_OK;                            // (a rank without blocks)
	// (the last run on this input at this capacity found the representatives not worth it: the same input gives the same answer)
	if (c->red_declined && c->red_declined_X == X && !c->tune.reduced_always) return FSEQ_OK;
	if (c->red_pin_words < 4 * (size_t) nbk + 64)
	{
		if (c->h_red_pin) (void) hipHostFree(c->h_red_pin);
		c->h_red_pin = nullptr; c->red_pin_words = 0;
		HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_red_pin), (4 * (size_t) nbk + 64) * 4, hipHostMallocDefault));
		c->red_pin_words = 4 * (size_t) nbk + 64;
	}
	if (!c->red_ev[0])
		for (auto &evt : c->red_ev) HIP_TRY(c, hipEventCreateWithFlags(&evt, hipEventDisableTiming));
	// LDS-resident row counts: the representatives' symbols come from the alignment's own columns (a block stages whole columns)
	c->red_direct = !c->use_stream;
	RedPrepArgs A{};
	A.bstate_a = c->d_bstate_a; A.bstate_d = c->d_bstate_d; A.rank = c->d_rank; A.blocks = nullptr;
	A.m = m; A.B = c->B; A.L = (uint32_t) L; A.cap = cap; A.block0 = b_lo; A.leaf_only = 0; A.n = n; A.direct = c->red_direct ? 1u : 0u;
	A.Xp = X + (c->tune.reduced_margin >= 0 ? (uint32_t) c->tune.reduced_margin : X / 4u + 8u);
	A.cnt = c->d_red_cnt; A.vmin = c->d_red_vmin; A.rows = c->d_red_rows; A.leaf = c->d_red_leaf; A.a = c->d_red_a; A.d = c->d_red_d;
	A.invalid = c->d_red_invalid; A.flags = c->d_red_invalid + nbk;
	HIP_TRY(c, launch_reduce_prep(st, my_blocks, A));
	if (c->red_plan_valid && c->red_plan_X == X && c->red_force_full.size() == nbk)
	{
		launch_reduce_check(st, c->d_red_cnt + b_lo, c->d_red_cnt_plan + b_lo, my_blocks, c->d_red_invalid + nbk);
		if (!c->red_direct)
			launch_reduce_msa(st, c->red_listed, c->red_max_rows, c->d_msa, c->ld, c->d_red_msa, c->red_ld, c->d_red_cnt, c->d_red_rows, cap, n, c->B, c->bsh, c->d_red_blocks, m, c->tune.reduced_msa_gather);
		c->tm.reduced_blocks = c->red_plan_blocks; c->tm.reduced_rows_mean = c->red_plan_rows_mean;
		*use = true;
		return FSEQ_OK;
	}
	uint32_t *const h_cnt = c->h_red_pin;
	HIP_TRY(c, hipMemcpyAsync(h_cnt + b_lo, c->d_red_cnt + b_lo, (size_t) my_blocks * 4, hipMemcpyDeviceToHost, st));
	HIP_TRY(c, hipMemcpyAsync(c->d_red_cnt_plan + b_lo, c->d_red_cnt + b_lo, (size_t) my_blocks * 4, hipMemcpyDeviceToDevice, st));
	HIP_TRY(c, hipStreamSynchronize(st));
	c->red_cnt_host.assign(nbk, RED_NONE);
	std::copy(h_cnt + b_lo, h_cnt + b_hi, c->red_cnt_host.begin() + b_lo);
	if (c->red_force_full.size() != nbk) c->red_force_full.assign(nbk, 0);
	c->red_full.assign(nbk, 0);
	c->red_config_of.assign(nbk, -1);
	c->red_config_snap_of.assign(nbk, -1);
	int const nconf = reduced_config_count();
	std::vector<ReducedSet> sets((size_t) nconf);
	std::vector<uint8_t> usable((size_t) nconf), usable_snap((size_t) nconf);
	for (int i = 0; i < nconf; ++i)
	{
		(void) reduced_config(i, &sets[(size_t) i]);
		ReducedSet const &rs = sets[(size_t) i];
		// (small blocks: one-wave workgroups, or two with the list on a wave of its own: FSEQ_REDUCED_EW picks the latter; from 256
		// threads on phase C takes the configurations with a list wave, pass 2's sweeps the others)
		bool const fit = columns_fit_reduced(c, rs, c->red_direct);
		usable[(size_t) i] = fit && (rs.T > 128u ? rs.ew : rs.ew == c->tune.reduced_ew);
		usable_snap[(size_t) i] = fit && (rs.T > 128u ? !rs.ew : rs.ew == c->tune.reduced_ew);
	}
	std::vector<std::vector<uint32_t>> per((size_t) nconf);
	uint32_t n_full = 0, max_rows = 0, listed = 0;
	uint64_t sum_rows = 0;
	uint32_t *const h_blocks = c->h_red_pin + nbk;             // [0, listed): every reduced block; then the configurations' lists
	for (uint32_t b = b_lo; b < b_hi; ++b)
	{
		uint32_t const r = c->red_cnt_host[b];
		if (r != RED_NONE)
		{
			h_blocks[listed++] = b;
			max_rows = std::max(max_rows, r);
			int cf = -1, cs = -1;
			for (int i = 0; i < nconf; ++i) if (usable[(size_t) i] && sets[(size_t) i].rows >= r) { cf = i; break; }
			for (int i = 0; i < nconf; ++i) if (usable_snap[(size_t) i] && sets[(size_t) i].rows >= r) { cs = i; break; }
			c->red_config_of[b] = cf;
			c->red_config_snap_of[b] = cs;
		}
		bool const full = r == RED_NONE || c->red_config_of[b] < 0 || c->red_force_full[b] || (uint64_t) r * 10u > (uint64_t) m * 7u;
		if (full) { c->red_full[b] = 1; ++n_full; }
		else { per[(size_t) c->red_config_of[b]].push_back(b); sum_rows += r; }
	}
	c->red_listed = listed; c->red_max_rows = max_rows;
	c->tm.reduced_blocks = my_blocks - n_full;
	c->tm.reduced_rows_mean = my_blocks > n_full ? (uint32_t) (sum_rows / (my_blocks - n_full)) : 0u;
	if (c->tune.debug)
		fprintf(stderr, "[fseq] reduced phase C: %u of %u blocks on their representatives (mean %u of %u rows, most %u), %u on all rows\n", my_blocks - n_full, my_blocks,
		        c->tm.reduced_rows_mean, m, max_rows, n_full);
	if (c->tune.debug)
		for (int i = 0; i < nconf; ++i)
			if (!per[(size_t) i].empty())
			{
				uint64_t sr = 0;
				for (uint32_t b : per[(size_t) i]) sr += c->red_cnt_host[b];
				fprintf(stderr, "[fseq]   configuration of %u rows: %zu blocks, %llu representatives on average\n", sets[(size_t) i].rows, per[(size_t) i].size(),
				        (unsigned long long) (sr / per[(size_t) i].size()));
			}
	// worth it?  The run on all rows is the tuned one (three workgroups per CU, stride states for pass 2), and a row of a small
	// reduced workgroup costs more than a row there: the representatives take over where they are clearly fewer -- rows to
	// update in all, a block on all rows counted as one and a half (its boundaries are reached from the block's start) -- below
	// a fifth of the rows (BASELINE C3 / C4 / C5: 8 / 7 / 6 %; C3's shape with ten times the mutations, 40 %: 11.4 ms against
	// 9.4 on all rows, tools/diversity_sweep.py; FSEQ_REDUCED_ALWAYS: tests)
	{
		uint64_t const rows_all = sum_rows + (uint64_t) n_full * m * 3u / 2u;
		if ((!c->tune.reduced_always && (rows_all * 5u > (uint64_t) my_blocks * m || (uint64_t) n_full * 4u > my_blocks)) || (uint64_t) n_full >= my_blocks)
		{
			c->red_declined = true; c->red_declined_X = X;
			return FSEQ_OK;
		}
	}
	if (n_full && c->use_stream && !c->s2.T) return FSEQ_OK;     // (the first form of the streamed kernel takes no block list)
	if (!c->red_direct)
	{
		// the reduced alignment: column k at d_red_msa + k * red_ld
		// (my columns only: column k at d_red_msa + k * red_ld)
		size_t const ldr = ((size_t) sym_bytes(max_rows ? max_rows : 1u, c->bsh) + 15) & ~size_t(15);
		uint64_t const k_lo = (uint64_t) b_lo * c->B, k_hi = std::min<uint64_t>(n, (uint64_t) b_hi * c->B);
		size_t const need = (size_t) (k_hi - k_lo) * ldr + 64;
		if (c->red_msa_bytes < need) { if ((rc = dev_alloc(c, &c->d_red_msa_alloc, need))) return rc; c->red_msa_bytes = need; }
		c->red_ld = ldr;
		c->d_red_msa = c->d_red_msa_alloc - (size_t) k_lo * ldr;
	}
	{
		// the reduced states for pass 2: every 16 columns (32: streamed rows), rows for the most representatives of a block
		uint32_t const stride_ = c->use_stream ? 32u : 16u;
		uint32_t const scap = (std::max(max_rows, 1u) + 63u) & ~63u;
		uint64_t const q_lo = (uint64_t) b_lo * c->B / stride_, q_hi = std::min<uint64_t>(n, (uint64_t) b_hi * c->B) / stride_;
		size_t const words = ((size_t) (q_hi - q_lo) + 2) * scap;
		if (c->red_ss_words < words)
		{
			if ((rc = dev_alloc(c, &c->d_red_ss_a_alloc, words))) return rc;
			if ((rc = dev_alloc(c, &c->d_red_ss_d_alloc, words))) return rc;
			c->red_ss_words = words;
		}
		c->red_ss_stride = stride_; c->red_ss_cap = scap;
		c->d_red_ss_a = c->d_red_ss_a_alloc - (size_t) q_lo * scap;       // (the state at column q * stride at [q][scap])
		c->d_red_ss_d = c->d_red_ss_d_alloc - (size_t) q_lo * scap;
	}
	c->red_bins.clear();
	uint32_t at = listed;
	for (int i = 0; i < nconf; ++i)
	{
		auto const &v = per[(size_t) i];
		if (v.empty()) continue;
		std::copy(v.begin(), v.end(), h_blocks + at);
		c->red_bins.push_back(fseq_ctx::RedBin{i, at, (uint32_t) v.size()});
		at += (uint32_t) v.size();
	}
	c->red_full_at = at; c->red_nfull = 0;
	for (uint32_t b = b_lo; b < b_hi; ++b) if (c->red_full[b]) h_blocks[at + c->red_nfull++] = b;
	at += c->red_nfull;
	HIP_TRY(c, hipMemcpyAsync(c->d_red_blocks, h_blocks, (size_t) at * 4, hipMemcpyHostToDevice, st));
	if (!c->red_direct)
		launch_reduce_msa(st, listed, max_rows, c->d_msa, c->ld, c->d_red_msa, c->red_ld, c->d_red_cnt, c->d_red_rows, cap, n, c->B, c->bsh, c->d_red_blocks, m, c->tune.reduced_msa_gather);
	c->red_plan_valid = true; c->red_plan_X = X;
	c->red_plan_blocks = c->tm.reduced_blocks; c->red_plan_rows_mean = c->tm.reduced_rows_mean;
	*use = true;
	return FSEQ_OK;
}

// launches of the reduced column kernel over lists of workgroups, one per configuration, side by side: the first on the
// context's stream, the others on streams of their own that wait for it and that it waits for
struct RedLaunch { int config; uint32_t first, count; };
int red_launch_all(fseq_ctx *c, std::vector<RedLaunch> const &ls, RedArgs const &base, uint32_t const *blocks, uint32_t const *wg_tasks, uint2 *ent, uint4 *hdr, uint32_t X, uint32_t stride)
{
	FSEQ_LONG_LOCALS(c);
	if (ls.empty()) return FSEQ_OK;
	// (side streams: the context's second stream first -- every further hardware queue in use slows the dependent launches of
	// phase B, measured on BASELINE C3: 0.71 ms with none, 0.97 with three)
	size_t const want_side = c->tune.reduced_serial ? 0u : (c->tune.reduced_side >= 0 ? (size_t) c->tune.reduced_side : 3u);
	size_t const nside = std::min<size_t>(ls.size() - 1, std::min<size_t>(want_side, 3));
	if (nside) HIP_TRY(c, hipEventRecord(c->red_ev[3], st));
	for (size_t i = 0; i < ls.size(); ++i)
	{
		ReducedSet rs;
		(void) reduced_config(ls[i].config, &rs);
		RedArgs RA = base;
		RA.symcap = red_symcap(c, rs, c->red_direct);
		size_t const lds = rs.lds(c->B, RA.symcap);
		HIP_TRY(c, rs.prepare(lds));
		RA.blocks = blocks + ls[i].first;
		if (wg_tasks) RA.wg_tasks = wg_tasks + 3 * (size_t) ls[i].first;
		// the largest launches on the side streams, the rest in turn on the context's
		hipStream_t s_ = (i >= 1 && i <= nside) ? (i == 1 ? c->stream2 : c->red_st[i - 2]) : st;
		if (s_ != st && s_ == nullptr)
		{
			HIP_TRY(c, hipStreamCreateWithFlags(&c->red_st[i - 2], hipStreamNonBlocking));
			s_ = c->red_st[i - 2];
		}
		if (s_ != st) HIP_TRY(c, hipStreamWaitEvent(s_, c->red_ev[3], 0));
		rs.launch(s_, ls[i].count, lds, c->red_direct ? c->d_msa : c->d_red_msa, c->red_direct ? c->ld : c->red_ld, n, c->B, (uint32_t) L, X, stride, ent, hdr, c->npass, c->bsh, RA);
		if (s_ != st) HIP_TRY(c, hipEventRecord(c->red_ev[i - 1], s_));
	}
	for (size_t i = 0; i < nside; ++i) HIP_TRY(c, hipStreamWaitEvent(st, c->red_ev[i], 0));
	HIP_TRY(c, hipGetLastError());
	return FSEQ_OK;
}

// the lists of the reduced blocks
int red_columns(fseq_ctx *c)
{
	RedArgs RA;
	red_fill_args(c, RA);
	std::vector<RedLaunch> ls;
	for (auto const &bin : c->red_bins) ls.push_back(RedLaunch{bin.config, bin.first, bin.count});
	// (the configuration with the most blocks first: it stays on the context's stream)
	std::stable_sort(ls.begin(), ls.end(), [](RedLaunch const &x, RedLaunch const &y) { return x.count > y.count; });
	return red_launch_all(c, ls, RA, c->d_red_blocks, nullptr, c->d_ent, c->d_hdr, c->X, c->stride);
}

int long_attempt(fseq_ctx *c, LongRun &R, bool *overflow_out)
{
	FSEQ_LONG_LOCALS(c);
	uint32_t &X = R.X;
	double &ms_c = R.ms_c, &ms_dp = R.ms_dp, &ms_host = R.ms_host;
	bool const keyspace = R.keyspace;
	// [r5] phase C on representative rows: the default wherever the lists are consumed by the speculative DP behind phase C
	// (sharded: a rank's own blocks; its halo block has no state behind it to take the classes from and runs on all rows)
	bool const red_candidate = !c->tune.no_reduced && n >= 2 * L;
	double const t_att = now_ms();
	auto mark = [&](char const *what) { if (c->tune.debug) fprintf(stderr, "[fseq]   attempt +%.3f ms %s\n", now_ms() - t_att, what); };
	if ((rc = ensure_work_buffers(c, X, !red_candidate))) return rc;
	mark("lists allocated");
	// ---- phase C + D
#if defined(FSEQ_DP_STAMPS) || defined(FSEQ_DP_STATS)
	HIP_TRY(c, hipMemsetAsync(c->d_flags, 0, 1024, st));
#else
	HIP_TRY(c, hipMemsetAsync(c->d_flags, 0, 16, st));
#endif
	if (c->tune.poison_lists)
	{
		// tests of the DP-beside-phase-C forms: a list read before it is written must not look right by accident
		HIP_TRY(c, hipMemsetAsync(c->d_ent_alloc, 0xFF, ((size_t) (held_hi(c) - held_lo(c)) * c->stride + 256) * sizeof(uint2), st));
		HIP_TRY(c, hipMemsetAsync(c->d_hdr, 0xFF, (size_t) n * sizeof(uint4), st));
	}
	DpSchedule const S = dp_schedule((uint32_t) L, (uint32_t) n);
	// the DP as chunk-speculative sweeps over the whole chip once every list is written (fseq_dpspec.hpp); the serial kernel for
	// inputs too short for three chunks (and FSEQ_DP_SERIAL).  [r5] the forms that ran the serial DP beside phase C (in parts, or
	// fed by host-visible flags) are gone: no default reached them
	SpecPlan const spec = spec_plan(c, S);
	bool const use_spec = sharded || spec.nchunks() > 0;
	if (sharded && spec.nchunks() < 1) return fail(c, FSEQ_E_UNSUPPORTED, "sharded run: no DP chunk plan");
	// columns phase C covers here: all, or my blocks plus the halo block's first columns (the lists my last DP round reads)
	uint64_t const n_c = sharded ? sh.c_end : n;
	uint32_t spec_overflow = 0, spec_sweeps = 0;
	HIP_TRY(c, hipEventRecord(c->ev[3], st));
	RangeScope range_cd("fseq pass 1: phases C + D (column updates + lists, segmentation DP)");
	// (list [r5]: workgroup i owns block list[i] instead of b0 + i -- the blocks the reduced phase C hands to the run on all rows)
	auto launch_columns = [&](uint32_t b0, uint32_t nb, uint32_t *done = nullptr, uint32_t epoch = 0, uint32_t const *list = nullptr) {
		if (c->use_stream && c->s2.T)
		{
			uint32_t pack_abits = 0;
			if (c->s2.pack) { pack_abits = 1; while ((1u << pack_abits) < m) ++pack_abits; }
			hipLaunchKernelGGL(k_columns_stream2_prologue, dim3(nb), dim3(ST), stream_lds_bytes(0, true), st, m, n_c, c->B, c->d_ws_c, c->d_bstate_a, c->d_bstate_d, b0, pack_abits,
			                   c->ss_ids ? c->d_bs_w : (uint32_t *) nullptr, c->ss_ids ? c->d_bs_h : (uint8_t *) nullptr, list);
			c->s2.launch(st, nb, c->s2_lds, c->d_msa, c->ld, m, n_c, c->B, c->npass, c->bsh, c->d_ws_c, (uint32_t) L, c->X, c->stride, c->d_ent, c->d_hdr,
			             c->snap_stride, c->d_ss_a, c->d_ss_d, b0, done, epoch, c->ss_pack | (c->ss_ids ? S2_SS_IDS : 0u), list);
		}
		else if (c->use_stream && (uint64_t) m + c->B < (1u << 19) && !c->tune.stream_plain_scan)
			hipLaunchKernelGGL(k_columns_stream<19>, dim3(nb), dim3(ST), stream_lds_bytes(sym_bytes(m, c->bsh), c->stream_staged), st, c->d_msa, c->ld, m, n_c, c->B, c->npass, c->bsh, c->d_ws_c, (uint32_t) c->stream_staged,
			                   c->d_bstate_a, c->d_bstate_d, (uint32_t) L, c->X, c->stride, c->d_ent, c->d_hdr, c->snap_stride, c->d_ss_a, c->d_ss_d, b0, done, epoch, c->ss_pack);
		else if (c->use_stream)
			hipLaunchKernelGGL(k_columns_stream<0>, dim3(nb), dim3(ST), stream_lds_bytes(sym_bytes(m, c->bsh), c->stream_staged), st, c->d_msa, c->ld, m, n_c, c->B, c->npass, c->bsh, c->d_ws_c, (uint32_t) c->stream_staged,
			                   c->d_bstate_a, c->d_bstate_d, (uint32_t) L, c->X, c->stride, c->d_ent, c->d_hdr, c->snap_stride, c->d_ss_a, c->d_ss_d, b0, done, epoch, c->ss_pack);
		else
			ks.columns(st, nb, c->lds_columns, c->d_msa, c->ld, m, n_c, c->B, c->N2, c->d_bstate_a, c->d_bstate_d, (uint32_t) L, c->X, c->stride, c->d_ent, c->d_hdr, c->npass, c->bsh,
			           c->snap_stride, c->d_ss_a, c->d_ss_d, b0, done, epoch, c->colmask_ready && c->colmask_use ? c->d_colmask : (uint32_t const *) nullptr, list);
	};
	{
		if (use_spec)
		{
			// the arrays the speculative DP starts from are reset on the second stream while phase C runs
			if ((rc = dp_spec_reset(c, spec, c->stream2))) return rc;
			HIP_TRY(c, hipEventRecord(c->ev_part[15], c->stream2));
		}
		c->red_active = false;
		if (red_candidate)
		{
			bool use = false;
			if ((rc = red_plan(c, X, &use))) return rc;
			mark("reduced plan");
			if (use)
			{
				c->red_active = true;
				if ((rc = red_columns(c))) return rc;
				mark("reduced columns queued");
				// the blocks that run on all rows, in one launch (no stride states: pass 2 reaches their boundaries from the block's start)
				if (c->red_nfull) launch_columns(0, c->red_nfull, nullptr, 0, c->d_red_blocks + c->red_full_at);
				// sharded: the block behind mine for as far as the halo reaches, on all rows (k_columns stops at n_c)
				if (sharded && sh.c_end > sh.c_hi) launch_columns(b_hi, 1u);
			}
			else if ((rc = ensure_work_buffers(c, X, true))) return rc;      // (the stride states after all)
		}
		if (!sharded && !c->red_active) launch_columns(0, c->nblocks);
		if (sync_at(c, 'C')) { fprintf(stderr, "[fseq] phase C queued\n"); HIP_TRY(c, hipStreamSynchronize(st)); fprintf(stderr, "[fseq] phase C done\n"); }
		if (sharded && my_blocks && !c->red_active)
		{
			// my blocks, and the block behind them for as far as the halo reaches (k_columns stops at n_c)
			uint32_t const nb = my_blocks + ((sh.c_end > sh.c_hi) ? 1u : 0u);
			launch_columns(b_lo, nb);
		}
		if (sharded && red_candidate)
		{
			// the ranks agree on whether the attempt stands BEFORE the DP's exchanges: a rank whose lists could not be proven
			// on the representatives (or whose plan's counts have changed) makes every rank run the attempt again
			uint32_t mine[2] = {0u, 0u};
			if (c->red_active)
			{
				HIP_TRY(c, hipMemcpyAsync(mine, c->d_red_invalid + c->nblocks, 8, hipMemcpyDeviceToHost, st));
				HIP_TRY(c, hipStreamSynchronize(st));
			}
			uint32_t word = (mine[0] ? 1u : 0u) | (mine[1] ? 2u : 0u);
			HIP_TRY(c, hipMemcpyAsync(sh.xbuf, &word, 4, hipMemcpyHostToDevice, st));
			if ((rc = shard_exchange(c, 1, 1))) return rc;
			uint32_t all = 0;
			HIP_TRY(c, hipMemcpy(&all, sh.xbuf, 4, hipMemcpyDeviceToHost));
			if (all)
			{
				if (mine[1]) c->red_plan_valid = false;
				if (mine[0])
				{
					std::vector<uint32_t> inv(c->nblocks);
					HIP_TRY(c, hipMemcpy(inv.data(), c->d_red_invalid, (size_t) c->nblocks * 4, hipMemcpyDeviceToHost));
					uint32_t cnt = 0;
					for (uint32_t b = b_lo; b < b_hi; ++b) if (inv[b] && !c->red_full[b]) { c->red_force_full[b] = 1; ++cnt; }
					c->red_plan_valid = false;
					R.redone += cnt;
				}
				R.redo = true; *overflow_out = false;
				HIP_TRY(c, hipEventRecord(c->ev[4], st));
				HIP_TRY(c, hipEventRecord(c->ev_dp[0], st));
				HIP_TRY(c, hipEventRecord(c->ev_dp[1], st));
				HIP_TRY(c, hipStreamWaitEvent(st, c->ev_part[15], 0));
				{ float f = 0; HIP_TRY(c, hipEventSynchronize(c->ev[4])); HIP_TRY(c, hipEventElapsedTime(&f, c->ev[3], c->ev[4])); R.ms_c += f; }
				return FSEQ_OK;
			}
		}
		HIP_TRY(c, hipEventRecord(c->ev[4], st));
		HIP_TRY(c, hipEventRecord(c->ev_dp[0], st));
		if (use_spec)
		{
			HIP_TRY(c, hipStreamWaitEvent(st, c->ev_part[15], 0));      // the DP arrays were reset beside phase C
			if ((rc = run_dp_spec(c, S, spec, st, &spec_overflow, &spec_sweeps, true))) return rc;
		}
		else
			hipLaunchKernelGGL(k_dp<DP_WHOLE>, dim3(1), dim3(1024), dp_lds_bytes(), st, c->dp, c->d_ent, c->d_hdr, c->stride, m, (uint32_t) n, (uint32_t) L,
			                   c->d_flags, 0u, S.nrounds, DpSpecArgs{});
		HIP_TRY(c, hipEventRecord(c->ev_dp[1], st));
	}
	HIP_TRY(c, hipEventRecord(c->ev[5], st));
	HIP_TRY(c, hipGetLastError());
	mark("DP queued");

	if ((rc = pin_reserve(c, 64))) return rc;
	uint32_t *const h_flags = pin_take<uint32_t>(c, 12);
	h_flags[4] = 0;
	HIP_TRY(c, hipMemcpyAsync(h_flags, c->d_flags, 16, hipMemcpyDeviceToHost, st));
	h_flags[5] = 0;
	h_flags[6] = 0;
	if (keyspace) HIP_TRY(c, hipMemcpyAsync(h_flags + 4, c->d_flags + 64, 12, hipMemcpyDeviceToHost, st));
	h_flags[6 + 1] = 0; h_flags[6 + 2] = 0;
	uint32_t *const h_red = h_flags + 7;                         // {a block's lists not proven, the plan's counts have changed}
	if (c->red_active && !sharded) HIP_TRY(c, hipMemcpyAsync(h_red, c->d_red_invalid + c->nblocks, 8, hipMemcpyDeviceToHost, st));      // (sharded: agreed on before the DP)
	HIP_TRY(c, hipStreamSynchronize(st));
	R.redo = false;
	if (c->red_active && h_red[1])
	{
		// (the counts are not what the plan was made from: plan afresh)
		c->red_plan_valid = false;
		R.redo = true; *overflow_out = false;
		return FSEQ_OK;
	}
	if (c->red_active && h_red[0])
	{
		std::vector<uint32_t> inv(c->nblocks);
		HIP_TRY(c, hipMemcpy(inv.data(), c->d_red_invalid, (size_t) c->nblocks * 4, hipMemcpyDeviceToHost));
		uint32_t cnt = 0;
		for (uint32_t b = 0; b < c->nblocks; ++b) if (inv[b] && !c->red_full[b]) { c->red_force_full[b] = 1; ++cnt; }
		if (c->tune.debug) fprintf(stderr, "[fseq] reduced phase C: the lists of %u blocks reach below what their representatives vouch for: those blocks again on all rows\n", cnt);
		if (cnt) { c->red_plan_valid = false; R.redo = true; R.redone += cnt; *overflow_out = false; return FSEQ_OK; }
	}
	if (keyspace)
	{
		if (!R.tree_ran) h_flags[5] = my_blocks;               // (no tree this time: every block went to the column sweep, as last time)
		c->tm.phase_a_fallbacks = h_flags[4];
		c->tm.phase_a_given_up = h_flags[5];
		// (the tree ran alone because no block was given up last time; the same input gives the same outcome)
		if (R.tree_alone && h_flags[5] != 0u) return fail(c, FSEQ_E_HIP, "internal: the key-space tree gave up blocks it ranked in the run before");
		c->bk_given_up = (int) h_flags[5];
		if (R.trie_ran)
		{
			if (R.trie_alone && h_flags[6] != 0u) return fail(c, FSEQ_E_HIP, "internal: the block trie gave up blocks it ranked in the run before");
			c->bt_given_up = (int) h_flags[6];
			c->tm.phase_a_trie_given_up = h_flags[6];
		}
	}
	{
		float f = 0;
		HIP_TRY(c, hipEventElapsedTime(&f, c->ev[3], c->ev[4])); ms_c += f;
		HIP_TRY(c, hipEventElapsedTime(&f, c->ev_dp[0], c->ev_dp[1])); ms_dp += f;
	}
#ifdef FSEQ_DP_STAMPS
	{
		unsigned long long stamps[96];
		HIP_TRY(c, hipMemcpy(stamps, c->d_flags + 8, sizeof(stamps), hipMemcpyDeviceToHost));
		for (int w = 0; w < 16; ++w)
		{
			unsigned long long const *q = stamps + 48 + 3 * w;
			double const nr = (double) (stamps[3 * w + 2] ? stamps[3 * w + 2] : 1);
			fprintf(stderr, "[dp stamps] wave %2d cycles/round: barrier 1 = %.0f, update = %.0f, barrier 2 = %.0f\n", w, q[0] / nr, q[1] / nr, q[2] / nr);
		}
		for (int w = 0; w < 16; ++w)
		{
			unsigned long long const *q = stamps + 3 * w;
			double const nr = (double) (q[2] ? q[2] : 1);
			fprintf(stderr, "[dp stamps] wave %2d rounds=%llu cycles/round: work=%.0f waits=%.0f\n", w, q[2], q[0] / nr, q[1] / nr);
		}
	}
#endif
#ifdef FSEQ_DP_STATS
	{
		uint32_t hist[34];
		HIP_TRY(c, hipMemcpy(hist, c->d_flags + 128, sizeof(hist), hipMemcpyDeviceToHost));
		fprintf(stderr, "[dp stats] list entries a cell needed (cell-pair path; last = more than 32):");
		for (int i = 0; i < 34; ++i) fprintf(stderr, " %u", hist[i]);
		fprintf(stderr, "\n");
	}
#endif
	range_cd.end();
	progress(c, FSEQ_STAGE_TRACEBACK, n, n);
	RangeScope range_tb("fseq traceback + find_segments_greedy");
	double const th0 = now_ms();
	bool overflow = (h_flags[0] & 1u) != 0 || spec_overflow != 0;
	c->tm.dp_sweeps = spec_sweeps;
	c->tm.dp_chunks = use_spec ? spec.nchunks() : 0u;

	if (!overflow && (rc = long_traceback_and_merge(c, R, th0, &overflow))) return rc;
	ms_host += now_ms() - th0;
	range_tb.end();
	if (!overflow) progress(c, FSEQ_STAGE_MERGE, c->traceback.size(), c->traceback.size());
	if (c->tune.debug) fprintf(stderr, "[fseq] host: traceback + merge %.3f ms\n", now_ms() - th0);
	*overflow_out = overflow;
	return FSEQ_OK;
}


// ---- [r5] pass 2 behind the reduced phase C: a boundary inside a block is ONE chain step from the block's boundary state
// (k_chain_snap), keyed by the classes the block's representatives form at that column (k_columns_red with the class
// tables as its output); a boundary on a block border is that border's state.  Blocks without representatives (more than a
// configuration holds) replay their columns on all rows from the block's start (k_colblock<MODE_SNAP>).
int long_pass2_reduced(fseq_ctx *c, LongRun &R)
{
	FSEQ_LONG_LOCALS(c);
	size_t const S2all = c->segments.size();
	if (!S2all) return FSEQ_OK;
	ChainSnapSet cs{};
	if (!c->use_stream)
	{
		if (!select_chain_snap(ks.T, ks.E, &cs)) return fail(c, FSEQ_E_UNSUPPORTED, "pass 2: no chain step for this configuration");
		HIP_TRY(c, cs.prepare());
	}
	c->snap_slot.assign(S2all, -1);
	// my tasks (sharded: a boundary belongs to the rank whose blocks hold the state in front of it), in the order of the boundaries
	std::vector<uint64_t> rbs;
	std::vector<uint32_t> task_blk, ncls0;
	// tasks of reduced blocks by configuration: workgroups {block, first task, count, start column}; tasks of the other blocks: old groups
	struct Wg { uint32_t blk, first, count, start; };
	std::vector<std::vector<Wg>> wgs((size_t) reduced_config_count());
	std::vector<uint64_t> o_rbs, o_srcs;
	std::vector<uint2> o_grp;
	std::vector<uint32_t> o_slot;
	uint64_t cells = 0;
	for (size_t si = 0; si < S2all; ++si)
	{
		uint64_t const rb = c->segments[si].rb;
		if (sharded)
		{
			uint32_t const owner = (uint32_t) std::min<uint64_t>(rb / ((uint64_t) sh.bpr * c->B), sh.active - 1u);
			if (owner != sh.rank) continue;
		}
		size_t const i = rbs.size();
		c->snap_slot[si] = (int64_t) i;
		rbs.push_back(rb);
		bool const border = rb % c->B == 0;
		uint32_t const blk = border ? (uint32_t) (rb / c->B) : (uint32_t) std::min<uint64_t>(rb / c->B, c->nblocks - 1u);
		task_blk.push_back(blk);
		ncls0.push_back(0u);                                       // a border: the copy; else the sweep fills it in
		if (border) continue;
		int const cf = blk < c->red_config_snap_of.size() ? c->red_config_snap_of[blk] : -1;
		if (cf >= 0 && c->red_cnt_host[blk] != RED_NONE)
		{
			// the sweep starts at the last state phase C dropped in front of the boundary (or at the block's first column)
			uint64_t start = c->red_ss_stride ? (rb - 1u) / c->red_ss_stride * c->red_ss_stride : 0u;
			// (a block that ran on all rows dropped none)
			if (start <= (uint64_t) blk * c->B || c->red_full[blk]) start = (uint64_t) blk * c->B;
			auto &v = wgs[(size_t) cf];
			if (!v.empty() && v.back().blk == blk && v.back().start == (uint32_t) start) ++v.back().count;
			else v.push_back(Wg{blk, (uint32_t) i, 1u, (uint32_t) start});
		}
		else
		{
			ncls0[i] = 0xFFFFFFFFu;                                // not this kernel's
			if (!o_srcs.empty() && o_srcs.back() == blk) ++o_grp.back().y;
			else { o_grp.push_back(make_uint2((uint32_t) o_rbs.size(), 1u)); o_srcs.push_back(blk); }
			o_rbs.push_back(rb); o_slot.push_back((uint32_t) i);
		}
	}
	size_t const S2 = rbs.size();
	auto finish = [&]() -> int {
		if (sharded)
		{
			// R of SURVEY.md 8(d) is the sum over the ranks: one slot pair per rank
			uint32_t slots[2] = {(uint32_t) R.pass2_cells, (uint32_t) (R.pass2_cells >> 32)};
			std::vector<uint32_t> all(2 * sh.world);
			HIP_TRY(c, hipMemsetAsync(sh.xbuf, 0, all.size() * 4, st));
			HIP_TRY(c, hipMemcpyAsync(sh.xbuf + 2 * sh.rank, slots, 8, hipMemcpyHostToDevice, st));
			int rc2;
			if ((rc2 = shard_exchange(c, all.size(), 0))) return rc2;
			HIP_TRY(c, hipMemcpy(all.data(), sh.xbuf, all.size() * 4, hipMemcpyDeviceToHost));
			R.pass2_cells = 0;
			for (uint32_t g = 0; g < sh.world; ++g) R.pass2_cells += (uint64_t) all[2 * g] | ((uint64_t) all[2 * g + 1] << 32);
			c->sh.closed = true;                                 // the last exchange of the run
		}
		return FSEQ_OK;
	};
	if (!S2) { R.pass2_cells = 0; return finish(); }
	if (c->snap_cap < S2)
	{
		if ((rc = dev_alloc(c, &c->d_snap_a, S2 * (size_t) m))) return rc;
		if ((rc = dev_alloc(c, &c->d_snap_d, S2 * (size_t) m))) return rc;
		c->snap_cap = S2;
	}
	if (c->red_task_cap < S2)
	{
		if ((rc = dev_alloc(c, &c->d_red_cls, S2 * (size_t) c->red_cap))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_headd, S2 * (size_t) c->red_cap))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_ncls, S2))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_taskblk, S2))) return rc;
		if ((rc = dev_alloc(c, &c->d_red_wgtasks, 4 * S2 + 64))) return rc;
		c->red_task_cap = S2;
	}
	if (c->cols_cap < S2) { if ((rc = dev_alloc(c, &c->d_cols, S2))) return rc; c->cols_cap = S2; }
	// the task lists through pinned memory of their own (live until the synchronisation behind the kernels)
	std::vector<uint32_t> hb, hw;
	std::vector<RedLaunch> ls;
	for (size_t cf = 0; cf < wgs.size(); ++cf)
	{
		if (wgs[cf].empty()) continue;
		ls.push_back(RedLaunch{(int) cf, (uint32_t) hb.size(), (uint32_t) wgs[cf].size()});
		for (auto const &w : wgs[cf])
		{
			hb.push_back(w.blk); hw.push_back(w.first); hw.push_back(w.count); hw.push_back(w.start);
			cells += (rbs[w.first + w.count - 1u] - (uint64_t) w.start) * c->red_cnt_host[w.blk];
		}
	}
	{
		size_t const need = S2 * 16 + hb.size() * 16 + 256;
		if (c->red_pin2_bytes < need)
		{
			if (c->h_red_pin2) (void) hipHostFree(c->h_red_pin2);
			c->h_red_pin2 = nullptr; c->red_pin2_bytes = 0;
			HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_red_pin2), need * 2, hipHostMallocDefault));
			c->red_pin2_bytes = need * 2;
		}
		uint8_t *pp = c->h_red_pin2;
		auto put = [&](void const *src, size_t bytes) { void *at = pp; memcpy(pp, src, bytes); pp += (bytes + 15) & ~size_t(15); return at; };
		HIP_TRY(c, hipMemcpyAsync(c->d_cols, put(rbs.data(), S2 * 8), S2 * 8, hipMemcpyHostToDevice, st));
		HIP_TRY(c, hipMemcpyAsync(c->d_red_taskblk, put(task_blk.data(), S2 * 4), S2 * 4, hipMemcpyHostToDevice, st));
		HIP_TRY(c, hipMemcpyAsync(c->d_red_ncls, put(ncls0.data(), S2 * 4), S2 * 4, hipMemcpyHostToDevice, st));
		if (!hb.empty())
		{
			HIP_TRY(c, hipMemcpyAsync(c->d_red_wgtasks + 3 * S2, put(hb.data(), hb.size() * 4), hb.size() * 4, hipMemcpyHostToDevice, st));
			HIP_TRY(c, hipMemcpyAsync(c->d_red_wgtasks, put(hw.data(), hw.size() * 4), hw.size() * 4, hipMemcpyHostToDevice, st));
		}
	}
	HIP_TRY(c, hipEventRecord(c->ev[6], st));
	progress(c, FSEQ_STAGE_SAMPLES, 0, S2);
	RangeScope range_p2("fseq pass 2: boundary states (update_pbwt_task)");
	// the class tables at the task columns, configuration by configuration
	if (!ls.empty())
	{
		RedArgs RA;
		red_fill_args(c, RA);
		RA.task_rb = reinterpret_cast<unsigned long long const *>(c->d_cols);
		RA.cls = c->d_red_cls; RA.headd = c->d_red_headd; RA.ncls = c->d_red_ncls;
		std::stable_sort(ls.begin(), ls.end(), [](RedLaunch const &x, RedLaunch const &y) { return x.count > y.count; });
		if ((rc = red_launch_all(c, ls, RA, c->d_red_wgtasks + 3 * S2, c->d_red_wgtasks, (uint2 *) nullptr, (uint4 *) nullptr, 0u, 0u))) return rc;
	}
	// one chain step per boundary (a copy for the borders)
	if (!c->use_stream)
		cs.launch(st, (uint32_t) S2, cs.lds, c->d_bstate_a, c->d_bstate_d, c->d_rank, m, c->d_red_taskblk, c->d_red_cls, c->d_red_headd, c->d_red_ncls, c->red_cap,
		          c->d_snap_a, c->d_snap_d, scan_keyed(c));
	else
	{
		// streamed rows: the step as a radix sort + range maxima in a workspace per workgroup (fseq_chainsort.hpp), the workgroups
		// take the tasks in turn
		int ncu = 0;
		(void) hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->p.device);
		size_t const fit = c->ws_words / chainsort_ws_words(m);
		uint32_t const grid = (uint32_t) std::min<size_t>(std::min<size_t>(S2, fit), (size_t) std::max(ncu, 1) * 2u);
		if (!grid) return fail(c, FSEQ_E_OOM, "pass 2: the workspace holds no chain step");
		hipLaunchKernelGGL(k_chain_snap_stream, dim3(grid), dim3(ST), chainsort_lds_bytes(), st, c->d_bstate_a, c->d_bstate_d, c->d_rank, m, c->d_red_taskblk, c->d_red_cls,
		                   c->d_red_headd, c->d_red_ncls, c->red_cap, (uint32_t) S2, c->d_snap_a, c->d_snap_d, c->d_ws);
	}
	for (size_t i = 0; i < S2; ++i)
		if (ncls0[i] == 0u && rbs[i] % c->B != 0) cells += (uint64_t) m * 4u;      // (a step is ~4 digit passes over the rows)
	// the boundaries of blocks without representatives: their columns on all rows from the block's start
	if (!o_grp.empty())
	{
		size_t const So = o_rbs.size();
		uint32_t *tmp_a = nullptr, *tmp_d = nullptr;
		uint64_t *d_orb = nullptr, *d_osrc = nullptr;
		uint2 *d_ogrp = nullptr;
		if ((rc = dev_alloc(c, &tmp_a, So * (size_t) m))) return rc;
		if ((rc = dev_alloc(c, &tmp_d, So * (size_t) m))) { dev_free(c, &tmp_a); return rc; }
		if ((rc = dev_alloc(c, &d_orb, So)) || (rc = dev_alloc(c, &d_osrc, o_srcs.size())) || (rc = dev_alloc(c, &d_ogrp, o_grp.size())))
		{ dev_free(c, &tmp_a); dev_free(c, &tmp_d); dev_free(c, &d_orb); dev_free(c, &d_osrc); dev_free(c, &d_ogrp); return rc; }
		HIP_TRY(c, hipMemcpyAsync(d_orb, o_rbs.data(), So * 8, hipMemcpyHostToDevice, st));
		HIP_TRY(c, hipMemcpyAsync(d_osrc, o_srcs.data(), o_srcs.size() * 8, hipMemcpyHostToDevice, st));
		HIP_TRY(c, hipMemcpyAsync(d_ogrp, o_grp.data(), o_grp.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
		if (!c->use_stream)
			ks.snap(st, (uint32_t) o_grp.size(), ks.lds_snap, c->d_msa, c->ld, m, n, c->B, c->nblocks, c->npass, c->bsh, c->d_bstate_a, c->d_bstate_d, d_orb, d_ogrp,
			        tmp_a, tmp_d, d_osrc, c->snap_stride, (uint32_t const *) nullptr, (uint32_t const *) nullptr, scan_keyed(c));
		else
		{
			// (the streamed sweep needs 4m workspace words per workgroup: as many groups per launch as d_ws holds)
			size_t const capg = std::max<size_t>(1, c->ws_words / (4 * (size_t) m));
			for (size_t g0 = 0; g0 < o_grp.size(); g0 += capg)
			{
				size_t const cntg = std::min(capg, o_grp.size() - g0);
				hipLaunchKernelGGL((stream_keyed(c) ? k_colblock_stream<MODE_SNAP, true> : k_colblock_stream<MODE_SNAP, false>), dim3((uint32_t) cntg), dim3(ST), stream_lds_bytes(sym_bytes(m, c->bsh), c->stream_staged), st, c->d_msa, c->ld, m, n, c->B,
				                   c->nblocks, c->npass, c->bsh, c->d_ws, (uint32_t) c->stream_staged, (uint32_t *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, c->d_bstate_a, c->d_bstate_d,
				                   d_orb, d_ogrp + g0, tmp_a, tmp_d, d_osrc + g0, c->snap_stride, (uint32_t *) nullptr, (uint32_t *) nullptr, (uint64_t) 0, 0u);
			}
		}
		for (size_t j = 0; j < So; ++j)
		{
			HIP_TRY(c, hipMemcpyAsync(c->d_snap_a + (size_t) o_slot[j] * m, tmp_a + j * (size_t) m, (size_t) m * 4, hipMemcpyDeviceToDevice, st));
			HIP_TRY(c, hipMemcpyAsync(c->d_snap_d + (size_t) o_slot[j] * m, tmp_d + j * (size_t) m, (size_t) m * 4, hipMemcpyDeviceToDevice, st));
		}
		for (size_t g = 0; g < o_grp.size(); ++g) cells += (o_rbs[o_grp[g].x + o_grp[g].y - 1] - o_srcs[g] * c->B) * m;
		HIP_TRY(c, hipStreamSynchronize(st));
		dev_free(c, &tmp_a); dev_free(c, &tmp_d); dev_free(c, &d_orb); dev_free(c, &d_osrc); dev_free(c, &d_ogrp);
	}
	HIP_TRY(c, hipEventRecord(c->ev[7], st));
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(st));
	range_p2.end();
	progress(c, FSEQ_STAGE_SAMPLES, S2, S2);
	float f = 0;
	HIP_TRY(c, hipEventElapsedTime(&f, c->ev[6], c->ev[7])); R.ms_p2 = f;
	R.pass2_cells = cells;
	return finish();
}

// ---- pass 2: (a, d) at the merged boundaries
int long_pass2(fseq_ctx *c, LongRun &R)
{
	FSEQ_LONG_LOCALS(c);
	if (c->red_active) return long_pass2_reduced(c, R);
	uint64_t &pass2_cells = R.pass2_cells;
	double &ms_p2 = R.ms_p2;
	size_t const S2 = c->segments.size();
	// ---- pass 2: (a,d) at the merged boundaries (update_pbwt_task.cc:13-35)
	if (S2)
	{
		if (c->cols_cap < S2) { if ((rc = dev_alloc(c, &c->d_cols, S2))) return rc; c->cols_cap = S2; }
		// every boundary starts from the nearest exact state at or below it: a block boundary state
		// (phase B) or one of the states phase C dropped every snap_stride columns; boundaries that share
		// a start state share one sweep (boundaries ascending)
		// sharded: a boundary belongs to the rank whose blocks hold the state in front of it
		std::vector<uint64_t> rbs, srcs;
		std::vector<uint2> grp;
		uint64_t const sstr = c->snap_stride;
		std::vector<uint64_t> starts;
		std::vector<uint32_t> grp_blk;                            // block whose columns a group replays
		c->snap_slot.assign(S2, -1);
		for (size_t i = 0; i < S2; ++i)
		{
			uint64_t const rb = c->segments[i].rb;
			if (sharded)
			{
				uint32_t const owner = (uint32_t) std::min<uint64_t>(rb / ((uint64_t) sh.bpr * c->B), sh.active - 1u);
				if (owner != sh.rank) continue;
			}
			c->snap_slot[i] = (int64_t) rbs.size();
			rbs.push_back(rb);
			// (states in id form belong to the block that made them: the boundary behind the last column starts inside the last block)
			uint64_t const blk = std::min<uint64_t>(rb / c->B, c->ss_ids ? c->nblocks - 1u : c->nblocks);
			uint64_t const q = rb / sstr;
			uint64_t src = blk, p0 = blk * c->B;
			if (c->d_ss_a && q >= 1 && q * sstr > p0) { src = q | (1ull << 63); p0 = q * sstr; }
			if (grp.empty() || srcs.back() != src) { grp.push_back(make_uint2((uint32_t) (rbs.size() - 1), 1u)); srcs.push_back(src); starts.push_back(p0); grp_blk.push_back((uint32_t) blk); }
			else ++grp.back().y;
		}
		for (size_t g = 0; g < grp.size(); ++g)
			pass2_cells += (rbs[grp[g].x + grp[g].y - 1] - starts[g]) * m;
		size_t const S2m = rbs.size();                            // boundaries that are mine (all of them when not sharded)
		if (c->snap_cap < S2m)
		{
			if ((rc = dev_alloc(c, &c->d_snap_a, S2m * (size_t) m))) return rc;
			if ((rc = dev_alloc(c, &c->d_snap_d, S2m * (size_t) m))) return rc;
			c->snap_cap = S2m;
		}
		if (c->src_cap < srcs.size()) { if ((rc = dev_alloc(c, &c->d_src, srcs.size()))) return rc; c->src_cap = srcs.size(); }
		if (c->grp_cap < grp.size()) { if ((rc = dev_alloc(c, &c->d_grp, grp.size()))) return rc; c->grp_cap = grp.size(); }
		{
			// (through the pinned stage: it stays untouched until the synchronisation behind the kernel)
			if ((rc = pin_reserve(c, (srcs.size() + S2m + grp.size()) * 8 + 256))) return rc;
			uint64_t *const psrc = pin_take<uint64_t>(c, srcs.size());
			uint64_t *const prb = pin_take<uint64_t>(c, S2m);
			uint2 *const pgrp = pin_take<uint2>(c, grp.size());
			std::copy(srcs.begin(), srcs.end(), psrc);
			std::copy(rbs.begin(), rbs.end(), prb);
			std::copy(grp.begin(), grp.end(), pgrp);
			if (!srcs.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_src, psrc, srcs.size() * 8, hipMemcpyHostToDevice, st));
			if (S2m) HIP_TRY(c, hipMemcpyAsync(c->d_cols, prb, S2m * 8, hipMemcpyHostToDevice, st));
			if (!grp.empty()) HIP_TRY(c, hipMemcpyAsync(c->d_grp, pgrp, grp.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
		}
		HIP_TRY(c, hipEventRecord(c->ev[6], st));
		progress(c, FSEQ_STAGE_SAMPLES, 0, S2);
		RangeScope range_p2("fseq pass 2: boundary states (update_pbwt_task)");
		if (grp.empty())
		{
		}
		else if (c->use_stream && c->ss_ids)
		{
			// pass 2 on phase C's tile step (fseq_stream2.hpp, S2_SNAP): one workgroup per block that has boundaries, the block's
			// groups one after the other in the block's own workspace (where V and D0 of its id space still are)
			std::vector<uint32_t> wgb;
			std::vector<uint2> wgg;
			std::vector<uint64_t> wgw;                             // columns a workgroup replays
			for (size_t g = 0; g < grp.size(); ++g)
			{
				if (wgb.empty() || wgb.back() != grp_blk[g]) { wgb.push_back(grp_blk[g]); wgg.push_back(make_uint2((uint32_t) g, 1u)); wgw.push_back(0); }
				else ++wgg.back().y;
				wgw.back() += rbs[grp[g].x + grp[g].y - 1] - starts[g] + 2;      // (+ the loads of the start state and the snapshots)
			}
			{
				// the longest first: the workgroups are handed out in launch order, and a long one at the end would run alone
				std::vector<uint32_t> order(wgb.size());
				for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
				std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return wgw[x] > wgw[y]; });
				std::vector<uint32_t> b2(wgb.size());
				std::vector<uint2> g2(wgg.size());
				for (size_t i = 0; i < order.size(); ++i) { b2[i] = wgb[order[i]]; g2[i] = wgg[order[i]]; }
				wgb.swap(b2); wgg.swap(g2);
			}
			if (c->wg_cap < wgb.size())
			{
				if ((rc = dev_alloc(c, &c->d_wgblk, wgb.size()))) return rc;
				if ((rc = dev_alloc(c, &c->d_wggrp, wgb.size()))) return rc;
				c->wg_cap = wgb.size();
			}
			// (pageable sources: the runtime stages them before the call returns)
			HIP_TRY(c, hipMemcpyAsync(c->d_wgblk, wgb.data(), wgb.size() * 4, hipMemcpyHostToDevice, st));
			HIP_TRY(c, hipMemcpyAsync(c->d_wggrp, wgg.data(), wgg.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
			S2SnapArgs SN;
			SN.wg_block = c->d_wgblk; SN.wg_groups = c->d_wggrp; SN.grp_tasks = c->d_grp; SN.grp_src = c->d_src; SN.task_rb = c->d_cols;
			SN.snap_a = c->d_snap_a; SN.snap_d = c->d_snap_d; SN.bs_w = c->d_bs_w; SN.bs_h = c->d_bs_h;
			c->s2.launch_snap(st, (uint32_t) wgb.size(), c->s2_lds, c->d_msa, c->ld, m, n, c->B, c->npass, c->bsh, c->d_ws_c, c->snap_stride, c->d_ss_a, c->d_ss_d, SN);
			HIP_TRY(c, hipStreamSynchronize(st));                 // (wgb / wgg must outlive their copies)
		}
		else if (c->use_stream)
		{
			// the streamed sweep needs 4m workspace words per workgroup: as many groups per launch as d_ws holds
			size_t const cap = std::max<size_t>(1, c->ws_words / (4 * (size_t) m));
			for (size_t g0 = 0; g0 < grp.size(); g0 += cap)
			{
				size_t const cnt = std::min(cap, grp.size() - g0);
				hipLaunchKernelGGL((stream_keyed(c) ? k_colblock_stream<MODE_SNAP, true> : k_colblock_stream<MODE_SNAP, false>), dim3((uint32_t) cnt), dim3(ST), stream_lds_bytes(sym_bytes(m, c->bsh), c->stream_staged), st, c->d_msa, c->ld, m, n, c->B,
				                   c->nblocks, c->npass, c->bsh, c->d_ws, (uint32_t) c->stream_staged, (uint32_t *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, c->d_bstate_a, c->d_bstate_d,
				                   c->d_cols, c->d_grp + g0, c->d_snap_a, c->d_snap_d, c->d_src + g0, c->snap_stride, c->d_ss_a, c->d_ss_d, (uint64_t) 0, c->ss_pack);
			}
		}
		else
			ks.snap(st, (uint32_t) grp.size(), ks.lds_snap, c->d_msa, c->ld, m, n, c->B, c->nblocks, c->npass, c->bsh, c->d_bstate_a, c->d_bstate_d, c->d_cols, c->d_grp,
			        c->d_snap_a, c->d_snap_d, c->d_src, c->snap_stride, c->d_ss_a, c->d_ss_d, scan_keyed(c));
		HIP_TRY(c, hipEventRecord(c->ev[7], st));
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipStreamSynchronize(st));
		range_p2.end();
		progress(c, FSEQ_STAGE_SAMPLES, S2, S2);
		float f = 0;
		HIP_TRY(c, hipEventElapsedTime(&f, c->ev[6], c->ev[7])); ms_p2 = f;
		if (sharded)
		{
			// R of SURVEY.md 8(d) is the sum over the ranks: one slot pair per rank
			uint32_t slots[2] = {(uint32_t) pass2_cells, (uint32_t) (pass2_cells >> 32)};
			std::vector<uint32_t> all(2 * sh.world);
			HIP_TRY(c, hipMemsetAsync(sh.xbuf, 0, all.size() * 4, st));
			HIP_TRY(c, hipMemcpyAsync(sh.xbuf + 2 * sh.rank, slots, 8, hipMemcpyHostToDevice, st));
			if ((rc = shard_exchange(c, all.size(), 0))) return rc;
			HIP_TRY(c, hipMemcpy(all.data(), sh.xbuf, all.size() * 4, hipMemcpyDeviceToHost));
			pass2_cells = 0;
			for (uint32_t g = 0; g < sh.world; ++g) pass2_cells += (uint64_t) all[2 * g] | ((uint64_t) all[2 * g + 1] << 32);
			c->sh.closed = true;                                 // the last exchange of the run
		}
	}

	return FSEQ_OK;
}

int run_long_path(fseq_ctx *c, fseq_result *res)
{
	FSEQ_LONG_LOCALS(c);
	LongRun R;
	R.X = p.list_cap ? p.list_cap : std::max(FSEQ_X_FLOOR, c->X_hint);
	c->tm = fseq_timings{};
	c->tm.block_len = c->B;
	c->tm.n_blocks = c->nblocks;
	double const t_begin = now_ms();

	// (FSEQ_DEBUG: where the host's wall time of a run goes -- a first run on a context allocates, loads code objects, plans)
	auto mark = [&](char const *what) { if (c->tune.debug) fprintf(stderr, "[fseq] +%.3f ms %s\n", now_ms() - t_begin, what); };
	if ((rc = ensure_work_buffers(c, 0))) return rc;
	mark("work buffers");
	auto close_ab = [&](int code) { if (R.range_ab_open) { FSEQ_RANGE_POP(); R.range_ab_open = false; } return code; };
	if ((rc = long_phase_a(c, R))) return close_ab(rc);
	mark("phase A queued");
	if ((rc = long_phase_b(c, R))) return close_ab(rc);
	mark("phase B queued");
	if ((rc = long_list_capacity(c, R))) return rc;
	mark("list capacity");
	while (true)
	{
		bool overflow = false;
		if ((rc = long_attempt(c, R, &overflow))) return rc;
		mark("attempt done");
		if (R.redo) continue;                  // (the same capacity; the blocks that were flagged run on all rows now)
		// (sharded: the thresholds are the same on every rank, so every rank takes the same way here)
		if (!overflow) break;
		if (R.X >= m) return fail(c, FSEQ_E_HIP, "internal: divergence lists complete but DP flagged overflow");
		R.X = (uint32_t) std::min<uint64_t>(m, (uint64_t) R.X * 2 + 1);
		++R.retries;
		if (c->tune.debug) fprintf(stderr, "[fseq] divergence lists too short, retry %u with X = %u\n", R.retries, R.X);
	}
	c->X_hint = R.X;                         // later runs on this context start with the capacity that worked
	c->res.segment_count = c->segments.size();
	if ((rc = long_pass2(c, R))) return rc;
	mark("pass 2 done");
	uint32_t const X = R.X, retries = R.retries;
	double const ms_c = R.ms_c, ms_dp = R.ms_dp, ms_host = R.ms_host, ms_p2 = R.ms_p2;
	uint64_t const pass2_cells = R.pass2_cells;
	size_t const S2 = c->segments.size();
	{
		float f = 0;
		HIP_TRY(c, hipEventElapsedTime(&f, c->ev[0], c->ev[1])); c->tm.ms_phase_a = f;
		HIP_TRY(c, hipEventElapsedTime(&f, c->ev[1], c->ev[2])); c->tm.ms_phase_b = f;
	}
	c->tm.ms_phase_c = ms_c;
	c->tm.ms_dp = ms_dp;
	c->tm.ms_pass2 = ms_p2;
	c->tm.ms_host = ms_host;
	c->tm.ms_colstep_kernels = c->tm.ms_phase_a + ms_c + ms_p2;
	c->tm.colstep_launches = 2 + retries + (S2 ? 1 : 0);
	c->tm.colstep_cells = (uint64_t) m * n * (2 + retries) + pass2_cells;
	c->tm.pass2_cells = pass2_cells;
	c->tm.list_cap_used = X;
	c->tm.retries = retries;
	c->tm.reduced_redone = R.redone;
	if (!c->red_active) { c->tm.reduced_blocks = 0; c->tm.reduced_rows_mean = 0; }
	c->tm.ms_total = now_ms() - t_begin;
	c->have_result = true;
	*res = c->res;
	if (!(c->res.max_segment_size < m))
		return fail(c, FSEQ_E_NO_REDUCTION, "Unable to reduce the number of sequences; the maximum segment size is equal to the number of input sequences.");
	return FSEQ_OK;
}

// segmentation_sp_context::process (segmentation_sp_context.cc:21-28): one sweep over all n columns
// from the identity; the distinct rows are the block keys of a single block [0, n).
int run_short_path(fseq_ctx *c, fseq_result *res)
{
	fseq_params const &p = c->p;
	uint32_t const m = p.m;
	hipStream_t st = c->stream;
	int rc;
	c->tm = fseq_timings{};
	double const t_begin = now_ms();
	uint32_t *d_rank = nullptr, *d_keyd = nullptr, *d_nk = nullptr;
	if ((rc = dev_alloc(c, &d_rank, m))) return rc;
	if ((rc = dev_alloc(c, &d_keyd, m))) { dev_free(c, &d_rank); return rc; }
	if ((rc = dev_alloc(c, &d_nk, 4))) { dev_free(c, &d_rank); dev_free(c, &d_keyd); return rc; }
	if (c->use_stream && !c->d_ws)
	{
		c->ws_words = (size_t) 4 * m;
		if ((rc = dev_alloc(c, &c->d_ws, c->ws_words))) { dev_free(c, &d_rank); dev_free(c, &d_keyd); dev_free(c, &d_nk); return rc; }
		c->d_ws_c = c->d_ws;
	}
	// one block [0, n): ranked in key space (fseq_blockkeys.hpp); FSEQ_PHASE_A_CLASSIC: the per-column sweep
	if (c->bk_cap_words && !c->tune.phase_a_classic)
	{
		if (c->use_stream)
		{
			size_t const per = (blockkeys_stream_ws_words(m, (uint32_t) p.n, c->bsh) + 15) & ~size_t(15);
			if (c->bkws_words < per)
			{
				if ((rc = dev_alloc(c, &c->d_bkws, per))) { dev_free(c, &d_rank); dev_free(c, &d_keyd); dev_free(c, &d_nk); return rc; }
				c->bkws_words = per;
			}
			hipLaunchKernelGGL(k_blockkeys_stream, dim3(1), dim3(1024), c->bk_lds, st, c->d_msa, c->ld, m, p.n, (uint32_t) p.n, c->bsh, 1u,
			                   d_rank, d_keyd, d_nk, (uint64_t) 0, c->d_bkws, per, c->bk_cap_words, (uint32_t *) nullptr, (c->tune.blockkeys_wide ? 1u : 0u) | (c->tune.blockkeys_single ? 2u : 0u), (uint32_t *) nullptr);
		}
		else
		{
			size_t const per = (blockkeys_scratch_halfwords(m, (uint32_t) p.n, c->bsh) + 7) & ~size_t(7);
			if (c->bk_per_block != per || c->bk_blocks < 1)
			{
				if ((rc = dev_alloc(c, &c->d_bk, per))) { dev_free(c, &d_rank); dev_free(c, &d_keyd); dev_free(c, &d_nk); return rc; }
				c->bk_per_block = per; c->bk_blocks = 1;
			}
			launch_blockkeys(c->bk_T, st, 1, c->bk_lds, c->d_msa, c->ld, m, p.n, (uint32_t) p.n, c->bsh, d_rank, d_keyd, d_nk, 0, c->d_bk, per, c->bk_cap_words, nullptr, nullptr);
		}
	}
	else
	{
		// the 16-bit LDS kernels keep block-relative divergences in 16 bits: one block of 65536 columns or more would wrap
		if (!c->use_stream && c->ks.cap > 7168u && p.n > 65535u)
		{
			dev_free(c, &d_rank); dev_free(c, &d_keyd); dev_free(c, &d_nk);
			return fail(c, FSEQ_E_UNSUPPORTED, "short path by column sweep: more than 65535 columns with 16-bit LDS state (unset FSEQ_PHASE_A_CLASSIC)");
		}
		launch_rank(c, 1, (uint32_t) p.n, 1, d_rank, d_keyd, d_nk);
	}
	std::vector<uint32_t> rank(m);
	uint32_t nk = 0;
	hipError_t e1 = hipMemcpyAsync(rank.data(), d_rank, (size_t) m * 4, hipMemcpyDeviceToHost, st);
	hipError_t e2 = hipMemcpyAsync(&nk, d_nk, 4, hipMemcpyDeviceToHost, st);
	hipError_t e3 = hipStreamSynchronize(st);
	dev_free(c, &d_rank); dev_free(c, &d_keyd); dev_free(c, &d_nk);
	if (e1 != hipSuccess) return fail(c, FSEQ_E_HIP, "short path copy", e1);
	if (e2 != hipSuccess) return fail(c, FSEQ_E_HIP, "short path copy", e2);
	if (e3 != hipSuccess) return fail(c, FSEQ_E_HIP, "short path sync", e3);
	// identical rows keep ascending row-id order in the pBWT, so a run's first row is its smallest id
	c->sp_first.assign(nk, 0xFFFFFFFFu);
	c->sp_len.assign(nk, 0);
	for (uint32_t r = 0; r < m; ++r)
	{
		uint32_t const k = rank[r];
		if (c->sp_first[k] == 0xFFFFFFFFu) c->sp_first[k] = r;
		++c->sp_len[k];
	}
	c->res = fseq_result{};
	c->res.max_segment_size = nk;
	c->res.short_path = 1;
	c->traceback.clear();
	c->segments.clear();
	c->tm.colstep_launches = 1;
	c->tm.colstep_cells = (uint64_t) m * p.n;
	c->tm.ms_total = now_ms() - t_begin;
	c->have_result = true;
	*res = c->res;
	if (!(nk < m))
		return fail(c, FSEQ_E_NO_REDUCTION, "Unable to reduce the number of sequences; the maximum segment size is equal to the number of input sequences.");
	return FSEQ_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

uint32_t fseq_abi_version(void) { return FSEQ_ABI_VERSION; }

char const *fseq_strerror(int code)
{
	switch (code)
	{
		case FSEQ_OK: return "ok";
		case FSEQ_E_ARG: return "bad argument";
		case FSEQ_E_NO_REDUCTION: return "unable to reduce the number of sequences";
		case FSEQ_E_HIP: return "HIP runtime error";
		case FSEQ_E_OOM: return "out of device memory";
		case FSEQ_E_UNSUPPORTED: return "unsupported shape for this build";
		case FSEQ_E_PEER: return "another rank of the sharded run failed";
		default: return "unknown";
	}
}

int fseq_create(fseq_params const *params, fseq_ctx **out)
{
	if (!params || !out) return FSEQ_E_ARG;
	*out = nullptr;
	if (0 == params->m || 0 == params->n || 0 == params->segment_length) return FSEQ_E_ARG;
	if (params->n >= 0xFFFFFFF0ull) return FSEQ_E_ARG;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FSEQ_E_HIP;
	if (params->device < 0 || params->device >= ndev) return FSEQ_E_ARG;
	if (hipSetDevice(params->device) != hipSuccess) return FSEQ_E_HIP;
	fseq_ctx *c = new fseq_ctx();
	c->p = *params;
	c->tune.from_environment();                  // the only look at the environment: fseq_debug_set_tuning changes a knob afterwards
	if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { fseq_destroy(c); return FSEQ_E_HIP; }
	if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { fseq_destroy(c); return FSEQ_E_HIP; }
	for (auto &e : c->ev_part)
		if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { fseq_destroy(c); return FSEQ_E_HIP; }
	for (auto &e : c->ev_dp)
		if (hipEventCreate(&e) != hipSuccess) { fseq_destroy(c); return FSEQ_E_HIP; }
	for (auto &e : c->ev)
		if (hipEventCreate(&e) != hipSuccess) { fseq_destroy(c); return FSEQ_E_HIP; }
	*out = c;
	return FSEQ_OK;
}

void fseq_destroy(fseq_ctx *c)
{
	if (!c) return;
	(void) hipSetDevice(c->p.device);
	if (c->stream) (void) hipStreamSynchronize(c->stream);
	free_msa(c);
	free_work(c);
	for (auto &e : c->ev) if (e) (void) hipEventDestroy(e);
	for (auto &e : c->ev_part) if (e) (void) hipEventDestroy(e);
	for (auto &e : c->ev_dp) if (e) (void) hipEventDestroy(e);
	if (c->h_pin) (void) hipHostFree(c->h_pin);
	if (c->h_red_pin) (void) hipHostFree(c->h_red_pin);
	if (c->h_red_pin2) (void) hipHostFree(c->h_red_pin2);
	for (auto &s_ : c->red_st) if (s_) (void) hipStreamDestroy(s_);
	for (auto &e_ : c->red_ev) if (e_) (void) hipEventDestroy(e_);
	if (c->stream2) (void) hipStreamDestroy(c->stream2);
	if (c->stream) (void) hipStreamDestroy(c->stream);
	delete c;
}

char const *fseq_last_error(fseq_ctx const *c) { return c ? c->err.c_str() : "null context"; }

int fseq_set_matrix(fseq_ctx *c, uint8_t const *base, size_t row_stride, size_t col_stride)
{
	if (!c || !base) return FSEQ_E_ARG;
	(void) hipSetDevice(c->p.device);
	if (1 == col_stride)
	{
		// row-major view: the same device path as fseq_set_rows
		std::vector<uint8_t const *> rows(c->p.m);
		for (uint32_t r = 0; r < c->p.m; ++r) rows[r] = base + (size_t) r * row_stride;
		return upload_rows_device(c, rows.data());
	}
	return set_alphabet_and_upload(c, base, row_stride, col_stride);     // any other layout: host encode + transpose
}

int fseq_set_rows(fseq_ctx *c, uint8_t const *const *rows)
{
	if (!c || !rows) return FSEQ_E_ARG;
	(void) hipSetDevice(c->p.device);
	for (uint32_t r = 0; r < c->p.m; ++r)
		if (!rows[r]) return fail(c, FSEQ_E_ARG, "null row pointer");
	return upload_rows_device(c, rows);
}

// borrowed columns: every code must be < sigma (the owned upload paths build the code table themselves)
static int check_borrowed_codes(fseq_ctx *c)
{
	if (c->sigma >= (1u << (8u >> c->bsh))) return FSEQ_OK;          // every code the width can hold is allowed
	uint64_t const ncols = held_hi(c) - held_lo(c);
	if (!ncols) return FSEQ_OK;
	uint32_t *d_mx = nullptr;
	int rc = dev_alloc(c, &d_mx, 1);
	if (rc) return rc;
	uint32_t mx = 0;
	uint32_t const col_bytes = sym_bytes(c->p.m, c->bsh), tail = c->p.m & ((1u << c->bsh) - 1u);
	hipError_t e = hipMemsetAsync(d_mx, 0, 4, c->stream);
	if (e == hipSuccess)
	{
		hipLaunchKernelGGL(k_max_code, dim3((uint32_t) std::min<uint64_t>(ncols, 4096)), dim3(256), 0, c->stream,
		                   c->d_msa + held_lo(c) * c->ld, c->ld, col_bytes, ncols, c->bsh, tail, d_mx);
		e = hipMemcpyAsync(&mx, d_mx, 4, hipMemcpyDeviceToHost, c->stream);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	dev_free(c, &d_mx);
	if (e != hipSuccess) return fail(c, FSEQ_E_HIP, "checking the borrowed columns", e);
	if (mx >= c->sigma)
	{
		c->have_input = false;
		char what[128];
		snprintf(what, sizeof(what), "borrowed device columns hold the code %u but sigma is %u", mx, c->sigma);
		return fail(c, FSEQ_E_ARG, what);
	}
	return FSEQ_OK;
}

int fseq_set_device_columns(fseq_ctx *c, void const *d_codes, size_t ld, uint32_t sigma)
{
	if (!c || !d_codes) return FSEQ_E_ARG;
	if (ld < c->p.m || (ld & 15) || (reinterpret_cast<uintptr_t>(d_codes) & 15))
		return fail(c, FSEQ_E_ARG, "device columns: ld must be >= m and a multiple of 16, base 16-byte aligned");
	if (sigma == 0 || sigma > 256) return fail(c, FSEQ_E_ARG, "sigma out of range");
	free_msa(c);
	c->d_msa = const_cast<uint8_t *>(static_cast<uint8_t const *>(d_codes)) - held_lo(c) * ld;
	c->ld = ld;
	c->bsh = 0;                              // borrowed columns are one code per byte
	c->own_msa = false;
	c->sigma = sigma;
	for (uint32_t i = 0; i < 256; ++i) c->code_to_byte[i] = (uint8_t) i;
	c->have_input = true;
	c->have_result = false;
	c->kernels_ready = false;
	c->X_hint = 0;
	c->bk_given_up = -1; c->bt_given_up = -1; c->colmask_ready = false; c->shard_dp_full_sticky = false; c->red_force_full.clear(); c->red_plan_valid = false; c->red_declined = false;
	(void) hipSetDevice(c->p.device);
	return check_borrowed_codes(c);
}

int fseq_set_device_columns_packed(fseq_ctx *c, void const *d_packed, size_t ld_bytes, uint32_t sigma, uint32_t bits)
{
	if (!c || !d_packed) return FSEQ_E_ARG;
	if (bits != 2 && bits != 4 && bits != 8) return fail(c, FSEQ_E_ARG, "packed device columns: bits must be 2, 4 or 8");
	uint32_t const bsh = bits == 2 ? 2u : bits == 4 ? 1u : 0u;
	if (sigma == 0 || sigma > (1u << bits)) return fail(c, FSEQ_E_ARG, "sigma does not fit the code width");
	if (ld_bytes < sym_bytes(c->p.m, bsh) || (ld_bytes & 15) || (reinterpret_cast<uintptr_t>(d_packed) & 15))
		return fail(c, FSEQ_E_ARG, "packed device columns: ld_bytes must cover a column and be a multiple of 16, base 16-byte aligned");
	free_msa(c);
	c->d_msa = const_cast<uint8_t *>(static_cast<uint8_t const *>(d_packed)) - held_lo(c) * ld_bytes;
	c->ld = ld_bytes;
	c->bsh = bsh;
	c->own_msa = false;
	c->sigma = sigma;
	for (uint32_t i = 0; i < 256; ++i) c->code_to_byte[i] = (uint8_t) i;
	c->have_input = true;
	c->have_result = false;
	c->kernels_ready = false;
	c->X_hint = 0;
	c->bk_given_up = -1; c->bt_given_up = -1; c->colmask_ready = false; c->shard_dp_full_sticky = false; c->red_force_full.clear(); c->red_plan_valid = false; c->red_declined = false;
	(void) hipSetDevice(c->p.device);
	return check_borrowed_codes(c);
}

uint64_t fseq_shard_xbuf_words(fseq_ctx const *c, uint32_t world)
{
	if (!c || 0 == world) return 0;
	// the largest exchanges: a whole DP array (keys, then lb, then size, one at a time), the hyper key blocks of
	// phase B (world x (2m + 1) words), the merge thresholds (2 words per traceback boundary <= n / L + 1)
	uint64_t const dp = c->p.n >= c->p.segment_length ? c->p.n - c->p.segment_length + 1 : 1;
	uint64_t const keys = (uint64_t) world * (2ull * c->p.m + 1);
	uint64_t const tb = 4 * (c->p.n / std::max<uint64_t>(1, c->p.segment_length) + 2) + 2;     // the gathered traceback entries (16 bytes each)
	return std::max<uint64_t>(std::max(std::max(dp, tb), keys), 1024) + 64;
}

int fseq_set_shard(fseq_ctx *c, uint32_t rank, uint32_t world, void *xbuf_device, uint64_t xbuf_words, fseq_allreduce_fn fn, void *user)
{
	if (!c || 0 == world || rank >= world) return FSEQ_E_ARG;
	if (c->have_input) return fail(c, FSEQ_E_ARG, "fseq_set_shard must be called before the input is set");
	if (1 == world) { c->sh = Shard{}; return FSEQ_OK; }
	if (!xbuf_device || !fn) return fail(c, FSEQ_E_ARG, "sharded run: exchange buffer and all-reduce function needed");
	if (c->p.n < 2 * c->p.segment_length) return fail(c, FSEQ_E_UNSUPPORTED, "the short path (n < 2L) is one sweep and does not shard");
	Shard sh;
	sh.on = true; sh.rank = rank; sh.world = world;
	sh.xbuf = static_cast<uint32_t *>(xbuf_device); sh.xwords = xbuf_words; sh.fn = fn; sh.user = user;
	c->sh = sh;
	block_geometry(c);
	uint64_t const need = fseq_shard_xbuf_words(c, world);
	if (xbuf_words < need) { c->sh = Shard{}; return fail(c, FSEQ_E_ARG, "exchange buffer too small (fseq_shard_xbuf_words)"); }
	// every rank that owns blocks must own at least one regular DP round, and the last one the final cell's column
	DpSchedule const S = dp_schedule((uint32_t) c->p.segment_length, (uint32_t) c->p.n);
	SpecPlan const P = spec_plan(c, S);
	bool ok = P.nchunks() >= 1;
	if (ok && c->sh.rank < c->sh.active && P.mine_hi <= P.mine_lo) ok = false;
	// (the plan is the same on every rank: check every rank's share here so that all ranks fail together)
	for (uint32_t g = 0; ok && g < c->sh.active; ++g)
	{
		Shard probe = c->sh; probe.rank = g;
		Shard const keep = c->sh; c->sh = probe;
		SpecPlan const Q = spec_plan(c, S);
		c->sh = keep;
		if (Q.mine_hi <= Q.mine_lo) ok = false;
		if (g + 1 == c->sh.active && Q.mine_hi != Q.nchunks()) ok = false;
	}
	if (!ok) { c->sh = Shard{}; return fail(c, FSEQ_E_UNSUPPORTED, "too few columns per rank for this segment length: use fewer ranks"); }
	return FSEQ_OK;
}

int fseq_shard_abort(fseq_ctx *c, int code)
{
	if (!c) return FSEQ_E_ARG;
	if (!c->sh.on) return FSEQ_OK;
	(void) hipSetDevice(c->p.device);
	shard_post_failure(c, code ? code : FSEQ_E_HIP);
	return FSEQ_OK;
}

int fseq_shard_columns(fseq_ctx const *c, uint64_t *first, uint64_t *last)
{
	if (!c || !first || !last) return FSEQ_E_ARG;
	*first = held_lo(c); *last = held_hi(c);
	return FSEQ_OK;
}

int fseq_shard_owner(fseq_ctx const *c, uint64_t rb, uint32_t *rank)
{
	if (!c || !rank || rb > c->p.n) return FSEQ_E_ARG;
	*rank = c->sh.on ? (uint32_t) std::min<uint64_t>(rb / ((uint64_t) c->sh.bpr * c->B), c->sh.active - 1u) : 0u;
	return FSEQ_OK;
}

int fseq_generate_synthetic(fseq_ctx *c, fseq_synth_spec const *spec)
{
	if (!c || !spec || 0 == spec->n_founders || 0 == spec->block_len || spec->kind > 1) return FSEQ_E_ARG;
	(void) hipSetDevice(c->p.device);
	char const *alpha = spec->kind ? "ACGTRYSWKMBDHVN-" : "ACGT";
	uint32_t const sigma = spec->kind ? 16u : 4u;
	c->sigma = sigma;
	int rc = alloc_msa(c);
	if (rc) { shard_post_failure(c, rc); return rc; }
	SynthArgs A;
	A.seed = spec->seed; A.n_founders = spec->n_founders; A.block_len = spec->block_len;
	A.mut_threshold = spec->mut_threshold; A.kind = spec->kind; A.sigma = sigma;
	for (uint32_t i = 0; i < 16; ++i) A.code_of_sym[i] = 0;
	for (uint32_t i = 0; i < sigma; ++i)
	{
		uint32_t rank = 0;
		for (uint32_t j = 0; j < sigma; ++j) rank += ((uint8_t) alpha[j] < (uint8_t) alpha[i]) ? 1u : 0u;
		A.code_of_sym[i] = (uint8_t) rank;
		c->code_to_byte[rank] = (uint8_t) alpha[i];
	}
	uint64_t const k_lo = held_lo(c), k_hi = held_hi(c);     // sharded: this rank's columns only
	if (k_hi > k_lo)
	{
		dim3 const grid((uint32_t) ((c->ld / 4 + 255) / 256), (uint32_t) std::min<uint64_t>(k_hi - k_lo, 65535));
		hipLaunchKernelGGL(k_synth, grid, dim3(256), 0, c->stream, A, c->d_msa, c->ld, c->p.m, k_lo, k_hi, c->bsh);
	}
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	c->have_input = true;
	c->have_result = false;
	c->kernels_ready = false;
	c->X_hint = 0;
	c->bk_given_up = -1; c->bt_given_up = -1; c->colmask_ready = false; c->shard_dp_full_sticky = false; c->red_force_full.clear(); c->red_plan_valid = false; c->red_declined = false;
	return FSEQ_OK;
}

int fseq_get_matrix(fseq_ctx *c, uint64_t c0, uint64_t c1, uint8_t *out, size_t row_stride, size_t col_stride)
{
	if (!c || !out || !c->have_input || c0 > c1 || c1 > c->p.n) return FSEQ_E_ARG;
	if (c0 < held_lo(c) || c1 > held_hi(c)) return fail(c, FSEQ_E_ARG, "columns not held by this rank");
	(void) hipSetDevice(c->p.device);
	std::vector<uint8_t> buf((c1 - c0) * c->ld);
	uint32_t const bsh = c->bsh, smask = (1u << bsh) - 1u, bits = 8u >> bsh, cmask = (1u << bits) - 1u;
	HIP_TRY(c, hipMemcpy(buf.data(), c->d_msa + c0 * c->ld, buf.size(), hipMemcpyDeviceToHost));
	for (uint64_t col = c0; col < c1; ++col)
		for (uint32_t r = 0; r < c->p.m; ++r)
			out[(size_t) r * row_stride + (col - c0) * col_stride] =
				c->code_to_byte[(buf[(col - c0) * c->ld + (r >> bsh)] >> ((r & smask) * bits)) & cmask];
	return FSEQ_OK;
}

int fseq_run_segmentation(fseq_ctx *c, fseq_result *res)
{
	if (!c || !res) return FSEQ_E_ARG;
	if (!c->have_input) return fail(c, FSEQ_E_ARG, "no input set");
	(void) hipSetDevice(c->p.device);
	c->have_result = false;
	c->sh.closed = false;
	if (!c->kernels_ready)
	{
		int rc = prepare_geometry(c);
		if (rc) { shard_post_failure(c, rc); return rc; }
	}
	// generate_context::calculate_segmentation, generate_context.cc:386-389
	if (c->p.n < 2 * c->p.segment_length)
	{
		if (c->sh.on) return fail(c, FSEQ_E_UNSUPPORTED, "the short path (n < 2L) is one sweep and does not shard");
		return run_short_path(c, res);
	}
	int const rc = run_long_path(c, res);
	shard_post_failure(c, rc);                   // sharded: the other ranks learn of it in their next exchange
	return rc;
}

/* replaces: nothing in the reference (one process, one address space).  A context that shares its device with other
 * contexts or ranks plans its pass-2 stride states inside `bytes` of device memory in all (0 = whatever is free). */
int fseq_debug_set_tuning(fseq_ctx *c, char const *name, char const *value)
{
	if (!c || !name) return FSEQ_E_ARG;
	// sharded: the input was laid out for the block partition of the knobs in force when it was set, and every rank must
	// plan the same partition -- the knobs of a sharded run are set (identically on every rank) before the input
	if (c->sh.on && c->have_input) return fail(c, FSEQ_E_ARG, "sharded run: set tuning knobs before the input is set (identically on every rank)");
	if (!c->tune.set(name, value)) return fail(c, FSEQ_E_ARG, "unknown tuning knob");
	// (the geometry and the kernel choice may depend on it: the work buffers of an earlier run were sized for the old one)
	(void) hipSetDevice(c->p.device);
	if (c->stream) (void) hipStreamSynchronize(c->stream);
	free_work(c);
	c->have_result = false;
	c->kernels_ready = false;
	// (what the last run saw belongs to the old geometry: a block the tree or the trie ranked then may be given up now)
	c->bk_given_up = -1; c->bt_given_up = -1; c->red_force_full.clear(); c->red_plan_valid = false; c->red_declined = false;
	return FSEQ_OK;
}

int fseq_set_progress(fseq_ctx *c, fseq_progress_fn fn, void *user)
{
	if (!c) return FSEQ_E_ARG;
	c->progress_fn = fn; c->progress_user = user;
	return FSEQ_OK;
}
uint64_t fseq_step_max(fseq_ctx const *c) { return c ? c->step_max.load(std::memory_order_relaxed) : 0; }
uint64_t fseq_current_step(fseq_ctx const *c) { return c ? c->current_step.load(std::memory_order_relaxed) : 0; }

int fseq_set_memory_budget(fseq_ctx *c, uint64_t bytes)
{
	if (!c) return FSEQ_E_ARG;
	c->mem_budget = bytes;
	return FSEQ_OK;
}

int fseq_run_segmentation_batch(fseq_ctx *const *ctxs, size_t count, fseq_result *results, int *return_codes)
{
	if ((count && (!ctxs || !results || !return_codes))) return FSEQ_E_ARG;
	for (size_t i = 0; i < count; ++i)
		for (size_t j = 0; j < i; ++j)
			if (ctxs[i] == ctxs[j]) return FSEQ_E_ARG;                 // a context runs one alignment at a time
	std::vector<std::thread> workers;
	workers.reserve(count);
	for (size_t i = 1; i < count; ++i)
		workers.emplace_back([=]() { return_codes[i] = fseq_run_segmentation(ctxs[i], &results[i]); });
	if (count) return_codes[0] = fseq_run_segmentation(ctxs[0], &results[0]);
	for (auto &w : workers) w.join();
	return FSEQ_OK;
}

int fseq_get_traceback(fseq_ctx *c, fseq_dp_arg *out)
{
	if (!c || !out || !c->have_result) return FSEQ_E_ARG;
	std::copy(c->traceback.begin(), c->traceback.end(), out);
	return FSEQ_OK;
}

int fseq_get_segments(fseq_ctx *c, fseq_segment *out)
{
	if (!c || !out || !c->have_result) return FSEQ_E_ARG;
	std::copy(c->segments.begin(), c->segments.end(), out);
	return FSEQ_OK;
}

int fseq_boundary_state(fseq_ctx *c, uint64_t i, uint32_t *a_out, uint32_t *d_out)
{
	if (!c || !c->have_result || i >= c->segments.size()) return FSEQ_E_ARG;
	(void) hipSetDevice(c->p.device);
	size_t const m = c->p.m;
	if (i >= c->snap_slot.size() || c->snap_slot[i] < 0) return fail(c, FSEQ_E_ARG, "boundary state held by another rank (fseq_shard_owner)");
	size_t const slot = (size_t) c->snap_slot[i];
	if (a_out) HIP_TRY(c, hipMemcpy(a_out, c->d_snap_a + slot * m, m * 4, hipMemcpyDeviceToHost));
	if (d_out) HIP_TRY(c, hipMemcpy(d_out, c->d_snap_d + slot * m, m * 4, hipMemcpyDeviceToHost));
	return FSEQ_OK;
}

int fseq_short_path_runs(fseq_ctx *c, uint32_t *first_idx, uint32_t *run_len)
{
	if (!c || !c->have_result || !c->res.short_path) return FSEQ_E_ARG;
	if (first_idx) std::copy(c->sp_first.begin(), c->sp_first.end(), first_idx);
	if (run_len) std::copy(c->sp_len.begin(), c->sp_len.end(), run_len);
	return FSEQ_OK;
}

int fseq_debug_dp(fseq_ctx *c, uint32_t *lb, uint32_t *max_size, uint32_t *size)
{
	if (!c || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	(void) hipSetDevice(c->p.device);
	if (lb) HIP_TRY(c, hipMemcpy(lb, c->dp.LB, c->dp_size * 4, hipMemcpyDeviceToHost));
	if (max_size) HIP_TRY(c, hipMemcpy(max_size, c->dp.M, c->dp_size * 4, hipMemcpyDeviceToHost));
	if (size) HIP_TRY(c, hipMemcpy(size, c->dp.SZ, c->dp_size * 4, hipMemcpyDeviceToHost));
	return FSEQ_OK;
}

int fseq_debug_dp_owned(fseq_ctx *c, uint64_t *first, uint64_t *last, int *final_cell, int *whole_arrays)
{
	if (!c || !c->have_result || c->res.short_path) return FSEQ_E_ARG;
	uint64_t lo = 0, hi = c->dp_size;
	int fin = 1, whole = 1;
	if (c->sh.on)
	{
		Shard const &sh = c->sh;
		bool const have = sh.rank < sh.active && sh.rank < c->own_lo.size();
		lo = have ? c->own_lo[sh.rank] : 0; hi = have ? c->own_hi[sh.rank] : 0;
		fin = have && sh.rank + 1u == sh.active ? 1 : 0;
		whole = c->dp_window_mode ? 0 : 1;
	}
	if (first) *first = lo;
	if (last) *last = hi;
	if (final_cell) *final_cell = fin;
	if (whole_arrays) *whole_arrays = whole;
	return FSEQ_OK;
}

int fseq_debug_ranges(uint64_t *pushes, uint64_t *pops, int *with_roctx)
{
	if (pushes) *pushes = g_range_pushes.load(std::memory_order_relaxed);
	if (pops) *pops = g_range_pops.load(std::memory_order_relaxed);
#ifdef FSEQ_WITH_ROCTX
	if (with_roctx) *with_roctx = 1;
#else
	if (with_roctx) *with_roctx = 0;
#endif
	return FSEQ_OK;
}

int fseq_debug_clock(fseq_ctx *c, double *ghz, uint32_t *workgroups)
{
	if (!c || !ghz) return FSEQ_E_ARG;
#ifdef FSEQ_CLOCK_STAMPS
	(void) hipSetDevice(c->p.device);
	std::vector<unsigned long long> st((size_t) FSEQ_CLOCK_SLOTS * 4);
	HIP_TRY(c, hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_clock_stamps), st.size() * 8));
	std::vector<double> f;
	for (uint32_t i = 0; i < FSEQ_CLOCK_SLOTS; ++i)
	{
		unsigned long long const t0 = st[4 * i], r0 = st[4 * i + 1], t1 = st[4 * i + 2], r1 = st[4 * i + 3];
		if (t1 > t0 && r1 > r0) f.push_back((double) (t1 - t0) / (double) (r1 - r0) * 0.1);      // cycles per 10 ns = GHz x 10
	}
	if (f.empty()) return fail(c, FSEQ_E_ARG, "no clock stamps: run a long-path segmentation first");
	std::nth_element(f.begin(), f.begin() + f.size() / 2, f.end());
	*ghz = f[f.size() / 2];
	if (workgroups) *workgroups = (uint32_t) f.size();
	return FSEQ_OK;
#else
	(void) workgroups;
	*ghz = 0.0;
	return fail(c, FSEQ_E_UNSUPPORTED, "built without -DFSEQ_CLOCK_STAMPS (the product kernels execute no stamp)");
#endif
}

int fseq_debug_block_state(fseq_ctx *c, uint64_t block_idx, uint32_t *a_out, uint32_t *d_out)
{
	if (!c || !c->have_result || c->res.short_path || block_idx > c->nblocks) return FSEQ_E_ARG;
	if (c->sh.on && (block_idx < c->sh.b_lo || block_idx > c->sh.b_hi)) return fail(c, FSEQ_E_ARG, "block state held by another rank");
	(void) hipSetDevice(c->p.device);
	size_t const m = c->p.m;
	if (a_out) HIP_TRY(c, hipMemcpy(a_out, c->d_bstate_a + block_idx * m, m * 4, hipMemcpyDeviceToHost));
	if (d_out) HIP_TRY(c, hipMemcpy(d_out, c->d_bstate_d + block_idx * m, m * 4, hipMemcpyDeviceToHost));
	return FSEQ_OK;
}

int fseq_debug_column_list(fseq_ctx *c, uint64_t col, uint32_t *values, uint32_t *counts,
                           uint32_t *n_entries, uint32_t *cnt0, uint32_t *complete)
{
	if (!c || !c->have_result || c->res.short_path || col >= c->p.n) return FSEQ_E_ARG;
	(void) hipSetDevice(c->p.device);
	uint4 h;
	HIP_TRY(c, hipMemcpy(&h, c->d_hdr + col, sizeof(h), hipMemcpyDeviceToHost));
	std::vector<uint2> e(h.x);
	if (h.x) HIP_TRY(c, hipMemcpy(e.data(), c->d_ent + col * (size_t) c->stride, h.x * sizeof(uint2), hipMemcpyDeviceToHost));
	for (uint32_t i = 0; i < h.x; ++i)
	{
		if (values) values[i] = e[i].x;
		if (counts) counts[i] = e[i].y;
	}
	if (n_entries) *n_entries = h.x;
	if (cnt0) *cnt0 = h.y;
	if (complete) *complete = h.z;
	return FSEQ_OK;
}

int fseq_debug_dp_schedule(uint64_t segment_length, uint64_t n, uint64_t col_hi, uint32_t *n_rounds, uint32_t *cells_per_round,
                           uint32_t *rounds_within, int *pipelined)
{
	if (0 == segment_length || n < 2 * segment_length || n >= 0xFFFFFFF0ull) return FSEQ_E_ARG;
	DpSchedule const S = dp_schedule((uint32_t) segment_length, (uint32_t) n);
	if (n_rounds) *n_rounds = S.nrounds;
	if (cells_per_round) *cells_per_round = S.RL;
	if (rounds_within) *rounds_within = dp_rounds_within(S, col_hi);
	if (pipelined) *pipelined = S.pipe ? 1 : 0;
	return FSEQ_OK;
}

// ---- row-sharded pBWT sweep: the north-star partition as a conformance path (fseq_rowshard.hpp) ------------------
uint64_t fseq_rowshard_xbuf_words(uint32_t m, uint32_t bits, uint32_t world)
{
	if (!m || !world || (bits != 2 && bits != 4 && bits != 8)) return 0;
	uint32_t const bsh = bits == 2 ? 2u : bits == 4 ? 1u : 0u;
	uint64_t const cw = (sym_bytes(m, bsh) + 3u) / 4u;
	return 2 * (cw + 2ull * m) + (uint64_t) RS_SLOT * world + 64;
}

int fseq_rowshard_rows(uint32_t m, uint32_t bits, uint32_t rank, uint32_t world, uint32_t *row_lo, uint32_t *row_hi)
{
	if (!m || !world || rank >= world || !row_lo || !row_hi || (bits != 2 && bits != 4 && bits != 8)) return FSEQ_E_ARG;
	uint32_t const bsh = bits == 2 ? 2u : bits == 4 ? 1u : 0u;
	uint64_t const cw = (sym_bytes(m, bsh) + 3u) / 4u, rpw = 32u / bits;
	*row_lo = (uint32_t) std::min<uint64_t>(m, cw * rank / world * rpw);
	*row_hi = (uint32_t) std::min<uint64_t>(m, cw * (rank + 1) / world * rpw);
	return FSEQ_OK;
}

int fseq_rowshard_pbwt(fseq_rowshard const *A, uint32_t *a_out, uint32_t *d_out, uint32_t *pos_lo, uint32_t *pos_hi,
                       double *ms, uint64_t *n_exchanges)
{
	if (!A || !A->m || !A->world || A->rank >= A->world || !A->d_cols || !a_out || !d_out) return FSEQ_E_ARG;
	if ((A->bits != 2 && A->bits != 4 && A->bits != 8) || A->sigma < 1 || A->sigma > (1u << A->bits)) return FSEQ_E_ARG;
	uint32_t const m = A->m, G = A->world, g = A->rank;
	uint32_t const bsh = A->bits == 2 ? 2u : A->bits == 4 ? 1u : 0u;
	uint64_t const cw = (sym_bytes(m, bsh) + 3u) / 4u;
	if (A->ld % 4 || A->ld < cw * 4) return FSEQ_E_ARG;
	if (A->ncols > 0xFFFFFFFEull) return FSEQ_E_ARG;
	if (G > 1 && (!A->xbuf || !A->fn)) return FSEQ_E_ARG;
	if (!A->xbuf || A->xbuf_words < fseq_rowshard_xbuf_words(m, A->bits, G)) return FSEQ_E_ARG;
	if (hipSetDevice(A->device) != hipSuccess) return FSEQ_E_HIP;
	uint32_t nbits = 1;
	while ((1u << nbits) < A->sigma) ++nbits;
	uint32_t const npass = (nbits + 1) / 2;
	uint64_t const reg = cw + 2ull * m;
	uint32_t *const xb = static_cast<uint32_t *>(A->xbuf);
	uint32_t *const slots = xb + 2 * reg;
	auto C_ = [&](uint32_t r) { return xb + r * reg; };
	auto A_ = [&](uint32_t r) { return xb + r * reg + cw; };
	auto D_ = [&](uint32_t r) { return xb + r * reg + cw + m; };
	uint32_t const p_lo = (uint32_t) ((uint64_t) m * g / G), p_hi = (uint32_t) ((uint64_t) m * (g + 1) / G), ml = p_hi - p_lo;
	uint32_t const w_lo = (uint32_t) (cw * g / G), w_hi = (uint32_t) (cw * (g + 1) / G);
	hipStream_t st = nullptr;
	if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return FSEQ_E_HIP;
	uint64_t nex = 0;
	bool ok = true;
	auto H = [&](hipError_t e) { if (e != hipSuccess) ok = false; return e == hipSuccess; };
	auto xch = [&](uint64_t off, uint64_t count) {
		++nex;
		if (G == 1 || !ok) return;
		if (!H(hipStreamSynchronize(st))) return;
		if (A->fn(A->user, off, count, 0) != 0) ok = false;
	};
	auto contrib = [&](uint64_t k, uint32_t r) {
		H(hipMemsetAsync(C_(r), 0, cw * 4, st));
		if (w_hi > w_lo)
			hipLaunchKernelGGL(k_rs_contrib, dim3((w_hi - w_lo + 255u) / 256u), dim3(256), 0, st,
			                   reinterpret_cast<uint32_t const *>(static_cast<uint8_t const *>(A->d_cols) + k * A->ld), w_lo, w_hi, C_(r));
	};
	double const t0 = now_ms();
	hipLaunchKernelGGL(k_rs_init, dim3((m + 255u) / 256u), dim3(256), 0, st, A_(0), D_(0), m);
	uint32_t colreg = 0, adreg = 0;
	if (A->ncols)
	{
		contrib(0, 0);
		xch(0, cw);                                              // X0 of column 0
	}
	for (uint64_t k = 0; k < A->ncols && ok; ++k)
		for (uint32_t pass = 0; pass < npass && ok; ++pass)
		{
			uint8_t const *col = reinterpret_cast<uint8_t const *>(C_(colreg));
			H(hipMemsetAsync(slots, 0, (size_t) RS_SLOT * G * 4, st));
			hipLaunchKernelGGL(k_rs_sweep<false>, dim3(1), dim3(ST), 0, st, col, A_(adreg) + p_lo, D_(adreg) + p_lo, ml, bsh, pass,
			                   (uint32_t) (k + 1), slots, g, G, (uint32_t *) nullptr, (uint32_t *) nullptr);
			xch(2 * reg, (uint64_t) RS_SLOT * G);                   // X1 + X2
			uint32_t const r2 = 1u - adreg;
			H(hipMemsetAsync(A_(r2), 0, (size_t) 2 * m * 4, st));
			hipLaunchKernelGGL(k_rs_sweep<true>, dim3(1), dim3(ST), 0, st, col, A_(adreg) + p_lo, D_(adreg) + p_lo, ml, bsh, pass,
			                   (uint32_t) (k + 1), slots, g, G, A_(r2), D_(r2));
			if (pass + 1 == npass && k + 1 < A->ncols)
			{
				contrib(k + 1, r2);
				xch(r2 * reg, reg);                                  // X3 + X0 of the next column
				colreg = r2;
			}
			else
				xch(r2 * reg + cw, 2ull * m);                        // X3
			adreg = r2;
		}
	if (ok) H(hipStreamSynchronize(st));
	double const t1 = now_ms();
	if (ok && ml)
	{
		H(hipMemcpy(a_out + p_lo, A_(adreg) + p_lo, (size_t) ml * 4, hipMemcpyDeviceToHost));
		H(hipMemcpy(d_out + p_lo, D_(adreg) + p_lo, (size_t) ml * 4, hipMemcpyDeviceToHost));
	}
	(void) hipStreamDestroy(st);
	if (pos_lo) *pos_lo = p_lo;
	if (pos_hi) *pos_hi = p_hi;
	if (ms) *ms = t1 - t0;
	if (n_exchanges) *n_exchanges = nex;
	return ok ? FSEQ_OK : FSEQ_E_HIP;
}

int fseq_debug_rmq(int device, uint32_t const *keys, uint32_t count, uint32_t const *beg, uint32_t const *end, uint32_t n_queries,
                   uint32_t *index_hbm, uint32_t *index_lds)
{
	if (!keys || !count || (n_queries && (!beg || !end || !index_hbm))) return FSEQ_E_ARG;
	for (uint32_t q = 0; q < n_queries; ++q)
		if (beg[q] >= end[q] || end[q] > count) return FSEQ_E_ARG;
	if (hipSetDevice(device) != hipSuccess) return FSEQ_E_HIP;
	DpArrays A{};
	A.tstride = count / 64 + 2;
	uint32_t *d_prev = nullptr, *d_lift = nullptr, *d_r0 = nullptr;
	uint2 *d_q = nullptr, *d_out = nullptr;
	SpecCtl *d_ctl = nullptr;
	int rc = FSEQ_E_HIP;
	auto A_free = [&]() {
		(void) hipFree(A.M); (void) hipFree(A.K); (void) hipFree(A.Tb); (void) hipFree(A.Tbv); (void) hipFree(d_prev); (void) hipFree(d_lift);
		(void) hipFree(d_r0); (void) hipFree(d_q); (void) hipFree(d_out); (void) hipFree(d_ctl);
	};
	std::vector<uint2> hq(n_queries), ho(n_queries);
	for (uint32_t q = 0; q < n_queries; ++q) hq[q] = make_uint2(beg[q], end[q]);
	uint32_t const r0[2] = {0u, count};                       // one "chunk" of `count` rounds of one entry
	do
	{
		if (hipMalloc((void **) &A.M, ((size_t) count + 64) * 4) != hipSuccess) break;
		if (hipMalloc((void **) &A.K, ((size_t) count + 64) * 8) != hipSuccess) break;
		if (hipMalloc((void **) &A.Tb, (size_t) 32 * A.tstride * 4) != hipSuccess) break;
		if (hipMalloc((void **) &A.Tbv, (size_t) 32 * A.tstride * 4) != hipSuccess) break;
		if (hipMalloc((void **) &d_prev, ((size_t) count + 64) * 4) != hipSuccess) break;
		if (hipMalloc((void **) &d_lift, 16) != hipSuccess) break;
		if (hipMalloc((void **) &d_r0, 16) != hipSuccess) break;
		if (hipMalloc((void **) &d_q, std::max<size_t>(1, n_queries) * 8) != hipSuccess) break;
		if (hipMalloc((void **) &d_out, std::max<size_t>(1, n_queries) * 8) != hipSuccess) break;
		if (hipMalloc((void **) &d_ctl, sizeof(SpecCtl)) != hipSuccess) break;
		if (hipMemcpy(A.M, keys, (size_t) count * 4, hipMemcpyHostToDevice) != hipSuccess) break;
		if (hipMemset(d_lift, 0, 16) != hipSuccess || hipMemset(d_ctl, 0, sizeof(SpecCtl)) != hipSuccess) break;
		if (hipMemcpy(d_r0, r0, 8, hipMemcpyHostToDevice) != hipSuccess) break;
		if (n_queries && hipMemcpy(d_q, hq.data(), (size_t) n_queries * 8, hipMemcpyHostToDevice) != hipSuccess) break;
		SpecGeom G;
		G.chunk_r0 = d_r0; G.RL = 1; G.nchunks = 1; G.NR = count; G.t_final = count + 32u; G.win = 1;
		hipLaunchKernelGGL(k_spec_rebuild, dim3((count + 255u) / 256u), dim3(256), 0, 0, A, d_prev, G, d_lift, d_ctl);
		hipLaunchKernelGGL(k_spec_table, dim3((count / 64u + 255u) / 256u + 1u), dim3(256), 0, 0, A, count / 64u, d_ctl);
		size_t const lds = (size_t) DPW * 12 + (size_t) DP_LEVELS * DP_TRN * 8 + 64;
		if (allow_lds(k_debug_rmq, lds) != hipSuccess) break;
		hipLaunchKernelGGL(k_debug_rmq, dim3(1), dim3(1024), lds, 0, A, count, d_q, n_queries, d_out);
		if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) break;
		if (n_queries && hipMemcpy(ho.data(), d_out, (size_t) n_queries * 8, hipMemcpyDeviceToHost) != hipSuccess) break;
		for (uint32_t q = 0; q < n_queries; ++q) { index_hbm[q] = ho[q].x; if (index_lds) index_lds[q] = ho[q].y; }
		rc = FSEQ_OK;
	} while (false);
	A_free();
	return rc;
}

int fseq_get_timings(fseq_ctx const *c, fseq_timings *out)
{
	if (!c || !out) return FSEQ_E_ARG;
	*out = c->tm;
	return FSEQ_OK;
}

} // extern "C"
