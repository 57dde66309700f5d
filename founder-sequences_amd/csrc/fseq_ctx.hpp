// fseq_ctx.hpp -- the context behind the C ABI (include/fseq.h) and the small helpers every translation unit of the library
// shares: csrc/fseq_api.hip (the path: geometry, buffers, phases, sharding, the ABI's entry points) and
// csrc/fseq_api_join.hip (the host joiners, their device front and the output writers).  Internal: nothing here is part of
// the boundary.
#pragma once

#include "../../include/fseq.h"
#include "../../include/fseq_debug.h"
#include "fseq_types.hpp"

#include <hip/hip_runtime.h>

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

namespace fseq {

struct KernelSet {
	uint32_t T, E, sigma, cap;
	size_t lds_colblock, lds_snap;
	uint32_t scan_shift;                     // partition steps of this configuration may scan keys while every divergence is < 2^scan_shift
	void (*rank)(hipStream_t, uint32_t grid, size_t lds, uint8_t const *, size_t, uint32_t, uint64_t, uint32_t, uint32_t, uint32_t npass, uint32_t bsh,
	             uint32_t *, uint32_t *, uint32_t *, uint64_t col0, uint32_t const *only);
	void (*snap)(hipStream_t, uint32_t grid, size_t lds, uint8_t const *, size_t, uint32_t, uint64_t, uint32_t, uint32_t, uint32_t npass, uint32_t bsh,
	             uint32_t const *, uint32_t const *, uint64_t const *, uint2 const *, uint32_t *, uint32_t *,
	             uint64_t const *task_src, uint32_t snap_stride, uint32_t const *ss_a, uint32_t const *ss_d, uint32_t keyed);
	size_t (*columns_lds)(uint32_t B);
	void (*columns)(hipStream_t, uint32_t grid, size_t lds, uint8_t const *, size_t, uint32_t, uint64_t, uint32_t, uint32_t,
	                uint32_t const *, uint32_t const *, uint32_t, uint32_t, uint32_t, uint2 *, uint4 *, uint32_t npass, uint32_t bsh,
	                uint32_t snap_stride, uint32_t *ss_a, uint32_t *ss_d, uint32_t block0, uint32_t *done_host, uint32_t epoch, uint32_t const *colmask, uint32_t const *blocklist);
	uint32_t (*columns_resident)(size_t lds);                 // workgroups of k_columns one CU holds
	size_t lds_chain;
	void (*chain)(hipStream_t, uint32_t grid, size_t lds, uint32_t const *rank, uint32_t const *keyd, uint32_t const *nkeys, uint32_t m,
	              uint32_t nb_total, uint32_t G, uint64_t cols_per_block, uint32_t const *start_a, uint32_t const *start_d,
	              uint32_t *out_a, uint32_t *out_d, uint32_t *out_rank, uint32_t *out_keyd, uint32_t *out_nkeys, uint32_t grp0, uint32_t keyed);
	hipError_t (*prepare)(size_t lds_columns);
	hipError_t (*prepare_columns)(size_t lds_columns);
};

template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes)
{
	if (bytes <= 64 * 1024) return hipSuccess;
	return hipFuncSetAttribute(reinterpret_cast<void const *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
}

struct Stream2Config { uint32_t T, E, key_shift, pack; size_t (*lds)(uint32_t colbytes); hipError_t (*prepare)(size_t lds);
	void (*launch)(hipStream_t, uint32_t grid, size_t lds, uint8_t const *, size_t, uint32_t, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t *,
	               uint32_t, uint32_t, uint32_t, uint2 *, uint4 *, uint32_t, uint32_t *, uint32_t *, uint32_t, uint32_t *, uint32_t, uint32_t, uint32_t const *blocklist);
	uint32_t (*resident)(size_t lds);
	// pass 2 on the same tile step (packed rows only; nullptr otherwise): k_columns_stream2<.., S2_SNAP>
	void (*launch_snap)(hipStream_t, uint32_t grid, size_t lds, uint8_t const *, size_t, uint32_t, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t *,
	                    uint32_t snap_stride, uint32_t *ss_a, uint32_t *ss_d, S2SnapArgs const &); };
// [r5] phase C on representative rows (fseq_reduced.hpp; the kernels live in csrc/fseq_reduced.hip)
struct ReducedSet {
	uint32_t T, E, rows;                     // rows: representatives a workgroup holds
	bool pk, ew;
	size_t (*lds)(uint32_t B, uint32_t symcap);   // symcap: bytes of each staged-column buffer (RedArgs)
	hipError_t (*prepare)(size_t lds);
	void (*launch)(hipStream_t, uint32_t grid, size_t lds, uint8_t const *red_msa, size_t red_ld, uint64_t n, uint32_t B, uint32_t L, uint32_t X, uint32_t stride,
	               uint2 *ent, uint4 *hdr, uint32_t npass, uint32_t bsh, RedArgs const &);
	uint32_t (*resident)(size_t lds);
};
// the smallest configuration that holds `rows` representatives (index into the list; -1: none)
int reduced_config_count();
bool reduced_config(int index, ReducedSet *out);
// pass 2's chain step for the base configuration <T, E> of a KernelSet
struct ChainSnapSet {
	size_t lds;
	hipError_t (*prepare)();
	void (*launch)(hipStream_t, uint32_t grid, size_t lds, uint32_t const *bstate_a, uint32_t const *bstate_d, uint32_t const *rank, uint32_t m,
	               uint32_t const *task_blk, uint32_t const *cls, uint32_t const *headd, uint32_t const *ncls, uint32_t cap, uint32_t *snap_a, uint32_t *snap_d, uint32_t keyed);
};
bool select_chain_snap(uint32_t T, uint32_t E, ChainSnapSet *out);

// the kernel configurations and their launchers (csrc/fseq_kernelsets.hip)
bool select_kernels(uint32_t m, uint32_t sigma, KernelSet *out, bool no_emitter_wave);
bool select_stream2(uint32_t T, uint32_t E, uint32_t pack, Stream2Config *out);
void launch_blockkeys(uint32_t T, hipStream_t st, uint32_t grid, size_t lds, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
                      uint32_t bsh, uint32_t *rank_, uint32_t *keyd, uint32_t *nkeys, uint64_t col0,
                      uint16_t *scratch, size_t scratch_per_block, uint32_t cap_words, uint32_t *sliced, uint32_t *todo, uint32_t const *only = nullptr);
hipError_t prepare_blockkeys(uint32_t T, size_t lds, bool debug);
uint32_t blocktrie_threads(uint32_t m, bool stream);
size_t blocktrie_lds(uint32_t T);
hipError_t launch_blocktrie(uint32_t bits, uint32_t T, hipStream_t st, uint32_t groups, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
                            uint32_t nblk, uint32_t *rank_, uint32_t *keyd, uint32_t *nkeys, uint64_t col0, uint32_t *ws, size_t per, uint32_t *given_up, uint32_t *todo);
hipError_t launch_reduce_prep(hipStream_t, uint32_t grid, RedPrepArgs const &);
void launch_reduce_check(hipStream_t, uint32_t const *cnt, uint32_t const *planned, uint32_t count, uint32_t *flags);
void launch_reduce_msa(hipStream_t, uint32_t nblocks_listed, uint32_t max_rows, uint8_t const *msa, size_t ld, uint8_t *red, size_t ldr, uint32_t const *cnt,
                       uint32_t const *rows, uint32_t cap, uint64_t n, uint32_t B, uint32_t bsh, uint32_t const *blocks, uint32_t m, bool gather_only);

inline double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}


} // namespace fseq

// Diagnostic / test knobs of the library.  They are read from the environment ONCE, when a context is created
// (fseq_create), and can be set per context with fseq_debug_set_tuning (include/fseq_debug.h); nothing on the run
// path looks at the environment.  Every knob selects among exact alternatives (results never depend on them).
struct Tuning {
	bool debug = false;                  // FSEQ_DEBUG: progress notes on stderr
	bool dp_serial = false;              // FSEQ_DP_SERIAL: the serial DP instead of the speculative sweeps
	int  dp_spec_win = 0, dp_spec_rounds = 0, dp_spec_max_sweeps = 0;      // FSEQ_DP_SPEC_*: tail window, chunk length, sweep budget
	bool stream_plain_scan = false;      // FSEQ_STREAM_PLAIN_SCAN: streamed phase C with the has-based scan (first form)
	bool plain_scan = false;             // FSEQ_PLAIN_SCAN: phase B / pass 2 never scan keys
	bool occurrence_keys = false;        // FSEQ_OCCURRENCE_KEYS: ... scan occurrence keys even where row counts fit the keys
	bool phase_a_classic = false;        // FSEQ_PHASE_A_CLASSIC: phase A as a column sweep
	int  chain_fan = 0;                  // FSEQ_CHAIN_FAN: group size of phase B's recursion
	bool two_level_chain = false;        // FSEQ_TWO_LEVEL_CHAIN
	bool chain_stream_passes = false;    // FSEQ_CHAIN_STREAM_PASSES: streamed phase B as two-bit digit passes (the form before fseq_chainsort.hpp)
	bool chain_stream_single = false;    // FSEQ_CHAIN_STREAM_SINGLE: ... as the sorted step on one workgroup per chain (not spread over the chip)
	bool blockkeys_wide = false;         // FSEQ_BLOCKKEYS_WIDE: 32-bit ids in the streamed key-space tree from the start
	bool blockkeys_single = false;       // FSEQ_BLOCKKEYS_SINGLE: its leaves one by one (no pair leaves)
	bool no_dense_columns = false;       // FSEQ_NO_DENSE_COLUMNS: every column of 4-bit symbols in two digit passes (by itself: one pass where at most four codes are present)
	bool no_blocktrie = false;           // FSEQ_NO_BLOCKTRIE: the streamed phase A without the trie over 16-column words (fseq_blocktrie.hpp): the key-space tree on every block
	bool blocktrie_always = false;       // FSEQ_BLOCKTRIE_ALWAYS: the trie for LDS-resident rows of any count (by itself: from 6,145 rows on)
	bool blockkeys_no_limit = false;     // FSEQ_BLOCKKEYS_NO_LIMIT: the key-space tree slices as often as it takes (never hands a block to the column sweep)
	int  blockkeys_cap = 0;              // FSEQ_BLOCKKEYS_CAP: words of the key-space tree's LDS bitmap
	std::string stream2;                 // FSEQ_STREAM2: "T,E[,pack]" configuration of the streamed phase C, "0" = first form
	bool ss_unpacked = false;            // FSEQ_SS_UNPACKED: 8-byte stride states in the streamed regime
	bool ss_absolute = false;            // FSEQ_SS_ABSOLUTE: stride states hold divergences and pass 2 runs the first form's tile step (the form before round 4)
	int  snap_stride = 0;                // FSEQ_SNAP_STRIDE: first stride tried for the stride states
	bool poison_lists = false;           // FSEQ_POISON_LISTS: lists and headers filled with 0xFF before phase C
	bool no_emitter_wave = false;        // FSEQ_NO_EMITTER_WAVE: phase C without the list wave
	bool join_host = false;              // FSEQ_JOIN_HOST: the greedy joiner's class tables and edges on the host
	bool shard_dp_full = false;          // FSEQ_SHARD_DP_FULL: the sharded DP gathers the whole key array after every sweep (round 2-3 form)
	int  shard_dp_window = 0;            // FSEQ_SHARD_DP_WINDOW: entries of the other ranks a rank holds in front of its own (tests: small windows)
	int  inject_failure_rank = -1;       // FSEQ_INJECT_FAILURE_RANK: this rank of a sharded run fails after phase A
	std::string sync_phases;             // FSEQ_SYNC_PHASES: "ABC": synchronise after these phases (a fault shows where it happened)
	bool check_phase_a = false;          // FSEQ_CHECK_PHASE_A: validate the key blocks on the host before phase B
	bool no_reduced = false;             // FSEQ_NO_REDUCED: phase C and pass 2 on all rows of every block (the form before round 5)
	int  reduced_margin = -1;            // FSEQ_REDUCED_MARGIN: counts beyond the list capacity the choice of vmin allows for (tests: 0 makes lists dig below it)
	bool reduced_always = false;         // FSEQ_REDUCED_ALWAYS: the representatives whenever some block has fewer of them than rows (tests of the mixed runs)
	int  reduced_side = -1;              // FSEQ_REDUCED_SIDE: side streams the configurations' launches may use (0 .. 3)
	bool reduced_serial = false;         // FSEQ_REDUCED_SERIAL: the configurations' launches one after the other on the context's stream (by itself: side by side)
	bool chain_no_xcd_map = false;       // FSEQ_CHAIN_NO_XCD_MAP: the streamed phase B's workgroups taken as they come (by itself: a chain's on one XCD)
	bool reduced_msa_gather = false;     // FSEQ_REDUCED_MSA_GATHER: the reduced alignment by gathers from memory (by itself: the column through LDS where it fits)
	bool reduced_ew = false;             // FSEQ_REDUCED_EW: small blocks on two-wave workgroups (the list on a wave of its own) instead of one wave
	int  stream_block = 0;               // FSEQ_STREAM_BLOCK: columns per block the streamed regime aims for when phase C runs on representatives
	int  reduced_cap = 0;                // FSEQ_REDUCED_CAP: most representatives a block may have (tests: small values send blocks to the run on all rows)

	// returns false for a name it does not know
	bool set(char const *name, char const *value)
	{
		std::string const n(name), v(value ? value : "");
		bool const on = value != nullptr;
		int const iv = atoi(v.c_str());
		if (n == "FSEQ_DEBUG") debug = on;
		else if (n == "FSEQ_DP_SERIAL") dp_serial = on;
		else if (n == "FSEQ_DP_SPEC_WIN") dp_spec_win = on ? std::max(1, iv) : 0;
		else if (n == "FSEQ_DP_SPEC_ROUNDS") dp_spec_rounds = on ? std::max(1, iv) : 0;
		else if (n == "FSEQ_DP_SPEC_MAX_SWEEPS") dp_spec_max_sweeps = on ? std::max(1, iv) : 0;
		else if (n == "FSEQ_STREAM_PLAIN_SCAN") stream_plain_scan = on;
		else if (n == "FSEQ_PLAIN_SCAN") plain_scan = on;
		else if (n == "FSEQ_OCCURRENCE_KEYS") occurrence_keys = on;
		else if (n == "FSEQ_PHASE_A_CLASSIC") phase_a_classic = on;
		else if (n == "FSEQ_CHAIN_FAN") chain_fan = on ? std::max(2, iv) : 0;
		else if (n == "FSEQ_TWO_LEVEL_CHAIN") two_level_chain = on;
		else if (n == "FSEQ_CHAIN_STREAM_PASSES") chain_stream_passes = on;
		else if (n == "FSEQ_CHAIN_STREAM_SINGLE") chain_stream_single = on;
		else if (n == "FSEQ_BLOCKKEYS_WIDE") blockkeys_wide = on;
		else if (n == "FSEQ_BLOCKKEYS_SINGLE") blockkeys_single = on;
		else if (n == "FSEQ_BLOCKKEYS_NO_LIMIT") blockkeys_no_limit = on;
		else if (n == "FSEQ_NO_BLOCKTRIE") no_blocktrie = on;
		else if (n == "FSEQ_NO_DENSE_COLUMNS") no_dense_columns = on;
		else if (n == "FSEQ_BLOCKTRIE_ALWAYS") blocktrie_always = on;
		else if (n == "FSEQ_BLOCKKEYS_CAP") blockkeys_cap = on ? std::max(2048, iv) : 0;
		else if (n == "FSEQ_STREAM2") stream2 = v;
		else if (n == "FSEQ_SS_UNPACKED") ss_unpacked = on;
		else if (n == "FSEQ_SS_ABSOLUTE") ss_absolute = on;
		else if (n == "FSEQ_SNAP_STRIDE") snap_stride = on ? std::max(1, iv) : 0;
		else if (n == "FSEQ_POISON_LISTS") poison_lists = on;
		else if (n == "FSEQ_NO_EMITTER_WAVE") no_emitter_wave = on;
		else if (n == "FSEQ_JOIN_HOST") join_host = on;
		else if (n == "FSEQ_SHARD_DP_FULL") shard_dp_full = on;
		else if (n == "FSEQ_SHARD_DP_WINDOW") shard_dp_window = on ? std::max(64, iv) : 0;
		else if (n == "FSEQ_INJECT_FAILURE_RANK") inject_failure_rank = on ? iv : -1;
		else if (n == "FSEQ_SYNC_PHASES") sync_phases = v;
		else if (n == "FSEQ_CHECK_PHASE_A") check_phase_a = on;
		else if (n == "FSEQ_NO_REDUCED") no_reduced = on;
		else if (n == "FSEQ_REDUCED_MARGIN") reduced_margin = on ? std::max(0, iv) : -1;
		else if (n == "FSEQ_REDUCED_CAP") reduced_cap = on ? std::max(1, iv) : 0;
		else if (n == "FSEQ_REDUCED_EW") reduced_ew = on;
		else if (n == "FSEQ_REDUCED_ALWAYS") reduced_always = on;
		else if (n == "FSEQ_REDUCED_SERIAL") reduced_serial = on;
		else if (n == "FSEQ_REDUCED_SIDE") reduced_side = on ? std::max(0, std::min(3, iv)) : -1;
		else if (n == "FSEQ_REDUCED_MSA_GATHER") reduced_msa_gather = on;
		else if (n == "FSEQ_CHAIN_NO_XCD_MAP") chain_no_xcd_map = on;
		else if (n == "FSEQ_STREAM_BLOCK") stream_block = on ? std::max(64, iv) : 0;
		else return false;
		return true;
	}

	void from_environment()
	{
		static char const *const names[] = {"FSEQ_DEBUG", "FSEQ_DP_SERIAL",
			"FSEQ_DP_SPEC_WIN", "FSEQ_DP_SPEC_ROUNDS", "FSEQ_DP_SPEC_MAX_SWEEPS", "FSEQ_STREAM_PLAIN_SCAN", "FSEQ_PLAIN_SCAN", "FSEQ_OCCURRENCE_KEYS", "FSEQ_PHASE_A_CLASSIC",
			"FSEQ_CHAIN_FAN", "FSEQ_TWO_LEVEL_CHAIN", "FSEQ_BLOCKKEYS_WIDE", "FSEQ_BLOCKKEYS_SINGLE", "FSEQ_BLOCKKEYS_CAP", "FSEQ_STREAM2", "FSEQ_SS_UNPACKED", "FSEQ_SNAP_STRIDE",
			"FSEQ_POISON_LISTS", "FSEQ_NO_EMITTER_WAVE", "FSEQ_JOIN_HOST", "FSEQ_INJECT_FAILURE_RANK", "FSEQ_SYNC_PHASES", "FSEQ_CHECK_PHASE_A",
			"FSEQ_SHARD_DP_FULL", "FSEQ_SHARD_DP_WINDOW", "FSEQ_BLOCKKEYS_NO_LIMIT", "FSEQ_CHAIN_STREAM_PASSES", "FSEQ_CHAIN_STREAM_SINGLE", "FSEQ_SS_ABSOLUTE",
			"FSEQ_NO_BLOCKTRIE", "FSEQ_BLOCKTRIE_ALWAYS", "FSEQ_NO_DENSE_COLUMNS", "FSEQ_NO_REDUCED", "FSEQ_REDUCED_MARGIN", "FSEQ_REDUCED_CAP", "FSEQ_REDUCED_EW", "FSEQ_STREAM_BLOCK", "FSEQ_REDUCED_ALWAYS", "FSEQ_REDUCED_SERIAL", "FSEQ_REDUCED_SIDE", "FSEQ_REDUCED_MSA_GATHER", "FSEQ_CHAIN_NO_XCD_MAP"};
		for (char const *nm : names)
			if (char const *v = getenv(nm)) (void) set(nm, v);
	}
};

// One alignment over several ranks (include/fseq.h, fseq_set_shard): which blocks / columns / DP chunks are mine
struct Shard {
	bool on = false;
	uint32_t rank = 0, world = 1;
	uint32_t *xbuf = nullptr;               // caller-owned exchange buffer (device)
	uint64_t xwords = 0;
	fseq_allreduce_fn fn = nullptr;
	void *user = nullptr;
	uint32_t bpr = 0;                       // blocks per rank = shard_q * chain_fan^shard_k (a rank is one hyper-block of phase B)
	uint32_t active = 1;                    // ranks that own blocks
	uint32_t b_lo = 0, b_hi = 0;            // my blocks
	uint64_t c_lo = 0, c_hi = 0, c_end = 0; // my columns [c_lo, c_hi); held: [c_lo, c_end) (halo for my last DP round)
	bool posted = false;                    // this rank has told the others that it failed (once per context)
	bool closed = false;                    // the run's last exchange is done: nobody is left to hear of a failure
};

struct fseq_ctx {
	fseq_params p{};
	Tuning tune;                             // read from the environment once, at fseq_create
	std::unordered_map<void *, size_t> alloc_sizes;   // device allocations of this context (dev_alloc / dev_free)
	size_t alloc_total = 0;
	uint64_t mem_budget = 0;                  // fseq_set_memory_budget: 0 = whatever is free on the device
	std::atomic<uint64_t> step_max{0}, current_step{0};      // fseq_step_max / fseq_current_step (segmentation_lp_context.hh:122-127)
	fseq_join_profile jp{};                  // the last joiner call (fseq_get_join_profile)
	fseq_progress_fn progress_fn = nullptr;
	void *progress_user = nullptr;
	hipStream_t stream = nullptr;
	std::string err;
	Shard sh;
	uint8_t *d_msa_alloc = nullptr;          // what was allocated; d_msa = d_msa_alloc - c_lo * ld (column k at d_msa + k * ld)
	uint2 *d_ent_alloc = nullptr;
	uint32_t *d_ss_a_alloc = nullptr, *d_ss_d_alloc = nullptr;
	uint32_t *d_bkws = nullptr;              // ... streamed rows: per-workgroup workspace (id arrays, group ids)
	size_t bkws_words = 0;
	uint16_t *d_bk = nullptr;                // phase A in key space (fseq_blockkeys.hpp): per-block scratch (leaf words, group ids)
	size_t bk_per_block = 0, bk_blocks = 0;
	uint32_t bk_cap_words = 0;
	uint32_t bk_T = 0;                       // threads of k_blockkeys (LDS-resident rows)
	uint32_t *d_todo = nullptr;              // phase A: blocks the key-space tree gave up on (the column sweep does them)
	size_t todo_cap = 0;
	int bk_given_up = -1;                    // ... in the last run on this input (-1: not run yet): later runs skip the sweep's launch when
	                                         // it was none, and the tree altogether when it was most blocks
	size_t bk_lds = 0;
	uint32_t *d_colmask_alloc = nullptr;     // 4-bit symbols: the codes present in every held column (k_column_presence), once per input;
	uint32_t *d_colmask = nullptr;           // d_colmask = d_colmask_alloc - c_lo (column k at d_colmask[k])
	bool colmask_ready = false, colmask_use = false;      // ... computed for this input; ... enough dense columns for the kernel that looks at it
	uint32_t *d_btws = nullptr;              // phase A, streamed rows, the trie (fseq_blocktrie.hpp): per-workgroup workspace (the nodes of the levels)
	size_t btws_words = 0;
	uint32_t *d_only = nullptr;              // ... blocks the trie gave up on (the key-space tree does them)
	size_t only_cap = 0;
	int bt_given_up = -1;                    // ... in the last run on this input (-1: not run yet)
	uint32_t *d_chunk_r0 = nullptr;          // speculative DP: first round of every chunk (+ the end)
	uint32_t chunk_cap = 0;
	uint2 *d_tau = nullptr;                  // merge thresholds (k_seg_tau) / counts
	size_t tau_cap = 0;
	std::vector<int64_t> snap_slot;          // segment index -> slot in d_snap_* (-1: another rank's)
	// sharded DP (run_dp_spec): the DP entries [own_lo[g], own_hi[g]) belong to rank g (the last active rank also owns the
	// final cell's); dp_window_mode: a rank holds its own entries and a window of the others' in front of them, not the
	// whole arrays (the traceback then runs rank by rank, follow_traceback_sharded)
	std::vector<uint32_t> own_lo, own_hi;
	bool dp_window_mode = false;
	bool shard_dp_full_sticky = false;       // a sweep of this input read below its window once: whole-array exchanges from then on
	uint64_t dp_exchange_words = 0;          // words the DP's sweep exchanges moved in the last run (diagnostics)

	// input
	uint8_t *d_msa = nullptr;
	size_t ld = 0;
	bool own_msa = false;
	bool have_input = false;
	uint32_t sigma = 0;
	uint8_t code_to_byte[256]{};

	// geometry
	uint32_t B = 0, nblocks = 0, N2 = 0, npass = 1;
	uint32_t auto_B = 0;                     // block length fitted to whole rounds of phase C's workgroups (short inputs)
	fseq::Stream2Config s2{};                      // streamed phase C, second form (T = 0: not in use)
	size_t s2_lds = 0;
	bool stream_staged = false;              // streamed kernels lay tiles out in LDS before writing them (needs 64 KiB more)
	uint32_t bsh = 0;                        // alignment packing: 8 >> bsh bits per symbol (fseq_kernels.hpp sym_bytes)
	fseq::KernelSet ks{};
	bool kernels_ready = false;
	bool use_stream = false;             // m too large for an LDS-resident order: HBM-streamed kernels (fseq_stream.hpp)
	size_t tb_guess = 0;                 // traceback entries of the last run (sizes the speculative copy of the next)
	std::vector<uint2> tau_host;         // merge thresholds that came back with the traceback (not sharded)
	uint32_t *d_ws = nullptr;            // their per-block workspaces
	size_t ws_words = 0;
	size_t lds_columns = 0;

	// device work buffers
	// per column block: key blocks (phase A) and boundary states (phase B), indexed by the block's place in the whole
	// alignment.  A rank of a sharded run allocates its own blocks [b_lo, b_hi] only (*_alloc) and shifts the pointer
	// (block b at d_rank + b * m as before): memory per rank falls with the rank count
	uint32_t *d_rank = nullptr, *d_keyd = nullptr, *d_nkeys = nullptr;
	uint32_t *d_bstate_a = nullptr, *d_bstate_d = nullptr;
	uint32_t *d_rank_alloc = nullptr, *d_keyd_alloc = nullptr, *d_nkeys_alloc = nullptr, *d_bstate_a_alloc = nullptr, *d_bstate_d_alloc = nullptr;
	uint32_t *d_cshist = nullptr;            // streamed phase B spread over the chip (fseq_chainsort.hpp): digit histograms [chain][part][bin]
	size_t cshist_words = 0;
	uint32_t *d_ws_c = nullptr;              // streamed phase C: the per-block workspaces, block b at d_ws_c + b * (words per block)
	uint32_t *d_hrank = nullptr, *d_hkeyd = nullptr, *d_hnkeys = nullptr, *d_hstate_a = nullptr, *d_hstate_d = nullptr;
	// not sharded: phase B over any number of levels (levels[i - 1] = the composites of chain_fan level-(i - 1) key blocks)
	struct ChainLevel { uint32_t count = 0; uint64_t cols = 0; uint32_t *rank = nullptr, *keyd = nullptr, *nkeys = nullptr, *state_a = nullptr, *state_d = nullptr;
	                    uint32_t *rank_alloc = nullptr, *keyd_alloc = nullptr, *nkeys_alloc = nullptr, *state_a_alloc = nullptr, *state_d_alloc = nullptr; };
	std::vector<ChainLevel> levels;
	uint32_t chain_fan = 0;
	uint32_t shard_k = 0, shard_q = 0;       // sharded: a rank's hyper-block = shard_q groups of chain_fan^shard_k blocks
	uint32_t chain_G = 0, n_super = 0;       // sharded: super-blocks of chain_G blocks
	uint32_t chain_G2 = 0, n_hyper = 0;      // third level: hyper-blocks of chain_G2 super-blocks (0 = two levels only)
	uint2 *d_ent = nullptr;
	uint4 *d_hdr = nullptr;
	uint32_t X = 0, stride = 0;
	uint32_t X_hint = 0;                     // list capacity that worked on the last run of this input
	fseq::DpArrays dp{};
	uint32_t *d_Mprev = nullptr;             // chunk-speculative DP: the iterate the last sweep started from
	uint32_t *d_spec = nullptr;              // its per-chunk words (active, changed, tailmin, floor, lift, 2 x ovf) + SpecCtl
	uint32_t spec_cap = 0;
	uint32_t *d_flags = nullptr;
	uint32_t *d_recent = nullptr;            // k_boundary_recent counts, one per block boundary
	uint64_t dp_size = 0;
	uint64_t *d_cols = nullptr;           // scratch: column / rb lists
	size_t cols_cap = 0;
	uint2 *d_grp = nullptr;
	size_t grp_cap = 0;
	uint64_t *d_src = nullptr;
	size_t src_cap = 0;
	uint32_t snap_stride = 16;            // phase C drops the exact (a,d) every snap_stride columns for pass 2
	uint32_t *d_ss_a = nullptr, *d_ss_d = nullptr;
	uint32_t ss_pack = 0;                 // streamed rows: stride states packed to 5 bytes per row (bits of a row id; fseq_stream.hpp)
	bool ss_ids = false;                  // ... and in ID form: the packed rows of phase C's workspace; pass 2 replays them on the same tile step (fseq_stream2.hpp, S2_SNAP)
	uint32_t *d_bs_w_alloc = nullptr, *d_bs_w = nullptr;      // ... with every block's start state in the same form (block b at d_bs_w + b * m)
	uint8_t *d_bs_h_alloc = nullptr, *d_bs_h = nullptr;
	uint32_t *d_wgblk = nullptr;          // pass 2 on the tile step: block and groups of every workgroup
	uint2 *d_wggrp = nullptr;
	size_t wg_cap = 0;
	uint2 *d_gent = nullptr;
	uint4 *d_ghdr = nullptr;
	size_t gather_cap = 0, gather_stride = 0;
	uint32_t *d_snap_a = nullptr, *d_snap_d = nullptr;
	size_t snap_cap = 0;

	// [r5] phase C / pass 2 on representative rows (fseq_reduced.hpp): per block [red_cap] representatives (ascending row id),
	// their block keys, the reduced start state; the reduced alignment (column k at d_red_msa + k * red_ld)
	uint32_t *d_red_cnt = nullptr, *d_red_vmin = nullptr, *d_red_rows = nullptr, *d_red_leaf = nullptr, *d_red_a = nullptr, *d_red_d = nullptr;
	uint32_t *d_red_rows_alloc = nullptr, *d_red_leaf_alloc = nullptr, *d_red_a_alloc = nullptr, *d_red_d_alloc = nullptr;      // (a rank holds its own blocks' rows)
	uint32_t *d_red_invalid = nullptr, *d_red_blocks = nullptr;
	uint32_t red_cap = 0, red_blocks_cap = 0;
	uint8_t *d_red_msa = nullptr, *d_red_msa_alloc = nullptr;
	size_t red_ld = 0, red_msa_bytes = 0;
	std::vector<uint32_t> red_cnt_host;      // representatives per block of the last prep (RED_NONE: not reduced)
	std::vector<uint8_t> red_full;           // blocks this run sends to the kernel on all rows
	std::vector<uint8_t> red_force_full;     // ... because an earlier run on this input could not prove their lists on the representatives
	bool red_active = false;                 // this run's phase C went through the representatives (pass 2 follows it)
	uint32_t *d_red_cls = nullptr, *d_red_headd = nullptr, *d_red_ncls = nullptr, *d_red_taskblk = nullptr, *d_red_wgtasks = nullptr;
	size_t red_task_cap = 0;
	uint32_t *d_red_ss_a_alloc = nullptr, *d_red_ss_d_alloc = nullptr;
	uint32_t *d_red_ss_a = nullptr, *d_red_ss_d = nullptr;      // the reduced states phase C drops every red_ss_stride columns ([q][red_ss_cap])
	uint32_t red_ss_stride = 0, red_ss_cap = 0;
	size_t red_ss_words = 0;
	uint32_t *h_red_pin = nullptr;           // pinned host staging of the plan (counts back, block lists out): its own buffer, live across the run
	size_t red_pin_words = 0;
	struct RedBin { int config; uint32_t first, count; };     // blocks [first, first + count) of d_red_blocks run on configuration `config`
	std::vector<RedBin> red_bins;
	std::vector<int> red_config_of;          // [block] configuration of a reduced block's phase C (-1: not reduced)
	std::vector<int> red_config_snap_of;     // ... and of its sweeps in pass 2 (no lists: no list wave)
	uint32_t red_full_at = 0, red_nfull = 0;     // d_red_blocks[red_full_at .. + red_nfull): the blocks that run on all rows
	uint32_t red_listed = 0, red_max_rows = 0;   // d_red_blocks[0 .. red_listed): every reduced block; the most representatives among them
	bool red_direct = false;                 // the representatives' symbols are read from the alignment's own columns (LDS-resident row counts): no reduced alignment
	// the plan (which block on which configuration) of the last run on this input at capacity red_plan_X: the next run launches
	// by it without waiting for the counts, and checks on the device that they are what the plan was made from
	bool red_declined = false;               // the last run on this input (at capacity red_declined_X) found too many representatives: no prep this time
	uint32_t red_declined_X = 0;
	bool red_plan_valid = false;
	uint32_t red_plan_X = 0, red_plan_blocks = 0, red_plan_rows_mean = 0;
	uint32_t *d_red_cnt_plan = nullptr;
	hipStream_t red_st[2]{};                 // the configurations' launches side by side: the context's second stream, then these
	hipEvent_t red_ev[4]{};
	uint8_t *h_red_pin2 = nullptr;           // pass 2's task lists (pinned)
	size_t red_pin2_bytes = 0;

	// results
	bool have_result = false;
	fseq_result res{};
	uint4 *d_tb = nullptr;                   // the traceback kernels' output {entry, lb, key, size} per segment, window heads, counts
	size_t tb_cap = 0;
	uint32_t tb_win = 0;
	std::vector<fseq_dp_arg> traceback;
	std::vector<fseq_segment> segments;
	std::vector<uint32_t> sp_first, sp_len;
	fseq_timings tm{};
	hipEvent_t ev[8]{};
	hipStream_t stream2 = nullptr;           // the DP, while phase C is still producing lists for later columns
	hipEvent_t ev_part[16]{};                // part c of phase C done
	hipEvent_t ev_dp[2]{};                   // DP begin / end on stream2
	uint8_t *h_pin = nullptr;                // pinned host staging of a step's small transfers (pin_reserve / pin_take)
	size_t pin_cap = 0, pin_used = 0;
};


namespace fseq {

inline int fail(fseq_ctx *c, int code, char const *what, hipError_t e = hipSuccess)
{
	char buf[512];
	if (e != hipSuccess)
		snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
	else
		snprintf(buf, sizeof(buf), "%s", what);
	c->err = buf;
	return code;
}

// progress (include/fseq.h, fseq_set_progress): counters another thread may poll + the caller's callback
inline void progress(fseq_ctx *c, int stage, uint64_t current, uint64_t max)
{
	c->step_max.store(max, std::memory_order_relaxed);
	c->current_step.store(current, std::memory_order_relaxed);
	if (c->progress_fn) c->progress_fn(c->progress_user, stage, current, max);
}

#define HIP_TRY(c, expr)                                                   \
	do {                                                                   \
		hipError_t e_ = (expr);                                            \
		if (e_ != hipSuccess) return fail((c), FSEQ_E_HIP, #expr, e_);     \
	} while (0)

template <typename U>
inline void dev_free(fseq_ctx *c, U **p)
{
	if (!*p) return;
	auto it = c->alloc_sizes.find(static_cast<void *>(*p));
	if (it != c->alloc_sizes.end()) { c->alloc_total -= it->second; c->alloc_sizes.erase(it); }
	(void) hipFree(*p);
	*p = nullptr;
}

// every device allocation of a context goes through here: alloc_total is what the context holds (the memory plan of
// the stride states stays inside fseq_set_memory_budget's figure when ranks share a card)
template <typename U>
inline int dev_alloc(fseq_ctx *c, U **p, size_t count)
{
	dev_free(c, p);
	size_t const bytes = std::max<size_t>(count, 1) * sizeof(U);
	hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes);
	if (e != hipSuccess)
	{
		*p = nullptr;
		size_t free_b = 0, total_b = 0;
		(void) hipMemGetInfo(&free_b, &total_b);
		(void) hipGetLastError();          // the runtime remembers the failure: the checks behind later launches (of this or any
		                                   // other context of the thread) must not find it
		char what[160];
		snprintf(what, sizeof(what), "hipMalloc of %zu bytes (%zu of %zu bytes free on the device)", bytes, free_b, total_b);
		return fail(c, e == hipErrorOutOfMemory ? FSEQ_E_OOM : FSEQ_E_HIP, what, e);
	}
	c->alloc_sizes[static_cast<void *>(*p)] = bytes;
	c->alloc_total += bytes;
	return FSEQ_OK;
}

// items [lo, hi) of an array of `per` words per item: *alloc owns the memory, *view is shifted so that item i sits at
// *view + i * per (a rank of a sharded run holds its own column blocks only, addressed by their place in the whole alignment)
template <typename U>
inline int dev_alloc_range(fseq_ctx *c, U **alloc, U **view, size_t lo, size_t hi, size_t per)
{
	int const rc = dev_alloc(c, alloc, (hi > lo ? hi - lo : 0) * per);
	*view = rc ? nullptr : *alloc - lo * per;
	return rc;
}


} // namespace fseq
