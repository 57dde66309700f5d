// fseq_types.hpp -- the few plain types the context (fseq_ctx.hpp) holds by value or names in a signature, apart from the
// kernel headers that use them.
#pragma once

#include <stdint.h>

namespace fseq {

// Phase D's arrays (fseq_dp.hpp): M (segment_max_size = the key), LB, SZ per DP entry; K[t]: 64-bit stack mask over the
// 64-block of t; Tb[p][j] / Tbv[p][j]: sparse-table sample (index / key) of rmq.hh's m_precalc[p][j].
struct DpArrays {
	uint32_t *M, *LB, *SZ, *Tb, *Tbv;
	unsigned long long *K;
	uint32_t tstride;
};

struct S2SnapArgs;                   // fseq_stream2.hpp: the boundaries of a pass-2 launch on the streamed tile step

// [r5] Phase C on a block's REPRESENTATIVE rows (fseq_reduced.hpp): what k_reduce_prep left for every block and where a
// workgroup of the reduced column kernel finds it.  Plain pointers into device memory.
constexpr uint32_t RED_NONE = 0xFFFFFFFFu;      // cnt[b]: block b is not reduced (more representatives than the kernel holds)
struct RedArgs {
	uint32_t const *cnt = nullptr;      // [block] representatives of the block
	uint32_t const *vmin = nullptr;     // [block] the values >= vmin are those of the run on all rows (1: every value is)
	uint32_t const *a = nullptr;        // [block][cap] start state: representative index (place among the block's representatives by row id) ...
	uint32_t const *d = nullptr;        // [block][cap] ... and the maximum of d0 over the positions skipped since the last kept row
	uint32_t const *leaf = nullptr;     // [block][cap] block-key rank of representative i (pass 2)
	uint32_t const *blocks = nullptr;   // [workgroup] block of workgroup i of the launch
	uint32_t *invalid = nullptr;        // [block] set when a list of the block took an entry the representatives cannot vouch for
	uint32_t cap = 0;                   // row stride of a / d / leaf (and of cls / headd)
	uint32_t m_true = 0;                // rows of the alignment
	uint32_t direct = 0;                // 1: a[] holds ROW IDS and msa / ld are the alignment itself (its whole column is staged: colbytes);
	                                    // 0: a[] holds representative indices and msa / ld are the reduced alignment (k_reduce_msa)
	uint32_t colbytes = 0;              // direct: bytes of a packed column of the alignment
	uint32_t symcap = 0;                // bytes of each of the two staged-column buffers in LDS (whole kilobytes)
	uint32_t const *rank = nullptr;     // direct, pass 2: [block][m_true] block-key rank of every row (leaf[] is not used)
	uint32_t *any_invalid = nullptr;    // one word: set with invalid[b]
	// pass 2: instead of lists, the class tables at the task columns of the block
	uint32_t const *wg_tasks = nullptr; // [workgroup][3] {first task, tasks, column of the start state (the block's first column, or a stride state's)}
	// the reduced states phase C drops every ss_stride columns (the state at column q * ss_stride at [q][ss_cap], where that
	// column lies strictly inside a block): where pass 2's sweeps start from
	uint32_t *ss_a = nullptr, *ss_d = nullptr;
	uint32_t ss_stride = 0, ss_cap = 0;
	unsigned long long const *task_rb = nullptr;   // [task] column (ascending inside a workgroup)
	uint32_t *cls = nullptr;            // [task][cap] class (rank among the distinct key prefixes) of every block key
	uint32_t *headd = nullptr;          // [task][cap] divergence in front of every class
	uint32_t *ncls = nullptr;           // [task]
};

// what k_reduce_prep reads and leaves (fseq_reduced.hpp)
struct RedPrepArgs {
	uint32_t const *bstate_a, *bstate_d;     // [blocks + 1][m]: the exact states in front of the blocks and behind the last
	uint32_t const *rank;                    // [blocks][m]: block-key rank of every row (phase A)
	uint32_t const *blocks;                  // [workgroup] block of workgroup i, or nullptr: block0 + i
	uint32_t m, B, L, Xp, cap, block0, leaf_only, direct;      // direct: a[] = row ids (else: indices among the block's representatives)
	uint32_t *invalid, *flags;               // invalid[b] = 0; flags[0 .. 1] = 0 (workgroup 0): what the column kernels and the plan check set
	uint64_t n;
	uint32_t *cnt, *vmin, *rows, *leaf, *a, *d;
};

} // namespace fseq
