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

} // namespace fseq
