// fseq_kernelsets_stream.hip -- launchers by configuration, second part (csrc/fseq_kernelsets.hip): phase A's key-space tree
// (k_blockkeys) and trie (k_blocktrie) by workgroup size and bits per symbol, the streamed phase C's tile configurations
// (k_columns_stream2, select_stream2).
#include "fseq_ctx.hpp"
#include "fseq_kernels.hpp"
#include "fseq_stream.hpp"
#include "fseq_stream2.hpp"
#include "fseq_blockkeys.hpp"
#include "fseq_blocktrie.hpp"

#include <cstdio>

namespace fseq {

namespace {

// phase C, streamed rows, second form (fseq_stream2.hpp): <threads, rows per thread, 5-byte rows>
#define FSEQ_S2_CONFIGS(X) X(512, 8, true) X(1024, 4, true) X(1024, 8, true) X(256, 8, true) X(256, 12, true) X(512, 8, false) X(1024, 6, false) X(1024, 8, false) X(256, 8, false) X(256, 12, false)
template <int T, int E, bool PACK>
struct LaunchS2 {
	static size_t lds(uint32_t colbytes) { return stream2_lds_bytes<T, E, PACK>(colbytes); }
	static hipError_t prepare(size_t bytes)
	{
		hipError_t const e = allow_lds(k_columns_stream2<T, E, PACK>, bytes);
		if (e != hipSuccess) return e;
		if constexpr (PACK) return allow_lds(k_columns_stream2<T, E, PACK, S2_SNAP>, bytes);
		return hipSuccess;
	}
	static void launch_snap(hipStream_t st, uint32_t grid, size_t bytes, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t npass, uint32_t bsh, uint32_t *ws,
	                        uint32_t snap_stride, uint32_t *ss_a, uint32_t *ss_d, S2SnapArgs const &SN)
	{
		if constexpr (PACK)
			hipLaunchKernelGGL((k_columns_stream2<T, E, PACK, S2_SNAP>), dim3(grid), dim3(T), bytes, st, msa, ld, m, n, B, npass, bsh, ws, 0u, 0u, 0u, (uint2 *) nullptr, (uint4 *) nullptr,
			                   snap_stride, ss_a, ss_d, 0u, (uint32_t *) nullptr, 0u, 0u, SN);
	}
	static void launch(hipStream_t st, uint32_t grid, size_t bytes, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B, uint32_t npass, uint32_t bsh, uint32_t *ws,
	                   uint32_t L, uint32_t X, uint32_t stride, uint2 *ent, uint4 *hdr, uint32_t snap_stride, uint32_t *ss_a, uint32_t *ss_d, uint32_t block0, uint32_t *done, uint32_t epoch, uint32_t ss_pack,
	                   uint32_t const *blocklist)
	{
		S2SnapArgs SN{};
		SN.wg_block = blocklist;
		hipLaunchKernelGGL((k_columns_stream2<T, E, PACK>), dim3(grid), dim3(T), bytes, st, msa, ld, m, n, B, npass, bsh, ws, L, X, stride, ent, hdr, snap_stride, ss_a, ss_d, block0, done, epoch, ss_pack, SN);
	}
	static uint32_t resident(size_t bytes)
	{
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_columns_stream2<T, E, PACK>, T, bytes) != hipSuccess || nb < 1) nb = 1;
		return (uint32_t) nb;
	}
	static Stream2Config make() { return Stream2Config{(uint32_t) T, (uint32_t) E, (uint32_t) s2_key_shift(T * E), PACK ? 1u : 0u, &lds, &prepare, &launch, &resident, PACK ? &launch_snap : nullptr}; }
};
} // namespace

// phase A in key space, LDS-resident rows (fseq_blockkeys.hpp): the kernel has its own workgroup size, one thread
// per 8 rows where that fits (blockkeys_threads)
#define FSEQ_BK_SIZES(X) X(256) X(320) X(512) X(768) X(1024)
void launch_blockkeys(uint32_t T, hipStream_t st, uint32_t grid, size_t lds, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
                      uint32_t bsh, uint32_t *rank_, uint32_t *keyd, uint32_t *nkeys, uint64_t col0,
                      uint16_t *scratch, size_t scratch_per_block, uint32_t cap_words, uint32_t *sliced, uint32_t *todo, uint32_t const *only)
{
	switch (T)
	{
#define X(T_) case T_: hipLaunchKernelGGL((k_blockkeys<T_>), dim3(grid), dim3(T_), lds, st, msa, ld, m, n, B, bsh, rank_, keyd, nkeys, col0, \
	                                          scratch, scratch_per_block, cap_words, sliced, todo, only); break;
		FSEQ_BK_SIZES(X)
#undef X
		default: break;
	}
}
hipError_t prepare_blockkeys(uint32_t T, size_t lds, bool debug)
{
	if (debug)
	{
		int nb = -1;
		switch (T)
		{
#define X(T_) case T_: (void) allow_lds(k_blockkeys<T_>, lds); (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_blockkeys<T_>, T_, lds); break;
			FSEQ_BK_SIZES(X)
#undef X
			default: break;
		}
		fprintf(stderr, "fseq: k_blockkeys<%u> with %zu bytes of LDS: %d workgroups per CU\n", T, lds, nb);
	}
	switch (T)
	{
#define X(T_) case T_: return allow_lds(k_blockkeys<T_>, lds);
		FSEQ_BK_SIZES(X)
#undef X
		default: return hipErrorInvalidValue;
	}
}

// phase A, the trie over 32-bit group words (fseq_blocktrie.hpp): T threads by the row count (12 T classes fit), bits per symbol
uint32_t blocktrie_threads(uint32_t m, bool stream) { return stream || m > 12u * 512u ? 1024u : m > 12u * 256u ? 512u : 256u; }
size_t blocktrie_lds(uint32_t T) { return T == 256u ? BtGeom<256>::LDS_BYTES : T == 512u ? BtGeom<512>::LDS_BYTES : BtGeom<1024>::LDS_BYTES; }
hipError_t launch_blocktrie(uint32_t bits, uint32_t T, hipStream_t st, uint32_t groups, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
                            uint32_t nblk, uint32_t *rank_, uint32_t *keyd, uint32_t *nkeys, uint64_t col0, uint32_t *ws, size_t per, uint32_t *given_up, uint32_t *todo)
{
#define FSEQ_BT_CASE(BITS_, T_) \
	if (bits == BITS_ && T == T_) \
	{ \
		hipError_t const e = allow_lds(k_blocktrie<BITS_, T_>, BtGeom<T_>::LDS_BYTES); \
		if (e != hipSuccess) return e; \
		hipLaunchKernelGGL((k_blocktrie<BITS_, T_>), dim3(groups), dim3(T_), BtGeom<T_>::LDS_BYTES, st, msa, ld, m, n, B, nblk, rank_, keyd, nkeys, col0, ws, per, given_up, todo); \
		return hipSuccess; \
	}
	FSEQ_BT_CASE(2, 256) FSEQ_BT_CASE(2, 512) FSEQ_BT_CASE(2, 1024)
	FSEQ_BT_CASE(4, 256) FSEQ_BT_CASE(4, 512) FSEQ_BT_CASE(4, 1024)
	FSEQ_BT_CASE(8, 256) FSEQ_BT_CASE(8, 512) FSEQ_BT_CASE(8, 1024)
#undef FSEQ_BT_CASE
	return hipErrorInvalidValue;
}

bool select_stream2(uint32_t T, uint32_t E, uint32_t pack, Stream2Config *out)
{
#define X(T_, E_, P_) if (T == T_ && E == E_ && (pack != 0) == P_) { *out = LaunchS2<T_, E_, P_>::make(); return true; }
	FSEQ_S2_CONFIGS(X)
#undef X
	return false;
}



} // namespace fseq
