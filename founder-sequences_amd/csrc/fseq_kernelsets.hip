// fseq_kernelsets.hip -- the kernel configurations of the path and their launchers: the LDS-resident column / chain / rank
// kernels by row count (select_kernels); phase A's key-space tree and trie by workgroup size and the streamed phase C's tile
// configurations (select_stream2) are csrc/fseq_kernelsets_stream.hip.  A translation unit of its own since round 5 (the review's "split fseq_api.hip"): these
// template instantiations were two thirds of what a rebuild of csrc/fseq_api.hip cost, and nothing in the orchestration
// touches them but through the function tables of fseq_ctx.hpp (KernelSet, Stream2Config).
#include "fseq_ctx.hpp"
#include "fseq_kernels.hpp"

#include <cstdio>

namespace fseq {

namespace {

// EW: phase C keeps wave 0 free of rows for the per-column list (k_columns, fseq_kernels.hpp): m <= (T - 64) * E
template <int T, int E, int SIGMA, bool PK, bool EW = false>
struct Launch {
	static void rank(hipStream_t st, uint32_t grid, size_t lds, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
	                 uint32_t nblocks, uint32_t npass, uint32_t bsh, uint32_t *rank_, uint32_t *keyd, uint32_t *nkeys, uint64_t col0, uint32_t const *only)
	{
		// (only: per-block filter, passed in the start-state slot the rank mode does not use -- k_colblock)
		hipLaunchKernelGGL((k_colblock<T, E, SIGMA, MODE_RANK, PK>), dim3(grid), dim3(T), lds, st, msa, ld, m, n, B, nblocks, npass, bsh, rank_, keyd, nkeys,
		                   only, (uint32_t const *) nullptr, (uint64_t const *) nullptr, (uint2 const *) nullptr,
		                   (uint32_t *) nullptr, (uint32_t *) nullptr, (uint64_t const *) nullptr, 0u, (uint32_t const *) nullptr, (uint32_t const *) nullptr, col0,
		                   (uint64_t) B < (1ull << scan_shift_for(T, E)) ? 1u : 0u);      // (divergences relative to the block start: <= B)
	}
	static void snap(hipStream_t st, uint32_t grid, size_t lds, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
	                 uint32_t nblocks, uint32_t npass, uint32_t bsh, uint32_t const *ba, uint32_t const *bd, uint64_t const *rb, uint2 const *grp, uint32_t *sa, uint32_t *sd,
	                 uint64_t const *task_src, uint32_t snap_stride, uint32_t const *ss_a, uint32_t const *ss_d, uint32_t keyed)
	{
		hipLaunchKernelGGL((k_colblock<T, E, SIGMA, MODE_SNAP, PK>), dim3(grid), dim3(T), lds, st, msa, ld, m, n, B, nblocks, npass, bsh,
		                   (uint32_t *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, ba, bd, rb, grp, sa, sd, task_src, snap_stride, ss_a, ss_d, (uint64_t) 0, keyed);
	}
	static size_t columns_lds(uint32_t B) { return columns_lds_bytes<T, E, SIGMA, PK>(B); }
	static void columns(hipStream_t st, uint32_t grid, size_t lds, uint8_t const *msa, size_t ld, uint32_t m, uint64_t n, uint32_t B,
	                    uint32_t N2, uint32_t const *ba, uint32_t const *bd, uint32_t L, uint32_t X, uint32_t stride, uint2 *ent, uint4 *hdr, uint32_t npass, uint32_t bsh,
	                    uint32_t snap_stride, uint32_t *ss_a, uint32_t *ss_d, uint32_t block0, uint32_t *done_host, uint32_t epoch, uint32_t const *colmask, uint32_t const *blocklist)
	{
		if (colmask)
			hipLaunchKernelGGL((k_columns<T, E, SIGMA, PK, EW, true>), dim3(grid), dim3(T), lds, st, msa, ld, m, n, B, N2, ba, bd, L, X, stride, ent, hdr, npass, bsh,
			                   snap_stride, ss_a, ss_d, block0, done_host, epoch, colmask, blocklist);
		else
			hipLaunchKernelGGL((k_columns<T, E, SIGMA, PK, EW, false>), dim3(grid), dim3(T), lds, st, msa, ld, m, n, B, N2, ba, bd, L, X, stride, ent, hdr, npass, bsh,
			                   snap_stride, ss_a, ss_d, block0, done_host, epoch, (uint32_t const *) nullptr, blocklist);
	}
	static uint32_t columns_resident(size_t lds)
	{
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_columns<T, E, SIGMA, PK, EW>, T, lds) != hipSuccess || nb < 1) nb = 1;
		return (uint32_t) nb;
	}
	static void chain(hipStream_t st, uint32_t grid, size_t lds, uint32_t const *rank_, uint32_t const *keyd, uint32_t const *nkeys, uint32_t m,
	                  uint32_t nb_total, uint32_t G, uint64_t cols_per_block, uint32_t const *start_a, uint32_t const *start_d,
	                  uint32_t *out_a, uint32_t *out_d, uint32_t *out_rank, uint32_t *out_keyd, uint32_t *out_nkeys, uint32_t grp0, uint32_t keyed)
	{
		hipLaunchKernelGGL((k_chain<T, E, PK>), dim3(grid), dim3(T), lds, st, rank_, keyd, nkeys, m, nb_total, G, cols_per_block,
		                   start_a, start_d, out_a, out_d, out_rank, out_keyd, out_nkeys, grp0, keyed);
	}
	static hipError_t prepare(size_t lds_columns)
	{
		hipError_t e;
		if ((e = allow_lds(k_colblock<T, E, SIGMA, MODE_RANK, PK>, colblock_lds_bytes<T, E, SIGMA, MODE_RANK, PK>())) != hipSuccess) return e;
		if ((e = allow_lds(k_colblock<T, E, SIGMA, MODE_SNAP, PK>, colblock_lds_bytes<T, E, SIGMA, MODE_SNAP, PK>())) != hipSuccess) return e;
		if ((e = allow_lds(k_chain<T, E, PK>, chain_lds_bytes<T, E, PK>())) != hipSuccess) return e;
		(void) lds_columns;
		return hipSuccess;
	}
	static hipError_t prepare_columns(size_t lds_columns)
	{
		hipError_t const e = allow_lds(k_columns<T, E, SIGMA, PK, EW, false>, lds_columns);
		return e != hipSuccess ? e : allow_lds(k_columns<T, E, SIGMA, PK, EW, true>, lds_columns);
	}
	static KernelSet make()
	{
		KernelSet k;
		k.T = T; k.E = E; k.sigma = SIGMA; k.cap = T * E; k.scan_shift = (uint32_t) scan_shift_for(T, E);
		k.lds_colblock = colblock_lds_bytes<T, E, SIGMA, MODE_RANK, PK>();
		k.lds_snap = colblock_lds_bytes<T, E, SIGMA, MODE_SNAP, PK>();
		k.rank = &rank; k.snap = &snap; k.columns_lds = &columns_lds; k.columns = &columns; k.columns_resident = &columns_resident;
		k.lds_chain = chain_lds_bytes<T, E, PK>();
		k.chain = &chain; k.prepare = &prepare; k.prepare_columns = &prepare_columns;
		return k;
	}
};

// phase C from another configuration than phases A, B and pass 2 (the emitter-wave kernels give their threads one
// row more; the latency-bound chain and snapshot kernels are better off without it)
template <typename Base, typename Col>
KernelSet compose_kernels()
{
	KernelSet k = Base::make();
	k.columns_lds = &Col::columns_lds; k.columns = &Col::columns; k.columns_resident = &Col::columns_resident; k.prepare_columns = &Col::prepare_columns;
	return k;
}

} // namespace

bool select_kernels(uint32_t m, uint32_t sigma, KernelSet *out, bool no_emitter_wave)
{
	if (sigma > 256) return false;
	// 1024-thread configurations spare wave 0 for the per-column lists when the rows allow it (measured: C5 phase C
	// 86 -> 75 ms with it, while 512-thread workgroups lose as much to the longer per-thread chunks as they gain)
	bool const ew_ok = !no_emitter_wave;
#define FSEQ_TRY_EW(T_, E_, PK_)                                                               \
	if (ew_ok && m <= (uint32_t) ((T_) - 64) * (E_))                                           \
	{                                                                                          \
		*out = Launch<T_, E_, 4, PK_, true>::make();                                            \
		return true;                                                                           \
	}
#define FSEQ_TRY(T_, E_, PK_)                                                                  \
	if (m <= (uint32_t) (T_) * (E_))                                                           \
	{                                                                                          \
		*out = Launch<T_, E_, 4, PK_>::make();                                                  \
		return true;                                                                           \
	}
	FSEQ_TRY(64, 1, false)
	FSEQ_TRY(64, 7, false)
	FSEQ_TRY(256, 5, false)
	// 512 threads: the list wave pays since the partition step's scan became cheap (BASELINE C3: phase C 8.2 -> 7.7 ms
	// with six rows on seven waves and the list on the eighth; 576 threads would keep five rows per thread, but nine
	// waves per workgroup place three on one SIMD and only one workgroup fits a CU)
	// [late r3] ... with 16-bit LDS words (value ids are < m + B < 65536 anyway): 48 KiB instead of 70 per workgroup and 80
	// registers by launch bounds put THREE workgroups on a CU -- a column step is a chain of three barriers and ~6 LDS round
	// trips, and two workgroups left the SIMDs idle 43 % of the time (BASELINE C3: phase C 5.61 -> 5.08 ms; the unpacking
	// costs less than the third workgroup brings)
	if (ew_ok && m > 448u * 5u && m <= 448u * 6u && m <= 512u * 5u)
	{
		*out = compose_kernels<Launch<512, 5, 4, false>, Launch<512, 6, 4, true, true>>();
		return true;
	}
	if (ew_ok && m > 256u * 5u && m <= 448u * 5u)
	{
		*out = compose_kernels<Launch<512, 5, 4, false>, Launch<512, 5, 4, true, true>>();
		return true;
	}
	FSEQ_TRY(512, 5, false)
	FSEQ_TRY(512, 7, false)
	FSEQ_TRY_EW(1024, 5, false)    // (3,585 .. 5,120 rows: e.g. the 5,008 haplotypes of 2,504 diploid samples)
	FSEQ_TRY(1024, 5, false)
	FSEQ_TRY_EW(1024, 7, false)
	FSEQ_TRY(1024, 7, false)
	// 16-bit LDS state (m <= 11,264).  These kernels want more registers than a wave of a 1024-thread workgroup gets
	// (~13 per row of E): the fewest rows per thread that hold m, the list wave where the same E allows it
	// (BASELINE C5, m = 10,000: (1024,10) 56.9 ms of phase C against 61.9 with (1024,11) and the list wave)
	FSEQ_TRY_EW(1024, 9, true)
	FSEQ_TRY(1024, 9, true)
	FSEQ_TRY_EW(1024, 10, true)
	FSEQ_TRY(1024, 10, true)
	FSEQ_TRY_EW(1024, 11, true)
	FSEQ_TRY(1024, 11, true)
#undef FSEQ_TRY
#undef FSEQ_TRY_EW
	return false;
}


} // namespace fseq
