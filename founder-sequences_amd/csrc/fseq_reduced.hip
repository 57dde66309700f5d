// fseq_reduced.hip -- [r5] the kernels of phase C / pass 2 on representative rows (fseq_reduced.hpp) and their launchers: a
// translation unit of its own, compiled beside fseq_api.hip (which reaches them through the function tables below).
#include "fseq_ctx.hpp"
#include "fseq_reduced.hpp"

namespace fseq {

namespace {

template <int T, int E, bool PK, bool EW>
struct LaunchRed {
	static size_t lds(uint32_t B, uint32_t symcap)
	{
		return columns_lds_bytes<T, E, 4, PK>(B) - 2 * carve_bytes((size_t) T * E, 1) + 2 * carve_bytes(symcap, 1);
	}
	static hipError_t prepare(size_t bytes) { return allow_lds(k_columns_red<T, E, 4, PK, EW>, bytes); }
	static void launch(hipStream_t st, uint32_t grid, size_t bytes, uint8_t const *msa, size_t ld, uint64_t n, uint32_t B, uint32_t L, uint32_t X, uint32_t stride,
	                   uint2 *ent, uint4 *hdr, uint32_t npass, uint32_t bsh, RedArgs const &red)
	{
		hipLaunchKernelGGL((k_columns_red<T, E, 4, PK, EW>), dim3(grid), dim3(T), bytes, st, msa, ld, n, B, L, X, stride, ent, hdr, npass, bsh, red);
	}
	static uint32_t resident(size_t bytes)
	{
		int nb = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_columns_red<T, E, 4, PK, EW>, T, bytes) != hipSuccess || nb < 1) nb = 1;
		return (uint32_t) nb;
	}
	static ReducedSet make() { return ReducedSet{(uint32_t) T, (uint32_t) E, EW ? (uint32_t) (T - 64) * E : (uint32_t) T * E, PK, EW, &lds, &prepare, &launch, &resident}; }
};

// ascending by the rows they hold.  EW: wave 0 holds no rows and emits the lists while the row waves are in the next column (from 256
// threads on: what phase C runs -- the lists of a block with thousands of representatives are a thousand entries long, and
// on a wave that also holds rows they were 40 % of a column of BASELINE C4 --; pass 2's sweeps, which emit none, take the others)
#define FSEQ_RED_CONFIGS(X) \
	X(64, 3, false, false) X(128, 3, false, true) X(64, 5, false, false) X(128, 5, false, true) X(64, 7, false, false) X(128, 7, false, true) \
	X(256, 3, false, true) X(256, 3, false, false) X(256, 5, false, true) X(256, 5, false, false) X(512, 5, false, true) X(512, 5, false, false) \
	X(512, 7, false, true) X(512, 7, false, false) X(1024, 5, false, true) X(1024, 5, false, false) X(1024, 7, true, true) X(1024, 7, true, false) \
	X(1024, 8, true, true) X(1024, 9, true, true) X(1024, 9, true, false) X(1024, 10, true, true) X(1024, 10, true, false) X(1024, 11, true, true) \
	X(1024, 11, true, false) X(1024, 12, true, true)

template <int T, int E, bool PK>
struct LaunchChainSnap {
	static hipError_t prepare() { return allow_lds(k_chain_snap<T, E, PK>, chain_snap_lds_bytes<T, E, PK>()); }
	static void launch(hipStream_t st, uint32_t grid, size_t bytes, uint32_t const *ba, uint32_t const *bd, uint32_t const *rank, uint32_t m, uint32_t const *task_blk,
	                   uint32_t const *cls, uint32_t const *headd, uint32_t const *ncls, uint32_t cap, uint32_t *snap_a, uint32_t *snap_d, uint32_t keyed)
	{
		hipLaunchKernelGGL((k_chain_snap<T, E, PK>), dim3(grid), dim3(T), bytes, st, ba, bd, rank, m, task_blk, cls, headd, ncls, cap, snap_a, snap_d, keyed);
	}
	static ChainSnapSet make() { return ChainSnapSet{chain_snap_lds_bytes<T, E, PK>(), &prepare, &launch}; }
};

// the base configurations of select_kernels (fseq_api.hip)
#define FSEQ_CHAIN_SNAP_CONFIGS(X) \
	X(64, 1, false) X(64, 7, false) X(256, 5, false) X(512, 5, false) X(512, 7, false) X(1024, 5, false) X(1024, 7, false) X(1024, 9, true) X(1024, 10, true) X(1024, 11, true)

} // namespace

int reduced_config_count()
{
	int n = 0;
#define X(T_, E_, PK_, EW_) ++n;
	FSEQ_RED_CONFIGS(X)
#undef X
	return n;
}

bool reduced_config(int index, ReducedSet *out)
{
	int i = 0;
#define X(T_, E_, PK_, EW_) if (i++ == index) { *out = LaunchRed<T_, E_, PK_, EW_>::make(); return true; }
	FSEQ_RED_CONFIGS(X)
#undef X
	return false;
}

bool select_chain_snap(uint32_t T, uint32_t E, ChainSnapSet *out)
{
#define X(T_, E_, PK_) if (T == T_ && E == E_) { *out = LaunchChainSnap<T_, E_, PK_>::make(); return true; }
	FSEQ_CHAIN_SNAP_CONFIGS(X)
#undef X
	return false;
}

hipError_t launch_reduce_prep(hipStream_t st, uint32_t grid, RedPrepArgs const &A)
{
	size_t const bytes = reduce_prep_lds_bytes(A.m);
	if (bytes > 160 * 1024) return hipErrorInvalidValue;
	hipError_t const e = allow_lds(k_reduce_prep, bytes);
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(k_reduce_prep, dim3(grid), dim3(RED_PREP_T), bytes, st, A);
	return hipSuccess;
}

void launch_reduce_check(hipStream_t st, uint32_t const *cnt, uint32_t const *planned, uint32_t count, uint32_t *flags)
{
	if (count) hipLaunchKernelGGL(k_reduce_check, dim3((count + 255u) / 256u), dim3(256), 0, st, cnt, planned, count, flags);
}

template <int BSH, int R>
static bool launch_reduce_msa_lds(hipStream_t st, uint32_t nlisted, uint8_t const *msa, size_t ld, uint8_t *red, size_t ldr, uint32_t const *cnt,
                                  uint32_t const *rows, uint32_t cap, uint64_t n, uint32_t B, uint32_t const *blocks, uint32_t colbytes)
{
	static bool prepared = false;
	size_t const lds = (size_t) 2 * R * 16384;
	if (!prepared)
	{
		if (hipFuncSetAttribute(reinterpret_cast<void const *>(&k_reduce_msa_lds<BSH, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds) != hipSuccess)
		{
			(void) hipGetLastError();
			return false;
		}
		prepared = true;
	}
	// (a quarter of a block's columns per workgroup: the rows are read four times, the blocks' columns are enough workgroups)
	hipLaunchKernelGGL((k_reduce_msa_lds<BSH, R>), dim3(nlisted, 4), dim3(1024), lds, st, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes);
	return true;
}

void launch_reduce_msa(hipStream_t st, uint32_t nlisted, uint32_t max_rows, uint8_t const *msa, size_t ld, uint8_t *red, size_t ldr, uint32_t const *cnt,
                       uint32_t const *rows, uint32_t cap, uint64_t n, uint32_t B, uint32_t bsh, uint32_t const *blocks, uint32_t m, bool gather_only)
{
	if (!nlisted) return;
	uint32_t const colbytes = sym_bytes(m, bsh);
	// the column through LDS where it fits two buffers of at most 32 KB and the representatives the registers of 1,024 threads
	if (!gather_only && colbytes <= 32768u && cap <= 12288u)
	{
		bool const one = colbytes <= 16384u;
		bool ok = false;
		switch (bsh)
		{
		case 0: ok = one ? launch_reduce_msa_lds<0, 1>(st, nlisted, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes) : launch_reduce_msa_lds<0, 2>(st, nlisted, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes); break;
		case 1: ok = one ? launch_reduce_msa_lds<1, 1>(st, nlisted, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes) : launch_reduce_msa_lds<1, 2>(st, nlisted, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes); break;
		case 2: ok = one ? launch_reduce_msa_lds<2, 1>(st, nlisted, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes) : launch_reduce_msa_lds<2, 2>(st, nlisted, msa, ld, red, ldr, cnt, rows, cap, n, B, blocks, colbytes); break;
		default: break;
		}
		if (ok) return;
	}
	uint32_t const nq = (max_rows + (1u << bsh) - 1u) >> bsh;
	hipLaunchKernelGGL(k_reduce_msa, dim3(nlisted, (nq + 63u) / 64u), dim3(256), 0, st, msa, ld, red, ldr, cnt, rows, cap, n, B, bsh, blocks);
}

} // namespace fseq
