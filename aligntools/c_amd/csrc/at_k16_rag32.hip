#include "at_launch.h"
/* packed kernels for RAGGED batches of reads of 305..608 bases: two groups of 32 lanes (4 alignments per wavefront), one strip of
 * 32 x K rows -- K = 10 (320), 12 (384), 13 (416) for every mode, 16 (512) for local and global, 19 (608) for local (the classes
 * of the uniform kernels, at_k16_g32*.hip).  Frames as in at_k16_rag16b.hip: a work item sweeps the extents its alignments need,
 * every alignment keeps its own.  The pointer matrix lives in the per-wave global slots. */
template <int MODE, int K>
static at_sweep16_fn y3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 32, K, 4, true, true, false, true, AT_BITS16>;
	return at::at_sweep16<MODE, 32, K, 4, true, false, true, true, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn y2(int k, bool tb)
{
	switch (k) {
	case 10: return y3<MODE, 10>(tb);
	case 12: return y3<MODE, 12>(tb);
	case 13: return y3<MODE, 13>(tb);
	case 16: if constexpr (MODE == at::K_LOCAL || MODE == at::K_GLOBAL) return y3<MODE, 16>(tb); else return nullptr;
	case 19: if constexpr (MODE == at::K_LOCAL) return y3<MODE, 19>(tb); else return nullptr;
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_rag32)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;
	switch (kmode) {
	case at::K_GLOBAL: return y2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return y2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return y2<at::K_FITJ>(k, tb);
	default: return y2<at::K_FIT>(k, tb);
	}
}
/* ragged packed overlap (alignment.h:926-964 with tracebacks): one 64-lane group, two alignments of equal l1 per wavefront, 4 or
 * 16 rows per lane in one strip (reads of up to 256 / 1 024 bases) */
at_sweep16_fn AT_NAME(at_pick16_ragovl)(int k, int store)
{
	if (store != 1) return nullptr;
	if (k == 16) return at::at_sweep16<at::K_OVERLAP, 64, 16, 2, true, false, true, true, AT_BITS16>;
	if (k == 4) return at::at_sweep16<at::K_OVERLAP, 64, 4, 2, true, false, true, true, AT_BITS16>;
	return nullptr;
}
