#include "at_launch.h"
/* packed kernels for RAGGED batches, eight groups of 8 lanes (at_sweep16.hip.h, RAG): every work item sweeps the frame its
 * 16 alignments need: the global / fit alignments of one item share their l1 (the host puts reads of equal length together)
 * and keep their own l2.  This unit: K in {6, 8} */
template <int MODE, int K>
static at_sweep16_fn v3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 8, K, 4, true, true, false, true, AT_BITS16>;
	return at::at_sweep16<MODE, 8, K, 4, true, false, true, true, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn v2(int k, bool tb)
{
	switch (k) {
	case 6: return v3<MODE, 6>(tb);
	case 8: return v3<MODE, 8>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_rag8d)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;   /* no all-LDS and no all-HBM variant */
	switch (kmode) {
	case at::K_GLOBAL: return v2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return nullptr;               /* ragged local batches: the 16-lane groups (at_k16_rag.hip) */
	case at::K_FITJ: return v2<at::K_FITJ>(k, tb);
	default: return v2<at::K_FIT>(k, tb);
	}
}
