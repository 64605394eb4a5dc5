#include "at_launch.h"
/* packed kernels for RAGGED batches of global / fit alignments of reads of 153..208 bases: four groups of 16 lanes, 10 or 13
 * rows per lane (shorter reads use the 8-lane groups, at_k16_rag8*.hip; ragged local batches at_k16_rag.hip) */
template <int MODE, int K>
static at_sweep16_fn w3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 16, K, 4, true, true, false, true, AT_BITS16>;
	return at::at_sweep16<MODE, 16, K, 4, true, false, true, true, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn w2(int k, bool tb)
{
	switch (k) {
	case 10: return w3<MODE, 10>(tb);
	case 13: return w3<MODE, 13>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_rag16)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;
	switch (kmode) {
	case at::K_GLOBAL: return w2<at::K_GLOBAL>(k, tb);
	case at::K_FITJ: return w2<at::K_FITJ>(k, tb);
	case at::K_FIT: return w2<at::K_FIT>(k, tb);
	default: return nullptr;   /* local: at_k16_rag.hip */
	}
}
