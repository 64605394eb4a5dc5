#include "at_launch.h"
/* packed kernels, four groups of 16 lanes (8 alignments per wavefront): K rows per lane = ceil(l1 / 16) rounded up to
 * one of {4,5,6,7,10,13}; the pointer matrix always lives in the per-wave global slot (8 alignments do not fit LDS) */
template <int MODE, int K>
static at_sweep16_fn q3(int store, bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 16, K, 4, true, true, false, false, AT_BITS16>;
	if (store == 0) return at::at_sweep16<MODE, 16, K, 4, true, true, true, false, AT_BITS16>;
	return at::at_sweep16<MODE, 16, K, 4, true, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn q2(int k, int store, bool tb)
{
	switch (k) {
	case 10: return q3<MODE, 10>(store, tb);
	case 13: return q3<MODE, 13>(store, tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g16b)(int kmode, int k, int store, bool tb)
{
	switch (kmode) {
	case at::K_GLOBAL: return q2<at::K_GLOBAL>(k, store, tb);
	case at::K_LOCAL: return q2<at::K_LOCAL>(k, store, tb);
	case at::K_FITJ: return q2<at::K_FITJ>(k, store, tb);
	default: return q2<at::K_FIT>(k, store, tb);
	}
}
