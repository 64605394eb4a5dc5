#include "at_launch.h"
/* packed kernels, two groups of 32 lanes (4 alignments per wavefront) for reads of 209..416 bases: K rows per lane =
 * ceil(l1 / 32) rounded up to one of {7, 8, 10, 13}; pointer matrix in the per-wave global slot */
template <int MODE, int K>
static at_sweep16_fn g3(int store, bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 32, K, 4, true, true, false, false, AT_BITS16>;
	if (store == 0) return at::at_sweep16<MODE, 32, K, 4, true, true, true, false, AT_BITS16>;
	return at::at_sweep16<MODE, 32, K, 4, true, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn g2(int k, int store, bool tb)
{
	switch (k) {
	case 7: return g3<MODE, 7>(store, tb);
	case 8: return g3<MODE, 8>(store, tb);
	case 10: return g3<MODE, 10>(store, tb);
	case 13: return g3<MODE, 13>(store, tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g32b)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g32)(int kmode, int k, int store, bool tb)
{
	if (k == 12 || k >= 16) return AT_NAME(at_pick16_g32b)(kmode, k, store, tb);
	switch (kmode) {
	case at::K_GLOBAL: return g2<at::K_GLOBAL>(k, store, tb);
	case at::K_LOCAL: return g2<at::K_LOCAL>(k, store, tb);
	case at::K_FITJ: return g2<at::K_FITJ>(k, store, tb);
	default: return g2<at::K_FIT>(k, store, tb);
	}
}
