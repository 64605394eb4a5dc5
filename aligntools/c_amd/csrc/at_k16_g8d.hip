#include "at_launch.h"
/* packed kernels, eight groups of 8 lanes (16 alignments per wavefront) for reads of up to 152 bases: K rows per lane =
 * ceil(l1 / 8) rounded up to one of {5, 6, 7, 8, 10, 13, 16, 19}.  150 x 150: 150 of 152 rows and 150 of 157 steps carry cells
 * (94 percent; four groups of 16 lanes x 10 rows: 85), and the per-step overhead is spread over 19 rows.  The pointer
 * matrix always lives in the per-wave global slots (16 alignments do not fit LDS); this unit: K in {6, 8} */
template <int MODE, int K>
static at_sweep16_fn h3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 8, K, 4, true, true, false, false, AT_BITS16>;
	return at::at_sweep16<MODE, 8, K, 4, true, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn h2(int k, bool tb)
{
	switch (k) {
	case 6: return h3<MODE, 6>(tb);
	case 8: return h3<MODE, 8>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g8d)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;   /* no all-LDS and no all-HBM variant */
	switch (kmode) {
	case at::K_GLOBAL: return h2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return h2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return h2<at::K_FITJ>(k, tb);
	default: return h2<at::K_FIT>(k, tb);
	}
}
