#include "at_launch.h"
/* packed kernels, sixteen groups of 4 lanes (32 alignments per wavefront): K = 16 or 19 rows per lane for reads of 53..76
 * bases (75-base reads: 75 of 76 rows, 75 of 78 steps; eight groups of 8 lanes x 10 rows: 75 of 80 rows, 75 of 82 steps) */
template <int MODE, int K>
static at_sweep16_fn f3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 4, K, 4, true, true, false, false, AT_BITS16>;
	return at::at_sweep16<MODE, 4, K, 4, true, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn f2(int k, bool tb)
{
	switch (k) {
	case 16: return f3<MODE, 16>(tb);
	case 19: return f3<MODE, 19>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g4b)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;   /* no all-LDS and no all-HBM variant */
	switch (kmode) {
	case at::K_GLOBAL: return f2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return f2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return f2<at::K_FITJ>(k, tb);
	default: return f2<at::K_FIT>(k, tb);
	}
}
