#include "at_launch.h"
/* packed kernels, sixteen groups of 4 lanes (32 alignments per wavefront) for reads of up to 52 bases: K = 9, 10 or 13 rows per
 * lane (36 / 40 / 52 rows).  36 x 36: all 36 rows and 36 of 39 steps carry cells, the per-step overhead is spread over 9 rows
 * (eight groups of 8 lanes x 5 rows: 36 of 40 rows, 36 of 43 steps, 5 rows).  Pointers in the per-wave global slot only. */
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
	case 9: return f3<MODE, 9>(tb);
	case 10: return f3<MODE, 10>(tb);
	case 13: return f3<MODE, 13>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g4b)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g4)(int kmode, int k, int store, bool tb)
{
	if (k >= 16) return AT_NAME(at_pick16_g4b)(kmode, k, store, tb);
	if (tb && store != 1) return nullptr;   /* no all-LDS and no all-HBM variant */
	switch (kmode) {
	case at::K_GLOBAL: return f2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return f2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return f2<at::K_FITJ>(k, tb);
	default: return f2<at::K_FIT>(k, tb);
	}
}
