#include "at_launch.h"
/* packed kernels, four groups of 16 lanes (8 alignments per wavefront) for reads of 209 .. 304 bases: K = 16 rows per lane
 * (256 rows: 250-base reads use 250 of 256 rows and 250 of 265 steps; two groups of 32 lanes x 8 rows: 250 of 281 steps, 4
 * alignments per wave) or K = 19 (304 rows: 300-base reads).  Pointers in the per-wave global slot only. */
template <int MODE, int K>
static at_sweep16_fn r3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 16, K, 4, true, true, false, false, AT_BITS16>;
	return at::at_sweep16<MODE, 16, K, 4, true, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn r2(int k, bool tb)
{
	switch (k) {
	case 16: return r3<MODE, 16>(tb);
	case 19: return r3<MODE, 19>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g16c)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;   /* no all-LDS and no all-HBM variant */
	switch (kmode) {
	case at::K_GLOBAL: return r2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return r2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return r2<at::K_FITJ>(k, tb);
	default: return r2<at::K_FIT>(k, tb);
	}
}
