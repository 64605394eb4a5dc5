#include "at_launch.h"
/* packed kernels, two groups of 32 lanes (4 alignments per wavefront) for reads of 321..608 bases: K = 12 (384 rows),
 * 16 (512) or 19 (608) rows per lane in ONE strip -- a 64-lane group would sweep reads of 417..512 bases in two strips of 256
 * rows and those of 513..608 in three, two alignments per wave.  Pointers in the per-wave global slot only. */
template <int MODE, int K>
static at_sweep16_fn s3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 32, K, 4, true, true, false, false, AT_BITS16>;
	return at::at_sweep16<MODE, 32, K, 4, true, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn s2(int k, bool tb)
{
	switch (k) {
	case 12: return s3<MODE, 12>(tb);
	case 16: return s3<MODE, 16>(tb);
	case 19: return s3<MODE, 19>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_g32b)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;   /* no all-LDS and no all-HBM variant */
	switch (kmode) {
	case at::K_GLOBAL: return s2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return s2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return s2<at::K_FITJ>(k, tb);
	default: return s2<at::K_FIT>(k, tb);
	}
}
