#include "at_launch.h"
/* packed kernels for RAGGED batches of reads of 209..304 bases: four groups of 16 lanes, 16 or 19 rows per lane -- global /
 * fit (work items of one read length) and local (frames that mix lengths) */
template <int MODE, int K>
static at_sweep16_fn x3(bool tb)
{
	if (!tb) return at::at_sweep16<MODE, 16, K, 4, true, true, false, true, AT_BITS16>;
	return at::at_sweep16<MODE, 16, K, 4, true, false, true, true, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn x2(int k, bool tb)
{
	switch (k) {
	case 16: return x3<MODE, 16>(tb);
	case 19: return x3<MODE, 19>(tb);
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_rag16b)(int kmode, int k, int store, bool tb)
{
	if (tb && store != 1) return nullptr;
	switch (kmode) {
	case at::K_GLOBAL: return x2<at::K_GLOBAL>(k, tb);
	case at::K_LOCAL: return x2<at::K_LOCAL>(k, tb);
	case at::K_FITJ: return x2<at::K_FITJ>(k, tb);
	default: return x2<at::K_FIT>(k, tb);
	}
}
