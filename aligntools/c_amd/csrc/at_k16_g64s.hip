#include "at_launch.h"
/* packed kernels, one group of 64 lanes, K in 1..4 rows per lane, scores scaled by 4 (TS = 2) */
template <int MODE, int K>
static at_sweep16_fn p3(int store, bool tb)
{
	if (!tb) return store < 2 ? at::at_sweep16<MODE, 64, K, 2, true, true, false, false, AT_BITS16> : at::at_sweep16<MODE, 64, K, 2, false, false, false, false, AT_BITS16>;
	if (store == 0) return at::at_sweep16<MODE, 64, K, 2, true, true, true, false, AT_BITS16>;
	if (store == 1) return at::at_sweep16<MODE, 64, K, 2, true, false, true, false, AT_BITS16>;
	return at::at_sweep16<MODE, 64, K, 2, false, false, true, false, AT_BITS16>;
}
template <int MODE>
static at_sweep16_fn p2(int k, int store, bool tb)
{
	switch (k) {
	case 1: return p3<MODE, 1>(store, tb);
	case 2: return p3<MODE, 2>(store, tb);
	case 3: return p3<MODE, 3>(store, tb);
	case 16:   /* scores only: no pointer registers, so 16 rows per lane still leave 3 waves per SIMD, and 1 024 rows are one strip
	            * (C3 scores only 5.6 -> 6.5 TCUPS); with pointers the same geometry needs 256 VGPRs (1.8 instead of 3.0 TCUPS) */
		if (tb) return nullptr;
		return store < 2 ? at::at_sweep16<MODE, 64, 16, 2, true, true, false, false, AT_BITS16> : at::at_sweep16<MODE, 64, 16, 2, false, false, false, false, AT_BITS16>;
	default: return p3<MODE, 4>(store, tb);
	}
}
/* overlap, packed: with pointers only, 4 or 16 rows per lane */
template <int K>
static at_sweep16_fn ov3(int store)
{
	if (store == 0) return at::at_sweep16<at::K_OVERLAP, 64, K, 2, true, true, true, false, AT_BITS16>;
	if (store == 1) return at::at_sweep16<at::K_OVERLAP, 64, K, 2, true, false, true, false, AT_BITS16>;
	return at::at_sweep16<at::K_OVERLAP, 64, K, 2, false, false, true, false, AT_BITS16>;
}
at_sweep16_fn AT_NAME(at_pick16_g64_ts2)(int kmode, int k, int store, bool tb)
{
	if (kmode == at::K_OVERLAP) return !tb ? nullptr : k == 16 ? ov3<16>(store) : k == 4 ? ov3<4>(store) : nullptr;
	switch (kmode) {
	case at::K_GLOBAL: return p2<at::K_GLOBAL>(k, store, tb);
	case at::K_LOCAL: return p2<at::K_LOCAL>(k, store, tb);
	case at::K_FITJ: return p2<at::K_FITJ>(k, store, tb);
	default: return p2<at::K_FIT>(k, store, tb);
	}
}
