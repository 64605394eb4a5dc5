#include "at_launch.h"
/* packed kernels: group width 64 with K in 1..4 rows per lane, group width 16 with K in {7, 10, 13} (reads of 97..208 bases) */
template <int MODE, int G, int K>
static at_sweep16_fn p3(int store, bool tb)
{
	if (!tb) return store < 2 ? at::at_sweep16<MODE, G, K, true, true, false> : at::at_sweep16<MODE, G, K, false, false, false>;
	if (store == 0) return at::at_sweep16<MODE, G, K, true, true, true>;
	if (store == 1) return at::at_sweep16<MODE, G, K, true, false, true>;
	return at::at_sweep16<MODE, G, K, false, false, true>;
}
template <int MODE>
static at_sweep16_fn p2(int g, int k, int store, bool tb)
{
	if (g == 16) {
		switch (k) {
		case 7: return p3<MODE, 16, 7>(store, tb);
		case 10: return p3<MODE, 16, 10>(store, tb);
		default: return p3<MODE, 16, 13>(store, tb);
		}
	}
	switch (k) {
	case 1: return p3<MODE, 64, 1>(store, tb);
	case 2: return p3<MODE, 64, 2>(store, tb);
	case 3: return p3<MODE, 64, 3>(store, tb);
	default: return p3<MODE, 64, 4>(store, tb);
	}
}
at_sweep16_fn at_pick16(int kmode, int g, int k, int store, bool tb)
{
	switch (kmode) {
	case at::K_GLOBAL: return p2<at::K_GLOBAL>(g, k, store, tb);
	case at::K_LOCAL: return p2<at::K_LOCAL>(g, k, store, tb);
	case at::K_FITJ: return p2<at::K_FITJ>(g, k, store, tb);
	default: return p2<at::K_FIT>(g, k, store, tb);
	}
}
