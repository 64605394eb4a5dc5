#include "at_launch.h"
template <int MODE, int K>
static at_sweep16_fn p3(bool small, bool tb)
{
	if (small) return tb ? at::at_sweep16<MODE, K, true, true> : at::at_sweep16<MODE, K, true, false>;
	return tb ? at::at_sweep16<MODE, K, false, true> : at::at_sweep16<MODE, K, false, false>;
}
template <int MODE>
static at_sweep16_fn p2(int k, bool small, bool tb)
{
	switch (k) {
	case 1: return p3<MODE, 1>(small, tb);
	case 2: return p3<MODE, 2>(small, tb);
	case 3: return p3<MODE, 3>(small, tb);
	default: return p3<MODE, 4>(small, tb);
	}
}
at_sweep16_fn at_pick16(int kmode, int k, bool small, bool tb)
{
	switch (kmode) {
	case at::K_GLOBAL: return p2<at::K_GLOBAL>(k, small, tb);
	case at::K_LOCAL: return p2<at::K_LOCAL>(k, small, tb);
	default: return p2<at::K_FIT>(k, small, tb);
	}
}
