/* at_launch.h -- kernel tables; each at_k*.hip translation unit instantiates one slice so the
 * slices compile in parallel (make -j). */
#pragma once
#include "at_sweep16.hip.h"

typedef void (*at_sweep_fn)(const at::SweepArgs);
typedef void (*at_sweep16_fn)(const at::Sweep16Args);

at_sweep_fn at_pick32_b2(int kmode, int k, bool small, bool tb);
at_sweep_fn at_pick32_b8(int kmode, int k, bool small, bool tb);
at_sweep16_fn at_pick16(int kmode, int k, bool small, bool tb);   /* kmode in {K_GLOBAL, K_LOCAL, K_FIT} */

template <int MODE, int BITS, int K>
static at_sweep_fn at_pick3(bool small, bool tb)
{
	if constexpr (MODE == at::K_EDIT) {
		return small ? at::at_sweep<MODE, BITS, K, true, false> : at::at_sweep<MODE, BITS, K, false, false>;
	} else {
		if (small) return tb ? at::at_sweep<MODE, BITS, K, true, true> : at::at_sweep<MODE, BITS, K, true, false>;
		return tb ? at::at_sweep<MODE, BITS, K, false, true> : at::at_sweep<MODE, BITS, K, false, false>;
	}
}
template <int MODE, int BITS>
static at_sweep_fn at_pick2(int k, bool small, bool tb)
{
	switch (k) {
	case 1: return at_pick3<MODE, BITS, 1>(small, tb);
	case 2: return at_pick3<MODE, BITS, 2>(small, tb);
	case 3: return at_pick3<MODE, BITS, 3>(small, tb);
	default: return at_pick3<MODE, BITS, 4>(small, tb);
	}
}
template <int BITS>
static at_sweep_fn at_pick1(int kmode, int k, bool small, bool tb)
{
	switch (kmode) {
	case at::K_GLOBAL: return at_pick2<at::K_GLOBAL, BITS>(k, small, tb);
	case at::K_LOCAL: return at_pick2<at::K_LOCAL, BITS>(k, small, tb);
	case at::K_FIT: return at_pick2<at::K_FIT, BITS>(k, small, tb);
	case at::K_FITJ: return at_pick2<at::K_FITJ, BITS>(k, small, tb);
	case at::K_OVERLAP: return at_pick2<at::K_OVERLAP, BITS>(k, small, tb);
	default: return at_pick2<at::K_EDIT, BITS>(k, small, tb);
	}
}
