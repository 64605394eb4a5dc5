#include "at_launch.h"
#include "at_walk16.hip.h"
/* pass 2 of the two-pass tracebacks as a kernel of its own (at_walk16.hip.h).  This unit: the walks behind the sweeps of one group of
 * 64 lanes x 16 rows (reads of up to 1 024 bases: C3), scores x4 or x16 */
template <int MODE, int TS>
static at_walk16_fn walk64(int k)
{
	if (k == 16) return at::at_walk16<MODE, 64, 16, TS, AT_BITS16, at::ck_steps(64)>;
	return nullptr;
}
at_walk16_fn AT_NAME(at_pick_walk16_g64)(int kmode, int k, int ts)
{
	if (ts == 2) {
		switch (kmode) {
		case at::K_GLOBAL: return walk64<at::K_GLOBAL, 2>(k);
		case at::K_LOCAL: return walk64<at::K_LOCAL, 2>(k);
		case at::K_FIT: return walk64<at::K_FIT, 2>(k);
		default: return nullptr;
		}
	}
	switch (kmode) {
	case at::K_GLOBAL: return walk64<at::K_GLOBAL, 4>(k);
	case at::K_LOCAL: return walk64<at::K_LOCAL, 4>(k);
	case at::K_FITJ: return walk64<at::K_FITJ, 4>(k);
	default: return walk64<at::K_FIT, 4>(k);
	}
}
