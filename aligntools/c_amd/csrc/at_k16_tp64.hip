#include "at_launch.h"
/* two-pass traceback kernels (at_sweep16.hip.h, CK), one group of 64 lanes x 16 rows in one strip (reads of up to 1 024 bases: C3),
 * scores x4 or x16.  (fit -s with scores x4 keeps byte cells in the one-pass kernels and has no two-pass form.) */
template <int MODE, int TS>
static at_sweep16_fn tp64(int k, int split)
{
	switch (k) {
	case 16: return split ? at::at_sweep16<MODE, 64, 16, TS, true, false, false, false, AT_BITS16, at::ck_steps(64), 1>
	                      : at::at_sweep16<MODE, 64, 16, TS, true, false, false, false, AT_BITS16, at::ck_steps(64)>;
	default: return nullptr;
	}
}
at_sweep16_fn AT_NAME(at_pick16_tp64)(int kmode, int k, int ts, int split)
{
	if (ts == 2) {
		switch (kmode) {
		case at::K_GLOBAL: return tp64<at::K_GLOBAL, 2>(k, split);
		case at::K_LOCAL: return tp64<at::K_LOCAL, 2>(k, split);
		case at::K_FIT: return tp64<at::K_FIT, 2>(k, split);
		default: return nullptr;
		}
	}
	switch (kmode) {   /* scores x16: reads of 609 .. 1 024 bases whose scores stay below 2 048 */
	case at::K_GLOBAL: return tp64<at::K_GLOBAL, 4>(k, split);
	case at::K_LOCAL: return tp64<at::K_LOCAL, 4>(k, split);
	case at::K_FITJ: return tp64<at::K_FITJ, 4>(k, split);
	default: return tp64<at::K_FIT, 4>(k, split);
	}
}
