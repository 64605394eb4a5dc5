#include "at_launch.h"
/* two-pass traceback kernels (at_sweep16.hip.h, CK): the scores-only sweep with checkpoints, pointers rebuilt block by block where the
 * walks need them.  This unit: eight groups of 8 lanes x 19 rows (reads of 129 .. 152 bases: C2, C4), scores x16 */
template <int MODE>
static at_sweep16_fn tp8(int k, int split)
{
	if (k != 19) return nullptr;
	return split ? at::at_sweep16<MODE, 8, 19, 4, true, false, false, false, AT_BITS16, at::ck_steps(8), 1>
	             : at::at_sweep16<MODE, 8, 19, 4, true, false, false, false, AT_BITS16, at::ck_steps(8)>;
}
at_sweep16_fn AT_NAME(at_pick16_tp8)(int kmode, int k, int split)   /* split: pass 2 is a kernel of its own (at_walk16.hip.h) */
{
	switch (kmode) {
	case at::K_GLOBAL: return tp8<at::K_GLOBAL>(k, split);
	case at::K_LOCAL: return tp8<at::K_LOCAL>(k, split);
	case at::K_FITJ: return tp8<at::K_FITJ>(k, split);
	default: return tp8<at::K_FIT>(k, split);
	}
}
