#include "at_launch.h"
#include "at_walk16.hip.h"
/* pass 2 of the two-pass tracebacks as a kernel of its own (at_walk16.hip.h).  This unit: the walks behind the sweeps of eight groups
 * of 8 lanes x 19 rows (reads of 129 .. 152 bases: C2, C4) and behind their sliver items (two groups of 32 lanes x 5 rows), scores x16 */
#ifndef AT_WALK_TEAM8
#define AT_WALK_TEAM8 2   /* lanes per pair of alignments of the walk kernel's teams on the 8-lane groups */
#endif
template <int MODE>
static at_walk16_fn walk8(int g, int k, bool teams)
{
	if (g == 8 && k == 19) return teams ? at::at_walk16<MODE, 8, 19, 4, AT_BITS16, at::ck_steps(8), at::AT_TAIL_G, at::at_tail_k(8, 19), AT_WALK_TEAM8>
	                                    : at::at_walk16<MODE, 8, 19, 4, AT_BITS16, at::ck_steps(8), at::AT_TAIL_G, at::at_tail_k(8, 19)>;
	return nullptr;
}
at_walk16_fn AT_NAME(at_pick_walk16_g8)(int kmode, int g, int k, bool teams)
{
	switch (kmode) {
	case at::K_GLOBAL: return walk8<at::K_GLOBAL>(g, k, teams);
	case at::K_LOCAL: return walk8<at::K_LOCAL>(g, k, teams);
	case at::K_FITJ: return walk8<at::K_FITJ>(g, k, teams);
	default: return walk8<at::K_FIT>(g, k, teams);
	}
}
int AT_NAME(at_walk16_team_lanes8)() { return AT_WALK_TEAM8; }
