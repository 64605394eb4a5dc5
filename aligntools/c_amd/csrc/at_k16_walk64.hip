#include "at_launch.h"
#include "at_walk16.hip.h"
#ifndef AT_WALK_TEAM
#define AT_WALK_TEAM 4   /* lanes per pair of alignments of the walk kernel's teams on the 64-lane groups */
#endif
/* pass 2 of the two-pass tracebacks as a kernel of its own (at_walk16.hip.h).  This unit: the walks behind the sweeps of one group of
 * 64 lanes x 16 rows (reads of up to 1 024 bases: C3), scores x4 or x16 */
template <int MODE, int TS>
static at_walk16_fn walk64(int k, bool teams)
{
	if (k == 16) return teams ? at::at_walk16<MODE, 64, 16, TS, AT_BITS16, at::ck_steps(64), 0, 0, AT_WALK_TEAM>
	                          : at::at_walk16<MODE, 64, 16, TS, AT_BITS16, at::ck_steps(64)>;
	return nullptr;
}
at_walk16_fn AT_NAME(at_pick_walk16_g64)(int kmode, int k, int ts, bool teams)   /* teams: AT_WALK_TEAM lanes per pair of alignments (walk16_team_wave) */
{
	if (ts == 2) {
		switch (kmode) {
		case at::K_GLOBAL: return walk64<at::K_GLOBAL, 2>(k, teams);
		case at::K_LOCAL: return walk64<at::K_LOCAL, 2>(k, teams);
		case at::K_FIT: return walk64<at::K_FIT, 2>(k, teams);
		default: return nullptr;
		}
	}
	switch (kmode) {
	case at::K_GLOBAL: return walk64<at::K_GLOBAL, 4>(k, teams);
	case at::K_LOCAL: return walk64<at::K_LOCAL, 4>(k, teams);
	case at::K_FITJ: return walk64<at::K_FITJ, 4>(k, teams);
	default: return walk64<at::K_FIT, 4>(k, teams);
	}
}
int AT_NAME(at_walk16_team_lanes)() { return AT_WALK_TEAM; }
