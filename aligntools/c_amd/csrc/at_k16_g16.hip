#include "at_launch.h"
at_sweep16_fn AT_NAME(at_pick16_g16a)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g16b)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g16c)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g16)(int kmode, int k, int store, bool tb)
{
	if (k >= 16) return AT_NAME(at_pick16_g16c)(kmode, k, store, tb);
	return k >= 10 ? AT_NAME(at_pick16_g16b)(kmode, k, store, tb) : AT_NAME(at_pick16_g16a)(kmode, k, store, tb);
}
