#include "at_launch.h"
at_sweep16_fn AT_NAME(at_pick16_g8a)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g8b)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g8c)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g8d)(int kmode, int k, int store, bool tb);
at_sweep16_fn AT_NAME(at_pick16_g8)(int kmode, int k, int store, bool tb)
{
	if (k == 6 || k == 8) return AT_NAME(at_pick16_g8d)(kmode, k, store, tb);
	return k >= 19 ? AT_NAME(at_pick16_g8c)(kmode, k, store, tb) : k >= 13 ? AT_NAME(at_pick16_g8b)(kmode, k, store, tb) : AT_NAME(at_pick16_g8a)(kmode, k, store, tb);
}
