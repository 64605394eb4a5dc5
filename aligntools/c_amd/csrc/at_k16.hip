#include "at_launch.h"
at_sweep16_fn at_pick16_g64_ts4(int kmode, int k, int store, bool tb);
at_sweep16_fn at_pick16_g64_ts2(int kmode, int k, int store, bool tb);
at_sweep16_fn at_pick16_g16(int kmode, int k, int store, bool tb);
at_sweep16_fn at_pick16(int kmode, int g, int k, int ts, int store, bool tb)
{
	if (g == 16) return ts == 4 ? at_pick16_g16(kmode, k, store, tb) : nullptr;
	return ts == 4 ? at_pick16_g64_ts4(kmode, k, store, tb) : at_pick16_g64_ts2(kmode, k, store, tb);
}
