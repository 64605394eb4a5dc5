#include "at_launch.h"
at_sweep_fn at_pick32_b2(int kmode, int k, int store, bool tb) { return at_pick1<2>(kmode, k, store, tb); }
