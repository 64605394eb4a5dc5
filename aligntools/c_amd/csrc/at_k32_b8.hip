#include "at_launch.h"
at_sweep_fn at_pick32_b8(int kmode, int k, int store, bool tb) { return at_pick1<8>(kmode, k, store, tb); }
