/* at_launch.h -- kernel tables; each at_k*.hip translation unit instantiates one slice so the
 * slices compile in parallel (make -j). */
#pragma once
#include "at_sweep16.hip.h"

typedef void (*at_sweep_fn)(const at::SweepArgs);
typedef void (*at_sweep16_fn)(const at::Sweep16Args, const at::Sweep16Args);   /* (batch, its sliver on 64-lane items) */
namespace at { struct MyersArgs; }
typedef void (*at_myers_fn)(const at::MyersArgs);
at_myers_fn at_pick_myers(int w, int g);
at_myers_fn at_pick_myers_semi(int w);   /* the overlap filter: one alignment per lane, w in {2, 3, 4, 5, 8, 16, 32} */   /* w 32-bit words per lane, g lanes per alignment: (1,2,4,8 x 32), (1 x 8), (5,8,16,32 x 1) */

/* store: 0 = everything in LDS, 1 = s2/boundary in LDS + pointers in the global slot, 2 = everything global */
at_sweep_fn at_pick32_b2(int kmode, int k, int store, bool tb);
at_sweep_fn at_pick32_b8(int kmode, int k, int store, bool tb);
at_sweep16_fn at_pick16_rag(int kmode, int g, int k, int store, bool tb, int bits);   /* ragged frames: g in {8, 16} */
at_sweep16_fn at_pick16(int kmode, int g, int k, int ts, int store, bool tb, int bits);
at_sweep16_fn at_pick16_tp(int kmode, int g, int k, int ts, int bits, int split = 0);   /* two-pass tracebacks (CK kernels; split = 1: pass 2 is a kernel of its own), or nullptr */
typedef void (*at_walk16_fn)(const at::Sweep16Args, const at::Sweep16Args);   /* (launch, its sliver) */
at_walk16_fn at_pick_walk16(int kmode, int g, int k, int ts, int bits, int teams = 0);   /* their pass 2 as a kernel of its own (at_walk16.hip.h; teams: several lanes per pair of alignments -- the 64-lane groups), or nullptr */
int at_walk16_team_lanes(int g);
/* every packed translation unit is compiled twice: -DAT_BITS16=2 (16 codes per sequence word, score LUT) and
 * -DAT_BITS16=8 (4 bytes per word, compare); its entry points carry the suffix _b2 / _b8 */
#ifndef AT_BITS16
#define AT_BITS16 2
#endif
#define AT_CAT_(a, b) a##b
#define AT_CAT(a, b) AT_CAT_(a, b)
#define AT_NAME(f) AT_CAT(f, AT_CAT(_b, AT_BITS16))   /* g: lanes per group (64 or 16); ts: 4 (x16) or 2 (x4) */   /* kmode in {K_GLOBAL, K_LOCAL, K_FIT} */

template <int MODE, int BITS, int K>
static at_sweep_fn at_pick3(int store, bool tb)
{
	if constexpr (MODE == at::K_EDIT) {
		return store < 2 ? at::at_sweep<MODE, BITS, K, true, true, false> : at::at_sweep<MODE, BITS, K, false, false, false>;
	} else {
		if (!tb) return store < 2 ? at::at_sweep<MODE, BITS, K, true, true, false> : at::at_sweep<MODE, BITS, K, false, false, false>;
		if (store == 0) return at::at_sweep<MODE, BITS, K, true, true, true>;
		if (store == 1) return at::at_sweep<MODE, BITS, K, true, false, true>;
		return at::at_sweep<MODE, BITS, K, false, false, true>;
	}
}
template <int MODE, int BITS>
static at_sweep_fn at_pick2(int k, int store, bool tb)
{
	if constexpr (MODE == at::K_OVERLAP || MODE == at::K_EDIT) {
		/* deep lanes for the kernels that keep one value per row and no pointer matrix */
		if (!tb || MODE == at::K_EDIT) {
			if (k == 8) return store < 2 ? at::at_sweep<MODE, BITS, 8, true, true, false> : at::at_sweep<MODE, BITS, 8, false, false, false>;
			if (k == 16) return store < 2 ? at::at_sweep<MODE, BITS, 16, true, true, false> : at::at_sweep<MODE, BITS, 16, false, false, false>;
		}
	}
	switch (k) {
	case 1: return at_pick3<MODE, BITS, 1>(store, tb);
	case 2: return at_pick3<MODE, BITS, 2>(store, tb);
	case 3: return at_pick3<MODE, BITS, 3>(store, tb);
	default: return at_pick3<MODE, BITS, 4>(store, tb);
	}
}
template <int BITS>
static at_sweep_fn at_pick1(int kmode, int k, int store, bool tb)
{
	switch (kmode) {
	case at::K_GLOBAL: return at_pick2<at::K_GLOBAL, BITS>(k, store, tb);
	case at::K_LOCAL: return at_pick2<at::K_LOCAL, BITS>(k, store, tb);
	case at::K_FIT: return at_pick2<at::K_FIT, BITS>(k, store, tb);
	case at::K_FITJ: return at_pick2<at::K_FITJ, BITS>(k, store, tb);
	case at::K_OVERLAP: return at_pick2<at::K_OVERLAP, BITS>(k, store, tb);
	default: return at_pick2<at::K_EDIT, BITS>(k, store, tb);
	}
}
