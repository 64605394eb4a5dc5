#include "at_launch.h"
#define AT_DECL(b)                                                                 \
	at_sweep16_fn at_pick16_g64_ts4_b##b(int kmode, int k, int store, bool tb);    \
	at_sweep16_fn at_pick16_g64_ts2_b##b(int kmode, int k, int store, bool tb);    \
	at_sweep16_fn at_pick16_g16_b##b(int kmode, int k, int store, bool tb);        \
	at_sweep16_fn at_pick16_g32_b##b(int kmode, int k, int store, bool tb);        \
	at_sweep16_fn at_pick16_g8_b##b(int kmode, int k, int store, bool tb);         \
	at_sweep16_fn at_pick16_g4_b##b(int kmode, int k, int store, bool tb);         \
	at_sweep16_fn at_pick16_rag_impl_b##b(int k, int store, bool tb);                \
	at_sweep16_fn at_pick16_rag8a_b##b(int kmode, int k, int store, bool tb);      \
	at_sweep16_fn at_pick16_rag8b_b##b(int kmode, int k, int store, bool tb);      \
	at_sweep16_fn at_pick16_rag8c_b##b(int kmode, int k, int store, bool tb);      \
	at_sweep16_fn at_pick16_rag8d_b##b(int kmode, int k, int store, bool tb);      \
	at_sweep16_fn at_pick16_rag16_b##b(int kmode, int k, int store, bool tb);      \
	at_sweep16_fn at_pick16_rag16b_b##b(int kmode, int k, int store, bool tb);     \
	at_sweep16_fn at_pick16_rag32_b##b(int kmode, int k, int store, bool tb);      \
	at_sweep16_fn at_pick16_ragovl_b##b(int k, int store);                           \
	at_sweep16_fn at_pick16_tp8_b##b(int kmode, int k, int split);                             \
	at_sweep16_fn at_pick16_tp64_b##b(int kmode, int k, int ts, int split);                  \
	at_walk16_fn at_pick_walk16_g8_b##b(int kmode, int g, int k, bool teams);                    \
	at_walk16_fn at_pick_walk16_g64_b##b(int kmode, int k, int ts, bool teams);
AT_DECL(2)
AT_DECL(8)
#undef AT_DECL
at_sweep16_fn at_pick16(int kmode, int g, int k, int ts, int store, bool tb, int bits)
{
	if (g == 8) return ts != 4 ? nullptr : bits == 8 ? at_pick16_g8_b8(kmode, k, store, tb) : at_pick16_g8_b2(kmode, k, store, tb);
	if (g == 4) return ts != 4 ? nullptr : bits == 8 ? at_pick16_g4_b8(kmode, k, store, tb) : at_pick16_g4_b2(kmode, k, store, tb);
	if (g == 32) return ts != 4 ? nullptr : bits == 8 ? at_pick16_g32_b8(kmode, k, store, tb) : at_pick16_g32_b2(kmode, k, store, tb);
	if (bits == 8) {
		if (g == 16) return ts == 4 ? at_pick16_g16_b8(kmode, k, store, tb) : nullptr;
		return ts == 4 ? at_pick16_g64_ts4_b8(kmode, k, store, tb) : at_pick16_g64_ts2_b8(kmode, k, store, tb);
	}
	if (g == 16) return ts == 4 ? at_pick16_g16_b2(kmode, k, store, tb) : nullptr;
	return ts == 4 ? at_pick16_g64_ts4_b2(kmode, k, store, tb) : at_pick16_g64_ts2_b2(kmode, k, store, tb);
}
at_sweep16_fn at_pick16_rag(int kmode, int g, int k, int store, bool tb, int bits)
{
	if (g == 8) {
		if (k == 6 || k == 8) return bits == 8 ? at_pick16_rag8d_b8(kmode, k, store, tb) : at_pick16_rag8d_b2(kmode, k, store, tb);
		if (bits == 8) return k >= 19 ? at_pick16_rag8c_b8(kmode, k, store, tb) : k >= 13 ? at_pick16_rag8b_b8(kmode, k, store, tb) : at_pick16_rag8a_b8(kmode, k, store, tb);
		return k >= 19 ? at_pick16_rag8c_b2(kmode, k, store, tb) : k >= 13 ? at_pick16_rag8b_b2(kmode, k, store, tb) : at_pick16_rag8a_b2(kmode, k, store, tb);
	}
	if (kmode == at::K_OVERLAP) return g != 64 || !tb ? nullptr : bits == 8 ? at_pick16_ragovl_b8(k, store) : at_pick16_ragovl_b2(k, store);
	if (g == 32) return bits == 8 ? at_pick16_rag32_b8(kmode, k, store, tb) : at_pick16_rag32_b2(kmode, k, store, tb);
	if (g != 16) return nullptr;
	if (k >= 16) return bits == 8 ? at_pick16_rag16b_b8(kmode, k, store, tb) : at_pick16_rag16b_b2(kmode, k, store, tb);   /* reads of 209..304 bases */
	if (kmode == at::K_LOCAL) return bits == 8 ? at_pick16_rag_impl_b8(k, store, tb) : at_pick16_rag_impl_b2(k, store, tb);
	return bits == 8 ? at_pick16_rag16_b8(kmode, k, store, tb) : at_pick16_rag16_b2(kmode, k, store, tb);
}
/* two-pass traceback kernels (CK): global slot for the checkpoints, LDS for the s2 windows */
at_sweep16_fn at_pick16_tp(int kmode, int g, int k, int ts, int bits, int split)
{
	if (g == 8) return ts != 4 ? nullptr : bits == 8 ? at_pick16_tp8_b8(kmode, k, split) : at_pick16_tp8_b2(kmode, k, split);
	if (g == 64) return bits == 8 ? at_pick16_tp64_b8(kmode, k, ts, split) : at_pick16_tp64_b2(kmode, k, ts, split);
	return nullptr;
}
int at_walk16_team_lanes_b2();
int at_walk16_team_lanes8_b2();
int at_walk16_team_lanes(int g) { return g == 64 ? at_walk16_team_lanes_b2() : at_walk16_team_lanes8_b2(); }
at_walk16_fn at_pick_walk16(int kmode, int g, int k, int ts, int bits, int teams)
{
	if (g == 64) return bits == 8 ? at_pick_walk16_g64_b8(kmode, k, ts, teams != 0) : at_pick_walk16_g64_b2(kmode, k, ts, teams != 0);
	if (ts != 4) return nullptr;
	return bits == 8 ? at_pick_walk16_g8_b8(kmode, g, k, teams != 0) : at_pick_walk16_g8_b2(kmode, g, k, teams != 0);
}
