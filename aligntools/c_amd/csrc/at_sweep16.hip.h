/*
 * at_sweep16.hip.h -- packed-int16 variant of the anti-diagonal sweep (gfx950).
 *
 * Same geometry as at_sweep.hip.h (K rows per lane, DPP shift for the cell
 * above, time-major pointers), but every 32-bit register carries TWO alignments
 * of the same shape (l1, l2): pair A in bits [15:0], pair B in bits [31:16].
 * With G = 64 one wavefront sweeps one such pair of alignments; with G = 16 the
 * wavefront is four independent groups of 16 lanes (DPP row_shr:1 stays inside a
 * row of 16), each sweeping its own two alignments with K = ceil(l1/16) rows per
 * lane -- 8 alignments per wavefront.  For 150 x 150 that cuts the ramp from
 * 49 + 150 steps on 50 lanes (59 % of lane-steps inside the matrix) to 14 + 150
 * steps on 15 of 16 lanes (85 %).  gfx950 issues v_pk_add_i16 / v_pk_max_i16 at
 * the same rate as the 32-bit integer max (measured: tools/valu_rate*.hip), and
 * the tag/clean/pointer logic is bitwise, so one instruction stream fills two
 * DP matrices.  Used for batches of uniform shape whose scores provably fit:
 * 16 * (|score| range) < 2^15 with the -inf sentinel at -32768 (saturating
 * adds keep it there); everything else takes the int32 kernel.
 *
 * Score lookup: s2 is staged in LDS one BYTE per base (codes 0..3, unpacked
 * from the 2-bit HBM format); x = window ^ query gives 0 for a match; one
 * v_perm_b32 gathers step k's two x bytes into a selector and a second one
 * reads the 16-bit scores of both pairs from a byte LUT {m,u,u,u}.
 *
 * Tags, pointer bits and tie-breaking are exactly those of at_sweep.hip.h,
 * per 16-bit half (reference max5 first-wins order, alignment.h:90-100), when
 * TS = 4 (scores scaled by 16, |score| < 2048).  TS = 2 trades pointer bits for
 * range: scores scaled by 4 (|score| < 8192, e.g. 1024 x 1024 global), tags
 * L 3 / M 2 / U 1 / J,HOME 0; the pointer nibble then takes bit 0 of the L and U
 * winners' tags (two packed shifts more than TS = 4).
 * Local arg-max (alignment.h:830-833, first in row-major order) is tracked
 * once per step: the K cells of the column are reduced with the row index in
 * the (otherwise constant) tag bits, then folded into a per-lane running
 * (best, step) pair with a packed compare built from a saturating subtract.
 */
#pragma once
#include "at_sweep.hip.h"
#include <utility>

#ifndef AT_WALK_AHEAD
#define AT_WALK_AHEAD 4   /* fit walks: pointer words loaded ahead along the current run (C4: 2 -> 1.96, 4 -> 2.07, 8 -> 2.01 TCUPS) */
#endif
#ifndef AT_DIAG_NO_WALK
#define AT_DIAG_NO_WALK 0   /* 1: throw-away build without the traceback walks (what do they cost?); every pair reports 0 ops */
#endif
#ifndef AT_GLOBAL_WALK_AHEAD
#define AT_GLOBAL_WALK_AHEAD 1   /* global walks cross the whole matrix like fit ones: the same look-ahead along the current run, where 4 .. 16
                                  * walks share a wavefront (150 x 150: 2 448 -> 2 778 GCUPS, 250 x 250: 2 065 -> 2 344).  Not on the 64-lane
                                  * groups, whose two walks per wavefront gain nothing (C3 with launches in flight 2 977 -> 2 997) and lose
                                  * alone on the chip (2 138 -> 1 915) */
#endif
#ifndef AT_LOCAL_WALK_AHEAD
#define AT_LOCAL_WALK_AHEAD 1    /* and local ones, although those of unrelated reads are a dozen ops long (C2 on the 8-lane groups: 2 966 ->
                                  * 3 042 GCUPS with launches in flight, 2 266 -> 2 447 alone; on 16-lane groups round 1 had measured -4 %) */
#endif
#ifndef AT_WALK_PRIO
#define AT_WALK_PRIO 2    /* s_setprio of a wave while it walks: the walk is a chain of dependent loads with a few instructions in
                          * between, which should not queue behind the other waves' sweeps (C3 +2.5 %, C4 +1 %; 0 = off) */
#endif
#ifndef AT_JPLANE
#define AT_JPLANE 1       /* fit -s, scores x16: 4-bit pointer cells + a bit plane for the jump state's pointers (5 bits per cell and
                          * alignment instead of 8; 0 = the round-2 byte cells, for A/B runs) */
#endif
#ifndef AT_OVL_BITS
#define AT_OVL_BITS 2     /* packed overlap: bits per pointer cell (the walk reads 2; 4 = the round-2 nibble cells, for A/B runs) */
#endif

namespace at {

typedef short s16x2 __attribute__((ext_vector_type(2)));

struct Sweep16Args {
	long long npairs;
	const uint32_t *seq;
	const long long *woff1;
	const long long *woff2;
	const int *len1, *len2;        /* per pair, only to verify the caller's uniform-shape promise */
	int l1, l2;                    /* uniform shape of the whole batch                    */
	int m16, u16, o16, e16, g16;   /* scores * 16 (each fits int16); g16: jump penalty    */
	const uint32_t *sitemask;      /* fit -s: bit (j + 64) set = M->J may open at column j */
	int thresh16;                  /* values <= thresh16 are "-inf" (fit end-cell scan)   */
	int *score, *end_i, *end_j, *state;
	uint8_t *ops;
	const long long *ops_off;
	int *nops;
	uint32_t *ws;
	long long ws_slot_words;
	int off_refb, off_bound, off_ptr, off_sm, nsm;   /* off_sm/nsm: site mask words staged behind the boundary row */
	int ptr_lanes;
	unsigned long long *queue;     /* work counter, zeroed before every launch */
	/* RAG kernels (ragged batch): l1 / l2 are the FRAME the launch is sized for, len1 / len2 the pairs' own
	 * extents (<= the frame); work item w takes pairs order[2*NG*w ...] (pairs of similar size, chosen by the host) */
	const int *order;
	const int *only_if;            /* optional guard, see SweepArgs */
	int only_val;
	/* two-pass tracebacks (CK kernels, see replay16_block): regions of the per-wave global slot, word offsets --
	 * border row 0, row checkpoints, column checkpoints, the replayed blocks' pointer words (+ jump plane) */
	int off_brow, off_rck, off_cck, off_rptr, off_rjpl;
	/* ... with pass 2 as a kernel of its own (at_walk16.hip.h): the checkpoints of work item w of this launch go to ck + w * ck_item_words
	 * instead of the wave's slot (off_rck / off_cck count from there), row 0 of the matrix -- the same for every alignment of the
	 * batch -- to ck_brow, and every alignment's end cell, state and verdict to tp_end; the sweep then ends where the rounds would
	 * begin.  ck = nullptr: the rounds run inside the sweep's kernel. */
	uint32_t *ck;
	long long ck_item_words;
	uint32_t *ck_brow;
	int4 *tp_end;
};

AT_DEV uint32_t pk2(int v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }
AT_DEV uint32_t pminu(uint32_t a, uint32_t b)
{
	uint32_t d;
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
AT_DEV uint32_t pmad(uint32_t a, uint32_t b, uint32_t c)   /* per half: a * b + c (low 16 bits) */
{
	uint32_t d;
	asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
AT_DEV uint32_t padd(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
AT_DEV uint32_t psub(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
AT_DEV uint32_t pmax(uint32_t a, uint32_t b)
{
	return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
/* shift each 16-bit half left by 4 (v_pk_lshlrev_b16) */
AT_DEV uint32_t pshl4(uint32_t a)
{
	return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) << (s16x2)(4));
}
template <int N>
AT_DEV uint32_t pshln(uint32_t a)
{
	return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) << (s16x2)(N));
}
AT_DEV uint32_t pshl8(uint32_t a)
{
	return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) << (s16x2)(8));
}
/* 0xffff in every half that is negative */
AT_DEV uint32_t pneg(uint32_t a)
{
	return __builtin_bit_cast(uint32_t, __builtin_bit_cast(s16x2, a) >> (s16x2)(15));
}
AT_DEV int half(uint32_t v, int h) { return (int)(short)(h ? (v >> 16) : (v & 0xffffu)); }

constexpr uint32_t kClean2 = 0xfff0fff0u, kTagL2 = 0x000f000fu, kTagM2 = 0x000a000au, kTagU2 = 0x00010001u;
constexpr int kNeg16 = -32768;

/* border cells, scaled by 16, int16 range (see border<> in at_sweep.hip.h) */
template <int MODE>
AT_DEV void border16(int i, int j, int o16, int e16, int &L, int &M, int &U)
{
	if constexpr (MODE == K_GLOBAL) {
		if (i == 0 && j == 0) { L = o16; M = 0; U = o16; }
		else if (j == 0) { L = o16 + e16 * i; M = kNeg16; U = kNeg16; }
		else { L = kNeg16; M = kNeg16; U = o16 + e16 * j; }
	} else if constexpr (MODE == K_LOCAL) {
		L = 0; M = 0; U = 0;
	} else {
		if (i == 0) { L = kNeg16; M = 0; U = 0; }
		else { L = kNeg16; M = kNeg16; U = kNeg16; }
	}
}
AT_DEV int sat16(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* second __launch_bounds__ argument = waves per SIMD the register allocator must leave room for */
#ifndef AT_WAVES16
#define AT_WAVES16(G, K) ((G) == 8 || (G) == 4 ? ((K) >= 16 ? 2 : (K) >= 10 ? 3 : 1) : ((G) == 16 && (K) >= 16) || ((G) == 32 && (K) >= 10) ? 2 : (G) == 16 && (K) >= 10 ? 3 : 1)
#endif

/* shift up by one lane inside a group of G lanes; lane 0 of each group keeps `old` */
template <int G>
AT_DEV uint32_t grp_up1(uint32_t old, uint32_t src)
{
	if constexpr (G == 64) return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
	else if constexpr (G == 32) {
		/* two DPP rows per group: shift across the whole wave, then give lane 32 (lane 0 of the second group) its own `old` */
		const uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
		return threadIdx.x == 32 ? old : v;
	}
	else if constexpr (G == 8) {
		/* two groups per DPP row of 16: lane 8 of a row (lane 0 of the second group) keeps its own `old` */
		const uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
		return (threadIdx.x & 15) == 8 ? old : v;
	}
	else if constexpr (G == 4) {
		/* four groups per DPP row of 16: lanes 4, 8 and 12 of a row (lane 0 of their groups) keep their own `old` */
		const uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
		return (threadIdx.x & 3) == 0 ? old : v;
	}
	else return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}

/* Word (wr, r, lane) of a strip's pointer matrix: wr = block of SPW steps, r = row in lane.  In LDS the lanes of one
 * (wr, r) are adjacent.  In the HBM slot four rows of a lane are adjacent, then the lanes (the last K % 4 rows form a
 * narrower group of their own): a lane stores 16 bytes at once (about K / 4 stores per block instead of K, each covering
 * 1 KiB of consecutive addresses), and a 128-byte line holds 8 lanes x 4 rows, so a pointer walk, which climbs one row
 * per op, finds up to four ops in a line instead of a new line per op.  NL is a multiple of 4 (16-byte alignment). */
/* bit offset of step `sw` of a pointer word inside its 16-bit half (see the assembly of acc[] in the step body) */
template <int PB>
AT_DEV int cell_shift(int sw)
{
	/* 8-bit cells: two steps per half.  4-bit cells: nibbles [step 3 | step 1 | step 2 | step 0].  2-bit cells (overlap): two such
	 * nibble words on top of each other, steps 0..3 in bits [1:0] of the nibbles, steps 4..7 in bits [3:2] */
	return PB == 8 ? 8 * sw : PB == 4 ? 8 * (sw & 1) + 4 * (sw >> 1) : 8 * (sw & 1) + 4 * ((sw >> 1) & 1) + 2 * (sw >> 2);
}

template <bool LDS, int K>
AT_DEV int pidx(int wr, int r, int lane, int NL)
{
	constexpr int KF = K / 4 * 4, KR = K - KF;
	/* 24-bit multiplies (v_mul_u32_u24, one issue slot; the 32-bit v_mul_lo_u32 takes four): the host keeps a slot below
	 * 2^24 words for the packed kernels, so every index here is */
	if constexpr (LDS) return __mul24(wr * K + r, NL) + lane;
	else return r < KF ? __mul24(wr, NL * K) + __mul24(r & ~3, NL) + lane * 4 + (r & 3) : __mul24(wr, NL * K) + KF * NL + lane * KR + (r - KF);
}

/* ======================================================================================================================
 * Two-pass tracebacks (round 4; CK kernels).
 *
 * The forward sweep of a CK kernel is the scores-only sweep (no priority tags, no pointer words: a third of the
 * instructions of a row-step less) plus CHECKPOINTS -- in the wave's global slot, or (SPLIT: pass 2 is the walk kernel of
 * at_walk16.hip.h, the default wherever two-pass tracebacks are the default) in the work item's region of a per-launch buffer:
 *   row checkpoints     every lane, every step: its LAST row as the lane below sees it (ck_entry: X' = max(L, M, U[, J]) with
 *                       the winner's tag, L of the row below), both alignments packed, 8 bytes.  Entry e = t + 1 is the state
 *                       after step t, entry 0 the border.
 *   column checkpoints  every lane, every CB steps: (M + o, U[, J]) of its K rows -- what a lane needs of its own past.
 * (Layout: ck_rck_word / ck_cck_word.)  Without the walk kernel the traceback then runs in ROUNDS inside this kernel.  A block is (lane band, CB steps of that lane): K rows x CB columns.  In a round
 * every lane replays two blocks, one per 16-bit half, each for the alignment of that half of its group, on its own:
 * the cell above comes from the row checkpoint instead of a DPP move, so nothing ties the lanes (or the halves)
 * together and G lanes give every alignment G blocks per round -- laid out along the direction its walk is taking
 * (CkPattern).  The replay is the tagged arithmetic of the one-pass kernels (same first-wins order, same pointer
 * cells), its pointer words go to a small region of the slot, and the alignment's walker consumes them until it leaves
 * the blocks of the round.  Pointers depend on nothing but the exact boundary values, so the ops are those of the
 * one-pass kernels (trace_back_gla / _local_affine / _fit_affine_jump, alignment.h:372-412, 558-592, 766-800).
 *
 * A round, in order: (1) every walker's anchor goes to the lanes of its group (shuffles), each lane works out its two
 * blocks; (2) replay16_block: the CB + 1 row-checkpoint entries above the block are loaded together and staged in LDS
 * (the halves of the lane's two blocks joined), the state at the block's first step is rebuilt from the column
 * checkpoint (or the border), CB steps are swept, the pointer words stored; (3) the walks, in PHASES: every walker names
 * the block it stands in, the wave copies those blocks from the slot into LDS together (one round trip), every walker
 * walks its block -- four cells ahead along its direction per look, the state machine in bit arithmetic -- until it
 * leaves what it holds; a walker that stands in a block the round has not replayed waits for the next round.
 *
 * What it buys (one MI355X, profiles/r04/two_pass_ab.jsonl): the 64-lane group with 16 rows per lane -- reads of 609 .. 1 024
 * bases, where the one-pass kernels hold 4 rows per lane in four strips -- C3 2 940 -> 3 170 GCUPS with launches in flight,
 * 2 110 -> 2 860 one launch at a time.  (Since pass 2 exists as a kernel of its own -- at_walk16.hip.h, teams of walker lanes: 4 360 -
 * 4 530 / 3 290 -- that form is the default there and the rounds below are what AT_TP_SPLIT=0 and a failed allocation fall back
 * to.)  The 8-lane groups x 19 rows lose (C2 3 050 -> 2 850, C4 2 200 ->
 * 1 580: a round replays 19 x 16 cells in every lane for the dozen cells a short walk needs, and long walks need ten rounds)
 * and stay on the one-pass kernels unless AT_TWO_PASS=2 asks.
 * ====================================================================================================================== */
#ifndef AT_TP_STATS
#define AT_TP_STATS 0     /* 1: throw-away build that counts rounds / items / walk ops / cycles in the words behind the work counter */
#endif
#ifndef AT_TP_ONEBODY
#define AT_TP_ONEBODY 1   /* the forward sweep of a CK kernel keeps one (masked) step body: 45 KB of code instead of 58 on the 64-lane group */
#endif
#ifndef AT_CK_STEPS
#define AT_CK_STEPS 16    /* CB: steps between two column checkpoints = columns of a replayed block (groups of up to 32 lanes) */
#endif
#ifndef AT_CK_STEPS64
#define AT_CK_STEPS64 32  /* ... on the 64-lane group (its LDS holds a staging area of 33 steps): half the column checkpoints of the forward
                           * sweep, half the block visits of a walk, 4.3 rounds per C3 alignment instead of 7.1 -- C3 2 987 -> 3 159 GCUPS */
#endif
#ifndef AT_TP_WAVES64
#define AT_TP_WAVES64 2   /* wavefronts per SIMD the two-pass kernels of the 64-lane group are compiled for */
#endif
#ifndef AT_CK_W64
#define AT_CK_W64 3       /* t-blocks per band of a round on the 64-lane group */
#endif
constexpr int ck_steps(int g) { return g == 64 ? AT_CK_STEPS64 : AT_CK_STEPS; }
template <int MODE> constexpr int ck_es() { return 2; }                                          /* words of a row checkpoint entry (ck_entry) */
template <int MODE, int K> constexpr int ck_nq() { return ((MODE == K_FITJ ? 3 : 2) * K + 3) / 4; }   /* 16-byte chunks of a column checkpoint */
constexpr int ck_log2(int v) { return v <= 1 ? 0 : 1 + ck_log2(v / 2); }
/* Where the checkpoints lie (words from off_rck / off_cck).  The sweep produces them step by step, 64 lanes at a time; a replay (the walk
 * kernel's above all: one walker per half-lane, every lane somewhere else) reads CB + 1 consecutive steps of ONE lane and one lane's column
 * checkpoint.  [step][lane], the sweep's own order, makes every entry a replay reads a cache line of its own (17 + 10 lines for 360 bytes
 * on the 8-lane groups: the walk kernels of C3 fetched as many bytes as the sweep wrote).  So:
 *   row checkpoints     entry 0 of every lane first (the border), then TILES of AT_CK_ROW_TILE (4) steps: [tile][lane][step][ES words] --
 *                       a replay reads 5 (9) lines where it read 17 (33); the sweep's store of one step is 64 x 12 bytes, 48 bytes apart
 *                       (whole tiles of CB steps cost the sweep 17 .. 29 %: every store 64 lines)
 *   column checkpoints  [c][chunk / Q][lane][chunk % Q] with Q = AT_CK_COL_QUAD (4) chunks of 16 bytes adjacent */
#ifndef AT_CK_ROW_TILE
#define AT_CK_ROW_TILE 4
#endif
#ifndef AT_CK_COL_QUAD
#define AT_CK_COL_QUAD 4
#endif
template <int ES>
AT_DEV int ck_rck_word(int e, int slot_lane)   /* entry e = the state after step e - 1 */
{
	constexpr int RT = AT_CK_ROW_TILE;
	const int t = e - 1;
	return e <= 0 ? slot_lane * ES : 64 * ES + ((t / RT) * 64 + slot_lane) * (RT * ES) + (t % RT) * ES;
}
template <int ES> constexpr int ck_rck_step() { return AT_CK_ROW_TILE > 1 ? ES : 64 * ES; }   /* words between the entries of two steps of one tile */
template <int NQ>
AT_DEV int ck_cck_word(int c, int slot_lane, int q)
{
	constexpr int Q = AT_CK_COL_QUAD, NQQ = (NQ + Q - 1) / Q;
	return (((c * NQQ + q / Q) * 64 + slot_lane) * Q + q % Q) * 4;
}

/* Which blocks a round replays for one alignment, seen from its walker's cell (ci, cj) -- the anchor.  Band b = the lane
 * that owns row ci, t = (cj - 1) + band = the step at which that lane swept column cj, t-block c = t / CB.
 *   diagonal: the walk is in M or L -- it climbs.  Along the diagonal ray from the anchor band b - dr is crossed at steps
 *             t_hi(dr) = T0 - rho0 - 2 - (dr - 1)(K + 1) down to t_hi - (K - 1); W t-blocks per band around that, G / W bands.
 *   horizontal: the walk is in U or J -- it runs left along its row: the G t-blocks of band b from the anchor's downwards. */
template <int G, int K, int CB>
struct CkPattern {
	static constexpr int LCB = ck_log2(CB);
	/* one group of 64 lanes: W = 3 t-blocks around the ray in each of 21 bands (closed forms: a walk of a thousand ops looks its
	 * blocks up a hundred times).  Narrower groups have 4 .. 32 slots and spend them exactly: band by band the t-blocks the ray
	 * crosses, MU / ML steps of margin above / below it (an L op moves a walk above the ray, a U op below). */
	static constexpr bool FIXED = G == 64;
	static constexpr int W = AT_CK_W64, NB = G / W;
	static constexpr int MHr = (W * CB - K) / 2, MH = MHr < 0 ? 0 : MHr > CB ? CB : MHr;
	static constexpr int MU = 3, ML = 3;
	int b, c0, thi1, horiz;
	AT_DEV void set(int ci, int cj, bool hz)
	{
		b = (ci - 1) / K;
		const int rho = (ci - 1) - b * K, T0 = cj - 1 + b;
		c0 = T0 >> LCB; thi1 = T0 - rho - 2; horiz = hz ? 1 : 0;
	}
	AT_DEV int ctop(int dr) const { return dr == 0 ? c0 : (thi1 - (dr - 1) * (K + 1) + MH) >> LCB; }   /* (arithmetic shift: floor) */
	/* band b - dr holds the t-blocks chi, chi - 1, ... chi - n + 1 of this round */
	AT_DEV void band(int dr, int &chi, int &n) const
	{
		if constexpr (FIXED) { chi = ctop(dr); n = W; }
		else {
			const int thi = thi1 - (dr - 1) * (K + 1);                     /* (dr = 0: the ray starts at the anchor's own step, c0) */
			const int tlo = dr == 0 ? thi1 + 2 : thi - (K - 1);
			chi = dr == 0 ? c0 : (thi + MU) >> LCB;
			const int clo = imax((tlo - ML) >> LCB, 0);
			n = imax(chi - clo + 1, 0);
		}
	}
	AT_DEV void block(int q, int T, int &bl, int &c, bool &valid) const   /* T: steps of the sweep */
	{
		bl = -1; c = 0;
		if (horiz) { bl = b; c = c0 - q; }
		else if constexpr (FIXED) { const int dr = q / W; if (dr < NB) { bl = b - dr; c = ctop(dr) - q % W; } }
		else {
			int ql = q;
			for (int dr = 0; dr <= b; ++dr) {
				int chi, n;
				band(dr, chi, n);
				if (ql < n) { bl = b - dr; c = chi - ql; break; }
				ql -= n;
			}
		}
		valid = bl >= 0 && c >= 0 && c * CB < T && (c + 1) * CB > bl;   /* (the block holds a step at which its lane is inside the matrix) */
	}
	/* the slot that holds block (bl, c) this round, or -1 */
	AT_DEV int slot(int bl, int c) const
	{
		const int dr = b - bl;
		if (horiz) { const int w = c0 - c; return dr == 0 && w >= 0 && w < G ? w : -1; }
		if constexpr (FIXED) {
			if (dr < 0 || dr >= NB) return -1;
			const int w = ctop(dr) - c;
			return w >= 0 && w < W ? dr * W + w : -1;
		} else {
			if (dr < 0) return -1;
			int chi, n, s0 = 0;
			for (int d = 0; d < dr; ++d) { band(d, chi, n); s0 += n; if (s0 >= G) return -1; }
			band(dr, chi, n);
			const int w = chi - c;
			return w >= 0 && w < n && s0 + w < G ? s0 + w : -1;
		}
	}
};

template <typename F, int... Q>
AT_DEV void static_for_impl(F &f, std::integer_sequence<int, Q...>) { (f(std::integral_constant<int, Q>{}), ...); }
template <int N, typename F>
AT_DEV void static_for(F &f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

/* Word (s4, r, lane) of a round's replayed pointer words: s4 = group of 4 steps of the block, r = row in lane; four rows of a lane
 * adjacent (the rows padded to a multiple of 4), so that a lane stores 16 bytes at once and a walk fetches a tile of 4 rows x 4
 * steps with one load.  KQ = 16-byte groups per s4. */
template <int KQ>
AT_DEV int ridx(int s4, int r, int lane) { return (s4 * KQ + (r >> 2)) * 256 + lane * 4 + (r & 3); }
constexpr int ck_stage_words(int cb) { return 64 * (cb + 1) * 2; }   /* LDS: the row above the block, step by step, as the lane below sees it */
/* LDS words of one walker: the pointer words of the block it walks (K rows padded to 4 x CB steps / 4) and the jump plane's */
template <int G> constexpr int ck_walk_blocks() { return G == 64 ? 4 : 1; }   /* blocks a walker holds in LDS at a time */
template <int MODE, int K, int CB>
constexpr int ck_walk_words() { return (K + 3) / 4 * CB + (MODE == K_FITJ ? ((K + 3) / 4 + 3) / 4 * CB : 0); }

AT_DEV uint32_t pk2h(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
AT_DEV uint32_t lohi(uint32_t lo, uint32_t hi)   /* low half of `lo`, high half of `hi` */
{
	return __builtin_amdgcn_perm(hi, lo, 0x07060100u);
}
struct ck_u3 { uint32_t x, y, z; };
/* A row checkpoint entry: (L, M, U[, J]) of a lane's last row AS THE LANE BELOW SEES THEM -- X' = max(L, M, U[, J]) with the winner's
 * priority tag (the diagonal input of its first row) and L of the row below = max(L + e, M + o) (tagged: extended or opened) -- 8 bytes
 * for the two alignments of a lane where (L, M, U[, J]) were 12 / 16.  The sweep is the scores-only sweep (no tags): it ORs them in
 * here, nine instructions per step.  (C3's sweep writes 5.6 GB of checkpoints per launch at 2.9 TB/s -- that, not its instructions, is
 * what bounds it once pass 2 is out of the way: DESIGN.md 3.6.1.) */
template <bool HASJ>
AT_DEV uint2 ck_entry(uint32_t eL, uint32_t eM, uint32_t eU, uint32_t eJ, uint32_t e2, uint32_t o2, uint32_t tagL, uint32_t tagM, uint32_t tagU)
{
	const uint32_t tl = eL | tagL, tm = eM | tagM;
	uint32_t xx = pmax(pmax(tl, tm), eU | tagU);
	if constexpr (HASJ) xx = pmax(xx, eJ);
	return make_uint2(xx, pmax(padd(tl, e2), padd(tm, o2)));
}

/* One round's replay: this lane's two blocks -- (blA, cA) of the alignment in the low halves, (blB, cB) of the one in the
 * high halves; band = lane-in-group of the forward sweep, c = t-block -- swept with tags from their checkpoints, pointer
 * words (4-bit cells [+ the jump plane]) to the slot's replay region at [step / 4][row][this lane].  A block starts at
 * step org = max(c * CB, band) -- a lane's first step inside the matrix is t = band, where its state is the border's --
 * and runs CB steps; the cell of step t sits at s = t - org.  Steps behind column l2 and rows behind l1 compute cells
 * nobody reads, as in the forward sweep. */
#ifndef AT_REPLAY_INLINE
#define AT_REPLAY_INLINE 1    /* 0: the replay as a function call (A/B on the first version: C2 1 943 against 2 120 GCUPS inlined) */
#endif
#if AT_REPLAY_INLINE
#define AT_REPLAY_FN AT_DEV
#else
#define AT_REPLAY_FN __device__ __noinline__
#endif
template <int MODE, int G, int K, int TS, bool SMALL, int BITS, int CB>
AT_REPLAY_FN void replay16_block(const Sweep16Args &a, const Slot<SMALL> &mem, uint32_t *gs, const int refoff,
                           const uint32_t *qA, const uint32_t *qB, const int l1A, const int l1B,
                           const int blA, const int cA, const int blB, const int cB)
{
	constexpr bool HASJ = MODE == K_FITJ;
	static_assert(!HASJ || (TS == 4 && AT_JPLANE), "two-pass jump state: scores x16, 4-bit cells + bit plane");
	static_assert(MODE == K_GLOBAL || MODE == K_LOCAL || MODE == K_FIT || MODE == K_FITJ, "two-pass tracebacks: the affine modes");
	static_assert(CB % 4 == 0, "pointer words hold 4 steps");
	constexpr int TMASK = (1 << TS) - 1;
	constexpr int TGL = TS == 4 ? 15 : 3, TGM = TS == 4 ? 10 : 2, TGU = 1;
	constexpr int KG = (K + 3) / 4, ES = ck_es<MODE>(), NQ = ck_nq<MODE, K>();
	constexpr int PADW = kPad / 4;
	(void)PADW;
	const int lane = threadIdx.x, grp = lane / G;
	const int o16 = a.o16, e16 = a.e16;
	uint32_t o2 = pk2(o16), e2 = pk2(e16);
	uint32_t lut_lo = ((uint32_t)a.m16 & 0xffu) | (((uint32_t)a.u16 & 0xffu) * 0x01010100u);
	uint32_t lut_hi = (((uint32_t)a.m16 >> 8) & 0xffu) | ((((uint32_t)a.u16 >> 8) & 0xffu) * 0x01010100u);
	uint32_t cClean = (uint32_t)(0xffff & ~TMASK) * 0x00010001u, cTagM = (uint32_t)TGM * 0x00010001u;
	uint32_t cTagL = (uint32_t)TGL * 0x00010001u, cTagU = (uint32_t)TGU * 0x00010001u;
	uint32_t cM3 = 0x00030003u, cM7 = 0x00070007u, cF0 = 0x00f000f0u, cF000 = 0xf000f000u, c8888 = 0x88888888u;
	uint32_t gmo2 = pk2(a.g16 - a.o16), neg2 = 0x80008000u;
	uint32_t c1 = 0x00010001u, umm2 = pk2(a.u16 - a.m16), m2 = pk2(a.m16);
	asm volatile("" : "+v"(c8888), "+v"(gmo2), "+v"(neg2), "+v"(c1), "+v"(umm2), "+v"(m2));
	asm volatile("" : "+v"(o2), "+v"(e2), "+v"(lut_lo), "+v"(lut_hi));
	asm volatile("" : "+v"(cClean), "+v"(cTagM), "+v"(cTagL), "+v"(cTagU), "+v"(cM3), "+v"(cM7), "+v"(cF0), "+v"(cF000));

	const int orgA = imax(cA * CB, blA), orgB = imax(cB * CB, blB);
	/* halves that start from a column checkpoint (the others start at their lane's first step, from the border) */
	const uint32_t mck = (cA * CB > blA ? 0x0000ffffu : 0u) | (cB * CB > blB ? 0xffff0000u : 0u);
	const int i0A = blA * K, i0B = blB * K;
	/* row checkpoint entries of the row above my band: band 0 reads the border row (entry = column), the others the lane above
	 * (entry e = state after step e - 1).  p = the entry of step org - 1 -- the cell diagonally above my first one */
	auto entA = [&](int x) { return blA == 0 ? a.off_brow + (orgA + x) * ES : a.off_rck + ck_rck_word<ES>(orgA - 1 + x, grp * G + blA - 1); };
	auto entB = [&](int x) { return blB == 0 ? a.off_brow + (orgB + x) * ES : a.off_rck + ck_rck_word<ES>(orgB - 1 + x, grp * G + blB - 1); };
	/* ---- the row above my two blocks, CB + 1 entries from step org - 1 on, staged in LDS as the lane below sees it: X' = max(L, M,
	 *      U[, J]) with the winner's tag, and L of the row below = max(L + e, M + o).  All loads are issued before the state arrays
	 *      exist (registers are free now) and cost one round trip; the step loop then reads two words per step from LDS. ---- */
	const int stg = a.off_bound + lane * 2;
	{
		static_assert(ES == 2, "row checkpoint entries: (X', L of the row below), ck_entry");
		uint2 ra[CB + 1], rb[CB + 1];
#pragma unroll
		for (int x = 0; x <= CB; ++x) { ra[x] = *(const uint2 *)(gs + entA(x)); rb[x] = *(const uint2 *)(gs + entB(x)); }
#pragma unroll
		for (int x = 0; x <= CB; ++x) mem.st2(stg + x * 128, lohi(ra[x].x, rb[x].x), lohi(ra[x].y, rb[x].y));
	}

	uint32_t Mo_l[K], U_l[K], Xl[2][K], J_l[HASJ ? K : 1], qsel[K], acc[K];
	uint32_t jA[HASJ ? KG : 1], jB[HASJ ? KG : 1];
	(void)jA; (void)jB; (void)J_l;
	/* ---- the state at step org - 1 ---- */
	uint32_t Ad, lraw0;
	{
		const uint2 e0 = mem.ld2(stg);
		Ad = e0.x; lraw0 = e0.y;
	}
	{
		/* the column checkpoint's values (M + o of my K rows, then U, then J: untagged) straight into the state arrays */
		const int ckA = a.off_cck, ckB = a.off_cck;
		auto put = [&](auto XC, uint32_t val) {
			constexpr int x = decltype(XC)::value;
			if constexpr (x < K) Mo_l[x] = val;
			else if constexpr (x < 2 * K) U_l[x - K] = val;
			else if constexpr (HASJ && x < 3 * K) J_l[x - 2 * K] = val;
		};
		uint4 cva[NQ], cvb[NQ];
#pragma unroll
		for (int q = 0; q < NQ; ++q) { cva[q] = *(const uint4 *)(gs + ckA + ck_cck_word<NQ>(cA, grp * G + blA, q)); cvb[q] = *(const uint4 *)(gs + ckB + ck_cck_word<NQ>(cB, grp * G + blB, q)); }
		auto chunk = [&](auto QC) {
			constexpr int q = decltype(QC)::value;
			const uint4 va = cva[q], vb = cvb[q];
			put(std::integral_constant<int, 4 * q>{}, lohi(va.x, vb.x)); put(std::integral_constant<int, 4 * q + 1>{}, lohi(va.y, vb.y));
			put(std::integral_constant<int, 4 * q + 2>{}, lohi(va.z, vb.z)); put(std::integral_constant<int, 4 * q + 3>{}, lohi(va.w, vb.w));
		};
		static_for<NQ>(chunk);
		uint32_t lraw = lraw0;                 /* L of my first row in the checkpoint's column */
		/* the query words my bands' rows lie in (a band of K rows spans NWQ sequence words at most), fetched together */
		constexpr int BPW = 32 / BITS, LBP = BITS == 2 ? 4 : 2, NWQ = (K + BPW - 2) / BPW + 1;
		const int wqa = imin(i0A, l1A - 1) >> LBP, wqb = imin(i0B, l1B - 1) >> LBP;
		const int lwa = (l1A - 1) >> LBP, lwb = (l1B - 1) >> LBP;
		uint32_t qwa[NWQ], qwb[NWQ];
#pragma unroll
		for (int x = 0; x < NWQ; ++x) { qwa[x] = qA[imin(wqa + x, lwa)]; qwb[x] = qB[imin(wqb + x, lwb)]; }
#pragma unroll
		for (int r = 0; r < K; ++r) {
			const uint32_t vMo = Mo_l[r], vU = U_l[r];
			uint32_t vJ = neg2;
			if constexpr (HASJ) vJ = J_l[r];
			int La, Ma, Ua, Lb, Mb, Ub;
			border16<MODE>(i0A + r + 1, 0, o16, e16, La, Ma, Ua);
			border16<MODE>(i0B + r + 1, 0, o16, e16, Lb, Mb, Ub);
			La = sat16(La); Lb = sat16(Lb);
			const uint32_t bMo = pk2h(sat16((Ma | TGM) + o16), sat16((Mb | TGM) + o16));
			const uint32_t bU = pk2h(Ua | TGU, Ub | TGU);
			const uint32_t bX = pk2h(imax3(La | TGL, Ma | TGM, Ua | TGU), imax3(Lb | TGL, Mb | TGM, Ub | TGU));
			/* from the checkpoint: M + o and U as stored (untagged), L by the chain down the column */
			const uint32_t kMo = vMo | cTagM, kU = vU | cTagU;
			const uint32_t Lc = lraw | cTagL, Mc = psub(vMo, o2) | cTagM;
			uint32_t kX = pmax(pmax(Lc, Mc), kU);
			if constexpr (HASJ) kX = pmax(kX, vJ);
			lraw = pmax(padd(Lc, e2), kMo);
			Mo_l[r] = vbfi(mck, kMo, bMo);
			U_l[r] = vbfi(mck, kU, bU);
			Xl[0][r] = vbfi(mck, kX, bX);
			Xl[1][r] = Xl[0][r];
			if constexpr (HASJ) J_l[r] = vbfi(mck, vJ, neg2);
			/* my query bases, per half its own band's rows */
			const int qi = imin(i0A + r, l1A - 1), qj = imin(i0B + r, l1B - 1);
			const uint32_t wa = pick<NWQ>(qwa, (qi >> LBP) - wqa), wb = pick<NWQ>(qwb, (qj >> LBP) - wqb);
			uint32_t ca, cb;
			if constexpr (BITS == 2) { ca = (wa >> ((qi & 15) * 2)) & 3u; cb = (wb >> ((qj & 15) * 2)) & 3u; }
			else { ca = (wa >> ((qi & 3) * 8)) & 0xffu; cb = (wb >> ((qj & 3) * 8)) & 0xffu; }
			qsel[r] = (ca * 0x00000101u + cb * 0x01010000u) | (BITS == 2 ? 0x04000400u : 0u);
			acc[r] = 0;
		}
	}
	if constexpr (HASJ) {
#pragma unroll
		for (int g = 0; g < KG; ++g) { jA[g] = 0; jB[g] = 0; }
	}
	const int jA0 = orgA - blA, jB0 = orgB - blB;      /* column - 1 of step org */

	for (int s4 = 0; s4 < CB / 4; ++s4) {
		/* ---- s2 windows of the block's next 4 steps, per half from its own alignment's bytes ---- */
		uint32_t wA, wB, smA = 0, smB = 0;
		{
			const int eA = jA0 + 4 * s4 + kPad, eB = jB0 + 4 * s4 + kPad;
			const int xa = refoff + (eA >> 2), xb = refoff + a.off_refb + (eB >> 2);
			const uint32_t a0 = mem.ld(xa), a1 = mem.ld(xa + 1), b0 = mem.ld(xb), b1 = mem.ld(xb + 1);
			wA = __builtin_amdgcn_alignbit(a1, a0, (eA & 3) * 8);
			wB = __builtin_amdgcn_alignbit(b1, b0, (eB & 3) * 8);
		}
		if constexpr (HASJ) {
			const int eA = jA0 + 4 * s4 + 1 + 64, eB = jB0 + 4 * s4 + 1 + 64;
			smA = __builtin_amdgcn_alignbit(mem.ld(a.off_sm + (eA >> 5) + 1), mem.ld(a.off_sm + (eA >> 5)), eA & 31);
			smB = __builtin_amdgcn_alignbit(mem.ld(a.off_sm + (eB >> 5) + 1), mem.ld(a.off_sm + (eB >> 5)), eB & 31);
		}
		auto step = [&](auto KC) {
			constexpr int k = decltype(KC)::value;
			constexpr uint32_t SELK = (uint32_t)k * 0x00000101u + (uint32_t)(4 + k) * 0x01010000u;
			const uint2 eup = mem.ld2(stg + (1 + 4 * s4 + k) * 128);   /* the row above in this step's column */
			const uint32_t Aup = eup.x, Bup = eup.y;
			uint32_t gopen = neg2;
			if constexpr (HASJ) gopen = lohi(((smA >> k) & 1u) ? gmo2 : neg2, ((smB >> k) & 1u) ? gmo2 : neg2);
			const uint32_t selw = __builtin_amdgcn_perm(wB, wA, SELK);
			uint32_t diag = Ad, lraw = Bup, jn = 0;
			(void)jn;
#pragma unroll
			for (int r = 0; r < K; ++r) {
				/* (the row-step of the one-pass kernels with pointers, sweep16_items: same candidates, same tags, same cells) */
				uint32_t S;
				if constexpr (BITS == 2) S = __builtin_amdgcn_perm(lut_hi, lut_lo, selw ^ qsel[r]);
				else S = pmad(pminu(selw ^ qsel[r], c1), umm2, m2);
				uint32_t Mraw = padd(diag, S);
				if constexpr (MODE == K_LOCAL) Mraw = pmax(Mraw, 0u);
				const uint32_t Mc = vandor(Mraw, cClean, cTagM);
				const uint32_t Lc = lraw | cTagL;
				const uint32_t Uraw = pmax(Mo_l[r], padd(U_l[r], e2));
				const uint32_t Uc = vandor(Uraw, cClean, cTagU);
				const uint32_t Mo = padd(Mc, o2);
				uint32_t Xo = pmax(pmax(Lc, Mc), Uc);
				uint32_t Jraw = 0;
				if constexpr (HASJ) {
					Jraw = pmax(padd(Mo_l[r], gopen), J_l[r]);
					const uint32_t Jc = Jraw & cClean;
					Xo = pmax(Xo, Jc);
					J_l[r] = Jc;
				}
				const uint32_t Ld = pmax(padd(Lc, e2), Mo);
				uint32_t c;
				if constexpr (TS == 4) c = vbfi(cM7, vbfi(cM3, Mraw, lraw), Uraw);
				else c = vbfi(cM7, vbfi(cM3, Mraw, pshln<2>(lraw)), pshln<3>(Uraw));
				if constexpr (HASJ) {
					const int q = r & 3, gq = r >> 2;
					if (q == 0) jn = Jraw;
					else if (q == 1) jn = __builtin_amdgcn_perm(Jraw, jn, 0x06020400u);
					else if (q == 2) jn = vbfi(cF0, pshln<4>(Jraw), jn);
					else jn = vbfi(cF000, pshln<12>(Jraw), jn);
					if (q == 3 || r == K - 1) {
						if (q == 0) jn &= 0x000f000fu; else if (q == 1) jn &= 0x0f0f0f0fu; else if (q == 2) jn &= 0x0fff0fffu;
						if constexpr (k == 0) jA[gq] = jn;
						else if constexpr (k == 1) jA[gq] = vbfi(c8888, jn, jA[gq]);
						else if constexpr (k == 2) jB[gq] = jn;
						else jB[gq] = vbfi(c8888, jn, jB[gq]);
					}
				}
				if constexpr (k == 0) acc[r] = c;
				else if constexpr (k == 1) acc[r] = __builtin_amdgcn_perm(c, acc[r], 0x06020400u);
				else if constexpr (k == 2) acc[r] = vbfi(cF0, pshln<4>(c), acc[r]);
				else acc[r] = vbfi(cF000, pshln<12>(c), acc[r]);
				diag = Xl[k & 1][r];
				Xl[(k & 1) ^ 1][r] = Xo;
				lraw = Ld;
				Mo_l[r] = Mo; U_l[r] = Uc;
			}
			Ad = Aup;
		};
		step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
		step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
		/* ---- the pointer words of these 4 steps: [s4][row][lane], four rows of a lane adjacent (pidx) ---- */
		{
			constexpr int KQ = (K + 3) / 4;
#pragma unroll
			for (int r = 0; r < K; r += 4)
				*(uint4 *)(gs + a.off_rptr + ridx<KQ>(s4, r, lane)) = make_uint4(acc[r], r + 1 < K ? acc[r + 1 < K ? r + 1 : 0] : 0u, r + 2 < K ? acc[r + 2 < K ? r + 2 : 0] : 0u, r + 3 < K ? acc[r + 3 < K ? r + 3 : 0] : 0u);
		}
		if constexpr (HASJ) {
			constexpr int GQ = (KG + 3) / 4;
			uint32_t jO[KG];
#pragma unroll
			for (int g = 0; g < KG; ++g) jO[g] = jA[g] | (jB[g] >> 1);
#pragma unroll
			for (int g = 0; g < KG; g += 4)
				*(uint4 *)(gs + a.off_rjpl + ridx<GQ>(s4, g, lane)) = make_uint4(jO[g], g + 1 < KG ? jO[g + 1 < KG ? g + 1 : 0] : 0u, g + 2 < KG ? jO[g + 2 < KG ? g + 2 : 0] : 0u, g + 3 < KG ? jO[g + 3 < KG ? g + 3 : 0] : 0u);
		}
	}
}

/* The work items [wbase, wbase + items of `a`) of one launch, pulled from the launch's work counter: `wnext` is the item this wave
 * holds when it gets here (its block index, or what an earlier call left over); returns the first item beyond the range. */
template <int MODE, int G, int K, int TS, bool SMALL, bool PTRLDS, bool TB, bool RAG = false, int BITS = 2, int CK = 0, int SPLIT = 0>
AT_DEV long long sweep16_items(const Sweep16Args &a, long long wnext, const long long wbase)
{
	/* SPLIT = 1: pass 2 of the two-pass tracebacks is a kernel of its own (at_walk16.hip.h): this one ends where the rounds would begin */
	static_assert(!SPLIT || CK > 0, "a walk kernel follows a sweep that leaves checkpoints");
	/* CK > 0: two-pass tracebacks -- the scores-only sweep leaves checkpoints every CK steps, the pointers are rebuilt block by block
	 * where the walks need them (replay16_block above) */
	constexpr bool TP = CK > 0;
	static_assert(!TP || (!TB && !RAG && MODE != K_OVERLAP && !PTRLDS), "two-pass tracebacks: uniform batches of the affine modes, checkpoints in the global slot");
	static_assert(BITS == 2 || BITS == 8, "sequence words: 16 two-bit codes or 4 bytes");
	static_assert(!RAG || G <= 32 || MODE == K_OVERLAP, "ragged frames: one strip (the host sizes the frame for it)");
	static_assert(MODE == K_GLOBAL || MODE == K_LOCAL || MODE == K_FIT || MODE == K_FITJ || MODE == K_OVERLAP, "packed path: the affine modes, overlap");
	/* overlap (align_overlap, alignment.h:926-964): ONE state with a linear gap, M = max5(M(i,j-1) + o, M(i-1,j-1) + s,
	 * M(i-1,j) + o), first wins in that order -> tags LEFT 3 / DIAGONAL 2 / RIGHT 1 on the three candidates (the diagonal
	 * one through the score LUT), the winner's tag is the cell's pointer.  Xl holds the row's clean M of the previous
	 * column (left neighbour, and diagonal input of the row below); 9.25 instructions per row-step with pointers.  Only
	 * built with pointers: without them the int32 kernel's two instructions per cell (SDWA add, v_max3_i32: the sweep on the gap ramp, at_sweep.hip.h) win. */
	constexpr bool OVL = MODE == K_OVERLAP;
	static_assert(!OVL || (TB && TS == 2), "packed overlap: scores x4 with 2-bit tags, tracebacks");
	constexpr bool HASJ = MODE == K_FITJ;
	constexpr bool ISFIT = MODE == K_FIT || MODE == K_FITJ;
	static_assert(TS == 4 || TS == 2, "scores x16 or x4");
	constexpr int TMASK = (1 << TS) - 1;      /* tag bits of a score */
	/* local arg-max: the key of a cell carries its row-in-lane in the tag bits, so one running maximum covers TMASK + 1
	 * rows; lanes with more rows (8-lane groups: up to 19) keep one chain per TMASK + 1 rows */
	constexpr int CS = TMASK + 1, NCH = (K + CS - 1) / CS;
	/* priority tags in the low bits of every score make v_pk_max pick the reference's first-wins candidate.  They decide
	 * pointers, never values: the kernels without a pointer matrix (TB = false) run without them, three instructions
	 * per row-step less, and tag the three end-cell candidates of global only to report the start state. */
	constexpr int OTGL = TS == 4 ? 15 : 3, OTGM = TS == 4 ? 10 : 2, OTGU = 1;
	constexpr int TGL = TB ? OTGL : 0, TGM = TB ? OTGM : 0, TGU = TB ? OTGU : 0;
	/* Jump state with scores x16: the cell keeps the 4 bits of the other modes and the jump state's own pointer (did J open from
	 * M here?) goes to a BIT PLANE beside the cells -- one word per 4 rows x 4 steps and half -- 5 bits per cell instead of the 8 of a
	 * byte cell.  (Scores x4 carry score bits in the low nibble of J: they keep the byte cells.) */
	constexpr bool JPL = HASJ && TB && TS == 4 && AT_JPLANE;
	constexpr int KG = (K + 3) / 4;           /* groups of 4 rows of the plane */
	constexpr int PB = OVL ? AT_OVL_BITS : (HASJ && !JPL) ? 8 : 4;   /* pointer bits per cell and alignment */
	static_assert(!OVL || G == 64, "packed overlap: one 64-lane group");
	constexpr int SPW = 16 / PB;              /* steps per pointer word (each half holds its own alignment) */
	static_assert(G == 64 || G == 32 || G == 16 || G == 8 || G == 4, "group width");
	constexpr int NG = 64 / G;                /* groups per wavefront, 2 alignments each */
	/* steps per unrolled block.  The 16-lane kernels carry 7..13 rows per lane, so 4 steps (one pointer word) already
	 * unroll to ~6 KB of code and only one (masked) body is emitted: every launch starts with a cold instruction
	 * cache, and at ~1 ms per launch the first pass through tens of KB of straight-line code is measurable. */
	constexpr int BLK = G <= 16 ? 4 : 8;
	/* (two-pass kernels carry the replay and the walks' code as well: one body keeps the whole kernel inside the instruction cache,
	 * 64 KB for two CUs -- C3's forward sweep, replay and walks together are 58 KB with two bodies) */
	constexpr bool ONEBODY = G <= 16 || (CK > 0 && AT_TP_ONEBODY);
	constexpr int RPB = BLK / SPW;            /* pointer word rows per block */
	constexpr int RS = G * K;                 /* rows per strip (G < 64: the only strip) */
	constexpr int PADW = kPad / 4;            /* s2 bytes: 4 per word */
	const int lane = threadIdx.x;
	const int grp = lane / G, lg = lane % G;
	Slot<SMALL> mem;
	mem.g = SMALL ? nullptr : a.ws + (long long)blockIdx.x * a.ws_slot_words;
	PtrStore<PTRLDS> pm;
	pm.g = PTRLDS ? nullptr : a.ws + (long long)blockIdx.x * a.ws_slot_words;
	const int l1 = a.l1, l2 = a.l2, NL = a.ptr_lanes;
	const int o16 = a.o16, e16 = a.e16;
	uint32_t o2 = pk2(a.o16), e2 = pk2(a.e16);
	/* byte LUT for v_perm: pool bytes 0..3 = low bytes of {m,u,u,u}, 4..7 = high bytes */
	const int mS = OVL ? a.m16 | 2 : a.m16, uS = OVL ? a.u16 | 2 : a.u16;   /* overlap: the diagonal candidate's tag rides on the score */
	uint32_t lut_lo = ((uint32_t)mS & 0xffu) | (((uint32_t)uS & 0xffu) * 0x01010100u);
	uint32_t lut_hi = (((uint32_t)mS >> 8) & 0xffu) | ((((uint32_t)uS >> 8) & 0xffu) * 0x01010100u);
	uint32_t oL2 = pk2(a.o16 | 3), oR2 = pk2(a.o16 | 1);                    /* overlap: gap + LEFT / RIGHT tag (o16 is a multiple of 4) */
	asm volatile("" : "+v"(oL2), "+v"(oR2));
	/* constants live in VGPRs: VOP3 encodings take no 32-bit literals, and a literal would split and_or into two ops */
	uint32_t cClean = (uint32_t)(0xffff & ~TMASK) * 0x00010001u, cTagM = (uint32_t)TGM * 0x00010001u;
	uint32_t cTagL = (uint32_t)TGL * 0x00010001u, cTagU = (uint32_t)TGU * 0x00010001u;
	uint32_t cM3 = 0x00030003u, cM7 = 0x00070007u, cNib = 0x000f000fu, cF0 = 0x00f000f0u, cF000 = 0xf000f000u;
	uint32_t c8888 = 0x88888888u, c3333 = 0x33333333u;
	asm volatile("" : "+v"(c8888), "+v"(c3333));
	/* two-pass: the tags the replay's arithmetic carries, for the row checkpoint entries (ck_entry) of a sweep that runs without them */
	uint32_t cRTagL = (uint32_t)OTGL * 0x00010001u, cRTagM = (uint32_t)OTGM * 0x00010001u, cRTagU = (uint32_t)OTGU * 0x00010001u;
	(void)cRTagL; (void)cRTagM; (void)cRTagU;
	if constexpr (CK > 0) asm volatile("" : "+v"(cRTagL), "+v"(cRTagM), "+v"(cRTagU));
	/* jump state: J(i,j) = max(M(i,j-1) + g, J(i,j-1)) where the column may open, else J(i,j-1) (alignment.h:658-666);
	 * the left state holds M + o, so the opening candidate is (M + o) + (g - o), or -inf where opening is barred */
	uint32_t gmo2 = pk2(a.g16 - a.o16), neg2 = 0x80008000u;
	asm volatile("" : "+v"(gmo2), "+v"(neg2));
	uint32_t c1 = 0x00010001u, umm2 = pk2(a.u16 - a.m16), m2 = pk2(mS);       /* 8-bit alphabets: compare instead of LUT */
	asm volatile("" : "+v"(c1), "+v"(umm2), "+v"(m2));
	asm volatile("" : "+v"(o2), "+v"(e2), "+v"(lut_lo), "+v"(lut_hi));
	asm volatile("" : "+v"(cClean), "+v"(cTagM), "+v"(cTagL), "+v"(cTagU), "+v"(cM3), "+v"(cM7), "+v"(cNib), "+v"(cF0), "+v"(cF000));
	const int nstrips = (l1 + RS - 1) / RS;   /* host guarantees 1 when G < 64 */
	const int tbk_frame = (l2 + G - 1 + BLK - 1) / BLK;
	const int wps = tbk_frame * RPB * K;      /* pointer word rows per strip (the layout of the slot: always the frame's) */
	const int wpj = tbk_frame * (BLK / 4) * KG;   /* jump plane: word rows per strip, behind the cells of all strips */
	const int jpl_base0 = a.off_ptr + __mul24(nstrips * wps, NL);
	const long long nwork = (a.npairs + 2 * NG - 1) / (2 * NG);
	const int refoff = grp * 2 * a.off_refb;  /* my group's two s2 byte arrays */
	/* two-pass tracebacks: this wave's slot holds the checkpoints; row 0 of the matrix, the same for every alignment of the batch, is
	 * written once as the row checkpoint of a lane above lane 0: entry j = (L, M, U[, J]) of cell (0, j) */
	uint32_t *gs = TP ? a.ws + (long long)blockIdx.x * a.ws_slot_words : nullptr;
	constexpr int ES = ck_es<MODE>();
	const int ck_T = tbk_frame * BLK;         /* steps of a sweep */
	(void)gs; (void)ck_T;
	if constexpr (TP) {
		uint32_t *const brow = SPLIT ? a.ck_brow : gs + a.off_brow;   /* (split: every wave writes the batch's one copy, the same words) */
		for (int j = lane; j <= l2; j += 64) {
			int L, M, U;
			border16<MODE>(0, j, o16, e16, L, M, U);
			const uint2 en = ck_entry<HASJ>(pk2(sat16(L)), pk2(M), pk2(U), 0x80008000u, e2, o2, cRTagL, cRTagM, cRTagU);
			brow[j * ES] = en.x; brow[j * ES + 1] = en.y;
		}
	}

	while (wnext - wbase < nwork) {
		const long long wk = wnext - wbase;
		long long st_item = 0;
		(void)st_item;
		if (TP && AT_TP_STATS) st_item = (long long)__builtin_amdgcn_s_memtime();
		wnext = next_work(a.queue, lane);   /* consumed at the end of this work item: latency hidden */
		if constexpr (TP && SPLIT) gs = a.ck + wk * a.ck_item_words;
		const long long last = a.npairs - 1;
		long long pA = (wk * NG + grp) * 2 < a.npairs ? (wk * NG + grp) * 2 : last;
		long long pB = (wk * NG + grp) * 2 + 1 < a.npairs ? (wk * NG + grp) * 2 + 1 : last;
		int l1A = l1, l1B = l1, l2A = l2, l2B = l2;   /* the alignments' own extents (RAG: inside the frame l1 x l2) */
		/* the extents this work item is swept in.  Uniform batches: the batch's shape.  RAG: il2 = the largest l2 of the
		 * item's alignments (columns behind an alignment's own l2 feed nothing inside it); il1 = the frame's l1 for local
		 * (rows behind an alignment's own l1 are masked out of its arg-max), and for global / fit, whose end cells lie in
		 * row l1, the COMMON l1 of the item's alignments -- the host puts reads of equal length together. */
		int il1 = l1, il2 = l2;
		long long pout = 0;                            /* lanes 0 .. 2*NG-1: where the results of alignment `lane` go */
		bool filler = false;                           /* RAG: this lane's alignment only fills its work item up (a repeat of a pair another lane owns) */
		int own_len = l1 + l2;                         /* the ops slot of this lane's alignment holds this many (RAG: the pair's own len1 + len2) */
		if constexpr (RAG) {
			const long long pc = wk * 2 * NG + lane < a.npairs ? wk * 2 * NG + lane : last;
			/* the host marks the repeats with which it pads a run of equal l1 to whole work items as ~index: they are swept (their
			 * lanes cannot idle) but store nothing -- several lanes writing one pair's results at once was correct only because
			 * the values were equal */
			const int oraw = lane < 2 * NG ? a.order[pc] : 0;
			filler = oraw < 0;
			pout = filler ? (long long)~oraw : (long long)oraw;
			const int o1 = lane < 2 * NG ? a.len1[pout] : 0, o2l = lane < 2 * NG ? a.len2[pout] : 0;
			own_len = o1 + o2l;
			pA = __shfl((int)pout, 2 * grp); pB = __shfl((int)pout, 2 * grp + 1);
			l1A = __shfl(o1, 2 * grp); l1B = __shfl(o1, 2 * grp + 1);
			l2A = __shfl(o2l, 2 * grp); l2B = __shfl(o2l, 2 * grp + 1);
			int mx = o2l;
#pragma unroll
			for (int d = 1; d < 2 * NG; d <<= 1) mx = imax(mx, __shfl_xor(mx, d));
			il2 = __builtin_amdgcn_readfirstlane(mx);
			if constexpr (MODE != K_LOCAL) il1 = __builtin_amdgcn_readfirstlane(o1);
			const bool bad = lane < 2 * NG && (o1 > l1 || o2l > l2 || o1 < 1 || o2l < 1 || (MODE != K_LOCAL && o1 != il1));
			if (__any(bad)) {   /* a pair that does not fit the frame, is empty, or (global / fit) breaks the item's common l1 */
				if (lane < 2 * NG && wk * 2 * NG + lane < a.npairs && !filler) {
					a.score[pout] = INT32_MIN;
					if (a.nops) a.nops[pout] = -1;
				}
				continue;
			}
		} else {
			/* the caller promised one shape for the whole batch; a pair that breaks the promise would be swept with the
			 * wrong extents, so its work item is refused (domain error on its pairs) instead */
			const long long pc = wk * 2 * NG + lane;
			pout = pc;
			const bool bad = lane < 2 * NG && pc < a.npairs && (a.len1[pc] != l1 || a.len2[pc] != l2);
			if (__any(bad)) {
				if (lane < 2 * NG && pc < a.npairs) {
					a.score[pc] = INT32_MIN;
					if (a.nops) a.nops[pc] = -1;
				}
				continue;
			}
		}
		const int tbk = RAG ? (il2 + G - 1 + BLK - 1) / BLK : tbk_frame;
		const int lastlane = il1 > 0 ? ((il1 - 1) % RS) / K : 0;   /* lane-in-group owning row l1 */
		const int rl = il1 > 0 ? ((il1 - 1) % RS) % K : 0;
		const uint32_t *qA = a.seq + a.woff1[pA], *qB = a.seq + a.woff1[pB];
		const uint32_t *rA = a.seq + a.woff2[pA], *rB = a.seq + a.woff2[pB];

		/* ---- stage both s2 of my group as bytes (coalesced int32 reads of the 2-bit words) ---- */
		{
			constexpr int BPW = 32 / BITS;         /* bases per sequence word */
			const int nw2 = (il2 + BPW - 1) / BPW;
			const int nwA = (l2A + BPW - 1) / BPW, nwB = (l2B + BPW - 1) / BPW;   /* RAG: never read behind an alignment's own words */
			for (int w = lg; w < nw2; w += G) {
				const uint32_t va = rA[RAG ? imin(w, nwA - 1) : w], vb = rB[RAG ? imin(w, nwB - 1) : w];
				if constexpr (BITS == 8) {         /* already one byte per base */
					mem.st(refoff + PADW + w, va);
					mem.st(refoff + a.off_refb + PADW + w, vb);
				} else {
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const uint32_t ba = (va >> (8 * q)) & 0xffu, bb = (vb >> (8 * q)) & 0xffu;
						mem.st(refoff + PADW + 4 * w + q, (ba & 3u) | ((ba & 0xcu) << 6) | ((ba & 0x30u) << 12) | ((ba & 0xc0u) << 18));
						mem.st(refoff + a.off_refb + PADW + 4 * w + q, (bb & 3u) | ((bb & 0xcu) << 6) | ((bb & 0x30u) << 12) | ((bb & 0xc0u) << 18));
					}
				}
			}
		}
		if constexpr (HASJ) {
			for (int w = lane; w < a.nsm; w += 64) mem.st(a.off_sm + w, a.sitemask[w]);
		}
		/* ---- boundary row 0 (identical for every alignment of the batch) ---- */
		for (int j = lane; j <= l2; j += 64) {
			int L, M, U;
			border16<MODE>(0, j, o16, e16, L, M, U);
			int x = imax3(L | TGL, M | TGM, U | TGU);
			const int ld = imax(sat16((L | TGL) + e16), sat16((M | TGM) + o16));
			if constexpr (OVL) x = j == 0 ? 0 : kNeg16;        /* M(0,j) = -inf, then M(i,0) = 0 for all i (:937-938) */
			mem.st2(a.off_bound + 2 * j, pk2(x), pk2(ld));
		}
		mem.sync();

		/* running results, per half */
		int gbi[2] = {INT32_MAX, INT32_MAX}, gbj[2] = {INT32_MAX, INT32_MAX}, gbs[2] = {INT32_MIN, INT32_MIN};
		uint32_t bestM = pk2(a.thresh16), bestMj = 0, bestL = pk2(a.thresh16), bestLj = 0;   /* fit scans */
		uint32_t capL = 0, capM = 0, capU = 0;         /* RAG global: the three states of each alignment's own cell (l1, l2) */
		/* Xl: X' of my rows at the previous column (the diagonal input of the row below).  Two copies used in turn
		 * (step parity), so that the old value can be read while the new one is written without register moves. */
		uint32_t Mo_l[K], U_l[K], Xl[2][K], L_l[K], J_l[K];
		uint32_t Lrow = 0;                             /* fit: L of row l1 at the previous column (what the end-cell scan reads) */

		for (int s = 0; s < nstrips; ++s) {
			const int base = s * RS;
			const int i0 = base + lg * K;
			const int nl = imin(G, (il1 - base + K - 1) / K);
			const bool laststrip = s == nstrips - 1;
			const bool wb = G == 64 && !laststrip;
			uint32_t qsel[K], acc[K], keymask[K];
			uint32_t accH[OVL && PB == 2 ? K : 1];   /* overlap, 2-bit cells: the nibble word of steps 4..7 until it joins that of steps 0..3 */
			uint32_t jA[JPL ? KG : 1], jB[JPL ? KG : 1];   /* jump plane: per group of 4 rows, the J nibbles of the block's steps on their way into one word */
			(void)accH; (void)jA; (void)jB;
			if constexpr (JPL) {
#pragma unroll
				for (int g = 0; g < KG; ++g) { jA[g] = 0; jB[g] = 0; }   /* (their words are OR-ed together: no stray bits from before the lane's first step) */
			}
			const int jpl_base = jpl_base0 + __mul24(s * wpj, NL);
#pragma unroll
			for (int r = 0; r < K; ++r) {
				const int qi = imin(i0 + r, l1A - 1), qj = imin(i0 + r, l1B - 1);
				uint32_t ca, cb;
				if constexpr (BITS == 2) { ca = (qA[qi >> 4] >> ((qi & 15) * 2)) & 3u; cb = (qB[qj >> 4] >> ((qj & 15) * 2)) & 3u; }
				else { ca = (qA[qi >> 2] >> ((qi & 3) * 8)) & 0xffu; cb = (qB[qj >> 2] >> ((qj & 3) * 8)) & 0xffu; }
				/* my query bases of both alignments as the bytes {qB|4, qB, qA|4, qA}: xor-ing the s2 bytes {b,b,a,a}
				 * onto it gives the LUT selector directly (codes are < 4, so the |4 survives the xor) */
				qsel[r] = (ca * 0x00000101u + cb * 0x01010000u) | (BITS == 2 ? 0x04000400u : 0u);
				acc[r] = 0;                       /* (steps outside the matrix leave their bits as they are: never read) */
				/* rows past l1 (only in the last lane that owns rows): their key collapses to the bare row tag,
				 * which every real row of the lane beats (smaller r = larger tag, score >= 0) */
				keymask[r] = (i0 + r < l1A ? (uint32_t)(0xffff & ~TMASK) : 0u) | (i0 + r < l1B ? (uint32_t)(0xffff & ~TMASK) << 16 : 0u);
				int L, M, U;
				border16<MODE>(i0 + r + 1, 0, o16, e16, L, M, U);
				L = sat16(L);
				Mo_l[r] = pk2(sat16((M | TGM) + o16));
				U_l[r] = pk2(U | TGU);
				L_l[r] = pk2(L | TGL);
				if constexpr (ISFIT) Lrow = L_l[r];       /* (fit: the same border value in every row) */
				J_l[r] = neg2;                    /* J is -inf on both borders (:616, :622) */
				Xl[0][r] = pk2(imax3(L | TGL, M | TGM, U | TGU));
				if constexpr (OVL) Xl[0][r] = 0;  /* M(i,0) = 0 */
				Xl[1][r] = Xl[0][r];
			}
			/* two-pass: (L, M, U[, J]) of my last row as of my latest step -- the border's until my first one; entry 0 of my row checkpoints */
			uint32_t ckL = 0, ckM = 0, ckU = 0, ckJ = neg2;
			int ck_p = 0;                          /* where the entries of the current block of steps go (ck_rck_word: all in one chunk) */
			(void)ckL; (void)ckM; (void)ckU; (void)ckJ; (void)ck_p;
			if constexpr (TP) {
				int L, M, U;
				border16<MODE>(i0 + K, 0, o16, e16, L, M, U);
				ckL = pk2(sat16(L)); ckM = pk2(M); ckU = pk2(U);
				const int p0 = a.off_rck + ck_rck_word<ES>(0, lane);
				const uint2 en = ck_entry<HASJ>(ckL, ckM, ckU, ckJ, e2, o2, cRTagL, cRTagM, cRTagU);
				gs[p0] = en.x; gs[p0 + 1] = en.y;
			}
			uint32_t A_prev = Xl[0][K - 1], B_prev = 0, Ad;
			{
				int L, M, U;
				border16<MODE>(base, 0, o16, e16, L, M, U);
				L = sat16(L);
				Ad = pk2(imax3(L | TGL, M | TGM, U | TGU));
				if constexpr (OVL) Ad = 0;
			}
			uint32_t best[NCH], bt[NCH];           /* local: per-lane (key, step) of this strip, per chain of rows */
#pragma unroll
			for (int c = 0; c < NCH; ++c) { best[c] = 0x80008000u; bt[c] = 0; }
			auto load_bound = [&](int t0, uint32_t &bx, uint32_t &bl) {
				const int jn = imin(t0 + 1 + (lane & (BLK - 1)), l2);   /* lane k of a group: the boundary cell of step k (row_shl<k> brings it to lane 0) */
				const uint2 v = mem.ld2(a.off_bound + 2 * jn);
				bx = v.x; bl = v.y;
			};
			uint32_t bx, bl, bxn, bln;
			load_bound(0, bx, bl);
			const int ptr_base = a.off_ptr + s * wps * NL;

			for (int blk = 0; blk < tbk; ++blk) {
				const int t0 = blk * BLK;
				if constexpr (TP) ck_p = a.off_rck + ck_rck_word<ES>(t0 + 1, lane);
				load_bound(t0 + BLK, bxn, bln);
				/* ---- s2 windows: bytes t0-lg .. t0-lg+7 of both alignments ---- */
				uint32_t wA[2], wB[2];
				{
					const int e0 = t0 - lg + kPad;
					const int w = refoff + (e0 >> 2);
					const int sh = (e0 & 3) * 8;
					const uint32_t a0 = mem.ld(w), a1 = mem.ld(w + 1);
					const uint32_t b0 = mem.ld(a.off_refb + w), b1 = mem.ld(a.off_refb + w + 1);
					wA[0] = __builtin_amdgcn_alignbit(a1, a0, sh);
					wB[0] = __builtin_amdgcn_alignbit(b1, b0, sh);
					wA[1] = 0; wB[1] = 0;
					if constexpr (BLK == 8) {
						const uint32_t a2 = mem.ld(w + 2), b2 = mem.ld(a.off_refb + w + 2);
						wA[1] = __builtin_amdgcn_alignbit(a2, a1, sh);
						wB[1] = __builtin_amdgcn_alignbit(b2, b1, sh);
					}
				}
				uint32_t sm = 0;
				if constexpr (HASJ) {
					const int e0 = t0 - lg + 1 + 64;
					const uint32_t w0 = mem.ld(a.off_sm + (e0 >> 5)), w1 = mem.ld(a.off_sm + (e0 >> 5) + 1);
					sm = __builtin_amdgcn_alignbit(w1, w0, e0 & 31);
				}
				const int jm1_0 = t0 - lg;

				auto step = [&](auto KC, auto MASKED) {
					constexpr int k = decltype(KC)::value;
					constexpr bool masked = decltype(MASKED)::value;
					constexpr int kk = k & 3, hw = k >> 2;
					/* selector picking byte kk of the A window (pool 0..3) twice and of the B window (pool 4..7) twice */
					constexpr uint32_t SELK = (uint32_t)kk * 0x00000101u + (uint32_t)(4 + kk) * 0x01010000u;
					const int t = t0 + k;
					uint32_t tpk = pk2(t);
					asm("" : "+v"(tpk));
					const uint32_t Aup = grp_up1<G>((uint32_t)row_shl<k>((int)bx), A_prev);
					const uint32_t Bup = grp_up1<G>((uint32_t)row_shl<k>((int)bl), B_prev);
					const int jm1 = jm1_0 + k;
					bool active = true;
					if constexpr (masked) active = lg < nl && (unsigned)jm1 < (unsigned)il2;
					if (active) {
						uint32_t gopen = neg2;
						if constexpr (HASJ) gopen = ((sm >> k) & 1u) ? gmo2 : neg2;
						if constexpr (OVL) {
							/* end-cell scan of row l1, columns 0..l2-1 (:951-959), one column behind the sweep */
							if (laststrip) {
								const uint32_t jpk = pk2(jm1);
								uint32_t vM = pick<K>(Xl[k & 1], rl);
								if constexpr (RAG) {
									/* an alignment's scan ends at its own column l2 - 1 */
									const uint32_t cm = (((uint32_t)((jm1 - l2A) >> 31)) & 0xffffu) | (((uint32_t)((jm1 - l2B) >> 31)) << 16);
									vM = vbfi(cm, vM, neg2);
								}
								uint32_t dM = psub(bestM, vM);
								asm("" : "+v"(dM));
								bestMj = vbfi(pneg(dM), jpk, bestMj); bestM = pmax(bestM, vM);
							}
						}
						if constexpr (ISFIT) {
							/* end-cell scan of row l1, columns 0..l2-1 (:676-690), one column behind the sweep;
							 * every lane scans its own row rl, only the owner of row l1 is read at the end.  (rl is wave-uniform, but a
							 * scalar branch per row instead of the two select chains measured 3 % slower with pointers, 8 % without:
							 * it cuts the unrolled step into 19 basic blocks.) */
							if (laststrip) {
								const uint32_t jpk = pk2(jm1);
								uint32_t vM = psub(pick<K>(Mo_l, rl), o2);
								uint32_t vL = Lrow;
								if constexpr (RAG) {
									/* an alignment's scan ends at its own column l2 - 1 */
									const uint32_t cm = (((uint32_t)((jm1 - l2A) >> 31)) & 0xffffu) | (((uint32_t)((jm1 - l2B) >> 31)) << 16);
									vM = vbfi(cm, vM, neg2); vL = vbfi(cm, vL, neg2);
								}
								uint32_t dM = psub(bestM, vM);
								asm("" : "+v"(dM));
								bestMj = vbfi(pneg(dM), jpk, bestMj); bestM = pmax(bestM, vM);
								uint32_t dL = psub(bestL, vL);
								asm("" : "+v"(dL));
								bestLj = vbfi(pneg(dL), jpk, bestLj); bestL = pmax(bestL, vL);
							}
						}
						/* step k's byte of each window against each of my query bases */
						const uint32_t selw = __builtin_amdgcn_perm(wB[hw], wA[hw], SELK);   /* [b, b, a, a] */
						uint32_t diag = Ad, lraw = Bup, up = 0, cmax[NCH];
						uint32_t capmask = 0;
						if constexpr (RAG && MODE == K_GLOBAL) {
							/* the sweep passes each alignment's end cell (l1, l2 of its own) on its way through the frame: the
							 * lane that owns row l1 keeps the cell's three states when its column comes by */
							if (lg == lastlane) capmask = (jm1 + 1 == l2A ? 0xffffu : 0u) | (jm1 + 1 == l2B ? 0xffff0000u : 0u);
						}
						uint32_t upv = Aup;                         /* overlap: M of the row above in this column */
						uint32_t jn = 0;                            /* jump plane: the J nibbles of the current group of 4 rows */
						(void)jn;
#pragma unroll
						for (int r = 0; r < K; ++r) {
							/* x = s2 byte ^ query byte (0 = match): selector {xB+4, xB, xA+4, xA} -> the two 16-bit scores */
							uint32_t S;
							if constexpr (BITS == 2) {
								S = __builtin_amdgcn_perm(lut_hi, lut_lo, selw ^ qsel[r]);
							} else {
								/* bytes: x = {b^qB, b^qB, a^qA, a^qA}; per 16-bit half 0 = match.  z = min(x, 1), S = m + z * (u - m) */
								const uint32_t z = pminu(selw ^ qsel[r], c1);
								S = pmad(z, umm2, m2);
							}
							if constexpr (OVL) {
								const uint32_t left = Xl[k & 1][r];
								const uint32_t x = pmax(pmax(padd(left, oL2), padd(diag, S)), padd(upv, oR2));
								const uint32_t Mcl = x & cClean;
								constexpr int sw = k % SPW;         /* pointer word: see the affine modes below */
								if constexpr (PB == 4) {
									if constexpr (sw == 0) acc[r] = x;
									else if constexpr (sw == 1) acc[r] = __builtin_amdgcn_perm(x, acc[r], 0x06020400u);
									else if constexpr (sw == 2) acc[r] = vbfi(cF0, pshln<4>(x), acc[r]);
									else acc[r] = vbfi(cF000, pshln<12>(x), acc[r]);
								} else {
									/* 2-bit cells: the walk reads the tag alone, so the nibble words of steps 0..3 and 4..7 share one word --
									 * bits [1:0] of every nibble from the first, [3:2] from the second (two instructions per 8 steps more,
									 * half the pointer bytes) */
									if constexpr (sw == 0) acc[r] = x;
									else if constexpr (sw == 1) acc[r] = __builtin_amdgcn_perm(x, acc[r], 0x06020400u);
									else if constexpr (sw == 2) acc[r] = vbfi(cF0, pshln<4>(x), acc[r]);
									else if constexpr (sw == 3) acc[r] = vbfi(cF000, pshln<12>(x), acc[r]);
									else if constexpr (sw == 4) accH[r] = x;
									else if constexpr (sw == 5) accH[r] = __builtin_amdgcn_perm(x, accH[r], 0x06020400u);
									else if constexpr (sw == 6) accH[r] = vbfi(cF0, pshln<4>(x), accH[r]);
									else accH[r] = vbfi(cF000, pshln<12>(x), accH[r]);   /* (the two words join where they are stored: a lane may sit out step 7) */
								}
								diag = left;
								upv = Mcl;
								up = Mcl;
								Xl[(k & 1) ^ 1][r] = Mcl;
								continue;
							}
							uint32_t Mraw = padd(diag, S);
							if constexpr (MODE == K_LOCAL) Mraw = pmax(Mraw, 0u);
							const uint32_t Mc = TB ? vandor(Mraw, cClean, cTagM) : Mraw;
							const uint32_t Lc = TB ? (lraw | cTagL) : lraw;
							const uint32_t Uraw = pmax(Mo_l[r], padd(U_l[r], e2));
							const uint32_t Uc = TB ? vandor(Uraw, cClean, cTagU) : Uraw;
							const uint32_t Mo = padd(Mc, o2);
							uint32_t Xo = pmax(pmax(Lc, Mc), Uc);
							uint32_t Jraw = 0;
							if constexpr (HASJ) {
								Jraw = pmax(padd(Mo_l[r], gopen), J_l[r]);     /* M first: tag 10 beats J's tag 0 on ties */
								const uint32_t Jc = TB ? (Jraw & cClean) : Jraw;
								Xo = pmax(Xo, Jc);
								J_l[r] = Jc;
							}
							const uint32_t Ld = pmax(padd(Lc, e2), Mo);
							if (TP && r == K - 1) {                /* what the lane below needs of this row */
								ckL = Lc; ckM = Mc; ckU = Uc;
								if constexpr (HASJ) ckJ = J_l[r];
							}
							if constexpr (RAG && MODE == K_GLOBAL) {
								if (r == rl) { capL = vbfi(capmask, Lc, capL); capM = vbfi(capmask, Mc, capM); capU = vbfi(capmask, Uc, capU); }
							}
							if constexpr (TB) {
								/* the cell's pointer bits, in the low bits of each 16-bit half (what lies above them is junk
								 * until the word is assembled): [1:0] pM, bit 2: L extended, bit 3: the U winner's tag bit, bit 4: J opened */
								uint32_t c;
								if constexpr (TS == 4) {
									/* tags L 15 / M 10 / U 1: bit 2 of the L winner (1111 ext / 1010 open), bit 3 of the U winner
									 * (1010 open / 0001 ext).  (Not their xor: a saturated -inf carries no tag at all.) */
									c = vbfi(cM7, vbfi(cM3, Mraw, lraw), Uraw);
								} else {
									/* 2-bit tags: bit 0 of the L winner (3 ext / 2 open) and of the U winner (1 ext / 2 open) */
									c = vbfi(cM7, vbfi(cM3, Mraw, pshln<2>(lraw)), pshln<3>(Uraw));
								}
								if constexpr (HASJ && !JPL) c = vbfi(cNib, c, pshln<TS == 4 ? 1 : 3>(Jraw));   /* J winner's tag: M's (open) or 0 */
								if constexpr (JPL) {
									/* The low nibble of the J winner is M's tag 1010 (J opened here) or 0000 (extended, or -inf): the nibbles of 4
									 * rows are gathered like the steps of a cell word -- [row 3 | row 1 | row 2 | row 0] -- and because bits 2 and 0
									 * of every nibble are 0, the gathered words of the block's 4 steps interleave into ONE word: bit 1 step 0,
									 * bit 3 step 1, bit 0 step 2, bit 2 step 3 of each row's nibble (1.5 instructions per row-step). */
									const int q = r & 3, gq = r >> 2;
									if (q == 0) jn = Jraw;
									else if (q == 1) jn = __builtin_amdgcn_perm(Jraw, jn, 0x06020400u);
									else if (q == 2) jn = vbfi(cF0, pshln<4>(Jraw), jn);
									else jn = vbfi(cF000, pshln<12>(Jraw), jn);
									if (q == 3 || r == K - 1) {
										/* (a last group of fewer than 4 rows: what the gather left in the other nibbles must not leak into the interleave) */
										if (q == 0) jn &= 0x000f000fu; else if (q == 1) jn &= 0x0f0f0f0fu; else if (q == 2) jn &= 0x0fff0fffu;
										constexpr int sj = k & 3;
										if constexpr (sj == 0) jA[gq] = jn;
										else if constexpr (sj == 1) jA[gq] = vbfi(c8888, jn, jA[gq]);
										else if constexpr (sj == 2) jB[gq] = jn;
										else jB[gq] = vbfi(c8888, jn, jB[gq]);   /* (jA: steps 1 | 0, jB: steps 3 | 2 in bits 3 | 1 of every nibble; they interleave
										                                          * where they are stored -- a lane may sit out the block's last step.  Stale
										                                          * nibbles of a step the lane sat out only ever reach that step's own bits.) */
									}
								}
								/* assembling the word of SPW steps costs 1.25 instructions per step (4-bit cells) or 0.5 (8-bit cells)
								 * instead of a mask and a shift-or per step: bytes are gathered with v_perm, which does not care what
								 * the unused bits of a byte hold; half h of the word: 8-bit cells [step 1 | step 0], 4-bit cells
								 * [step 3 | step 1 | step 2 | step 0] (nibbles, most significant first) */
								constexpr int sw = k % SPW;
								if constexpr (sw == 0) acc[r] = c;
								else if constexpr (sw == 1) acc[r] = __builtin_amdgcn_perm(c, acc[r], 0x06020400u);
								else if constexpr (sw == 2) acc[r] = vbfi(cF0, pshln<4>(c), acc[r]);
								else acc[r] = vbfi(cF000, pshln<12>(c), acc[r]);
							}
							if constexpr (MODE == K_LOCAL) {
								uint32_t rt = (uint32_t)(TMASK - r % CS) * 0x00010001u;
								const uint32_t key = (Mraw & keymask[r]) | rt;
								cmax[r / CS] = r % CS == 0 ? key : pmax(cmax[r / CS], key);
							}
							diag = Xl[k & 1][r];
							Xl[(k & 1) ^ 1][r] = Xo;
							lraw = Ld;
							up = Xo;
							Mo_l[r] = Mo; U_l[r] = Uc;
							if constexpr (MODE == K_GLOBAL) L_l[r] = Lc;
							if constexpr (ISFIT) Lrow = r == rl ? Lc : Lrow;   /* (one register instead of K: the scan reads row l1 only) */
						}
						if constexpr (MODE == K_LOCAL) {
							if constexpr (RAG) {
								/* columns behind an alignment's own l2 are swept (they feed nothing inside it) but must
								 * not win its arg-max */
								const uint32_t cm = (((uint32_t)((jm1 - l2A) >> 31)) & 0xffffu) | (((uint32_t)((jm1 - l2B) >> 31)) << 16);
#pragma unroll
								for (int c = 0; c < NCH; ++c) cmax[c] = vbfi(cm, cmax[c], neg2);
							}
#pragma unroll
							for (int c = 0; c < NCH; ++c) {
								uint32_t dlt = psub(best[c], cmax[c]);
								asm("" : "+v"(dlt));                    /* keep hipcc from turning this into 2 cmp + 2 cndmask + perm */
								const uint32_t g = pneg(dlt);           /* 0xffff where cmax > best */
								bt[c] = vbfi(g, tpk, bt[c]);
								best[c] = pmax(best[c], cmax[c]);
							}
						}
						A_prev = up; B_prev = lraw;
						if (wb && lane == 63) mem.st2(a.off_bound + 2 * (jm1 + 1), up, lraw);
					}
					Ad = Aup;
					if constexpr (TP) {
						/* row checkpoint entry t + 1: my last row after this step (a step outside the matrix repeats the entry before) */
						static_assert(!TP || AT_CK_ROW_TILE <= 1 || BLK % AT_CK_ROW_TILE == 0 || AT_CK_ROW_TILE % BLK == 0, "a tile of row checkpoints and a block of steps: one holds the other");
						constexpr int RT = AT_CK_ROW_TILE > 1 ? AT_CK_ROW_TILE : 1;
						/* (ck_p: the entry of step t0.  Tiles of up to BLK steps: step k lies k / RT tiles on; longer tiles hold the whole block of steps) */
						constexpr int so = RT >= BLK ? k * ES : (k / RT) * 64 * RT * ES + (k % RT) * ck_rck_step<ES>();
						*(uint2 *)(gs + ck_p + so) = ck_entry<HASJ>(ckL, ckM, ckU, ckJ, e2, o2, cRTagL, cRTagM, cRTagU);
					}
					if constexpr (TB) {
						if constexpr (OVL && PB == 2) {
							if constexpr ((k + 1) % SPW == 0) {
#pragma unroll
								for (int r = 0; r < K; ++r) acc[r] = vbfi(c3333, acc[r], accH[r] << 2);
							}
						}
						if ((k + 1) % SPW == 0 && lane < NL) {
							if constexpr (PTRLDS) {
#pragma unroll
								for (int r = 0; r < K; ++r) pm.st(ptr_base + pidx<true, K>(blk * RPB + k / SPW, r, lane, NL), acc[r]);
							} else {
								const int wr = blk * RPB + k / SPW;
#pragma unroll
								for (int r = 0; r + 3 < K; r += 4)
									pm.st4(ptr_base + pidx<false, K>(wr, r, lane, NL), acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
								constexpr int R4 = K / 4 * 4;
								if constexpr (K - R4 == 3) {
									pm.st(ptr_base + pidx<false, K>(wr, R4, lane, NL), acc[R4]);
									pm.st(ptr_base + pidx<false, K>(wr, R4 + 1, lane, NL), acc[R4 + 1]);
									pm.st(ptr_base + pidx<false, K>(wr, R4 + 2, lane, NL), acc[R4 + 2]);
								} else if constexpr (K - R4 == 2) {
									pm.st2(ptr_base + pidx<false, K>(wr, R4, lane, NL), acc[R4], acc[R4 + 1]);
								} else if constexpr (K - R4 == 1) {
									pm.st(ptr_base + pidx<false, K>(wr, R4, lane, NL), acc[R4]);
								}
							}
						}
						if constexpr (JPL) {
							if ((k & 3) == 3 && lane < NL) {
								const int wr = blk * (BLK / 4) + k / 4;
								uint32_t jO[KG];
#pragma unroll
								for (int g = 0; g < KG; ++g) jO[g] = jA[g] | (jB[g] >> 1);
								if constexpr (PTRLDS) {
#pragma unroll
									for (int g = 0; g < KG; ++g) pm.st(jpl_base + pidx<true, KG>(wr, g, lane, NL), jO[g]);
								} else {
#pragma unroll
									for (int g = 0; g + 3 < KG; g += 4)
										pm.st4(jpl_base + pidx<false, KG>(wr, g, lane, NL), jO[g], jO[g + 1], jO[g + 2], jO[g + 3]);
									constexpr int G4 = KG / 4 * 4;
									if constexpr (KG - G4 == 3) {
										pm.st(jpl_base + pidx<false, KG>(wr, G4, lane, NL), jO[G4]);
										pm.st(jpl_base + pidx<false, KG>(wr, G4 + 1, lane, NL), jO[G4 + 1]);
										pm.st(jpl_base + pidx<false, KG>(wr, G4 + 2, lane, NL), jO[G4 + 2]);
									} else if constexpr (KG - G4 == 2) {
										pm.st2(jpl_base + pidx<false, KG>(wr, G4, lane, NL), jO[G4], jO[G4 + 1]);
									} else if constexpr (KG - G4 == 1) {
										pm.st(jpl_base + pidx<false, KG>(wr, G4, lane, NL), jO[G4]);
									}
								}
							}
						}
					}
				};
				using T = std::true_type;
				using F = std::false_type;
#define AT_STEPS16(M)                                                                                    \
	step(std::integral_constant<int, 0>{}, M{}); step(std::integral_constant<int, 1>{}, M{});            \
	step(std::integral_constant<int, 2>{}, M{}); step(std::integral_constant<int, 3>{}, M{});            \
	if constexpr (BLK == 8) {                                                                            \
		step(std::integral_constant<int, 4>{}, M{}); step(std::integral_constant<int, 5>{}, M{});        \
		step(std::integral_constant<int, 6>{}, M{}); step(std::integral_constant<int, 7>{}, M{});        \
	}
				if constexpr (ONEBODY) { AT_STEPS16(T) }
				else if (t0 >= nl - 1 && t0 + BLK <= il2) { AT_STEPS16(F) }
				else { AT_STEPS16(T) }
#undef AT_STEPS16
				bx = bxn; bl = bln;
				if constexpr (TP) {
					/* column checkpoint c: (M + o, U[, J]) of my K rows after step c * CK - 1, in 16-byte chunks [c][chunk][lane] */
					if ((t0 + BLK) % CK == 0) {
						constexpr int NV = HASJ ? 3 : 2, NQ = ck_nq<MODE, K>();
						const int cb = a.off_cck, cidx = (t0 + BLK) / CK;
						auto val = [&](auto XC) -> uint32_t {
							constexpr int x = decltype(XC)::value;
							if constexpr (x < K) return Mo_l[x];
							else if constexpr (x < 2 * K) return U_l[x - K];
							else if constexpr (x < NV * K) return J_l[x - 2 * K];
							else return 0u;
						};
						auto chunk = [&](auto QC) {
							constexpr int q = decltype(QC)::value;
							*(uint4 *)(gs + cb + ck_cck_word<NQ>(cidx, lane, q)) = make_uint4(val(std::integral_constant<int, 4 * q>{}), val(std::integral_constant<int, 4 * q + 1>{}),
							                                          val(std::integral_constant<int, 4 * q + 2>{}), val(std::integral_constant<int, 4 * q + 3>{}));
						};
						static_for<NQ>(chunk);
					}
				}
			}
			if constexpr (MODE == K_LOCAL) {
				/* fold this strip's per-lane winner into the running one (earlier strips = smaller i win ties) */
#pragma unroll
				for (int c = 0; c < NCH; ++c) {        /* chains in row order: the earlier chain keeps a tie */
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int key = half(best[c], h);
						const int sc = key & ~TMASK;
						const int row = i0 + c * CS + (TMASK - (key & TMASK)) + 1;
						if (key != kNeg16 && row <= (h == 0 ? l1A : l1B) && sc > gbs[h]) {
							gbs[h] = sc;
							gbi[h] = row;
							gbj[h] = (int)((bt[c] >> (16 * h)) & 0xffffu) - lg + 1;
						}
					}
				}
			}
			mem.sync();
		}

		pm.ready();
		/* ================= end cells: lane gh receives the result of alignment gh = 2*group + half ================= */
		int my_sc = 0, my_ci = 0, my_cj = 0, my_st = 2;
		bool my_ok = true;
		if constexpr (MODE == K_LOCAL) {
			/* every group folds its lanes' winners at once, both halves: afterwards each lane of a group holds the group's */
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				int bs = gbs[h], bi = gbi[h], bj = gbj[h];
				for (int d = G / 2; d >= 1; d >>= 1) {
					const int ob = __shfl_xor(bs, d), oi = __shfl_xor(bi, d), oj = __shfl_xor(bj, d);
					const bool take = ob > bs || (ob == bs && (oi < bi || (oi == bi && oj < bj)));
					if (take) { bs = ob; bi = oi; bj = oj; }
				}
				gbs[h] = bs; gbi[h] = bi; gbj[h] = bj;
			}
		}
		{
			/* lane a < 2*NG fetches alignment a's result from its group (a >> 1) and half (a & 1) with lane-indexed reads: one pass for
			 * all 4 .. 32 alignments of the item instead of a loop over them */
			const int h = lane & 1;
			const int glane = (lane >> 1) * G;     /* lane 0 of the group (lanes >= 2*NG read garbage they never use) */
			const int own = (glane + lastlane) & 63;
			int sc16 = 0, ci = 0, cj = 0, st = 2;
			bool ok = true;
			auto from = [&](uint32_t v, int src) { return (uint32_t)__shfl((int)v, src & 63); };
			if constexpr (MODE == K_LOCAL) {
				/* (every read is executed by all lanes, then selected: a lane-indexed read inside a branch on h would find its
				 * source lane switched off) */
				const int s0 = (int)from((uint32_t)gbs[0], glane), s1 = (int)from((uint32_t)gbs[1], glane);
				const int i0 = (int)from((uint32_t)gbi[0], glane), i1 = (int)from((uint32_t)gbi[1], glane);
				const int j0 = (int)from((uint32_t)gbj[0], glane), j1 = (int)from((uint32_t)gbj[1], glane);
				sc16 = h ? s1 : s0; ci = h ? i1 : i0; cj = h ? j1 : j0;
				st = 2;
			} else if constexpr (MODE == K_GLOBAL) {
				int eL, eM, eU;
				if constexpr (RAG) {
					eL = half(from(capL, own), h);
					eM = half(from(capM, own), h);
					eU = half(from(capU, own), h);
					const int ca = (int)from((uint32_t)l2A, glane), cb = (int)from((uint32_t)l2B, glane);
					cj = h ? cb : ca;
				} else {
					eL = half(from(pick<K>(L_l, rl), own), h);
					eM = half(psub(from(pick<K>(Mo_l, rl), own), pk2(o16)), h);
					eU = half(from(pick<K>(U_l, rl), own), h);
					cj = l2;
				}
				const int x = TB ? imax3(eL, eM, eU) : imax3(eL | OTGL, eM | OTGM, eU | OTGU);   /* max5(L,M,U) first-wins :466 */
				sc16 = x; st = x & 3; ci = il1;
			} else if constexpr (OVL) {
				sc16 = half(from(bestM, own), h);   /* >= 0: column 0 holds 0 (:951-959) */
				cj = half(from(bestMj, own), h);
				ci = il1; st = 2;
				ok = sc16 > a.thresh16;
			} else {
				const int bM = half(from(bestM, own), h);
				const int jM = half(from(bestMj, own), h);
				const int bL = half(from(bestL, own), h);
				const int jL = half(from(bestLj, own), h);
				ci = il1;
				if ((bL >> TS) > (bM >> TS) && bL > a.thresh16) { sc16 = bL; st = 3; cj = jL; }
				else { sc16 = bM; st = 2; cj = jM; }
				ok = sc16 > a.thresh16;
			}
			my_sc = sc16; my_ci = ci; my_cj = cj; my_st = st; my_ok = ok;
		}
		if constexpr (TP && SPLIT) {
			/* ================= two-pass tracebacks, pass 2 a kernel of its own (at_walk16.hip.h): leave what it starts from ================= */
			const long long pin = wk * 2 * NG + lane;
			if (lane < 2 * NG && pin < a.npairs) {
				a.score[pin] = my_ok ? (my_sc >> TS) : INT32_MIN;
				if (a.end_i) a.end_i[pin] = my_ci;
				if (a.end_j) a.end_j[pin] = my_cj;
				if (a.state) a.state[pin] = my_st == 3 ? 1 : my_st == 2 ? 2 : 3;
				a.tp_end[pin] = make_int4(my_ci, my_cj, my_st, my_ok ? 1 : 0);
			}
		} else if constexpr (TP) {
			/* ================= two-pass tracebacks: rounds of { replay the blocks the walks are heading for; walk } ================= */
			constexpr int LCB = ck_log2(CK);
			constexpr int KG2 = (K + 3) / 4;
			const int g2 = lane >> 1, h = lane & 1;
			const long long pin = wk * 2 * NG + lane;
			const bool mine = lane < 2 * NG && pin < a.npairs;
			int ci = my_ci, cj = my_cj, st = my_st, cnt = 0;
			bool ok = my_ok;
			const bool walks = mine && a.nops != nullptr;
			uint8_t *ops = walks ? a.ops + a.ops_off[pin] : nullptr;
			/* the checkpoints of this item have been written by other lanes of this wave: stores done, stale L1 lines dropped */
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			if (AT_WALK_PRIO) __builtin_amdgcn_s_setprio(AT_WALK_PRIO);
			CkPattern<G, K, CK> pat;
			(void)KG2;
			long long st_t0 = 0, st_rep = 0, st_walk = 0;
			int st_rounds = 0, st_fetch = 0, st_iters = 0;
			(void)st_t0; (void)st_rep; (void)st_walk; (void)st_rounds; (void)st_fetch; (void)st_iters;
			if (AT_TP_STATS) st_t0 = (long long)__builtin_amdgcn_s_memtime();
			for (int rounds = 0;; ++rounds) {
				/* has my walk arrived?  local: HOME (:788-791) or a border; global: a border (then the padding loops); fit: row 0 */
				const bool fin = !walks || !ok || ci <= 0 || (!ISFIT && cj <= 0) || (MODE == K_LOCAL && st == 0) || AT_DIAG_NO_WALK;
				if (!fin && (cj <= 0 || cnt >= own_len)) ok = false;         /* (fit: the walk left the matrix; or more ops than the slot holds) */
				bool go = !fin && ok;
				if (rounds > own_len + 8) { if (go) ok = false; go = false; }   /* (every round moves every walk: cannot happen) */
				if (!__any(go)) break;
				/* a walk in U (1) or the jump state (0, fit -s only) runs left along its row; in M or L it climbs */
				pat.set(go ? ci : 1, go ? cj : 1, st == 1 || (HASJ && st == 0));
				/* the anchors of my group's two alignments (their walkers are lanes 2 * grp and 2 * grp + 1) -> my two blocks */
				int blk_bl[2], blk_c[2];
#pragma unroll
				for (int hh = 0; hh < 2; ++hh) {
					CkPattern<G, K, CK> q;
					q.b = __shfl(pat.b, 2 * grp + hh); q.c0 = __shfl(pat.c0, 2 * grp + hh);
					q.thi1 = __shfl(pat.thi1, 2 * grp + hh); q.horiz = __shfl(pat.horiz, 2 * grp + hh);
					const int alive = __shfl(go ? 1 : 0, 2 * grp + hh);
					bool v;
					q.block(lg, ck_T, blk_bl[hh], blk_c[hh], v);
					if (!v || !alive) { blk_bl[hh] = 0; blk_c[hh] = 0; }   /* (a slot nobody reads: the first block of band 0) */
				}
				long long st_a = 0;
				if (AT_TP_STATS) { st_a = (long long)__builtin_amdgcn_s_memtime(); ++st_rounds; }
				replay16_block<MODE, G, K, TS, SMALL, BITS, CK>(a, mem, gs, refoff, qA, qB, l1A, l1B, blk_bl[0], blk_c[0], blk_bl[1], blk_c[1]);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        /* the blocks' pointer words are in the L2 */
				if (AT_TP_STATS) st_rep += (long long)__builtin_amdgcn_s_memtime() - st_a;
				/* (the replay's staging area in LDS is free again: every walker copies the block it is in -- K x CK cells, KQ * CK words [+ the
				 * jump plane's] -- from the slot into its own piece of it with one round trip, and walks it from there) */
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      /* drop this CU's L1 lines of the replay region: last round's words */
				if (AT_TP_STATS) st_a = (long long)__builtin_amdgcn_s_memtime();
				{
					constexpr int KQ = (K + 3) / 4, GQ = (KG2 + 3) / 4, S4N = CK / 4;
					constexpr int WSCR = ck_walk_words<MODE, K, CK>();
					constexpr int NBLK = ck_walk_blocks<G>();                             /* blocks a walker holds in LDS at a time */
					constexpr int NCHP = S4N * KQ, NCH = NCHP + (HASJ ? S4N * GQ : 0);   /* 16-byte chunks of a block: pointer words, then the plane's */
					constexpr int LPW = 64 / (2 * NG), CPL = (NBLK * NCH + LPW - 1) / LPW;   /* lanes that copy for one walker, chunks per lane */
					const int sl0 = g2 * G;                               /* lane 0 of my alignment's group: slot q = lane sl0 + q */
					const int scr0 = a.off_bound + lane * (NBLK * WSCR);
					bool wgo = go;
					(void)GQ;
					/* phases: every walker names the block it is in (and, where LDS has room, the blocks up and left of it); the wave copies
					 * them from the slot into LDS together (LPW lanes per walker, one round trip for all of them); every walker walks until
					 * it leaves what it holds.  A walker leaves the phases when it has arrived or stands in a block this round has not
					 * replayed */
					for (;;) {
						int need[NBLK], tbl[NBLK], tc[NBLK];
#pragma unroll
						for (int k = 0; k < NBLK; ++k) { need[k] = -1; tbl[k] = -1; tc[k] = -1; }
						if (wgo) {
							if (ci <= 0 || (!ISFIT && cj <= 0) || (MODE == K_LOCAL && st == 0)) wgo = false;
							else if (cj <= 0 || cnt >= own_len || (st == 0 && !HASJ)) { ok = false; wgo = false; }   /* (st = 0 without a jump state: a corrupt pointer) */
							else {
								const int bl = (ci - 1) / K, rr = (ci - 1) - bl * K;
								const int c = ((cj - 1) + bl) >> LCB;
								const int q = pat.slot(bl, c);
								if (q < 0) wgo = false;                       /* (outside this round's blocks: the next round starts here) */
								else {
									need[0] = sl0 + q; tbl[0] = bl; tc[0] = c;
									if constexpr (NBLK == 4) {
										/* the block left of mine, and the two the diagonal through my cell meets in the band above */
										const int c2 = ((cj - 1) - (rr + 1) + (bl - 1)) >> LCB;
										const int q1 = pat.slot(bl, c - 1), q2 = bl > 0 ? pat.slot(bl - 1, c2) : -1, q3 = bl > 0 ? pat.slot(bl - 1, c2 - 1) : -1;
										if (q1 >= 0) { need[1] = sl0 + q1; tbl[1] = bl; tc[1] = c - 1; }
										if (q2 >= 0) { need[2] = sl0 + q2; tbl[2] = bl - 1; tc[2] = c2; }
										if (q3 >= 0) { need[3] = sl0 + q3; tbl[3] = bl - 1; tc[3] = c2 - 1; }
									}
								}
							}
						}
						if (!__any(need[0] >= 0)) break;
						long long st_f0 = 0;
						if (AT_TP_STATS) st_f0 = (long long)__builtin_amdgcn_s_memtime();
						{
							const int w = lane / LPW, part = lane % LPW;
							const int dst = a.off_bound + w * (NBLK * WSCR);
							int src[NBLK];
#pragma unroll
							for (int k = 0; k < NBLK; ++k) src[k] = __shfl(need[k], w);
							uint4 tw[CPL];
#pragma unroll
							for (int x = 0; x < CPL; ++x) {
								const int id = part + x * LPW, k = id / NCH, ch = id - k * NCH;   /* chunk ch of block k: words [ch][slot lane][4] of its region */
								const int sl = imax(NBLK == 1 ? src[0] : pick<NBLK>(src, imin(k, NBLK - 1)), 0);
								const uint32_t *from = gs + (ch < NCHP ? a.off_rptr + (ch * 64 + sl) * 4 : a.off_rjpl + ((ch - NCHP) * 64 + sl) * 4);
								asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(tw[x]) : "v"(from) : "memory");
							}
							asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
							for (int x = 0; x < CPL; ++x) {
								const int id = part + x * LPW, k = id / NCH, ch = id - k * NCH;
								const int sl = NBLK == 1 ? src[0] : pick<NBLK>(src, imin(k, NBLK - 1));
								if (sl >= 0 && k < NBLK) *(uint4 *)(&at_lds[dst + k * WSCR + 4 * ch]) = tw[x];
							}
							__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
						}
						if (AT_TP_STATS) st_fetch += (int)((long long)__builtin_amdgcn_s_memtime() - st_f0);
						if (need[0] >= 0) {
							int kb = 0;                                       /* the block of mine I am in */
							for (;;) {
								const int bl = NBLK == 1 ? tbl[0] : pick<NBLK>(tbl, kb), c = NBLK == 1 ? tc[0] : pick<NBLK>(tc, kb);
								const int scr = scr0 + kb * WSCR, scj = scr + KQ * CK;
								(void)scj;
								const int i_lo = bl * K, t_lo = imax(c << LCB, bl) - bl;   /* row (0-based) and column - 1 of the block's first cell */
								for (;;) {
									/* inside the block both offsets are >= 0.  st = 0: HOME, or nothing at all (the phase loop sorts it out), or the jump state */
									const int rr = ci - 1 - i_lo, ss = cj - 1 - t_lo;
									if ((rr | ss) < 0 || cnt >= own_len || (st == 0 && !HASJ)) break;
									if constexpr (HASJ) {
										if (st == 0) {
											/* jump state (:579-583): left along the row until the column where J opened from M.  A plane word holds my row's
											 * bits of 4 steps, {step 1, step 3, step 0, step 2}; up to 4 words -- 16 columns -- put in column order, the steps
											 * behind mine shifted out: the first set bit is where J opened */
											constexpr unsigned long long ORD = 0xfbea7362d9c85140ull;
											const int s4 = ss >> 2, tk = ss & 3;
											uint32_t cols = 0;
#pragma unroll
											for (int x = 0; x < 4; ++x) {
												const uint32_t w = at_lds[scj + imax(s4 - x, 0) * (GQ * 4) + (rr >> 2)];
												const uint32_t nib = (w >> (16 * h + cell_shift<4>(rr & 3))) & 15u;
												cols = (cols << 4) | ((uint32_t)(ORD >> (4 * nib)) & 15u);
											}
											cols = (cols << (3 - tk)) & 0xffffu;                         /* bit 15 = column cj */
											const int avail = 4 * imin(4, s4 + 1) - (3 - tk);          /* columns of this block from mine leftwards */
											const int lim = imin(imin(avail, cj), own_len - cnt);
											const int n = __clz((int)((cols << 16) | 0x8000u));
											const int steps = n < lim ? n + 1 : lim;
											if (n < lim) st = 2;
											for (int x = 0; x < steps; ++x) ops[cnt + x] = 3;
											cnt += steps; cj -= steps;
											continue;
										}
									}
									/* while its state does not change a walk keeps its direction (LOW up, MID diagonal, UPP left) and its op: the cells of
									 * the next four ops along it are read together; n of them are consumed -- up to the first one that changes the state,
									 * the block's edge, the end of the ops slot */
									if (AT_TP_STATS) ++st_iters;
									/* (the state machine without compares: written with selects hipcc makes branches of it and puts a wait between the four
									 * reads; written with compares every result crosses from the vector unit to a scalar mask and back.  st is 1, 2 or 3 here) */
									const uint32_t ust = (uint32_t)st;
									const uint32_t inL = ust & (ust >> 1), inM = (ust >> 1) & ~ust & 1u;   /* st == 3, st == 2 */
									const int di = (int)(ust >> 1), dj = (int)(inL ^ 1u);
									const uint32_t mL = 0u - inL, mM = 0u - inM, mU = ~(mL | mM);
									uint32_t nbv[4], nst[4];
									int shv[4];
#pragma unroll
									for (int x = 0; x < 4; ++x) {
										const int rx = imax(rr - x * di, 0), sx = imax(ss - x * dj, 0);
										nbv[x] = at_lds[scr + (sx >> 2) * (KQ * 4) + rx];
										shv[x] = 16 * h + ((sx & 1) << 3) + ((sx & 2) << 1);
									}
#pragma unroll
									for (int x = 0; x < 4; ++x) {
										/* {bit 3: U winner, bit 2: L extended, pM[1:0]} -- the cells of the one-pass kernels.  In L: LOW 3 if it extended, else
										 * MID 2; in M: pM; in U: MID 2 if it opened (bit 3 with scores x16, its complement with x4), else UPP 1 */
										const uint32_t nb = nbv[x] >> shv[x];
										const uint32_t lres = 2u + ((nb >> 2) & 1u), mres = nb & 3u, ures = TS == 4 ? 1u + ((nb >> 3) & 1u) : 2u - ((nb >> 3) & 1u);
										nst[x] = (lres & mL) | (mres & mM) | (ures & mU);
									}
									/* e_x = 1 while the state stays what it was: ((a ^ b) - 1) >> 31 is a == b */
									const uint32_t e0 = ((nst[0] ^ ust) - 1u) >> 31, e1 = e0 & (((nst[1] ^ ust) - 1u) >> 31), e2 = e1 & (((nst[2] ^ ust) - 1u) >> 31);
									const int mdi = -di, mdj = -dj;
									const int lim = imin(imin((rr & mdi) | (3 & ~mdi), (ss & mdj) | (3 & ~mdj)), own_len - cnt - 1);   /* ops beyond the first that stay inside */
									const int n = imin(1 + (int)(e0 + e1 + e2), lim + 1);
									const uint32_t op4 = (inL | (2u & mU)) * 0x01010101u;                  /* LOW 1, MID 0, UPP 2 */
									if (__builtin_expect(cnt + 4 <= own_len, 1)) __builtin_memcpy(ops + cnt, &op4, 4);   /* (bytes behind the walk's end are rewritten or never read) */
									else {
#pragma nounroll
										for (int x = 0; x < n; ++x) ops[cnt + x] = (uint8_t)op4;
									}
									st = (int)(((nst[0] | (nst[1] << 4) | (nst[2] << 8) | (nst[3] << 12)) >> (4 * (n - 1))) & 15u);
									ci -= n * di; cj -= n * dj; cnt += n;
								}
								if constexpr (NBLK == 1) break;
								else {
									/* off this block: into another one I hold? */
									if (ci <= 0 || cj <= 0 || cnt >= own_len || st == 0) break;
									const int nbl = (ci - 1) / K, nc = ((cj - 1) + nbl) >> LCB;
									int kn = -1;
#pragma unroll
									for (int k = 0; k < NBLK; ++k) kn = (tbl[k] == nbl && tc[k] == nc) ? k : kn;
									if (kn < 0 || kn == kb) break;
									kb = kn;
								}
							}
						}
					}
				}
				if (AT_TP_STATS) st_walk += (long long)__builtin_amdgcn_s_memtime() - st_a;
			}
			if (AT_WALK_PRIO) __builtin_amdgcn_s_setprio(0);
			if (AT_TP_STATS) {
				const long long now = (long long)__builtin_amdgcn_s_memtime();
				int mxf = st_fetch, mxi = st_iters;
				for (int d = 1; d < 64; d <<= 1) mxi = imax(mxi, __shfl_xor(mxi, d));
				if (lane == 0) {
					atomicAdd(a.queue + 1, 1ull); atomicAdd(a.queue + 2, (unsigned long long)st_rounds);
					atomicAdd(a.queue + 3, (unsigned long long)st_walk); atomicAdd(a.queue + 4, (unsigned long long)mxf);
					atomicAdd(a.queue + 5, (unsigned long long)(now - st_t0)); atomicAdd(a.queue + 6, (unsigned long long)st_rep);
					if (AT_TP_STATS == 2) atomicAdd(a.queue + 7, (unsigned long long)mxi);
					else atomicAdd(a.queue + 7, (unsigned long long)(st_t0 - st_item));
				}
			}
			if (mine) {
				if constexpr (MODE == K_GLOBAL) {                         /* padding loops :398-407 */
					if (walks && ok && !AT_DIAG_NO_WALK) {
						while (cj > 0 && cnt < own_len) { ops[cnt++] = 2; --cj; }
						while (ci > 0 && cnt < own_len) { ops[cnt++] = 1; --ci; }
						if (ci > 0 || cj > 0) ok = false;
					}
				}
				a.score[pin] = ok ? (my_sc >> TS) : INT32_MIN;
				if (a.end_i) a.end_i[pin] = my_ci;
				if (a.end_j) a.end_j[pin] = my_cj;
				if (a.state) a.state[pin] = my_st == 3 ? 1 : my_st == 2 ? 2 : 3;
				if (a.nops) a.nops[pin] = ok ? cnt : -1;
			}
		} else
		/* ================= results; tracebacks: the 2*NG pointer walks run side by side, one per lane, so their
		 *                   dependent pointer loads overlap instead of queueing behind each other ================= */
		{
			const int g = lane >> 1, h = lane & 1;
			const long long pin = (wk * NG + g) * 2 + h;          /* = wk * 2 * NG + lane */
			const bool mine = lane < 2 * NG && pin < a.npairs && !filler;
			const long long p = RAG ? pout : pin;                  /* RAG: the host's order array says which pair this is */
			if (mine) {
				int ci = my_ci, cj = my_cj, st = my_st, cnt = 0;
				bool ok = my_ok;
				if constexpr (TB) {
					uint8_t *ops = a.ops + a.ops_off[p];
					const int glane = g * G;
					int guard = own_len + 2;
					if (AT_WALK_PRIO) __builtin_amdgcn_s_setprio(AT_WALK_PRIO);
					if (AT_DIAG_NO_WALK) {
					} else if (ok && OVL) {
						/* overlap (trace_back_overlap :896-922): the cell's pointer is the move -- LEFT 3: ('-', s2[--j]),
						 * DIAGONAL 2: both, RIGHT 1: (s1[--i], '-') -- until column 0 */
						while (cj > 0 && --guard >= 0) {
							if (ci <= 0) { ok = false; break; }               /* (row 0 is -inf beyond column 0: no path leads there) */
							const int ss = G == 64 ? (ci - 1) / RS : 0, li = G == 64 ? (ci - 1) % RS : ci - 1;
							const int ln = li / K, r = li % K;
							const int t = (cj - 1) + ln;
							const uint32_t w = pm.ld(a.off_ptr + (G == 64 ? __mul24(ss, wps * NL) : 0) + pidx<PTRLDS, K>(t / SPW, r, glane + ln, NL));
							const uint32_t nb = (w >> (16 * h + cell_shift<PB>(t % SPW))) & 3u;
							if (nb == 0) { ok = false; break; }
							ops[cnt++] = (uint8_t)(nb == 3 ? 2 : nb == 2 ? 0 : 1);
							ci -= nb != 3; cj -= nb != 1;
						}
						if (guard < 0) ok = false;
					} else if (ok && (ISFIT || (AT_GLOBAL_WALK_AHEAD && MODE == K_GLOBAL && G < 64) || (AT_LOCAL_WALK_AHEAD && MODE == K_LOCAL && G < 64))) {
						/* fit: the walk crosses the whole read, and with the jump state a run of JUMP ops crosses hundreds of
						 * columns (C4: 380 ops per pair on average), every op a dependent load from HBM.  Runs are predictable:
						 * while the state does not change the walk keeps its direction (LOW up, MID diagonal, UPP / JUMP left).
						 * So the pointer words of the next four cells along the current direction are loaded together and
						 * consumed while the state stays what it was: one round trip to HBM per run of four instead of one per op. */
						constexpr int AHEAD = AT_WALK_AHEAD;
						while (ci > 0 && (ISFIT || cj > 0)) {     /* (global, trace_back_gla :384-397: until either index is 0, then the padding loops) */
							if (MODE == K_LOCAL && st == 0) break;              /* HOME :788-791 (the cell that pointed home has been emitted) */
							if (cj <= 0 || cnt >= own_len) { ok = false; break; }   /* (a walk never has more ops than its slot holds) */
							/* jump state with the bit plane (:579-583): the walk runs left along its row until the column where J opened from M;
							 * the plane holds 4 columns of the row per word, so the same four loads -- issued together with those of the walks in
							 * the other states: one round trip to HBM for all 2 .. 16 walks of the wavefront -- cover up to 16 columns */
							const bool inJ = JPL && st == 0;
							const int di = st >= 2 ? 1 : 0, dj = st == 3 ? 0 : 1;
							/* the cell (ci, cj): strip, lane in group, row in lane -- the cells ahead along the run differ from it by at most
							 * AHEAD - 1 rows, so one division serves all of them */
							const int ss0 = G == 64 ? (ci - 1) / RS : 0, li0 = G == 64 ? (ci - 1) % RS : ci - 1;
							const int ln0 = li0 / K, r0 = li0 % K;
							uint32_t w[AHEAD];
							int sh[AHEAD];
#pragma unroll
							for (int q = 0; q < AHEAD; ++q) {
								int ss = ss0, ln = ln0, r = r0;
								if (q * di > 0 && q * di < ci) {                 /* (cells beyond row 1 are never consumed: any address will do) */
									if constexpr (K >= AHEAD) {
										r = r0 - q * di;
										if (r < 0) { r += K; --ln; }
										if (G == 64 && ln < 0) { ln += 64; --ss; }
									} else {
										const int qi = ci - q * di;
										ss = G == 64 ? (qi - 1) / RS : 0;
										const int li = G == 64 ? (qi - 1) % RS : qi - 1;
										ln = li / K; r = li % K;
									}
								}
								const int t = (imax(cj - q * dj, 1) - 1) + ln;
								int idx = a.off_ptr + (G == 64 ? __mul24(ss, wps * NL) : 0) + pidx<PTRLDS, K>(t / SPW, r, glane + ln, NL);
								sh[q] = 16 * h + cell_shift<PB>(t % SPW);
								if constexpr (JPL) {
									/* (in the jump state every q has the walk's own row and step: word q is the block q to the left) */
									const int jidx = jpl_base0 + (G == 64 ? __mul24(ss0, wpj * NL) : 0) +
									                 pidx<PTRLDS, KG>(imax((((cj - 1) + ln0) >> 2) - q, 0), r0 >> 2, glane + ln0, NL);
									idx = inJ ? jidx : idx;
									sh[q] = inJ ? 16 * h + cell_shift<4>(r0 & 3) : sh[q];
								}
								w[q] = pm.ld(idx);
							}
							if constexpr (JPL) {
								if (inJ) {
									/* the row's 16 plane bits in column order (bit 15 = the newest step of block b0), the steps behind the walk's own
									 * shifted out: the first set bit is the column where J opened -- every column up to it is one JUMP op */
									constexpr unsigned long long ORD = 0xfbea7362d9c85140ull;   /* nibble {k1,k3,k0,k2} -> {k3,k2,k1,k0}, 16 entries of 4 bits */
									uint32_t cols = 0;
#pragma unroll
									for (int q = 0; q < AHEAD; ++q) cols = (cols << 4) | ((uint32_t)(ORD >> (4 * ((w[q] >> sh[q]) & 15u))) & 15u);
									const int tk = ((cj - 1) + ln0) & 3;
									cols = (cols << (16 - 4 * AHEAD + 3 - tk)) & 0xffffu;            /* bit 15 = column cj */
									const int avail = 4 * AHEAD - (3 - tk);
									const int lim = imin(imin(avail, cj), own_len - cnt);
									const int n = __clz((int)((cols << 16) | 0x8000u));
									const int steps = n < lim ? n + 1 : lim;
									if (n < lim) st = 2;
									for (int x = 0; x < steps; ++x) ops[cnt + x] = 3;
									cnt += steps; cj -= steps;
									continue;
								}
							}
							bool go = true;
#pragma unroll
							for (int q = 0; q < AHEAD; ++q) {
								if (go) {
									const uint32_t nb = (w[q] >> sh[q]) & ((1u << PB) - 1u);
									const int was = st;
									int op = 0;
									if (st == 3) { st = (nb & 4u) ? 3 : 2; op = 1; --ci; }
									else if (st == 2) { st = (int)(nb & 3u); op = 0; --ci; --cj; }
									else if (st == 1) { st = (TS == 4 ? (nb & 8u) != 0 : (nb & 8u) == 0) ? 2 : 1; op = 2; --cj; }
									else if (HASJ && !JPL) { st = (nb & 16u) ? 2 : 0; op = 3; --cj; }   /* jump state :579-583 (byte cells) */
									else { ok = false; go = false; }
									if (ok) {
										ops[cnt++] = (uint8_t)op;
										go = st == was && ci > 0 && cj > 0 && cnt < own_len;   /* the next prefetched cell is the next cell */
									}
								}
							}
							if (!ok) break;
						}
						if constexpr (MODE == K_GLOBAL) {                     /* padding loops :398-407 */
							if (ok) {
								while (cj > 0) { ops[cnt++] = 2; --cj; }
								while (ci > 0) { ops[cnt++] = 1; --ci; }
							}
						}
					} else if (ok) {
						/* local / global: one pointer cell per op.  The state machine is written with selects, not branches: the
						 * walk is the only part of a work item in which 2 .. 16 lanes keep a whole wavefront busy, so its length
						 * in instructions counts (C3: a sixth of all vector instructions) */
						const int wpsNL = wps * NL;
						while (ci > 0 && cj > 0 && --guard >= 0) {
							if (MODE == K_LOCAL && st == 0) break;            /* HOME :788-791 */
							const int ss = G == 64 ? (ci - 1) / RS : 0, li = G == 64 ? (ci - 1) % RS : ci - 1;
							const int ln = li / K, r = li % K;
							const int t = (cj - 1) + ln;
							const uint32_t w = pm.ld(a.off_ptr + (G == 64 ? __mul24(ss, wpsNL) : 0) + pidx<PTRLDS, K>(t / SPW, r, glane + ln, NL));
							/* {bit 3: U winner, bit 2: L extended, pM[1:0]}.  Bit 3 is bit 3 of the U winner's tag for TS = 4 (M's 10
							 * has it: opened) and bit 0 for TS = 2 (U's 1 has it: extended). */
							const uint32_t nb = (w >> (16 * h + cell_shift<PB>(t % SPW))) & ((1u << PB) - 1u);
							if (st == 0) { ok = false; break; }               /* (no jump state here: a corrupt pointer) */
							const bool u_open = TS == 4 ? (nb & 8u) != 0 : (nb & 8u) == 0;
							const int inL = st == 3, inM = st == 2;
							const int op = inL ? 1 : inM ? 0 : 2;
							st = inL ? ((nb & 4u) ? 3 : 2) : inM ? (int)(nb & 3u) : (u_open ? 2 : 1);
							ci -= inL | inM; cj -= inL ^ 1;
							ops[cnt++] = (uint8_t)op;
						}
						if constexpr (MODE == K_GLOBAL) {                     /* padding loops :398-407 */
							while (cj > 0) { ops[cnt++] = 2; --cj; }
							while (ci > 0) { ops[cnt++] = 1; --ci; }
						}
						if (guard < 0) ok = false;
					}
				}
				a.score[p] = ok ? (my_sc >> TS) : INT32_MIN;
				if (a.end_i) a.end_i[p] = my_ci;
				if (a.end_j) a.end_j[p] = my_cj;
				if (a.state) a.state[p] = my_st == 3 ? 1 : my_st == 2 ? 2 : 3;
				if (a.nops) a.nops[p] = ok ? cnt : -1;
			}
			if (TB && AT_WALK_PRIO) __builtin_amdgcn_s_setprio(0);
		}
		mem.sync();
	}
	return wnext;
}

/* The items that finish a batch of narrow-group items: two groups of 32 lanes (4 alignments), rows per lane for the longest read of
 * the main items' class in one strip.  (One 64-lane group makes the shortest items -- C2's sliver done in 34 us instead of 42 -- but
 * needs 2.3 times the instructions per alignment where 32-lane groups need 1.4 times: a caller who keeps launches in flight, whose
 * next batch would have filled the idle SIMDs anyway, paid 1.5 % for it on C2.) */
constexpr int AT_TAIL_G = 32;
constexpr int at_tail_k(int g, int k)
{
	return (g * k + AT_TAIL_G - 1) / AT_TAIL_G < 2 ? 2 : (g * k + AT_TAIL_G - 1) / AT_TAIL_G;
}

/* The kernel.  `a`: the batch.  Kernels with narrow groups (G <= 16: 8 .. 32 alignments per work item, items of ~100 us) also take
 * `t`: the SLIVER of the batch behind its whole rounds of work items, as items of two 32-lane groups (four alignments, a third as
 * long) that follow the main items in the same work queue -- a launch that has the chip to itself then ends with a fifth of the
 * SIMDs busy for a short item instead of a tenth of them working through one long item more (C2: 6 250 items of 16 pairs on 2 048
 * resident waves = 3.05 rounds).  t.npairs = 0: no sliver.  Other kernels ignore `t`. */
template <int MODE, int G, int K, int TS, bool SMALL, bool PTRLDS, bool TB, bool RAG = false, int BITS = 2, int CK = 0, int SPLIT = 0>
__global__ __launch_bounds__(64, (CK > 0 && G == 64 ? AT_TP_WAVES64 : AT_WAVES16(G, K))) void at_sweep16(const Sweep16Args a, const Sweep16Args t)
{
	if (a.only_if && __builtin_amdgcn_readfirstlane(*a.only_if) != a.only_val) return;
	long long w = sweep16_items<MODE, G, K, TS, SMALL, PTRLDS, TB, RAG, BITS, CK, SPLIT>(a, (long long)blockIdx.x, 0);
	if constexpr (G <= 16 && !RAG && MODE != K_OVERLAP) {
		if (t.npairs > 0) {
			constexpr int NGm = 64 / G;
			const long long nmain = (a.npairs + 2 * NGm - 1) / (2 * NGm);
			sweep16_items<MODE, AT_TAIL_G, at_tail_k(G, K), TS, SMALL, PTRLDS, TB, false, BITS, CK, SPLIT>(t, w, nmain);
		}
	}
}

} /* namespace at */
