/*
 * at_walk16.hip.h -- pass 2 of the two-pass tracebacks as a kernel of its own (gfx950).
 *
 * The forward sweep (at_sweep16.hip.h, CK kernels with Sweep16Args.ck set) leaves, per work item, the row and column
 * checkpoints of its alignments and, per alignment, the end cell the walk starts from.  Here ONE WALKER OWNS ONE 16-BIT
 * HALF OF A LANE: a wavefront carries 128 walks instead of the 2 .. 32 of a sweep's work item, so the instructions of a
 * walk -- a chain of dependent little steps that 2 of 64 lanes used to execute -- are shared by 64 lanes, and only the
 * blocks a walk really crosses are replayed (C3: ~100 per alignment where the in-kernel rounds replay 274; C2: 1.5 where
 * they replay 19).
 *
 * A round, per lane: the block (lane band, CB steps) each of my two walkers stands in; replay16_lane rebuilds both blocks
 * from their checkpoints -- the tagged arithmetic of the one-pass kernels, one block per half, each with its own
 * position, sequences and checkpoints -- and leaves the pointer words in my column of LDS; both walkers walk their block
 * until they leave it.  A lane touches only its own words of LDS: no barrier anywhere.  Pointers depend on nothing but
 * the exact boundary values, so the ops are those of the one-pass kernels (trace_back_gla / _local_affine /
 * _fit_affine_jump, alignment.h:372-412, 558-592, 766-800).
 *
 * The even alignment of a group of the sweep lives in the low halves of its checkpoint words, the odd one in the high
 * halves: half 0 of a lane walks even alignments, half 1 odd ones, and `lohi` joins two unrelated blocks exactly as the
 * sweep packed them.
 *
 * Two forms.  walk16_wave (the 8-lane groups: the default for fit against a long second sequence -- C4 2 200 -> 2 390 - 2 480
 * GCUPS --, AT_TWO_PASS=2 AT_TP_SPLIT=1 elsewhere): every half-lane a walker of its own, a wavefront per 128 alignments
 * (AT_WALK_WAVES_PER_CU: persistent wavefronts that refill finished halves from two counters).  walk16_team_wave (the
 * 64-lane groups, where a walk crosses a hundred blocks; the default there): teams of NT lanes share one pair of
 * alignments, a round replays the NT blocks per alignment the walk is heading for, and the chain of dependent rounds per
 * alignment is a third as long (C3: 2 940 -> 4 360 - 4 530 GCUPS with launches in flight, 2 090 -> 3 290 one at a time,
 * against the one-pass kernels).  DESIGN.md 3.6.1.
 */
#pragma once
#include "at_sweep16.hip.h"

#ifndef AT_WALK_KPRIO
#define AT_WALK_KPRIO 3
#endif

namespace at {

/* LDS words of the kernel: the site mask (fit -s), the staging area of the rows above the blocks ((CB + 1) steps x 64 lanes x 2 words),
 * every lane's pointer words (K rows x CB / 4 words [+ the jump plane's]) as [word][lane] */
template <int K> constexpr bool walk16_no_staging() { return K > 16; }   /* (replay16_lane: phases of 4 steps) */
template <int MODE, int K, int CB>
constexpr int walk16_lane_words() { return K * (CB / 4) + (MODE == K_FITJ ? (K + 3) / 4 * (CB / 4) : 0); }
template <int MODE, int K, int CB>
constexpr int walk16_lds_words(int nsm) { return ((MODE == K_FITJ ? nsm : 0) + 1) / 2 * 2 + (walk16_no_staging<K>() ? 0 : (CB + 1) * 128) + 64 * walk16_lane_words<MODE, K, CB>(); }

/* This lane's two blocks -- (blA, cA) of the alignment in the low halves, (blB, cB) of the one in the high halves; band = lane-in-group
 * of the forward sweep, c = t-block -- swept with tags from their checkpoints in the item's region `gi` (lanes lane0 .. of it), pointer
 * words (4-bit cells [+ the jump plane]) to LDS at [P0 + word * 64 + lane].  A block starts at step org = max(c * CB, band) and runs CB
 * steps; the cell of step t sits at s = t - org.  Steps behind column l2 and rows behind l1 compute cells nobody reads. */
template <int MODE, int K, int TS, int BITS, int CB>
AT_DEV void replay16_lane(const Sweep16Args &a, const uint32_t *giA, const uint32_t *giB, const int lane0A, const int lane0B,
                          const uint32_t *qA, const uint32_t *qB, const uint32_t *rA, const uint32_t *rB,
                          const int blA, const int cA, const int blB, const int cB,
                          const int SM0, const int S0, const int P0, const int PJ0, long long *stt = nullptr)
{
	/* (stt: -DAT_TP_STATS=2 -- cycles until the first loads are staged, of the state's set-up, of the step loop) */
	long long stt0 = 0;
	(void)stt0;
	if (AT_TP_STATS == 2) stt0 = (long long)__builtin_amdgcn_s_memtime();
	constexpr bool HASJ = MODE == K_FITJ;
	static_assert(!HASJ || (TS == 4 && AT_JPLANE), "two-pass jump state: scores x16, 4-bit cells + bit plane");
	static_assert(MODE == K_GLOBAL || MODE == K_LOCAL || MODE == K_FIT || MODE == K_FITJ, "two-pass tracebacks: the affine modes");
	static_assert(CB % 16 == 0, "a block's columns: whole sequence words");
	constexpr int TMASK = (1 << TS) - 1;
	constexpr int TGL = TS == 4 ? 15 : 3, TGM = TS == 4 ? 10 : 2, TGU = 1;
	constexpr int KG = (K + 3) / 4, ES = ck_es<MODE>(), NQ = ck_nq<MODE, K>();
	const int lane = threadIdx.x;
	const int l1 = a.l1, l2 = a.l2;
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
	const uint32_t *const grA = giA + a.off_rck, *const grB = giB + a.off_rck;
	auto entA = [&](int x) { return blA == 0 ? a.ck_brow + (orgA + x) * ES : grA + ck_rck_word<ES>(orgA - 1 + x, lane0A + blA - 1); };
	auto entB = [&](int x) { return blB == 0 ? a.ck_brow + (orgB + x) * ES : grB + ck_rck_word<ES>(orgB - 1 + x, lane0B + blB - 1); };
	/* ---- the row above my two blocks, CB + 1 entries from step org - 1 on, staged in LDS as the lane below sees it: X' = max(L, M,
	 *      U[, J]) with the winner's tag, and L of the row below = max(L + e, M + o).  The entries arrive in PHASES of PH steps: the first
	 *      (the seed and PH steps) is asked for together with everything else the block starts from -- one round trip --, every other one
	 *      a phase ahead of the steps that read it, so that it travels while the phase before it is swept (a wavefront of walkers is a
	 *      chain of dependent round trips beside the sweeps of other launches, whose stores fill the memory pipes). ---- */
	const int stg = S0 + lane * 2;
	constexpr int PH = K > 16 ? 4 : 8;   /* (19 rows per lane: the registers hold a phase of 4 steps) */
	/* phases of 4 steps = the 4 steps of one pass of the step loop: the entries go from the registers they arrive in straight into the
	 * steps that read them, a pass later -- no staging area in LDS (8.7 KB of a wavefront's 28: five wavefronts of walkers per CU become
	 * eight, 2 048 on the chip; C2's 1 564 team wavefronts were two batches of 1 024) */
	constexpr bool NOSTG = walk16_no_staging<K>();
	static_assert(!NOSTG || PH == 4, "entries without staging: a phase per pass of the step loop");
	static_assert(CB % PH == 0 && PH % 4 == 0, "phases of whole pointer words");
	static_assert(ES == 2, "row checkpoint entries: (X', L of the row below), ck_entry");
	typedef uint2 ent_t;
	ent_t ra[PH], rb[PH];
	auto ask = [&](const int x0) {       /* entries x0 .. x0 + PH - 1 */
#pragma unroll
		for (int x = 0; x < PH; ++x) { ra[x] = *(const ent_t *)entA(x0 + x); rb[x] = *(const ent_t *)entB(x0 + x); }
	};
	auto stage1 = [&](const ent_t &va, const ent_t &vb, const int x) {
		*reinterpret_cast<uint2 *>(&at_lds[stg + x * 128]) = make_uint2(lohi(va.x, vb.x), lohi(va.y, vb.y));
	};
	auto stage = [&](const int x0) {
#pragma unroll
		for (int x = 0; x < PH; ++x) stage1(ra[x], rb[x], x0 + x);
	};
	const ent_t seedA = *(const ent_t *)entA(0), seedB = *(const ent_t *)entB(0);
	ask(1);

	uint32_t Mo_l[K], U_l[K], Xl[2][K], J_l[HASJ ? K : 1], qsel[K], acc[K];
	uint32_t jA[HASJ ? KG : 1], jB[HASJ ? KG : 1];
	(void)jA; (void)jB; (void)J_l;
	/* ---- the state at step org - 1 ---- */
	uint32_t Ad, lraw0;
	const int jA0 = orgA - blA, jB0 = orgB - blB;      /* column - 1 of step org */
	/* my blocks' columns of s2: CB bases from base jA0 / jB0 on, as the low bits of a run of words that is shifted along with the steps */
	constexpr int NW2 = BITS == 2 ? CB / 16 : 1;
	uint32_t winA[NW2], winB[NW2];
	uint32_t ta[NW2 + 1], tb[NW2 + 1];
	if constexpr (BITS == 2) {
		const int lw = (l2 - 1) >> 4;
		const int wa = jA0 >> 4, wb = jB0 >> 4;
#pragma unroll
		for (int x = 0; x <= NW2; ++x) { ta[x] = rA[imin(wa + x, lw)]; tb[x] = rB[imin(wb + x, lw)]; }
	}
	{
		/* the column checkpoint's values (M + o of my K rows, then U, then J: untagged) straight into the state arrays */
		const uint32_t *ckA = giA + a.off_cck, *ckB = giB + a.off_cck;
		auto put = [&](auto XC, uint32_t val) {
			constexpr int x = decltype(XC)::value;
			if constexpr (x < K) Mo_l[x] = val;
			else if constexpr (x < 2 * K) U_l[x - K] = val;
			else if constexpr (HASJ && x < 3 * K) J_l[x - 2 * K] = val;
		};
		uint4 cva[NQ], cvb[NQ];
#pragma unroll
		for (int q = 0; q < NQ; ++q) { cva[q] = *(const uint4 *)(ckA + ck_cck_word<NQ>(cA, lane0A + blA, q)); cvb[q] = *(const uint4 *)(ckB + ck_cck_word<NQ>(cB, lane0B + blB, q)); }
		auto chunk = [&](auto QC) {
			constexpr int q = decltype(QC)::value;
			const uint4 va = cva[q], vb = cvb[q];
			put(std::integral_constant<int, 4 * q>{}, lohi(va.x, vb.x)); put(std::integral_constant<int, 4 * q + 1>{}, lohi(va.y, vb.y));
			put(std::integral_constant<int, 4 * q + 2>{}, lohi(va.z, vb.z)); put(std::integral_constant<int, 4 * q + 3>{}, lohi(va.w, vb.w));
		};
		/* (every load of the block's start has been asked for: the seed entry and the first phase, the checkpoint, the sequence words) */
		constexpr int BPW0 = 32 / BITS, LBP0 = BITS == 2 ? 4 : 2, NWQ0 = (K + BPW0 - 2) / BPW0 + 1;
		const int wqa0 = imin(i0A, l1 - 1) >> LBP0, wqb0 = imin(i0B, l1 - 1) >> LBP0, lwq0 = (l1 - 1) >> LBP0;
		uint32_t qwa[NWQ0], qwb[NWQ0];
#pragma unroll
		for (int x = 0; x < NWQ0; ++x) { qwa[x] = qA[imin(wqa0 + x, lwq0)]; qwb[x] = qB[imin(wqb0 + x, lwq0)]; }
		if constexpr (!NOSTG) {
			stage1(seedA, seedB, 0);
			stage(1);
		}
		if (AT_TP_STATS == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long n = (long long)__builtin_amdgcn_s_memtime(); stt[0] += n - stt0; stt0 = n; }
		if constexpr (!NOSTG && CB > PH) ask(1 + PH);
		if constexpr (BITS == 2) {
			const int sha = (jA0 & 15) * 2, shb = (jB0 & 15) * 2;
#pragma unroll
			for (int x = 0; x < NW2; ++x) { winA[x] = __builtin_amdgcn_alignbit(ta[x + 1], ta[x], sha); winB[x] = __builtin_amdgcn_alignbit(tb[x + 1], tb[x], shb); }
		} else { winA[0] = 0; winB[0] = 0; }
		if constexpr (NOSTG) { Ad = lohi(seedA.x, seedB.x); lraw0 = lohi(seedA.y, seedB.y); }
		else {
			const uint2 e0 = *reinterpret_cast<const uint2 *>(&at_lds[stg]);
			Ad = e0.x; lraw0 = e0.y;
		}
		static_for<NQ>(chunk);
		uint32_t lraw = lraw0;                 /* L of my first row in the checkpoint's column */
		/* the query words my bands' rows lie in (a band of K rows spans NWQ sequence words at most), fetched together */
		constexpr int LBP = LBP0, NWQ = NWQ0;
		const int wqa = wqa0, wqb = wqb0;
		/* (blocks that start at their lane's first step, from the border, are a few per walk: the border's values are worked out only when
		 * some lane of the wavefront has one) */
		const bool some_border = __any(mck != 0xffffffffu);
		auto setup = [&](auto BORDER) {
			constexpr bool border = decltype(BORDER)::value;
#pragma unroll
			for (int r = 0; r < K; ++r) {
				const uint32_t vMo = Mo_l[r], vU = U_l[r];
				uint32_t vJ = neg2;
				if constexpr (HASJ) vJ = J_l[r];
				/* from the checkpoint: M + o and U as stored (untagged), L by the chain down the column */
				const uint32_t kMo = vMo | cTagM, kU = vU | cTagU;
				const uint32_t Lc = lraw | cTagL, Mc = psub(vMo, o2) | cTagM;
				uint32_t kX = pmax(pmax(Lc, Mc), kU);
				if constexpr (HASJ) kX = pmax(kX, vJ);
				lraw = pmax(padd(Lc, e2), kMo);
				if constexpr (border) {
					int La, Ma, Ua, Lb, Mb, Ub;
					border16<MODE>(i0A + r + 1, 0, o16, e16, La, Ma, Ua);
					border16<MODE>(i0B + r + 1, 0, o16, e16, Lb, Mb, Ub);
					La = sat16(La); Lb = sat16(Lb);
					const uint32_t bMo = pk2h(sat16((Ma | TGM) + o16), sat16((Mb | TGM) + o16));
					const uint32_t bU = pk2h(Ua | TGU, Ub | TGU);
					const uint32_t bX = pk2h(imax3(La | TGL, Ma | TGM, Ua | TGU), imax3(Lb | TGL, Mb | TGM, Ub | TGU));
					Mo_l[r] = vbfi(mck, kMo, bMo);
					U_l[r] = vbfi(mck, kU, bU);
					Xl[0][r] = vbfi(mck, kX, bX);
					if constexpr (HASJ) J_l[r] = vbfi(mck, vJ, neg2);
				} else {
					Mo_l[r] = kMo; U_l[r] = kU; Xl[0][r] = kX;
					if constexpr (HASJ) J_l[r] = vJ;
				}
				Xl[1][r] = Xl[0][r];
				/* my query bases, per half its own band's rows */
				const int qi = imin(i0A + r, l1 - 1), qj = imin(i0B + r, l1 - 1);
				const uint32_t wa = pick<NWQ>(qwa, (qi >> LBP) - wqa), wb = pick<NWQ>(qwb, (qj >> LBP) - wqb);
				uint32_t ca, cb;
				if constexpr (BITS == 2) { ca = (wa >> ((qi & 15) * 2)) & 3u; cb = (wb >> ((qj & 15) * 2)) & 3u; }
				else { ca = (wa >> ((qi & 3) * 8)) & 0xffu; cb = (wb >> ((qj & 3) * 8)) & 0xffu; }
				qsel[r] = (ca * 0x00000101u + cb * 0x01010000u) | (BITS == 2 ? 0x04000400u : 0u);
				acc[r] = 0;
			}
		};
		if (some_border) setup(std::true_type{});
		else setup(std::false_type{});
	}
	if constexpr (HASJ) {
#pragma unroll
		for (int g = 0; g < KG; ++g) { jA[g] = 0; jB[g] = 0; }
	}

	if (AT_TP_STATS == 2) { const long long n = (long long)__builtin_amdgcn_s_memtime(); stt[1] += n - stt0; stt0 = n; }
	for (int s4 = 0; s4 < CB / 4; ++s4) {
		if constexpr (!NOSTG && CB > PH) {
			/* a phase begins (not the first): its entries have arrived -- stage them, ask for the next phase's */
			if (s4 > 0 && s4 % (PH / 4) == 0) {
				stage(1 + 4 * s4);
				if (4 * s4 + PH < CB) ask(1 + 4 * s4 + PH);
			}
		}
		/* ---- s2 windows of the block's next 4 steps as one byte per base, per half from its own alignment ---- */
		uint32_t wA, wB, smA = 0, smB = 0;
		if constexpr (BITS == 2) {
			const uint32_t va = winA[0], vb = winB[0];
			wA = (va & 3u) | ((va & 0xcu) << 6) | ((va & 0x30u) << 12) | ((va & 0xc0u) << 18);
			wB = (vb & 3u) | ((vb & 0xcu) << 6) | ((vb & 0x30u) << 12) | ((vb & 0xc0u) << 18);
#pragma unroll
			for (int x = 0; x + 1 < NW2; ++x) { winA[x] = __builtin_amdgcn_alignbit(winA[x + 1], winA[x], 8); winB[x] = __builtin_amdgcn_alignbit(winB[x + 1], winB[x], 8); }
			winA[NW2 - 1] >>= 8; winB[NW2 - 1] >>= 8;
		} else {
			const int lw = (l2 - 1) >> 2;
			const int eA = jA0 + 4 * s4, eB = jB0 + 4 * s4;
			const uint32_t a0 = rA[imin(eA >> 2, lw)], a1 = rA[imin((eA >> 2) + 1, lw)], b0 = rB[imin(eB >> 2, lw)], b1 = rB[imin((eB >> 2) + 1, lw)];
			wA = __builtin_amdgcn_alignbit(a1, a0, (eA & 3) * 8);
			wB = __builtin_amdgcn_alignbit(b1, b0, (eB & 3) * 8);
		}
		if constexpr (HASJ) {
			const int eA = jA0 + 4 * s4 + 1 + 64, eB = jB0 + 4 * s4 + 1 + 64;
			smA = __builtin_amdgcn_alignbit(at_lds[SM0 + (eA >> 5) + 1], at_lds[SM0 + (eA >> 5)], eA & 31);
			smB = __builtin_amdgcn_alignbit(at_lds[SM0 + (eB >> 5) + 1], at_lds[SM0 + (eB >> 5)], eB & 31);
		}
		uint2 eup4[4];                     /* the row above in this pass's four columns */
		if constexpr (NOSTG) {
			/* (asked for a pass ago; the next pass's are asked for now -- a pass behind the last one: entries nobody reads, inside the slack
			 * the sweep's layout leaves behind a lane's last step) */
#pragma unroll
			for (int k = 0; k < 4; ++k) eup4[k] = make_uint2(lohi(ra[k].x, rb[k].x), lohi(ra[k].y, rb[k].y));
			if (s4 + 1 < CB / 4) ask(1 + 4 * (s4 + 1));
		} else {
			/* (the four reads together: one LDS latency per four steps instead of one per step) */
#pragma unroll
			for (int k = 0; k < 4; ++k) eup4[k] = *reinterpret_cast<const uint2 *>(&at_lds[stg + (1 + 4 * s4 + k) * 128]);
		}
		auto step = [&](auto KC) {
			constexpr int k = decltype(KC)::value;
			constexpr uint32_t SELK = (uint32_t)k * 0x00000101u + (uint32_t)(4 + k) * 0x01010000u;
			const uint32_t Aup = eup4[k].x, Bup = eup4[k].y;                     /* the row above in this step's column */
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
		/* ---- the pointer words of these 4 steps: [s4][row] of my column of LDS ---- */
#pragma unroll
		for (int r = 0; r < K; ++r) at_lds[P0 + (s4 * K + r) * 64 + lane] = acc[r];
		if constexpr (HASJ) {
#pragma unroll
			for (int g = 0; g < KG; ++g) at_lds[PJ0 + (s4 * KG + g) * 64 + lane] = jA[g] | (jB[g] >> 1);
		}
	}
	if (AT_TP_STATS == 2) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stt[2] += (long long)__builtin_amdgcn_s_memtime() - stt0; }
}

/* One walker inside the block (bl, c) whose pointer words lie in column `src` of LDS (half h of every word): while its state does not
 * change a walk keeps its direction (LOW up, MID diagonal, UPP / JUMP left) and its op: the cells of the next four ops along it are
 * read together; n of them are consumed -- up to the first one that changes the state, the block's edge, the end of the ops slot */
template <int MODE, int K, int TS, int CB>
AT_DEV void walk16_block(const int h, const int src, int &wci, int &wcj, int &wst, int &wcnt, uint8_t *wops, const int own_len,
                         const int bl, const int c, const int P0, const int PJ0)
{
	constexpr bool HASJ = MODE == K_FITJ;
	constexpr int LCB = ck_log2(CB), KG = (K + 3) / 4;
	(void)KG;
	const int i_lo = bl * K, t_lo = imax(c << LCB, bl) - bl;   /* row (0-based) and column - 1 of the block's first cell */
	for (;;) {
		/* inside the block both offsets are >= 0.  st = 0: HOME, or a corrupt pointer (the round loop sorts it out), or the jump state */
		const int rr = wci - 1 - i_lo, ss = wcj - 1 - t_lo;
		if ((rr | ss) < 0 || wcnt >= own_len || (wst == 0 && !HASJ)) break;
		if constexpr (HASJ) {
			if (wst == 0) {
				/* jump state (:579-583): left along the row until the column where J opened from M.  A plane word holds my row's
				 * bits of 4 steps, {step 1, step 3, step 0, step 2}; up to 4 words -- 16 columns -- put in column order, the steps
				 * behind mine shifted out: the first set bit is where J opened */
				constexpr unsigned long long ORD = 0xfbea7362d9c85140ull;
				const int s4 = ss >> 2, tk = ss & 3;
				uint32_t cols = 0;
#pragma unroll
				for (int x = 0; x < 4; ++x) {
					const uint32_t w = at_lds[PJ0 + (imax(s4 - x, 0) * KG + (rr >> 2)) * 64 + src];
					const uint32_t nib = (w >> (16 * h + cell_shift<4>(rr & 3))) & 15u;
					cols = (cols << 4) | ((uint32_t)(ORD >> (4 * nib)) & 15u);
				}
				cols = (cols << (3 - tk)) & 0xffffu;                         /* bit 15 = column cj */
				const int avail = 4 * imin(4, s4 + 1) - (3 - tk);          /* columns of this block from mine leftwards */
				const int lim = imin(imin(avail, wcj), own_len - wcnt);
				const int n = __clz((int)((cols << 16) | 0x8000u));
				const int steps = n < lim ? n + 1 : lim;
				if (n < lim) wst = 2;
				for (int x = 0; x < steps; ++x) wops[wcnt + x] = 3;
				wcnt += steps; wcj -= steps;
				continue;
			}
		}
		/* (the state machine without compares: written with selects hipcc makes branches of it and puts a wait between the four
		 * reads.  st is 1, 2 or 3 here) */
		const uint32_t ust = (uint32_t)wst;
		const uint32_t inL = ust & (ust >> 1), inM = (ust >> 1) & ~ust & 1u;   /* st == 3, st == 2 */
		const int di = (int)(ust >> 1), dj = (int)(inL ^ 1u);
		const uint32_t mL = 0u - inL, mM = 0u - inM, mU = ~(mL | mM);
		uint32_t nbv[4], nst[4];
		int shv[4];
#pragma unroll
		for (int x = 0; x < 4; ++x) {
			const int rx = imax(rr - x * di, 0), sx = imax(ss - x * dj, 0);
			nbv[x] = at_lds[P0 + ((sx >> 2) * K + rx) * 64 + src];
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
		const int lim = imin(imin((rr & mdi) | (3 & ~mdi), (ss & mdj) | (3 & ~mdj)), own_len - wcnt - 1);   /* ops beyond the first that stay inside */
		const int n = imin(1 + (int)(e0 + e1 + e2), lim + 1);
		const uint32_t op4 = (inL | (2u & mU)) * 0x01010101u;                  /* LOW 1, MID 0, UPP 2 */
		if (__builtin_expect(wcnt + 4 <= own_len, 1)) __builtin_memcpy(wops + wcnt, &op4, 4);   /* (bytes behind the walk's end are rewritten or never read) */
		else {
#pragma nounroll
			for (int x = 0; x < n; ++x) wops[wcnt + x] = (uint8_t)op4;
		}
		wst = (int)(((nst[0] | (nst[1] << 4) | (nst[2] << 8) | (nst[3] << 12)) >> (4 * (n - 1))) & 15u);
		wci -= n * di; wcj -= n * dj; wcnt += n;
	}
}

/* A wavefront of walkers.  Half h of lane l starts with alignment 2 (wave * 64 + l) + h of the launch and, when its walk has arrived,
 * takes the next alignment OF ITS PARITY from the launch's counters (`counter` != nullptr: [0] the even alignments, [1] the odd ones,
 * from pair number `handed` on): the even alignment of a group of the sweep lives in the low halves of its checkpoint words, the odd one
 * in the high halves, and a walker reads its alignment's words where they lie.  (A walk of unrelated reads crosses a few blocks, the
 * longest of 128 of them a dozen: a wavefront that kept its 128 walkers to the end ran 8.6 rounds on C2, most of them for a handful.) */
template <int MODE, int G, int K, int TS, int BITS, int CB>
AT_DEV void walk16_wave(const Sweep16Args &a, const long long wave, unsigned long long *counter, const long long handed)
{
	constexpr bool HASJ = MODE == K_FITJ;
	constexpr bool ISFIT = MODE == K_FIT || MODE == K_FITJ;
	constexpr int NG = 64 / G, LCB = ck_log2(CB), KG = (K + 3) / 4, S4N = CB / 4;
	static_assert(CB == ck_steps(G), "the block the forward sweep of this group width checkpoints");
	const int lane = threadIdx.x;
	const int nsm = HASJ ? a.nsm : 0;
	const int SM0 = 0, S0 = (nsm + 1) / 2 * 2, P0 = S0 + (walk16_no_staging<K>() ? 0 : (CB + 1) * 128), PJ0 = P0 + K * S4N * 64;
	(void)KG;
	long long stt[3] = {0, 0, 0};
	long long st_t0 = 0, st_t1 = 0, st_rep = 0, st_walk = 0;   /* (-DAT_TP_STATS=1: cycles of this wavefront's start-up, replays and walks) */
	(void)st_t0; (void)st_t1; (void)st_rep; (void)st_walk;
	if (AT_TP_STATS) st_t0 = (long long)__builtin_amdgcn_s_memtime();
	if constexpr (HASJ) {
		for (int w = lane; w < nsm; w += 64) at_lds[SM0 + w] = a.sitemask[w];
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	}
	const long long nhalf[2] = {(a.npairs + 1) / 2, a.npairs / 2};   /* even / odd alignments of the launch */
	const int own_len = a.l1 + a.l2;
	/* my two walkers */
	long long pp[2] = {0, 0};
	bool have[2] = {false, false}, ok[2] = {false, false}, ok0[2] = {false, false};
	const uint32_t *gi[2] = {a.ck, a.ck}, *qq[2] = {a.seq, a.seq}, *rr2[2] = {a.seq, a.seq};
	int lane0[2] = {0, 0}, myrounds[2] = {0, 0};
	int ci[2] = {0, 0}, cj[2] = {0, 0}, st[2] = {2, 2}, cnt[2] = {0, 0};
	uint8_t *ops[2] = {a.ops, a.ops};
	auto take = [&](const int h, const long long x) {      /* walker h takes pair number x's alignment of its parity */
		const long long p = 2 * x + h;
		pp[h] = p; have[h] = true; myrounds[h] = 0;
		const long long wk = p / (2 * NG);
		lane0[h] = (int)((p - wk * (2 * NG)) >> 1) * G;      /* lane 0 of my alignment's group in its work item */
		gi[h] = a.ck + wk * a.ck_item_words;
		qq[h] = a.seq + a.woff1[p]; rr2[h] = a.seq + a.woff2[p];
		const int4 e = a.tp_end[p];
		ci[h] = e.x; cj[h] = e.y; st[h] = e.z; ok[h] = e.w != 0; ok0[h] = ok[h]; cnt[h] = 0;
		ops[h] = a.ops + a.ops_off[p];
	};
	/* walk h has arrived: the padding loops of global (:398-407), the results */
	auto retire = [&](const int h) {
		if constexpr (MODE == K_GLOBAL) {
			if (ok[h]) {
				while (cj[h] > 0 && cnt[h] < own_len) { ops[h][cnt[h]++] = 2; --cj[h]; }
				while (ci[h] > 0 && cnt[h] < own_len) { ops[h][cnt[h]++] = 1; --ci[h]; }
				if (ci[h] > 0 || cj[h] > 0) ok[h] = false;
			}
		}
		if (ok0[h] && !ok[h]) a.score[pp[h]] = INT32_MIN;          /* (a walk that failed takes its alignment's score with it, as in the one-pass kernels) */
		a.nops[pp[h]] = ok[h] ? cnt[h] : -1;
		have[h] = false;
	};
	/* has walk h arrived?  local: HOME (:788-791) or a border; global: a border (then the padding loops); fit: row 0 */
	auto going = [&](const int h) -> bool {
		bool fin = !have[h] || !ok[h] || ci[h] <= 0 || (!ISFIT && cj[h] <= 0) || (MODE == K_LOCAL && st[h] == 0);
		/* (fit: the walk left the matrix; more ops than the slot holds; st = 0 without a jump state: a corrupt pointer; every round
		 * moves every walk, so more rounds than ops cannot happen) */
		if (!fin && (cj[h] <= 0 || cnt[h] >= own_len || (st[h] == 0 && !HASJ) || myrounds[h] > own_len + 8)) { ok[h] = false; fin = true; }
		return !fin;
	};
#pragma unroll
	for (int h = 0; h < 2; ++h)
		if (wave * 64 + lane < nhalf[h]) take(h, wave * 64 + lane);
	bool more[2] = {counter != nullptr, counter != nullptr};

	int nrounds = 0;
	if (AT_TP_STATS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st_t1 = (long long)__builtin_amdgcn_s_memtime(); }
	for (;;) {
		bool go[2] = {going(0), going(1)};
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (have[h] && !go[h]) retire(h);
			if (more[h]) {
				/* halves without a walk take the next alignments of their parity: one atomic per wavefront, parity and round */
				const unsigned long long need = __ballot(!have[h]);
				if (need) {
					unsigned long long base = 0;
					if (lane == 0) base = atomicAdd(counter + h, (unsigned long long)__popcll(need));
					const long long b = handed + (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
					                                         (unsigned)__builtin_amdgcn_readfirstlane((int)base));
					const long long x = b + (long long)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
					if (!have[h] && x < nhalf[h]) { take(h, x); go[h] = going(h); }
					if (b + __popcll(need) >= nhalf[h]) more[h] = false;
				}
			}
		}
		if (!__any(go[0] || go[1])) {
			if (__any(have[0] || have[1])) continue;             /* (walks that arrive where they start: retired on the next pass) */
			break;
		}
		int bl[2], c[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			bl[h] = go[h] ? (ci[h] - 1) / K : 0;
			c[h] = go[h] ? ((cj[h] - 1) + bl[h]) >> LCB : 0;
		}
		++myrounds[0]; ++myrounds[1];
		long long st_a = 0, st_b = 0;
		(void)st_a; (void)st_b;
		if (AT_TP_STATS) { st_a = (long long)__builtin_amdgcn_s_memtime(); ++nrounds; }
		replay16_lane<MODE, K, TS, BITS, CB>(a, gi[0], gi[1], lane0[0], lane0[1], qq[0], qq[1], rr2[0], rr2[1], bl[0], c[0], bl[1], c[1], SM0, S0, P0, PJ0, stt);
		if (AT_TP_STATS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_b = (long long)__builtin_amdgcn_s_memtime(); st_rep += st_b - st_a; }
		if (go[0]) walk16_block<MODE, K, TS, CB>(0, lane, ci[0], cj[0], st[0], cnt[0], ops[0], own_len, bl[0], c[0], P0, PJ0);
		if (go[1]) walk16_block<MODE, K, TS, CB>(1, lane, ci[1], cj[1], st[1], cnt[1], ops[1], own_len, bl[1], c[1], P0, PJ0);
		if (AT_TP_STATS) st_walk += (long long)__builtin_amdgcn_s_memtime() - st_b;
	}
	if (AT_TP_STATS && lane == 0) {
		/* the words behind the work counter: wavefronts, rounds, cycles (whole wavefront, start-up, replays, walks), the longest wavefront */
		const long long now = (long long)__builtin_amdgcn_s_memtime();
		atomicAdd(a.queue + 1, 1ull); atomicAdd(a.queue + 2, (unsigned long long)nrounds);
		atomicAdd(a.queue + 3, (unsigned long long)(now - st_t0));
		if (AT_TP_STATS == 2) {   /* the replays apart: first loads + staging, the state's set-up, the step loop; the walks */
			atomicAdd(a.queue + 4, (unsigned long long)stt[0]); atomicAdd(a.queue + 5, (unsigned long long)stt[1]);
			atomicAdd(a.queue + 6, (unsigned long long)stt[2]); atomicAdd(a.queue + 7, (unsigned long long)st_walk);
		} else {
			atomicAdd(a.queue + 4, (unsigned long long)(st_t1 - st_t0));
			atomicAdd(a.queue + 5, (unsigned long long)st_rep); atomicAdd(a.queue + 6, (unsigned long long)st_walk);
			atomicMax(a.queue + 7, (unsigned long long)(now - st_t0));
		}
	}
}

/* A wavefront of walker TEAMS (the 64-lane groups: alignments whose walks cross a hundred blocks).  NT lanes share the two alignments of
 * one work item of the sweep: a round replays NT blocks per alignment -- those its walk is heading for, CkPattern's exact per-band lists
 * seen from the walker's cell, as in the rounds inside the sweep's kernel -- and the walkers (team lane 0: the even alignment's, lane 1:
 * the odd one's) walk from block to block through the columns of LDS their team-mates filled, until they stand in a block the round
 * has not replayed.  A chain of ~100 dependent rounds per alignment becomes one of ~35: C3's walk kernel 3.3 -> ~1 ms. */
template <int MODE, int G, int K, int TS, int BITS, int CB, int NT>
AT_DEV void walk16_team_wave(const Sweep16Args &a, const long long wave)
{
	constexpr bool HASJ = MODE == K_FITJ;
	constexpr bool ISFIT = MODE == K_FIT || MODE == K_FITJ;
	constexpr int NG = 64 / G, LCB = ck_log2(CB), KG = (K + 3) / 4, S4N = CB / 4, TPW = 64 / NT;
	static_assert(NT >= 2 && NT <= 32 && 64 % NT == 0, "team width");
	const int lane = threadIdx.x, tl = lane % NT, t0 = lane - tl;      /* my place in my team, my team's lane 0 */
	const int nsm = HASJ ? a.nsm : 0;
	const int SM0 = 0, S0 = (nsm + 1) / 2 * 2, P0 = S0 + (walk16_no_staging<K>() ? 0 : (CB + 1) * 128), PJ0 = P0 + K * S4N * 64;
	(void)KG; (void)LCB;
	if constexpr (HASJ) {
		for (int w = lane; w < nsm; w += 64) at_lds[SM0 + w] = a.sitemask[w];
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	}
	const int own_len = a.l1 + a.l2;
	constexpr int BLKS = G <= 16 ? 4 : 8;
	const int ck_T = (a.l2 + G - 1 + BLKS - 1) / BLKS * BLKS;          /* steps of the sweep (sweep16_items) */
	/* my team's pair of alignments: 2 u (even, low halves) and 2 u + 1 (odd, high halves) */
	const long long u = wave * TPW + lane / NT;
	const long long npair2 = (a.npairs + 1) / 2;
	const bool team = u < npair2;
	const long long pA = team ? 2 * u : 0, pB = team && 2 * u + 1 < a.npairs ? 2 * u + 1 : pA;
	const long long wk = pA / (2 * NG);
	const int lane0 = (int)((pA - wk * (2 * NG)) >> 1) * G;
	const uint32_t *gi = a.ck + wk * a.ck_item_words;
	const uint32_t *qA = a.seq + a.woff1[pA], *qB = a.seq + a.woff1[pB], *rA = a.seq + a.woff2[pA], *rB = a.seq + a.woff2[pB];
	/* the walker I am, if any: team lane 0 walks the even alignment (h = 0), team lane 1 the odd one (h = 1) */
	const int h = tl & 1;
	const long long pw = h ? 2 * u + 1 : 2 * u;
	const bool walker = team && tl < 2 && pw < a.npairs;
	int ci = 0, cj = 0, st = 2, cnt = 0, myrounds = 0;
	bool ok = false, ok0 = false;
	uint8_t *ops = a.ops;
	if (walker) {
		const int4 e = a.tp_end[pw];
		ci = e.x; cj = e.y; st = e.z; ok = e.w != 0; ok0 = ok;
		ops = a.ops + a.ops_off[pw];
	}
	for (;;) {
		/* has my walk arrived?  local: HOME (:788-791) or a border; global: a border (then the padding loops); fit: row 0 */
		bool fin = !walker || !ok || ci <= 0 || (!ISFIT && cj <= 0) || (MODE == K_LOCAL && st == 0);
		if (!fin && (cj <= 0 || cnt >= own_len || (st == 0 && !HASJ) || myrounds > own_len + 8)) { ok = false; fin = true; }
		const bool go = !fin;
		if (!__any(go)) break;
		++myrounds;
		/* a walk in U (1) or the jump state (0, fit -s only) runs left along its row; in M or L it climbs */
		CkPattern<NT, K, CB> pat;
		pat.set(go ? ci : 1, go ? cj : 1, st == 1 || (HASJ && st == 0));
		/* the anchors of my team's two walkers -> my two blocks */
		int bl[2], c[2];
#pragma unroll
		for (int hh = 0; hh < 2; ++hh) {
			CkPattern<NT, K, CB> q;
			q.b = __shfl(pat.b, t0 + hh); q.c0 = __shfl(pat.c0, t0 + hh);
			q.thi1 = __shfl(pat.thi1, t0 + hh); q.horiz = __shfl(pat.horiz, t0 + hh);
			const int alive = __shfl(go ? 1 : 0, t0 + hh);
			bool v;
			q.block(tl, ck_T, bl[hh], c[hh], v);
			if (!v || !alive) { bl[hh] = 0; c[hh] = 0; }      /* (a slot nobody reads: the first block of band 0) */
		}
		replay16_lane<MODE, K, TS, BITS, CB>(a, gi, gi, lane0, lane0, qA, qB, rA, rB, bl[0], c[0], bl[1], c[1], SM0, S0, P0, PJ0);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      /* my team-mates' columns of LDS */
		if (go) {
			for (;;) {
				if (ci <= 0 || cj <= 0 || cnt >= own_len || (st == 0 && (!HASJ || MODE == K_LOCAL))) break;
				const int wb = (ci - 1) / K, wc = ((cj - 1) + wb) >> LCB;
				const int q = pat.slot(wb, wc);
				if (q < 0) break;                                  /* (outside this round's blocks: the next round starts here) */
				const int before = cnt, bi = ci, bj = cj, bs = st;
				walk16_block<MODE, K, TS, CB>(h, t0 + q, ci, cj, st, cnt, ops, own_len, wb, wc, P0, PJ0);
				if (cnt == before && ci == bi && cj == bj && st == bs) break;   /* (nothing moved: the round loop sorts it out) */
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      /* (the walks have read this round's words before the next replay writes) */
	}
	if (walker) {
		if constexpr (MODE == K_GLOBAL) {                             /* padding loops :398-407 */
			if (ok) {
				while (cj > 0 && cnt < own_len) { ops[cnt++] = 2; --cj; }
				while (ci > 0 && cnt < own_len) { ops[cnt++] = 1; --ci; }
				if (ci > 0 || cj > 0) ok = false;
			}
		}
		if (ok0 && !ok) a.score[pw] = INT32_MIN;
		a.nops[pw] = ok ? cnt : -1;
	}
}

/* The kernel.  `a`: the launch's main work items; `t`: its sliver (the items of two 32-lane groups that end a launch of narrow-group
 * items, at_sweep16): their walks take the first wavefronts of the same launch -- a launch of their own was a second chain of cold
 * instruction fetches and round trips behind the first (C2: 0.2 ms for 2 % of the alignments).  G2 = 0: no sliver. */
template <int MODE, int G, int K, int TS, int BITS, int CB, int G2 = 0, int K2 = 0, int NT = 1>
__global__ __launch_bounds__(64, 2) void at_walk16(const Sweep16Args a, const Sweep16Args t)
{
	if (a.only_if && __builtin_amdgcn_readfirstlane(*a.only_if) != a.only_val) return;
	/* a few wavefronts, each a long chain of dependent steps, beside the sweeps of the launches around them: they go first at the issue */
	if (AT_WALK_KPRIO) __builtin_amdgcn_s_setprio(AT_WALK_KPRIO);
	long long wave = blockIdx.x, nmain = gridDim.x;
	if constexpr (G2 > 0) {
		const long long nt = (t.npairs + 127) / 128;
		if (wave < nt) { walk16_wave<MODE, G2, K2, TS, BITS, ck_steps(G2)>(t, wave, nullptr, 0); return; }
		wave -= nt; nmain -= nt;
	}
	if constexpr (NT > 1) {   /* teams of NT lanes per pair of alignments: one wavefront per 64 / NT pairs */
		walk16_team_wave<MODE, G, K, TS, BITS, CB, NT>(a, wave);
		return;
	}
	/* the launch's main units: 64 per wavefront to begin with, the others from the counter behind the sweep's (zeroed by the host) */
	walk16_wave<MODE, G, K, TS, BITS, CB>(a, wave, a.queue + 8, nmain * 64);   /* (queue[8], [9]: zeroed by the host) */
}

} /* namespace at */
