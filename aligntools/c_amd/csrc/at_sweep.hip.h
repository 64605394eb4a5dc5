/*
 * at_sweep.hip.h -- the anti-diagonal DP sweep + traceback kernel (gfx950).
 *
 * One 64-lane wavefront owns one alignment.  Query rows are cut into strips of
 * 64 rows; lane l of a strip owns row i = 64*s + l + 1 and, at step t, computes
 * column j = t - l + 1, so the wave walks anti-diagonals.  The three (four with
 * the fit jump state) DP values of the cell above arrive from lane l-1 through
 * a DPP wave_shr:1 move (the hardware form of __shfl_up(x, 1)); the left
 * neighbour is the lane's own previous step and lives in registers.  Lane 0
 * takes the row above its strip from a boundary row buffer that lane 63 of the
 * previous strip filled.  Reference sequence s2 is staged 2-bit (or 8-bit)
 * packed and read through a sliding 16-base register window; the lane's query
 * base is one register.  Pointers (4 bit per cell, 8 with the jump state) are
 * accumulated 8 steps per dword and stored time-major: ptr[strip][t/8][lane],
 * i.e. one fully coalesced 256-byte row per 8 anti-diagonals.
 *
 * Storage class `SMALL`: reference window source, boundary row and pointer
 * matrix all live in LDS (one wave per workgroup, no barriers needed -- LDS
 * operations of one wave execute in order).  `!SMALL`: the same three regions
 * live in a per-wave global workspace slot (HBM/L2) for pairs whose pointer
 * matrix does not fit LDS.
 *
 * Arithmetic: the reference computes in fp64 that only ever holds integers or
 * -inf (alignment.h:58-62,483-486).  Here every score is an int32 scaled by 16
 * whose low 4 bits carry a priority tag, so that the reference's first-wins
 * strict-'>' arg-max (max5, alignment.h:90-100) becomes a plain integer max:
 *     L cells carry tag 15, M cells 10, U cells 1, J cells / the local 0: 0
 *   * M(i,j) = max5(L'+s, M'+s, U'+s, 0|J'+s)   alignment.h:451,635,825
 *       -> max3(L',M',U') + s, ties resolved L > M > U > (J | 0)  by the tag;
 *          bits[1:0] of the winner = 3 LOW, 2 MID, 1 UPP, 0 HOME/JUMP.
 *   * L(i,j) = max5(L(i-1,j)+e, M(i-1,j)+o)     alignment.h:456   L first
 *       -> bit 2 of the winner: 1 = LOW, 0 = MID.
 *   * U(i,j) = max5(-, M(i,j-1)+o, U(i,j-1)+e)  alignment.h:460   M first
 *       -> bit 3 of the winner: 1 = MID, 0 = UPP.
 *   * J(i,j) = max5(-, M(i,j-1)+g, -, J(i,j-1)) alignment.h:660   M first
 *       -> bit 3 of the winner: 1 = MID, 0 = JUMP.
 *   -inf is the sentinel -2^26 (scaled -2^30); the host rejects inputs whose
 *   real scores could come within 2^24 of it (AT_ERR_RANGE), so a sentinel
 *   can never beat a real score nor wrap (verified
 *   bit-exact against the fp64 reference).
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace at {

enum KMode { K_GLOBAL = 0, K_LOCAL = 1, K_FIT = 2, K_FITJ = 3, K_OVERLAP = 4, K_EDIT = 5 };

constexpr int kShift = 4;                 /* scores are scaled by 16            */
constexpr int kNeg = -(1 << 30);          /* -inf sentinel, scaled              */
constexpr int kNegThresh = -(1 << 29);    /* anything below is "-inf"           */
constexpr int kTagL = 15, kTagM = 10, kTagU = 1;
constexpr int kPad = 64;                  /* bases of slack in front of s2      */
constexpr int kBlk = 8;                   /* steps per unrolled block           */

struct SweepArgs {
	long long npairs;
	const uint32_t *seq;
	const long long *woff1;
	const int *len1;
	const long long *woff2;
	const int *len2;
	int m16, u16, o16, e16, g16;   /* scores * 16                                  */
	int u_raw;                     /* edit: mismatch cost, unscaled                */
	const uint32_t *sitemask;      /* fit -s: bit (j + 64) set = M->J may open at column j */
	int *score, *end_i, *end_j, *state;
	uint8_t *ops;
	const long long *ops_off;
	int *nops;
	uint32_t *ws;                  /* !SMALL: per-wave workspace slots             */
	long long ws_slot_words;
	int off_bound, off_ptr;        /* word offsets of the regions inside a slot    */
};

extern __shared__ uint32_t at_lds[];

#define AT_DEV __device__ __forceinline__

/* __shfl_up(x, 1) as one DPP move; lane 0 (no source lane) keeps `old`. */
AT_DEV int shfl_up1(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
AT_DEV int imax(int a, int b) { return a > b ? a : b; }
AT_DEV int imin(int a, int b) { return a < b ? a : b; }
AT_DEV int imax3(int a, int b, int c) { return imax(imax(a, b), c); }
AT_DEV int imin3(int a, int b, int c) { return imin(imin(a, b), c); }
AT_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* (a & mask) | (b & ~mask) */
AT_DEV uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

template <bool SMALL>
struct Slot {
	uint32_t *g;
	AT_DEV uint32_t ld(int i) const
	{
		if constexpr (SMALL) return at_lds[i];
		else return g[i];
	}
	AT_DEV void st(int i, uint32_t v) const
	{
		if constexpr (SMALL) at_lds[i] = v;
		else g[i] = v;
	}
	/* make this wave's earlier stores visible to its later loads from other lanes */
	AT_DEV void sync() const
	{
		if constexpr (SMALL) {
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		} else {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		}
	}
};

/* Border cells (i == 0 or j == 0), scaled, untagged.
 * global  alignment.h:428-441 | local: calloc zeros
 * fit     alignment.h:612-624 (row 0 written after column 0, so (0,0) is a row-0 cell) */
template <int MODE>
AT_DEV void border(int i, int j, int o16, int e16, int &L, int &M, int &U, int &J)
{
	J = kNeg;
	if constexpr (MODE == K_GLOBAL) {
		if (i == 0 && j == 0) { L = o16; M = 0; U = o16; }
		else if (j == 0) { L = o16 + e16 * i; M = kNeg; U = kNeg; }
		else { L = kNeg; M = kNeg; U = o16 + e16 * j; }
	} else if constexpr (MODE == K_LOCAL) {
		L = 0; M = 0; U = 0;
	} else {
		if (i == 0) { L = kNeg; M = 0; U = 0; }
		else { L = kNeg; M = kNeg; U = kNeg; }
	}
}

template <int MODE>
AT_DEV int xo_of(int L, int M, int U, int J)
{
	int x = imax3(L | kTagL, M | kTagM, U | kTagU);
	if constexpr (MODE == K_FITJ) x = imax(x, J);
	return x;
}

template <int MODE, int BITS, bool SMALL, bool TB>
__global__ __launch_bounds__(64) void at_sweep(const SweepArgs a)
{
	constexpr bool AFFINE = MODE <= K_FITJ;
	constexpr bool HASJ = MODE == K_FITJ;
	constexpr int PB = HASJ ? 8 : 4;          /* pointer bits per cell        */
	constexpr int SPD = 32 / PB;              /* steps per pointer dword      */
	constexpr int RPB = kBlk / SPD;           /* pointer word rows per block  */
	constexpr int BPW = 32 / BITS;            /* bases per packed word        */
	constexpr int PADW = kPad / BPW;
	constexpr uint32_t BMASK = (1u << BITS) - 1u;

	const int lane = threadIdx.x;
	Slot<SMALL> mem;
	mem.g = SMALL ? nullptr : a.ws + (long long)blockIdx.x * a.ws_slot_words;
	const int m16 = a.m16, u16 = a.u16, o16 = a.o16, e16 = a.e16, g16 = a.g16;

	for (long long p = blockIdx.x; p < a.npairs; p += gridDim.x) {
		const int l1 = uni(a.len1[p]);
		const int l2 = uni(a.len2[p]);
		const uint32_t *q_words = a.seq + a.woff1[p];
		const uint32_t *r_words = a.seq + a.woff2[p];
		const int nstrips = (l1 + 63) >> 6;
		const int tbk = (l2 + 63 + kBlk - 1) / kBlk;   /* blocks per strip         */
		const int wps = tbk * RPB;                     /* pointer word rows / strip */

		/* ---- stage s2 (coalesced int32 reads) in front of kPad slack bases ---- */
		{
			const int nw2 = (l2 + BPW - 1) / BPW;
			for (int w = lane; w < nw2; w += 64) mem.st(PADW + w, r_words[w]);
		}
		/* ---- boundary row 0 ---- */
		for (int j = lane; j <= l2; j += 64) {
			if constexpr (AFFINE) {
				int L, M, U, J;
				border<MODE>(0, j, o16, e16, L, M, U, J);
				mem.st(a.off_bound + 2 * j, (uint32_t)xo_of<MODE>(L, M, U, J));
				mem.st(a.off_bound + 2 * j + 1, (uint32_t)imax((L | kTagL) + e16, (M | kTagM) + o16));
			} else if constexpr (MODE == K_OVERLAP) {
				mem.st(a.off_bound + 2 * j, (uint32_t)((j == 0 ? 0 : kNeg) + o16));   /* :937-938 */
			} else {
				mem.st(a.off_bound + 2 * j, (uint32_t)j);                              /* :302 */
			}
		}
		mem.sync();

		/* running results */
		int best = INT32_MIN, best_tau = 0;        /* local: arg-max over M; fit/overlap: last row M */
		int bestL = kNegThresh, bestL_tau = 0;     /* fit: last row L                                 */
		int endL = kNeg, endM = kNeg, endU = kNeg; /* global: the three states of (l1,l2); edit: D    */
		if constexpr (MODE == K_FIT || MODE == K_FITJ) best = kNegThresh;

		for (int s = 0; s < nstrips; ++s) {
			const int base = s << 6;
			const int i = base + lane + 1;
			const bool rowok = i <= l1;
			const bool lastrow = i == l1;
			/* my query base, replicated across the window width */
			uint32_t qrep;
			{
				const int qi = rowok ? i - 1 : 0;
				const uint32_t qw = q_words[qi / BPW];
				qrep = ((qw >> ((qi % BPW) * BITS)) & BMASK) * (BITS == 2 ? 0x55555555u : 0x01010101u);
			}
			/* column-0 state of my row, and what I hand to the lane below */
			int Mo_l = 0, U_l = 0, Mg_l = 0, J_l = kNeg;   /* affine left state        */
			int P_l = 0;                                   /* overlap / edit left state */
			int A_prev, B_prev = 0, Ad;
			if constexpr (AFFINE) {
				int L, M, U, J;
				border<MODE>(i, 0, o16, e16, L, M, U, J);
				Mo_l = (M | kTagM) + o16;
				Mg_l = (M | kTagM) + g16;
				U_l = U | kTagU;
				J_l = J;
				A_prev = xo_of<MODE>(L, M, U, J);
				border<MODE>(base, 0, o16, e16, L, M, U, J);
				Ad = xo_of<MODE>(L, M, U, J);
			} else if constexpr (MODE == K_OVERLAP) {
				P_l = o16;               /* M(i,0) = 0  -> P = M + o */
				A_prev = o16;
				Ad = o16;
				if (lastrow && l2 >= 1) { best = 0; best_tau = -1; }   /* M(l1,0) = 0 enters the scan :954 */
			} else {
				P_l = i;                 /* D(i,0) = i */
				A_prev = i;
				Ad = base;
			}
			/* lane 0 reads the row above from the boundary buffer, one step ahead */
			int bx = 0, bl = 0;
			if (lane == 0) {
				bx = (int)mem.ld(a.off_bound + 2 * imin(1, l2));
				if constexpr (AFFINE) bl = (int)mem.ld(a.off_bound + 2 * imin(1, l2) + 1);
			}
			uint32_t acc = 0;

			for (int blk = 0; blk < tbk; ++blk) {
				const int t0 = blk * kBlk;
				/* ---- reference window: bases t0-lane .. t0-lane+7 ---- */
				uint32_t xlo, xhi = 0;
				{
					const int e0 = t0 - lane + kPad;
					const int w = e0 / BPW;
					const int sh = (e0 % BPW) * BITS;
					const uint32_t w0 = mem.ld(w), w1 = mem.ld(w + 1);
					xlo = __builtin_amdgcn_alignbit(w1, w0, sh) ^ qrep;
					if constexpr (BITS == 8) {
						const uint32_t w2 = mem.ld(w + 2);
						xhi = __builtin_amdgcn_alignbit(w2, w1, sh) ^ qrep;
					}
				}
				uint32_t sm = 0;
				if constexpr (HASJ) {
					const int e0 = t0 - lane + 1 + 64;
					const uint32_t w0 = a.sitemask[e0 >> 5], w1 = a.sitemask[(e0 >> 5) + 1];
					sm = __builtin_amdgcn_alignbit(w1, w0, e0 & 31);
				}
				const int jm1_0 = t0 - lane;   /* 0-based column of step 0 of this block */

				auto step = [&](auto KC) {
					constexpr int k = decltype(KC)::value;
					const int t = t0 + k;
					/* values of the cell above: lane-1's previous outputs (lane 0: boundary row) */
					const int Aup = shfl_up1(bx, A_prev);
					int Bup = 0;
					if constexpr (AFFINE) Bup = shfl_up1(bl, B_prev);
					/* lane 0 prefetches the boundary entry of the next step */
					if (lane == 0) {
						const int jn = imin(t + 2, l2);
						bx = (int)mem.ld(a.off_bound + 2 * jn);
						if constexpr (AFFINE) bl = (int)mem.ld(a.off_bound + 2 * jn + 1);
					}
					const int jm1 = jm1_0 + k;
					const bool active = rowok && (unsigned)jm1 < (unsigned)l2;
					uint32_t nib = 0;   /* pointer bits of this cell (don't care when inactive) */
					if (active) {
						uint32_t mis;
						if constexpr (BITS == 2) mis = (xlo >> (2 * k)) & 3u;
						else mis = ((k < 4 ? xlo : xhi) >> (8 * (k & 3))) & 0xffu;
						if constexpr (AFFINE) {
							const int s16 = mis ? u16 : m16;
							int Mraw = Ad + s16;
							if constexpr (MODE == K_LOCAL) Mraw = imax(Mraw, 0);
							const int Mc = (Mraw & ~15) | kTagM;
							const int Lraw = Bup;
							const int Lc = Lraw | kTagL;
							const int Uraw = imax(Mo_l, U_l + e16);
							const int Uc = (Uraw & ~15) | kTagU;
							int Jraw = 0, Jc = kNeg;
							if constexpr (HASJ) {
								const bool open_ok = (sm >> k) & 1u;
								Jraw = open_ok ? imax(Mg_l, J_l) : J_l;
								Jc = Jraw & ~15;
							}
							const int Mo = Mc + o16;
							int Xo = imax3(Lc, Mc, Uc);
							if constexpr (HASJ) Xo = imax(Xo, Jc);
							const int Ld = imax(Lc + e16, Mo);
							/* left state for my next column, outputs for the lane below */
							Mo_l = Mo; U_l = Uc;
							if constexpr (HASJ) { Mg_l = Mc + g16; J_l = Jc; }
							A_prev = Xo; B_prev = Ld;
							if constexpr (TB) {
								nib = bfi(3u, (uint32_t)Mraw, (uint32_t)Lraw);
								nib = bfi(7u, nib, (uint32_t)Uraw);
								if constexpr (HASJ) nib = (nib & 15u) | (((uint32_t)Jraw & 8u) << 1);
							}
							if constexpr (MODE == K_LOCAL) {
								if (Mc > best) { best = Mc; best_tau = s * (tbk * kBlk) + t; }   /* :830-833 */
							} else if constexpr (MODE == K_GLOBAL) {
								if (lastrow && jm1 + 1 == l2) { endL = Lc; endM = Mc; endU = Uc; }
							} else {
								/* fit end-cell scan over j = 1..l2-1 of row l1 (:676-690; j = 0 holds -inf) */
								if (lastrow && jm1 + 1 < l2) {
									if (Mc > best) { best = Mc; best_tau = t; }
									if (Lc > bestL) { bestL = Lc; bestL_tau = t; }
								}
							}
							if (lane == 63) {
								mem.st(a.off_bound + 2 * (jm1 + 1), (uint32_t)Xo);
								mem.st(a.off_bound + 2 * (jm1 + 1) + 1, (uint32_t)Ld);
							}
						} else if constexpr (MODE == K_OVERLAP) {
							/* max5(M(i,j-1)+o, M(i-1,j-1)+s, M(i-1,j)+o): LEFT, DIAGONAL, RIGHT  :944 */
							const int sp = mis ? (u16 - o16) : (m16 - o16);
							const int Mraw = imax3(P_l | 3, (Ad + sp) | 2, Aup | 1);
							const int Mc = Mraw & ~15;
							const int P = Mc + o16;
							P_l = P; A_prev = P;
							nib = (uint32_t)Mraw;
							if (lastrow && jm1 + 1 < l2 && Mc > best) { best = Mc; best_tau = t; }
							if (lane == 63) mem.st(a.off_bound + 2 * (jm1 + 1), (uint32_t)P);
						} else {
							/* min3(D(i,j-1)+1, D(i-1,j-1)+cost, D(i-1,j)+1)  :306-309 */
							const int cost = mis ? a.u_raw : 0;
							const int D = imin3(P_l + 1, Ad + cost, Aup + 1);
							P_l = D; A_prev = D;
							if (lastrow && jm1 + 1 == l2) endM = D;
							if (lane == 63) mem.st(a.off_bound + 2 * (jm1 + 1), (uint32_t)D);
						}
					}
					Ad = Aup;
					/* every lane pushes every step so that nibble k of a word is step k */
					if constexpr (TB) acc = __builtin_amdgcn_alignbit(nib, acc, PB);
					if constexpr (TB && SPD < kBlk) {
						if ((k + 1) % SPD == 0)
							mem.st(a.off_ptr + ((s * wps + blk * RPB + k / SPD) << 6) + lane, acc);
					}
				};
				step(std::integral_constant<int, 0>{});
				step(std::integral_constant<int, 1>{});
				step(std::integral_constant<int, 2>{});
				step(std::integral_constant<int, 3>{});
				step(std::integral_constant<int, 4>{});
				step(std::integral_constant<int, 5>{});
				step(std::integral_constant<int, 6>{});
				step(std::integral_constant<int, 7>{});
				if constexpr (TB && SPD == kBlk) mem.st(a.off_ptr + ((s * wps + blk) << 6) + lane, acc);
			}
			mem.sync();
		}

		/* ================= end cell (uniform from here on) ================= */
		int sc16 = 0, ci = 0, cj = 0, st = 2;   /* st: 3 LOW, 2 MID, 1 UPP, 0 JUMP/HOME */
		bool ok = true;
		const int own = (l1 - 1) & 63;          /* lane that owns row l1 */
		if constexpr (MODE == K_LOCAL) {
			/* first cell in row-major order among the maxima: (max M, min i, min j) */
			int bi = 0, bj = 0;
			{
				const int period = tbk * kBlk;
				const int bs = best_tau / period, bt = best_tau % period;
				bi = (bs << 6) + lane + 1;
				bj = bt - lane + 1;
				if (best == INT32_MIN) { bi = INT32_MAX; bj = INT32_MAX; }
			}
			for (int d = 32; d >= 1; d >>= 1) {
				const int ob = __shfl_xor(best, d), oi = __shfl_xor(bi, d), oj = __shfl_xor(bj, d);
				const bool take = ob > best || (ob == best && (oi < bi || (oi == bi && oj < bj)));
				if (take) { best = ob; bi = oi; bj = oj; }
			}
			sc16 = uni(best); ci = uni(bi); cj = uni(bj); st = 2;
			ok = l1 >= 1 && l2 >= 1;
		} else if constexpr (MODE == K_GLOBAL) {
			int eL, eM, eU;
			if (l1 >= 1 && l2 >= 1) {
				eL = __builtin_amdgcn_readlane(endL, own);
				eM = __builtin_amdgcn_readlane(endM, own);
				eU = __builtin_amdgcn_readlane(endU, own);
			} else {
				int L, M, U, J;
				border<MODE>(l1, l2, o16, e16, L, M, U, J);
				eL = L | kTagL; eM = M | kTagM; eU = U | kTagU;
			}
			const int x = imax3(eL, eM, eU);   /* max5(L,M,U) first-wins :466 */
			sc16 = x; st = x & 3; ci = l1; cj = l2;
		} else if constexpr (MODE == K_FIT || MODE == K_FITJ) {
			const int bM = __builtin_amdgcn_readlane(best, own), tM = __builtin_amdgcn_readlane(best_tau, own);
			const int bL = __builtin_amdgcn_readlane(bestL, own), tL = __builtin_amdgcn_readlane(bestL_tau, own);
			ci = l1;
			if ((bL >> kShift) > (bM >> kShift) && bL > kNegThresh) { sc16 = bL; st = 3; cj = tL - own + 1; }   /* L only if strictly greater :684-690 */
			else { sc16 = bM; st = 2; cj = tM - own + 1; }
			ok = sc16 > kNegThresh;
		} else if constexpr (MODE == K_OVERLAP) {
			if (l1 >= 1) {
				const int bM = __builtin_amdgcn_readlane(best, own), tM = __builtin_amdgcn_readlane(best_tau, own);
				sc16 = bM; cj = tM < 0 ? 0 : tM - own + 1;
			} else { sc16 = 0; cj = 0; }   /* row 0: only M(0,0)=0 is finite */
			ci = l1; st = 2;
			ok = l2 >= 1;
		} else {
			int d;
			if (l1 >= 1 && l2 >= 1) d = __builtin_amdgcn_readlane(endM, own);
			else d = l1 + l2;              /* border: D(i,0)=i, D(0,j)=j */
			sc16 = d << kShift; ci = l1; cj = l2;
		}

		/* ================= traceback (uniform pointer walk) ================= */
		int cnt = 0;
		const int ei = ci, ej = cj, est = st;
		if constexpr (TB && MODE != K_EDIT) {
			uint8_t *ops = a.ops + a.ops_off[p];
			uint32_t opreg = 0;
			auto emit = [&](int op) {
				if (lane == (cnt & 63)) opreg = (uint32_t)op;
				if ((cnt & 63) == 63) ops[cnt - 63 + lane] = (uint8_t)opreg;
				++cnt;
			};
			auto fetch = [&](int ii, int jj) -> uint32_t {
				const int ss = (ii - 1) >> 6, ln = (ii - 1) & 63;
				const int t = (jj - 1) + ln;
				const uint32_t w = mem.ld(a.off_ptr + ((ss * wps + t / SPD) << 6) + ln);
				return (uint32_t)uni((int)((w >> ((t % SPD) * PB)) & ((1u << PB) - 1u)));
			};
			int guard = l1 + l2 + 2;
			if (ok) {
				if constexpr (AFFINE) {
					/* trace_back_gla :377-397, _local_affine :771-795, _fit_affine_jump :562-587 */
					while (ci > 0 && (MODE == K_FIT || MODE == K_FITJ || cj > 0) && --guard >= 0) {
						if (MODE == K_LOCAL && st == 0) break;            /* HOME :788-791 */
						if (cj <= 0) { ok = false; break; }              /* reference would index column -1 */
						const uint32_t nb = fetch(ci, cj);
						if (st == 3) { st = (nb & 4u) ? 3 : 2; emit(1); --ci; }
						else if (st == 2) { st = (int)(nb & 3u); emit(0); --ci; --cj; }
						else if (st == 1) { st = (nb & 8u) ? 2 : 1; emit(2); --cj; }
						else { st = (nb & 16u) ? 2 : 0; emit(3); --cj; }
					}
					if constexpr (MODE == K_GLOBAL) {                     /* padding loops :398-407 */
						while (cj > 0) { emit(2); --cj; }
						while (ci > 0) { emit(1); --ci; }
					}
				} else {
					/* trace_back_overlap :899-916 */
					while (cj > 0 && --guard >= 0) {
						if (ci <= 0) { ok = false; break; }
						const uint32_t nb = fetch(ci, cj) & 3u;
						if (nb == 3u) { emit(2); --cj; }
						else if (nb == 2u) { emit(0); --ci; --cj; }
						else if (nb == 1u) { emit(1); --ci; }
						else { ok = false; break; }
					}
				}
				if (guard < 0) ok = false;
			}
			if ((cnt & 63) != 0 && lane < (cnt & 63)) ops[(cnt & ~63) + lane] = (uint8_t)opreg;
		}
		if (lane == 0) {
			a.score[p] = ok ? (sc16 >> kShift) : INT32_MIN;
			if (a.end_i) a.end_i[p] = ei;
			if (a.end_j) a.end_j[p] = ej;
			if (a.state) a.state[p] = est == 3 ? 1 : est == 2 ? 2 : 3;   /* AT_ST_LOW / MID / UPP */
			if (a.nops) a.nops[p] = ok ? cnt : -1;
		}
		mem.sync();   /* the slot is reused by the next pair */
	}
}

} /* namespace at */
