/*
 * at_sweep.hip.h -- the anti-diagonal DP sweep + traceback kernel (gfx950).
 *
 * One 64-lane wavefront owns one alignment.  Query rows are cut into strips of
 * 64*K rows; lane l of a strip owns the K consecutive rows 64*K*s + K*l + 1 ..
 * + K and, at step t, computes their cells in column j = t - l + 1, so the wave
 * walks anti-diagonals.  Inside a lane the K cells of a column are chained
 * through registers (the cell above is the previous iteration of the unrolled
 * row loop); across lanes the values of the cell above arrive from lane l-1
 * through a DPP wave_shr:1 move (the hardware form of __shfl_up(x, 1)); the
 * left neighbours are the lane's own previous step and live in registers.
 * Lane 0 takes the row above its strip from a boundary row buffer that lane 63
 * of the previous strip filled; 8 entries are prefetched per block of 8 steps
 * and handed to lane 0 with a DPP row_shl.  The reference sequence s2 is staged
 * 2-bit (or 8-bit) packed and read through a sliding register window (16 bases
 * per int32); the lane's K query bases live in registers.  Pointers (4 bit per
 * cell, 8 with the fit jump state) are accumulated 8 steps per dword and stored
 * time-major: ptr[strip][t/8][row-in-lane][lane] -- for the HBM storage class
 * one fully coalesced row per 8 anti-diagonals.
 *
 * Storage class `SMALL`: s2 window source, boundary row and pointer matrix all
 * live in LDS (one wave per workgroup, no barriers -- LDS operations of one
 * wave execute in order).  `!SMALL`: the same three regions live in a per-wave
 * global workspace slot (HBM/L2) for pairs whose pointer matrix exceeds LDS.
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
	int off_sm, nsm;               /* fit -s: site mask words staged behind the boundary row */
	int ptr_lanes;                 /* lanes per pointer row (min(64, ceil(max_l1/K))) */
	int max_l1, max_l2;            /* the bounds the caller gave: the LDS / slot regions are sized from them */
	unsigned long long *queue;     /* work counter, zeroed before every launch     */
	/* all-vs-all mode (ap_n > 0): woff1/len1 describe ap_n READS; work item p is the ordered pair (a < b)
	 * with linear triangle index ap_first + p; woff2/len2 are unused */
	long long ap_n, ap_first;
	/* optional processing order (largest pairs first, so that ragged batches end without a long straggler) */
	const int *order;
	/* optional guard: the launch does nothing unless *only_if == only_val (at_align_batch_device queues the packed and
	 * the int32 kernel behind a device-side check of the batch's shapes and lets the check decide) */
	const int *only_if;
	int only_val;
	/* optional: the number of work items lives on the device (a candidate list a kernel before this one has filled); npairs bounds it */
	const int *npairs_dev;
};

extern __shared__ uint32_t at_lds[];

#define AT_DEV __device__ __forceinline__

/* linear index t of the strict upper triangle of an n x n matrix (row-major) -> (a, b), a < b */
AT_DEV void tri_pair(long long t, long long n, long long &ia, long long &ib)
{
	const double d = (double)(2 * n - 1);
	long long r = (long long)((d - sqrt(d * d - 8.0 * (double)t)) * 0.5);
	if (r < 0) r = 0;
	if (r > n - 2) r = n - 2;
	/* pairs before row r: r*(2n - r - 1)/2; correct the floating-point guess */
	while (r > 0 && r * (2 * n - r - 1) / 2 > t) --r;
	while ((r + 1) * (2 * n - r - 2) / 2 <= t) ++r;
	ia = r;
	ib = t - r * (2 * n - r - 1) / 2 + r + 1;
}

/* __shfl_up(x, 1) as one DPP move; lane 0 (no source lane) keeps `old`. */
AT_DEV int shfl_up1(int old, int src)
{
	return __builtin_amdgcn_update_dpp(old, src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
/* lane l reads lane l+N of its row of 16 (only lane 0 <- lane N is used) */
template <int N>
AT_DEV int row_shl(int v)
{
	if constexpr (N == 0) return v;
	else return __builtin_amdgcn_update_dpp(0, v, 0x100 + N /* row_shl:N */, 0xf, 0xf, true);
}
AT_DEV int imax(int a, int b) { return a > b ? a : b; }
AT_DEV int imin(int a, int b) { return a < b ? a : b; }
AT_DEV int imax3(int a, int b, int c) { return imax(imax(a, b), c); }
AT_DEV int imin3(int a, int b, int c) { return imin(imin(a, b), c); }
AT_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* (a & mask) | (b & ~mask) */
AT_DEV uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }
/* v_bfi_b32 / v_and_or_b32 spelled out: hipcc otherwise re-associates nested selects into longer and/or3 chains */
AT_DEV uint32_t vbfi(uint32_t mask, uint32_t a, uint32_t b)
{
	uint32_t d;
	asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "v"(mask), "v"(a), "v"(b));
	return d;
}
AT_DEV uint32_t vandor(uint32_t a, uint32_t m, uint32_t o)
{
	uint32_t d;
	asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "v"(o));
	return d;
}

template <bool SMALL>
struct Slot {
	uint32_t *g;
	AT_DEV uint32_t ld(int i) const
	{
		if constexpr (SMALL) return at_lds[i];
		else return g[i];
	}
	AT_DEV uint2 ld2(int i) const   /* i even: 8-byte aligned */
	{
		if constexpr (SMALL) return *reinterpret_cast<const uint2 *>(&at_lds[i]);
		else return *reinterpret_cast<const uint2 *>(&g[i]);
	}
	AT_DEV void st(int i, uint32_t v) const
	{
		if constexpr (SMALL) at_lds[i] = v;
		else g[i] = v;
	}
	AT_DEV void st2(int i, uint32_t v0, uint32_t v1) const
	{
		if constexpr (SMALL) *reinterpret_cast<uint2 *>(&at_lds[i]) = make_uint2(v0, v1);
		else *reinterpret_cast<uint2 *>(&g[i]) = make_uint2(v0, v1);
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

/* The pointer matrix: LDS, or a per-wave global slot (stores are fire-and-forget coalesced rows;
 * the traceback reads them back L2-served after the wave's stores have been acknowledged). */
#ifndef AT_DIAG_NO_STORE
#define AT_DIAG_NO_STORE 0   /* 1: throw-away build that drops the pointer stores to HBM (what do they cost?); results are no alignments */
#endif
template <bool LDS>
struct PtrStore {
	uint32_t *g;
	AT_DEV void st(int i, uint32_t v) const
	{
		if constexpr (LDS) at_lds[i] = v;
		else if (!AT_DIAG_NO_STORE) g[i] = v;
	}
	AT_DEV uint32_t ld(int i) const
	{
		if constexpr (LDS) return at_lds[i];
		else return __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* sc1: bypasses this CU's L1 */
	}
	/* 16 / 8 bytes of one lane at once (global slot only; i is a multiple of 4 / 2 words) */
	AT_DEV void st4(int i, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) const { if (!AT_DIAG_NO_STORE) *(uint4 *)(g + i) = make_uint4(v0, v1, v2, v3); }
	AT_DEV void st2(int i, uint32_t v0, uint32_t v1) const { if (!AT_DIAG_NO_STORE) *(uint2 *)(g + i) = make_uint2(v0, v1); }
	AT_DEV void ready() const   /* before the first ld of a pair */
	{
		if constexpr (LDS) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
};

/* Dynamic work distribution: one returning atomic per work item, issued one item ahead.  The first item of a
 * wave is its block index (no atomic: thousands of waves hitting one counter in the same microsecond queue up at
 * ~90 dequeues/us), the counter hands out items gridDim.x, gridDim.x + 1, ... */
AT_DEV long long next_work(unsigned long long *queue, int lane)
{
	unsigned long long v = 0;
	if (lane == 0) v = atomicAdd(queue, 1ull) + gridDim.x;
	return (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
	                   (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}

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

template <int MODE, bool TAGS = true>
AT_DEV int xo_of(int L, int M, int U, int J)
{
	int x = TAGS ? imax3(L | kTagL, M | kTagM, U | kTagU) : imax3(L, M, U);
	if constexpr (MODE == K_FITJ) x = imax(x, J);
	return x;
}

/* v[r] for a wave-uniform r, written so that every array index is a compile-time constant: a variable index
 * (even inside a loop that is later unrolled) makes hipcc keep the whole register array in scratch memory. */
template <int K, int Q = 0, typename T>
AT_DEV T pick(const T (&v)[K], int r)
{
	if constexpr (Q == K - 1) return v[Q];
	else return r == Q ? v[Q] : pick<K, Q + 1, T>(v, r);
}

template <int MODE, int BITS, int K, bool SMALL, bool PTRLDS, bool TB>
__global__ __launch_bounds__(64) void at_sweep(const SweepArgs a)
{
	constexpr bool AFFINE = MODE <= K_FITJ;
	constexpr bool HASJ = MODE == K_FITJ;
	constexpr bool KEEPL = MODE == K_GLOBAL || MODE == K_FIT || MODE == K_FITJ;
	constexpr int PB = HASJ ? 8 : 4;          /* pointer bits per cell        */
	constexpr int SPD = 32 / PB;              /* steps per pointer dword      */
	constexpr int RPB = kBlk / SPD;           /* pointer word rows per block  */
	constexpr int BPW = 32 / BITS;            /* bases per packed word        */
	constexpr int PADW = kPad / 4;            /* s2 is staged one byte per base */
	constexpr int RS = 64 * K;                /* rows per strip               */
	constexpr uint32_t BMASK = (1u << BITS) - 1u;
	/* the priority tags decide pointers, never values: kernels without a pointer matrix run without them */
	constexpr int TgL = TB ? kTagL : 0, TgM = TB ? kTagM : 0, TgU = TB ? kTagU : 0;

	const int lane = threadIdx.x;
	Slot<SMALL> mem;
	mem.g = SMALL ? nullptr : a.ws + (long long)blockIdx.x * a.ws_slot_words;
	PtrStore<PTRLDS> pm;
	pm.g = PTRLDS ? nullptr : a.ws + (long long)blockIdx.x * a.ws_slot_words;
	const int m16 = a.m16, u16 = a.u16, o16 = a.o16, e16 = a.e16, g16 = a.g16;
	const int NL = a.ptr_lanes;
	/* Overlap without pointers and edit distance sweep the matrix MINUS ITS GAP RAMP: with M'(i,j) = M(i,j) - o (i + j),
	 *   M'(i,j) = max3(M'(i,j-1), M'(i-1,j-1) + s - 2 o, M'(i-1,j))
	 * -- the two gap candidates lose their additions (every candidate of a cell carries the same offset, so the maximum and its
	 * first-wins order are what they were), a cell is two instructions (v_add_u32_sdwa, v_max3_i32) instead of three; the true value
	 * is restored where it is compared across columns (the scan of row l1) and at the end.  Nothing here needs the x16 room of the
	 * priority tags, so these sweeps run on unscaled scores (s - 2 o must fit the byte table).  Edit distance: D' = D - (i + j),
	 * both borders become 0, D' = min3(D'(i,j-1), D'(i-1,j-1) + cost - 2, D'(i-1,j)). */
	constexpr bool RAMP = (MODE == K_OVERLAP && !TB) || MODE == K_EDIT;
	/* On the ramp a row never lies below the row above it (the RIGHT candidate costs nothing), so the rows a lane holds BEHIND l1 can be
	 * made exact copies of row l1: their query code selects a score of -128 from the table's second word and their column 0 holds
	 * row l1's -- LEFT and DIAGONAL then never beat RIGHT.  The scan of row l1 (:951-959) reads the lane's last row, a fixed register,
	 * instead of picking row (l1 - 1) % K every step (a branch tree of four levels per step: a quarter of the sweep's time on C5). */
	constexpr bool COPYROWS = MODE == K_OVERLAP && !TB && BITS == 2;
	const uint32_t lutneg = COPYROWS ? 0x80808080u : 0u;
	const int o1 = MODE == K_EDIT ? 1 : (o16 >> kShift);   /* the ramp's slope (edit: unit gaps) */
	/* keep the mismatch score in a VGPR the compiler will not re-materialise per step */
	int u16v = MODE == K_EDIT ? a.u_raw - 2 : (MODE == K_OVERLAP ? (TB ? u16 - o16 : (u16 - 2 * o16) >> kShift) : u16);
	asm volatile("" : "+v"(u16v));
	const int m16s = MODE == K_EDIT ? -2 : (MODE == K_OVERLAP ? (TB ? m16 - o16 : (m16 - 2 * o16) >> kShift) : m16);
	/* signed-byte score LUT {match, mismatch x3} for the 2-bit path (the host guarantees both fit a byte) */
	uint32_t lut8 = ((uint32_t)m16s & 0xffu) | (((uint32_t)u16v & 0xffu) * 0x01010100u);
	/* penalties and bit masks in VGPRs: an SGPR operand makes v_add_u32 half-rate on gfx950 (tools/valu_rate.hip) */
	int e16v = e16, o16v = o16, g16v = g16;
	uint32_t cM3 = 3u, cM7 = 7u;
	asm volatile("" : "+v"(lut8), "+v"(e16v), "+v"(o16v), "+v"(g16v), "+v"(cM3), "+v"(cM7));

	if (a.only_if && uni(*a.only_if) != a.only_val) return;
	long long pnext = blockIdx.x;
	const long long np = a.npairs_dev ? (long long)imin((int)(a.npairs < 0x7fffffff ? a.npairs : 0x7fffffff), uni(*a.npairs_dev)) : a.npairs;
	while (pnext < np) {
		const long long p = a.order ? (long long)a.order[pnext] : pnext;
		pnext = next_work(a.queue, lane);   /* consumed at the end of this pair: latency hidden */
		long long ia = p, ib = p;
		if (a.ap_n > 0) tri_pair(a.ap_first + p, a.ap_n, ia, ib);
		const int l1 = uni(a.len1[ia]);
		const int l2 = uni(a.ap_n > 0 ? a.len1[ib] : a.len2[ib]);
		if (l1 > a.max_l1 || l2 > a.max_l2 || l1 < 0 || l2 < 0) {
			/* longer than the bound the regions were sized from: refuse the pair instead of overrunning them */
			if (lane == 0) {
				a.score[p] = INT32_MIN;
				if (a.nops) a.nops[p] = -1;
			}
			continue;
		}
		const uint32_t *q_words = a.seq + a.woff1[ia];
		const uint32_t *r_words = a.seq + (a.ap_n > 0 ? a.woff1[ib] : a.woff2[ib]);
		const int nstrips = (l1 + RS - 1) / RS;
		const int tbk = (l2 + 63 + kBlk - 1) / kBlk;   /* blocks per strip           */
		const int wps = tbk * RPB * K;                 /* pointer word rows per strip */

		/* ---- stage s2 (coalesced int32 reads) as one byte per base in front of kPad slack bytes;
		 *      2-bit input is expanded to byte codes 0..3 so that v_perm_b32 can look scores up ---- */
		{
			const int nw2 = (l2 + BPW - 1) / BPW;
			for (int w = lane; w < nw2; w += 64) {
				const uint32_t v = r_words[w];
				if constexpr (BITS == 8) {
					mem.st(PADW + w, v);
				} else {
#pragma unroll
					for (int q = 0; q < 4; ++q) {
						const uint32_t b = (v >> (8 * q)) & 0xffu;
						mem.st(PADW + 4 * w + q, (b & 3u) | ((b & 0xcu) << 6) | ((b & 0x30u) << 12) | ((b & 0xc0u) << 18));
					}
				}
			}
		}
		if constexpr (HASJ) {
			for (int w = lane; w < a.nsm; w += 64) mem.st(a.off_sm + w, a.sitemask[w]);
		}
		/* ---- boundary row 0 ---- */
		for (int j = lane; j <= l2; j += 64) {
			if constexpr (AFFINE) {
				int L, M, U, J;
				border<MODE>(0, j, o16, e16, L, M, U, J);
				mem.st2(a.off_bound + 2 * j, (uint32_t)xo_of<MODE, TB>(L, M, U, J),
				        (uint32_t)imax((L | TgL) + e16, (M | TgM) + o16));
			} else if constexpr (MODE == K_OVERLAP) {
				mem.st(a.off_bound + 2 * j, (uint32_t)((j == 0 ? 0 : kNeg) + (TB ? o16 : 0)));   /* :937-938 */
			} else {
				mem.st(a.off_bound + 2 * j, 0u);                                       /* D(0,j) = j :302, minus the ramp */
			}
		}
		mem.sync();

		/* the cell (l1, .) lives in strip nstrips-1, lane lastlane, row-in-lane rl */
		const int lastlane = l1 > 0 ? ((l1 - 1) % RS) / K : 0;
		const int rl = l1 > 0 ? ((l1 - 1) % RS) % K : 0;

		/* running results */
		int best = INT32_MIN, best_i = INT32_MAX, best_j = INT32_MAX;   /* local arg-max; fit/overlap: last-row M scan */
		int bestL = kNegThresh, bestL_j = 0;                            /* fit: last-row L scan                        */
		if constexpr (MODE == K_FIT || MODE == K_FITJ) best = kNegThresh;
		/* per-row left state (registers).  Xl = X' of my rows at the previous column, kept in two copies used in turn by
		 * step parity: the old value is read while the new one is written, without register moves.  (A lane is active
		 * for one contiguous run of steps and both copies start with the column-0 value, so inactive steps touch nothing.) */
		int Mo_l[K], U_l[K], Xl[2][K], L_l[K], Mg_l[K], J_l[K];

		for (int s = 0; s < nstrips; ++s) {
			const int base = s * RS;
			const int i0 = base + lane * K;                 /* 0-based index of my first row */
			const int nl = imin(64, (l1 - base + K - 1) / K);   /* lanes holding at least one row */
			const bool laststrip = s == nstrips - 1;
			const bool wb = !laststrip;                     /* lane 63 feeds the next strip */
			uint32_t qrep[K];
			int best_r[K], bt_r[K];
			uint32_t acc[K];
#pragma unroll
			for (int r = 0; r < K; ++r) {
				const int qi = imin(i0 + r, l1 - 1);
				const uint32_t qw = q_words[qi / BPW];
				qrep[r] = ((qw >> ((qi % BPW) * BITS)) & BMASK) * 0x01010101u;
				if constexpr (COPYROWS) qrep[r] = i0 + r >= l1 ? 0x04040404u : qrep[r];   /* selectors 4..7: the -128 word */
				best_r[r] = INT32_MIN; bt_r[r] = 0; acc[r] = 0;
				const int i = i0 + r + 1;
				if constexpr (AFFINE) {
					int L, M, U, J;
					border<MODE>(i, 0, o16, e16, L, M, U, J);
					Mo_l[r] = (M | TgM) + o16;
					Mg_l[r] = (M | TgM) + g16;
					U_l[r] = U | TgU;
					J_l[r] = J;
					L_l[r] = L | TgL;
					Xl[0][r] = xo_of<MODE, TB>(L, M, U, J);
					Xl[1][r] = Xl[0][r];
				} else if constexpr (MODE == K_OVERLAP) {
					Mo_l[r] = TB ? o16 : -o1 * (COPYROWS ? imin(i, l1) : i);   /* M(i,0) = 0 -> P = M + o (with pointers) or M - o i (ramp) */
				} else {
					Mo_l[r] = 0;             /* D(i,0) = i, minus the ramp */
				}
			}
			/* what the lane below sees as "row above, column 0", and my own diagonal */
			int A_prev, B_prev = 0, Ad;
			if constexpr (AFFINE) {
				A_prev = Xl[0][K - 1];
				int L, M, U, J;
				border<MODE>(base, 0, o16, e16, L, M, U, J);
				Ad = xo_of<MODE, TB>(L, M, U, J);
			} else if constexpr (MODE == K_OVERLAP) {
				A_prev = TB ? o16 : -o1 * (i0 + K); Ad = TB ? o16 : -o1 * i0;
			} else {
				A_prev = 0; Ad = 0;
			}
			/* boundary entries of columns t0+1+lane (lanes 0..7 matter), one block ahead */
			auto load_bound = [&](int t0, int &bx, int &bl) {
				const int jn = imin(t0 + 1 + (lane & 7), l2);
				if constexpr (AFFINE) {
					const uint2 v = mem.ld2(a.off_bound + 2 * jn);
					bx = (int)v.x; bl = (int)v.y;
				} else {
					bx = (int)mem.ld(a.off_bound + 2 * jn); bl = 0;
				}
			};
			int bx, bl, bxn, bln;
			load_bound(0, bx, bl);
			const int ptr_base = a.off_ptr + s * wps * NL;

			for (int blk = 0; blk < tbk; ++blk) {
				const int t0 = blk * kBlk;
				load_bound(t0 + kBlk, bxn, bln);
				/* ---- reference window: bases t0-lane .. t0-lane+7 against my K query bases ---- */
				/* 2-bit input: xs[h][r] = four signed score bytes (steps 4h..4h+3 of row r) read from the byte LUT
				 * {m,u,u,u} with the xor of s2 and query codes as selector.  8-bit input: the raw xor (0 = match). */
				uint32_t xs[2][K];
				{
					const int e0 = t0 - lane + kPad;
					const int w = e0 >> 2;
					const int sh = (e0 & 3) * 8;
					const uint32_t w0 = mem.ld(w), w1 = mem.ld(w + 1), w2 = mem.ld(w + 2);
					const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
#pragma unroll
					for (int r = 0; r < K; ++r) {
						if constexpr (BITS == 2) {
							xs[0][r] = __builtin_amdgcn_perm(lutneg, lut8, lo ^ qrep[r]);
							xs[1][r] = __builtin_amdgcn_perm(lutneg, lut8, hi ^ qrep[r]);
						} else {
							xs[0][r] = lo ^ qrep[r]; xs[1][r] = hi ^ qrep[r];
						}
					}
				}
				uint32_t sm = 0;
				if constexpr (HASJ) {
					const int e0 = t0 - lane + 1 + 64;
					const uint32_t w0 = mem.ld(a.off_sm + (e0 >> 5)), w1 = mem.ld(a.off_sm + (e0 >> 5) + 1);
					sm = __builtin_amdgcn_alignbit(w1, w0, e0 & 31);
				}
				const int jm1_0 = t0 - lane;   /* 0-based column of step 0 of this block */

				auto step = [&](auto KC, auto MASKED) {
					constexpr int k = decltype(KC)::value;
					constexpr bool masked = decltype(MASKED)::value;
					const int t = t0 + k;
					/* values of the cell above my first row: lane-1's previous outputs (lane 0: boundary row) */
					const int Aup = shfl_up1(row_shl<k>(bx), A_prev);
					int Bup = 0;
					if constexpr (AFFINE) Bup = shfl_up1(row_shl<k>(bl), B_prev);
					const int jm1 = jm1_0 + k;
					bool active = true;
					if constexpr (masked) active = lane < nl && (unsigned)jm1 < (unsigned)l2;
					uint32_t nib[K];
#pragma unroll
					for (int r = 0; r < K; ++r) nib[r] = 0;
					if (active) {
						if constexpr (MODE == K_FIT || MODE == K_FITJ || MODE == K_OVERLAP) {
							/* end-cell scan of row l1 over columns 0..l2-1 (:676-690, :954-959), one column
							 * behind the sweep: the left state still holds column j-1 = jm1 */
							if (laststrip) {
								/* (ramp: the true M(l1, jm1) = M' + o (l1 + jm1), scaled like every result) */
								const int vM = RAMP ? ((COPYROWS ? Mo_l[K - 1] : pick<K>(Mo_l, rl)) + o1 * (l1 + jm1)) << kShift : pick<K>(Mo_l, rl) - o16;
								if (lane == lastlane && vM > best) { best = vM; best_j = jm1; }
								if constexpr (AFFINE) {
									const int vL = pick<K>(L_l, rl);
									if (lane == lastlane && vL > bestL) { bestL = vL; bestL_j = jm1; }
								}
							}
						}
						int diag = Ad, up = Aup, lraw = Bup;
#pragma unroll
						for (int r = 0; r < K; ++r) {
							int s16;
							if constexpr (BITS == 2) {
								s16 = (int)(signed char)(xs[k >> 2][r] >> (8 * (k & 3)));   /* folds into an SDWA add */
							} else {
								const uint32_t mis = (xs[k >> 2][r] >> (8 * (k & 3))) & 0xffu;
								s16 = mis ? u16v : m16s;
							}
							if constexpr (AFFINE) {
								int Mraw = diag + s16;
								if constexpr (MODE == K_LOCAL) Mraw = imax(Mraw, 0);
								const int Mc = TB ? ((Mraw & ~15) | kTagM) : Mraw;
								const int Lc = TB ? (lraw | kTagL) : lraw;
								const int Uraw = imax(Mo_l[r], U_l[r] + e16v);
								const int Uc = TB ? ((Uraw & ~15) | kTagU) : Uraw;
								int Jraw = 0, Jc = kNeg;
								if constexpr (HASJ) {
									const bool open_ok = (sm >> k) & 1u;
									Jraw = open_ok ? imax(Mg_l[r], J_l[r]) : J_l[r];
									Jc = TB ? (Jraw & ~15) : Jraw;
								}
								const int Mo = Mc + o16v;
								int Xo = imax3(Lc, Mc, Uc);
								if constexpr (HASJ) Xo = imax(Xo, Jc);
								const int Ld = imax(Lc + e16v, Mo);
								if constexpr (TB) {
									nib[r] = vbfi(cM3, (uint32_t)Mraw, (uint32_t)lraw);
									nib[r] = vbfi(cM7, nib[r], (uint32_t)Uraw);
									if constexpr (HASJ) nib[r] = (nib[r] & 15u) | (((uint32_t)Jraw & 8u) << 1);
								}
								if constexpr (MODE == K_LOCAL) {
									if (Mc > best_r[r]) { best_r[r] = Mc; bt_r[r] = t; }   /* :830-833 */
								}
								/* hand down / right */
								diag = Xl[k & 1][r];
								Xl[(k & 1) ^ 1][r] = Xo;
								lraw = Ld;
								up = Xo;
								Mo_l[r] = Mo; U_l[r] = Uc;
								if constexpr (KEEPL) L_l[r] = Lc;
								if constexpr (HASJ) { Mg_l[r] = Mc + g16v; J_l[r] = Jc; }
							} else if constexpr (MODE == K_OVERLAP) {
								/* max5(M(i,j-1)+o, M(i-1,j-1)+s, M(i-1,j)+o): LEFT, DIAGONAL, RIGHT  :944 */
								const int old = Mo_l[r];
								int P;
								if constexpr (TB) {
									const int Mraw = imax3(old | 3, (diag + s16) | 2, up | 1);
									P = (Mraw & ~15) + o16v;
									nib[r] = (uint32_t)Mraw;
								} else {
									/* scores only: the priority tags decide pointers, never values; no gap additions on the ramp */
									P = imax3(old, diag + s16, up);
								}
								Mo_l[r] = P; diag = old; up = P;
							} else {
								/* min3(D(i,j-1)+1, D(i-1,j-1)+cost, D(i-1,j)+1)  :306-309 */
								const int old = Mo_l[r];
								const int D = imin3(old, diag + s16, up);
								Mo_l[r] = D; diag = old; up = D;
							}
						}
						A_prev = up; B_prev = lraw;
						if (wb && lane == 63) {
							if constexpr (AFFINE) mem.st2(a.off_bound + 2 * (jm1 + 1), (uint32_t)up, (uint32_t)lraw);
							else mem.st(a.off_bound + 2 * (jm1 + 1), (uint32_t)up);
						}
					}
					Ad = Aup;
					/* every lane pushes every step so that nibble k of a word is step k */
					if constexpr (TB) {
#pragma unroll
						for (int r = 0; r < K; ++r) acc[r] = __builtin_amdgcn_alignbit(nib[r], acc[r], PB);
						if constexpr (SPD < kBlk) {
							if ((k + 1) % SPD == 0 && lane < NL) {
#pragma unroll
								for (int r = 0; r < K; ++r)
									pm.st(ptr_base + ((blk * RPB + k / SPD) * K + r) * NL + lane, acc[r]);
							}
						}
					}
				};
				using T = std::true_type;
				using F = std::false_type;
#define AT_STEPS(M)                                                                                      \
	step(std::integral_constant<int, 0>{}, M{}); step(std::integral_constant<int, 1>{}, M{});            \
	step(std::integral_constant<int, 2>{}, M{}); step(std::integral_constant<int, 3>{}, M{});            \
	step(std::integral_constant<int, 4>{}, M{}); step(std::integral_constant<int, 5>{}, M{});            \
	step(std::integral_constant<int, 6>{}, M{}); step(std::integral_constant<int, 7>{}, M{});
				/* steady block: every lane that owns rows is inside the matrix for all 8 steps
				 * (lanes >= nl then compute cells nobody reads: nothing of theirs is stored or merged) */
				if (t0 >= nl - 1 && t0 + kBlk <= l2) { AT_STEPS(F) }
				else { AT_STEPS(T) }
#undef AT_STEPS
				if constexpr (TB && SPD == kBlk) {
					if (lane < NL) {
#pragma unroll
						for (int r = 0; r < K; ++r) pm.st(ptr_base + (blk * K + r) * NL + lane, acc[r]);
					}
				}
				bx = bxn; bl = bln;
			}
			if constexpr (MODE == K_LOCAL) {
				/* rows ascend inside a lane; strict '>' keeps the first in row-major order */
#pragma unroll
				for (int r = 0; r < K; ++r) {
					if (i0 + r < l1 && best_r[r] > best) { best = best_r[r]; best_i = i0 + r + 1; best_j = bt_r[r] - lane + 1; }
				}
			}
			mem.sync();
		}

		/* ================= end cell (uniform from here on) ================= */
		int sc16 = 0, ci = 0, cj = 0, st = 2;   /* st: 3 LOW, 2 MID, 1 UPP, 0 JUMP/HOME */
		bool ok = true;
		if constexpr (MODE == K_LOCAL) {
			/* first cell in row-major order among the maxima: (max M, min i, min j) */
			int bi = best_i, bj = best_j;
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
				eL = __builtin_amdgcn_readlane(pick<K>(L_l, rl), lastlane);
				eM = __builtin_amdgcn_readlane(pick<K>(Mo_l, rl), lastlane) - o16;
				eU = __builtin_amdgcn_readlane(pick<K>(U_l, rl), lastlane);
			} else {
				int L, M, U, J;
				border<MODE>(l1, l2, o16, e16, L, M, U, J);
				eL = L | kTagL; eM = M | kTagM; eU = U | kTagU;
			}
			if constexpr (!TB) { eL = (eL & ~15) | kTagL; eM = (eM & ~15) | kTagM; eU = (eU & ~15) | kTagU; }   /* start state */
			const int x = imax3(eL, eM, eU);   /* max5(L,M,U) first-wins :466 */
			sc16 = x; st = x & 3; ci = l1; cj = l2;
		} else if constexpr (MODE == K_FIT || MODE == K_FITJ) {
			const int bM = __builtin_amdgcn_readlane(best, lastlane), jM = __builtin_amdgcn_readlane(best_j, lastlane);
			const int bL = __builtin_amdgcn_readlane(bestL, lastlane), jL = __builtin_amdgcn_readlane(bestL_j, lastlane);
			ci = l1;
			if ((bL >> kShift) > (bM >> kShift) && bL > kNegThresh) { sc16 = bL; st = 3; cj = jL; }   /* L only if strictly greater :684-690 */
			else { sc16 = bM; st = 2; cj = jM; }
			ok = sc16 > kNegThresh;
		} else if constexpr (MODE == K_OVERLAP) {
			if (l1 >= 1) {
				sc16 = __builtin_amdgcn_readlane(best, lastlane);
				cj = __builtin_amdgcn_readlane(best_j, lastlane);
			} else { sc16 = 0; cj = 0; }   /* row 0: only M(0,0)=0 is finite */
			ci = l1; st = 2;
			ok = l2 >= 1;
		} else {
			int d;
			if (l1 >= 1 && l2 >= 1) d = __builtin_amdgcn_readlane(pick<K>(Mo_l, rl), lastlane) + l1 + l2;   /* (the ramp back) */
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
			/* Pointer cells are read through a register cache of one time-major block row: the K*NL words of
			 * (strip, t/SPD) are one coalesced load (each lane keeps its own K words); the walk picks the word of
			 * lane `ln` with v_readlane.  A path moves by at most 2 anti-diagonals per step, so a row serves >= SPD/2
			 * steps, and the row below it (t decreasing) is prefetched while the current one is walked. */
			uint32_t cw[K], cwn[K];
			int ckey = -1, nkey = -1;       /* row index ss*wps/K + t/SPD currently in cw / cwn */
			const int rows_per_strip = wps / K;
			auto load_row = [&](int key, uint32_t (&dst)[K]) {
				const int base = a.off_ptr + key * K * NL;
#pragma unroll
				for (int r = 0; r < K; ++r) dst[r] = lane < NL ? pm.ld(base + r * NL + lane) : 0u;
			};
			auto fetch = [&](int ii, int jj) -> uint32_t {
				const int ss = (ii - 1) / RS, li = (ii - 1) % RS;
				const int ln = li / K, r = li % K;
				const int t = (jj - 1) + ln;
				const int key = ss * rows_per_strip + t / SPD;
				if (key != ckey) {
					if (key == nkey) {
#pragma unroll
						for (int q = 0; q < K; ++q) cw[q] = cwn[q];
					} else {
						load_row(key, cw);
					}
					ckey = key;
					nkey = key - 1;
					if (nkey >= 0) load_row(nkey, cwn);
				}
				const uint32_t mine = pick<K>(cw, r);
				const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)mine, ln);
				return (w >> ((t % SPD) * PB)) & ((1u << PB) - 1u);
			};
			int guard = l1 + l2 + 2;
			pm.ready();
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
