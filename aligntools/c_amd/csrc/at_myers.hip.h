/*
 * at_myers.hip.h -- bit-parallel edit distance for the unit-cost case of
 * edit_dist (reference src/alignment.h:291-315): D(i,0) = i, D(0,j) = j,
 * D(i,j) = min3(D(i,j-1) + 1, D(i-1,j-1) + (s1[i-1] == s2[j-1] ? 0 : u), D(i-1,j) + 1), returns D(l1,l2).
 * With the mismatch cost u == 1 (`edit -u 1`) this is the Levenshtein distance, and the column of vertical
 * differences D(i,j) - D(i-1,j) in {-1,0,+1} fits two bit vectors (Myers 1999, block form of Hyyro 2003):
 * one 32-bit word holds 32 rows, a column step costs 14 instructions per word (below) instead of 32 cells.
 * Any other u keeps the cell-by-cell kernel (at_sweep.hip.h, K_EDIT): a negative u, the reference's
 * default, is not a distance and has no such encoding.
 *
 * Mapping: G lanes per alignment, W consecutive words (32 * W rows) per lane, the words of a column chained through the
 * horizontal difference (the word above's Ph / Mh), every alignment with its own lengths.
 *   G = 1 (round 3): ONE ALIGNMENT PER LANE, 64 per wavefront, W = 2 / 3 / 4 / 5 / 8 / 16 / 32 words for reads of up to 64 / 96 / 128 /
 *          160 / 256 / 512 / 1 024 bases -- no skew, no idle lanes, a column step is W chained word steps inside the lane; the 64 s2
 *          windows of a wavefront in LDS (odd stride), one LDS word per sixteen columns;
 *   G = 32: two alignments per wavefront, l1 <= 1 024 * W with W up to 32 (32 768 bases); G = 8: eight alignments of reads up to 256
 *          bases (the round-2 form, kept for A/B runs and for second sequences too long for 64 LDS windows).  Like in the sweep
 *          kernels the lanes are skewed: lane l works on column t - l at step t and takes the horizontal-difference words of the word
 *          above from lane l - 1's previous step by one DPP move each.
 * The first word's horizontal input is +1 (the border D(0,j) = j).  When a lane has passed its last column its Pv / Mv are the vertical
 * differences of column l2: D(l1, l2) = l2 + popcount(Pv & rows <= l1) - popcount(Mv & rows <= l1), summed over the alignment's words
 * and lanes once.  s2 is staged in LDS as packed 2-bit words.  All-vs-all over one read set: the pairs of the triangle are enumerated
 * here (ap_n / ap_first), like in at_sweep.
 */
#pragma once
#include "at_sweep.hip.h"

namespace at {

struct MyersArgs {
	long long npairs;
	const uint32_t *seq;           /* 2-bit packed words (at_pack_batch, bits = 2) */
	const long long *woff1, *woff2;
	const int *len1, *len2;
	int max_l1, max_l2;            /* bounds: max_l1 <= 32 * G * W, the LDS region is sized from max_l2 */
	int *score, *end_i, *end_j, *state, *nops;
	const int *order;              /* optional processing order (largest pairs first) */
	unsigned long long *queue;
	/* all-vs-all over one read set (woff1 / len1 per READ): pair p of this launch is entry ap_first + p of the strict upper triangle of
	 * ap_n reads, enumerated here like in at_sweep (no per-pair descriptors); ap_n = 0: pairs from the four arrays */
	long long ap_n, ap_first;
	/* SEMI kernels (the overlap filter, below): 2 m, 2 c - m and 2 T of the bound; candidates are appended to cand_order[] */
	int semi_m2, semi_k, semi_min2;
	int *cand_order, *cand_count;
};

/* bit k of the result = bit 2k of `w`, k = 0..15 (one bit plane of sixteen 2-bit codes) */
AT_DEV uint32_t even16(uint32_t w)
{
	uint32_t m = w & 0x55555555u;
	m = (m | (m >> 1)) & 0x33333333u;
	m = (m | (m >> 2)) & 0x0f0f0f0fu;
	m = (m | (m >> 4)) & 0x00ff00ffu;
	m = (m | (m >> 8)) & 0x0000ffffu;
	return m;
}
/* One column step of one word costs 14 instructions (round 2: 27):
 *   - s1 is kept as two bit planes (B0 / B1 = low / high bit of the code of each row) instead of four match masks: Eq = (B0 ^ ~C0) & (B1 ^ ~C1)
 *     with the column's code spread to two all-ones / all-zeros words once per column -- three instructions like the selects before, half
 *     the registers (the 32-word form: 128 instead of 192 + 64);
 *   - the boolean forms are written plainly: gfx950's v_bitop3_b32 takes any function of three words, and hipcc finds them
 *     (Mv | ~(sum | Pv | Eq) in two, Mhs | ~(Xv | Phs) and Phs & (Eq | Mv) in one each);
 *   - the shifted horizontal differences are one v_alignbit each over (this word, the word above): no carry bits are extracted;
 *   - nothing is accumulated per step: when a lane has passed its last column its Pv / Mv ARE the vertical differences of column l2, and
 *     D(l1, l2) = D(0, l2) + sum over rows <= l1 = l2 + popcount(Pv & rows) - popcount(Mv & rows), summed over the words and lanes of
 *     the alignment once, at the end.
 * Rows behind l1 hold code 0 and may match: differences only ever travel towards higher rows (the add's carry, the shifts), so they never
 * reach a row <= l1, and the final count masks them. */
/* SEMI = true: the OVERLAP FILTER (round 4; SURVEY.md 8(f) rank 3, "cuts C5's 1.25e15 cells").  align_overlap (alignment.h:926-964)
 * aligns a suffix of s1 (a rows) with a prefix of s2 (b <= l2 - 1 columns: the scan of row l1, :951-959) under a linear gap.  With x
 * matches, y mismatches and g gap columns, a + b = 2x + 2y + g and the score is m x + u y + o g = m (a + b) / 2 - (m - u) y -
 * (m / 2 - o) g <= m (a + b) / 2 - c (y + g) with c = min(m - u, m / 2 - o); y + g is at least the unit-cost edit distance ED of the
 * two pieces, and a <= b + ED, so
 *       2 score(.., b)  <=  2 m b - (2 c - m) D'(l1, b),      D'(i, j) = min over i0 of ED(s1[i0 .. i), s2[0 .. j))
 * -- the edit distance with a free start in s1, i.e. this kernel with the border D'(i, 0) = 0 instead of i (Pv = 0).  The kernel keeps
 * D'(l1, j) column by column (s1 is loaded so that row l1 is the last bit of the lane's last word; the rows in front of it hold zeros,
 * which can only lower D' -- the bound stays a bound) and reports ub = max over b <= l2 - 1 of the right-hand side, halved.  A pair with
 * ub < T has an overlap score below T and is not swept; the others are appended to cand_order[] and swept exactly.  At 14 instructions
 * per 32 cells the filter runs seven times faster than the two instructions per cell of the exact sweep. */
template <int W, int G, bool SEMI = false>
__global__ __launch_bounds__(64) void at_myers(const MyersArgs a)
{
	static_assert(G == 32 || G == 8 || G == 1, "lanes per alignment");
	static_assert(!SEMI || G == 1, "the overlap filter: one alignment per lane");
	constexpr int NG = 64 / G;                         /* alignments per wavefront */
	const int lane = threadIdx.x;
	const int grp = lane / G, lg = lane % G;
	const int nw2max = (((a.max_l2 + 15) >> 4) + 2) | 1;   /* odd: the windows of the 64 lanes start in different LDS banks */
	uint32_t *ref = at_lds + grp * nw2max;             /* my alignment's s2 words */
	const long long nwork = (a.npairs + NG - 1) / NG;
	long long wnext = blockIdx.x;
	while (wnext < nwork) {
		const long long wk = wnext;
		wnext = next_work(a.queue, lane);
		const long long pin = wk * NG + grp;
		const bool have = pin < a.npairs;
		const long long p = a.order ? (long long)a.order[have ? pin : a.npairs - 1] : (have ? pin : a.npairs - 1);
		long long ia = p, ib = p;
		if (a.ap_n > 0) tri_pair(a.ap_first + p, a.ap_n, ia, ib);
		const int l1 = a.len1[ia], l2 = a.ap_n > 0 ? a.len1[ib] : a.len2[ib];
		const uint32_t *q = a.seq + a.woff1[ia], *r = a.seq + (a.ap_n > 0 ? a.woff1[ib] : a.woff2[ib]);
		const bool fits = l1 >= 0 && l2 >= 0 && l1 <= 32 * G * W && l2 <= a.max_l2;
		/* ---- stage s2 ---- */
		const int nw2 = fits ? (l2 + 15) >> 4 : 0;
		for (int w = lg; w < nw2; w += G) ref[w] = r[w];
		/* ---- my W words of s1 as two bit planes, all-ones vertical state (D(i,0) = i) ---- */
		uint32_t B0[W], B1[W], Pv[W], Mv[W];
		const int nw1 = fits ? (l1 + 15) >> 4 : 0;
#pragma unroll
		for (int w = 0; w < W; ++w) {
			const int b = lg * W + w;                      /* word index: rows 32b+1 .. 32b+32 */
			uint32_t lo, hi;
			if constexpr (SEMI) {
				/* row l1 is the last bit of word W - 1: word w holds s1[32 w - pad ..), zeros in front of s1 */
				const int P = 32 * w - (32 * W - l1), kq = P >> 4, sh = (P & 15) * 2;
				const uint32_t x0 = kq >= 0 && kq < nw1 ? q[kq] : 0u, x1 = kq + 1 >= 0 && kq + 1 < nw1 ? q[kq + 1] : 0u;
				const uint32_t x2 = kq + 2 >= 0 && kq + 2 < nw1 ? q[kq + 2] : 0u;
				lo = __builtin_amdgcn_alignbit(x1, x0, sh); hi = __builtin_amdgcn_alignbit(x2, x1, sh);
			} else {
				lo = 2 * b < nw1 ? q[2 * b] : 0u; hi = 2 * b + 1 < nw1 ? q[2 * b + 1] : 0u;
			}
			B0[w] = even16(lo) | (even16(hi) << 16);
			B1[w] = even16(lo >> 1) | (even16(hi >> 1) << 16);
			Pv[w] = SEMI ? 0u : 0xffffffffu; Mv[w] = 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   /* LDS writes of this wave before its reads */
		const int nlanes = l1 > 0 ? (((l1 - 1) >> 5) / W) + 1 : 0;   /* lanes of the alignment that hold rows <= l1 */
		uint32_t hp_out = 0, hm_out = 0;                   /* horizontal differences of my last word (previous step), whole words */
		/* the alignments of the wave step together: the longest one decides the trip count */
		const int steps_mine = (fits && have && l1 > 0 && l2 > 0) ? l2 + nlanes - 1 : 0;
		int nsteps = steps_mine;
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) nsteps = imax(nsteps, __shfl_xor(nsteps, d));
		uint32_t cw = 0;                                   /* (G == 1) the sixteen codes of s2 around column t */
		int semi_d = 0, semi_ub2 = 0;                      /* SEMI: D'(l1, column), and the bound so far (column 0 gives 0) */
		(void)semi_d; (void)semi_ub2;
		for (int t = 0; t < nsteps; ++t) {
			/* the word above my first one: lane - 1's last word one step ago; first lane of an alignment: the border, +1 */
			uint32_t pP, pM;
			if constexpr (G == 1) { pP = 0x80000000u; pM = 0u; }
			else {
				pP = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hp_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
				pM = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm_out, 0x138, 0xf, 0xf, false);
				if (lg == 0) { pP = 0x80000000u; pM = 0u; }
			}
			const int j = t - lg;
			const bool active = lg < nlanes && j >= 0 && j < l2 && t < steps_mine;
			uint32_t c = 0;
			if constexpr (G == 1) {                        /* every lane is at column t: one LDS word per sixteen columns */
				if ((t & 15) == 0) cw = t < l2 ? ref[t >> 4] : 0u;
				c = cw; cw >>= 2;
			}
			if (active) {
				if constexpr (G != 1) c = ref[j >> 4] >> ((j & 15) * 2);
				const uint32_t nC0 = (c & 1u) - 1u, nC1 = ((c >> 1) & 1u) - 1u;   /* ~(bit of the code spread over the word) */
#pragma unroll
				for (int w = 0; w < W; ++w) {
					const uint32_t Eq = (B0[w] ^ nC0) & (B1[w] ^ nC1);
					const uint32_t Eqh = Eq | (pM >> 31);      /* hin < 0 */
					const uint32_t sum = (Eqh & Pv[w]) + Pv[w];
					const uint32_t Xh = (sum ^ Pv[w]) | Eqh;
					const uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
					const uint32_t Mh = Pv[w] & Xh;
					const uint32_t Phs = __builtin_amdgcn_alignbit(Ph, pP, 31);   /* (Ph << 1) | hin > 0 */
					const uint32_t Mhs = __builtin_amdgcn_alignbit(Mh, pM, 31);
					const uint32_t Xv = Eq | Mv[w];
					Pv[w] = Mhs | ~(Xv | Phs);
					Mv[w] = Phs & Xv;
					pP = Ph; pM = Mh;
				}
				hp_out = pP; hm_out = pM;
				if constexpr (SEMI) {
					/* the last word's horizontal difference in its last row = D'(l1, t + 1) - D'(l1, t) */
					semi_d += (int)(pP >> 31) - (int)(pM >> 31);
					if (t + 1 <= l2 - 1) semi_ub2 = imax(semi_ub2, a.semi_m2 * (t + 1) - a.semi_k * semi_d);
				}
			}
		}
		if constexpr (SEMI) {
			if (have) {
				const bool cand = !fits || semi_ub2 >= a.semi_min2;
				a.score[p] = semi_ub2 >> 1;                /* an upper bound of the score; the exact sweep overwrites it for candidates */
				if (a.end_i) a.end_i[p] = l1;
				if (a.end_j) a.end_j[p] = 0;
				if (a.state) a.state[p] = 0;               /* 0: not swept (score < T proven) */
				if (cand) a.cand_order[atomicAdd(a.cand_count, 1)] = (int)p;
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			continue;
		}
		/* ---- result: D(l1, l2) = l2 + the vertical differences of the last column over rows 1 .. l1 ---- */
		int sum = 0;
#pragma unroll
		for (int w = 0; w < W; ++w) {
			const int valid = l1 - 32 * (lg * W + w);      /* rows of this word inside s1 */
			const uint32_t vm = valid >= 32 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << valid) - 1u);
			sum += __popc(Pv[w] & vm) - __popc(Mv[w] & vm);
		}
#pragma unroll
		for (int d = G / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
		int d;
		if (l1 <= 0 || l2 <= 0) d = imax(l1, 0) + imax(l2, 0);   /* border: D(i,0) = i, D(0,j) = j */
		else d = l2 + sum;
		if (have && lg == 0) {
			a.score[p] = fits ? d : INT32_MIN;
			if (a.end_i) a.end_i[p] = l1;
			if (a.end_j) a.end_j[p] = l2;
			if (a.state) a.state[p] = 2;
			if (a.nops) a.nops[p] = fits ? 0 : -1;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   /* the LDS region is reused by the next item */
	}
}

} /* namespace at */
