/*
 * at_myers.hip.h -- bit-parallel edit distance for the unit-cost case of
 * edit_dist (reference src/alignment.h:291-315): D(i,0) = i, D(0,j) = j,
 * D(i,j) = min3(D(i,j-1) + 1, D(i-1,j-1) + (s1[i-1] == s2[j-1] ? 0 : u), D(i-1,j) + 1), returns D(l1,l2).
 * With the mismatch cost u == 1 (`edit -u 1`) this is the Levenshtein distance, and the column of vertical
 * differences D(i,j) - D(i-1,j) in {-1,0,+1} fits two bit vectors (Myers 1999, block form of Hyyro 2003):
 * one 32-bit word holds 32 rows, a column step costs ~17 word operations per word instead of 32 cells.
 * Any other u keeps the cell-by-cell kernel (at_sweep.hip.h, K_EDIT): a negative u, the reference's
 * default, is not a distance and has no such encoding.
 *
 * Mapping: G lanes per alignment (G = 32: two alignments per wavefront, l1 <= 1024 * W; G = 8: eight alignments of
 * reads up to 256 bases; G = 1 (round 3): ONE ALIGNMENT PER LANE, 64 per wavefront, W = 5 or 8 words for reads of up to 160 / 256
 * bases -- no skew, no idle lanes, a column step of W chained word steps per lane: 230 instead of 965 instructions per alignment
 * of 150 x 150; each alignment with its own lengths), W consecutive words (32*W rows) per lane, the words of
 * a column chained through the horizontal difference (hin/hout).
 * Like the sweep kernels the lanes are skewed: lane l works on column t - l at step t and takes the
 * horizontal difference of the word above from lane l-1's previous step by one DPP move.  The first word's
 * hin is +1 (the border D(0,j) = j).  The lane that holds row l1 adds its horizontal differences up:
 * D(l1,j) = l1 + sum.  s2 is staged in LDS as packed 2-bit words.
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
};

/* bit k of the result = (2-bit code k of `w` == c), k = 0..15 */
AT_DEV uint32_t eq16(uint32_t w, uint32_t c)
{
	const uint32_t v = w ^ (c * 0x55555555u);
	uint32_t m = ~(v | (v >> 1)) & 0x55555555u;      /* even bits: code equal */
	m = (m | (m >> 1)) & 0x33333333u;
	m = (m | (m >> 2)) & 0x0f0f0f0fu;
	m = (m | (m >> 4)) & 0x00ff00ffu;
	m = (m | (m >> 8)) & 0x0000ffffu;
	return m;
}

template <int W, int G>
__global__ __launch_bounds__(64) void at_myers(const MyersArgs a)
{
	static_assert(G == 32 || G == 8 || G == 1, "lanes per alignment");
	constexpr int NG = 64 / G;                         /* alignments per wavefront */
	const int lane = threadIdx.x;
	const int grp = lane / G, lg = lane % G;
	const int nw2max = ((a.max_l2 + 15) >> 4) + 2;
	uint32_t *ref = at_lds + grp * nw2max;             /* my alignment's s2 words */
	const long long nwork = (a.npairs + NG - 1) / NG;
	long long wnext = blockIdx.x;
	while (wnext < nwork) {
		const long long wk = wnext;
		wnext = next_work(a.queue, lane);
		const long long pin = wk * NG + grp;
		const bool have = pin < a.npairs;
		const long long p = a.order ? (long long)a.order[have ? pin : a.npairs - 1] : (have ? pin : a.npairs - 1);
		const int l1 = a.len1[p], l2 = a.len2[p];
		const uint32_t *q = a.seq + a.woff1[p], *r = a.seq + a.woff2[p];
		const bool fits = l1 >= 0 && l2 >= 0 && l1 <= 32 * G * W && l2 <= a.max_l2;
		/* ---- stage s2 ---- */
		const int nw2 = fits ? (l2 + 15) >> 4 : 0;
		for (int w = lg; w < nw2; w += G) ref[w] = r[w];
		/* ---- my W words of s1: match masks per code, all-ones vertical state ---- */
		uint32_t P0[W], P1[W], P2[W], P3[W], Pv[W], Mv[W];
		const int nw1 = fits ? (l1 + 15) >> 4 : 0;
#pragma unroll
		for (int w = 0; w < W; ++w) {
			const int b = lg * W + w;                      /* word index: rows 32b+1 .. 32b+32 */
			const uint32_t lo = 2 * b < nw1 ? q[2 * b] : 0u, hi = 2 * b + 1 < nw1 ? q[2 * b + 1] : 0u;
			/* rows behind l1 must match nothing (they never feed a lower bit anyway) */
			const int valid = l1 - 32 * b;                 /* rows of this word inside s1 */
			const uint32_t vm = valid >= 32 ? 0xffffffffu : valid <= 0 ? 0u : ((1u << valid) - 1u);
			P0[w] = (eq16(lo, 0) | (eq16(hi, 0) << 16)) & vm;
			P1[w] = (eq16(lo, 1) | (eq16(hi, 1) << 16)) & vm;
			P2[w] = (eq16(lo, 2) | (eq16(hi, 2) << 16)) & vm;
			P3[w] = (eq16(lo, 3) | (eq16(hi, 3) << 16)) & vm;
			Pv[w] = 0xffffffffu; Mv[w] = 0u;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   /* LDS writes of this wave before its reads */
		/* the word and bit that hold row l1 */
		const int ob = l1 > 0 ? (l1 - 1) >> 5 : 0;
		const int olane = ob / W, oword = ob % W, obit = l1 > 0 ? (l1 - 1) & 31 : 0;
		const int nlanes = l1 > 0 ? olane + 1 : 0;
		int acc = 0;                                       /* sum of horizontal differences at row l1 (owner lane) */
		uint32_t hp_out = 0, hm_out = 0;                   /* horizontal difference leaving my last word (previous step) */
		/* the alignments of the wave step together: the longest one decides the trip count */
		const int steps_mine = (fits && have && l1 > 0 && l2 > 0) ? l2 + nlanes - 1 : 0;
		int nsteps = steps_mine;
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) nsteps = imax(nsteps, __shfl_xor(nsteps, d));
		for (int t = 0; t < nsteps; ++t) {
			/* what the word above me (lane - 1's last word) sent out one step ago; lane 0 of an alignment: +1 */
			uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hp_out, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
			uint32_t hm = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hm_out, 0x138, 0xf, 0xf, false);
			if (lg == 0) { hp = 1u; hm = 0u; }
			const int j = t - lg;
			const bool active = lg < nlanes && j >= 0 && j < l2 && t < steps_mine;
			if (active) {
				const uint32_t c = (ref[j >> 4] >> ((j & 15) * 2)) & 3u;
				const bool c0 = (c & 1u) != 0, c1 = (c & 2u) != 0;
#pragma unroll
				for (int w = 0; w < W; ++w) {
					const uint32_t e01 = c0 ? P1[w] : P0[w], e23 = c0 ? P3[w] : P2[w];
					uint32_t Eq = c1 ? e23 : e01;
					const uint32_t Xv = Eq | Mv[w];
					Eq |= hm;                                  /* hin < 0 */
					const uint32_t Xh = (((Eq & Pv[w]) + Pv[w]) ^ Pv[w]) | Eq;
					uint32_t Ph = Mv[w] | ~(Xh | Pv[w]);
					uint32_t Mh = Pv[w] & Xh;
					if (w == oword && lg == olane) acc += (int)((Ph >> obit) & 1u) - (int)((Mh >> obit) & 1u);
					const uint32_t ph_o = Ph >> 31, mh_o = Mh >> 31;
					Ph = (Ph << 1) | hp;
					Mh = (Mh << 1) | hm;
					Pv[w] = Mh | ~(Xv | Ph);
					Mv[w] = Ph & Xv;
					hp = ph_o; hm = mh_o;
				}
				hp_out = hp; hm_out = hm;
			}
		}
		/* ---- result ---- */
		int d;
		if (l1 <= 0 || l2 <= 0) d = imax(l1, 0) + imax(l2, 0);   /* border: D(i,0) = i, D(0,j) = j */
		else d = l1 + __shfl(acc, grp * G + olane);
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
