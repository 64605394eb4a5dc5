/*
 * at_render.hip.h -- GPU-side output rendering.
 *
 * The sweep kernels leave, per pair, the traceback as op codes in END -> START order plus the cell the
 * traceback started from.  The reference turns the same walk into two gapped strings character by
 * character and then reverses them (trace_back_* + strrev, alignment.h:372-412, 558-592, 766-800, 896-922,
 * 172-184).  This kernel produces those two strings directly in forward order, in HBM, from the packed
 * sequences: one wavefront per pair, 64 ops per pass; the row / column index an op consumes is the end
 * cell minus the number of row / column consuming ops before it, i.e. a wave-wide prefix count (ballot +
 * popcount), so no lane walks the op list serially.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace at {

struct RenderArgs {
	long long npairs;
	const uint32_t *seq;
	const long long *woff1, *woff2;
	const int *end_i, *end_j;
	const uint8_t *ops;
	const long long *ops_off;
	const int *nops;
	uint8_t *r1, *r2;
	const long long *str_off;      /* where pair k's strings go; NULL = ops_off */
	int nul;                       /* write a terminating 0 behind each string */
	int *bad;                      /* set to 1 if an op list walks off its sequences */
};

template <int BITS>
__device__ __forceinline__ uint32_t render_base(const uint32_t *seq, long long woff, int idx)
{
	if constexpr (BITS == 2) {
		const uint32_t w = seq[woff + (idx >> 4)];
		const uint32_t code = (w >> (2 * (idx & 15))) & 3u;
		return (0x54474341u >> (8 * code)) & 0xffu;            /* 0..3 -> 'A','C','G','T' */
	} else {
		const uint32_t w = seq[woff + (idx >> 2)];
		return (w >> (8 * (idx & 3))) & 0xffu;
	}
}

/* W lanes per pair, 64 / W pairs per wavefront side by side.  An alignment of unrelated 150-base reads has a dozen ops, one of
 * 36-base reads fewer: with one pair per wavefront the kernel is a chain of four dependent loads per pair with most lanes idle
 * (1.7 M pairs of 36 bases: 0.63 ms, more than half of their sweep); with W = 16 four such chains share a wavefront. */
template <int BITS, int W>
__global__ __launch_bounds__(256) void at_render_k(const RenderArgs a)
{
	constexpr int NGR = 64 / W;
	const int lane = threadIdx.x & 63, g = lane / W, lg = lane % W;
	const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
	const unsigned long long gm = W == 64 ? ~0ull : ((1ull << W) - 1ull) << (W * g);   /* my group's lanes */
	const unsigned long long lt = ((1ull << lg) - 1ull) << (W * g);                     /* ... below me */
	int bad = 0;
	for (long long kk = wave * NGR; kk < a.npairs; kk += nwaves * NGR) {
		const long long k = kk + g;
		const bool valid = k < a.npairs;
		const int nraw = valid ? a.nops[k] : -1;               /* < 0: flagged by the sweep kernel already (or no pair) */
		const int n = nraw < 0 ? 0 : nraw;
		int nmax = n;
#pragma unroll
		for (int d = W; d < 64; d <<= 1) nmax = max(nmax, __shfl_xor(nmax, d));
		long long oo = 0, so = 0, w1 = 0, w2 = 0;
		int i = 0, j = 0;
		if (nraw >= 0) {
			oo = a.ops_off[k];
			so = a.str_off ? a.str_off[k] : oo;
			w1 = a.woff1[k]; w2 = a.woff2[k];
			i = a.end_i[k]; j = a.end_j[k];
		}
		for (int base = 0; base < nmax; base += W) {
			const int p = base + lg;
			const bool act = p < n;
			const uint32_t op = act ? a.ops[oo + p] : 0xffu;
			const bool di = act && op <= 1u;                   /* MID, LOW consume a row    */
			const bool dj = act && op != 1u;                   /* MID, UPP, JUMP a column   */
			const unsigned long long mi = __ballot(di), mj = __ballot(dj);
			const int ii = i - __popcll(mi & lt) - 1;
			const int jj = j - __popcll(mj & lt) - 1;
			uint32_t c1 = '-', c2 = '-';
			if (di) { if (ii >= 0) c1 = render_base<BITS>(a.seq, w1, ii); else bad = 1; }
			if (dj) { if (jj >= 0) c2 = render_base<BITS>(a.seq, w2, jj); else bad = 1; }
			if (act) {
				if (op > 3u) bad = 1;
				a.r1[so + (n - 1 - p)] = (uint8_t)c1;
				a.r2[so + (n - 1 - p)] = (uint8_t)c2;
			}
			i -= __popcll(mi & gm);
			j -= __popcll(mj & gm);
		}
		if (a.nul && lg == 0 && nraw >= 0) { a.r1[so + n] = 0; a.r2[so + n] = 0; }
	}
	if (__any(bad) && lane == 0) atomicOr(a.bad, 1);
}

/*
 * CIGAR compaction for the result gather (sizes first, then one payload): the sweep kernels write
 * pair k's ops into a slot of len1+len2 bytes; a batch of short alignments is mostly slack.  at_scan_tiles + at_scan_nops turn
 * the counts into exclusive offsets (off[npairs] = total), at_compact_k copies the used part of every slot to
 * packed[off[k] ..).  Pairs that would end beyond `cap` are skipped; the total tells the caller.
 */
constexpr int SCAN_TILE = 1024;   /* counts per workgroup: 256 threads x 4 */

__device__ __forceinline__ void scan_load4(const int *nops, long long n, long long k0, int v[4])
{
#pragma unroll
	for (int q = 0; q < 4; ++q) v[q] = k0 + q < n ? max(nops[k0 + q], 0) : 0;
}

/* pass 1: one total per tile of 1024 counts */
__global__ __launch_bounds__(256) void at_scan_tiles(const int *nops, long long n, long long *tile_sum)
{
	__shared__ long long wsum[4];
	const int t = threadIdx.x;
	int v[4];
	scan_load4(nops, n, (long long)blockIdx.x * SCAN_TILE + 4 * t, v);
	long long s = (long long)v[0] + v[1] + v[2] + v[3];
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
	if ((t & 63) == 0) wsum[t >> 6] = s;
	__syncthreads();
	if (t == 0) tile_sum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

/* pass 2: exclusive offsets; a tile's base is the sum of the tile totals before it */
__global__ __launch_bounds__(256) void at_scan_nops(const int *nops, long long n, const long long *tile_sum, long long *off)
{
	__shared__ long long wsum[4], wbase[4];
	const int t = threadIdx.x, lane = t & 63, w = t >> 6;
	long long base = 0;
	for (int b = t; b < (int)blockIdx.x; b += 256) base += tile_sum[b];
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) base += __shfl_xor(base, d);
	if (lane == 0) wbase[w] = base;
	const long long k0 = (long long)blockIdx.x * SCAN_TILE + 4 * t;
	int v[4];
	scan_load4(nops, n, k0, v);
	const long long mine = (long long)v[0] + v[1] + v[2] + v[3];
	long long inc = mine;                                  /* inclusive scan over the wavefront */
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const long long up = __shfl_up(inc, d);
		if (lane >= d) inc += up;
	}
	if (lane == 63) wsum[w] = inc;
	__syncthreads();
	long long run = wbase[0] + wbase[1] + wbase[2] + wbase[3] + inc - mine;
	for (int q = 0; q < w; ++q) run += wsum[q];
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		if (k0 + q < n) off[k0 + q] = run;
		run += v[q];
	}
	/* the thread that owns the last count also writes the total */
	if (k0 <= n - 1 && n - 1 < k0 + 4) off[n] = run;
}

__global__ __launch_bounds__(256) void at_compact_k(long long npairs, const uint8_t *ops, const long long *ops_off, const int *nops,
                                                    uint8_t *packed, const long long *off, long long cap)
{
	/* 16 lanes per pair: local alignments of unrelated reads are a dozen ops long */
	const int l = threadIdx.x & 15;
	const long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
	const long long ngrp = ((long long)gridDim.x * blockDim.x) >> 4;
	for (long long k = grp; k < npairs; k += ngrp) {
		const int n = nops[k];
		const long long dst = off[k];
		if (n <= 0 || dst + n > cap) continue;
		const uint8_t *src = ops + ops_off[k];
		for (int p = l; p < n; p += 16) packed[dst + p] = src[p];
	}
}

/* dst[k] = src[k] + k: where string k starts when every packed string is followed by one NUL */
__global__ __launch_bounds__(256) void at_add_index(const long long *src, long long *dst, long long n)
{
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) dst[k] = src[k] + k;
}

} /* namespace at */
