/*
 * at_pack.hip.h -- GPU-side input packing.
 *
 * The host entry at_align_batch uploads the raw sequence bytes once; these kernels turn them into the
 * packed words the sweep kernels read -- 2 bits per base (A,C,G,T -> 0..3, 16 bases per int32) or,
 * when the batch contains any other byte (the reference compares raw bytes, alignment.h:449, and two of
 * its own fixtures are protein), 4 bytes per int32.  One wavefront per sequence, one lane per output word,
 * byte loads of consecutive lanes are consecutive 16-byte (4-byte) runs.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace at {

struct PackArgs {
	long long nseq;                /* 2 * npairs: sequence 2k = s1 of pair k, 2k+1 = s2 */
	const uint8_t *blob;
	const long long *off;          /* [nseq] byte offset */
	const int *len;                /* [nseq] */
	const long long *woff;         /* [nseq] word offset in `words` */
	uint32_t *words;
	int *not_acgt;                 /* set to 1 if a byte outside ACGT is seen (2-bit kernel only) */
};

template <int BITS>
__global__ __launch_bounds__(256) void at_pack(const PackArgs a)
{
	constexpr int BPW = 32 / BITS;
	const int lane = threadIdx.x & 63;
	const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
	int bad = 0;
	for (long long s = wave; s < a.nseq; s += nwaves) {
		const uint8_t *src = a.blob + a.off[s];
		const int len = a.len[s];
		uint32_t *dst = a.words + a.woff[s];
		const int nw = (len + BPW - 1) / BPW + 1;   /* one zero word of slack behind every sequence */
		for (int w = lane; w < nw; w += 64) {
			uint32_t v = 0;
#pragma unroll
			for (int b = 0; b < BPW; ++b) {
				const int idx = w * BPW + b;
				if (idx < len) {
					const uint32_t c = src[idx];
					if constexpr (BITS == 2) {
						const uint32_t raw = (c >> 1) & 3u;                       /* A0 C1 T2 G3 */
						bad |= ((0x47544341u >> (8 * raw)) & 0xffu) != c;
						v |= (raw ^ (raw >> 1)) << (2 * b);                        /* A0 C1 G2 T3 */
					} else {
						v |= c << (8 * b);
					}
				}
			}
			dst[w] = v;
		}
	}
	if constexpr (BITS == 2) {
		if (__any(bad) && lane == 0) atomicOr(a.not_acgt, 1);
	}
}

/* per-sequence word offsets / lengths (s1 of pair k at 2k, s2 at 2k + 1) -> the per-pair arrays the sweep kernels take.  The host
 * uploads the interleaved form once; this saves it from uploading the same numbers a second time */
__global__ __launch_bounds__(256) void at_split_desc(const long long *swoff, const int *slen, long long n,
                                                     long long *woff1, long long *woff2, int *len1, int *len2)
{
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
		woff1[k] = swoff[2 * k]; woff2[k] = swoff[2 * k + 1];
		len1[k] = slen[2 * k]; len2[k] = slen[2 * k + 1];
	}
}

/* *flag (preset to 1) becomes 0 unless every pair has exactly the lengths (l1, l2) */
__global__ __launch_bounds__(256) void at_check_uniform(const int *len1, const int *len2, long long n, int l1, int l2, int *flag)
{
	int bad = 0;
	for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x)
		bad |= len1[k] != l1 || len2[k] != l2;
	if (__any(bad) && (threadIdx.x & 63) == 0) atomicAnd(flag, 0);
}

} /* namespace at */
