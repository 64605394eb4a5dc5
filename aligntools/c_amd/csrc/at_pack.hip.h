/*
 * at_pack.hip.h -- GPU-side input packing.
 *
 * The host entry at_align_batch uploads the raw sequence bytes once; these kernels turn them into the
 * packed words the sweep kernels read -- 2 bits per base (A,C,G,T -> 0..3, 16 bases per int32) or,
 * when the batch contains any other byte (the reference compares raw bytes, alignment.h:449, and two of
 * its own fixtures are protein), 4 bytes per int32.  Sixteen lanes per sequence, one lane per output word, aligned dword loads.
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

/* Sixteen lanes per sequence (sixteen sequences per workgroup: a 150-base read is 10 words + the slack word), one output word
 * per lane and pass.  The bytes of a word are fetched as ALIGNED dwords and brought into place with v_alignbit, whatever the
 * alignment of the sequence in the blob: four or five dword loads per word instead of sixteen byte loads (the byte-wise form took
 * 33 us per 33k reads alone and 140-180 us beside the sweeps of the other chunks of a batch -- before its own chunk's sweep could
 * start).  The blob must be readable up to 20 bytes behind its last base (the callers allocate 32). */
template <int BITS>
__global__ __launch_bounds__(256) void at_pack(const PackArgs a)
{
	constexpr int BPW = 32 / BITS;
	const int l16 = threadIdx.x & 15;
	const long long first = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
	const long long stride = (long long)gridDim.x * 16;
	int bad = 0;
	for (long long s = first; s < a.nseq; s += stride) {
		const uint8_t *src = a.blob + a.off[s];
		const int len = a.len[s];
		uint32_t *dst = a.words + a.woff[s];
		const int nw = (len + BPW - 1) / BPW + 1;   /* one zero word of slack behind every sequence */
		for (int w = l16; w < nw; w += 16) {
			uint32_t v = 0;
			const int left = len - w * BPW;          /* bases of this word and beyond */
			if (left > 0) {
				const uintptr_t p = (uintptr_t)(src + (size_t)w * BPW);
				const uint32_t *q = (const uint32_t *)(p & ~(uintptr_t)3);
				const unsigned sh = (unsigned)(p & 3) * 8;
				if constexpr (BITS == 8) {
					const uint32_t d0 = q[0], d1 = sh ? q[1] : 0u;   /* (4 bytes from p: inside the blob's slack at the end) */
					v = __builtin_amdgcn_alignbit(d1, d0, sh);
					if (left < 4) v &= (1u << (8 * left)) - 1u;
				} else {
					uint32_t d[5];                               /* dword i is needed by output dword i and, when the bytes are shifted, by i - 1 */
#pragma unroll
					for (int i = 0; i < 5; ++i) d[i] = (i < 4 && left > 4 * i) || (sh && i > 0 && left > 4 * (i - 1)) ? q[i] : 0u;
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						const int cnt = left - 4 * i;                 /* bases in this dword */
						if (cnt > 0) {
							const uint32_t x = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
							const uint32_t m = cnt >= 4 ? 0xffffffffu : (1u << (8 * cnt)) - 1u;
							const uint32_t raw = (x >> 1) & 0x03030303u;                      /* A0 C1 T2 G3 */
							bad |= ((__builtin_amdgcn_perm(0u, 0x47544341u, raw) ^ x) & m) != 0;
							const uint32_t c = (raw ^ (raw >> 1)) & 0x03030303u & m;         /* A0 C1 G2 T3 */
							uint32_t t = c | (c >> 6);
							t |= t >> 12;
							v |= (t & 0xffu) << (8 * i);
						}
					}
				}
			}
			dst[w] = v;
		}
	}
	if constexpr (BITS == 2) {
		if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(a.not_acgt, 1);
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
