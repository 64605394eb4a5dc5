// Microbenchmark: sustained issue rate of the integer VALU ops the DP cell uses
// (v_add_u32, v_max_i32, v_max3_i32, v_and_or_b32, v_cndmask, DPP mov) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int KIND>
__global__ __launch_bounds__(64) void k(int *out, int iters, int a, int b)
{
	int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (KIND == 0) { // add + max, 8 independent chains
				x0 = max(x0 + a, b); x1 = max(x1 + a, b); x2 = max(x2 + a, b); x3 = max(x3 + a, b);
				x4 = max(x4 + a, b); x5 = max(x5 + a, b); x6 = max(x6 + a, b); x7 = max(x7 + a, b);
			} else if (KIND == 1) { // one dependent chain
				x0 = max(x0 + a, b); x0 = max(x0 + a, b); x0 = max(x0 + a, b); x0 = max(x0 + a, b);
				x0 = max(x0 + a, b); x0 = max(x0 + a, b); x0 = max(x0 + a, b); x0 = max(x0 + a, b);
			} else if (KIND == 2) { // max3 + and_or (VOP3, 8-byte encodings)
				x0 = (max(max(x0, x1), x2) & ~15) | 10; x1 = (max(max(x1, x2), x3) & ~15) | 10;
				x2 = (max(max(x2, x3), x4) & ~15) | 10; x3 = (max(max(x3, x4), x5) & ~15) | 10;
				x4 = (max(max(x4, x5), x6) & ~15) | 10; x5 = (max(max(x5, x6), x7) & ~15) | 10;
				x6 = (max(max(x6, x7), x0) & ~15) | 10; x7 = (max(max(x7, x0), x1) & ~15) | 10;
			} else { // DPP wave_shr + add
				x0 = __builtin_amdgcn_update_dpp(x1, x0, 0x138, 0xf, 0xf, false) + a;
				x1 = __builtin_amdgcn_update_dpp(x2, x1, 0x138, 0xf, 0xf, false) + a;
				x2 = __builtin_amdgcn_update_dpp(x3, x2, 0x138, 0xf, 0xf, false) + a;
				x3 = __builtin_amdgcn_update_dpp(x4, x3, 0x138, 0xf, 0xf, false) + a;
				x4 = __builtin_amdgcn_update_dpp(x5, x4, 0x138, 0xf, 0xf, false) + a;
				x5 = __builtin_amdgcn_update_dpp(x6, x5, 0x138, 0xf, 0xf, false) + a;
				x6 = __builtin_amdgcn_update_dpp(x7, x6, 0x138, 0xf, 0xf, false) + a;
				x7 = __builtin_amdgcn_update_dpp(x0, x7, 0x138, 0xf, 0xf, false) + a;
			}
		}
	}
	out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND>
static void run(const char *name, int waves_per_simd, int ops_per_iter)
{
	int *out; int nblk = 256 * 4 * waves_per_simd; int iters = 20000;
	hipMalloc(&out, nblk * 64 * sizeof(int));
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<KIND><<<nblk, 64>>>(out, 100, 1, 3);
	hipEventRecord(e0);
	k<KIND><<<nblk, 64>>>(out, iters, 1, 3);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double inst = (double)nblk * iters * ops_per_iter;   // wave-instructions
	printf("%-28s waves/SIMD=%d  %.3f ms  %.1f G wave-inst/s  = %.2f cycles/inst/SIMD @2.4GHz\n", name, waves_per_simd, ms,
	       inst / ms / 1e6, 1024.0 * 2.4e9 / (inst / (ms * 1e-3)));
	hipFree(out);
}
int main()
{
	for (int w : {1, 2, 4, 8}) {
		run<0>("add+max 8 indep chains", w, 8 * 16);
		run<1>("add+max 1 dependent chain", w, 8 * 16);
		run<2>("max3+and_or (VOP3)", w, 8 * 16);
		run<3>("dpp wave_shr mov + add", w, 8 * 16);
	}
	return 0;
}
