// Per-instruction issue rate on gfx950 (cycles per wave64 instruction per SIMD), measured with
// inline asm so the instruction mix is exact.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(INS)                                                                                   \
	for (int i = 0; i < iters; ++i) {                                                               \
		asm volatile(REP8(INS "\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sa)); \
	}
#define KERNEL(NAME, INS)                                                                           \
	__global__ __launch_bounds__(64) void NAME(int *out, int iters, int a, int b, int sa)           \
	{                                                                                               \
		int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
		BODY(INS)                                                                                   \
		out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                \
	}
// each INS string = 8 independent instructions (one per accumulator); REP8 -> 64 instructions per iteration
KERNEL(k_add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8")
KERNEL(k_max_i32, "v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8")
KERNEL(k_max3_i32, "v_max3_i32 %0, %0, %8, %9\n v_max3_i32 %1, %1, %8, %9\n v_max3_i32 %2, %2, %8, %9\n v_max3_i32 %3, %3, %8, %9\n v_max3_i32 %4, %4, %8, %9\n v_max3_i32 %5, %5, %8, %9\n v_max3_i32 %6, %6, %8, %9\n v_max3_i32 %7, %7, %8, %9")
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9")
KERNEL(k_and_b32, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8")
KERNEL(k_bfi, "v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %8, %0, 4\n v_alignbit_b32 %1, %8, %1, 4\n v_alignbit_b32 %2, %8, %2, 4\n v_alignbit_b32 %3, %8, %3, 4\n v_alignbit_b32 %4, %8, %4, 4\n v_alignbit_b32 %5, %8, %5, 4\n v_alignbit_b32 %6, %8, %6, 4\n v_alignbit_b32 %7, %8, %7, 4")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9")
KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n v_mad_i32_i24 %4, %4, %8, %9\n v_mad_i32_i24 %5, %5, %8, %9\n v_mad_i32_i24 %6, %6, %8, %9\n v_mad_i32_i24 %7, %7, %8, %9")
KERNEL(k_add_f32, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8")
KERNEL(k_max_f32, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8")
KERNEL(k_max3_f32, "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9")
KERNEL(k_pk_add_i16, "v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %8\n v_pk_add_i16 %3, %3, %8\n v_pk_add_i16 %4, %4, %8\n v_pk_add_i16 %5, %5, %8\n v_pk_add_i16 %6, %6, %8\n v_pk_add_i16 %7, %7, %8")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc")
KERNEL(k_cmp, "v_cmp_gt_i32 vcc, %0, %8\n v_cmp_gt_i32 vcc, %1, %8\n v_cmp_gt_i32 vcc, %2, %8\n v_cmp_gt_i32 vcc, %3, %8\n v_cmp_gt_i32 vcc, %4, %8\n v_cmp_gt_i32 vcc, %5, %8\n v_cmp_gt_i32 vcc, %6, %8\n v_cmp_gt_i32 vcc, %7, %8")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8")
KERNEL(k_add_sgpr, "v_add_u32 %0, %10, %0\n v_add_u32 %1, %10, %1\n v_add_u32 %2, %10, %2\n v_add_u32 %3, %10, %3\n v_add_u32 %4, %10, %4\n v_add_u32 %5, %10, %5\n v_add_u32 %6, %10, %6\n v_add_u32 %7, %10, %7")
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 2, 2\n v_bfe_u32 %1, %1, 2, 2\n v_bfe_u32 %2, %2, 2, 2\n v_bfe_u32 %3, %3, 2, 2\n v_bfe_u32 %4, %4, 2, 2\n v_bfe_u32 %5, %5, 2, 2\n v_bfe_u32 %6, %6, 2, 2\n v_bfe_u32 %7, %7, 2, 2")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8")

typedef void (*kfn)(int *, int, int, int, int);
static void run(const char *name, kfn f, int wps)
{
	int *out; int nblk = 256 * 4 * wps, iters = 4000;
	(void)hipMalloc(&out, nblk * 64 * sizeof(int));
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	f<<<nblk, 64>>>(out, 10, 1, 3, 5);
	(void)hipEventRecord(e0);
	f<<<nblk, 64>>>(out, iters, 1, 3, 5);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	double inst = (double)nblk * iters * 64;
	printf("%-14s w/SIMD=%d  %.2f cyc/inst/SIMD (@2.4GHz)\n", name, wps, 1024.0 * 2.4e9 / (inst / (ms * 1e-3)));
	(void)hipFree(out);
}
#define RUN(n) run(#n, k_##n, w)
int main()
{
	for (int w : {1, 2, 4}) {
		RUN(add_u32); RUN(max_i32); RUN(max3_i32); RUN(and_or); RUN(and_b32); RUN(bfi); RUN(alignbit); RUN(lshl_add); RUN(add3);
		RUN(mad_i24); RUN(add_f32); RUN(max_f32); RUN(max3_f32); RUN(fma_f32); RUN(pk_add_i16); RUN(pk_max_i16); RUN(mov_dpp);
		RUN(cndmask); RUN(cmp); RUN(mov); RUN(add_sgpr); RUN(bfe); RUN(pk_add_f16);
	}
	return 0;
}
