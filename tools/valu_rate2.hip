// More issue-rate probes (see valu_rate.hip): cndmask forms, cmp+cndmask pairs, v_perm, SDWA add, pk ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define KERNEL(NAME, INS)                                                                           \
	__global__ __launch_bounds__(64) void NAME(int *out, int iters, int a, int b, int sa)           \
	{                                                                                               \
		int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
		unsigned long long m = 0x5555555555555555ull;                                               \
		for (int i = 0; i < iters; ++i) {                                                           \
			asm volatile(REP8(INS "\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(sa), "s"(m) : "vcc"); \
		}                                                                                           \
		out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                \
	}
KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc")
KERNEL(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %0, %8, %11\n v_cndmask_b32_e64 %1, %1, %8, %11\n v_cndmask_b32_e64 %2, %2, %8, %11\n v_cndmask_b32_e64 %3, %3, %8, %11\n v_cndmask_b32_e64 %4, %4, %8, %11\n v_cndmask_b32_e64 %5, %5, %8, %11\n v_cndmask_b32_e64 %6, %6, %8, %11\n v_cndmask_b32_e64 %7, %7, %8, %11")
KERNEL(k_cmp_cnd, "v_cmp_gt_i32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_i32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_i32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_i32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc")
KERNEL(k_cmp_cnd_far, "v_cmp_gt_i32 vcc, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8")
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9")
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %3, %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %4, %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %5, %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %6, %6, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %7, %7, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL(k_or_b32, "v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8")
KERNEL(k_xor_b32, "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8")
KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8")
KERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n v_ashrrev_i32 %4, 1, %4\n v_ashrrev_i32 %5, 1, %5\n v_ashrrev_i32 %6, 1, %6\n v_ashrrev_i32 %7, 1, %7")
KERNEL(k_min_u32, "v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8")
KERNEL(k_max_u16, "v_max_u16 %0, %0, %8\n v_max_u16 %1, %1, %8\n v_max_u16 %2, %2, %8\n v_max_u16 %3, %3, %8\n v_max_u16 %4, %4, %8\n v_max_u16 %5, %5, %8\n v_max_u16 %6, %6, %8\n v_max_u16 %7, %7, %8")
KERNEL(k_pk_add_sat, "v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n v_pk_add_i16 %4, %4, %8 clamp\n v_pk_add_i16 %5, %5, %8 clamp\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_add_i16 %7, %7, %8 clamp")
KERNEL(k_mix_add_max, "v_add_u32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_max_i32 %7, %7, %8")
KERNEL(k_dep_add, "v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %0, %0, %8")
KERNEL(k_dep_max, "v_max_i32 %0, %0, %8\n v_max_i32 %0, %0, %9\n v_max_i32 %0, %0, %8\n v_max_i32 %0, %0, %9\n v_max_i32 %0, %0, %8\n v_max_i32 %0, %0, %9\n v_max_i32 %0, %0, %8\n v_max_i32 %0, %0, %9")
KERNEL(k_snop, "v_add_u32 %0, %0, %8\n s_nop 0\n v_add_u32 %1, %1, %8\n s_nop 0\n v_add_u32 %2, %2, %8\n s_nop 0\n v_add_u32 %3, %3, %8\n s_nop 0")
typedef void (*kfn)(int *, int, int, int, int);
static void run(const char *name, kfn f, int wps, int inst_per_iter)
{
	int *out; int nblk = 256 * 4 * wps, iters = 4000;
	(void)hipMalloc(&out, nblk * 64 * sizeof(int));
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	f<<<nblk, 64>>>(out, 10, 1, 3, 5);
	(void)hipEventRecord(e0);
	f<<<nblk, 64>>>(out, iters, 1, 3, 5);
	(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	double inst = (double)nblk * iters * inst_per_iter;
	printf("%-14s w/SIMD=%d  %.2f cyc/inst/SIMD (@2.4GHz)\n", name, wps, 1024.0 * 2.4e9 / (inst / (ms * 1e-3))); fflush(stdout);
	(void)hipFree(out);
}
#define RUN(n) run(#n, k_##n, w, 64)
int main()
{
	for (int w : {1, 2, 3, 4}) {
		RUN(cnd_vcc); RUN(cnd_sgpr); RUN(cmp_cnd); RUN(cmp_cnd_far); RUN(perm); RUN(add_sdwa); RUN(or_b32); RUN(xor_b32); RUN(sub_u32);
		RUN(lshl); RUN(ashr); RUN(min_u32); RUN(max_u16); RUN(pk_add_sat); RUN(mix_add_max); RUN(dep_add); RUN(dep_max); RUN(snop);
	}
	return 0;
}
