// valu_issue.hip -- issue cost of the VALU instructions the sweep kernels are made of, on gfx950, in SHADER CYCLES
// measured inside the kernel (s_memtime around the loop) together with the clock the chip actually held
// (s_memtime / s_memrealtime, the latter ticks at 100 MHz), so that nothing rests on an assumed 2.4 GHz.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue > profiles/r02/valu_issue.txt
//
// Every kernel runs ITERS iterations of 64 independent instructions (8 accumulators x 8) in one wavefront per
// workgroup; the grid is 256 CUs x 4 SIMDs x W workgroups, all resident at once, so every SIMD holds W waves.
// Reported per instruction and W: `wave` = shader cycles between two consecutive instructions of ONE wave (median over
// waves of delta s_memtime / (ITERS x 64)); `simd` = cycles per wave64 instruction per SIMD = the launch's HIP-event
// time x the measured clock / (ITERS x 64 x W) -- the number a roofline needs; and the median clock in GHz.
// The output of this program is the `cyc` table bench.py prices SQ_INSTS_VALU with (profiles/r02/valu_issue.json).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define A2(OP, SFX)                                                                                                        \
	OP " %0, %0, %8" SFX "\n" OP " %1, %1, %8" SFX "\n" OP " %2, %2, %8" SFX "\n" OP " %3, %3, %8" SFX "\n"               \
	OP " %4, %4, %8" SFX "\n" OP " %5, %5, %8" SFX "\n" OP " %6, %6, %8" SFX "\n" OP " %7, %7, %8" SFX "\n"
#define A3(OP, SFX)                                                                                                        \
	OP " %0, %0, %8, %9" SFX "\n" OP " %1, %1, %8, %9" SFX "\n" OP " %2, %2, %8, %9" SFX "\n" OP " %3, %3, %8, %9" SFX "\n" \
	OP " %4, %4, %8, %9" SFX "\n" OP " %5, %5, %8, %9" SFX "\n" OP " %6, %6, %8, %9" SFX "\n" OP " %7, %7, %8, %9" SFX "\n"
// shift forms: dst = src0(imm) op src1
#define AS(OP)                                                                                                             \
	OP " %0, 1, %0\n" OP " %1, 1, %1\n" OP " %2, 1, %2\n" OP " %3, 1, %3\n" OP " %4, 1, %4\n" OP " %5, 1, %5\n" OP " %6, 1, %6\n" OP " %7, 1, %7\n"
// sgpr as src0
#define AG(OP)                                                                                                             \
	OP " %0, %10, %0\n" OP " %1, %10, %1\n" OP " %2, %10, %2\n" OP " %3, %10, %3\n" OP " %4, %10, %4\n" OP " %5, %10, %5\n" OP " %6, %10, %6\n" OP " %7, %10, %7\n"
// mov-like: dst = op(src)
#define A1(OP, SFX)                                                                                                        \
	OP " %0, %8" SFX "\n" OP " %1, %8" SFX "\n" OP " %2, %8" SFX "\n" OP " %3, %8" SFX "\n" OP " %4, %8" SFX "\n" OP " %5, %8" SFX "\n" OP " %6, %8" SFX "\n" OP " %7, %8" SFX "\n"
// two different ops alternating (4 + 4)
#define AM(OPA, OPB)                                                                                                       \
	OPA " %0, %0, %8\n" OPB " %1, %1, %8\n" OPA " %2, %2, %8\n" OPB " %3, %3, %8\n" OPA " %4, %4, %8\n" OPB " %5, %5, %8\n" OPA " %6, %6, %8\n" OPB " %7, %7, %8\n"
// seven of A, one of B
#define A71(OPA, OPB)                                                                                                      \
	OPA " %0, %0, %8\n" OPA " %1, %1, %8\n" OPA " %2, %2, %8\n" OPA " %3, %3, %8\n" OPA " %4, %4, %8\n" OPA " %5, %5, %8\n" OPA " %6, %6, %8\n" OPB " %7, %7, %8\n"

#define REP8(x) x x x x x x x x
constexpr int ITERS = 2000;

#define KERNEL(NAME, BODY)                                                                                                 \
	__global__ __launch_bounds__(64) void k_##NAME(unsigned long long *st, int *out, int a, int b, int sa)               \
	{                                                                                                                      \
		int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;   \
		const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                   \
		const unsigned long long c0 = __builtin_amdgcn_s_memtime();                                                       \
		for (int i = 0; i < ITERS; ++i) {                                                                                  \
			asm volatile(REP8(BODY) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)       \
			             : "v"(a), "v"(b), "s"(sa) : "vcc");                                                              \
		}                                                                                                                  \
		const unsigned long long c1 = __builtin_amdgcn_s_memtime();                                                       \
		const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                   \
		if (threadIdx.x == 0) { st[2 * blockIdx.x] = c1 - c0; st[2 * blockIdx.x + 1] = r1 - r0; }                          \
		out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                       \
	}

/* ---- 32-bit encodings (VOP1 / VOP2), VGPR operands ---- */
KERNEL(v_add_u32, A2("v_add_u32", ""))
KERNEL(v_sub_u32, A2("v_sub_u32", ""))
KERNEL(v_and_b32, A2("v_and_b32", ""))
KERNEL(v_or_b32, A2("v_or_b32", ""))
KERNEL(v_xor_b32, A2("v_xor_b32", ""))
KERNEL(v_mov_b32, A1("v_mov_b32", ""))
KERNEL(v_max_i32, A2("v_max_i32", ""))
KERNEL(v_min_i32, A2("v_min_i32", ""))
KERNEL(v_max_u32, A2("v_max_u32", ""))
KERNEL(v_lshlrev_b32, AS("v_lshlrev_b32"))
KERNEL(v_ashrrev_i32, AS("v_ashrrev_i32"))
KERNEL(v_add_f32, A2("v_add_f32", ""))
KERNEL(v_max_f32, A2("v_max_f32", ""))
KERNEL(v_mul_f32, A2("v_mul_f32", ""))
KERNEL(v_add_u16, A2("v_add_u16", ""))
KERNEL(v_max_i16, A2("v_max_i16", ""))
KERNEL(v_cndmask_vcc, A2("v_cndmask_b32", ", vcc"))
/* ---- the same operations in the 64-bit VOP3 encoding / with an SGPR source ---- */
KERNEL(v_add_u32_e64, A2("v_add_u32_e64", ""))
KERNEL(v_max_i32_e64, A2("v_max_i32_e64", ""))
KERNEL(v_or_b32_e64, A2("v_or_b32_e64", ""))
KERNEL(v_add_u32_sgpr, AG("v_add_u32"))
KERNEL(v_max_i32_sgpr, AG("v_max_i32"))
KERNEL(v_or_b32_sgpr, AG("v_or_b32"))
/* ---- VOP3 three-operand integer ---- */
KERNEL(v_max3_i32, A3("v_max3_i32", ""))
KERNEL(v_med3_i32, A3("v_med3_i32", ""))
KERNEL(v_and_or_b32, A3("v_and_or_b32", ""))
KERNEL(v_or3_b32, A3("v_or3_b32", ""))
KERNEL(v_bfi_b32, A3("v_bfi_b32", ""))
KERNEL(v_perm_b32, A3("v_perm_b32", ""))
KERNEL(v_alignbit_b32, A3("v_alignbit_b32", ""))
KERNEL(v_lshl_or_b32, A3("v_lshl_or_b32", ""))
KERNEL(v_lshl_add_u32, A3("v_lshl_add_u32", ""))
KERNEL(v_add3_u32, A3("v_add3_u32", ""))
KERNEL(v_xad_u32, A3("v_xad_u32", ""))
KERNEL(v_fma_f32, A3("v_fma_f32", ""))
KERNEL(v_max3_f32, A3("v_max3_f32", ""))
KERNEL(v_max3_i16, A3("v_max3_i16", ""))
KERNEL(v_mad_i32_i24, A3("v_mad_i32_i24", ""))
KERNEL(v_bfe_u32, A3("v_bfe_u32", ""))
KERNEL(v_bitop3_b32, A3("v_bitop3_b32", " bitop3:0x96"))   /* gfx950: any function of three words (the bit-parallel edit kernel's step) */
/* ---- VOP3P packed 16-bit ---- */
KERNEL(v_pk_add_i16, A2("v_pk_add_i16", ""))
KERNEL(v_pk_add_i16_clamp, A2("v_pk_add_i16", " clamp"))
KERNEL(v_pk_sub_i16_clamp, A2("v_pk_sub_i16", " clamp"))
KERNEL(v_pk_max_i16, A2("v_pk_max_i16", ""))
KERNEL(v_pk_min_u16, A2("v_pk_min_u16", ""))
KERNEL(v_pk_mad_i16, A3("v_pk_mad_i16", ""))
KERNEL(v_pk_lshlrev_b16, A2("v_pk_lshlrev_b16", ""))
KERNEL(v_pk_ashrrev_i16, A2("v_pk_ashrrev_i16", ""))
KERNEL(v_pk_add_f16, A2("v_pk_add_f16", ""))
KERNEL(v_pk_max_f16, A2("v_pk_max_f16", ""))
/* ---- DPP / SDWA ---- */
KERNEL(v_mov_dpp_wave_shr1, A1("v_mov_b32_dpp", " wave_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(v_mov_dpp_row_shr1, A1("v_mov_b32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(v_mov_dpp_row_shl1, A1("v_mov_b32_dpp", " row_shl:1 row_mask:0xf bank_mask:0xf"))
KERNEL(v_add_u32_dpp_row_shr1, A2("v_add_u32_dpp", " row_shr:1 row_mask:0xf bank_mask:0xf"))
KERNEL(v_add_u32_sdwa, A2("v_add_u32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"))
/* ---- mixes ---- */
KERNEL(mix_pkadd_pkmax, AM("v_pk_add_i16", "v_pk_max_i16"))
KERNEL(mix_pkmax_or, AM("v_pk_max_i16", "v_or_b32"))
KERNEL(mix_pkmax_addu32, AM("v_pk_max_i16", "v_add_u32"))
KERNEL(mix_7pkmax_1or, A71("v_pk_max_i16", "v_or_b32"))
KERNEL(mix_addu32_maxi32, AM("v_add_u32", "v_max_i32"))
KERNEL(mix_addf32_maxf32, AM("v_add_f32", "v_max_f32"))
/* the two instructions of a cell of the ramp sweeps (overlap scores only, cell-by-cell edit distance): v_add_u32_sdwa, v_max3_i32 */
#define ASM3                                                                                                              \
	"v_add_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" "v_max3_i32 %1, %1, %8, %9\n"   \
	"v_add_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" "v_max3_i32 %3, %3, %8, %9\n"   \
	"v_add_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" "v_max3_i32 %5, %5, %8, %9\n"   \
	"v_add_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n" "v_max3_i32 %7, %7, %8, %9\n"
KERNEL(mix_sdwaadd_max3, ASM3)
KERNEL(mix_bitop3_alignbit, "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n" "v_alignbit_b32 %1, %1, %8, %9\n" "v_bitop3_b32 %2, %2, %8, %9 bitop3:0xe8\n" "v_alignbit_b32 %3, %3, %8, %9\n"
       "v_bitop3_b32 %4, %4, %8, %9 bitop3:0x96\n" "v_add_u32 %5, %5, %8\n" "v_bitop3_b32 %6, %6, %8, %9 bitop3:0xe8\n" "v_xor_b32 %7, %7, %8\n")

typedef void (*kfn)(unsigned long long *, int *, int, int, int);
struct Entry { const char *name; kfn f; };
#define E(n) {#n, k_##n}
static const Entry table[] = {
	E(v_add_u32), E(v_sub_u32), E(v_and_b32), E(v_or_b32), E(v_xor_b32), E(v_mov_b32), E(v_max_i32), E(v_min_i32), E(v_max_u32),
	E(v_lshlrev_b32), E(v_ashrrev_i32), E(v_add_f32), E(v_max_f32), E(v_mul_f32), E(v_add_u16), E(v_max_i16), E(v_cndmask_vcc),
	E(v_add_u32_e64), E(v_max_i32_e64), E(v_or_b32_e64), E(v_add_u32_sgpr), E(v_max_i32_sgpr), E(v_or_b32_sgpr),
	E(v_max3_i32), E(v_med3_i32), E(v_and_or_b32), E(v_or3_b32), E(v_bfi_b32), E(v_perm_b32), E(v_alignbit_b32), E(v_lshl_or_b32),
	E(v_lshl_add_u32), E(v_add3_u32), E(v_xad_u32), E(v_fma_f32), E(v_max3_f32), E(v_max3_i16), E(v_mad_i32_i24), E(v_bfe_u32), E(v_bitop3_b32),
	E(v_pk_add_i16), E(v_pk_add_i16_clamp), E(v_pk_sub_i16_clamp), E(v_pk_max_i16), E(v_pk_min_u16), E(v_pk_mad_i16),
	E(v_pk_lshlrev_b16), E(v_pk_ashrrev_i16), E(v_pk_add_f16), E(v_pk_max_f16),
	E(v_mov_dpp_wave_shr1), E(v_mov_dpp_row_shr1), E(v_mov_dpp_row_shl1), E(v_add_u32_dpp_row_shr1), E(v_add_u32_sdwa),
	E(mix_pkadd_pkmax), E(mix_pkmax_or), E(mix_pkmax_addu32), E(mix_7pkmax_1or), E(mix_addu32_maxi32), E(mix_addf32_maxf32), E(mix_sdwaadd_max3), E(mix_bitop3_alignbit),
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
	const char *json_path = argc > 1 ? argv[1] : nullptr;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("# device: %s, %d CUs, clockRate %d kHz; ITERS %d x 64 instructions per wave\n", prop.name, cus, prop.clockRate, ITERS);
	printf("# wave = shader cycles between consecutive instructions of one wave (median over waves of delta s_memtime / instructions)\n");
	printf("# simd = cycles per wave64 instruction per SIMD: HIP-event time of the launch x measured clock / (instructions per wave x W)\n");
	printf("# GHz  = median of delta s_memtime / delta s_memrealtime x 0.1\n");
	printf("%-26s", "instruction");
	const int Ws[] = {1, 2, 3, 4};
	for (int w : Ws) printf("  W=%d wave  simd   GHz", w);
	printf("\n");
	std::string js = "{\n \"device\": \"" + std::string(prop.name) + "\", \"cus\": " + std::to_string(cus) + ",\n \"unit\": \"shader cycles per wave64 instruction per SIMD\",\n \"cyc\": {\n";
	const int maxblk = cus * 4 * 4;
	unsigned long long *st; int *out;
	CK(hipMalloc(&st, maxblk * 2 * sizeof(unsigned long long)));
	CK(hipMalloc(&out, maxblk * 64 * sizeof(int)));
	std::vector<unsigned long long> h(maxblk * 2);
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	bool first = true;
	std::vector<double> all_ghz, half_rate_cost;
	double bitpar_cost = 1e9;   /* cheapest wall cost of the bit-parallel kernel's instruction mix */
	for (const Entry &en : table) {
		printf("%-26s", en.name);
		js += std::string(first ? "" : ",\n") + "  \"" + en.name + "\": {";
		first = false;
		for (int wi = 0; wi < 4; ++wi) {
			const int w = Ws[wi], nblk = cus * 4 * w;
			en.f<<<nblk, 64>>>(st, out, 1, 3, 5);          /* warm the instruction cache and the clock */
			en.f<<<nblk, 64>>>(st, out, 1, 3, 5);
			CK(hipEventRecord(e0));
			en.f<<<nblk, 64>>>(st, out, 1, 3, 5);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			CK(hipMemcpy(h.data(), st, nblk * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
			std::vector<double> cyc(nblk), ghz(nblk);
			for (int b = 0; b < nblk; ++b) {
				cyc[b] = (double)h[2 * b] / ((double)ITERS * 64);
				ghz[b] = h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0;
			}
			std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
			const double c = cyc[nblk / 2], g = ghz[nblk / 2];
			const double wall = ms * 1e-3 * g * 1e9 / ((double)ITERS * 64 * w);   /* cycles per instruction per SIMD from wall time */
			printf("  %8.2f %5.2f %5.2f", c, wall, g);
			char buf[96];
			snprintf(buf, sizeof buf, "%s\"%d\": %.3f", wi ? ", " : "", w, wall);
			js += buf;
			if (wi == 3) { snprintf(buf, sizeof buf, ", \"ghz\": %.3f", g); js += buf; }
			if (wi >= 1) {
				all_ghz.push_back(g);
				if (strncmp(en.name, "v_pk_", 5) == 0 || strcmp(en.name, "mix_pkadd_pkmax") == 0) half_rate_cost.push_back(wall);
				if (strcmp(en.name, "mix_bitop3_alignbit") == 0) bitpar_cost = std::min(bitpar_cost, wall);
			}
		}
		js += "}";
		printf("\n"); fflush(stdout);
	}
	/* what bench.py prices SQ_INSTS_VALU with.  The step body of the sweep kernels is made of VOP3P packed 16-bit
	 * operations, VOP3 bit-field operations and DPP moves: each occupies a SIMD for 4 cycles (64 lanes over 16 lanes per
	 * cycle), and the 2-cycle VOP2 operations mixed in between cost 4 as well (the mix_* rows): cycles_per_inst.packed16 is
	 * that architectural 4.0.  The bit-parallel edit kernel's word step is eleven VOP3 operations (v_bitop3_b32, v_alignbit_b32) and
	 * three VOP2 ones, which pair up across waves now and then: cycles_per_inst.bitparallel is the cheapest this run saw for that mix
	 * (row mix_bitop3_alignbit: six VOP3, two VOP2), a little under 4.  The int32 kernels mix 2-cycle VOP2 operations into 4-cycle
	 * ones in every proportion -- the ramp sweeps (overlap scores only, cell-by-cell edit distance: v_add_u32_sdwa + v_max3_i32 per
	 * cell, rows mix_sdwaadd_max3 and mix_addu32_maxi32) run at 3.7 cycles per instruction, what 85 percent 4-cycle operations come
	 * to -- and keep the 2.0 of the full-rate class, the only bound that holds for every mix (int32, int32_ramp).
	 * measured_packed16_cost is what this run saw for the packed operations alone at 2-4 waves per SIMD (loop overhead and clock
	 * ramp included); clock_ghz is the median clock the chip held under these loads. */
	std::sort(all_ghz.begin(), all_ghz.end());
	std::sort(half_rate_cost.begin(), half_rate_cost.end());
	if (!(bitpar_cost > 2.0 && bitpar_cost < 4.0)) bitpar_cost = 4.0;
	char rb[512];
	snprintf(rb, sizeof rb, "\n },\n \"roofline\": {\"cycles_per_inst\": {\"packed16\": 4.0, \"int32\": 2.0, \"bitparallel\": %.3f, \"int32_ramp\": 2.0}, \"clock_ghz\": %.3f, \"simds\": %d, "
	         "\"measured_packed16_cost\": {\"min\": %.3f, \"median\": %.3f, \"max\": %.3f}}\n}\n",
	         bitpar_cost, all_ghz[all_ghz.size() / 2], cus * 4, half_rate_cost.front(), half_rate_cost[half_rate_cost.size() / 2], half_rate_cost.back());
	js += rb;
	if (json_path) {
		FILE *f = fopen(json_path, "w");
		if (f) { fputs(js.c_str(), f); fclose(f); }
	}
	return 0;
}
