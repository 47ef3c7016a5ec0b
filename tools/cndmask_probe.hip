// tools/cndmask_probe.hip -- why does v_cndmask_b32 measure at 0.044 wave64 instructions per SIMD-cycle in valu_issue.hip (a fifth of every
// other opcode)?  The same harness (8 chains, 8 waves per SIMD, wall clock) on variants: mask in vcc / in an SGPR pair, dependent chain /
// independent results, with the compare that produces the mask, and the arithmetic forms that can stand in for a select.
//   hipcc --offload-arch=gfx950 -O2 -o tools/cndmask_probe tools/cndmask_probe.hip && tools/cndmask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITERS 4096
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define KERNEL(NAME, PRE, ASM8, N)                                                                            \
__global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned seed)                                     \
{                                                                                                              \
	unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x55u, a3 = a0 + 77u,                    \
	         a4 = a0 * 5u, a5 = a0 ^ 0x1234u, a6 = a0 + 9u, a7 = a0 * 7u + 3u;                               \
	unsigned k0 = seed * 2654435761u + 12345u, k1 = seed ^ 0x00070003u;                                     \
	asm volatile(PRE : : "v"(k0), "v"(k1) : "vcc", "s20", "s21");                                          \
	for (int it = 0; it < ITERS; it++) {                                                                    \
		asm volatile(ASM8 ASM8 ASM8 ASM8                                                                    \
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
		             : "v"(k0), "v"(k1) : "vcc", "s20", "s21");                                             \
	}                                                                                                       \
	unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                     \
	if (r == 0x13579bdfu) out[blockIdx.x] = r;                                                              \
}                                                                                                              \
static const int NAME##_n = N;

#define EACH(A, B) A "%0" B "\n\t" A "%1" B "\n\t" A "%2" B "\n\t" A "%3" B "\n\t" A "%4" B "\n\t" A "%5" B "\n\t" A "%6" B "\n\t" A "%7" B "\n\t"
#define EACH2(A, M, B) A "%0" M "%0" B "\n\t" A "%1" M "%1" B "\n\t" A "%2" M "%2" B "\n\t" A "%3" M "%3" B "\n\t" A "%4" M "%4" B "\n\t" A "%5" M "%5" B "\n\t" A "%6" M "%6" B "\n\t" A "%7" M "%7" B "\n\t"

#define SETMASK "v_cmp_gt_u32 vcc, %0, %1\n\ts_mov_b64 s[20:21], vcc\n\t"
KERNEL(k_add,            SETMASK, EACH2("v_add_u32 ", ", ", ", %8"), 8)
KERNEL(k_cnd_vcc_chain,  SETMASK, EACH2("v_cndmask_b32 ", ", ", ", %8, vcc"), 8)
KERNEL(k_cnd_sgpr_chain, SETMASK, EACH2("v_cndmask_b32_e64 ", ", ", ", %8, s[20:21]"), 8)
KERNEL(k_cnd_vcc_indep,  SETMASK, EACH("v_cndmask_b32 ", ", %8, %9, vcc"), 8)
KERNEL(k_cnd_sgpr_indep, SETMASK, EACH("v_cndmask_b32_e64 ", ", %8, %9, s[20:21]"), 8)
// the usual pair: a compare writing vcc, the select reading it (16 instructions per group)
KERNEL(k_cmp_cnd_vcc,    SETMASK, "v_cmp_gt_i32 vcc, %0, %8\n\tv_cndmask_b32 %0, %0, %9, vcc\n\tv_cmp_gt_i32 vcc, %1, %8\n\tv_cndmask_b32 %1, %1, %9, vcc\n\t"
                                  "v_cmp_gt_i32 vcc, %2, %8\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cmp_gt_i32 vcc, %3, %8\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
                                  "v_cmp_gt_i32 vcc, %4, %8\n\tv_cndmask_b32 %4, %4, %9, vcc\n\tv_cmp_gt_i32 vcc, %5, %8\n\tv_cndmask_b32 %5, %5, %9, vcc\n\t"
                                  "v_cmp_gt_i32 vcc, %6, %8\n\tv_cndmask_b32 %6, %6, %9, vcc\n\tv_cmp_gt_i32 vcc, %7, %8\n\tv_cndmask_b32 %7, %7, %9, vcc\n\t", 16)
KERNEL(k_cmp_only,       SETMASK, EACH("v_cmp_gt_i32 vcc, ", ", %8"), 8)
// stand-ins: sign mask + bit-field insert (2 instructions per select), min/max, and-or
KERNEL(k_ashr_bfi,       SETMASK, "v_ashrrev_i32 %0, 31, %0\n\tv_bfi_b32 %0, %0, %8, %9\n\tv_ashrrev_i32 %1, 31, %1\n\tv_bfi_b32 %1, %1, %8, %9\n\t"
                                  "v_ashrrev_i32 %2, 31, %2\n\tv_bfi_b32 %2, %2, %8, %9\n\tv_ashrrev_i32 %3, 31, %3\n\tv_bfi_b32 %3, %3, %8, %9\n\t"
                                  "v_ashrrev_i32 %4, 31, %4\n\tv_bfi_b32 %4, %4, %8, %9\n\tv_ashrrev_i32 %5, 31, %5\n\tv_bfi_b32 %5, %5, %8, %9\n\t"
                                  "v_ashrrev_i32 %6, 31, %6\n\tv_bfi_b32 %6, %6, %8, %9\n\tv_ashrrev_i32 %7, 31, %7\n\tv_bfi_b32 %7, %7, %8, %9\n\t", 16)
KERNEL(k_max,            SETMASK, EACH2("v_max_i32 ", ", ", ", %8"), 8)
KERNEL(k_med3,           SETMASK, EACH2("v_med3_i32 ", ", ", ", %8, %9"), 8)
KERNEL(k_sub_co,         SETMASK, EACH2("v_sub_co_u32 ", ", vcc, ", ", %8"), 8)
KERNEL(k_subb,           SETMASK, EACH2("v_subb_co_u32 ", ", vcc, ", ", %8, vcc"), 8)
KERNEL(k_cmpx,           "", EACH("v_cmp_gt_i32 s[20:21], ", ", %8"), 8)

typedef void (*kfn)(unsigned *, unsigned);
struct entry { const char *name; kfn fn; int per_group; };

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const double clk = prop.clockRate * 1e3;
	unsigned *out;
	CHK(hipMalloc(&out, 1 << 20));
#define E(n) {#n, n, n##_n}
	std::vector<entry> es = { E(k_add), E(k_cnd_vcc_chain), E(k_cnd_sgpr_chain), E(k_cnd_vcc_indep), E(k_cnd_sgpr_indep), E(k_cmp_cnd_vcc), E(k_cmp_only),
	                          E(k_ashr_bfi), E(k_max), E(k_med3), E(k_sub_co), E(k_subb), E(k_cmpx) };
	hipEvent_t ea, eb;
	CHK(hipEventCreate(&ea));
	CHK(hipEventCreate(&eb));
	printf("%-20s %10s %10s   wave64 instructions per SIMD-cycle (wall clock at %.2f GHz), 1 and 8 waves per SIMD\n", "variant", "w1", "w8", clk * 1e-9);
	for (auto &e : es) {
		double r[2];
		for (int k = 0; k < 2; k++) {
			const int wps = k ? 8 : 1, grid = cus * wps;
			hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 1u);
			CHK(hipDeviceSynchronize());
			CHK(hipEventRecord(ea));
			hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, 2u);
			CHK(hipEventRecord(eb));
			CHK(hipDeviceSynchronize());
			float ms = 0;
			CHK(hipEventElapsedTime(&ms, ea, eb));
			r[k] = (double)ITERS * 4 * e.per_group * wps / (ms * 1e-3 * clk);
		}
		printf("%-20s %10.4f %10.4f\n", e.name, r[0], r[1]);
	}
	return 0;
}
