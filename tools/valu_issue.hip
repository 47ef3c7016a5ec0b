// tools/valu_issue.hip -- the VALU issue ceiling of gfx950 per opcode class, measured: how many wave64 instructions one SIMD
// issues per cycle for the instructions the rx_power / channeliser butterflies and the rx_fm decimators are made of.
// Anchors the "valu" rooflines of bench.py.  Diagnostic only, not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue tools/valu_issue.hip && /tmp/valu_issue [json-out]
//
// Method: every kernel runs ITERS x 32 instructions of ONE opcode per wave, as 8 independent dependency chains (so the
// 4..8-cycle result latency never stalls issue), at 1, 2, 4 and 8 waves per SIMD on all 256 CUs; time from hipEvents, cycles
// from s_memtime deltas inside the kernel (the shader clock as the waves see it, 100 MHz REFCLK-independent) AND from wall time x the
// reported clock.  Reported: wave-instructions per SIMD-cycle (1.0 would be one wave64 instruction per cycle; the
// SIMD-32 datapath makes 0.5 the ceiling for full-rate ops).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define ITERS 4096
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 8 chains x 4 rounds per loop turn; OP uses %0..%7 as dst=src0 accumulators, %8 / %9 as loop-invariant operands
#define KERNEL(NAME, ASM8)                                                                                     \
__global__ __launch_bounds__(256) void NAME(unsigned *out, unsigned long long *cyc, unsigned seed)             \
{                                                                                                              \
	unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u + 1u, a2 = a0 ^ 0x55u, a3 = a0 + 77u,                    \
	         a4 = a0 * 5u, a5 = a0 ^ 0x1234u, a6 = a0 + 9u, a7 = a0 * 7u + 3u;                               \
	unsigned k0 = seed * 2654435761u + 12345u, k1 = seed ^ 0x00070003u;                                     \
	unsigned long long t0 = __builtin_readcyclecounter();                                                   \
	for (int it = 0; it < ITERS; it++) {                                                                    \
		asm volatile(ASM8 ASM8 ASM8 ASM8                                                                    \
		             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
		             : "v"(k0), "v"(k1));                                                                   \
	}                                                                                                       \
	unsigned long long t1 = __builtin_readcyclecounter();                                                   \
	unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                     \
	if (r == 0x13579bdfu) out[blockIdx.x] = r;                                                              \
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                        \
}

#define R8(FMT_A, FMT_B) \
	FMT_A "%0" FMT_B "%0" "\n\t" FMT_A "%1" FMT_B "%1" "\n\t" FMT_A "%2" FMT_B "%2" "\n\t" FMT_A "%3" FMT_B "%3" "\n\t" \
	FMT_A "%4" FMT_B "%4" "\n\t" FMT_A "%5" FMT_B "%5" "\n\t" FMT_A "%6" FMT_B "%6" "\n\t" FMT_A "%7" FMT_B "%7" "\n\t"

// dst, src0 = acc; the tail after the second acc is per-opcode
#define OP3(NAME, MNEM, TAIL) KERNEL(NAME, \
	MNEM " %0, %0" TAIL "\n\t" MNEM " %1, %1" TAIL "\n\t" MNEM " %2, %2" TAIL "\n\t" MNEM " %3, %3" TAIL "\n\t" \
	MNEM " %4, %4" TAIL "\n\t" MNEM " %5, %5" TAIL "\n\t" MNEM " %6, %6" TAIL "\n\t" MNEM " %7, %7" TAIL "\n\t")

OP3(k_add_u32,        "v_add_u32",          ", %8")
OP3(k_and_b32,        "v_and_b32",          ", %8")
OP3(k_lshlrev_b32,    "v_lshlrev_b32",      ", %9")          // dst = src1 << src0: acc as the shift count is harmless here
OP3(k_add3_u32,       "v_add3_u32",         ", %8, %9")
OP3(k_bfi_b32,        "v_bfi_b32",          ", %8, %9")
OP3(k_perm_b32,       "v_perm_b32",         ", %8, %9")
OP3(k_mad_u32_u24,    "v_mad_u32_u24",      ", %8, %9")
OP3(k_mad_i32_i24,    "v_mad_i32_i24",      ", %8, %9")
OP3(k_mul_u32_u24,    "v_mul_u32_u24",      ", %8")
OP3(k_mul_hi_u32_u24, "v_mul_hi_u32_u24",   ", %8")
OP3(k_mul_hi_i32_i24, "v_mul_hi_i32_i24",   ", %8")
OP3(k_mul_lo_u32,     "v_mul_lo_u32",       ", %8")
OP3(k_mul_hi_u32,     "v_mul_hi_u32",       ", %8")
OP3(k_mad_i32_i16,    "v_mad_i32_i16",      ", %8, %9")
OP3(k_mad_i32_i16_os, "v_mad_i32_i16",      ", %8, %9 op_sel:[1,0,0,0]")
OP3(k_dot2_i32_i16,   "v_dot2_i32_i16",     ", %8, %9")
OP3(k_pk_add_u16,     "v_pk_add_u16",       ", %8")
OP3(k_pk_sub_i16,     "v_pk_sub_i16",       ", %8")
OP3(k_pk_ashrrev_i16, "v_pk_ashrrev_i16",   ", %9")
OP3(k_pk_mul_lo_u16,  "v_pk_mul_lo_u16",    ", %8")
OP3(k_pk_mad_i16,     "v_pk_mad_i16",       ", %8, %9")
OP3(k_pk_fma_f32x,    "v_fma_f32",          ", %8, %9")
OP3(k_mul_f32,        "v_mul_f32",          ", %8")
OP3(k_cvt_f32_i32,    "v_cvt_f32_i32",      "")
OP3(k_cvt_i32_f32,    "v_cvt_i32_f32",      "")
OP3(k_rcp_f32,        "v_rcp_f32",          "")
OP3(k_add_u32_dpp,    "v_add_u32_dpp",      ", %8 row_shr:1 row_mask:0xf bank_mask:0xf")
OP3(k_mov_dpp_wshr,   "v_mov_b32_dpp",      " wave_shr:1 row_mask:0xf bank_mask:0xf")
OP3(k_cvt_f32_i32_sdwa, "v_cvt_f32_i32_sdwa", " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
OP3(k_bfe_i32,        "v_bfe_i32",          ", 16, 16")
OP3(k_ashrrev_i32,    "v_ashrrev_i32",      ", 16")          // dst = 16 >> acc ... rate only
OP3(k_alignbit,       "v_alignbit_b32",     ", %8, %9")
// the mask in an SGPR pair: with vcc as a long-lived mask the loop measured 0.044 per SIMD-cycle -- an artefact of that loop (tools/cndmask_probe.hip:
// cmp + cndmask through vcc runs at 0.247, a cndmask on an SGPR pair at 0.23), which round 3's table carried
OP3(k_cndmask,        "v_cndmask_b32_e64",  ", %8, s[20:21]")
OP3(k_max_i32,        "v_max_i32",          ", %8")
OP3(k_sub_u32,        "v_sub_u32",          ", %8")

// v_pk_fma_f32 works on 64-bit register pairs: its own kernel
__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned *out, unsigned long long *cyc, unsigned seed)
{
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 * 1.5f, a2 = a0 + 2.f, a3 = a0 * 0.5f, a4 = a0 - 3.f, a5 = a0 * 0.25f, a6 = a0 + 7.f, a7 = a0 * 3.f;
	f2 k0 = {1.0000001f + (float)seed * 1e-9f, 0.9999999f}, k1 = {1e-3f, -1e-3f};
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITERS; it++) {
#define PKF "v_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t" \
            "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t"
		asm volatile(PKF PKF PKF PKF : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k0), "v"(k1));
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	f2 r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (r.x + r.y == 1.2345f) out[blockIdx.x] = 1;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// LDS: ds_read_b128 / ds_read_b32 / ds_write_b32 issue (address per lane = lane * width: conflict-free)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(unsigned *out, unsigned long long *cyc, unsigned seed)
{
	__shared__ unsigned sm[256 * 4 + 64];
	for (int i = threadIdx.x; i < 256 * 4 + 64; i += 256) sm[i] = i * seed;
	__syncthreads();
	unsigned acc = 0;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < ITERS; it++) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			if (MODE == 0) { typedef unsigned u32x4 __attribute__((ext_vector_type(4))); u32x4 v = *reinterpret_cast<volatile u32x4 *>(&sm[threadIdx.x * 4]); acc += v.x ^ v.w; }
			else if (MODE == 1) { acc += *reinterpret_cast<volatile unsigned *>(&sm[threadIdx.x + u]); }
			else { *reinterpret_cast<volatile unsigned *>(&sm[threadIdx.x + u]) = acc + u; }
		}
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if (acc == 0x13579bdfu) out[blockIdx.x] = acc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(unsigned *, unsigned long long *, unsigned);
struct entry { const char *name; kfn fn; int per_iter; };

int main(int argc, char **argv)
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const double clk_ghz = prop.clockRate * 1e-6;
	unsigned *out;
	unsigned long long *cyc;
	CHK(hipMalloc(&out, 1 << 20));
	CHK(hipMalloc(&cyc, (size_t)8 * 8192));
	std::vector<entry> es = {
#define E(n) {#n, n, 32}
		E(k_add_u32), E(k_sub_u32), E(k_and_b32), E(k_lshlrev_b32), E(k_ashrrev_i32), E(k_add3_u32), E(k_bfi_b32), E(k_perm_b32), E(k_alignbit), E(k_bfe_i32),
		E(k_cndmask), E(k_max_i32),
		E(k_mad_u32_u24), E(k_mad_i32_i24), E(k_mul_u32_u24), E(k_mul_hi_u32_u24), E(k_mul_hi_i32_i24), E(k_mul_lo_u32), E(k_mul_hi_u32),
		E(k_mad_i32_i16), E(k_mad_i32_i16_os), E(k_dot2_i32_i16), E(k_pk_add_u16), E(k_pk_sub_i16), E(k_pk_ashrrev_i16), E(k_pk_mul_lo_u16), E(k_pk_mad_i16),
		E(k_pk_fma_f32x), E(k_mul_f32), E(k_pk_fma_f32), E(k_cvt_f32_i32), E(k_cvt_i32_f32), E(k_cvt_f32_i32_sdwa), E(k_rcp_f32),
		E(k_add_u32_dpp), E(k_mov_dpp_wshr),
		{"ds_read_b128", k_lds<0>, 8}, {"ds_read_b32", k_lds<1>, 8}, {"ds_write_b32", k_lds<2>, 8},
	};
	std::string json = "{\"device\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) + ", \"clock_ghz\": " + std::to_string(clk_ghz) +
	                   ", \"unit\": \"wave64 instructions per SIMD-cycle (in-kernel cycle counter)\", \"ops\": {";
	printf("%-22s %8s %8s %8s %8s   (wave-instr / SIMD-cycle at 1,2,4,8 waves per SIMD; cycles from the shader clock)   wall@8\n", "opcode", "w1", "w2", "w4", "w8");
	hipEvent_t ea, eb;
	CHK(hipEventCreate(&ea));
	CHK(hipEventCreate(&eb));
	bool first = true;
	for (auto &e : es) {
		double rate[4], wall8 = 0;
		for (int wi = 0; wi < 4; wi++) {
			const int wps = 1 << wi;                       // waves per SIMD = workgroups of 256 threads per CU
			const int grid = cus * wps;
			hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, cyc, 1u);       // warm
			CHK(hipDeviceSynchronize());
			CHK(hipEventRecord(ea));
			hipLaunchKernelGGL(e.fn, dim3(grid), dim3(256), 0, 0, out, cyc, 2u);
			CHK(hipEventRecord(eb));
			CHK(hipDeviceSynchronize());
			float ms = 0;
			CHK(hipEventElapsedTime(&ms, ea, eb));
			std::vector<unsigned long long> h(grid);
			CHK(hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
			double mean = 0;
			for (auto v : h) mean += (double)v;
			mean /= grid;
			// __builtin_readcyclecounter = s_memtime: counts at the shader clock on gfx9 (one SIMD hosts wps of the timed waves)
			const double instr_per_wave = (double)ITERS * e.per_iter;
			rate[wi] = instr_per_wave * wps / mean;
			if (wi == 3)
				wall8 = instr_per_wave * wps / (ms * 1e-3 * clk_ghz * 1e9);
		}
		printf("%-22s %8.3f %8.3f %8.3f %8.3f   %8.3f\n", e.name, rate[0], rate[1], rate[2], rate[3], wall8);
		char buf[256];
		snprintf(buf, sizeof(buf), "%s\"%s\": {\"w1\": %.4f, \"w2\": %.4f, \"w4\": %.4f, \"w8\": %.4f, \"wall_w8\": %.4f}", first ? "" : ", ", e.name + 2, rate[0], rate[1], rate[2], rate[3], wall8);
		json += buf;
		first = false;
	}
	json += "}}";
	if (argc > 1) {
		FILE *f = fopen(argv[1], "w");
		if (f) { fputs(json.c_str(), f); fputc('\n', f); fclose(f); }
	}
	return 0;
}
