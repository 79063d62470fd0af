// Micro-benchmark: per-instruction VALU issue rates on gfx950 that decide how the closest-hit scan
// should be written (plain vs packed f32, SGPR operands, compares/selects, rcp).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/valu_bench tools/valu_bench.hip
// Output: one line per instruction mix, G lane-instructions/s over the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float sa, float sb) {
	float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	float a = sa + threadIdx.x * 1e-7f, b = sb;
	typedef float v2 __attribute__((ext_vector_type(2)));
	v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
	v2 p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
	for (int i = 0; i < ITERS; ++i) {
		if (MODE == 0) {        // v_fma_f32, VGPR operands
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 1) { // v_pk_fma_f32
#define OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pa), "v"(pb));
			OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
		} else if (MODE == 2) { // v_mul_f32 + v_add_f32 alternating
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3)
#undef OP
		} else if (MODE == 3) { // v_pk_mul_f32 + v_pk_add_f32
#define OP(x) asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(x) : "v"(pa), "v"(pb));
			OP(p0) OP(p1) OP(p2) OP(p3)
#undef OP
		} else if (MODE == 4) { // v_fma_f32 with an SGPR operand
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(sa), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 5) { // v_rcp_f32
#define OP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 6) { // v_cmp + v_cndmask
#define OP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(a), "v"(b) : "vcc");
			OP(x0) OP(x1) OP(x2) OP(x3)
#undef OP
		} else if (MODE == 7) { // v_pk_mul_f32 with SGPR-pair operand
			v2 sp = {sa, sb};
#define OP(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "s"(sp));
			OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
		} else if (MODE == 8) { // v_mul_f32 only
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 9) { // v_min_f32 / v_max_f32 / v_med3
#define OP(x) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 10) { // v_fma_f32, one dependent chain (ILP 1)
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x0) OP(x0) OP(x0) OP(x0) OP(x0) OP(x0) OP(x0)
#undef OP
		} else if (MODE == 11) { // v_fma_f32, two dependent chains (ILP 2)
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x0) OP(x1) OP(x0) OP(x1) OP(x0) OP(x1)
#undef OP
		} else if (MODE == 12) { // v_fmac_f32 (VOP2 encoding), 8 chains
#define OP(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 13) { // v_fmac_f32, two chains
#define OP(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x0) OP(x1) OP(x0) OP(x1) OP(x0) OP(x1)
#undef OP
		} else if (MODE == 14) { // v_sub_f32 with |.| modifiers (VOP3)
#define OP(x) asm volatile("v_sub_f32 %0, |%0|, |%1|" : "+v"(x) : "v"(a));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 15) { // v_cmp_ngt_f32 into an SGPR pair + s_or_b64
			unsigned long long m0, m1;
#define OP(x, m) asm volatile("v_cmp_ngt_f32 %0, %1, %2" : "=s"(m) : "v"(x), "v"(a));
			OP(x0, m0) OP(x1, m1) m0 |= m1; OP(x2, m1) m0 |= m1; OP(x3, m1) m0 |= m1; OP(x4, m1) m0 |= m1; OP(x5, m1) m0 |= m1; OP(x6, m1) m0 |= m1; OP(x7, m1) m0 |= m1;
#undef OP
			if (m0 == 0x123456789ull) x0 += 1.0f;
		} else if (MODE == 16) { // 8 x v_cmp_ngt_f32 into distinct SGPR pairs, no SALU use inside the loop
			unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
#define OP(x, m) asm volatile("v_cmp_ngt_f32 %0, %1, %2" : "=s"(m) : "v"(x), "v"(a));
			OP(x0, m0) OP(x1, m1) OP(x2, m2) OP(x3, m3) OP(x4, m4) OP(x5, m5) OP(x6, m6) OP(x7, m7)
#undef OP
			asm volatile("" :: "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7));
		} else if (MODE == 17) { // v_cmp_ngt_f32 vcc + v_addc_co_u32 (per-lane bit mask), counted as 2 per pair
			unsigned int mask = 0;
#define OP(x) asm volatile("v_cmp_ngt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(x), "v"(a) : "vcc");
			OP(x0) OP(x1) OP(x2) OP(x3)
#undef OP
			if (mask == 0x12345u) x0 += 1.0f;
		} else if (MODE == 18) { // 8 x v_cmp into SGPR pairs + balanced s_or tree (7 s_or, no serial chain)
			unsigned long long m0, m1, m2, m3, m4, m5, m6, m7;
#define OP(x, m) asm volatile("v_cmp_ngt_f32 %0, %1, %2" : "=s"(m) : "v"(x), "v"(a));
			OP(x0, m0) OP(x1, m1) OP(x2, m2) OP(x3, m3) OP(x4, m4) OP(x5, m5) OP(x6, m6) OP(x7, m7)
#undef OP
			const unsigned long long r = ((m0 | m1) | (m2 | m3)) | ((m4 | m5) | (m6 | m7));
			if (r == 0x123456789ull) x0 += 1.0f;
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y
	                                     + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
int run(const char* name, int instr_per_iter, int blocks_per_cu, float* d_out) {
	const int grid = 256 * blocks_per_cu;
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	const double lane_instr = 5.0 * (double)grid * 256.0 * ITERS * instr_per_iter;
	printf("%-34s waves/SIMD=%d  %8.1f G lane-instr/s  (%.3f ms)\n", name, blocks_per_cu, lane_instr / (ms * 1e-3) / 1e9, ms / 5);
	return 0;
}

int main() {
	float* d_out;
	CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * sizeof(float)));
	hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
	printf("device %s %s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
	for (int bpc : {1, 2, 4, 8}) {   // 256-thread blocks per CU = waves per SIMD
		run<0>("v_fma_f32 vgpr", 8, bpc, d_out);
		run<1>("v_pk_fma_f32 (2 lanes-ops/instr)", 8, bpc, d_out);
		run<2>("v_mul_f32+v_add_f32", 8, bpc, d_out);
		run<3>("v_pk_mul_f32+v_pk_add_f32", 8, bpc, d_out);
		run<4>("v_fma_f32 sgpr operand", 8, bpc, d_out);
		run<5>("v_rcp_f32", 8, bpc, d_out);
		run<6>("v_cmp_lt_f32+v_cndmask_b32", 8, bpc, d_out);
		run<7>("v_pk_mul_f32 sgpr-pair operand", 8, bpc, d_out);
		run<8>("v_mul_f32", 8, bpc, d_out);
		run<9>("v_med3_f32", 8, bpc, d_out);
		run<10>("v_fma_f32 one chain", 8, bpc, d_out);
		run<11>("v_fma_f32 two chains", 8, bpc, d_out);
		run<12>("v_fmac_f32 (VOP2) 8 chains", 8, bpc, d_out);
		run<13>("v_fmac_f32 (VOP2) two chains", 8, bpc, d_out);
		run<14>("v_sub_f32 |a|,|b| (VOP3)", 8, bpc, d_out);
		run<15>("v_cmp_ngt_f32 -> sgpr + s_or_b64", 8, bpc, d_out);
		run<16>("v_cmp_ngt_f32 -> 8 sgpr pairs", 8, bpc, d_out);
		run<17>("v_cmp vcc + v_addc_co_u32", 8, bpc, d_out);
		run<18>("v_cmp -> sgpr + s_or tree", 8, bpc, d_out);
	}
	return 0;
}
