// Micro-benchmark, second set: the VALU instructions stage 1 of the default scan consists of (sp_cylm_scan.h, cylm_bits) and their
// alternatives, at the kernel's four waves per SIMD.  One line per mix: G lane-instructions/s over the whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/valu_bench2 tools/valu_bench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float sa, float sb) {
	float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	float a = sa + threadIdx.x * 1e-7f, b = sb, c = sb * 0.25f;
	unsigned int w0 = threadIdx.x, w1 = w0 * 3u;
	for (int i = 0; i < ITERS; ++i) {
		if (MODE == 0) {        // v_minimum3_f32 with |.| on all sources, 8 chains
#define OP(x) asm volatile("v_minimum3_f32 %0, |%0|, |%1|, |%2|" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 1) { // v_min_f32 (VOP3 for the modifiers), 8 chains
#define OP(x) asm volatile("v_min_f32_e64 %0, |%0|, |%1|" : "+v"(x) : "v"(a));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 2) { // v_alignbit_b32, 2 chains as in the kernel (one per ray block)
#define OP(w, x) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(w) : "v"(x));
			OP(w0, x0) OP(w1, x1) OP(w0, x2) OP(w1, x3) OP(w0, x4) OP(w1, x5) OP(w0, x6) OP(w1, x7)
#undef OP
		} else if (MODE == 3) { // v_fma_f32 with a negated source (VOP3), 8 chains
#define OP(x) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 4) { // v_sub_f32 (VOP2), 8 chains
#define OP(x) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(x) : "v"(c));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 5) { // the kernel's block for two groups: 2 x (minimum3, minimum3, fma, sub, alignbit) = 10 instructions
#define GRP(w, p, q, r, s) asm volatile("v_minimum3_f32 %1, |%1|, |%2|, |%2|\n v_minimum3_f32 %1, %1, |%3|, |%4|\n v_fma_f32 %1, -%5, %6, %1\n v_sub_f32_e32 %1, %1, %7\n v_alignbit_b32 %0, %0, %1, 31" \
	: "+v"(w), "+v"(p) : "v"(q), "v"(r), "v"(s), "v"(a), "v"(b), "v"(c));
			GRP(w0, x0, x1, x2, x3) GRP(w1, x4, x5, x6, x7)
#undef GRP
		} else if (MODE == 6) { // the same with three two-source minima: 2 x (min, min, min, fma, sub, alignbit) = 12 instructions
#define GRP(w, p, q, r, s) asm volatile("v_min_f32_e64 %1, |%1|, |%2|\n v_min_f32_e64 %1, %1, |%3|\n v_min_f32_e64 %1, %1, |%4|\n v_fma_f32 %1, -%5, %6, %1\n v_sub_f32_e32 %1, %1, %7\n v_alignbit_b32 %0, %0, %1, 31" \
	: "+v"(w), "+v"(p) : "v"(q), "v"(r), "v"(s), "v"(a), "v"(b), "v"(c));
			GRP(w0, x0, x1, x2, x3) GRP(w1, x4, x5, x6, x7)
#undef GRP
		} else if (MODE == 7) { // v_min3_f32 with |.| (the pre-gfx950 three-input minimum), 8 chains
#define OP(x) asm volatile("v_min3_f32 %0, |%0|, |%1|, |%2|" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 8) { // v_fmac_f32 (VOP2), 8 chains: the plain-issue reference
#define OP(x) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		} else if (MODE == 9) { // v_lshl_or_b32 (another three-source integer op), 2 chains
#define OP(w, x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(w) : "v"(x));
			OP(w0, x0) OP(w1, x1) OP(w0, x2) OP(w1, x3) OP(w0, x4) OP(w1, x5) OP(w0, x6) OP(w1, x7)
#undef OP
		} else if (MODE == 10) { // v_max_f32 VOP2 plain, 8 chains
#define OP(x) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(x) : "v"(a));
			OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#undef OP
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(w0 ^ w1);
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
	printf("%-52s waves/SIMD=%d  %8.1f G lane-instr/s  (%.3f ms)\n", name, blocks_per_cu, lane_instr / (ms * 1e-3) / 1e9, ms / 5);
	return 0;
}

int main() {
	float* d_out;
	CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * sizeof(float)));
	hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
	printf("device %s %s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
	for (int bpc : {2, 4, 8}) {   // 256-thread blocks per CU = waves per SIMD
		run<8>("v_fmac_f32 (VOP2), 8 chains", 8, bpc, d_out);
		run<0>("v_minimum3_f32 |a|,|b|,|c|", 8, bpc, d_out);
		run<7>("v_min3_f32 |a|,|b|,|c|", 8, bpc, d_out);
		run<1>("v_min_f32_e64 |a|,|b|", 8, bpc, d_out);
		run<10>("v_max_f32_e32", 8, bpc, d_out);
		run<2>("v_alignbit_b32 w, w, x, 31 (2 chains)", 8, bpc, d_out);
		run<9>("v_lshl_or_b32 w, w, 1, x (2 chains)", 8, bpc, d_out);
		run<3>("v_fma_f32 x, -a, b, x", 8, bpc, d_out);
		run<4>("v_sub_f32_e32", 8, bpc, d_out);
		run<5>("stage-1 block: 2 x (minimum3 x2, fma, sub, alignbit)", 10, bpc, d_out);
		run<6>("same with 3 x v_min_f32_e64 per group", 12, bpc, d_out);
	}
	return 0;
}
