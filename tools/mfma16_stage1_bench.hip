// Micro-benchmark: stage 1 of the cylinder-filter scan (sp_cyl_scan.h: x = |gm| - H*D, gm a 6-term side product) with gm on the
// FP16 matrix pipe.  Every float operand v is split as v = hi + lo (two halves; |v - hi - lo| <= 2^-22 |v|), a product a*b becomes
// a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (each exact in the f32 accumulator), and the 6 products of gm fill 18 of the K = 32 slots
// of two v_mfma_f32_32x32x16_f16 -- 32 triangles (A rows) x 32 rays (B columns) per pair of instructions, 1024 side products.
// The VALU is left with x = fma(-H, D, |gm|), the min over a group of four triangles and the sign bit: 2 instructions per pair
// instead of 7.06.
//   mode 0: the production hot loop (4 rays per lane, six ds_read_b128 per group of 4 triangles, 113 VALU per 16 x 64 pairs)
//   mode 1: the MFMA form (1 ray per lane; A fragments and H from LDS; 2 MFMA + 32 VALU per 1024 pairs)
// Both loop over one LDS-resident tile; the figure of merit is pairs per second per chip at 4 waves per SIMD, and what the
// board's clock does meanwhile.  Results are checksummed (bit words) so that nothing is optimised away; mode 1's survivors are
// compared with mode 0's on the same data (they may differ only within the margin).
//
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o build/mfma16_stage1_bench tools/mfma16_stage1_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kTile = 128;                 // triangles per LDS tile (4 blocks of 32)
constexpr int kGroups = kTile / 4;

// ---- mode 0: production form.  Tile: chunk-major groups, 8 float4 per group (sp_cyl_scan.h)
__device__ __forceinline__ float cyl_x(const float4 q0, float mz, float H, float Pa, float Pb, float Pc, float ndx, float ndy, float ndz, float D) {
	float gm = __builtin_fmaf(q0.x, Pb, Pa);
	gm = __builtin_fmaf(q0.y, Pc, gm);
	gm = __builtin_fmaf(ndx, q0.z, gm);
	gm = __builtin_fmaf(ndy, q0.w, gm);
	gm = __builtin_fmaf(ndz, mz, gm);
	return __builtin_fmaf(-H, D, __builtin_fabsf(gm));
}

__global__ void __launch_bounds__(256, 4) k_valu(const float4* __restrict__ tile_g, const float* __restrict__ rays, int iters, float eps, unsigned long long* out) {
	__shared__ float4 sm[kGroups * 8];
	for (int i = threadIdx.x; i < kGroups * 8; i += 256) sm[i] = tile_g[i];
	__syncthreads();
	float Pa[4], Pb[4], Pc[4], nx[4], ny[4], nz[4], D[4], Dq[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		const float* p = rays + ((size_t)(blockIdx.x * 1024 + r * 256 + threadIdx.x)) * 8;
		Pa[r] = p[0]; Pb[r] = p[1]; Pc[r] = p[2]; nx[r] = p[3]; ny[r] = p[4]; nz[r] = p[5]; D[r] = p[6]; Dq[r] = p[7];
	}
	uint32_t w[4] = {0, 0, 0, 0};
	unsigned long long cnt = 0;
	for (int it = 0; it < iters; ++it) {
		for (int g = 0; g < kGroups; ++g) {
			float4 q[4]; float4 m01 = sm[4 * kGroups + g], m23 = sm[5 * kGroups + g];
#pragma unroll
			for (int u = 0; u < 4; ++u) q[u] = sm[u * kGroups + g];
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const float x0 = cyl_x(q[0], m01.x, m01.y, Pa[r], Pb[r], Pc[r], nx[r], ny[r], nz[r], D[r]);
				const float x1 = cyl_x(q[1], m01.z, m01.w, Pa[r], Pb[r], Pc[r], nx[r], ny[r], nz[r], D[r]);
				const float x2 = cyl_x(q[2], m23.x, m23.y, Pa[r], Pb[r], Pc[r], nx[r], ny[r], nz[r], D[r]);
				const float x3 = cyl_x(q[3], m23.z, m23.w, Pa[r], Pb[r], Pc[r], nx[r], ny[r], nz[r], D[r]);
				const float m = __builtin_fminf(__builtin_fminf(x0, x1), __builtin_fminf(x2, x3));
				w[r] = __builtin_amdgcn_alignbit(w[r], __float_as_uint(m - Dq[r]), 31);
			}
		}
#pragma unroll
		for (int r = 0; r < 4; ++r) { cnt += __builtin_popcount(w[r]); Pa[r] += eps * (float)(w[r] & 1u); }
	}
	atomicAdd(out, cnt);
}

// ---- mode 1: MFMA form.  LDS: A fragments [tb][kk][lane] 16 B (8 halves), H [tb][h][16] floats
__global__ void __launch_bounds__(256, 4) k_mfma(const h8* __restrict__ afrag_g, const float* __restrict__ h_g, const float* __restrict__ rays, int iters, float eps, unsigned long long* out) {
	__shared__ h8 sa[(kTile / 32) * 2 * 64];
	__shared__ float sh[(kTile / 32) * 2 * 16];
	for (int i = threadIdx.x; i < (kTile / 32) * 2 * 64; i += 256) sa[i] = afrag_g[i];
	for (int i = threadIdx.x; i < (kTile / 32) * 2 * 16; i += 256) sh[i] = h_g[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, hh = lane >> 5;
	// the lane's own ray: 6 values -> 18 K slots: (P_a, 1) needs (hi, lo) x 1; the five products hi*hi, hi*lo, lo*hi
	//   B slots k: [Pa_hi, Pa_lo, Pb_hi, Pb_hi, Pb_lo, Pc_hi, Pc_hi, Pc_lo | nx_hi, nx_hi, nx_lo, ny_hi, ny_hi, ny_lo, nz_hi, nz_hi | nz_lo, 0...]
	//   A slots k: [1,     1,     b_hi,  b_lo,  b_hi,  c_hi,  c_lo,  c_hi  | Mx_hi, Mx_lo, Mx_hi, My_hi, My_lo, My_hi, Mz_hi, Mz_lo | Mz_hi, 0...]
	const float* p = rays + ((size_t)(blockIdx.x * 256 + threadIdx.x)) * 8;
	float v[6]; for (int i = 0; i < 6; ++i) v[i] = p[i];
	const float D = p[6], Dq = p[7];
	_Float16 hi[6], lo[6];
#pragma unroll
	for (int i = 0; i < 6; ++i) { hi[i] = (_Float16)v[i]; lo[i] = (_Float16)(v[i] - (float)hi[i]); }
	const _Float16 z = (_Float16)0.0f;
	h8 k0 = { hi[0], lo[0], hi[1], hi[1], lo[1], hi[2], hi[2], lo[2] };
	h8 k1 = { hi[3], hi[3], lo[3], hi[4], hi[4], lo[4], hi[5], hi[5] };
	h8 k2 = { lo[5], z, z, z, z, z, z, z };
	h8 k3 = { z, z, z, z, z, z, z, z };
	// B fragment of ray block rb, MFMA kk: lane l supplies column l&31 = ray 32*rb + (l&31), k = 8h + j of that MFMA's 16 slots.
	// The K vector of a ray is (k0 k1 | k2 k3); lane l needs, of ray 32*rb + (l&31), part (2*kk + h).  Own ray when (l>>5) == rb.
	h8 bfr[2][2];
#pragma unroll
	for (int rb = 0; rb < 2; ++rb)
#pragma unroll
		for (int kk = 0; kk < 2; ++kk) {
			const int src = (int)((lane & 31u) + 32u * rb);
			h8 part;
			typedef int i4 __attribute__((ext_vector_type(4)));
			const h8 lo_part = kk == 0 ? k0 : k2, hi_part = kk == 0 ? k1 : k3;
			i4 a = __builtin_bit_cast(i4, lo_part), b = __builtin_bit_cast(i4, hi_part), o;
#pragma unroll
			for (int c = 0; c < 4; ++c) { const int va = __shfl(a[c], src, 64), vb = __shfl(b[c], src, 64); o[c] = hh ? vb : va; }
			part = __builtin_bit_cast(h8, o);
			bfr[rb][kk] = part;
		}
	const float Dn[2] = { __shfl(D, (int)(lane & 31u), 64), __shfl(D, (int)(lane & 31u) + 32, 64) };
	const float Dqn[2] = { __shfl(Dq, (int)(lane & 31u), 64), __shfl(Dq, (int)(lane & 31u) + 32, 64) };
	uint32_t w[2] = {0, 0};
	unsigned long long cnt = 0;
	float bump = 0.0f;
	for (int it = 0; it < iters; ++it) {
		__asm__ volatile("" ::: "memory");          // the tile changes in the real kernel: re-read LDS every pass
#pragma unroll 1
		for (int tb = 0; tb < kTile / 32; ++tb) {
			const h8 a0 = sa[(tb * 2 + 0) * 64 + lane], a1 = sa[(tb * 2 + 1) * 64 + lane];
			const float4* hp = (const float4*)(sh + (tb * 2 + hh) * 16);
			const float4 H0 = hp[0], H1 = hp[1], H2 = hp[2], H3 = hp[3];
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				f16v acc;
#pragma unroll
				for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
				acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bfr[rb][0], acc, 0, 0, 0);
				acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bfr[rb][1], acc, 0, 0, 0);
				const float Dr = Dn[rb] + bump, Dqr = Dqn[rb];
#define GRP(J, HV) { \
				const float x0 = __builtin_fmaf(-HV.x, Dr, __builtin_fabsf(acc[4 * J + 0])), x1 = __builtin_fmaf(-HV.y, Dr, __builtin_fabsf(acc[4 * J + 1])); \
				const float x2 = __builtin_fmaf(-HV.z, Dr, __builtin_fabsf(acc[4 * J + 2])), x3 = __builtin_fmaf(-HV.w, Dr, __builtin_fabsf(acc[4 * J + 3])); \
				const float m = __builtin_fminf(__builtin_fminf(x0, x1), __builtin_fminf(x2, x3)); \
				w[rb] = __builtin_amdgcn_alignbit(w[rb], __float_as_uint(m - Dqr), 31); }
				GRP(0, H0) GRP(1, H1) GRP(2, H2) GRP(3, H3)
#undef GRP
			}
		}
		cnt += __builtin_popcount(w[0] & 0xffffu) + __builtin_popcount(w[1] & 0xffffu);
		bump += eps * (float)(w[0] & 1u);
	}
	atomicAdd(out, cnt);
}

// ---- mode 2: ONE MFMA per 32 x 32 pairs: the 15 cross products fill K = 16, P_a enters through the accumulator input (C = P_a of the
// lane's column ray in all 16 registers).  A fragments [tb][lane] 16 B, H [tb][h][16] floats
template <int PART>
__global__ void __launch_bounds__(256, 4) k_mfma1(const h8* __restrict__ afrag_g, const float* __restrict__ h_g, const float* __restrict__ rays, int iters, float eps, unsigned long long* out) {
	__shared__ h8 sa[(kTile / 32) * 64];
	__shared__ float sh[(kTile / 32) * 2 * 16];
	for (int i = threadIdx.x; i < (kTile / 32) * 64; i += 256) sa[i] = afrag_g[i];
	for (int i = threadIdx.x; i < (kTile / 32) * 2 * 16; i += 256) sh[i] = h_g[i];
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u, hh = lane >> 5;
	const float* p = rays + ((size_t)(blockIdx.x * 256 + threadIdx.x)) * 8;
	float v[6]; for (int i = 0; i < 6; ++i) v[i] = p[i];
	const float D = p[6], Dq = p[7];
	_Float16 hi[6], lo[6];
#pragma unroll
	for (int i = 1; i < 6; ++i) { hi[i] = (_Float16)v[i]; lo[i] = (_Float16)(v[i] - (float)hi[i]); }
	const _Float16 z = (_Float16)0.0f;
	//   B slots k: [Pb_hi, Pb_hi, Pb_lo, Pc_hi, Pc_hi, Pc_lo, nx_hi, nx_hi | nx_lo, ny_hi, ny_hi, ny_lo, nz_hi, nz_hi, nz_lo, 0]
	//   A slots k: [b_hi,  b_lo,  b_hi,  c_hi,  c_lo,  c_hi,  Mx_hi, Mx_lo | Mx_hi, My_hi, My_lo, My_hi, Mz_hi, Mz_lo, Mz_hi, 0]
	h8 k0 = { hi[1], hi[1], lo[1], hi[2], hi[2], lo[2], hi[3], hi[3] };
	h8 k1 = { lo[3], hi[4], hi[4], lo[4], hi[5], hi[5], lo[5], z };
	h8 bfr[2]; f16v cin[2];
	typedef int i4 __attribute__((ext_vector_type(4)));
#pragma unroll
	for (int rb = 0; rb < 2; ++rb) {
		const int src = (int)((lane & 31u) + 32u * rb);
		i4 a = __builtin_bit_cast(i4, k0), b = __builtin_bit_cast(i4, k1), o;
#pragma unroll
		for (int c = 0; c < 4; ++c) { const int va = __shfl(a[c], src, 64), vb = __shfl(b[c], src, 64); o[c] = hh ? vb : va; }
		bfr[rb] = __builtin_bit_cast(h8, o);
		const float pa = __shfl(v[0], src, 64);
#pragma unroll
		for (int i = 0; i < 16; ++i) cin[rb][i] = pa;
	}
	const float Dn[2] = { __shfl(D, (int)(lane & 31u), 64), __shfl(D, (int)(lane & 31u) + 32, 64) };
	const float Dqn[2] = { __shfl(Dq, (int)(lane & 31u), 64), __shfl(Dq, (int)(lane & 31u) + 32, 64) };
	uint32_t w[2] = {0, 0};
	unsigned long long cnt = 0;
	float bump = 0.0f;
	for (int it = 0; it < iters; ++it) {
		__asm__ volatile("" ::: "memory");
#pragma unroll 1
		for (int tb = 0; tb < kTile / 32; ++tb) {
			const h8 a0 = sa[tb * 64 + lane];
			const float4* hp = (const float4*)(sh + (tb * 2 + hh) * 16);
			const float4 H0 = hp[0], H1 = hp[1], H2 = hp[2], H3 = hp[3];
			f16v acc[2];
			if (PART != 2) {
				acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bfr[0], cin[0], 0, 0, 0);
				acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bfr[1], cin[1], 0, 0, 0);
			} else {                                                  // VALU only: something that depends on the loads, no matrix instruction
#pragma unroll
				for (int i = 0; i < 16; ++i) { acc[0][i] = cin[0][i] + H0.x * (float)i; acc[1][i] = cin[1][i] + H1.y * (float)i; }
			}
			if (PART == 1) {                                          // MFMA only: one cheap use of every result
				uint32_t x0 = 0, x1 = 0;
#pragma unroll
				for (int i = 0; i < 16; ++i) { x0 |= __float_as_uint(acc[0][i]); x1 |= __float_as_uint(acc[1][i]); }
				w[0] ^= x0 ^ __float_as_uint(H0.x + H1.x + H2.x + H3.x); w[1] ^= x1;
				continue;
			}
#pragma unroll
			for (int rb = 0; rb < 2; ++rb) {
				const float Dr = Dn[rb] + bump, Dqr = Dqn[rb];
#define GRP(J, HV) { \
				const float x0 = __builtin_fmaf(-HV.x, Dr, __builtin_fabsf(acc[rb][4 * J + 0])), x1 = __builtin_fmaf(-HV.y, Dr, __builtin_fabsf(acc[rb][4 * J + 1])); \
				const float x2 = __builtin_fmaf(-HV.z, Dr, __builtin_fabsf(acc[rb][4 * J + 2])), x3 = __builtin_fmaf(-HV.w, Dr, __builtin_fabsf(acc[rb][4 * J + 3])); \
				const float m = __builtin_fminf(__builtin_fminf(x0, x1), __builtin_fminf(x2, x3)); \
				w[rb] = __builtin_amdgcn_alignbit(w[rb], __float_as_uint(m - Dqr), 31); }
				GRP(0, H0) GRP(1, H1) GRP(2, H2) GRP(3, H3)
#undef GRP
			}
		}
		cnt += __builtin_popcount(w[0] & 0xffffu) + __builtin_popcount(w[1] & 0xffffu);
		bump += eps * (float)(w[0] & 1u);
	}
	atomicAdd(out, cnt);
}

static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; std::memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }

int main(int argc, char** argv) {
	const int iters = argc > 1 ? atoi(argv[1]) : 2000;
	const int blocks = 256 * 4 * 2;
	srand(7);
	auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
	// records: beta, gamma in [-1, 1], Mc in [-1, 1], H small
	std::vector<float> rec(kTile * 6), Hh(kTile);
	for (int t = 0; t < kTile; ++t) { for (int i = 0; i < 5; ++i) rec[t * 6 + i] = rnd(); Hh[t] = 0.01f + 0.02f * fabsf(rnd()); }
	// mode 0 tile
	std::vector<float4> tile0(kGroups * 8);
	for (int g = 0; g < kGroups; ++g) {
		for (int u = 0; u < 4; ++u) { const float* r = &rec[(4 * g + u) * 6]; tile0[u * kGroups + g] = make_float4(r[0], r[1], r[2], r[3]); }
		tile0[4 * kGroups + g] = make_float4(rec[(4 * g) * 6 + 4], Hh[4 * g], rec[(4 * g + 1) * 6 + 4], Hh[4 * g + 1]);
		tile0[5 * kGroups + g] = make_float4(rec[(4 * g + 2) * 6 + 4], Hh[4 * g + 2], rec[(4 * g + 3) * 6 + 4], Hh[4 * g + 3]);
		tile0[6 * kGroups + g] = tile0[7 * kGroups + g] = make_float4(0, 0, 0, 0);
	}
	// mode 1 fragments: A slots (see kernel comment); lane l: row l&31, k = 8h + j
	std::vector<uint16_t> afrag((kTile / 32) * 2 * 64 * 8);
	std::vector<float> hfrag((kTile / 32) * 2 * 16);
	for (int tb = 0; tb < kTile / 32; ++tb) {
		for (int row = 0; row < 32; ++row) {
			const float* r = &rec[(tb * 32 + row) * 6];
			uint16_t hi[5], lo[5];
			for (int i = 0; i < 5; ++i) { hi[i] = f2h(r[i]); lo[i] = f2h(r[i] - h2f(hi[i])); }
			const uint16_t one = f2h(1.0f), z = 0;
			const uint16_t K[32] = { one, one, hi[0], lo[0], hi[0], hi[1], lo[1], hi[1],  hi[2], lo[2], hi[2], hi[3], lo[3], hi[3], hi[4], lo[4],
			                         hi[4], z, z, z, z, z, z, z,  z, z, z, z, z, z, z, z };
			for (int kk = 0; kk < 2; ++kk) for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j)
				afrag[(((size_t)(tb * 2 + kk) * 64) + h * 32 + row) * 8 + j] = K[kk * 16 + h * 8 + j];
		}
		for (int h = 0; h < 2; ++h) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) hfrag[(tb * 2 + h) * 16 + 4 * j + i] = Hh[tb * 32 + 8 * j + 4 * h + i];
	}
	std::vector<uint16_t> afrag1((kTile / 32) * 64 * 8);
	for (int tb = 0; tb < kTile / 32; ++tb)
		for (int row = 0; row < 32; ++row) {
			const float* r = &rec[(tb * 32 + row) * 6];
			uint16_t hi[5], lo[5];
			for (int i = 0; i < 5; ++i) { hi[i] = f2h(r[i]); lo[i] = f2h(r[i] - h2f(hi[i])); }
			const uint16_t K[16] = { hi[0], lo[0], hi[0], hi[1], lo[1], hi[1], hi[2], lo[2],  hi[2], hi[3], lo[3], hi[3], hi[4], lo[4], hi[4], 0 };
			for (int h = 0; h < 2; ++h) for (int j = 0; j < 8; ++j) afrag1[(((size_t)tb * 64) + h * 32 + row) * 8 + j] = K[h * 8 + j];
		}
	h8* d_a1; CHECK(hipMalloc(&d_a1, afrag1.size() * 2));
	CHECK(hipMemcpy(d_a1, afrag1.data(), afrag1.size() * 2, hipMemcpyHostToDevice));
	const size_t n_rays = (size_t)blocks * 1024;
	std::vector<float> rays(n_rays * 8);
	for (size_t i = 0; i < n_rays; ++i) { for (int k = 0; k < 6; ++k) rays[i * 8 + k] = rnd(); rays[i * 8 + 6] = 1.0f; rays[i * 8 + 7] = 1e-4f; }
	float4* d_t0; h8* d_a; float* d_h; float* d_r; unsigned long long* d_o;
	CHECK(hipMalloc(&d_t0, tile0.size() * 16)); CHECK(hipMalloc(&d_a, afrag.size() * 2)); CHECK(hipMalloc(&d_h, hfrag.size() * 4));
	CHECK(hipMalloc(&d_r, rays.size() * 4)); CHECK(hipMalloc(&d_o, 16));
	CHECK(hipMemcpy(d_t0, tile0.data(), tile0.size() * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_a, afrag.data(), afrag.size() * 2, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_h, hfrag.data(), hfrag.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_r, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int mode = 0; mode < 5; ++mode) {
		for (int rep = 0; rep < 3; ++rep) {
			CHECK(hipMemset(d_o, 0, 16));
			CHECK(hipEventRecord(e0));
			if (mode == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, d_t0, d_r, iters, 0.0f, d_o);
			else if (mode == 1) hipLaunchKernelGGL(k_mfma, dim3(blocks * 4), dim3(256), 0, 0, d_a, d_h, d_r, iters, 0.0f, d_o);
			else if (mode == 2) hipLaunchKernelGGL(k_mfma1<0>, dim3(blocks * 4), dim3(256), 0, 0, d_a1, d_h, d_r, iters, 0.0f, d_o);
			else if (mode == 3) hipLaunchKernelGGL(k_mfma1<1>, dim3(blocks * 4), dim3(256), 0, 0, d_a1, d_h, d_r, iters, 0.0f, d_o);
			else           hipLaunchKernelGGL(k_mfma1<2>, dim3(blocks * 4), dim3(256), 0, 0, d_a1, d_h, d_r, iters, 0.0f, d_o);
			CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			unsigned long long cnt; CHECK(hipMemcpy(&cnt, d_o, 8, hipMemcpyDeviceToHost));
			const double pairs = (double)n_rays * kTile * iters;
			printf("mode %d (%s): %.2f ms, %.2f T pairs/s, popcount checksum %llu\n", mode, mode == 4 ? "as mode 2 WITHOUT the matrix instructions (16 adds instead)" : mode == 3 ? "as mode 2 WITHOUT the VALU post-processing (16 ors instead)" : mode == 2 ? "1 MFMA 32x32x16 f16 (C = P_a) + 32 VALU per 1024 pairs, 1 ray/lane" : mode ? "2 MFMA 32x32x16 f16 + 32 VALU per 1024 pairs, 1 ray/lane" : "VALU production form, 4 rays/lane", ms, pairs / ms / 1e9, cnt);
		}
	}
	return 0;
}
