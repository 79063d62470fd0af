// Device-side arithmetic of the spath hot path for gfx950.
//
// Everything here is written so that, compiled with -ffp-contract=off, each float operation is
// rounded separately and in the order the reference's C++ evaluates it (x86-64 SSE2, no FMA).
// That is what makes the HIP image bit-comparable with cpu_renderer.  Reference lines are cited
// as file:line relative to the reference's src/ directory.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define SP_DEV __device__ __forceinline__

namespace sp {

struct f3 { float x, y, z; };

SP_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
SP_DEV f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }            // geom.h:38-40
SP_DEV f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }            // geom.h:42-44
SP_DEV f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }            // geom.h:46-48
SP_DEV f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }             // geom.h:62-64
SP_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }            // geom.h:126-128
SP_DEV f3 cross3(f3 a, f3 b) {                                                         // geom.h:143-145
	return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Constants of the reference, as the float/double values its expressions round to
// (checked against the compiled reference by tests/test_oracle_*.py).
constexpr double kPi         = 3.14159265358979323846;                 // geom.h:160 std::acos(-1.0)
constexpr float  kEpsilon    = (float)0.00000000000001;                // geom.h:198
constexpr float  kMaxDist    = (float)1000000000000.0;                 // cpu_renderer.cpp:27
constexpr float  kInvPi      = (float)(1.0 / kPi);                     // cpu_renderer.cpp:63
constexpr float  kP          = (float)(1.0 / (kPi * 2.0));             // cpu_renderer.cpp:60
constexpr float  kInvP       = (float)(1.0 / (double)kP);              // cpu_renderer.cpp:67
// geom.h:217 compares (double)d < 1.0/(double)EPSILON = 100000001754833.03...; that double lies
// strictly between the floats 0x56b5e621 and 0x56b5e622, so for a float d
//   (double)d < 1.0/EPSILON   <=>   d < kInvEpsCeil  (= 0x56b5e622)
constexpr float  kInvEpsCeil = 100000008765440.0f;
static_assert(__builtin_bit_cast(uint32_t, kEpsilon) == 0x283424dcu, "EPSILON");
static_assert(__builtin_bit_cast(uint32_t, kMaxDist) == 0x5368d4a5u, "MAX_VALUE_DIST");
static_assert(__builtin_bit_cast(uint32_t, kInvPi) == 0x3ea2f983u, "1/PI");
static_assert(__builtin_bit_cast(uint32_t, kP) == 0x3e22f983u, "p");
static_assert(__builtin_bit_cast(uint32_t, kInvP) == 0x40c90fdbu, "1/p");
static_assert(__builtin_bit_cast(uint32_t, kInvEpsCeil) == 0x56b5e622u, "1/EPSILON ceiling");
static_assert((double)kInvEpsCeil >= 1.0 / (double)kEpsilon, "ceil above");
static_assert((double)__builtin_bit_cast(float, 0x56b5e621u) < 1.0 / (double)kEpsilon, "floor below");

// ---- 1.0f/a, correctly rounded.  v_rcp_f32 (1 ulp) + one Newton step in FMAs equals the IEEE
// divide for EVERY float with biased exponent in [1, 252] (|a| in [2^-126, 2^126)): checked
// exhaustively on gfx950 against the compiler's divide expansion, which was itself checked against
// the host's divide (tools/rcp_check.hip, profiles/r01_rcp_exhaustive.log).  3 VALU instead of 10.
// Outside that range (denormal a, or a so large that 1/a is denormal, inf, nan) take the full divide.
SP_DEV float recip_ieee(float a) {
	const float r = __builtin_amdgcn_rcpf(a);
	const float e = __builtin_fmaf(-a, r, 1.0f);
	float f = __builtin_fmaf(e, r, r);
	const uint32_t ex = (__float_as_uint(a) >> 23) & 0xffu;
	if (__builtin_expect(ex - 1u > 251u, 0)) f = 1.0f / a;
	return f;
}

// ---- Moeller-Trumbore exactly as geom::ray_intersect evaluates it (geom.h:197-222), branch-free.
// e1 = v1 - v0 and e2 = v2 - v0 are precomputed by the repack pass; each is one float subtraction,
// so they are the same bits the reference computes per test at geom.h:200-201.
// Returns the hit distance, or -1 where the reference returns -1.
SP_DEV float ray_tri_strict(f3 o, f3 dir, f3 v0, f3 e1, f3 e2) {
	const f3 h = cross3(dir, e2);                    // :202
	const float a = dot3(e1, h);                     // :203
	const float f = recip_ieee(a);                   // :206 (double divide rounded to float == IEEE float divide)
	const f3 s = sub3(o, v0);                        // :207
	const float u = f * dot3(s, h);                  // :208
	const f3 q = cross3(s, e1);                      // :211
	const float v = f * dot3(dir, q);                // :212
	const float d = f * dot3(e2, q);                 // :216
	const bool rej_a = (a > -kEpsilon) && (a < kEpsilon);          // :204
	const bool rej_u = (u < 0.0f) || (u > 1.0f);                   // :209
	const bool rej_v = (v < 0.0f) || ((u + v) > 1.0f);             // :213
	const bool acc_d = (d > kEpsilon) && (d < kInvEpsCeil);        // :217
	return (!rej_a && !rej_u && !rej_v && acc_d) ? d : -1.0f;
}

// ---- glibc 2.35 sinf/cosf (ARM optimized-routines sincosf), restated; see oracle/spath_oracle.c for
// the provenance note.  Double precision, separately rounded operations.  Valid for 0 <= y < 120.
SP_DEV float sc_poly(double x, double x2, int n, bool neg) {
	constexpr double C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
	                 C4 = 0x1.99343027bf8c3p-16;
	constexpr double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
	if ((n & 1) == 0) {
		const double x3 = x * x2;
		const double s1 = S2 + x2 * S3;
		const double x7 = x3 * x2;
		const double s = x + x3 * S1;
		return (float)(s + x7 * s1);
	}
	const double sg = neg ? -1.0 : 1.0;
	const double x4 = x2 * x2;
	const double c2 = sg * C3 + x2 * (sg * C4);
	const double c1 = sg * C1 + x2 * (sg * C2);
	const double x6 = x4 * x2;
	const double c = sg + x2 * c1;
	return (float)(c + x6 * c2);
}

SP_DEV uint32_t sc_abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }

SP_DEV void sincos_glibc(float y, float* sn, float* cs) {
	constexpr double HPI_INV = 0x1.45F306DC9C883p+23, HPI = 0x1.921FB54442D18p0;
	double x = (double)y;
	if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {
		if (sc_abstop12(y) < sc_abstop12(0x1p-12f)) { *sn = y; *cs = 1.0f; return; }
		const double x2 = x * x;
		*sn = sc_poly(x, x2, 0, false);
		*cs = sc_poly(x, x2, 1, false);
		return;
	}
	const double r = x * HPI_INV;
	const int n = ((int32_t)r + 0x800000) >> 24;
	x = x - (double)n * HPI;
	const double x2 = x * x;
	const int ns = n & 3, nc = (n + 1) & 3;
	const double sgs = (ns == 1 || ns == 2) ? -1.0 : 1.0;
	const double sgc = (nc == 1 || nc == 2) ? -1.0 : 1.0;
	*sn = sc_poly(x * sgs, x2, n, (n & 2) != 0);
	*cs = sc_poly(x * sgc, x2, n ^ 1, ((n + 1) & 2) != 0);
}

// ---- Philox4x32-10 counter RNG (Salmon et al., SC'11), replaces frand.h on the device.
// counter = (pixel, sample, depth, 'SPTH'), key = 64-bit seed; o[0], o[1] feed the two draws of
// geom::rand_unit_vec (geom.h:168-169) as 24-bit uniforms in [0,1).
SP_DEV void philox_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, double* r1, double* r2) {
	uint32_t c0 = pixel, c1 = sample, c2 = depth, c3 = 0x48545053u;
	uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
	for (int i = 0; i < 10; ++i) {
		const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
		const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
		const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
		c0 = n0; c1 = l1; c2 = n2; c3 = l0;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	*r1 = (double)(c0 >> 8) * (1.0 / 16777216.0);
	*r2 = (double)(c1 >> 8) * (1.0 / 16777216.0);
}

// ---- geom::rand_unit_vec (geom.h:164-177) with the two draws given
SP_DEV f3 rand_unit_vec(f3 n, double r1, double r2) {
	const float rv_xz = (float)(1.0 * r1 * kPi * 2.0);   // :168
	const float rv_y  = (float)(1.0 * r2 * kPi * 0.5);   // :169
	float f_x, f_y, cxz, sxz;
	sincos_glibc(rv_y, &f_y, &f_x);                      // :170-171
	sincos_glibc(rv_xz, &sxz, &cxz);
	const f3 out = mk3(cxz * f_x, f_y, sxz * f_x);       // :173
	return (dot3(n, out) < 0.0f) ? scale3(out, -1.0f) : out;  // :174-176
}

// ---- vec3::clamp (geom.h:151-157) and scene::vec3_RGBA (scene.h:32-39)
SP_DEV float clamp01(float x) { return (x > 1.0f) ? 1.0f : ((x < 0.0f) ? 0.0f : x); }

SP_DEV uint32_t quant8(float x) {
	const float c = clamp01(x) * 255.0f + 0.5f;
	return (c < 0.0f) ? 0u : ((c > 255.0f) ? 255u : ((uint32_t)c & 0xffu));
}

SP_DEV uint32_t vec3_rgba(f3 v) {   // r | g<<8 | b<<16 | a(=0)<<24
	return quant8(v.x) | (quant8(v.y) << 8) | (quant8(v.z) << 16);
}

} // namespace sp
