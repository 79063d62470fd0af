/* TEST INFRASTRUCTURE ONLY -- the parity oracle, never the product path.
 *
 * Plain-C restatement of the reference CPU path tracer.  See spath_oracle.h.
 * Build: gcc -O3 -ffp-contract=off (oracle/Makefile).  All `real` arithmetic of
 * the reference is float (geom.h:24); the places where the reference evaluates
 * in double are spelled out with casts below.
 *
 * Pinned bit-exact against oracle/_ref/spath_ref (the compiled reference) by
 * tests/test_oracle_vs_reference.py and against tests/golden/ fixtures by
 * tests/test_oracle_golden.py.
 */
#include "spath_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec3 (geom.h:27-158) */
static inline spo_vec3 v3(float x, float y, float z) { spo_vec3 r = { x, y, z }; return r; }
static inline spo_vec3 v_add(spo_vec3 a, spo_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }      /* :38-40 */
static inline spo_vec3 v_sub(spo_vec3 a, spo_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }      /* :42-44 */
static inline spo_vec3 v_mul(spo_vec3 a, spo_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }      /* :46-48 */
static inline spo_vec3 v_adds(spo_vec3 a, float s) { return v3(a.x + s, a.y + s, a.z + s); }              /* :54-56 */
static inline spo_vec3 v_muls(spo_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }              /* :62-64 */
static inline spo_vec3 v_divs(spo_vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }              /* :66-68 */
static inline float v_dot(spo_vec3 a, spo_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           /* :126-128 */
static inline spo_vec3 v_cross(spo_vec3 a, spo_vec3 b) {                                                 /* :143-145 */
	return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline spo_vec3 v_unit(spo_vec3 a) {                                                              /* :130-141 */
	const float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
	return v_divs(a, l);
}
static inline float clamp1(float x, float mn, float mx) { return (x > mx) ? mx : ((x < mn) ? mn : x); }
static inline spo_vec3 v_clamp01(spo_vec3 a) { return v3(clamp1(a.x, 0.0f, 1.0f), clamp1(a.y, 0.0f, 1.0f), clamp1(a.z, 0.0f, 1.0f)); } /* :151-157 */

/* geom.h:160  const double PI = std::acos(-1.0) */
static const double SPO_PI = 3.14159265358979323846;

/* ------------------------------------------------------------------ shared sincos
 * The reference calls the std::cos/std::sin float overloads, i.e. glibc's cosf/sinf, on angles
 * in [0, 2*pi] (geom.h:168-173).  That algorithm lives in a third-party dependency that is not
 * part of /root/reference: GNU libc 2.35 (this image's libm), sysdeps/ieee754/flt-32/
 * {s_sinf.c,s_cosf.c,sincosf.h,s_sincosf_data.c} -- the ARM "optimized routines" sincosf by
 * Szabolcs Nagy: double-precision argument reduction by pi/2 and two short double polynomials,
 * result rounded once to float.  It is NOT correctly rounded (about 0.1 % of arguments differ
 * from the exact rounding by one ulp), so the published algorithm is restated here with its
 * published coefficients rather than replaced by a better one.  The GPU has no glibc, so the
 * oracle and the HIP kernel share this evaluation (separately rounded * and +, no FMA).
 * Pinned: tests/test_oracle_kat.py (test_sincos_matches_this_libm_on_dense_sample, test_sincos_matches_reference_libm_on_every_lcg_angle)
 * checks it against libm on a dense sample and, through
 * the _ref KAT hash, on every angle the reference's 15-bit LCG can produce; an exhaustive
 * sweep of all 1,090,519,041 floats in [0, 8] found 0 mismatches against glibc 2.35
 * (oracle/sincos_exhaustive.c, run once, takes ~10 s).
 * Domain handled: 0 <= x < 120 (the fast-reduction range); the renderer only needs [0, 2*pi].
 */
static const double SC_HPI_INV = 0x1.45F306DC9C883p+23;   /* 2/pi * 2^24 */
static const double SC_HPI     = 0x1.921FB54442D18p0;     /* pi/2 */
static const double SC_C1 = -0x1.ffffffd0c621cp-2, SC_C2 = 0x1.55553e1068f19p-5,
                    SC_C3 = -0x1.6c087e89a359dp-10, SC_C4 = 0x1.99343027bf8c3p-16;
static const double SC_S1 = -0x1.555545995a603p-3, SC_S2 = 0x1.1107605230bc4p-7,
                    SC_S3 = -0x1.994eb3774cf24p-13;

static inline uint32_t sc_abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }

/* sinf_poly of sincosf.h: n even -> sine polynomial on x, n odd -> cosine polynomial;
 * `neg` selects the sign-flipped cosine coefficients of the second table entry */
static inline float sc_poly(double x, double x2, int n, int neg) {
	if ((n & 1) == 0) {
		const double x3 = x * x2;
		const double s1 = SC_S2 + x2 * SC_S3;
		const double x7 = x3 * x2;
		const double s = x + x3 * SC_S1;
		return (float)(s + x7 * s1);
	} else {
		const double sg = neg ? -1.0 : 1.0;
		const double x4 = x2 * x2;
		const double c2 = sg * SC_C3 + x2 * (sg * SC_C4);
		const double c1 = sg * SC_C1 + x2 * (sg * SC_C2);
		const double x6 = x4 * x2;
		const double c = sg * 1.0 + x2 * c1;
		return (float)(c + x6 * c2);
	}
}

static inline double sc_reduce(double x, int* np) {
	const double r = x * SC_HPI_INV;
	const int n = ((int32_t)r + 0x800000) >> 24;
	*np = n;
	return x - (double)n * SC_HPI;
}

static const double SC_SIGN[4] = { 1.0, -1.0, -1.0, 1.0 };

float spo_sinf(float y) {
	double x = (double)y;
	if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {            /* |y| < pi/4 */
		if (sc_abstop12(y) < sc_abstop12(0x1p-12f)) return y;
		return sc_poly(x, x * x, 0, 0);
	}
	int n;
	x = sc_reduce(x, &n);
	return sc_poly(x * SC_SIGN[n & 3], x * x, n, (n & 2) != 0);
}

float spo_cosf(float y) {
	double x = (double)y;
	if (sc_abstop12(y) < sc_abstop12(0x1.921FB6p-1f)) {
		if (sc_abstop12(y) < sc_abstop12(0x1p-12f)) return 1.0f;
		return sc_poly(x, x * x, 1, 0);
	}
	int n;
	x = sc_reduce(x, &n);
	return sc_poly(x * SC_SIGN[(n + 1) & 3], x * x, n ^ 1, ((n + 1) & 2) != 0);
}

/* ------------------------------------------------------------------ RNGs */
/* frand.h:59-62 */
double spo_seed_dist_next(uint32_t* state) {
	*state = (214013u * (*state) + 2531011u);
	return 1.0 * ((*state >> 16) & 0x7FFF) / 32767.0;
}

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11).
 * Not in the reference: it replaces frand.h on the GPU path (north_star: per-lane counter RNG). */
void spo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
	uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
	for (int i = 0; i < 10; ++i) {
		const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
		const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
		const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* the two uniforms one surface hit consumes (geom.h:168-169 draws rv_xz first, then rv_y).
 * counter = (pixel, sample, depth, 'SPTH'), key = 64-bit seed; 24-bit mantissas in [0,1). */
void spo_counter_uniforms(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t depth, double* r1, double* r2) {
	const uint32_t ctr[4] = { pixel, sample, depth, 0x48545053u };
	const uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
	uint32_t o[4];
	spo_philox4x32_10(ctr, key, o);
	*r1 = (double)(o[0] >> 8) * (1.0 / 16777216.0);
	*r2 = (double)(o[1] >> 8) * (1.0 / 16777216.0);
}

typedef struct {
	int kind;
	uint32_t lcg;                       /* SEED_DIST state */
	uint64_t seed; uint32_t pixel, sample, depth; int draw;  /* COUNTER context */
	double pending;
} spo_rng;

static inline double rng_next(spo_rng* g) {
	switch (g->kind) {
	case SPO_RNG_SEED_DIST: return spo_seed_dist_next(&g->lcg);
	case SPO_RNG_STD_RAND:  return 1.0 * rand() / RAND_MAX;             /* frand.h:27-29 */
	default: {
		if (g->draw == 0) {
			double a, b;
			spo_counter_uniforms(g->seed, g->pixel, g->sample, g->depth, &a, &b);
			g->pending = b; g->draw = 1;
			return a;
		}
		g->draw = 0;
		return g->pending;
	}
	}
}

/* ------------------------------------------------------------------ geometry */
/* geom.h:192-195 */
void spo_flat_normal(spo_tri* t) {
	const spo_vec3 dir = v_cross(v_sub(t->v1, t->v0), v_sub(t->v2, t->v0));
	t->n = v_unit(dir);
}

/* geom.h:197-222 */
float spo_ray_intersect(const spo_ray* r, const spo_tri* t, spo_vec3* point) {
	static const float EPSILON = 0.00000000000001;                       /* :198 */
	const spo_vec3 edge1 = v_sub(t->v1, t->v0), edge2 = v_sub(t->v2, t->v0), h = v_cross(r->dir, edge2); /* :200-202 */
	const float a = v_dot(edge1, h);                                     /* :203 */
	if (a > -EPSILON && a < EPSILON) return -1.0f;                       /* :204-205 */
	const float f = (float)(1.0 / (double)a);                            /* :206 double divide, rounded to real */
	const spo_vec3 s = v_sub(r->pos, t->v0);                             /* :207 */
	const float u = f * v_dot(s, h);                                     /* :208 */
	if ((double)u < 0.0 || (double)u > 1.0) return -1.0f;                /* :209-210 */
	const spo_vec3 q = v_cross(s, edge1);                                /* :211 */
	const float v = f * v_dot(r->dir, q);                                /* :212 */
	if ((double)v < 0.0 || (double)(u + v) > 1.0) return -1.0f;          /* :213-214 */
	const float d = f * v_dot(edge2, q);                                 /* :216 */
	if (d > EPSILON && (double)d < 1.0 / (double)EPSILON) {              /* :217 */
		*point = v_add(r->pos, v_muls(r->dir, d));                       /* :218 */
		return d;
	}
	return -1.0f;
}

/* geom.h:164-177 with the two uniform draws passed in (r1 drawn first) */
spo_vec3 spo_rand_unit_vec_from(spo_vec3 in, double r1, double r2) {
	const float rv_xz = (float)(1.0 * r1 * SPO_PI * 2.0);                /* :168 */
	const float rv_y  = (float)(1.0 * r2 * SPO_PI * 0.5);                /* :169 */
	const float f_x = spo_cosf(rv_y), f_y = spo_sinf(rv_y);              /* :170-171 */
	const spo_vec3 out = v3(spo_cosf(rv_xz) * f_x, f_y, spo_sinf(rv_xz) * f_x); /* :173 */
	if ((double)v_dot(in, out) < 0.0) return v_muls(out, -1.0f);         /* :174-175 */
	return out;
}

/* scene.h:32-39 */
spo_rgba spo_vec3_rgba(spo_vec3 in) {
	const spo_vec3 c = v_adds(v_muls(v_clamp01(in), 255.0f), 0.5f);
	spo_rgba o;
	o.r = (c.x < 0.0f) ? 0 : ((c.x > 255.0f) ? 255 : (uint8_t)c.x);
	o.g = (c.y < 0.0f) ? 0 : ((c.y > 255.0f) ? 255 : (uint8_t)c.y);
	o.b = (c.z < 0.0f) ? 0 : ((c.z > 255.0f) ? 255 : (uint8_t)c.z);
	o.a = 0;
	return o;
}

void spo_consts(float out[5], double* inv_eps) {
	const float p = (float)(1.0 / (SPO_PI * 2.0));
	out[0] = p;
	out[1] = (float)(1.0 / SPO_PI);
	out[2] = (float)(1.0 / (double)p);
	out[3] = (float)0.00000000000001;
	out[4] = (float)1000000000000.0;
	*inv_eps = 1.0 / (double)out[3];
}

/* ------------------------------------------------------------------ integrator */
static const float MAX_VALUE_DIST = 1000000000000.0;                     /* cpu_renderer.cpp:27 */

/* cpu_renderer.cpp:36-49 */
int spo_closest_hit(const spo_ray* r, const spo_tri* tris, size_t n_tris, int idx_source, float* d_out, spo_vec3* point_out) {
	float d = MAX_VALUE_DIST;
	int idx = -1;
	spo_vec3 best = v3(0, 0, 0);
	for (int i = 0; i < (int)n_tris; ++i) {
		if (idx_source == i) continue;                                   /* :40-41 */
		spo_vec3 t_pos;
		const float cur_d = spo_ray_intersect(r, &tris[i], &t_pos);
		if ((double)cur_d > 0.0 && cur_d < d) { d = cur_d; best = t_pos; idx = i; } /* :44-48 */
	}
	if (d_out) *d_out = d;
	if (point_out) *point_out = best;
	return idx;
}

/* the same scan for a batch of rays (test helper) */
void spo_closest_hit_batch(const spo_ray* rays, size_t n_rays, const spo_tri* tris, size_t n_tris, const int* src_idx,
                           int* out_idx, float* out_d) {
	for (size_t k = 0; k < n_rays; ++k)
		out_idx[k] = spo_closest_hit(&rays[k], tris, n_tris, src_idx ? src_idx[k] : -1, &out_d[k], 0);
}

typedef struct {
	const spo_tri* tris; const spo_mat* mats; size_t n_tris;
	uint64_t scans;
} spo_scene_ctx;

/* cpu_renderer.cpp:29-68, recursion kept as in the reference */
static spo_vec3 render_step(spo_scene_ctx* sc, const spo_ray* r, spo_rng* g, int idx_source, int depth) {
	if (depth >= 5) return v3(0, 0, 0);                                  /* :33-34 */
	spo_ray next_r;
	float d;
	sc->scans++;
	const int idx = spo_closest_hit(r, sc->tris, sc->n_tris, idx_source, &d, &next_r.pos);
	if (idx < 0) return v3(0, 0, 0);                                     /* :51-52 */
	spo_vec3 adj_n = sc->tris[idx].n;                                    /* :55 */
	if ((double)v_dot(adj_n, r->dir) > 0.0) adj_n = v_muls(adj_n, -1.0f);/* :56-57 */
	if (g->kind == SPO_RNG_COUNTER) { g->depth = (uint32_t)depth; g->draw = 0; }
	const double r1 = rng_next(g), r2 = rng_next(g);
	next_r.dir = spo_rand_unit_vec_from(adj_n, r1, r2);                  /* :58 */
	static const double PI = 3.14159265358979323846;
	const float p = (float)(1.0 / (PI * 2.0));                           /* :60 */
	const float cos_theta = v_dot(next_r.dir, adj_n);                    /* :62 */
	const spo_vec3 BRDF = v_muls(sc->mats[idx].refl, (float)(1.0 / PI)); /* :63 */
	const spo_vec3 rec = render_step(sc, &next_r, g, idx, depth + 1);    /* :65 */
	return v_add(sc->mats[idx].emit, v_muls(v_muls(v_mul(BRDF, rec), cos_theta), (float)(1.0 / (double)p))); /* :67 */
}

/* cpu_renderer.cpp:70-79; returns the averaged accumulator through *acc_out */
static spo_rgba render_core(spo_scene_ctx* sc, const spo_ray* primary, size_t n_samples, spo_rng* g, spo_vec3* acc_out) {
	spo_vec3 accum = v3(0, 0, 0);
	for (int j = 0; j < (int)n_samples; ++j) {
		if (g->kind == SPO_RNG_COUNTER) g->sample = (uint32_t)j;
		accum = v_add(accum, render_step(sc, primary, g, -1, 0));       /* :74-76 */
	}
	accum = v_muls(accum, (float)(1.0 / (double)n_samples));             /* :77 */
	if (acc_out) *acc_out = accum;
	return spo_vec3_rgba(v_clamp01(accum));                              /* :78 */
}

/* cpu_renderer.cpp:81-101 */
void spo_render_flat(const spo_ray* rays, size_t w, size_t h, const spo_tri* tris, const spo_mat* mats, size_t n_tris, spo_rgba* out) {
	const size_t n = w * h;
	for (size_t i = 0; i < n; ++i) {
		const spo_rgba zero = { 0, 0, 0, 0 };
		out[i] = zero;
		float d = MAX_VALUE_DIST;
		for (int j = 0; j < (int)n_tris; ++j) {
			spo_vec3 unused;
			const float cur_d = spo_ray_intersect(&rays[i], &tris[j], &unused);
			if ((double)cur_d > 0.0 && cur_d < d) { d = cur_d; out[i] = spo_vec3_rgba(mats[j].refl); }
		}
	}
}

/* ---- cpu_renderer.cpp:118-184: T simulated reference threads, strided 16-pixel chunks */
typedef struct {
	const spo_ray* rays; size_t total; spo_scene_ctx sc; size_t n_samples;
	int T, s_begin, s_end; spo_rgba* out;
} mt_job;

static void mt_run_sim_thread(mt_job* j, int s) {
	const int chunk_sz = 16;                                             /* :125 */
	const int n_chunks = (int)(j->total / chunk_sz);                     /* :126 */
	const int chunks_per_th = n_chunks / j->T;                           /* :127 */
	spo_rng g; memset(&g, 0, sizeof g);
	g.kind = SPO_RNG_SEED_DIST; g.lcg = (uint32_t)s;                     /* :147 */
	for (int c = 0; c < chunks_per_th; ++c) {
		const int cur_chunk = c * j->T + s;                              /* :149 */
		for (int r = cur_chunk * chunk_sz; r < (cur_chunk + 1) * chunk_sz; ++r)
			j->out[r] = render_core(&j->sc, &j->rays[r], j->n_samples, &g, 0);
	}
	if (s == j->T - 1) {                                                 /* :157-165 */
		const int beg = chunks_per_th * j->T * chunk_sz, end = (int)j->total;
		for (int r = beg; r < end; ++r)
			j->out[r] = render_core(&j->sc, &j->rays[r], j->n_samples, &g, 0);
	}
}

static void* mt_worker(void* p) {
	mt_job* j = (mt_job*)p;
	for (int s = j->s_begin; s < j->s_end; ++s) mt_run_sim_thread(j, s);
	return 0;
}

void spo_render_mt(const spo_ray* rays, size_t w, size_t h, const spo_tri* tris, const spo_mat* mats, size_t n_tris,
                   size_t n_samples, int T, int n_workers, spo_rgba* out) {
	const size_t total = w * h;
	if (T <= 1) {                                                        /* :128-131 -> render_pt :105-116 */
		spo_scene_ctx sc = { tris, mats, n_tris, 0 };
		spo_rng g; memset(&g, 0, sizeof g);
		g.kind = SPO_RNG_STD_RAND;
		srand(1);                                                        /* an unseeded process starts at srand(1) */
		for (size_t i = 0; i < total; ++i) out[i] = render_core(&sc, &rays[i], n_samples, &g, 0);
		return;
	}
	if (n_workers < 1) n_workers = 1;
	if (n_workers > T) n_workers = T;
	mt_job* jobs = (mt_job*)calloc((size_t)n_workers, sizeof(mt_job));
	pthread_t* th = (pthread_t*)calloc((size_t)n_workers, sizeof(pthread_t));
	for (int k = 0; k < n_workers; ++k) {
		mt_job* j = &jobs[k];
		j->rays = rays; j->total = total; j->n_samples = n_samples; j->T = T; j->out = out;
		j->sc.tris = tris; j->sc.mats = mats; j->sc.n_tris = n_tris; j->sc.scans = 0;
		j->s_begin = (int)((long)T * k / n_workers);
		j->s_end = (int)((long)T * (k + 1) / n_workers);
		pthread_create(&th[k], 0, mt_worker, j);
	}
	for (int k = 0; k < n_workers; ++k) pthread_join(th[k], 0);
	free(jobs); free(th);
}

/* ---- counter-RNG variant: the CPU twin of the HIP kernel */
typedef struct {
	const spo_ray* rays; size_t pix0, begin, end; spo_scene_ctx sc; size_t n_samples; uint64_t seed;
	spo_rgba* out; float* accum;
} ctr_job;

static void* ctr_worker(void* p) {
	ctr_job* j = (ctr_job*)p;
	spo_rng g; memset(&g, 0, sizeof g);
	g.kind = SPO_RNG_COUNTER; g.seed = j->seed;
	for (size_t i = j->begin; i < j->end; ++i) {
		g.pixel = (uint32_t)(j->pix0 + i);
		spo_vec3 acc;
		j->out[i] = render_core(&j->sc, &j->rays[j->pix0 + i], j->n_samples, &g, &acc);
		if (j->accum) { j->accum[3 * i] = acc.x; j->accum[3 * i + 1] = acc.y; j->accum[3 * i + 2] = acc.z; }
	}
	return 0;
}

void spo_render_counter(const spo_ray* rays, size_t pix0, size_t npix, const spo_tri* tris, const spo_mat* mats, size_t n_tris,
                        size_t n_samples, uint64_t seed, int n_workers, spo_rgba* out_rgba, float* out_accum, uint64_t* scans_out) {
	if (n_workers < 1) n_workers = 1;
	if ((size_t)n_workers > npix && npix > 0) n_workers = (int)npix;
	ctr_job* jobs = (ctr_job*)calloc((size_t)n_workers, sizeof(ctr_job));
	pthread_t* th = (pthread_t*)calloc((size_t)n_workers, sizeof(pthread_t));
	/* interleaved 64-pixel blocks would balance better; contiguous ranges keep it simple */
	for (int k = 0; k < n_workers; ++k) {
		ctr_job* j = &jobs[k];
		j->rays = rays; j->pix0 = pix0; j->n_samples = n_samples; j->seed = seed; j->out = out_rgba; j->accum = out_accum;
		j->sc.tris = tris; j->sc.mats = mats; j->sc.n_tris = n_tris; j->sc.scans = 0;
		j->begin = npix * (size_t)k / (size_t)n_workers;
		j->end = npix * (size_t)(k + 1) / (size_t)n_workers;
		pthread_create(&th[k], 0, ctr_worker, j);
	}
	uint64_t scans = 0;
	for (int k = 0; k < n_workers; ++k) { pthread_join(th[k], 0); scans += jobs[k].sc.scans; }
	if (scans_out) *scans_out = scans;
	free(jobs); free(th);
}

/* ------------------------------------------------------------------ camera (view.h, basic_renderer.h) */
static void cam_trig(spo_camera* c) {                                    /* view.h:87-92 */
	c->cosY = cosf(c->angle.y); c->sinY = sinf(c->angle.y);
	c->cosX = cosf(c->angle.x); c->sinX = sinf(c->angle.x);
}
static spo_vec3 cam_rY(const spo_camera* c, spo_vec3 in) {               /* view.h:54-60 */
	return v3(in.x * c->cosY + in.z * c->sinY, in.y, in.x * -c->sinY + in.z * c->cosY);
}
static spo_vec3 cam_rX(const spo_camera* c, spo_vec3 in) {               /* view.h:62-68 */
	return v3(in.x, in.y * c->cosX + in.z * -c->sinX, in.y * c->sinX + in.z * c->cosX);
}
static spo_vec3 cam_rel_move(const spo_camera* c, spo_vec3 in) { return cam_rY(c, cam_rX(c, in)); } /* view.h:83-85 */

void spo_camera_init(spo_camera* c, size_t w, size_t h) {                /* view.h:76-81 */
	c->pos = v3(0.0f, 0.0f, -3.0f); c->angle = v3(0, 0, 0); c->focal = 2.0f; c->res_x = w; c->res_y = h;
	cam_trig(c);
}
void spo_camera_set_viewport_size(spo_camera* c, int w, int h) { c->res_x = (size_t)w; c->res_y = (size_t)h; }
void spo_camera_delta_mov(spo_camera* c, spo_vec3 m) { c->pos = v_add(c->pos, cam_rel_move(c, m)); }
void spo_camera_delta_rot(spo_camera* c, spo_vec3 r) { c->angle = v_add(c->angle, r); cam_trig(c); }
void spo_camera_delta_focal(spo_camera* c, float f) { c->focal += f; }

void spo_camera_get_viewport(const spo_camera* c, spo_ray* rays) {       /* view.h:94-132 */
	const size_t res_x = c->res_x, res_y = c->res_y;
	const float x_size = (float)(1.0 * (double)res_x / (double)res_y),   /* :101 */
	            y_size = 1.0f,
	            x_max = (float)((double)x_size / 2.0),
	            x_step = x_size / (float)res_x,
	            h_x_step = (float)((double)x_step / 2.0),
	            y_max = (float)((double)y_size / 2.0),
	            y_step = y_size / (float)res_y,
	            h_y_step = (float)((double)y_step / 2.0);
	for (int i = 0; i < (int)res_x; ++i) {
		for (int j = 0; j < (int)res_y; ++j) {
			const spo_vec3 cur_pos = v3(x_max - x_step * (float)i - h_x_step, y_max - y_step * (float)j - h_y_step, 0.0f); /* :111 */
			spo_ray* r = &rays[(size_t)i + (size_t)j * res_x];
			r->pos = cur_pos;
			r->dir = v_unit(v_add(cur_pos, v3(0.0f, 0.0f, c->focal)));   /* :114 */
		}
	}
	for (size_t k = 0; k < res_x * res_y; ++k) {                         /* :125-128 */
		rays[k].pos = cam_rel_move(c, rays[k].pos);
		rays[k].dir = cam_rel_move(c, rays[k].dir);
	}
	for (size_t k = 0; k < res_x * res_y; ++k)                           /* :130-131 */
		rays[k].pos = v_add(rays[k].pos, c->pos);
}


/* ------------------------------------------------------------------ batch forms for the device self-test sweeps
 * (tests/test_hip_device_math.py compares sphip_selftest_device against these on up to 1e9 inputs) */
typedef struct { int what; size_t begin, end; const void* in; void* out; } batch_job;

static void* batch_worker(void* arg) {
	batch_job* j = (batch_job*)arg;
	for (size_t i = j->begin; i < j->end; ++i) {
		switch (j->what) {
		case 0: { const float x = ((const float*)j->in)[i]; ((float*)j->out)[2 * i] = spo_sinf(x); ((float*)j->out)[2 * i + 1] = spo_cosf(x); break; }
		case 2: { const uint32_t* q = (const uint32_t*)j->in + 5 * i;
		          spo_counter_uniforms((uint64_t)q[0] | ((uint64_t)q[1] << 32), q[2], q[3], q[4], (double*)j->out + 2 * i, (double*)j->out + 2 * i + 1); break; }
		case 3: { const double* q = (const double*)j->in + 5 * i; spo_vec3 n = { (float)q[0], (float)q[1], (float)q[2] };
		          const spo_vec3 v = spo_rand_unit_vec_from(n, q[3], q[4]); float* o = (float*)j->out + 3 * i; o[0] = v.x; o[1] = v.y; o[2] = v.z; break; }
		case 4: { const float* q = (const float*)j->in + 15 * i; spo_ray r = { { q[0], q[1], q[2] }, { q[3], q[4], q[5] } };
		          spo_tri t = { { q[6], q[7], q[8] }, { q[9], q[10], q[11] }, { q[12], q[13], q[14] }, { 0, 0, 0 } }; spo_vec3 pt;
		          ((float*)j->out)[i] = spo_ray_intersect(&r, &t, &pt); break; }
		case 5: { const float* q = (const float*)j->in + 3 * i; spo_vec3 v = { q[0], q[1], q[2] }; const spo_rgba c = spo_vec3_rgba(v);
		          ((uint32_t*)j->out)[i] = (uint32_t)c.r | ((uint32_t)c.g << 8) | ((uint32_t)c.b << 16) | ((uint32_t)c.a << 24); break; }
		default: break;
		}
	}
	return 0;
}

/* what: 0 sincos (in f32[n] -> out f32[2n]: sin, cos), 2 counter uniforms (u32[5n]: seed lo, seed hi, pixel, sample, depth -> f64[2n]),
 * 3 rand_unit_vec (f64[5n]: n.xyz, r1, r2 -> f32[3n]), 4 ray_intersect (f32[15n]: pos dir v0 v1 v2 -> f32[n]), 5 vec3_RGBA (f32[3n] -> u32[n]).
 * (1 = the IEEE reciprocal needs no oracle: numpy's float32 divide is it.) */
void spo_device_math_batch(int what, const void* in, size_t n, void* out, int n_workers) {
	if (n_workers < 1) n_workers = 1;
	if ((size_t)n_workers > n && n > 0) n_workers = (int)n;
	batch_job* jobs = (batch_job*)calloc((size_t)n_workers, sizeof(batch_job));
	pthread_t* th = (pthread_t*)calloc((size_t)n_workers, sizeof(pthread_t));
	for (int k = 0; k < n_workers; ++k) {
		jobs[k].what = what; jobs[k].in = in; jobs[k].out = out;
		jobs[k].begin = n * (size_t)k / (size_t)n_workers; jobs[k].end = n * (size_t)(k + 1) / (size_t)n_workers;
		pthread_create(&th[k], 0, batch_worker, &jobs[k]);
	}
	for (int k = 0; k < n_workers; ++k) pthread_join(th[k], 0);
	free(jobs); free(th);
}
