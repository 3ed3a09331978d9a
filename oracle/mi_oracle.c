/*
 * mi_oracle.c -- CPU restatement (plain C99 + OpenMP) of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mitsuba3-camera_amd/, csrc/,
 * libmi355.so) may include, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg use it, and there only as the checker / the reported
 * CPU baseline.
 *
 * PARITY STATUS: the reference (Mitsuba 3.5.0 fork + Dr.Jit 0.4.4) can neither be built nor
 * imported in this environment (ext/drjit, ext/embree, ... are empty submodules),
 * and it ships no reference images.  This restatement is therefore pinned ONLY by the
 * known-answer vectors the reference's own tests hold for the leaf functions of the path
 * (tests/golden/reference_kats.json, checked in tests/test_oracle_kats.py):
 *   TEA (src/core/tests/test_random.py:8-27), perspective_projection and parse_fov
 *   (src/render/tests/test_sensor.py:5-57), diffuse eval/pdf (src/bsdfs/tests/test_diffuse.py:14-37),
 *   rectangle hit grid (src/shapes/tests/test_rectangle.py:33-57), cube hit/normal/uv
 *   (src/shapes/tests/test_cube.py:26-70), gaussian filter / imageblock / hdrfilm identities.
 * PCG32 arithmetic lives in Dr.Jit (absent): pinned to the published pcg32 demo vector instead.
 * END-TO-END PIN: the one whole-render number the reference's tests hard-code for this path -- the particle tracer's
 * d mean(image) / d radiance = 0.33647 +- 1e-5 (src/integrators/tests/test_ptracer.py:204-225, JIT variants) -- is reproduced
 * (0.3364746) by orc_ptracer_accumulate / orc_ptracer_backward.  It depends on, and therefore pins together: TEA seeding of the
 * sampler streams and the PCG32 seeding order, the pass structure of the render loop, the order and masking of the sampler calls,
 * Rectangle::sample_position, PerspectiveCamera::sample_direction / importance, sample_scale and the normalised splat.
 * Whole-image radiance of the PATH integrator and PRB gradients against llvm_ad_rgb have no such number in the reference and stay
 * "parity unpinned"; they share the sampler, sensor, shape and film code pinned above and are otherwise checked by
 * self-consistency (finite differences, energy identities, agreement with the pinned light tracer).
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 * Arithmetic notes: Dr.Jit's dot/cross/matrix-vector products are fmadd chains, ray(t) is an
 * fmadd, `Array / scalar` is a multiplication by rcp(); this file spells those out with fmaf()
 * and must be compiled with -ffp-contract=off so that nothing else is fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/mi355.h"

#define ORC_API __attribute__((visibility("default")))

/* ----------------------------------------------------------------------------------------
 * constants (include/mitsuba/core/math.h:17-23; dr::Epsilon<float> = 2^-24 [Dr.Jit, recalled])
 * ---------------------------------------------------------------------------------------- */
#define ORC_EPS        5.9604644775390625e-8f
#define RAY_EPS        (ORC_EPS * 1500.f)
#define SHADOW_EPS     (RAY_EPS * 10.f)
#define ORC_PI         3.14159265358979323846f
#define ORC_INV_PI     0.31830988618379067154f
#define ORC_LARGEST    FLT_MAX

typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* dr::fmadd(a, s, c) on arrays */
static inline v3 fma3s(v3 a, float s, v3 c) { return V3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }
static inline v3 fma3(v3 a, v3 b, v3 c) { return V3(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)); }
/* Dr.Jit dot_(): x0*y0, then fmadd per further component [recalled] */
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* Dr.Jit cross(): fmsub(a.yzx, b.zxy, a.zxy * b.yzx) [recalled] */
static inline v3 cross3(v3 a, v3 b) {
    return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float rcpf_(float x) { return 1.0f / x; }
static inline float rsqrtf_(float x) { return 1.0f / sqrtf(x); }
static inline float sqnorm3(v3 a) { return dot3(a, a); }
static inline float norm3(v3 a) { return sqrtf(dot3(a, a)); }
/* dr::normalize(v) = v * rsqrt(squared_norm(v)) [recalled] */
static inline v3 normalize3(v3 a) { return scale3(a, rsqrtf_(sqnorm3(a))); }
static inline float max3(v3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
static inline float mulsign(float a, float b) {
    uint32_t ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    ua ^= (ub & 0x80000000u); float r; memcpy(&r, &ua, 4); return r;
}
static inline float signf_(float x) { return copysignf(1.0f, x); }

/* include/mitsuba/core/vector.h:116-136 */
static void coordinate_system(v3 n, v3 *s, v3 *t) {
    float sign = signf_(n.z), a = -rcpf_(sign + n.z), b = n.x * n.y * a;
    *s = V3(mulsign((n.x * n.x) * a, n.z) + 1.0f, mulsign(b, n.z), mulsign(-n.x, n.z));
    *t = V3(b, fmaf(n.y, n.y * a, sign), -n.y);
}

/* 3x4 affine transforms; include/mitsuba/core/transform.h:98-141 */
static inline v3 xf_point(const float *m, v3 p) {
    v3 r = V3(m[3], m[7], m[11]);
    r = fma3s(V3(m[0], m[4], m[8]), p.x, r);
    r = fma3s(V3(m[1], m[5], m[9]), p.y, r);
    r = fma3s(V3(m[2], m[6], m[10]), p.z, r);
    return r;
}
static inline v3 xf_vector(const float *m, v3 v) {
    v3 r = scale3(V3(m[0], m[4], m[8]), v.x);
    r = fma3s(V3(m[1], m[5], m[9]), v.y, r);
    r = fma3s(V3(m[2], m[6], m[10]), v.z, r);
    return r;
}
/* normals use the inverse transpose = transpose(to_object) (transform.h:146-155) */
static inline v3 xf_normal(const float *to_object, v3 n) {
    const float *m = to_object; /* (M^-1)^T columns are rows of M^-1 */
    v3 r = scale3(V3(m[0], m[1], m[2]), n.x);
    r = fma3s(V3(m[4], m[5], m[6]), n.y, r);
    r = fma3s(V3(m[8], m[9], m[10]), n.z, r);
    return r;
}

/* ----------------------------------------------------------------------------------------
 * RNG: TEA + PCG32
 * ---------------------------------------------------------------------------------------- */
/* include/mitsuba/core/random.h:76-91 */
ORC_API void orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t out[2]) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    out[0] = v0; out[1] = v1;
}
/* random.h:136-140 */
ORC_API float orc_tea_float32(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t o[2]; orc_tea32(v0, v1, rounds, o);
    uint32_t u = (o[1] >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f;
}
/* random.h:158-162 */
ORC_API double orc_tea_float64(uint32_t v0, uint32_t v1, int rounds) {
    uint32_t o[2]; orc_tea32(v0, v1, rounds, o);
    uint64_t v = (uint64_t) o[0] + ((uint64_t) o[1] << 32);
    uint64_t u = (v >> 12) | 0x3ff0000000000000ull; double d; memcpy(&d, &u, 8); return d - 1.0;
}

typedef struct { uint64_t state, inc; } pcg32;
#define PCG32_MULT 0x5851f42d4c957f2dULL
/* PCG32 (O'Neill; Dr.Jit random.h [recalled]) */
static inline uint32_t pcg32_next(pcg32 *r) {
    uint64_t old = r->state;
    r->state = old * PCG32_MULT + r->inc;
    uint32_t xorshift = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xorshift >> rot) | (xorshift << ((~rot + 1u) & 31u));
}
static inline void pcg32_seed(pcg32 *r, uint64_t initstate, uint64_t initseq) {
    r->state = 0; r->inc = (initseq << 1u) | 1u;
    pcg32_next(r); r->state += initstate; pcg32_next(r);
}
static inline float pcg32_float(pcg32 *r) {
    uint32_t u = (pcg32_next(r) >> 9) | 0x3f800000u; float f; memcpy(&f, &u, 4); return f - 1.0f;
}
ORC_API void orc_pcg32_raw(uint64_t initstate, uint64_t initseq, int n, uint32_t *out) {
    pcg32 r; pcg32_seed(&r, initstate, initseq);
    for (int i = 0; i < n; ++i) out[i] = pcg32_next(&r);
}
/* PCG32Sampler::seed, JIT branch: src/render/sampler.cpp:116-135 (base_seed = 0) */
static inline void sampler_seed(pcg32 *r, uint32_t seed, uint32_t lane) {
    uint32_t v[2]; orc_tea32(seed, lane, 4, v);
    pcg32_seed(r, (uint64_t) v[0], (uint64_t) v[1]);
}
/* IndependentSampler::next_1d: src/samplers/independent.cpp:88-97 */
ORC_API void orc_sampler_floats(uint32_t seed, uint32_t lane, int n, float *out) {
    pcg32 r; sampler_seed(&r, seed, lane);
    for (int i = 0; i < n; ++i) out[i] = pcg32_float(&r);
}

/* ----------------------------------------------------------------------------------------
 * warps, filter
 * ---------------------------------------------------------------------------------------- */
/* include/mitsuba/core/warp.h:54-90 */
static v2 square_to_uniform_disk_concentric(v2 s) {
    float x = fmaf(2.f, s.x, -1.f), y = fmaf(2.f, s.y, -1.f);
    int is_zero = (x == 0.f && y == 0.f), q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * ORC_PI * rp / r;
    if (q13) phi = 0.5f * ORC_PI - phi;
    if (is_zero) phi = 0.f;
    float sn = sinf(phi), cs = cosf(phi);
    v2 o = { r * cs, r * sn }; return o;
}
/* warp.h:412-420 */
static v3 square_to_cosine_hemisphere(v2 s) {
    v2 p = square_to_uniform_disk_concentric(s);
    float sq = fmaf(p.y, p.y, p.x * p.x);
    float z = sqrtf(fmaxf(1.f - sq, 0.f));
    return V3(p.x, p.y, z);
}
ORC_API void orc_square_to_cosine_hemisphere(const float *s, int n, float *out) {
    for (int i = 0; i < n; ++i) { v2 q = { s[2*i], s[2*i+1] }; v3 r = square_to_cosine_hemisphere(q);
        out[3*i] = r.x; out[3*i+1] = r.y; out[3*i+2] = r.z; }
}

/* Dr.Jit estrin_impl [recalled]: pairwise fmadd, recurse on x^2 */
static float estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fmaf(x, c[1], c[0]), a1 = fmaf(x, c[3], c[2]), a2 = fmaf(x, c[5], c[4]),
          a3 = fmaf(x, c[7], c[6]), a4 = fmaf(x, c[9], c[8]);
    float b0 = fmaf(x2, a1, a0), b1 = fmaf(x2, a3, a2), b2 = a4;
    float c0 = fmaf(x4, b1, b0), c1 = b2;
    return fmaf(x8, c1, c0);
}
/* GaussianFilter ctor: src/rfilters/gaussian.cpp:46-86 */
ORC_API void orc_gaussian_coeffs(float stddev, float *out) {
    static const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2,
                                      2.136713061e-3, -1.679873860e-4, 9.202145248e-6, -3.329417433e-7,
                                      7.128382794e-9, -6.821193280e-11 };
    double scale = 1; float radius = 4.f * stddev;
    for (int i = 0; i < 10; ++i) { out[i] = (float) (coeff[i] * scale); scale /= ((double) stddev * (double) stddev); }
    out[0] -= estrin10(radius * radius, out);
}
/* GaussianFilter::eval (non-CUDA): gaussian.cpp:88-98 */
static inline float gaussian_eval(const float *c, float x) { return fmaxf(estrin10(x * x, c), 0.f); }
ORC_API void orc_gaussian_eval(float stddev, const float *x, int n, float *out) {
    float c[10]; orc_gaussian_coeffs(stddev, c);
    for (int i = 0; i < n; ++i) out[i] = gaussian_eval(c, x[i]);
}
/* TentFilter::eval: src/rfilters/tent.cpp (max(0, 1 - |x / radius|)) */
static inline float tent_eval(float radius, float x) { return fmaxf(0.f, 1.f - fabsf(x * (1.f / radius))); }

/* ----------------------------------------------------------------------------------------
 * sensor
 * ---------------------------------------------------------------------------------------- */
static void mat4_mul(const float *a, const float *b, float *o) {
    float r[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = a[i*4+0] * b[0*4+j];
        for (int k = 1; k < 4; ++k) s = fmaf(a[i*4+k], b[k*4+j], s);
        r[i*4+j] = s;
    }
    memcpy(o, r, sizeof(r));
}
static void mat4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }
static void mat4_scale(float x, float y, float z, float *m) { mat4_identity(m); m[0] = x; m[5] = y; m[10] = z; }
static void mat4_translate(float x, float y, float z, float *m) { mat4_identity(m); m[3] = x; m[7] = y; m[11] = z; }
/* Transform::perspective: include/mitsuba/core/transform.h:216-233 */
static void mat4_perspective(float fov, float near_, float far_, float *m, float *inv) {
    float recip = 1.f / (far_ - near_);
    float tn = tanf((fov * .5f) * (ORC_PI / 180.f)), cot = 1.f / tn;
    memset(m, 0, 64); m[0] = cot; m[5] = cot; m[10] = far_ * recip; m[11] = -near_ * far_ * recip; m[14] = 1.f;
    memset(inv, 0, 64); inv[0] = tn; inv[5] = tn; inv[15] = 1.f / near_; inv[11] = 1.f;
    inv[14] = (near_ - far_) / (far_ * near_);
}
/* perspective_projection: include/mitsuba/render/sensor.h:228-265; returns matrix and its
   analytic inverse (Transform keeps both, transform.h:62-71) */
static void perspective_projection(const int32_t *film, const int32_t *crop, const int32_t *off,
                                   float fov_x, float near_, float far_, float *m, float *inv) {
    float fx = (float) film[0], fy = (float) film[1];
    float rsx = (float) crop[0] / fx, rsy = (float) crop[1] / fy;
    float rox = (float) off[0] / fx, roy = (float) off[1] / fy;
    float aspect = fx / fy;
    float S1[16], T1[16], S2[16], T2[16], P[16], Pi[16], t[16];
    float S1i[16], T1i[16], S2i[16], T2i[16];
    mat4_scale(1.f / rsx, 1.f / rsy, 1.f, S1);          mat4_scale(1.f / (1.f / rsx), 1.f / (1.f / rsy), 1.f, S1i);
    mat4_translate(-rox, -roy, 0.f, T1);                mat4_translate(rox, roy, 0.f, T1i);
    mat4_scale(-0.5f, -0.5f * aspect, 1.f, S2);         mat4_scale(1.f / -0.5f, 1.f / (-0.5f * aspect), 1.f, S2i);
    mat4_translate(-1.f, -1.f / aspect, 0.f, T2);       mat4_translate(1.f, 1.f / aspect, 0.f, T2i);
    mat4_perspective(fov_x, near_, far_, P, Pi);
    mat4_mul(S1, T1, t); mat4_mul(t, S2, t); mat4_mul(t, T2, t); mat4_mul(t, P, m);
    mat4_mul(Pi, T2i, t); mat4_mul(t, S2i, t); mat4_mul(t, T1i, t); mat4_mul(t, S1i, inv);
}
ORC_API void orc_perspective_projection(const int32_t film[2], const int32_t crop[2], const int32_t off[2],
                                        float fov_x, float near_, float far_, float *out16, float *out_inv16) {
    float inv[16]; perspective_projection(film, crop, off, fov_x, near_, far_, out16, inv);
    if (out_inv16) memcpy(out_inv16, inv, 64);
}

/* ----------------------------------------------------------------------------------------
 * scene
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t type, bsdf; int32_t emitter; uint32_t flip_normals;
    float to_world[12], to_object[12];
    v3 frame_s, frame_t, frame_n; float inv_area;        /* rectangle.cpp:115-126 */
    v3 center; float radius;                              /* sphere.cpp:158-159 */
    uint32_t vertex_count, face_count;
    float *pos, *nrm, *uv; uint32_t *faces;
    float *area_cdf;                                      /* Mesh::build_pmf (mesh.cpp:430-470): inclusive float prefix sums of the face areas */
    float *nrm_len; int nrm_dep;                          /* vertex normals recomputed from the positions by a parameter update (mesh.cpp:138-139): lengths of the sums */
    struct ONode *bvh; uint32_t *bvh_faces; uint32_t n_nodes;   /* meshes of more than ORC_BVH_MIN faces: a median-split BVH (speed only: same closest hit) */
} OShape;
typedef struct ONode { float lo[3], hi[3]; uint32_t left, count; } ONode;
#define ORC_BVH_MIN 64
typedef struct { uint32_t w, h, c, filter, wrap; float to_uv[6]; float *data; } OTexture;
typedef struct {
    uint32_t n_shapes, n_bsdfs, n_emitters, n_textures;
    OShape *shapes; mi355_bsdf_desc *bsdfs; mi355_emitter_desc *emitters; OTexture *textures;
    mi355_sensor_desc sensor;
    float sample_to_camera[16], camera_to_sample[16];
    float filter_coeff[10]; float filter_radius;
    uint32_t border;                                        /* Film::sample_border ? ReconstructionFilter::border_size : 0 */
    float *lens; uint32_t lens_count; float film_px, film_py;   /* realisticlens: interface table, physical half extents */
    float pupil[64 * 4];                                          /* exit-pupil bounds (min.x, min.y, max.x, max.y) per film-radius segment */
    uint64_t c_isect, c_shade, c_shadow;                  /* event counters */
    uint64_t c_isect_dead;                                /* light tracer: intersections the reference computes and never reads (orc_dead_intersections) */
    /* Scene::sensors(): every sensor of the scene; the fields above (`sensor` ... `pupil`) are those of sensor `cur_sensor`, the one the
       `Sensor *` argument of Integrator::render picks (mi355_render_params.sensor; integrator.h:74-97) */
    mi355_sensor_desc *all_sensors; float **all_lens; float *all_pupil; int *pupil_done; uint32_t n_sensors, cur_sensor;
} OScene;

static void *dupmem(const void *p, size_t n) { if (!p || !n) return NULL; void *q = malloc(n); memcpy(q, p, n); return q; }

static void exit_pupil_bounds(OScene *sc, float film_diagonal);
ORC_API int orc_select_sensor(OScene *s, uint32_t index);
ORC_API int orc_scene_set_sensor(OScene *s, uint32_t index, const mi355_sensor_desc *se);
static void orc_bvh_build(OShape *sh);
static void recompute_vertex_normals(OShape *sh);
ORC_API OScene *orc_scene_create(const mi355_scene_desc *d) {
    if (!d || d->abi_version != MI355_ABI_VERSION) return NULL;
    OScene *s = (OScene *) calloc(1, sizeof(OScene));
    s->n_shapes = d->n_shapes; s->n_bsdfs = d->n_bsdfs; s->n_emitters = d->n_emitters; s->n_textures = d->n_textures;
    s->shapes = (OShape *) calloc(d->n_shapes ? d->n_shapes : 1, sizeof(OShape));
    s->bsdfs = (mi355_bsdf_desc *) dupmem(d->bsdfs, sizeof(mi355_bsdf_desc) * d->n_bsdfs);
    s->emitters = (mi355_emitter_desc *) dupmem(d->emitters, sizeof(mi355_emitter_desc) * d->n_emitters);
    s->textures = (OTexture *) calloc(d->n_textures ? d->n_textures : 1, sizeof(OTexture));
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const mi355_texture_desc *t = &d->textures[i]; OTexture *o = &s->textures[i];
        o->w = t->width; o->h = t->height; o->c = t->channels; o->filter = t->filter; o->wrap = t->wrap;
        memcpy(o->to_uv, t->to_uv, sizeof(o->to_uv));
        o->data = (float *) dupmem(t->data, sizeof(float) * (size_t) t->width * t->height * t->channels);
    }
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const mi355_shape_desc *sd = &d->shapes[i]; OShape *o = &s->shapes[i];
        o->type = sd->type; o->bsdf = sd->bsdf; o->emitter = sd->emitter; o->flip_normals = sd->flip_normals;
        memcpy(o->to_world, sd->to_world, 48); memcpy(o->to_object, sd->to_object, 48);
        if (sd->type == MI355_SHAPE_DISK) {
            /* Disk::update, disk.cpp:113-127; surface_area :163-167 */
            v3 du = xf_vector(o->to_world, V3(1.f, 0.f, 0.f)), dv = xf_vector(o->to_world, V3(0.f, 1.f, 0.f));
            float lu = norm3(du), lv = norm3(dv);
            o->frame_s = scale3(du, 1.f / lu); o->frame_t = scale3(dv, 1.f / lv);
            o->frame_n = normalize3(xf_normal(o->to_object, V3(0.f, 0.f, 1.f)));
            float dd = dot3(scale3(o->frame_t, lv), o->frame_s), hgt = sqrtf(lv * lv - dd * dd);
            o->inv_area = rcpf_(ORC_PI * lu * hgt);
        } else if (sd->type == MI355_SHAPE_RECTANGLE) {
            /* Rectangle::update, rectangle.cpp:115-126 */
            o->frame_s = xf_vector(o->to_world, V3(2.f, 0.f, 0.f));
            o->frame_t = xf_vector(o->to_world, V3(0.f, 2.f, 0.f));
            o->frame_n = normalize3(xf_normal(o->to_object, V3(0.f, 0.f, 1.f)));
            o->inv_area = rcpf_(norm3(cross3(o->frame_s, o->frame_t)));
        } else if (sd->type == MI355_SHAPE_MESH) {
            o->vertex_count = sd->vertex_count; o->face_count = sd->face_count;
            o->pos = (float *) dupmem(sd->vertex_positions, 12u * sd->vertex_count);
            o->nrm = (float *) dupmem(sd->vertex_normals, 12u * sd->vertex_count);
            o->uv = (float *) dupmem(sd->vertex_texcoords, 8u * sd->vertex_count);
            o->faces = (uint32_t *) dupmem(sd->faces, 12u * sd->face_count);
            /* Mesh::build_pmf + DiscreteDistribution::compute_cdf (mesh.cpp:430-470, distr_1d.h:218-231): surface area = last CDF entry */
            o->area_cdf = (float *) malloc(4u * (o->face_count ? o->face_count : 1u));
            float area = 0.f;
            for (uint32_t f = 0; f < o->face_count; ++f) {
                const uint32_t *fi = o->faces + 3 * f;
                v3 p0 = V3(o->pos[3*fi[0]], o->pos[3*fi[0]+1], o->pos[3*fi[0]+2]);
                v3 p1 = V3(o->pos[3*fi[1]], o->pos[3*fi[1]+1], o->pos[3*fi[1]+2]);
                v3 p2 = V3(o->pos[3*fi[2]], o->pos[3*fi[2]+1], o->pos[3*fi[2]+2]);
                area += 0.5f * norm3(cross3(sub3(p1, p0), sub3(p2, p0)));
                o->area_cdf[f] = area;
            }
            o->inv_area = 1.f / area;
            orc_bvh_build(o);
        } else if (sd->type == MI355_SHAPE_SPHERE) {
            /* Sphere::update, sphere.cpp:150-172 (uniform positive scale assumed) */
            o->center = xf_point(o->to_world, V3(0.f, 0.f, 0.f));
            o->radius = norm3(xf_vector(o->to_world, V3(1.f, 0.f, 0.f)));
            o->inv_area = rcpf_(4.f * ORC_PI * (o->radius * o->radius));
        } else { free(s); return NULL; }
    }
    if (!d->n_sensors || !d->sensors) { free(s); return NULL; }
    s->n_sensors = d->n_sensors; s->cur_sensor = 0xffffffffu;
    s->all_sensors = (mi355_sensor_desc *) dupmem(d->sensors, sizeof(mi355_sensor_desc) * d->n_sensors);
    s->all_lens = (float **) calloc(d->n_sensors, sizeof(float *));
    s->all_pupil = (float *) calloc((size_t) d->n_sensors * 256, sizeof(float)); s->pupil_done = (int *) calloc(d->n_sensors, sizeof(int));
    for (uint32_t i = 0; i < d->n_sensors; ++i) if (d->sensors[i].type == MI355_SENSOR_REALISTICLENS) {
        s->all_lens[i] = (float *) dupmem(d->sensors[i].lens_data, 64u * d->sensors[i].lens_count); s->all_sensors[i].lens_data = s->all_lens[i];
    }
    orc_select_sensor(s, 0);
    return s;
}
/* Makes sensor `index` the current one: PerspectiveCamera::update_camera_transforms (perspective.cpp:173-199), the film / filter
   quantities, the realisticlens table and (once per sensor) its exit-pupil bounds. */
ORC_API int orc_select_sensor(OScene *s, uint32_t index) {
    if (index >= s->n_sensors) return -1;
    if (s->cur_sensor == index) return 0;
    const mi355_sensor_desc *se = &s->all_sensors[index];
    s->sensor = *se; s->cur_sensor = index;
    int32_t film[2] = { (int32_t) se->film_width, (int32_t) se->film_height };
    int32_t crop[2] = { (int32_t) se->crop_width, (int32_t) se->crop_height };
    int32_t off[2] = { (int32_t) se->crop_offset_x, (int32_t) se->crop_offset_y };
    perspective_projection(film, crop, off, se->x_fov, se->near_clip, se->far_clip, s->camera_to_sample, s->sample_to_camera);
    s->lens = NULL; s->lens_count = 0;
    if (se->rfilter == MI355_RFILTER_GAUSSIAN) {
        orc_gaussian_coeffs(se->rfilter_param, s->filter_coeff); s->filter_radius = 4.f * se->rfilter_param;
    } else if (se->rfilter == MI355_RFILTER_TENT) s->filter_radius = se->rfilter_param;
    else s->filter_radius = 0.5f;
    {   /* rfilter.cpp:23: border_size = ceil(radius - 0.5 - 2 RayEps) */
        float bs = ceilf(s->filter_radius - .5f - 2.f * RAY_EPS);
        s->border = se->sample_border ? (uint32_t) (bs > 0.f ? bs : 0.f) : 0u;
    }
    if (se->type == MI355_SENSOR_REALISTICLENS) {
        s->lens_count = se->lens_count; s->lens = s->all_lens[index];
        float aspect = (float) se->film_height / (float) se->film_width;      /* Film::get_physical_size, film.h:193-198 */
        float x = sqrtf(se->film_diagonal * se->film_diagonal / (1.f + aspect * aspect));
        s->film_px = x / 2; s->film_py = aspect * x / 2;
        if (se->sample_exit_pupil) {
            if (!s->pupil_done[index]) { exit_pupil_bounds(s, se->film_diagonal); memcpy(s->all_pupil + 256 * (size_t) index, s->pupil, sizeof(s->pupil)); s->pupil_done[index] = 1; }
            else memcpy(s->pupil, s->all_pupil + 256 * (size_t) index, sizeof(s->pupil));
        }
    }
    return 0;
}
/* mi355_scene_set_sensor: replace the description of sensor `index` (PerspectiveCamera::parameters_changed, perspective.cpp:163-171;
   a film resize, film.cpp:59-98). */
ORC_API int orc_scene_set_sensor(OScene *s, uint32_t index, const mi355_sensor_desc *se) {
    if (index >= s->n_sensors) return -1;
    float *lens = se->type == MI355_SENSOR_REALISTICLENS ? (float *) dupmem(se->lens_data, 64u * se->lens_count) : NULL;   /* (se may point into the slot) */
    mi355_sensor_desc nd = *se; nd.lens_data = lens;
    free(s->all_lens[index]); s->all_lens[index] = lens;
    s->all_sensors[index] = nd;
    s->pupil_done[index] = 0;
    if (s->cur_sensor == index) { s->cur_sensor = 0xffffffffu; return orc_select_sensor(s, index); }
    return 0;
}
ORC_API void orc_scene_destroy(OScene *s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->n_shapes; ++i) { free(s->shapes[i].pos); free(s->shapes[i].nrm); free(s->shapes[i].uv); free(s->shapes[i].faces);
                                                 free(s->shapes[i].area_cdf); free(s->shapes[i].nrm_len); free(s->shapes[i].bvh); free(s->shapes[i].bvh_faces); }
    for (uint32_t i = 0; i < s->n_textures; ++i) free(s->textures[i].data);
    for (uint32_t i = 0; i < s->n_sensors; ++i) free(s->all_lens[i]);
    free(s->all_lens); free(s->all_sensors); free(s->all_pupil); free(s->pupil_done);
    free(s->shapes); free(s->bsdfs); free(s->emitters); free(s->textures); free(s);
}
/* SceneParameters.update for one parameter: src/python/python/util.py:185-232 */
ORC_API int orc_scene_set_param(OScene *s, uint32_t kind, uint32_t index, const float *v, uint64_t n) {
    if (kind == MI355_PARAM_BSDF_REFLECTANCE && index < s->n_bsdfs && n == 3) { memcpy(s->bsdfs[index].reflectance, v, 12); return 0; }
    if (kind == MI355_PARAM_EMITTER_RADIANCE && index < s->n_emitters && n == 3) { memcpy(s->emitters[index].radiance, v, 12); return 0; }
    if (kind == MI355_PARAM_TEXTURE_DATA && index < s->n_textures) {
        OTexture *t = &s->textures[index]; if (n != (uint64_t) t->w * t->h * t->c) return -1;
        memcpy(t->data, v, 4 * n); return 0;
    }
    if (kind == MI355_PARAM_VERTEX_POSITIONS && index < s->n_shapes && s->shapes[index].type == MI355_SHAPE_MESH) {
        /* Mesh::parameters_changed (mesh.cpp:96-163): bbox / accel follow, vertex normals are recomputed when the mesh has them, the
           area table is rebuilt */
        OShape *o = &s->shapes[index]; if (n != 3ull * o->vertex_count) return -1;
        memcpy(o->pos, v, 4 * n);
        recompute_vertex_normals(o);
        float area = 0.f;
        for (uint32_t f = 0; f < o->face_count; ++f) {
            const uint32_t *fi = o->faces + 3 * f;
            v3 p0 = V3(o->pos[3*fi[0]], o->pos[3*fi[0]+1], o->pos[3*fi[0]+2]), p1 = V3(o->pos[3*fi[1]], o->pos[3*fi[1]+1], o->pos[3*fi[1]+2]), p2 = V3(o->pos[3*fi[2]], o->pos[3*fi[2]+1], o->pos[3*fi[2]+2]);
            area += 0.5f * norm3(cross3(sub3(p1, p0), sub3(p2, p0))); o->area_cdf[f] = area;
        }
        o->inv_area = 1.f / area;
        orc_bvh_build(o);
        return 0;
    }
    if (kind == MI355_PARAM_BSDF_IOR && index < s->n_bsdfs && s->bsdfs[index].type == MI355_BSDF_DIELECTRIC && n == 2) {
        s->bsdfs[index].int_ior = v[0]; s->bsdfs[index].ext_ior = v[1]; return 0;
    }
    if ((kind == MI355_PARAM_SENSOR_TO_WORLD && n == 12) || (kind == MI355_PARAM_SENSOR_X_FOV && n == 1)) {     /* perspective.cpp:155-171 */
        if (index >= s->n_sensors) return -1;
        mi355_sensor_desc nd = s->all_sensors[index];
        if (kind == MI355_PARAM_SENSOR_TO_WORLD) memcpy(nd.to_world, v, 48); else nd.x_fov = v[0];
        return orc_scene_set_sensor(s, index, &nd);
    }
    return -1;
}
ORC_API void orc_get_counters(OScene *s, uint64_t out[3], int reset) {
    out[0] = s->c_isect; out[1] = s->c_shade; out[2] = s->c_shadow;
    if (reset) s->c_isect = s->c_shade = s->c_shadow = 0;
}
/* Light tracer (orc_ptracer_render): how many intersections the reference's loop computed and never read (ptracer.cpp:264-269) -- NOT part of
   c_isect, so that c_isect is what an implementation that skips them performs. */
ORC_API uint64_t orc_dead_intersections(OScene *s, int reset) { uint64_t v = s->c_isect_dead; if (reset) s->c_isect_dead = 0; return v; }

/* ----------------------------------------------------------------------------------------
 * rays and intersections
 * ---------------------------------------------------------------------------------------- */
typedef struct { v3 o, d; float maxt; } Ray;
typedef struct { float t; v2 prim_uv; uint32_t prim_index, shape_index; } PI;   /* interaction.h:570-675 */
typedef struct {
    float t; v3 p, n; v2 uv; v3 sh_s, sh_t, sh_n; v3 dp_du, dp_dv; v3 wi;
    uint32_t prim_index; int32_t shape;                    /* shape = -1 when invalid */
} SI;                                                       /* interaction.h:171-492 */

static inline v3 ray_at(const Ray *r, float t) { return fma3s(r->d, t, r->o); }   /* ray.h:61 */

/* Rectangle::ray_intersect_preliminary_impl: src/shapes/rectangle.cpp:446-468 */
static inline int rect_intersect(const OShape *sh, const Ray *ray, float *t_out, v2 *uv_out) {
    v3 o = xf_point(sh->to_object, ray->o), d = xf_vector(sh->to_object, ray->d);
    float t = -o.z / d.z;
    v3 local = fma3s(d, t, o);
    int hit = t >= 0.f && t <= ray->maxt && fabsf(local.x) <= 1.f && fabsf(local.y) <= 1.f;
    *t_out = t; uv_out->x = local.x; uv_out->y = local.y;
    return hit;
}
/* Disk::ray_intersect_preliminary_impl / ray_test_impl: src/shapes/disk.cpp:392-431 */
static inline int disk_intersect(const OShape *sh, const Ray *ray, float *t_out, v2 *uv_out) {
    v3 o = xf_point(sh->to_object, ray->o), d = xf_vector(sh->to_object, ray->d);
    float t = -o.z / d.z;
    v3 local = fma3s(d, t, o);
    int hit = t >= 0.f && t <= ray->maxt && local.x * local.x + local.y * local.y <= 1.f;
    *t_out = t; uv_out->x = local.x; uv_out->y = local.y;
    return hit;
}
static inline v3 mesh_vertex(const OShape *sh, uint32_t i) { return V3(sh->pos[3*i], sh->pos[3*i+1], sh->pos[3*i+2]); }
/* Watertight ray / triangle test for the geometric tracer (orc_trace).  The callers of prb_basic.trace aim rays EXACTLY at mesh vertices and
   along mesh edges -- a field's central ray goes to the entrance-pupil position, which is the pole vertex of the front lens mesh
   (design_problem.py:152-153, :183-189), and the tracer check sends its rays in the symmetry plane y = 0 of the icosphere (:442-453) --
   and the reference intersects meshes through Embree / OptiX, which do not leak at shared edges.  Moeller-Trumbore below does (its u, v
   are computed relative to each triangle's own first vertex: a ray on a shared edge can be marginally outside BOTH neighbours).  Pluecker
   edge functions (Embree's PlueckerIntersector): for the edge (P, Q) f = dot(cross(P - Q, P + Q), d) with P, Q relative to the ray origin;
   the neighbour sees the same edge as (Q, P) and gets exactly -f in floating point (negation and commutation are exact, rounding is
   sign-symmetric), so a ray is inside at least one of the two.  u, v as mesh.h:430-453 defines them (weights of p1, p2). */
static int g_watertight = 0;   /* set by orc_trace only, which is serial: not thread-local, the per-triangle read stays a plain load */
static __attribute__((noinline)) int pluecker_watertight(const Ray *ray, v3 p0, v3 p1, v3 p2, float *t_out, v2 *uv_out) {
    v3 v0 = sub3(p0, ray->o), v1 = sub3(p1, ray->o), v2_ = sub3(p2, ray->o);
    v3 e0 = sub3(v2_, v0), e1 = sub3(v0, v1), e2 = sub3(v1, v2_);
    float U = dot3(cross3(e0, add3(v2_, v0)), ray->d), V = dot3(cross3(e1, add3(v0, v1)), ray->d), W = dot3(cross3(e2, add3(v1, v2_)), ray->d);
    float UVW = U + V + W, eps = 5.9604644775390625e-8f * fabsf(UVW);
    float mn = fminf(fminf(U, V), W), mx = fmaxf(fmaxf(U, V), W);
    int active = (mn >= -eps || mx <= eps) && UVW != 0.f;
    v3 ng = cross3(sub3(v1, v0), sub3(v2_, v0));
    float den = dot3(ng, ray->d), t = dot3(v0, ng) / den;
    active = active && den != 0.f && t >= 0.f && t <= ray->maxt;
    float inv = 1.f / UVW;
    *t_out = t; uv_out->x = fminf(fmaxf(U * inv, 0.f), 1.f); uv_out->y = fminf(fmaxf(V * inv, 0.f), 1.f);
    return active;
}
/* Mesh::moeller_trumbore: include/mitsuba/render/mesh.h:430-453 */
static inline int moeller_trumbore(const Ray *ray, v3 p0, v3 p1, v3 p2, float *t_out, v2 *uv_out) {
    if (__builtin_expect(g_watertight, 0)) return pluecker_watertight(ray, p0, p1, p2, t_out, uv_out);
    v3 e1 = sub3(p1, p0), e2 = sub3(p2, p0);
    v3 pvec = cross3(ray->d, e2);
    float inv_det = rcpf_(dot3(e1, pvec));
    v3 tvec = sub3(ray->o, p0);
    float u = dot3(tvec, pvec) * inv_det;
    int active = u >= 0.f && u <= 1.f;
    v3 qvec = cross3(tvec, e1);
    float v = dot3(ray->d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    float t = dot3(e2, qvec) * inv_det;
    active = active && t >= 0.f && t <= ray->maxt;
    *t_out = t; uv_out->x = u; uv_out->y = v;
    return active;
}

/* Acceleration of the oracle's own closest / any hit queries on large meshes (a median-split BVH; boxes are padded, so no hit
   is lost -- the result is the linear scan's closest hit, only equal-t ties may resolve to another face). */
static void orc_bvh_node(OShape *sh, uint32_t node, uint32_t begin, uint32_t end, float (*cent)[3]) {
    ONode *nd = &sh->bvh[node];
    float clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int k = 0; k < 3; ++k) { nd->lo[k] = INFINITY; nd->hi[k] = -INFINITY; }
    for (uint32_t i = begin; i < end; ++i) {
        const uint32_t *fi = sh->faces + 3 * sh->bvh_faces[i];
        for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) { float v = sh->pos[3 * fi[j] + k]; if (v < nd->lo[k]) nd->lo[k] = v; if (v > nd->hi[k]) nd->hi[k] = v; }
        for (int k = 0; k < 3; ++k) { float c = cent[sh->bvh_faces[i]][k]; if (c < clo[k]) clo[k] = c; if (c > chi[k]) chi[k] = c; }
    }
    for (int k = 0; k < 3; ++k) { float pad = 1e-5f * (1.f + fabsf(nd->lo[k]) + fabsf(nd->hi[k])); nd->lo[k] -= pad; nd->hi[k] += pad; }
    uint32_t n = end - begin;
    if (n <= 4) { nd->left = begin; nd->count = n; return; }
    int axis = 0; if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1; if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    float split = 0.5f * (clo[axis] + chi[axis]); uint32_t i = begin, j = end;
    while (i < j) { if (cent[sh->bvh_faces[i]][axis] < split) ++i; else { --j; uint32_t t = sh->bvh_faces[i]; sh->bvh_faces[i] = sh->bvh_faces[j]; sh->bvh_faces[j] = t; } }
    if (i == begin || i == end) i = begin + n / 2;
    uint32_t left = sh->n_nodes; sh->n_nodes += 2;
    nd->left = left; nd->count = 0;
    orc_bvh_node(sh, left, begin, i, cent); orc_bvh_node(sh, left + 1, i, end, cent);
}
static void orc_bvh_build(OShape *sh) {
    free(sh->bvh); free(sh->bvh_faces); sh->bvh = NULL; sh->bvh_faces = NULL; sh->n_nodes = 0;
    if (sh->type != MI355_SHAPE_MESH || sh->face_count <= ORC_BVH_MIN) return;
    sh->bvh = (ONode *) malloc(sizeof(ONode) * 2u * (size_t) sh->face_count); sh->bvh_faces = (uint32_t *) malloc(4u * (size_t) sh->face_count);
    float (*cent)[3] = (float (*)[3]) malloc(12u * (size_t) sh->face_count);
    for (uint32_t f = 0; f < sh->face_count; ++f) {
        sh->bvh_faces[f] = f; const uint32_t *fi = sh->faces + 3 * f;
        for (int k = 0; k < 3; ++k) cent[f][k] = (sh->pos[3 * fi[0] + k] + sh->pos[3 * fi[1] + k] + sh->pos[3 * fi[2] + k]) * (1.f / 3.f);
    }
    sh->n_nodes = 1; orc_bvh_node(sh, 0, 0, sh->face_count, cent);
    free(cent);
}
static inline int orc_box_hit(const ONode *nd, const Ray *ray, float maxt) {
    float tn = 0.f, tf = maxt;
    const float o[3] = { ray->o.x, ray->o.y, ray->o.z }, d[3] = { ray->d.x, ray->d.y, ray->d.z };
    for (int k = 0; k < 3; ++k) {
        if (d[k] == 0.f) { if (o[k] < nd->lo[k] || o[k] > nd->hi[k]) return 0; continue; }
        float inv = 1.f / d[k], t0 = (nd->lo[k] - o[k]) * inv, t1 = (nd->hi[k] - o[k]) * inv;
        if (t0 > t1) { float t = t0; t0 = t1; t1 = t; }
        if (t0 > tn) tn = t0; if (t1 < tf) tf = t1;
    }
    return tn <= tf * 1.0001f + 1e-6f;
}

/* math::solve_quadratic: include/mitsuba/core/math.h:360-402 */
static int solve_quadratic(float a, float b, float c, float *x0, float *x1) {
    int linear_case = a == 0.f, valid_linear = linear_case && b != 0.f;
    *x0 = *x1 = -c / b;
    float discrim = fmaf(b, b, -(4.f * a * c));
    int valid_quadratic = !linear_case && discrim >= 0.f;
    if (valid_quadratic) {
        float sqrt_discrim = sqrtf(discrim);
        float temp = -0.5f * (b + copysignf(sqrt_discrim, b));
        float x0p = temp / a, x1p = c / temp;
        *x0 = fminf(x0p, x1p); *x1 = fmaxf(x0p, x1p);
    }
    return valid_linear || valid_quadratic;
}
/* Sphere::ray_intersect_preliminary_impl: src/shapes/sphere.cpp:491-563 (float32 branch of the AD variants) */
static int sphere_intersect(const OShape *sh, const Ray *ray, float *t_out) {
    v3 l = sub3(ray->o, sh->center), d = ray->d;
    float plane_t = dot3(neg3(l), d) / norm3(d);
    int no_hit = plane_t == 0.f && (ray->o.x != sh->center.x && ray->o.y != sh->center.y && ray->o.z != sh->center.z);
    v3 plane_p = ray_at(ray, plane_t);
    no_hit = no_hit && (norm3(sub3(plane_p, sh->center)) > sh->radius);
    v3 o = sub3(plane_p, sh->center);
    float A = sqnorm3(d), B = 2.f * dot3(o, d), C = sqnorm3(o) - sh->radius * sh->radius;
    float near_t, far_t; int found = solve_quadratic(A, B, C, &near_t, &far_t);
    near_t += plane_t; far_t += plane_t;
    int out_bounds = !(near_t <= ray->maxt && far_t >= 0.f);
    int in_bounds = near_t < 0.f && far_t > ray->maxt;
    int active = found && !no_hit && !out_bounds && !in_bounds;
    *t_out = near_t < 0.f ? far_t : near_t;
    return active;
}
/* Sphere::ray_test_impl: sphere.cpp:565-601 */
static int sphere_test(const OShape *sh, const Ray *ray) {
    v3 o = sub3(ray->o, sh->center), d = ray->d;
    float A = sqnorm3(d), B = 2.f * dot3(o, d), C = sqnorm3(o) - sh->radius * sh->radius;
    float near_t, far_t; int found = solve_quadratic(A, B, C, &near_t, &far_t);
    int out_bounds = !(near_t <= ray->maxt && far_t >= 0.f);
    int in_bounds = near_t < 0.f && far_t > ray->maxt;
    return found && !out_bounds && !in_bounds;
}

/* Scene::ray_intersect_preliminary (closest hit): scene.cpp:192-199; traversal semantics of
   kdtree.h:2197-2293 (maxt shrinks to the closest hit so far) restated as a linear scan. */
static PI scene_intersect(OScene *sc, const Ray *ray_in) {
    Ray ray = *ray_in; PI pi; pi.t = INFINITY; pi.prim_uv.x = pi.prim_uv.y = 0; pi.prim_index = 0; pi.shape_index = 0xffffffffu;
    for (uint32_t i = 0; i < sc->n_shapes; ++i) {
        const OShape *sh = &sc->shapes[i];
        if (sh->type == MI355_SHAPE_RECTANGLE || sh->type == MI355_SHAPE_DISK) {
            float t; v2 uv;
            if ((sh->type == MI355_SHAPE_DISK ? disk_intersect(sh, &ray, &t, &uv) : rect_intersect(sh, &ray, &t, &uv)) && t < pi.t) { pi.t = t; pi.prim_uv = uv; pi.prim_index = 0; pi.shape_index = i; ray.maxt = t; }
        } else if (sh->type == MI355_SHAPE_SPHERE) {
            float t;
            if (sphere_intersect(sh, &ray, &t) && t < pi.t) { pi.t = t; pi.prim_uv.x = pi.prim_uv.y = 0.f; pi.prim_index = 0; pi.shape_index = i; ray.maxt = t; }
        } else if (sh->bvh) {
            uint32_t stack[64]; int sp = 0; stack[sp++] = 0;
            while (sp) {
                const ONode *nd = &sh->bvh[stack[--sp]];
                if (!orc_box_hit(nd, &ray, ray.maxt)) continue;
                if (!nd->count) { if (sp < 62) { stack[sp++] = nd->left; stack[sp++] = nd->left + 1; } continue; }
                for (uint32_t q = 0; q < nd->count; ++q) {
                    const uint32_t f = sh->bvh_faces[nd->left + q]; const uint32_t *fi = sh->faces + 3 * f; float t; v2 uv;
                    if (moeller_trumbore(&ray, mesh_vertex(sh, fi[0]), mesh_vertex(sh, fi[1]), mesh_vertex(sh, fi[2]), &t, &uv) && (t < pi.t || (t == pi.t && pi.shape_index == i && f < pi.prim_index))) {
                        pi.t = t; pi.prim_uv = uv; pi.prim_index = f; pi.shape_index = i; ray.maxt = t;
                    }
                }
            }
        } else {
            for (uint32_t f = 0; f < sh->face_count; ++f) {
                const uint32_t *fi = sh->faces + 3 * f; float t; v2 uv;
                if (moeller_trumbore(&ray, mesh_vertex(sh, fi[0]), mesh_vertex(sh, fi[1]), mesh_vertex(sh, fi[2]), &t, &uv) && t < pi.t) {
                    pi.t = t; pi.prim_uv = uv; pi.prim_index = f; pi.shape_index = i; ray.maxt = t;
                }
            }
        }
    }
    return pi;
}
/* Scene::ray_test (any hit): scene.cpp:201-210 */
static int scene_ray_test(OScene *sc, const Ray *ray) {
    for (uint32_t i = 0; i < sc->n_shapes; ++i) {
        const OShape *sh = &sc->shapes[i]; float t; v2 uv;
        if (sh->type == MI355_SHAPE_RECTANGLE) { if (rect_intersect(sh, ray, &t, &uv)) return 1; }
        else if (sh->type == MI355_SHAPE_DISK) { if (disk_intersect(sh, ray, &t, &uv)) return 1; }
        else if (sh->type == MI355_SHAPE_SPHERE) { if (sphere_test(sh, ray)) return 1; }
        else if (sh->bvh) {
            uint32_t stack[64]; int sp = 0; stack[sp++] = 0;
            while (sp) {
                const ONode *nd = &sh->bvh[stack[--sp]];
                if (!orc_box_hit(nd, ray, ray->maxt)) continue;
                if (!nd->count) { if (sp < 62) { stack[sp++] = nd->left; stack[sp++] = nd->left + 1; } continue; }
                for (uint32_t q = 0; q < nd->count; ++q) {
                    const uint32_t *fi = sh->faces + 3 * sh->bvh_faces[nd->left + q];
                    if (moeller_trumbore(ray, mesh_vertex(sh, fi[0]), mesh_vertex(sh, fi[1]), mesh_vertex(sh, fi[2]), &t, &uv)) return 1;
                }
            }
        }
        else for (uint32_t f = 0; f < sh->face_count; ++f) {
            const uint32_t *fi = sh->faces + 3 * f;
            if (moeller_trumbore(ray, mesh_vertex(sh, fi[0]), mesh_vertex(sh, fi[1]), mesh_vertex(sh, fi[2]), &t, &uv)) return 1;
        }
    }
    return 0;
}

/* PreliminaryIntersection::compute_surface_interaction (interaction.h:642-668) ->
   Rectangle::compute_surface_interaction (rectangle.cpp:497-567, IsDiff branch of the
   llvm_ad_rgb target: si.p = ray(t), no re-projection) / Mesh::compute_surface_interaction
   (mesh.cpp:1375-1548) -> finalize_surface_interaction (interaction.h:464-484) */
static SI compute_si(const OScene *sc, const Ray *ray, const PI *pi) {
    SI si; memset(&si, 0, sizeof(si)); si.t = INFINITY; si.shape = -1;
    if (!(pi->t != INFINITY)) { si.wi = neg3(ray->d); return si; }
    const OShape *sh = &sc->shapes[pi->shape_index];
    si.t = pi->t; si.shape = (int32_t) pi->shape_index; si.prim_index = pi->prim_index;
    if (sh->type == MI355_SHAPE_DISK) {
        /* Disk::compute_surface_interaction, disk.cpp:436-519 (IsDiff && !FollowShape: si.p = ray(t)) */
        si.p = ray_at(ray, pi->t);
        float px = pi->prim_uv.x, py = pi->prim_uv.y, r = sqrtf(px * px + py * py), inv_r = 1.f / r;
        float v = atan2f(py, px) * (0.5f * ORC_INV_PI); if (v < 0.f) v += 1.f;
        si.uv.x = r; si.uv.y = v;
        float cos_phi = r != 0.f ? px * inv_r : 1.f, sin_phi = r != 0.f ? py * inv_r : 0.f;
        si.dp_du = xf_vector(sh->to_world, V3(cos_phi, sin_phi, 0.f)); si.dp_dv = xf_vector(sh->to_world, V3(-sin_phi, cos_phi, 0.f));
        si.n = sh->frame_n; si.sh_n = sh->frame_n;
    } else if (sh->type == MI355_SHAPE_RECTANGLE) {
        si.p = ray_at(ray, pi->t);
        si.n = sh->frame_n; si.sh_n = sh->frame_n; si.dp_du = sh->frame_s; si.dp_dv = sh->frame_t;
        si.uv.x = fmaf(pi->prim_uv.x, .5f, .5f); si.uv.y = fmaf(pi->prim_uv.y, .5f, .5f);
    } else if (sh->type == MI355_SHAPE_SPHERE) {
        /* Sphere::compute_surface_interaction, sphere.cpp:609-712, IsDiff && !FollowShape branch */
        si.p = ray_at(ray, pi->t);
        si.sh_n = normalize3(sub3(si.p, sh->center));
        v3 local = xf_point(sh->to_object, si.p);
        float rd_2 = local.x * local.x + local.y * local.y;
        /* dr::unit_angle_z [Dr.Jit, recalled]: robust acos(z) */
        v3 dz = V3(local.x, local.y, local.z - mulsign(1.f, local.z));
        float temp = 2.f * asinf(.5f * norm3(dz));
        float theta = local.z >= 0.f ? temp : ORC_PI - temp;
        float phi = atan2f(local.y, local.x); if (phi < 0.f) phi += 2.f * ORC_PI;
        si.uv.x = phi * (0.5f * ORC_INV_PI); si.uv.y = theta * ORC_INV_PI;
        si.dp_du = V3(-local.y, local.x, 0.f);
        float rd = sqrtf(rd_2), inv_rd = rcpf_(rd), cos_phi = local.x * inv_rd, sin_phi = local.y * inv_rd;
        si.dp_dv = V3(local.z * cos_phi, local.z * sin_phi, -rd);
        if (rd == 0.f) si.dp_dv = V3(1.f, 0.f, 0.f);
        si.dp_du = scale3(xf_vector(sh->to_world, si.dp_du), 2.f * ORC_PI);
        si.dp_dv = scale3(xf_vector(sh->to_world, si.dp_dv), ORC_PI);
        if (sh->flip_normals) si.sh_n = neg3(si.sh_n);
        si.n = si.sh_n;
    } else {
        const uint32_t *fi = sh->faces + 3 * pi->prim_index;
        v3 p0 = mesh_vertex(sh, fi[0]), p1 = mesh_vertex(sh, fi[1]), p2 = mesh_vertex(sh, fi[2]);
        float b1 = pi->prim_uv.x, b2 = pi->prim_uv.y, b0 = 1.f - b1 - b2;
        si.p = fma3s(p0, b0, fma3s(p1, b1, scale3(p2, b2)));
        si.n = normalize3(cross3(sub3(p1, p0), sub3(p2, p0)));          /* mesh.h:120-128 */
        si.uv.x = b1; si.uv.y = b2;
        coordinate_system(si.n, &si.dp_du, &si.dp_dv);
        v3 dp0 = sub3(p1, p0), dp1 = sub3(p2, p0);
        if (sh->uv) {
            v2 uv0 = { sh->uv[2*fi[0]], sh->uv[2*fi[0]+1] }, uv1 = { sh->uv[2*fi[1]], sh->uv[2*fi[1]+1] },
               uv2 = { sh->uv[2*fi[2]], sh->uv[2*fi[2]+1] };
            si.uv.x = fmaf(uv2.x, b2, fmaf(uv1.x, b1, uv0.x * b0));
            si.uv.y = fmaf(uv2.y, b2, fmaf(uv1.y, b1, uv0.y * b0));
            v2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
            float det = fmaf(duv0.x, duv1.y, -(duv0.y * duv1.x)), inv_det = rcpf_(det);
            if (det != 0.f) {
                si.dp_du = scale3(V3(fmaf(duv1.y, dp0.x, -(duv0.y * dp1.x)), fmaf(duv1.y, dp0.y, -(duv0.y * dp1.y)), fmaf(duv1.y, dp0.z, -(duv0.y * dp1.z))), inv_det);
                si.dp_dv = scale3(V3(fmaf(-duv1.x, dp0.x, duv0.x * dp1.x), fmaf(-duv1.x, dp0.y, duv0.x * dp1.y), fmaf(-duv1.x, dp0.z, duv0.x * dp1.z)), inv_det);
            }
        }
        if (sh->nrm) {
            v3 n0 = V3(sh->nrm[3*fi[0]], sh->nrm[3*fi[0]+1], sh->nrm[3*fi[0]+2]),
               n1 = V3(sh->nrm[3*fi[1]], sh->nrm[3*fi[1]+1], sh->nrm[3*fi[1]+2]),
               n2 = V3(sh->nrm[3*fi[2]], sh->nrm[3*fi[2]+1], sh->nrm[3*fi[2]+2]);
            v3 n = fma3s(n2, b2, fma3s(n1, b1, scale3(n0, b0)));
            float il = rsqrtf_(sqnorm3(n));
            si.sh_n = scale3(n, il);
        } else si.sh_n = si.n;
        if (sh->flip_normals) { si.n = neg3(si.n); si.sh_n = neg3(si.sh_n); }
    }
    /* initialize_sh_frame: interaction.h:245-255 */
    si.sh_s = normalize3(fma3s(si.sh_n, -dot3(si.sh_n, si.dp_du), si.dp_du));
    if (si.dp_du.x == 0.f && si.dp_du.y == 0.f && si.dp_du.z == 0.f) { v3 tt; coordinate_system(si.sh_n, &si.sh_s, &tt); }
    si.sh_t = cross3(si.sh_n, si.sh_s);
    v3 md = neg3(ray->d);
    si.wi = V3(dot3(md, si.sh_s), dot3(md, si.sh_t), dot3(md, si.sh_n));           /* frame.h:33-35 */
    return si;
}
static inline v3 si_to_world(const SI *si, v3 v) { return fma3s(si->sh_n, v.z, fma3s(si->sh_t, v.y, scale3(si->sh_s, v.x))); } /* frame.h:38-40 */
static inline v3 si_to_local(const SI *si, v3 v) { return V3(dot3(v, si->sh_s), dot3(v, si->sh_t), dot3(v, si->sh_n)); }
/* Interaction::offset_p: interaction.h:158-162 */
static inline v3 offset_p(v3 p, v3 n, v3 d) {
    float mag = (1.f + fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z))) * RAY_EPS;
    mag = mulsign(mag, dot3(n, d));
    return fma3s(n, mag, p);
}
/* Interaction::spawn_ray: interaction.h:133-135 */
static inline Ray spawn_ray(v3 p, v3 n, v3 d) { Ray r; r.o = offset_p(p, n, d); r.d = d; r.maxt = ORC_LARGEST; return r; }
/* Interaction::spawn_ray_to: interaction.h:138-145 */
static inline Ray spawn_ray_to(v3 p, v3 n, v3 target) {
    Ray r; r.o = offset_p(p, n, sub3(target, p));
    v3 d = sub3(target, r.o); float dist = norm3(d);
    r.d = scale3(d, rcpf_(dist)); r.maxt = dist * (1.f - SHADOW_EPS); return r;
}

/* ----------------------------------------------------------------------------------------
 * textures / BSDF / emitter
 * ---------------------------------------------------------------------------------------- */
static inline int32_t wrap_i(int32_t p, int32_t n, uint32_t mode) {
    if (mode == MI355_WRAP_CLAMP) return p < 0 ? 0 : (p > n - 1 ? n - 1 : p);
    if (mode == MI355_WRAP_REPEAT) { int32_t m = p % n; return m < 0 ? m + n : m; }
    int32_t m = p % (2 * n); if (m < 0) m += 2 * n; return m < n ? m : 2 * n - 1 - m;
}
/* BitmapTexture::interpolate_3 (bitmap.cpp:696-710) -> dr::Texture<Float,2>::eval_nonaccel
   [Dr.Jit, recalled]: pos = uv*res - 0.5, 4-tap lerp; tensor is (H, W, C), uv.x <-> W.
   Optionally returns the 4 tap indices/weights for the adjoint scatter. */
static v3 texture_eval(const OTexture *t, v2 uv_in, int32_t idx[4], float wgt[4]) {
    v2 uv = { fmaf(t->to_uv[0], uv_in.x, fmaf(t->to_uv[1], uv_in.y, t->to_uv[2])),
              fmaf(t->to_uv[3], uv_in.x, fmaf(t->to_uv[4], uv_in.y, t->to_uv[5])) };
    int32_t W = (int32_t) t->w, H = (int32_t) t->h, C = (int32_t) t->c;
    float out[3] = { 0, 0, 0 };
    if (t->filter == MI355_FILTER_NEAREST) {
        int32_t x = wrap_i((int32_t) floorf(uv.x * (float) W), W, t->wrap), y = wrap_i((int32_t) floorf(uv.y * (float) H), H, t->wrap);
        for (int c = 0; c < 3; ++c) out[c] = t->data[(y * W + x) * C + (C == 3 ? c : 0)];
        if (idx) { idx[0] = y * W + x; wgt[0] = 1.f; idx[1] = idx[2] = idx[3] = -1; wgt[1] = wgt[2] = wgt[3] = 0.f; }
        return V3(out[0], out[1], out[2]);
    }
    float px = fmaf(uv.x, (float) W, -.5f), py = fmaf(uv.y, (float) H, -.5f);
    float fx = floorf(px), fy = floorf(py);
    int32_t ix = (int32_t) fx, iy = (int32_t) fy;
    float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    int32_t x0 = wrap_i(ix, W, t->wrap), x1 = wrap_i(ix + 1, W, t->wrap), y0 = wrap_i(iy, H, t->wrap), y1 = wrap_i(iy + 1, H, t->wrap);
    int32_t i00 = y0 * W + x0, i10 = y0 * W + x1, i01 = y1 * W + x0, i11 = y1 * W + x1;
    for (int c = 0; c < 3; ++c) {
        int cc = C == 3 ? c : 0;
        float v00 = t->data[i00 * C + cc], v10 = t->data[i10 * C + cc], v01 = t->data[i01 * C + cc], v11 = t->data[i11 * C + cc];
        out[c] = fmaf(w0y, fmaf(w0x, v00, w1x * v10), w1y * fmaf(w0x, v01, w1x * v11));
    }
    if (idx) { idx[0] = i00; idx[1] = i10; idx[2] = i01; idx[3] = i11;
               wgt[0] = w0y * w0x; wgt[1] = w0y * w1x; wgt[2] = w1y * w0x; wgt[3] = w1y * w1x; }
    return V3(out[0], out[1], out[2]);
}
static inline v3 bsdf_reflectance(const OScene *sc, const mi355_bsdf_desc *b, const SI *si) {
    if (b->reflectance_texture >= 0) return texture_eval(&sc->textures[b->reflectance_texture], si->uv, NULL, NULL);
    return V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]);
}
/* SmoothDiffuse::eval_pdf: src/bsdfs/diffuse.cpp:160-180 */
static void diffuse_eval_pdf(v3 refl, v3 wi, v3 wo, int active, v3 *val, float *pdf) {
    float ci = wi.z, co = wo.z;
    active = active && ci > 0.f && co > 0.f;
    if (active) { *val = scale3(scale3(refl, ORC_INV_PI), co); *pdf = ORC_INV_PI * co; }
    else { *val = V3(0, 0, 0); *pdf = 0.f; }
}
/* SmoothDiffuse::sample: diffuse.cpp:101-125 (JIT semantics: no early-out, weight masked) */
static void diffuse_sample(v3 refl, v3 wi, v2 s2, int active, v3 *wo, float *pdf, v3 *weight) {
    active = active && wi.z > 0.f;
    *wo = square_to_cosine_hemisphere(s2);
    *pdf = ORC_INV_PI * wo->z;
    *weight = (active && *pdf > 0.f) ? refl : V3(0, 0, 0);
}
ORC_API void orc_diffuse_eval_pdf(const float *refl, const float *wi, const float *wo, float *val3, float *pdf) {
    v3 v; diffuse_eval_pdf(V3(refl[0], refl[1], refl[2]), V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]), 1, &v, pdf);
    val3[0] = v.x; val3[1] = v.y; val3[2] = v.z;
}
ORC_API void orc_diffuse_sample(const float *refl, const float *wi, const float *s2, float *wo3, float *pdf, float *w3) {
    v3 wo, w; v2 s = { s2[0], s2[1] };
    diffuse_sample(V3(refl[0], refl[1], refl[2]), V3(wi[0], wi[1], wi[2]), s, 1, &wo, pdf, &w);
    wo3[0] = wo.x; wo3[1] = wo.y; wo3[2] = wo.z; w3[0] = w.x; w3[1] = w.y; w3[2] = w.z;
}

/* ----------------------------------------------------------------------------------------
 * microfacet distribution (include/mitsuba/render/microfacet.h) and Fresnel (fresnel.h)
 * erf / erfinv: Dr.Jit polynomial approximations are absent from the tree -> libm erff and Giles' single-precision
 * erfinv ("Approximating the erfinv function", 2010) [parity unpinned]
 * ---------------------------------------------------------------------------------------- */
static float erfinv_(float x) {
    float w = -logf((1.0f - x) * (1.0f + x)), p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f; p = fmaf(p, w, 3.43273939e-07f); p = fmaf(p, w, -3.5233877e-06f); p = fmaf(p, w, -4.39150654e-06f);
        p = fmaf(p, w, 0.00021858087f); p = fmaf(p, w, -0.00125372503f); p = fmaf(p, w, -0.00417768164f); p = fmaf(p, w, 0.246640727f);
        p = fmaf(p, w, 1.50140941f);
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f; p = fmaf(p, w, 0.000100950558f); p = fmaf(p, w, 0.00134934322f); p = fmaf(p, w, -0.00367342844f);
        p = fmaf(p, w, 0.00573950773f); p = fmaf(p, w, -0.0076224613f); p = fmaf(p, w, 0.00943887047f); p = fmaf(p, w, 1.00167406f);
        p = fmaf(p, w, 2.83297682f);
    }
    return p * x;
}
typedef struct { uint32_t type; float au, av; int sample_visible; } MF;      /* microfacet.h:63-180 */
static MF mf_make(uint32_t type, float au, float av, int sv) { MF m = { type, fmaxf(au, 1e-4f), fmaxf(av, 1e-4f), sv }; return m; }  /* configure(), :424-427 */
static inline float sqr_(float x) { return x * x; }
static inline float safe_sqrt_(float x) { return sqrtf(fmaxf(x, 0.f)); }
/* MicrofacetDistribution::eval, microfacet.h:185-207 */
static float mf_eval(const MF *d, v3 m) {
    float alpha_uv = d->au * d->av, ct = m.z, ct2 = ct * ct, result;
    if (d->type == 0)
        result = expf(-(sqr_(m.x / d->au) + sqr_(m.y / d->av)) / ct2) / (ORC_PI * alpha_uv * sqr_(ct2));
    else
        result = rcpf_(ORC_PI * alpha_uv * sqr_(sqr_(m.x / d->au) + sqr_(m.y / d->av) + sqr_(m.z)));
    return result * ct > 1e-20f ? result : 0.f;
}
/* smith_g1, microfacet.h:341-365 */
static float mf_smith_g1(const MF *d, v3 v, v3 m) {
    float xy_alpha_2 = sqr_(d->au * v.x) + sqr_(d->av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr_(v.z), result;
    if (d->type == 0) {
        float a = rsqrtf_(tan_theta_alpha_2), a_sqr = a * a;
        result = a >= 1.6f ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else result = 2.f / (1.f + sqrtf(1.f + tan_theta_alpha_2));
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot3(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
/* pdf, microfacet.h:219-228 */
static float mf_pdf(const MF *d, v3 wi, v3 m) {
    float result = mf_eval(d, m);
    if (d->sample_visible) result *= mf_smith_g1(d, wi, m) * fabsf(dot3(wi, m)) / wi.z;
    else result *= m.z;
    return result;
}
/* sample_visible_11, microfacet.h:368-421 */
static v2 mf_sample_visible_11(const MF *d, float cos_theta_i, v2 sample) {
    v2 r;
    if (d->type == 0) {
        float tan_theta_i = safe_sqrt_(fmaf(-cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i, cot_theta_i = rcpf_(tan_theta_i);
        float maxval = erff(cot_theta_i);
        sample.x = fmaxf(fminf(sample.x, 1.f - 1e-6f), 1e-6f); sample.y = fmaxf(fminf(sample.y, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * erff(sqrtf(-logf(sample.x)));
        const float inv_sqrt_pi = 0.56418958354775628695f;
        sample.x *= 1.f + maxval + inv_sqrt_pi * tan_theta_i * expf(-sqr_(cot_theta_i));
        for (int i = 0; i < 3; ++i) {
            float slope = erfinv_(x), value = 1.f + x + inv_sqrt_pi * tan_theta_i * expf(-sqr_(slope)) - sample.x,
                  derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        r.x = erfinv_(x); r.y = erfinv_(fmaf(2.f, sample.y, -1.f));
    } else {
        v2 p = square_to_uniform_disk_concentric(sample);
        float sfac = 0.5f * (1.f + cos_theta_i);
        float a = safe_sqrt_(1.f - sqr_(p.x));
        p.y = fmaf(p.y - a, sfac, a);                                   /* dr::lerp(a, b, t) = fmadd(b - a, t, a) [recalled] */
        float x = p.x, y = p.y, z = safe_sqrt_(1.f - fmaf(p.y, p.y, p.x * p.x));
        float sin_theta_i = safe_sqrt_(1.f - sqr_(cos_theta_i));
        float norm = rcpf_(fmaf(sin_theta_i, y, cos_theta_i * z));
        r.x = fmaf(cos_theta_i, y, -(sin_theta_i * z)) * norm; r.y = x * norm;
    }
    return r;
}
/* sample, microfacet.h:244-326 */
static v3 mf_sample(const MF *d, v3 wi, v2 sample, float *pdf) {
    if (!d->sample_visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (d->au == d->av) { float ph = (2.f * ORC_PI) * sample.y; sin_phi = sinf(ph); cos_phi = cosf(ph); alpha_2 = d->au * d->au; }
        else {
            float ratio = d->av / d->au, tmp = ratio * tanf((2.f * ORC_PI) * sample.y);
            cos_phi = rsqrtf_(fmaf(tmp, tmp, 1.f)); cos_phi = mulsign(cos_phi, fabsf(sample.y - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = rcpf_(sqr_(cos_phi / d->au) + sqr_(sin_phi / d->av));
        }
        if (d->type == 0) {
            cos_theta = rsqrtf_(fmaf(-alpha_2, logf(1.f - sample.x), 1.f)); cos_theta_2 = cos_theta * cos_theta;
            float cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            *pdf = (1.f - sample.x) / (ORC_PI * d->au * d->av * cos_theta_3);
        } else {
            float tan_theta_m_2 = alpha_2 * sample.x / (1.f - sample.x);
            cos_theta = rsqrtf_(1.f + tan_theta_m_2); cos_theta_2 = cos_theta * cos_theta;
            float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            *pdf = rcpf_(ORC_PI * d->au * d->av * cos_theta_3 * sqr_(temp));
        }
        float sin_theta = sqrtf(1.f - cos_theta_2);
        return V3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    }
    v3 wi_p = normalize3(V3(d->au * wi.x, d->av * wi.y, wi.z));
    /* Frame3f::sincos_phi, frame.h:111-122 */
    float st2 = fmaf(wi_p.x, wi_p.x, wi_p.y * wi_p.y), ist = rsqrtf_(st2);
    float cos_phi = fminf(fmaxf(wi_p.x * ist, -1.f), 1.f), sin_phi = fminf(fmaxf(wi_p.y * ist, -1.f), 1.f);
    if (fabsf(st2) <= 4.f * ORC_EPS) { cos_phi = 1.f; sin_phi = 0.f; }
    v2 slope = mf_sample_visible_11(d, wi_p.z, sample);
    float sx = fmaf(cos_phi, slope.x, -(sin_phi * slope.y)) * d->au, sy = fmaf(sin_phi, slope.x, cos_phi * slope.y) * d->av;
    v3 m = normalize3(V3(-sx, -sy, 1.f));
    *pdf = mf_eval(d, m) * mf_smith_g1(d, wi, m) * fabsf(dot3(wi, m)) / wi.z;
    return m;
}
/* fresnel(), include/mitsuba/render/fresnel.h:34-72 */
static void fresnel_(float cos_theta_i, float eta, float *r_out, float *cos_theta_t_out, float *eta_it_out, float *eta_ti_out) {
    int outside = cos_theta_i >= 0.f;
    float rcp_eta = rcpf_(eta), eta_it = outside ? eta : rcp_eta, eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fmaf(-fmaf(-cos_theta_i, cos_theta_i, 1.f), eta_ti * eta_ti, 1.f);
    float cia = fabsf(cos_theta_i), cta = safe_sqrt_(cos_theta_t_sqr);
    int index_matched = eta == 1.f, special = index_matched || cia == 0.f;
    float r_sc = index_matched ? 0.f : 1.f;
    float a_s = fmaf(-eta_it, cta, cia) / fmaf(eta_it, cta, cia);
    float a_p = fmaf(-eta_it, cia, cta) / fmaf(eta_it, cia, cta);
    float r = 0.5f * (a_s * a_s + a_p * a_p);
    if (special) r = r_sc;
    *r_out = r; *cos_theta_t_out = mulsign(-cta, cos_theta_i); *eta_it_out = eta_it; *eta_ti_out = eta_ti;
}
/* fresnel_conductor(), fresnel.h:92-118 */
static float fresnel_conductor_(float cos_theta_i, float eta_r, float eta_i) {
    float ci2 = cos_theta_i * cos_theta_i, si2 = 1.f - ci2, si4 = si2 * si2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - si2, a_2_pb_2 = safe_sqrt_(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a = safe_sqrt_(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + ci2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * ci2 + si4, term_4 = term_2 * si2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}
ORC_API void orc_fresnel(float cos_theta_i, float eta, float *out4) { fresnel_(cos_theta_i, eta, &out4[0], &out4[1], &out4[2], &out4[3]); }
ORC_API float orc_fresnel_conductor(float cos_theta_i, float eta_r, float eta_i) { return fresnel_conductor_(cos_theta_i, eta_r, eta_i); }
ORC_API void orc_microfacet(uint32_t type, float au, float av, int sample_visible, const float *wi3, const float *v3s, const float *samples2, int n,
                            float *eval_out, float *pdf_out, float *g1_out, float *sample_m3, float *sample_pdf) {
    MF d = mf_make(type, au, av, sample_visible); v3 wi = V3(wi3[0], wi3[1], wi3[2]);
    for (int i = 0; i < n; ++i) {
        if (v3s) { v3 v = V3(v3s[3*i], v3s[3*i+1], v3s[3*i+2]);
            if (eval_out) eval_out[i] = mf_eval(&d, v); if (pdf_out) pdf_out[i] = mf_pdf(&d, wi, v); if (g1_out) g1_out[i] = mf_smith_g1(&d, v, wi); }
        if (samples2) { v2 sm = { samples2[2*i], samples2[2*i+1] }; float pdf; v3 m = mf_sample(&d, wi, sm, &pdf);
            sample_m3[3*i] = m.x; sample_m3[3*i+1] = m.y; sample_m3[3*i+2] = m.z; sample_pdf[i] = pdf; }
    }
}

/* ---- generic BSDF interface: BSDF::eval_pdf / sample (bsdf.h:371-470) dispatched on the plugin type ---- */
typedef struct { v3 wo; float pdf, eta; int delta; } BSample;
static int bsdf_is_smooth(const mi355_bsdf_desc *b) { return b->type != MI355_BSDF_DIELECTRIC && b->type != MI355_BSDF_CONDUCTOR; }   /* BSDFFlags::Smooth, bsdf.h:96 */
/* RoughConductor::eval_pdf, roughconductor.cpp:428-517 */
static void roughconductor_eval_pdf(const mi355_bsdf_desc *b, v3 wi, v3 wo, int active, v3 *val, float *pdf) {
    *val = V3(0, 0, 0); *pdf = 0.f;
    v3 H = normalize3(add3(wo, wi));
    active = active && wi.z > 0.f && wo.z > 0.f && dot3(wi, H) > 0.f && dot3(wo, H) > 0.f;
    if (!active) return;
    MF d = mf_make(b->distribution, b->alpha_u, b->alpha_v, (int) b->sample_visible);
    float D = mf_eval(&d, H);
    if (D == 0.f) return;
    float g1_wi = mf_smith_g1(&d, wi, H), G = g1_wi * mf_smith_g1(&d, wo, H);
    float value = D * G / (4.f * wi.z);
    float ct = dot3(wi, H);
    v3 F = V3(fresnel_conductor_(ct, b->eta[0], b->k[0]), fresnel_conductor_(ct, b->eta[1], b->k[1]), fresnel_conductor_(ct, b->eta[2], b->k[2]));
    v3 sr = V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]);
    *val = mul3(F, scale3(sr, value));
    *pdf = d.sample_visible ? D * g1_wi / (4.f * wi.z) : mf_pdf(&d, wi, H) / (4.f * dot3(wo, H));
}
/* RoughConductor::sample, roughconductor.cpp:226-312 */
static void roughconductor_sample(const mi355_bsdf_desc *b, v3 wi, v2 s2, int active, BSample *bs, v3 *weight) {
    memset(bs, 0, sizeof(*bs)); *weight = V3(0, 0, 0);
    active = active && wi.z > 0.f;
    MF d = mf_make(b->distribution, b->alpha_u, b->alpha_v, (int) b->sample_visible);
    float pdf; v3 m = mf_sample(&d, wi, s2, &pdf);
    bs->wo = sub3(scale3(m, 2.f * dot3(wi, m)), wi);                     /* reflect(wi, m), fresnel.h:281-284 */
    bs->eta = 1.f; bs->delta = 0;
    active = active && pdf != 0.f && bs->wo.z > 0.f;
    float w = d.sample_visible ? mf_smith_g1(&d, bs->wo, m) : mf_smith_g1(&d, wi, m) * mf_smith_g1(&d, bs->wo, m) * dot3(wi, m) / (wi.z * m.z);
    bs->pdf = pdf / (4.f * dot3(bs->wo, m));
    float ct = dot3(wi, m);
    v3 F = V3(fresnel_conductor_(ct, b->eta[0], b->k[0]), fresnel_conductor_(ct, b->eta[1], b->k[1]), fresnel_conductor_(ct, b->eta[2], b->k[2]));
    v3 sr = V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]);
    if (active) *weight = mul3(F, scale3(sr, w));
}
/* SmoothDielectric::sample, dielectric.cpp:246-371 (unpolarized, radiance transport); eval/pdf are zero (delta lobes) */
static int g_importance_mode = 0;   /* TransportMode::Importance for the particle tracer (set per thread region; the oracle's drivers never mix modes) */
#pragma omp threadprivate(g_importance_mode)
static void dielectric_sample(const mi355_bsdf_desc *b, v3 wi, float s1, int active, BSample *bs, v3 *weight) {
    float eta = b->int_ior / b->ext_ior;
    float r_i, cos_theta_t, eta_it, eta_ti; fresnel_(wi.z, eta, &r_i, &cos_theta_t, &eta_it, &eta_ti);
    float t_i = 1.f - r_i;
    int selected_r = s1 <= r_i && active;
    bs->pdf = selected_r ? r_i : t_i;
    bs->wo = selected_r ? V3(-wi.x, -wi.y, wi.z) : V3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t);
    bs->eta = selected_r ? 1.f : eta_it; bs->delta = 1;
    v3 w = selected_r ? V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]) : V3(b->transmittance[0], b->transmittance[1], b->transmittance[2]);
    if (!selected_r && !g_importance_mode) w = scale3(w, eta_ti * eta_ti);   /* radiance scaling (TransportMode::Radiance only), dielectric.cpp:361-366 */
    *weight = active ? w : V3(0, 0, 0);
}
/* SmoothConductor::sample, conductor.cpp:248-307 (unpolarized); eval / pdf are zero (delta lobe) */
static void conductor_sample(const mi355_bsdf_desc *b, v3 wi, int active, BSample *bs, v3 *weight) {
    active = active && wi.z > 0.f;
    bs->wo = V3(-wi.x, -wi.y, wi.z); bs->eta = 1.f; bs->pdf = 1.f; bs->delta = 1;         /* reflect(wi), fresnel.h:275-278 */
    v3 F = V3(fresnel_conductor_(wi.z, b->eta[0], b->k[0]), fresnel_conductor_(wi.z, b->eta[1], b->k[1]), fresnel_conductor_(wi.z, b->eta[2], b->k[2]));
    v3 sr = V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]);
    *weight = active ? mul3(sr, F) : V3(0, 0, 0);
    if (!active) { memset(bs, 0, sizeof(*bs)); }
}
static void bsdf_eval_pdf(const mi355_bsdf_desc *b, v3 refl, v3 wi, v3 wo, int active, v3 *val, float *pdf) {
    if (b->type == MI355_BSDF_DIFFUSE) diffuse_eval_pdf(refl, wi, wo, active, val, pdf);
    else if (b->type == MI355_BSDF_ROUGHCONDUCTOR) roughconductor_eval_pdf(b, wi, wo, active, val, pdf);
    else { *val = V3(0, 0, 0); *pdf = 0.f; }
}
static void bsdf_sample(const mi355_bsdf_desc *b, v3 refl, v3 wi, float s1, v2 s2, int active, BSample *bs, v3 *weight) {
    if (b->type == MI355_BSDF_DIFFUSE) { diffuse_sample(refl, wi, s2, active, &bs->wo, &bs->pdf, weight); bs->eta = 1.f; bs->delta = 0; }
    else if (b->type == MI355_BSDF_ROUGHCONDUCTOR) roughconductor_sample(b, wi, s2, active, bs, weight);
    else if (b->type == MI355_BSDF_CONDUCTOR) conductor_sample(b, wi, active, bs, weight);
    else dielectric_sample(b, wi, s1, active, bs, weight);
}
ORC_API void orc_bsdf_eval_pdf(const mi355_bsdf_desc *b, const float *wi, const float *wo, float *val3, float *pdf) {
    v3 v; v3 refl = V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]);
    bsdf_eval_pdf(b, refl, V3(wi[0], wi[1], wi[2]), V3(wo[0], wo[1], wo[2]), 1, &v, pdf); val3[0] = v.x; val3[1] = v.y; val3[2] = v.z;
}
ORC_API void orc_bsdf_sample(const mi355_bsdf_desc *b, const float *wi, float s1, const float *s2, float *wo3, float *pdf_eta_delta3, float *w3) {
    BSample bs; v3 w; v2 q = { s2[0], s2[1] }; v3 refl = V3(b->reflectance[0], b->reflectance[1], b->reflectance[2]);
    bsdf_sample(b, refl, V3(wi[0], wi[1], wi[2]), s1, q, 1, &bs, &w);
    wo3[0] = bs.wo.x; wo3[1] = bs.wo.y; wo3[2] = bs.wo.z; pdf_eta_delta3[0] = bs.pdf; pdf_eta_delta3[1] = bs.eta; pdf_eta_delta3[2] = (float) bs.delta;
    w3[0] = w.x; w3[1] = w.y; w3[2] = w.z;
}

/* Mesh::sample_position (mesh.cpp:749-803): face by area (DiscreteDistribution::sample_reuse, distr_1d.h:104-182: first CDF
   entry >= sample * sum; the sample is re-stretched over that face), then square_to_uniform_triangle (warp.h:153-156) */
static void mesh_sample_position(const OShape *sh, v2 sample, v3 *p, v3 *n, float *pdf) {
    const float sum = sh->area_cdf[sh->face_count - 1], norm = 1.f / sum, target = sample.y * sum;
    uint32_t lo = 0, hi = sh->face_count - 1;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (sh->area_cdf[mid] < target) lo = mid + 1; else hi = mid; }
    const uint32_t f = lo;
    float cdf_prev = f > 0 ? sh->area_cdf[f - 1] * norm : 0.f, pmf = (sh->area_cdf[f] - (f > 0 ? sh->area_cdf[f - 1] : 0.f)) * norm;
    sample.y = (sample.y - cdf_prev) / pmf;
    const uint32_t *fi = sh->faces + 3 * f;
    v3 p0 = V3(sh->pos[3*fi[0]], sh->pos[3*fi[0]+1], sh->pos[3*fi[0]+2]), p1 = V3(sh->pos[3*fi[1]], sh->pos[3*fi[1]+1], sh->pos[3*fi[1]+2]),
       p2 = V3(sh->pos[3*fi[2]], sh->pos[3*fi[2]+1], sh->pos[3*fi[2]+2]);
    v3 e0 = sub3(p1, p0), e1 = sub3(p2, p0);
    float t = safe_sqrt_(1.f - sample.x), bx = 1.f - t, by = t * sample.y;
    *p = fma3s(e0, bx, fma3s(e1, by, p0));
    v3 nn;
    if (sh->nrm) {
        v3 n0 = V3(sh->nrm[3*fi[0]], sh->nrm[3*fi[0]+1], sh->nrm[3*fi[0]+2]), n1 = V3(sh->nrm[3*fi[1]], sh->nrm[3*fi[1]+1], sh->nrm[3*fi[1]+2]),
           n2 = V3(sh->nrm[3*fi[2]], sh->nrm[3*fi[2]+1], sh->nrm[3*fi[2]+2]);
        nn = fma3s(n0, 1.f - bx - by, fma3s(n1, bx, scale3(n2, by)));
    } else nn = cross3(e0, e1);
    nn = normalize3(nn);
    *n = sh->flip_normals ? neg3(nn) : nn; *pdf = norm;
}

/* PathIntegrator::mis_weight: src/integrators/path.cpp:300-305 */
static inline float mis_weight(float a, float b) { a *= a; b *= b; float w = a / (a + b); return isfinite(w) ? w : 0.f; }

typedef struct { v3 p, n; v2 uv; float pdf; v3 d; float dist; int32_t emitter; } DS;   /* records.h:93-195 */

/* Scene::sample_emitter_direction (scene.cpp:299-353) -> AreaLight::sample_direction
   (area.cpp:117-168, uniform-radiance branch) -> Shape::sample_direction (shape.cpp:358-375)
   -> Rectangle::sample_position (rectangle.cpp:166-180); visibility via ray_test. */
static void sample_emitter_direction(OScene *sc, const SI *si, v2 sample, int active, DS *ds, v3 *spec, uint64_t *n_shadow) {
    memset(ds, 0, sizeof(*ds)); *spec = V3(0, 0, 0); ds->emitter = -1;
    if (sc->n_emitters == 0) return;
    uint32_t index = 0; float emitter_weight = 1.f, pmf = 1.f;
    if (sc->n_emitters > 1) {                                           /* sample_emitter, scene.cpp:227-250 */
        float cf = (float) sc->n_emitters, scaled = sample.x * cf;
        index = (uint32_t) scaled; if (index > sc->n_emitters - 1) index = sc->n_emitters - 1;
        emitter_weight = cf; sample.x = scaled - (float) index; pmf = 1.f / cf;
    }
    const mi355_emitter_desc *em = &sc->emitters[index];
    const OShape *sh = &sc->shapes[em->shape];
    ds->emitter = (int32_t) index;
    if (em->type == MI355_EMITTER_DIRECTIONALAREA) { ds->pdf = 0.f; return; }        /* DirectionalArea::sample_direction fails by design, directionalarea.cpp:126-130 */
    if (sh->type == MI355_SHAPE_RECTANGLE || sh->type == MI355_SHAPE_MESH || sh->type == MI355_SHAPE_DISK) {   /* Shape::sample_direction, shape.cpp:358-375 */
        if (sh->type == MI355_SHAPE_MESH) mesh_sample_position(sh, sample, &ds->p, &ds->n, &ds->pdf);
        else if (sh->type == MI355_SHAPE_DISK) { v2 q = square_to_uniform_disk_concentric(sample); ds->p = xf_point(sh->to_world, V3(q.x, q.y, 0.f)); ds->n = sh->frame_n; ds->pdf = sh->inv_area; }
        else { ds->p = xf_point(sh->to_world, V3(sample.x * 2.f - 1.f, sample.y * 2.f - 1.f, 0.f)); ds->n = sh->frame_n; ds->pdf = sh->inv_area; }
        ds->uv = sample;
        ds->d = sub3(ds->p, si->p);
        float dist2 = sqnorm3(ds->d); ds->dist = sqrtf(dist2);
        ds->d = scale3(ds->d, rcpf_(ds->dist));
        float dp = fabsf(dot3(ds->d, ds->n));
        float x = dist2 / dp;
        ds->pdf *= isfinite(x) ? x : 0.f;
    } else if (sh->type == MI355_SHAPE_SPHERE) {
        /* Sphere::sample_direction, sphere.cpp:236-311 */
        v3 dc_v = sub3(sh->center, si->p); float dc_2 = sqnorm3(dc_v);
        float radius_adj = sh->radius * (sh->flip_normals ? (1.f + RAY_EPS) : (1.f - RAY_EPS));
        v3 dn;
        if (dc_2 > radius_adj * radius_adj) {
            float inv_dc = rsqrtf_(dc_2), sin_theta_max = sh->radius * inv_dc, sin_theta_max_2 = sin_theta_max * sin_theta_max,
                  inv_sin_theta_max = rcpf_(sin_theta_max), cos_theta_max = safe_sqrt_(1.f - sin_theta_max_2);
            float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - sqr_(fmaf(cos_theta_max - 1.f, sample.x, 1.f)) : sin_theta_max_2 * sample.x,
                  cos_theta = safe_sqrt_(1.f - sin_theta_2);
            float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * safe_sqrt_(fmaf(-sin_theta_2, sqr_(inv_sin_theta_max), 1.f)),
                  sin_alpha = safe_sqrt_(fmaf(-cos_alpha, cos_alpha, 1.f));
            float ph = sample.y * (2.f * ORC_PI), sin_phi = sinf(ph), cos_phi = cosf(ph);
            v3 fn = scale3(dc_v, -inv_dc), fs, ft; coordinate_system(fn, &fs, &ft);
            v3 loc = V3(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha);
            dn = fma3s(fn, loc.z, fma3s(ft, loc.y, scale3(fs, loc.x)));
            ds->p = fma3s(dn, sh->radius, sh->center); ds->n = dn;
            ds->d = sub3(ds->p, si->p);
            float dist2 = sqnorm3(ds->d); ds->dist = sqrtf(dist2); ds->d = scale3(ds->d, rcpf_(ds->dist));
            ds->pdf = (0.5f * ORC_INV_PI) / (1.f - cos_theta_max);
            if (ds->dist == 0.f) ds->pdf = 0.f;
        } else {
            float z = fmaf(-2.f, sample.y, 1.f), r = safe_sqrt_(fmaf(-z, z, 1.f)), ph = 2.f * ORC_PI * sample.x;   /* warp.h:250-255 */
            dn = V3(r * cosf(ph), r * sinf(ph), z);
            ds->p = fma3s(dn, sh->radius, sh->center); ds->n = dn;
            ds->d = sub3(ds->p, si->p);
            float dist2 = sqnorm3(ds->d); ds->dist = sqrtf(dist2); ds->d = scale3(ds->d, rcpf_(ds->dist));
            ds->pdf = sh->inv_area * dist2 / fabsf(dot3(ds->d, ds->n));
        }
        if (sh->flip_normals) ds->n = neg3(ds->n);
        ds->uv = sample;
    } else { ds->pdf = 0.f; return; }                                   /* mesh emitters: outside the path's configs */
    active = active && dot3(ds->d, ds->n) < 0.f && ds->pdf != 0.f;      /* area.cpp:136 */
    v3 rad = V3(em->radiance[0], em->radiance[1], em->radiance[2]);
    *spec = active ? scale3(rad, rcpf_(ds->pdf)) : V3(0, 0, 0);          /* area.cpp:165-167 */
    if (sc->n_emitters > 1) { ds->pdf *= pmf; *spec = scale3(*spec, emitter_weight); }
    active = active && ds->pdf != 0.f;
    if (active) {                                                       /* scene.cpp:341-346 */
        Ray sr = spawn_ray_to(si->p, si->n, ds->p);
        (*n_shadow)++;
        if (scene_ray_test(sc, &sr)) { *spec = V3(0, 0, 0); ds->pdf = 0.f; }
    }
}
/* Scene::pdf_emitter_direction (scene.cpp:355-366) -> AreaLight::pdf_direction (area.cpp:170-197)
   -> Shape::pdf_direction (shape.cpp:377-388) */
static float pdf_emitter_direction(const OScene *sc, int32_t emitter, v3 ref_p, v3 d, v3 n, float dist) {
    const OShape *sh = &sc->shapes[sc->emitters[emitter].shape];
    float dp = dot3(d, n);
    if (!(dp < 0.f)) return 0.f;
    if (sc->emitters[emitter].type == MI355_EMITTER_DIRECTIONALAREA) return 0.f;     /* directionalarea.cpp:132-136 */
    float adp = fabsf(dp), pdf;
    if (sh->type == MI355_SHAPE_SPHERE) {                               /* Sphere::pdf_direction, sphere.cpp:313-325 */
        float sin_alpha = sh->radius * rcpf_(norm3(sub3(sh->center, ref_p))), cos_alpha = safe_sqrt_(1.f - sin_alpha * sin_alpha);
        pdf = sin_alpha < 0.99999994f ? (0.5f * ORC_INV_PI) / (1.f - cos_alpha) : sh->inv_area * (dist * dist) / adp;
    } else pdf = sh->inv_area * (adp != 0.f ? (dist * dist) / adp : 0.f);
    return pdf * (1.f / (float) sc->n_emitters);
}

/* ----------------------------------------------------------------------------------------
 * primary rays: SamplingIntegrator::render_sample head (integrator.cpp:419-455) and
 * PerspectiveCamera::sample_ray_differential (perspective.cpp:239-280)
 * ---------------------------------------------------------------------------------------- */
static void lane_to_pixel(const OScene *sc, uint32_t lane, uint32_t spp, int32_t *px, int32_t *py) {
    uint32_t b = sc->border, idx = lane / spp, W = sc->sensor.crop_width + 2u * b;   /* integrator.cpp:293-310; film_size += 2 border :133-134 */
    *py = (int32_t) (idx / W); *px = (int32_t) (idx - W * (idx / W));
    *px -= (int32_t) b; *py -= (int32_t) b;                              /* pos -= border_size, :307-308 */
    *px += (int32_t) sc->sensor.crop_offset_x; *py += (int32_t) sc->sensor.crop_offset_y;
}
static Ray sample_primary_ray(const OScene *sc, v2 sample_pos) {
    float sx = 1.f / (float) sc->sensor.crop_width, sy = 1.f / (float) sc->sensor.crop_height;
    float ox = -(float) sc->sensor.crop_offset_x * sx, oy = -(float) sc->sensor.crop_offset_y * sy;
    float ax = fmaf(sample_pos.x, sx, ox), ay = fmaf(sample_pos.y, sy, oy);
    const float *m = sc->sample_to_camera;
    /* Transform * Point with perspective divide, transform.h:113-121 */
    float r[4];
    for (int i = 0; i < 4; ++i) { float v = m[i*4+3]; v = fmaf(m[i*4+0], ax, v); v = fmaf(m[i*4+1], ay, v); v = fmaf(m[i*4+2], 0.f, v); r[i] = v; }
    float iw = rcpf_(r[3]);
    v3 near_p = V3(r[0] * iw, r[1] * iw, r[2] * iw);
    v3 d = normalize3(near_p);
    const float *tw = sc->sensor.to_world;
    Ray ray; ray.o = V3(tw[3], tw[7], tw[11]); ray.d = xf_vector(tw, d);
    float inv_z = rcpf_(d.z), near_t = sc->sensor.near_clip * inv_z, far_t = sc->sensor.far_clip * inv_z;
    ray.o = add3(ray.o, scale3(ray.d, near_t));
    ray.maxt = far_t - near_t;
    return ray;
}

/* ----------------------------------------------------------------------------------------
 * RealisticLensCamera::sample_ray -- src/sensors/realisticlens.cpp:2209-2292 (rear-element sampling, :1870-1878),
 * trace_ray_from_film :1654-1683, LensInterface::compute_interaction :269-297, SpheroidLens::intersect :359-397,
 * PlanoLens::intersect :444-468, ApertureStop::compute_interaction :505-517.  Returns the ray weight (cos^4, 0 if blocked).
 * ---------------------------------------------------------------------------------------- */
/* AsphericalLens (realisticlens.cpp:553-821) with per-ray ("scalar variant") semantics: the reference's intersect_conic /
 * intersect branch on dr::none / dr::any over the whole wavefront in JIT variants, which has no per-lane meaning.
 * Row: L[6] = kappa, L[7] = c (unitless curvature r_scale / R_mm), L[8..14] = a_i * r_scale^(2i+3) (ctor :559-572). */
#define ORC_LENS_ROW 16
static float asph_conic(const float *L, float r2_) {                   /* _eval_conic :683-687 */
    float c = L[7], K = L[6];
    float sqr_term = 1.f - (1.f + K) * (c * c) * r2_;
    return c * r2_ * (1.f / (1.f + sqrtf(sqr_term)));
}
static float asph_eval(const float *L, float r2) {                      /* eval_asph :698-707 */
    float er = L[2], r2_ = r2 * (1.f / (er * er)), z_ = 0.f;
    for (int i = 6; i >= 0; --i) z_ = fmaf(z_, r2_, L[8 + i]);
    z_ *= r2_ * r2_;
    z_ += asph_conic(L, r2_);
    return -z_ * er + L[3];
}
static float asph_grad_(const float *L, float r2_) {                    /* _eval_asph_grad :722-731, _eval_conic_grad :690-695 */
    float r_ = sqrtf(r2_), z_ = 0.f;
    for (int i = 6; i >= 0; --i) z_ = fmaf(z_, r2_, (2.f * (float) i + 4.f) * L[8 + i]);
    z_ *= r2_ * r_;
    float cr = L[7] * r_, sqr_term = 1.f - (1.f + L[6]) * (cr * cr);
    z_ += cr * (1.f / sqrtf(sqr_term));
    return -z_;
}
static v3 asph_normal(const float *L, v3 p) {                           /* normal :646-655 */
    float inv = 1.f / L[2], rx = p.x * inv, ry = p.y * inv, r2_ = rx * rx + ry * ry;
    if (r2_ >= 4.f * ORC_EPS) { float s = 1.f / sqrtf(r2_); rx *= s; ry *= s; } else { rx = 0.f; ry = 0.f; }
    float zg = asph_grad_(L, r2_);
    return normalize3(V3(zg * rx, zg * ry, -1.f));
}
static int asph_conic_hit(const float *L, v3 o, v3 d, float *t_out) {   /* intersect_conic :770-819 */
    float R = L[2], c = L[7], K = L[6], z0 = L[3];
    v3 on = scale3(sub3(o, V3(0.f, 0.f, z0)), 1.f / R);
    float A = c * (1.f + K * (d.z * d.z)), B = 2.f * (c * (dot3(on, d) + K * on.z * d.z) - d.z), C = c * (dot3(on, on) + K * (on.z * on.z)) - 2.f * on.z;
    float t0, t1; if (!solve_quadratic(A, B, C, &t0, &t1)) return 0;
    t0 *= R; t1 *= R;
    float z0t = c * (o.z + t0 * d.z - z0), z1t = c * (o.z + t1 * d.z - z0), lim = K > -1.f ? R * (1.f / (1.f + K)) : INFINITY;
    int v0 = z0t > 0.f && z0t <= lim, v1 = z1t > 0.f && z1t <= lim;
    if (!(v0 || v1)) return 0;
    *t_out = (v0 && v1) ? (t0 > 0.f ? t0 : t1) : (v0 ? t0 : t1);
    return 1;
}
static int asph_intersect(const float *L, v3 o, v3 d, float *t_out, v3 *p_out, v3 *n_out) {   /* intersect :574-644 */
    const float TOL = ORC_EPS * 10.f;
    float t;
    if (!asph_conic_hit(L, o, d, &t)) t = (L[3] - o.z) * (1.f / d.z);
    v3 p = fma3s(d, t, o); float r2 = p.x * p.x + p.y * p.y, err = INFINITY;
    for (int itr = 0; itr < 10; ++itr) {                                /* Newton on tangent planes */
        float z_asph = asph_eval(L, r2);
        v3 plane_p = V3(p.x, p.y, z_asph);
        err = fabsf(p.z - z_asph);
        v3 pn = asph_normal(L, p);
        t = dot3(pn, sub3(plane_p, o)) / dot3(pn, d);
        p = fma3s(d, t, o); r2 = p.x * p.x + p.y * p.y;
        if (!(err > TOL)) break;
    }
    if (!(err < TOL)) return 0;
    *t_out = t; *p_out = p; *n_out = asph_normal(L, p);
    return isfinite(t);
}
static int lens_interact(const float *L, v3 *o, v3 *d) {
    uint32_t kind = (uint32_t) L[0]; float R = L[1], er = L[2], z = L[3], n_left = L[4], n_right = L[5];
    v3 p, n; float t;
    if (kind == 3) {
        if (!asph_intersect(L, *o, *d, &t, &p, &n)) return 0;
    } else if (kind == 0) {
        v3 center = V3(0.f, 0.f, z + R);
        v3 pc = sub3(center, *o);
        float center_proj = dot3(pc, *d);
        float perp = norm3(sub3(pc, scale3(*d, center_proj)));
        float disc = R * R - perp * perp;
        if (!(disc >= 0.f)) return 0;
        float sq = sqrtf(disc), near_t = center_proj - sq, far_t = center_proj + sq;
        if (!(far_t >= 0.f)) return 0;
        int convex = R > 0.f;
        t = (convex != (d->z < 0.f)) ? (near_t >= 0.f ? near_t : INFINITY) : far_t;
        p = fma3s(*d, t, *o);
        n = normalize3(sub3(p, center)); if (!convex) n = neg3(n);
        if (!(t != INFINITY)) return 0;
    } else {
        n = V3(0.f, 0.f, -1.f);
        float n_dot_d = dot3(n, *d);
        if (!(fabsf(n_dot_d) >= ORC_EPS)) return 0;
        t = (n.z * z - dot3(n, *o)) / n_dot_d;
        if (!(t >= 0.f)) return 0;
        p = fma3s(*d, t, *o);
    }
    if (!((p.x * p.x + p.y * p.y) < er * er)) return 0;
    if (kind == 2) { *o = p; return 1; }
    v3 wi = neg3(*d);
    float cos_theta_i = dot3(wi, n), eta = n_right / n_left;
    float r, cos_theta_t, eta_it, eta_ti; fresnel_(cos_theta_i, eta, &r, &cos_theta_t, &eta_it, &eta_ti);
    if (!(r <= 0.99999994f)) return 0;
    float k = fmaf(dot3(wi, n), eta_ti, cos_theta_t);                   /* refract(wi, m, cos_theta_t, eta_ti), fresnel.h:309-313 */
    *d = V3(fmaf(n.x, k, -(wi.x * eta_ti)), fmaf(n.y, k, -(wi.y * eta_ti)), fmaf(n.z, k, -(wi.z * eta_ti)));
    *o = p;
    return 1;
}
/* RadicalInverse::eval (include/mitsuba/core/qmc.h:67-92), prime bases 2 and 3, recip = 1.f / base (qmc.cpp:50) */
static float radical_inverse(uint32_t base, uint64_t index) {
    uint64_t value = 0; float factor = 1.f, recip = 1.f / (float) base;
    while (index != 0) {
        uint64_t next = index / base;
        factor = factor * recip;
        value = (value - next) * base + index;
        index = next;
    }
    return fminf(0.99999994f, (float) value * factor);
}
static float lerp_(float a, float b, float t) { return fmaf(b, t, fmaf(-a, t, a)); }   /* dr::lerp = fmadd(b, t, fnmadd(a, t, a)) */
/* bound_exit_pupil + compute_exit_pupil_bounds (realisticlens.cpp:1685-1806): for 64 rings of film radii, the bounding box on
 * the rear plane of the rays that make it through the lens.  (The reference skips samples that already lie inside the box;
 * that only saves work.) */
static void exit_pupil_bounds(OScene *sc, float film_diagonal) {
    const uint32_t NSEG = 64, N = 1u << 20;
    const float rear_radius = sc->lens[2] * 1.5f, rear_z = sc->lens[3];
    for (uint32_t seg = 0; seg < NSEG; ++seg) {
        float r0 = (float) seg * film_diagonal / (float) NSEG, r1 = (float) (seg + 1) * film_diagonal / (float) NSEG;
        float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY; uint64_t transmitted = 0;
        #pragma omp parallel for reduction(min:mnx, mny) reduction(max:mxx, mxy) reduction(+:transmitted) schedule(static)
        for (int64_t i = 0; i < (int64_t) N; ++i) {
            v3 p_film = V3(lerp_(r0, r1, ((float) i + 0.5f) / (float) N), 0.f, 0.f);
            v3 p_rear = V3(lerp_(-rear_radius, rear_radius, radical_inverse(2, (uint64_t) i)),
                           lerp_(-rear_radius, rear_radius, radical_inverse(3, (uint64_t) i)), rear_z);
            v3 o = p_film, d = normalize3(sub3(p_rear, p_film));
            int active = 1;
            for (uint32_t k = 0; k < sc->lens_count && active; ++k) active = lens_interact(sc->lens + ORC_LENS_ROW * k, &o, &d);
            if (active) {
                mnx = fminf(mnx, p_rear.x); mny = fminf(mny, p_rear.y); mxx = fmaxf(mxx, p_rear.x); mxy = fmaxf(mxy, p_rear.y);
                transmitted++;
            }
        }
        if (transmitted == 0) { mnx = mny = -rear_radius; mxx = mxy = rear_radius; }
        float spacing = 4.f * rear_radius * sqrtf(2.f / (float) N);
        float *b = sc->pupil + 4 * seg;
        b[0] = mnx - spacing; b[1] = mny - spacing; b[2] = mxx + spacing; b[3] = mxy + spacing;
    }
}
static float sample_lens_ray(const OScene *sc, v2 sample_pos, v2 aperture_sample, Ray *ray_out) {
    float sx = 1.f / (float) sc->sensor.crop_width, sy = 1.f / (float) sc->sensor.crop_height;
    float ax = fmaf(sample_pos.x, sx, -(float) sc->sensor.crop_offset_x * sx), ay = fmaf(sample_pos.y, sy, -(float) sc->sensor.crop_offset_y * sy);
    /* m_sample_to_film = inverse of film_to_crop_transform (sensor.h:308-345): crop -> NDC -> physical film */
    float rsx = (float) sc->sensor.crop_width / (float) sc->sensor.film_width, rsy = (float) sc->sensor.crop_height / (float) sc->sensor.film_height;
    float rox = (float) sc->sensor.crop_offset_x / (float) sc->sensor.film_width, roy = (float) sc->sensor.crop_offset_y / (float) sc->sensor.film_height;
    v3 film_p = V3(sc->film_px * (2.f * fmaf(ax, rsx, rox) - 1.f), sc->film_py * (2.f * fmaf(ay, rsy, roy) - 1.f), 0.f);
    const float *L0 = sc->lens;
    v3 aperture_p; float bounds_area = 0.f;
    if (sc->sensor.sample_exit_pupil) {                                 /* sample_exit_pupil, :1808-1836 */
        float r_film = sqrtf(film_p.x * film_p.x + film_p.y * film_p.y);
        uint32_t r_idx = (uint32_t) (int32_t) floorf(r_film / sc->sensor.film_diagonal * 64.f);
        if (r_idx > 63u) r_idx = 63u;
        const float *b = sc->pupil + 4 * r_idx;
        bounds_area = (b[2] - b[0]) * (b[3] - b[1]);
        float px = lerp_(b[0], b[2], aperture_sample.x), py = lerp_(b[1], b[3], aperture_sample.y);
        float sin_t = r_film > 0.f ? film_p.y * (1.f / r_film) : 0.f, cos_t = r_film > 0.f ? film_p.x * (1.f / r_film) : 1.f;
        aperture_p = V3(cos_t * px - sin_t * py, sin_t * px + cos_t * py, L0[3]);
    } else {
        v2 dk = square_to_uniform_disk_concentric(aperture_sample);
        aperture_p = V3(L0[2] * dk.x, L0[2] * dk.y, L0[3]);             /* sample_rear_element, :1870-1878 */
    }
    v3 o = film_p, d = normalize3(sub3(aperture_p, film_p));
    int active = 1;
    for (uint32_t i = 0; i < sc->lens_count && active; ++i) active = lens_interact(sc->lens + ORC_LENS_ROW * i, &o, &d);
    const float *tw = sc->sensor.to_world;
    ray_out->o = xf_point(tw, o); ray_out->d = xf_vector(tw, d);
    float inv_z = rcpf_(d.z), near_t = sc->sensor.near_clip * inv_z, far_t = sc->sensor.far_clip * inv_z;
    ray_out->o = add3(ray_out->o, scale3(ray_out->d, near_t)); ray_out->maxt = far_t - near_t;
    float ct = d.z, c2 = ct * ct;
    if (sc->sensor.sample_exit_pupil)                                   /* :2282-2285 */
        return active ? sc->sensor.shutter_open_time * bounds_area * (1.f / (L0[3] * L0[3])) : 0.f;
    return active ? c2 * c2 : 0.f;
}

/* ----------------------------------------------------------------------------------------
 * PathIntegrator::sample -- src/integrators/path.cpp:95-287 (JIT masking semantics)
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t isect, shade, shadow, isect_dead; } Cnt;

/* keep_stream (samples_per_pass): every sampler call in the loop of path.cpp:142-281 is unmasked and the loop masks by `active` alone, so a
   lane draws six numbers in EVERY iteration it enters -- also in the one that leaves the scene (:196, :212-213, :272) */
static int g_path_keep_stream = 0;
#pragma omp threadprivate(g_path_keep_stream)
static void path_sample(OScene *sc, pcg32 *rng, Ray ray, const mi355_render_params *rp, v3 *result_out, int *valid_out, Cnt *cnt) {
    uint32_t max_depth = (uint32_t) rp->max_depth, rr_depth = (uint32_t) rp->rr_depth;
    v3 throughput = V3(1, 1, 1), result = V3(0, 0, 0); float eta = 1.f; uint32_t depth = 0;
    int valid_ray = 0;                                                  /* no environment emitter (path.cpp:115) */
    v3 prev_p = V3(0, 0, 0); float prev_bsdf_pdf = 1.f; int prev_bsdf_delta = 1;
    int active = 1;
    if (max_depth == 0) { *result_out = result; *valid_out = 0; return; }
    for (uint32_t it = 0; active && it < max_depth; ++it) {
        PI pi = scene_intersect(sc, &ray); cnt->isect++;
        SI si = compute_si(sc, &ray, &pi);
        int si_valid = si.t != INFINITY;
        int32_t emitter = si_valid ? sc->shapes[si.shape].emitter : -1;
        /* ---- direct emission (path.cpp:158-174) */
        if (emitter >= 0) {
            v3 rel = sub3(si.p, prev_p); float dist = norm3(rel);
            v3 d = scale3(rel, rcpf_(dist));                            /* records.h:173-180 */
            float em_pdf = 0.f;
            if (!prev_bsdf_delta) em_pdf = pdf_emitter_direction(sc, emitter, prev_p, d, si.sh_n, dist);
            float mis_bsdf = mis_weight(prev_bsdf_pdf, em_pdf);
            const float *r = sc->emitters[emitter].radiance;
            v3 le = (si.wi.z > 0.f && prev_bsdf_pdf > 0.f && sc->emitters[emitter].type == MI355_EMITTER_AREA) ? V3(r[0], r[1], r[2]) : V3(0, 0, 0);   /* area.cpp:83-90; DirectionalArea::eval = 0 */
            result = fma3(throughput, scale3(le, mis_bsdf), result);
        }
        int active_next = (depth + 1 < max_depth) && si_valid;          /* path.cpp:177 */
        if (!si_valid) { if (g_path_keep_stream) for (int k = 0; k < 6; ++k) (void) pcg32_float(rng); break; }   /* nothing below can affect the result */
        cnt->shade++;
        const mi355_bsdf_desc *bsdf = &sc->bsdfs[sc->shapes[si.shape].bsdf];
        v3 refl = bsdf_reflectance(sc, bsdf, &si);
        /* ---- emitter sampling (path.cpp:186-208) */
        int active_em = active_next && bsdf_is_smooth(bsdf);             /* path.cpp:187 */
        DS ds; v3 em_weight; v2 s_em; s_em.x = pcg32_float(rng); s_em.y = pcg32_float(rng);
        sample_emitter_direction(sc, &si, s_em, active_em, &ds, &em_weight, &cnt->shadow);
        active_em = active_em && ds.pdf != 0.f;
        v3 wo = si_to_local(&si, ds.d);
        /* ---- BSDF eval + sample (path.cpp:212-216, bsdf.cpp:22-33) */
        float s1 = pcg32_float(rng); v2 s2; s2.x = pcg32_float(rng); s2.y = pcg32_float(rng);
        v3 bsdf_val; float bsdf_pdf; bsdf_eval_pdf(bsdf, refl, si.wi, wo, 1, &bsdf_val, &bsdf_pdf);
        BSample bsm; v3 bsdf_weight; bsdf_sample(bsdf, refl, si.wi, s1, s2, 1, &bsm, &bsdf_weight);
        v3 bs_wo = bsm.wo; float bs_pdf = bsm.pdf;
        if (active_em) {                                                /* path.cpp:220-230 */
            float mis_em = mis_weight(ds.pdf, bsdf_pdf);
            result = fma3(throughput, scale3(mul3(bsdf_val, em_weight), mis_em), result);
        }
        /* ---- BSDF sampling (path.cpp:234-262) */
        ray = spawn_ray(si.p, si.n, si_to_world(&si, bs_wo));
        throughput = mul3(throughput, bsdf_weight);
        eta *= bsm.eta;
        valid_ray = valid_ray || si_valid;
        prev_p = si.p; prev_bsdf_pdf = bs_pdf; prev_bsdf_delta = bsm.delta;
        /* ---- stopping criterion (path.cpp:266-280) */
        depth += 1;
        float tmax = max3(throughput);
        float rr_prob = fminf(tmax * (eta * eta), .95f);
        int rr_active = depth >= rr_depth, rr_continue = pcg32_float(rng) < rr_prob;
        if (rr_active) throughput = scale3(throughput, rcpf_(rr_prob));
        active = active_next && (!rr_active || rr_continue) && tmax != 0.f;
    }
    *result_out = valid_ray ? result : V3(0, 0, 0); *valid_out = valid_ray;
}

/* ----------------------------------------------------------------------------------------
 * PRBIntegrator.sample -- src/python/python/ad/integrators/prb.py:59-257
 * mode 0 = Primal (returns L), mode 1 = Backward (state_in = L, accumulates parameter
 * gradients; closed form of dr.backward_from(dL * Lo), see DESIGN.md "PRB adjoint")
 * ---------------------------------------------------------------------------------------- */
typedef struct { float *grad; const mi355_param_info *layout; uint32_t n_params; const int32_t *bsdf_slot, *tex_slot, *em_slot;
                 /* Forward mode (prb.py mode == dr.ADMode.Forward, driven by RBIntegrator.render_forward, common.py:498-629): `grad` is then
                    READ as the tangent of every parameter and the walk accumulates dL = sum d(Le + Lr_dir + Lr_ind)/d(theta) * tangent */
                 int forward; v3 *dl_acc; } GradCtx;

static void prb_sample(OScene *sc, pcg32 *rng, Ray ray, const mi355_render_params *rp, int backward,
                       v3 dL, v3 state_in, const GradCtx *g, v3 *L_out, int *valid_out, Cnt *cnt) {
    uint32_t max_depth = (uint32_t) rp->max_depth, rr_depth = (uint32_t) rp->rr_depth;
    uint32_t depth = 0; v3 L = backward ? state_in : V3(0, 0, 0); v3 beta = V3(1, 1, 1); float eta = 1.f;
    v3 prev_p = V3(0, 0, 0); float prev_bsdf_pdf = 1.f; int prev_bsdf_delta = 1; int active = 1;
    for (uint32_t it = 0; active && it < max_depth; ++it) {
        int active_next = 1;
        PI pi = scene_intersect(sc, &ray); cnt->isect++;
        SI si = compute_si(sc, &ray, &pi);
        int si_valid = si.t != INFINITY;
        if ((rp->flags & MI355_FLAG_HIDE_EMITTERS) && depth == 0 && !si_valid) active_next = 0;   /* prb.py:122-123 */
        int32_t emitter = si_valid ? sc->shapes[si.shape].emitter : -1;
        /* ---- direct emission (prb.py:126-134) */
        v3 Le = V3(0, 0, 0); float le_mis = 0.f; int le_front = 0;
        if (emitter >= 0) {
            v3 rel = sub3(si.p, prev_p); float dist = norm3(rel); v3 d = scale3(rel, rcpf_(dist));
            float em_pdf = prev_bsdf_delta ? 0.f : pdf_emitter_direction(sc, emitter, prev_p, d, si.sh_n, dist);
            le_mis = mis_weight(prev_bsdf_pdf, em_pdf);
            const float *r = sc->emitters[emitter].radiance;
            le_front = si.wi.z > 0.f && active_next && sc->emitters[emitter].type == MI355_EMITTER_AREA;   /* DirectionalArea::eval = 0 */
            v3 e = le_front ? V3(r[0], r[1], r[2]) : V3(0, 0, 0);
            Le = mul3(scale3(beta, le_mis), e);
        }
        active_next = active_next && (depth + 1 < max_depth) && si_valid;   /* prb.py:139 */
        if (!si_valid) { if (!backward) L = add3(L, Le); break; }
        cnt->shade++;
        const mi355_bsdf_desc *bsdf = &sc->bsdfs[sc->shapes[si.shape].bsdf];
        int32_t tidx[4]; float twgt[4]; v3 refl;
        if (bsdf->reflectance_texture >= 0) refl = texture_eval(&sc->textures[bsdf->reflectance_texture], si.uv, tidx, twgt);
        else refl = V3(bsdf->reflectance[0], bsdf->reflectance[1], bsdf->reflectance[2]);
        /* ---- emitter sampling (prb.py:142-162) */
        int active_em = active_next && bsdf_is_smooth(bsdf);
        DS ds; v3 em_weight; v2 s_em; s_em.x = pcg32_float(rng); s_em.y = pcg32_float(rng);
        sample_emitter_direction(sc, &si, s_em, active_em, &ds, &em_weight, &cnt->shadow);
        active_em = active_em && ds.pdf != 0.f;
        v3 wo = si_to_local(&si, ds.d);
        v3 bsdf_value_em; float bsdf_pdf_em; bsdf_eval_pdf(bsdf, refl, si.wi, wo, active_em, &bsdf_value_em, &bsdf_pdf_em);
        float mis_em = mis_weight(ds.pdf, bsdf_pdf_em);
        v3 Lr_dir = mul3(mul3(scale3(beta, mis_em), bsdf_value_em), em_weight);
        /* ---- detached BSDF sampling (prb.py:166-169) */
        float s1 = pcg32_float(rng); v2 s2; s2.x = pcg32_float(rng); s2.y = pcg32_float(rng);
        BSample bsm; v3 bsdf_weight; bsdf_sample(bsdf, refl, si.wi, s1, s2, active_next, &bsm, &bsdf_weight);
        v3 bs_wo = bsm.wo; float bs_pdf = bsm.pdf;
        /* ---- update (prb.py:173-182) */
        v3 beta_pre = beta;
        L = backward ? sub3(sub3(L, Le), Lr_dir) : add3(add3(L, Le), Lr_dir);
        ray = spawn_ray(si.p, si.n, si_to_world(&si, bs_wo));
        eta *= bsm.eta; beta = mul3(beta, bsdf_weight);
        prev_p = si.p; prev_bsdf_pdf = bs_pdf; prev_bsdf_delta = bsm.delta;
        /* ---- stopping criterion (prb.py:186-198) */
        float bmax = max3(beta);
        active_next = active_next && bmax != 0.f;
        float rr_prob = fminf(bmax * (eta * eta), .95f);
        int rr_active = depth >= rr_depth;
        if (rr_active) beta = scale3(beta, rcpf_(rr_prob));
        int rr_continue = pcg32_float(rng) < rr_prob;
        active_next = active_next && (!rr_active || rr_continue);
        /* ---- differential phase (prb.py:202-247), closed form */
        if (backward && g) {
            /* bsdf.eval(si, wo_sampled, active_next) and its detached twin (prb.py:213-227) */
            float co = bs_wo.z; int ev_active = active_next && si.wi.z > 0.f && co > 0.f && bsdf->type == MI355_BSDF_DIFFUSE;
            v3 det = scale3(bsdf_weight, bs_pdf);
            v3 inv = V3(det.x != 0.f ? rcpf_(det.x) : 0.f, det.y != 0.f ? rcpf_(det.y) : 0.f, det.z != 0.f ? rcpf_(det.z) : 0.f);
            float dval_dalb = ev_active ? ORC_INV_PI * co : 0.f;                         /* d bsdf_val / d albedo */
            v3 dLo_dalb = scale3(mul3(L, inv), dval_dalb);                               /* Lr_ind term */
            /* Lr_dir term: beta_pre * mis_em * (InvPi cos_o_em) * em_weight */
            int em_active = active_em && si.wi.z > 0.f && wo.z > 0.f && bsdf->type == MI355_BSDF_DIFFUSE;
            float dval_em = em_active ? ORC_INV_PI * wo.z : 0.f;
            dLo_dalb = add3(dLo_dalb, mul3(scale3(scale3(beta_pre, mis_em), dval_em), em_weight));
            v3 ga = mul3(dL, dLo_dalb);
            uint32_t bidx = sc->shapes[si.shape].bsdf;
            if (g->forward) {
                v3 dalb = V3(0, 0, 0);
                if (bsdf->reflectance_texture >= 0) {
                    int32_t slot = g->tex_slot[bsdf->reflectance_texture];
                    if (slot >= 0) {
                        const OTexture *t = &sc->textures[bsdf->reflectance_texture];
                        const float *tb = g->grad + g->layout[slot].offset;
                        for (int k = 0; k < 4; ++k) if (tidx[k] >= 0) {
                            if (t->c == 3) { dalb.x = fmaf(twgt[k], tb[tidx[k]*3+0], dalb.x); dalb.y = fmaf(twgt[k], tb[tidx[k]*3+1], dalb.y); dalb.z = fmaf(twgt[k], tb[tidx[k]*3+2], dalb.z); }
                            else { float v = twgt[k] * tb[tidx[k]]; dalb.x += v; dalb.y += v; dalb.z += v; }
                        }
                    }
                } else if (bsdf->type == MI355_BSDF_DIFFUSE && g->bsdf_slot[bidx] >= 0) {
                    const float *tb = g->grad + g->layout[g->bsdf_slot[bidx]].offset; dalb = V3(tb[0], tb[1], tb[2]);
                }
                if (bsdf->type == MI355_BSDF_DIFFUSE) *g->dl_acc = add3(*g->dl_acc, mul3(dLo_dalb, dalb));   /* (other lobes carry no parameter; their inv * 0 may be NaN) */
                if (emitter >= 0 && g->em_slot[emitter] >= 0 && le_front) {
                    const float *tb = g->grad + g->layout[g->em_slot[emitter]].offset;
                    *g->dl_acc = add3(*g->dl_acc, mul3(scale3(beta_pre, le_mis), V3(tb[0], tb[1], tb[2])));
                }
                if (active_em && ds.emitter >= 0 && g->em_slot[ds.emitter] >= 0 && dot3(ds.d, ds.n) < 0.f) {
                    const float *tb = g->grad + g->layout[g->em_slot[ds.emitter]].offset;
                    v3 w = scale3(mul3(scale3(beta_pre, mis_em), bsdf_value_em), rcpf_(ds.pdf));
                    *g->dl_acc = add3(*g->dl_acc, mul3(w, V3(tb[0], tb[1], tb[2])));
                }
            } else {
            if (bsdf->reflectance_texture >= 0) {
                int32_t slot = g->tex_slot[bsdf->reflectance_texture];
                if (slot >= 0) {
                    const OTexture *t = &sc->textures[bsdf->reflectance_texture];
                    float *gb = g->grad + g->layout[slot].offset;
                    for (int k = 0; k < 4; ++k) if (tidx[k] >= 0) {
                        if (t->c == 3) { gb[tidx[k]*3+0] += twgt[k] * ga.x; gb[tidx[k]*3+1] += twgt[k] * ga.y; gb[tidx[k]*3+2] += twgt[k] * ga.z; }
                        else gb[tidx[k]] += twgt[k] * (ga.x + ga.y + ga.z);
                    }
                }
            } else if (bsdf->type == MI355_BSDF_DIFFUSE) {
                int32_t slot = g->bsdf_slot[bidx];
                if (slot >= 0) { float *gb = g->grad + g->layout[slot].offset; gb[0] += ga.x; gb[1] += ga.y; gb[2] += ga.z; }
            }
            /* emitter radiance: d Le / d rad = beta_pre*mis*[front];  d Lr_dir / d rad = beta_pre*mis_em*bsdf_value_em / ds.pdf */
            if (emitter >= 0 && g->em_slot[emitter] >= 0 && le_front) {
                float *gb = g->grad + g->layout[g->em_slot[emitter]].offset;
                gb[0] += dL.x * beta_pre.x * le_mis; gb[1] += dL.y * beta_pre.y * le_mis; gb[2] += dL.z * beta_pre.z * le_mis;
            }
            /* eval_emitter_direction masks back-facing samples (area.cpp:199-209): their ds.pdf is non-zero
               (only occlusion zeroes it, scene.cpp:343-345) but em_val = 0, so they carry no gradient */
            if (active_em && ds.emitter >= 0 && g->em_slot[ds.emitter] >= 0 && dot3(ds.d, ds.n) < 0.f) {
                float *gb = g->grad + g->layout[g->em_slot[ds.emitter]].offset;
                /* em_weight = em_val / ds.pdf (prb.py:154-155); ds.pdf already carries the emitter pmf (scene.cpp:324) */
                v3 w = scale3(mul3(scale3(beta_pre, mis_em), bsdf_value_em), rcpf_(ds.pdf));
                gb[0] += dL.x * w.x; gb[1] += dL.y * w.y; gb[2] += dL.z * w.z;
            }
            }
        }
        depth += 1;                                                       /* prb.py:249 (si valid here) */
        active = active_next;
    }
    *L_out = L; *valid_out = depth != 0;
}

/* ----------------------------------------------------------------------------------------
 * film: ImageBlock::put coalesced path (imageblock.cpp:418-531), HDRFilm::develop (hdrfilm.cpp:304-409)
 * ---------------------------------------------------------------------------------------- */
static inline float filter_eval(const OScene *sc, float x) {
    if (sc->sensor.rfilter == MI355_RFILTER_GAUSSIAN) return gaussian_eval(sc->filter_coeff, x);
    return tent_eval(sc->filter_radius, x);
}
/* returns the number of taps per axis and fills weights; (x0, y0) = top-left tap (film-local) */
static int splat_weights(const OScene *sc, v2 pos, int32_t *x0, int32_t *y0, float *wx, float *wy) {
    if (sc->sensor.rfilter == MI355_RFILTER_BOX) {                       /* imageblock.cpp:210-233 */
        *x0 = (int32_t) floorf(pos.x) - (int32_t) sc->sensor.crop_offset_x;
        *y0 = (int32_t) floorf(pos.y) - (int32_t) sc->sensor.crop_offset_y;
        wx[0] = wy[0] = 1.f; return 1;
    }
    int n = (int) ceilf(sc->filter_radius - .5f), count = 2 * n + 1;
    int32_t pix = (int32_t) floorf(pos.x) - n, piy = (int32_t) floorf(pos.y) - n;
    *x0 = pix - (int32_t) sc->sensor.crop_offset_x; *y0 = piy - (int32_t) sc->sensor.crop_offset_y;
    float rx = (float) pix + .5f - pos.x, ry = (float) piy + .5f - pos.y;
    for (int i = 0; i < count; ++i) { wx[i] = filter_eval(sc, rx); wy[i] = filter_eval(sc, ry); rx += 1.f; ry += 1.f; }
    return count;
}
static void film_put(const OScene *sc, float *film, int channels, v2 pos, const float *values) {
    int32_t x0, y0; float wx[16], wy[16];
    int count = splat_weights(sc, pos, &x0, &y0, wx, wy);
    int32_t W = (int32_t) sc->sensor.crop_width, H = (int32_t) sc->sensor.crop_height;
    for (int ys = 0; ys < count; ++ys) for (int xs = 0; xs < count; ++xs) {
        int32_t x = x0 + xs, y = y0 + ys;
        if (x < 0 || y < 0 || x >= W || y >= H) continue;
        float w = wy[ys] * wx[xs];
        for (int k = 0; k < channels; ++k) film[((size_t) y * W + x) * channels + k] += values[k] * w;
    }
}
ORC_API void orc_film_develop(const OScene *sc, const float *film_rgbw, float *image_rgb) {
    size_t n = (size_t) sc->sensor.crop_width * sc->sensor.crop_height;
    for (size_t i = 0; i < n; ++i) {
        float w = film_rgbw[4*i+3]; float dv = (w == 0.f) ? 1.f : w;
        for (int c = 0; c < 3; ++c) image_rgb[3*i+c] = film_rgbw[4*i+c] / dv;
    }
}
/* test hook for ImageBlock::put KATs */
ORC_API void orc_exit_pupil_bounds(const OScene *sc, float *out) { memcpy(out, sc->pupil, sizeof(sc->pupil)); }
ORC_API float orc_radical_inverse(uint32_t base, uint64_t index) { return radical_inverse(base, index); }
ORC_API void orc_film_put(const OScene *sc, float *film_rgbw, const float *pos2, const float *values4, int n) {
    for (int i = 0; i < n; ++i) { v2 p = { pos2[2*i], pos2[2*i+1] }; film_put(sc, film_rgbw, 4, p, values4 + 4*i); }
}

/* ----------------------------------------------------------------------------------------
 * Particle tracer: src/integrators/ptracer.cpp:77-371 driven by AdjointIntegrator::render (integrator.cpp:545-745, JIT
 * branch: one lane per sample, lanes = film_size * spp, sampler seeded like the path tracer's).  RGB, perspective sensor.
 * ---------------------------------------------------------------------------------------- */
/* ImageBlock::put with normalize = true (imageblock.cpp:460-474): the sample's filter weights are divided by their sum, the
   weight channel is not touched (ptracer.cpp:366-368) */
static void film_put_normalized(const OScene *sc, float *film_rgbw, v2 pos, v3 rgb) {
    int32_t x0, y0; float wx[16], wy[16];
    int count = splat_weights(sc, pos, &x0, &y0, wx, wy);
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < count; ++i) { sx += wx[i]; sy += wy[i]; }
    float factor = sx * sy; factor = factor != 0.f ? 1.f / factor : 0.f;
    for (int i = 0; i < count; ++i) wx[i] *= factor;
    int32_t W = (int32_t) sc->sensor.crop_width, H = (int32_t) sc->sensor.crop_height;
    for (int ys = 0; ys < count; ++ys) for (int xs = 0; xs < count; ++xs) {
        int32_t x = x0 + xs, y = y0 + ys;
        if (x < 0 || y < 0 || x >= W || y >= H) continue;
        float w = wy[ys] * wx[xs];
        float *px = film_rgbw + ((size_t) y * W + x) * 4;
        px[0] += rgb.x * w; px[1] += rgb.y * w; px[2] += rgb.z * w;
    }
}
/* Adjoint of the normalised splat: the film value is linear in the splatted colour, so dL/d(colour) = sum of taps w * dL/dfilm. */
static v3 film_gather_normalized(const OScene *sc, const float *grad_rgb, v2 pos) {
    int32_t x0, y0; float wx[16], wy[16];
    int count = splat_weights(sc, pos, &x0, &y0, wx, wy);
    float sx = 0.f, sy = 0.f;
    for (int i = 0; i < count; ++i) { sx += wx[i]; sy += wy[i]; }
    float factor = sx * sy; factor = factor != 0.f ? 1.f / factor : 0.f;
    for (int i = 0; i < count; ++i) wx[i] *= factor;
    int32_t W = (int32_t) sc->sensor.crop_width, H = (int32_t) sc->sensor.crop_height;
    v3 g = V3(0.f, 0.f, 0.f);
    for (int ys = 0; ys < count; ++ys) for (int xs = 0; xs < count; ++xs) {
        int32_t x = x0 + xs, y = y0 + ys;
        if (x < 0 || y < 0 || x >= W || y >= H) continue;
        float w = wy[ys] * wx[xs];
        const float *px = grad_rgb + ((size_t) y * W + x) * 3;
        g.x = fmaf(px[0], w, g.x); g.y = fmaf(px[1], w, g.y); g.z = fmaf(px[2], w, g.z);
    }
    return g;
}
/* Gradient mode of the particle tracer (Integrator::render_backward, integrator.cpp:65-84, applied to ptracer.cpp): every splat's
   value is  radiance_c * unit_c  with unit_c free of the radiance, so d(sum film * grad) / d radiance_c = sum unit_c * gather_c.
   `acc` holds 3 doubles per emitter.  (Russian roulette: the reference differentiates 1/q THROUGH the throughput because
   ptracer.cpp:277 has no detach, unlike path.cpp:272; this restatement keeps q detached -- identical while depth <= rr_depth,
   which covers every case the reference tests.) */
typedef struct { const float *grad_image; double *acc;
                 /* all parameters (`grad` != NULL): every light-path vertex j with a diffuse BSDF also receives
                        d/d albedo_j = (1 / albedo_j) * sum_{k >= j} <image adjoint gathered at splat k, splat k>
                    -- albedo_j is a linear factor of the connection made AT j (bsdf eval) and of every later one (bsdf sample weight) */
                 float *grad; const mi355_param_info *layout; const int32_t *bsdf_slot, *tex_slot; v3 last_dv; } PtGrad;
static __thread PtGrad g_ptg = { NULL, NULL, NULL, NULL, NULL, NULL, { 0.f, 0.f, 0.f } };
#define PT_MAXV 64
static __thread int g_geo_conn = 0;                                       /* set by connect_sensor: a diffuse connection reached the film (geometry derivative) */
static v3 xf_point_proj(const float *m, v3 p) {                        /* Transform4f * Point3f with the homogeneous divide */
    float r[4];
    for (int i = 0; i < 4; ++i) { float v = m[i*4+3]; v = fmaf(m[i*4+0], p.x, v); v = fmaf(m[i*4+1], p.y, v); v = fmaf(m[i*4+2], p.z, v); r[i] = v; }
    float iw = 1.f / r[3];
    return V3(r[0] * iw, r[1] * iw, r[2] * iw);
}
/* PerspectiveCamera::sample_direction + importance (perspective.cpp:282-386); image_rect / normalization :190-199.
   Returns 0 when the point is not seen; else uv (pixels, crop-relative), the sensor position, direction, weight. */
/* RealisticLensCamera::sample_direction (realisticlens.cpp:2296-2345): a point of the FRONT element is sampled, the ray from the
   reference point through it is traced world -> film (interfaces in reverse, :2061-2075), the film hit gives the pixel. */
static int lens_sample_direction(const OScene *sc, v3 ref_p, v2 sample, v2 *uv, v3 *ds_p_local, float *inv_dist, float *weight) {
    const float *Lf = sc->lens + ORC_LENS_ROW * (sc->lens_count - 1);
    v2 dk = square_to_uniform_disk_concentric(sample);
    v3 aperture_p = V3(dk.x * Lf[2], dk.y * Lf[2], Lf[3]);
    v3 dir = sub3(ref_p, aperture_p); float dist_ref = norm3(dir), inv_dist_ref = 1.f / dist_ref;
    dir = scale3(dir, inv_dist_ref);
    v3 o = ref_p, d = neg3(dir);
    for (int32_t k = (int32_t) sc->lens_count - 1; k >= 0; --k) if (!lens_interact(sc->lens + ORC_LENS_ROW * k, &o, &d)) return 0;
    float tt = -o.z / d.z; v3 fp = fma3s(d, tt, o);
    /* m_film_to_sample: physical film -> [0,1]^2 of the crop window (inverse of the mapping in sample_lens_ray) */
    float rsx = (float) sc->sensor.crop_width / (float) sc->sensor.film_width, rsy = (float) sc->sensor.crop_height / (float) sc->sensor.film_height;
    float rox = (float) sc->sensor.crop_offset_x / (float) sc->sensor.film_width, roy = (float) sc->sensor.crop_offset_y / (float) sc->sensor.film_height;
    float sx = ((fp.x / sc->film_px + 1.f) * 0.5f - rox) / rsx, sy = ((fp.y / sc->film_py + 1.f) * 0.5f - roy) / rsy;
    int valid = sx >= 0.f && sx <= 1.f && sy >= 0.f && sy <= 1.f;
    if (!valid) return 0;
    float normalization = 1.f / ((2.f * sc->film_px * rsx) * (2.f * sc->film_py * rsy));                  /* 1 / area of the crop window on the film, :2197-2203 */
    float ct_film = -d.z, inv_ct = 1.f / ct_film, rear_z = sc->lens[3];
    float value = normalization * (inv_ct * inv_ct) * (inv_ct * inv_ct) * (rear_z * rear_z);
    uv->x = sx * (float) sc->sensor.crop_width; uv->y = sy * (float) sc->sensor.crop_height;
    *ds_p_local = aperture_p; *inv_dist = inv_dist_ref;
    *weight = value * inv_dist_ref * inv_dist_ref * dir.z;                                                  /* ... * ct_ref, :2344 */
    return 1;
}
static int sensor_sample_direction(const OScene *sc, v3 p, v2 sample, v2 *uv, v3 *ds_p, v3 *ds_d, float *weight) {
    const float *tw = sc->sensor.to_world;
    /* inverse of the rigid camera frame (no scale: perspective.cpp:144-145) */
    v3 rel = sub3(p, V3(tw[3], tw[7], tw[11]));
    v3 ref_p = V3(tw[0] * rel.x + tw[4] * rel.y + tw[8] * rel.z, tw[1] * rel.x + tw[5] * rel.y + tw[9] * rel.z, tw[2] * rel.x + tw[6] * rel.y + tw[10] * rel.z);
    if (!(ref_p.z >= sc->sensor.near_clip && ref_p.z <= sc->sensor.far_clip)) return 0;
    if (sc->sensor.type == MI355_SENSOR_REALISTICLENS) {
        v3 ap; float inv_dist;
        if (!lens_sample_direction(sc, ref_p, sample, uv, &ap, &inv_dist, weight)) return 0;
        *ds_p = xf_point(tw, ap); *ds_d = scale3(sub3(*ds_p, p), inv_dist);
        return 1;
    }
    v3 screen = xf_point_proj(sc->camera_to_sample, ref_p);
    if (!(screen.x >= 0.f && screen.x <= 1.f && screen.y >= 0.f && screen.y <= 1.f)) return 0;
    uv->x = screen.x * (float) sc->sensor.crop_width; uv->y = screen.y * (float) sc->sensor.crop_height;
    float dist = norm3(ref_p), inv_dist = 1.f / dist;
    v3 local_d = scale3(ref_p, inv_dist);
    *ds_p = V3(tw[3], tw[7], tw[11]);
    *ds_d = scale3(sub3(*ds_p, p), inv_dist);
    v3 pmin = xf_point_proj(sc->sample_to_camera, V3(0.f, 0.f, 0.f)), pmax = xf_point_proj(sc->sample_to_camera, V3(1.f, 1.f, 0.f));
    float ax = pmin.x / pmin.z, ay = pmin.y / pmin.z, bx = pmax.x / pmax.z, by = pmax.y / pmax.z;
    float lox = fminf(ax, bx), hix = fmaxf(ax, bx), loy = fminf(ay, by), hiy = fmaxf(ay, by);
    float normalization = 1.f / ((hix - lox) * (hiy - loy));
    float ct = local_d.z, inv_ct = 1.f / ct, qx = local_d.x * inv_ct, qy = local_d.y * inv_ct;
    int valid = ct > 0.f && qx >= lox && qx <= hix && qy >= loy && qy <= hiy;
    float importance = valid ? normalization * inv_ct * inv_ct * inv_ct : 0.f;
    *weight = importance * inv_dist * inv_dist;
    return 1;
}
/* Shape::sample_position: rectangle.cpp:166-180, sphere.cpp:210-229, mesh.cpp:749-803 */
static void shape_sample_position(const OShape *sh, v2 sample, v3 *p, v3 *n, float *pdf) {
    if (sh->type == MI355_SHAPE_MESH) { mesh_sample_position(sh, sample, p, n, pdf); return; }
    if (sh->type == MI355_SHAPE_DISK) {                                 /* disk.cpp:173-191 */
        v2 q = square_to_uniform_disk_concentric(sample);
        *p = xf_point(sh->to_world, V3(q.x, q.y, 0.f)); *n = sh->frame_n; *pdf = sh->inv_area; return;
    }
    if (sh->type == MI355_SHAPE_RECTANGLE) {
        *p = xf_point(sh->to_world, V3(sample.x * 2.f - 1.f, sample.y * 2.f - 1.f, 0.f)); *n = sh->frame_n; *pdf = sh->inv_area;
    } else {
        float z = fmaf(-2.f, sample.y, 1.f), r = safe_sqrt_(fmaf(-z, z, 1.f)), ph = 2.f * ORC_PI * sample.x;   /* square_to_uniform_sphere, warp.h:250-255 */
        v3 local = V3(r * cosf(ph), r * sinf(ph), z);
        *p = fma3s(local, sh->radius, sh->center); *n = sh->flip_normals ? neg3(local) : local; *pdf = sh->inv_area;
    }
}
/* a SurfaceInteraction built from a PositionSample: frame from the normal (interaction.h: SurfaceInteraction(ps, wavelengths)) */
static SI si_from_position(v3 p, v3 n, int32_t shape) {
    SI si; memset(&si, 0, sizeof(si));
    si.p = p; si.n = n; si.sh_n = n; coordinate_system(n, &si.sh_s, &si.sh_t); si.shape = shape; si.t = 0.f;
    return si;
}
/* connect_sensor, ptracer.cpp:286-371.  bsdf = NULL for a point on an emitter. */
static void connect_sensor(OScene *sc, const SI *si, const mi355_bsdf_desc *bsdf, int have_sensor, v2 uv, v3 ds_p, v3 weight,
                           float sample_scale, float *film_rgbw, Cnt *cnt, v3 weight_unit, uint32_t emitter) {
    if (!have_sensor || (weight.x == 0.f && weight.y == 0.f && weight.z == 0.f)) return;
    Ray sensor_ray = spawn_ray_to(si->p, si->n, ds_p);
    cnt->shadow++;
    if (scene_ray_test(sc, &sensor_ray)) return;
    v3 local_d = si_to_local(si, sensor_ray.d);
    v3 surface_weight = V3(1.f, 1.f, 1.f);
    if (bsdf == NULL) surface_weight = scale3(surface_weight, fmaxf(0.f, local_d.z));                  /* foreshortening of the emitter */
    else {
        float wi_dot_geo_n = dot3(si->n, si_to_world(si, si->wi)), wo_dot_geo_n = dot3(si->n, sensor_ray.d);
        int valid = wi_dot_geo_n * si->wi.z > 0.f && wo_dot_geo_n * local_d.z > 0.f;
        float correction = valid ? fabsf((si->wi.z * wo_dot_geo_n) / (local_d.z * wi_dot_geo_n)) : 0.f;
        v3 refl = bsdf_reflectance(sc, bsdf, si), val; float pdf;
        bsdf_eval_pdf(bsdf, refl, si->wi, local_d, 1, &val, &pdf);
        surface_weight = scale3(val, correction);
        g_geo_conn = bsdf->type == MI355_BSDF_DIFFUSE && valid && si->wi.z > 0.f && local_d.z > 0.f;   /* (geometry derivative: a diffuse connection was made) */
    }
    v3 result = scale3(mul3(weight, surface_weight), sample_scale);
    v2 pos = { uv.x + (float) sc->sensor.crop_offset_x, uv.y + (float) sc->sensor.crop_offset_y };
    if (g_ptg.grad_image) {
        v3 ru = scale3(mul3(weight_unit, surface_weight), sample_scale), g = film_gather_normalized(sc, g_ptg.grad_image, pos);
        double *a = g_ptg.acc + 3 * (size_t) emitter;
        a[0] += (double) (ru.x * g.x); a[1] += (double) (ru.y * g.y); a[2] += (double) (ru.z * g.z);
        g_ptg.last_dv = mul3(result, g);
        return;
    }
    film_put_normalized(sc, film_rgbw, pos, result);
}
static uint32_t pick_emitter(const OScene *sc, float *sample, float *weight) {                     /* Scene::sample_emitter, scene.cpp:227-250 */
    *weight = 1.f;
    if (sc->n_emitters < 2) return 0;
    float cf = (float) sc->n_emitters, scaled = *sample * cf;
    uint32_t index = (uint32_t) scaled; if (index > sc->n_emitters - 1) index = sc->n_emitters - 1;
    *weight = cf; *sample = scaled - (float) index;
    return index;
}
#include "mi_oracle_geo.inc"
static void ptracer_sample(OScene *sc, const mi355_render_params *rp, pcg32 *rng_io, float sample_scale, float *film_rgbw, Cnt *cnt);
/* One sampler stream: seeded once (integrator.cpp:708), then one sample per pass from the SAME stream (:719-727; the independent
   sampler's advance() does not touch the generator).  Masked sampler calls of the reference never advance a masked-out lane, which
   the per-lane control flow below reproduces, so the second pass starts exactly where the reference's does. */
static void ptracer_lane(OScene *sc, const mi355_render_params *rp, uint32_t lane, float sample_scale, float *film_rgbw, Cnt *cnt) {
    pcg32 rng; sampler_seed(&rng, rp->seed, lane);
    uint32_t passes = rp->passes ? rp->passes : 1u;
    for (uint32_t i = 0; i < passes; ++i) ptracer_sample(sc, rp, &rng, sample_scale, film_rgbw, cnt);
}
static void ptracer_sample(OScene *sc, const mi355_render_params *rp, pcg32 *rng_io, float sample_scale, float *film_rgbw, Cnt *cnt) {
    pcg32 rng = *rng_io;
    const int timed = sc->sensor.shutter_open_time > 0.f;
    const int32_t max_depth = rp->max_depth;
    /* ---- sample_visible_emitters, ptracer.cpp:99-171 */
    if (max_depth != 0 && !(rp->flags & MI355_FLAG_HIDE_EMITTERS)) {
        if (timed) (void) pcg32_float(&rng);
        float es = pcg32_float(&rng), idx_weight; uint32_t ei = pick_emitter(sc, &es, &idx_weight);
        const mi355_emitter_desc *em = &sc->emitters[ei]; const OShape *sh = &sc->shapes[em->shape];
        /* "Don't connect delta emitters with sensor" (:114): for such a lane the MASKED sampler calls below (next_2d(active),
           next_1d(active)) leave its stream where it is -- only the sensor's unmasked next_2d() advances it */
        const int connect = em->type != MI355_EMITTER_DIRECTIONALAREA;
        /* 3.a (:121-141): in the JIT variants `dr::any_or<true>(active_e)` is true whatever the mask holds, so the infinite-emitter
           branch runs and its `sampler->next_2d(active)` -- masked by `active`, not by `active_e` -- consumes two numbers of every
           connectable lane although no emitter here is infinite */
        if (connect) { (void) pcg32_float(&rng); (void) pcg32_float(&rng); }
        v2 s2 = { 0.f, 0.f }; if (connect) { s2.x = pcg32_float(&rng); s2.y = pcg32_float(&rng); }
        v3 p, n; float pdf; shape_sample_position(sh, s2, &p, &n, &pdf);
        float pos_weight = (connect && pdf > 0.f) ? 1.f / pdf : 0.f;
        SI si = si_from_position(p, n, (int32_t) em->shape);
        v2 ss; ss.x = pcg32_float(&rng); ss.y = pcg32_float(&rng);                                 /* sensor->sample_direction's 2D sample (front-element point of a lens) */
        v2 uv; v3 ds_p, ds_d; float sensor_weight = 0.f;
        int seen = connect && sensor_sample_direction(sc, si.p, ss, &uv, &ds_p, &ds_d, &sensor_weight);
        si.wi = ds_d;
        if (connect) (void) pcg32_float(&rng);                                                     /* sample_wavelengths: radiance in RGB mode */
        v3 rad = V3(em->radiance[0], em->radiance[1], em->radiance[2]);
        v3 weight = scale3(rad, idx_weight * pos_weight * sensor_weight);
        float wu = idx_weight * pos_weight * sensor_weight;
        connect_sensor(sc, &si, NULL, seen, uv, ds_p, weight, sample_scale, film_rgbw, cnt, V3(wu, wu, wu), ei);
    }
    /* ---- prepare_ray, ptracer.cpp:174-191 -> Scene::sample_emitter_ray (scene.cpp:262-297) -> AreaLight::sample_ray (area.cpp:92-115) */
    if (timed) (void) pcg32_float(&rng);
    float wav = pcg32_float(&rng), idx_weight; uint32_t ei = pick_emitter(sc, &wav, &idx_weight);
    v2 sa, sb; sa.x = pcg32_float(&rng); sa.y = pcg32_float(&rng); sb.x = pcg32_float(&rng); sb.y = pcg32_float(&rng);
    const mi355_emitter_desc *em = &sc->emitters[ei]; const OShape *esh = &sc->shapes[em->shape];
    v3 ep, en; float epdf; shape_sample_position(esh, sa, &ep, &en, &epdf);
    SI esi = si_from_position(ep, en, (int32_t) em->shape);
    v3 local = square_to_cosine_hemisphere(sb);
    Ray ray = spawn_ray(esi.p, esi.n, si_to_world(&esi, local));
    v3 throughput = scale3(V3(em->radiance[0], em->radiance[1], em->radiance[2]), (epdf > 0.f ? 1.f / epdf : 0.f) * ORC_PI * idx_weight);
    float tu0 = (epdf > 0.f ? 1.f / epdf : 0.f) * ORC_PI * idx_weight;
    if (em->type == MI355_EMITTER_DIRECTIONALAREA) tu0 = (1.f / esh->inv_area) * idx_weight;
    v3 tu = V3(tu0, tu0, tu0);                                          /* throughput / radiance (gradient mode) */
    if (em->type == MI355_EMITTER_DIRECTIONALAREA) {                    /* DirectionalArea::sample_ray, directionalarea.cpp:88-113: along the normal, weight = area * radiance */
        ray = spawn_ray(esi.p, esi.n, esi.n);
        throughput = scale3(V3(em->radiance[0], em->radiance[1], em->radiance[2]), (1.f / esh->inv_area) * idx_weight);
    }
    int active = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z)) != 0.f;
    if (g_geo.on) { g_geo.nv = 0; g_geo.o0 = ray.o; g_geo.d0 = ray.d; g_geo.T0 = throughput; }
    v3 v_dv[PT_MAXV], v_rho[PT_MAXV]; int32_t v_bsdf[PT_MAXV], v_tidx[PT_MAXV][4]; float v_twgt[PT_MAXV][4]; int nv = 0;
    /* ---- trace_light_ray, ptracer.cpp:200-284 */
    float eta = 1.f; int32_t depth = 1;
    PI pi; SI si;
    if (active) { pi = scene_intersect(sc, &ray); cnt->isect++; si = compute_si(sc, &ray, &pi); active = si.shape >= 0; }
    if (max_depth >= 0) active = active && depth < max_depth;
    while (active) {
        cnt->shade++;
        const mi355_bsdf_desc *bsdf = &sc->bsdfs[sc->shapes[si.shape].bsdf];
        v2 ss; ss.x = pcg32_float(&rng); ss.y = pcg32_float(&rng);
        v2 uv; v3 ds_p, ds_d; float sensor_weight = 0.f;
        int seen = sensor_sample_direction(sc, si.p, ss, &uv, &ds_p, &ds_d, &sensor_weight);
        g_ptg.last_dv = V3(0.f, 0.f, 0.f); g_geo_conn = 0;
        connect_sensor(sc, &si, bsdf, seen, uv, ds_p, scale3(throughput, sensor_weight), sample_scale, film_rgbw, cnt, scale3(tu, sensor_weight), ei);
        float s1 = pcg32_float(&rng); v2 s2; s2.x = pcg32_float(&rng); s2.y = pcg32_float(&rng);
        v3 refl = bsdf_reflectance(sc, bsdf, &si); BSample bs; v3 bsdf_val;
        GeoVtx *gvx = (g_geo.on && g_geo.nv < GEO_MAXV) ? &g_geo.v[g_geo.nv++] : NULL;
        if (g_ptg.grad && nv < PT_MAXV) {
            v_dv[nv] = g_ptg.last_dv; v_rho[nv] = refl; v_bsdf[nv] = (int32_t) sc->shapes[si.shape].bsdf;
            if (bsdf->reflectance_texture >= 0) (void) texture_eval(&sc->textures[bsdf->reflectance_texture], si.uv, v_tidx[nv], v_twgt[nv]);
            nv++;
        }
        bsdf_sample(bsdf, refl, si.wi, s1, s2, 1, &bs, &bsdf_val);
        v3 wo_world = si_to_world(&si, bs.wo);
        float wi_dot_geo_n = dot3(si.n, neg3(ray.d)), wo_dot_geo_n = dot3(si.n, wo_world);
        active = active && wi_dot_geo_n * si.wi.z > 0.f && wo_dot_geo_n * bs.wo.z > 0.f;
        float correction = fabsf((si.wi.z * wo_dot_geo_n) / (bs.wo.z * wi_dot_geo_n));
        throughput = mul3(throughput, scale3(bsdf_val, correction)); tu = mul3(tu, scale3(bsdf_val, correction));
        eta *= bs.eta;
        active = active && (throughput.x != 0.f || throughput.y != 0.f || throughput.z != 0.f);
        if (gvx) {
            gvx->shape = (uint32_t) si.shape; gvx->prim = si.prim_index; gvx->t_hit = si.t; gvx->sh_s = si.sh_s; gvx->sh_t = si.sh_t;
            gvx->lobe = bs.delta ? (bs.wo.z * si.wi.z > 0.f ? 2 : 1) : 0; gvx->conn = g_geo_conn; gvx->cont = active;
            gvx->l = bs.wo; gvx->W = bsdf_val; gvx->refl = refl; gvx->rr = 1.f;
            float mag = (1.f + fmaxf(fmaxf(fabsf(si.p.x), fabsf(si.p.y)), fabsf(si.p.z))) * RAY_EPS;
            gvx->mag = mulsign(mag, dot3(si.n, wo_world));
        }
        if (!active) break;
        ray = spawn_ray(si.p, si.n, wo_world);
        /* :264-269: the reference intersects the sampled ray and THEN masks by depth < max_depth, so the intersection after the last vertex that may
           be shaded is computed and never read.  It is performed here as written, but counted apart (c_isect_dead): the device skips it. */
        pi = scene_intersect(sc, &ray); if (max_depth >= 0 && depth + 1 >= max_depth) cnt->isect_dead++; else cnt->isect++;
        si = compute_si(sc, &ray, &pi);
        depth++;
        if (max_depth >= 0) active = active && depth < max_depth;
        active = active && si.shape >= 0;
        if (depth > rp->rr_depth) {                                                                  /* :271-277 */
            float q = fminf(fmaxf(throughput.x, fmaxf(throughput.y, throughput.z)) * (eta * eta), 0.95f);
            active = active && pcg32_float(&rng) < q;       /* next_1d(active): dead lanes still advance their stream -- irrelevant, they are done */
            throughput = scale3(throughput, 1.f / q); tu = scale3(tu, 1.f / q);
            if (gvx) gvx->rr = 1.f / q;
        }
    }
    if (g_ptg.grad) {                                                   /* suffix sums over the path's connections */
        v3 S = V3(0.f, 0.f, 0.f);
        for (int j = nv - 1; j >= 0; --j) {
            S = add3(S, v_dv[j]);
            const mi355_bsdf_desc *b = &sc->bsdfs[v_bsdf[j]];
            if (b->type != MI355_BSDF_DIFFUSE) continue;
            v3 ga = V3(v_rho[j].x != 0.f ? S.x / v_rho[j].x : 0.f, v_rho[j].y != 0.f ? S.y / v_rho[j].y : 0.f, v_rho[j].z != 0.f ? S.z / v_rho[j].z : 0.f);
            if (b->reflectance_texture >= 0) {
                int32_t slot = g_ptg.tex_slot[b->reflectance_texture];
                if (slot < 0) continue;
                const OTexture *t = &sc->textures[b->reflectance_texture];
                float *gb = g_ptg.grad + g_ptg.layout[slot].offset;
                for (int k = 0; k < 4; ++k) if (v_tidx[j][k] >= 0) {
                    if (t->c == 3) { gb[v_tidx[j][k]*3+0] += v_twgt[j][k] * ga.x; gb[v_tidx[j][k]*3+1] += v_twgt[j][k] * ga.y; gb[v_tidx[j][k]*3+2] += v_twgt[j][k] * ga.z; }
                    else gb[v_tidx[j][k]] += v_twgt[j][k] * (ga.x + ga.y + ga.z);
                }
            } else {
                int32_t slot = g_ptg.bsdf_slot[v_bsdf[j]];
                if (slot >= 0) { float *gb = g_ptg.grad + g_ptg.layout[slot].offset; gb[0] += ga.x; gb[1] += ga.y; gb[2] += ga.z; }
            }
        }
    }
    if (g_geo.on) geo_eval_path(sc, g_ptg.grad_image);
    *rng_io = rng;
}
/* The shard of the particle tracer's lane grid (film pixels x sampler streams per pixel) a call covers: mi355_render_params's pixel /
   tile / spp fields over the FILM grid, exactly as the engine's make_args reads them. */
static inline uint32_t shard_pixel(const mi355_render_params *rp, uint32_t q);
static uint64_t pt_shard_lanes(const OScene *sc, const mi355_render_params *rp) {
    uint32_t passes = rp->passes ? rp->passes : 1u, streams = rp->spp / passes, npix = sc->sensor.film_width * sc->sensor.film_height;
    uint32_t pn = rp->pixel_count ? rp->pixel_count : npix - rp->pixel_begin, sn = rp->spp_count ? rp->spp_count : streams - rp->spp_begin;
    if (rp->pixel_begin >= npix && rp->pixel_count == 0) return 0;
    return (uint64_t) pn * sn;
}
static uint32_t pt_shard_lane(const OScene *sc, const mi355_render_params *rp, uint64_t i) {
    uint32_t passes = rp->passes ? rp->passes : 1u, streams = rp->spp / passes;
    uint32_t sn = rp->spp_count ? rp->spp_count : streams - rp->spp_begin;
    (void) sc;
    return shard_pixel(rp, (uint32_t) (i / sn)) * streams + rp->spp_begin + (uint32_t) (i % sn);
}
/* AdjointIntegrator::render: lanes = FILM size * spp (not crop), sample_scale = prod(crop) / (spp * prod(film)).  The
   accumulator is RGBW with W untouched = 0, so orc_film_develop divides by 1 (hdrfilm.cpp:396-400). */
ORC_API uint64_t orc_ptracer_accumulate(OScene *sc, const mi355_render_params *rp, float *film_rgbw, int n_threads) {
    orc_select_sensor(sc, rp->sensor);
    size_t fsz = (size_t) sc->sensor.crop_width * sc->sensor.crop_height * 4;
    uint32_t passes = rp->passes ? rp->passes : 1u;
    if (rp->spp % passes) return 0;
    uint64_t lanes = pt_shard_lanes(sc, rp);
    if (sc->n_emitters == 0) return 0;                                                              /* integrator.cpp:590-600 */
    float sample_scale = (float) (sc->sensor.crop_width * sc->sensor.crop_height) / (float) ((uint64_t) rp->spp * sc->sensor.film_width * sc->sensor.film_height);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    float *films = (float *) calloc(fsz * (size_t) n_threads, sizeof(float));
    uint64_t ci = 0, cs = 0, csh = 0;
    #pragma omp parallel num_threads(n_threads) reduction(+:ci, cs, csh)
    {
#ifdef _OPENMP
        float *film = films + fsz * (size_t) omp_get_thread_num();
#else
        float *film = films;
#endif
        Cnt cnt = { 0, 0, 0, 0 };
        g_importance_mode = 1;
        #pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < (int64_t) lanes; ++i) ptracer_lane(sc, rp, pt_shard_lane(sc, rp, (uint64_t) i), sample_scale, film, &cnt);
        g_importance_mode = 0;
        ci += cnt.isect; cs += cnt.shade; csh += cnt.shadow;
        #pragma omp atomic
        sc->c_isect_dead += cnt.isect_dead;
    }
    for (int t = 0; t < n_threads; ++t) for (size_t i = 0; i < fsz; ++i) film_rgbw[i] += films[fsz * (size_t) t + i];
    free(films);
    sc->c_isect += ci; sc->c_shade += cs; sc->c_shadow += csh;
    return lanes;
}

static int64_t emitter_grad_offset(const OScene *sc, uint32_t e) {       /* orc_param_layout: reflectances, textures, then radiances */
    uint64_t off = 0;
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) if (sc->bsdfs[i].type == MI355_BSDF_DIFFUSE && sc->bsdfs[i].reflectance_texture < 0) off += 3;
    for (uint32_t i = 0; i < sc->n_textures; ++i) off += (uint64_t) sc->textures[i].w * sc->textures[i].h * sc->textures[i].c;
    return (int64_t) (off + 3 * (uint64_t) e);
}
/* render_backward of the particle tracer for the emitter radiances: grad_image is dL/d(image) (crop_h x crop_w x 3; the developed
   image IS the RGB accumulator because W = 0), grad_out the concatenated parameter-gradient buffer (orc_scene_param_layout); only
   the MI355_PARAM_EMITTER_RADIANCE slots are written (+=). */
ORC_API uint64_t orc_ptracer_backward(OScene *sc, const mi355_render_params *rp, const float *grad_image, float *grad_out, int n_threads) {
    orc_select_sensor(sc, rp->sensor);
    const int all_params = !(rp->flags & MI355_FLAG_EMITTER_GRADS_ONLY);
    mi355_param_info layout[4096]; uint64_t total = 0; int np = orc_param_layout(sc, layout, 4096, &total);
    int32_t *bsdf_slot = (int32_t *) malloc(4 * (sc->n_bsdfs + 1)), *tex_slot = (int32_t *) malloc(4 * (sc->n_textures + 1));
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) bsdf_slot[i] = -1;
    for (uint32_t i = 0; i < sc->n_textures; ++i) tex_slot[i] = -1;
    for (int i = 0; i < np; ++i) {
        if (layout[i].kind == MI355_PARAM_BSDF_REFLECTANCE) bsdf_slot[layout[i].index] = i;
        else if (layout[i].kind == MI355_PARAM_TEXTURE_DATA) tex_slot[layout[i].index] = i;
    }
    uint32_t passes = rp->passes ? rp->passes : 1u;
    if (rp->spp % passes || sc->n_emitters == 0) return 0;
    uint64_t lanes = pt_shard_lanes(sc, rp);
    float sample_scale = (float) (sc->sensor.crop_width * sc->sensor.crop_height) / (float) ((uint64_t) rp->spp * sc->sensor.film_width * sc->sensor.film_height);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    size_t na = 3 * (size_t) sc->n_emitters;
    double *accs = (double *) calloc(na * (size_t) n_threads, sizeof(double));
    float *grads = all_params ? (float *) calloc((size_t) total * (size_t) n_threads, sizeof(float)) : NULL;
    /* MI355_FLAG_GEOMETRY_GRADS: vertex positions and IORs (mi_oracle_geo.inc) */
    const int geo = (rp->flags & MI355_FLAG_GEOMETRY_GRADS) != 0;
    uint64_t *pos_off = (uint64_t *) malloc(8 * (sc->n_shapes + 1)), *nrm_base = (uint64_t *) malloc(8 * (sc->n_shapes + 1)), *ior_off = (uint64_t *) malloc(8 * (sc->n_bsdfs + 1));
    uint64_t nrm_total = 0;
    for (uint32_t i = 0; i < sc->n_shapes; ++i) { pos_off[i] = (uint64_t) -1; nrm_base[i] = nrm_total; if (sc->shapes[i].type == MI355_SHAPE_MESH) nrm_total += 3ull * sc->shapes[i].vertex_count; }
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) ior_off[i] = (uint64_t) -1;
    for (int i = 0; i < np; ++i) {
        if (layout[i].kind == MI355_PARAM_VERTEX_POSITIONS) pos_off[layout[i].index] = layout[i].offset;
        else if (layout[i].kind == MI355_PARAM_BSDF_IOR) ior_off[layout[i].index] = layout[i].offset;
    }
    double *geo_grad = geo ? (double *) calloc((size_t) total * (size_t) n_threads + 1, sizeof(double)) : NULL;
    double *geo_nrm = geo ? (double *) calloc((size_t) nrm_total * (size_t) n_threads + 1, sizeof(double)) : NULL;
    #pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        memset(&g_geo, 0, sizeof(g_geo));
        if (geo) { g_geo.on = 1; g_geo.grad = geo_grad + (size_t) total * (size_t) tid; g_geo.nrm_adj = geo_nrm + (size_t) nrm_total * (size_t) tid;
                   g_geo.pos_off = pos_off; g_geo.ior_off = ior_off; g_geo.nrm_base = nrm_base; g_geo.sample_scale = sample_scale; }
#ifdef _OPENMP
        double *acc = accs + na * (size_t) omp_get_thread_num();
        float *gl = grads ? grads + total * (size_t) omp_get_thread_num() : NULL;
#else
        double *acc = accs; float *gl = grads;
#endif
        Cnt cnt = { 0, 0, 0, 0 };
        g_importance_mode = 1; g_ptg.grad_image = grad_image; g_ptg.acc = acc;
        g_ptg.grad = gl; g_ptg.layout = layout; g_ptg.bsdf_slot = bsdf_slot; g_ptg.tex_slot = tex_slot;
        #pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < (int64_t) lanes; ++i) ptracer_lane(sc, rp, pt_shard_lane(sc, rp, (uint64_t) i), sample_scale, NULL, &cnt);
        g_importance_mode = 0; g_ptg.grad_image = NULL; g_ptg.acc = NULL; g_ptg.grad = NULL; g_geo.on = 0;
    }
    if (geo) {
        for (int t = 1; t < n_threads; ++t) {
            for (uint64_t i = 0; i < total; ++i) geo_grad[i] += geo_grad[(size_t) total * (size_t) t + i];
            for (uint64_t i = 0; i < nrm_total; ++i) geo_nrm[i] += geo_nrm[(size_t) nrm_total * (size_t) t + i];
        }
        for (uint32_t i = 0; i < sc->n_shapes; ++i) {       /* vertex normals that follow their positions: carry their adjoint on */
            const OShape *sh = &sc->shapes[i];
            if (sh->type == MI355_SHAPE_MESH && sh->nrm && sh->nrm_dep && pos_off[i] != (uint64_t) -1) geo_normals_to_positions(sh, geo_nrm + nrm_base[i], geo_grad + pos_off[i]);
        }
        for (int i = 0; i < np; ++i) if (layout[i].kind == MI355_PARAM_VERTEX_POSITIONS || layout[i].kind == MI355_PARAM_BSDF_IOR)
            for (uint64_t k = 0; k < layout[i].size; ++k) grad_out[layout[i].offset + k] += (float) geo_grad[layout[i].offset + k];
    }
    free(geo_grad); free(geo_nrm); free(pos_off); free(nrm_base); free(ior_off);
    if (grads) { for (int t = 0; t < n_threads; ++t) for (uint64_t i = 0; i < total; ++i) grad_out[i] += grads[total * (size_t) t + i]; free(grads); }
    free(bsdf_slot); free(tex_slot);
    for (uint32_t e = 0; e < sc->n_emitters; ++e) {
        int64_t off = emitter_grad_offset(sc, e);
        if (off < 0) continue;
        for (int c = 0; c < 3; ++c) { double v = 0.0; for (int t = 0; t < n_threads; ++t) v += accs[na * (size_t) t + 3 * e + c]; grad_out[off + c] += (float) v; }
    }
    free(accs);
    return lanes;
}

/* ----------------------------------------------------------------------------------------
 * drivers
 * ---------------------------------------------------------------------------------------- */
static void shard_bounds(const OScene *sc, const mi355_render_params *rp, uint32_t *p0, uint32_t *pn, uint32_t *s0, uint32_t *sn) {
    uint32_t npix = (sc->sensor.crop_width + 2u * sc->border) * (sc->sensor.crop_height + 2u * sc->border);
    *p0 = rp->pixel_begin; *pn = rp->pixel_count ? rp->pixel_count : npix - rp->pixel_begin;
    uint32_t streams = rp->spp / (rp->passes ? rp->passes : 1u);         /* sampler streams per pixel (mi355.h: passes) */
    *s0 = rp->spp_begin; *sn = rp->spp_count ? rp->spp_count : streams - rp->spp_begin;
}
static inline uint32_t shard_pixel(const mi355_render_params *rp, uint32_t q) {     /* mi355.h: pixel_tile */
    if (rp->pixel_tile == 0u) return rp->pixel_begin + q;
    return rp->pixel_begin + (q / rp->pixel_tile) * rp->pixel_tile_stride + q % rp->pixel_tile;
}
/* render_sample head (integrator.cpp:433-455): pixel jitter, aperture sample if the sensor needs one, ray + ray weight */
static float camera_ray(const OScene *sc, pcg32 *rng, uint32_t lane, uint32_t spp, v2 *sp, Ray *ray) {
    int32_t px, py; lane_to_pixel(sc, lane, spp, &px, &py);
    sp->x = (float) px + pcg32_float(rng); sp->y = (float) py + pcg32_float(rng);          /* integrator.cpp:435 */
    if (sc->sensor.type == MI355_SENSOR_REALISTICLENS) {
        v2 ap; ap.x = pcg32_float(rng); ap.y = pcg32_float(rng);                         /* needs_aperture_sample, :438-440 */
        if (sc->sensor.shutter_open_time > 0.f) (void) pcg32_float(rng);                 /* time sample, :442-444 */
        return sample_lens_ray(sc, *sp, ap, ray);
    }
    if (sc->sensor.shutter_open_time > 0.f) (void) pcg32_float(rng);                     /* time sample, :442-444 */
    *ray = sample_primary_ray(sc, *sp); return 1.f;
}
static void trace_lane(OScene *sc, const mi355_render_params *rp, uint32_t lane, v3 *rgb, int *valid, v2 *pos, Cnt *cnt) {
    pcg32 rng; sampler_seed(&rng, rp->seed, lane);
    Ray ray; float w = camera_ray(sc, &rng, lane, rp->spp, pos, &ray);
    *rgb = V3(0, 0, 0); *valid = 0;
    if (w == 0.f) return;                                               /* blocked by the lens barrel: weight 0, path not traced */
    if (rp->integrator == MI355_INTEGRATOR_PRB) prb_sample(sc, &rng, ray, rp, 0, V3(0,0,0), V3(0,0,0), NULL, rgb, valid, cnt);
    else path_sample(sc, &rng, ray, rp, rgb, valid, cnt);
    *rgb = scale3(*rgb, w);                                             /* spec_u = ray_weight * spec, integrator.cpp:461 */
}
/* per-lane radiance for lanes [lane_begin, lane_begin+count): out = count x 4 (r,g,b,valid) */
ORC_API void orc_lane_radiance(OScene *sc, const mi355_render_params *rp, uint32_t lane_begin, uint32_t count, float *out) {
    orc_select_sensor(sc, rp->sensor);
    #pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t) count; ++i) {
        v3 rgb; int valid; v2 pos; Cnt c = { 0, 0, 0 };
        trace_lane(sc, rp, lane_begin + (uint32_t) i, &rgb, &valid, &pos, &c);
        out[4*i] = rgb.x; out[4*i+1] = rgb.y; out[4*i+2] = rgb.z; out[4*i+3] = (float) valid;
    }
}
/* SamplingIntegrator::render JIT branch (integrator.cpp:246-360): accumulate the shard's lanes
   into film_rgbw (H x W x 4, not cleared).  Returns the number of lanes traced. */
ORC_API uint64_t orc_render_accumulate(OScene *sc, const mi355_render_params *rp, float *film_rgbw, int n_threads) {
    orc_select_sensor(sc, rp->sensor);
    uint32_t p0, pn, s0, sn; shard_bounds(sc, rp, &p0, &pn, &s0, &sn);
    size_t fsz = (size_t) sc->sensor.crop_width * sc->sensor.crop_height * 4;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    float *films = (float *) calloc(fsz * (size_t) n_threads, sizeof(float));
    uint64_t ci = 0, cs = 0, csh = 0;
    #pragma omp parallel num_threads(n_threads) reduction(+:ci, cs, csh)
    {
#ifdef _OPENMP
        float *film = films + fsz * (size_t) omp_get_thread_num();
#else
        float *film = films;
#endif
        Cnt cnt = { 0, 0, 0, 0 };
        #pragma omp for schedule(dynamic, 64)
        for (int64_t q = 0; q < (int64_t) pn; ++q) {
            const uint32_t passes = rp->passes ? rp->passes : 1u, streams = rp->spp / passes;
            for (uint32_t s = s0; s < s0 + sn; ++s) {
                uint32_t lane = shard_pixel(rp, (uint32_t) q) * streams + s;
                if (passes == 1u) {
                    v3 rgb; int valid; v2 pos;
                    trace_lane(sc, rp, lane, &rgb, &valid, &pos, &cnt);
                    float vals[4] = { rgb.x, rgb.y, rgb.z, 1.f };       /* integrator.cpp:477-486 (no alpha) */
                    film_put(sc, film, 4, pos, vals);
                    continue;
                }
                /* samples_per_pass (integrator.cpp:283, :319-329): one seeding, `passes` samples from the same stream */
                pcg32 rng; sampler_seed(&rng, rp->seed, lane);
                for (uint32_t pass = 0; pass < passes; ++pass) {
                    v2 pos; Ray ray; float w = camera_ray(sc, &rng, lane, streams, &pos, &ray);
                    v3 rgb = V3(0, 0, 0); int valid = 0;
                    g_path_keep_stream = 1; path_sample(sc, &rng, ray, rp, &rgb, &valid, &cnt); g_path_keep_stream = 0;
                    rgb = scale3(rgb, w);
                    float vals[4] = { rgb.x, rgb.y, rgb.z, 1.f };
                    film_put(sc, film, 4, pos, vals);
                }
            }
        }
        ci += cnt.isect; cs += cnt.shade; csh += cnt.shadow;
    }
    for (int t = 0; t < n_threads; ++t) for (size_t i = 0; i < fsz; ++i) film_rgbw[i] += films[fsz * (size_t) t + i];
    free(films);
    sc->c_isect += ci; sc->c_shade += cs; sc->c_shadow += csh;
    return (uint64_t) pn * sn * (rp->passes ? rp->passes : 1u);
}

/* parameter layout shared with the engine's definition (include/mi355.h: mi355_scene_param_layout):
   bsdf reflectances (constant ones), then textures, then emitter radiances, then the vertex positions of the meshes, then the IOR pairs
   of the dielectrics, in index order */
ORC_API int orc_param_layout(const OScene *sc, mi355_param_info *out, uint32_t cap, uint64_t *total) {
    uint32_t n = 0; uint64_t off = 0;
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) if (sc->bsdfs[i].type == MI355_BSDF_DIFFUSE && sc->bsdfs[i].reflectance_texture < 0) {
        if (n < cap) { out[n].kind = MI355_PARAM_BSDF_REFLECTANCE; out[n].index = i; out[n].offset = off; out[n].size = 3; } n++; off += 3; }
    for (uint32_t i = 0; i < sc->n_textures; ++i) {
        uint64_t sz = (uint64_t) sc->textures[i].w * sc->textures[i].h * sc->textures[i].c;
        if (n < cap) { out[n].kind = MI355_PARAM_TEXTURE_DATA; out[n].index = i; out[n].offset = off; out[n].size = sz; } n++; off += sz; }
    for (uint32_t i = 0; i < sc->n_emitters; ++i) {
        if (n < cap) { out[n].kind = MI355_PARAM_EMITTER_RADIANCE; out[n].index = i; out[n].offset = off; out[n].size = 3; } n++; off += 3; }
    for (uint32_t i = 0; i < sc->n_shapes; ++i) if (sc->shapes[i].type == MI355_SHAPE_MESH) {
        uint64_t sz = 3ull * sc->shapes[i].vertex_count;
        if (n < cap) { out[n].kind = MI355_PARAM_VERTEX_POSITIONS; out[n].index = i; out[n].offset = off; out[n].size = sz; } n++; off += sz; }
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) if (sc->bsdfs[i].type == MI355_BSDF_DIELECTRIC) {
        if (n < cap) { out[n].kind = MI355_PARAM_BSDF_IOR; out[n].index = i; out[n].offset = off; out[n].size = 2; } n++; off += 2; }
    if (total) *total = off;
    return (int) n;
}

/* RBIntegrator.render_backward step 1a (common.py:741-758): W[p] = sum of filter weights */
ORC_API void orc_prb_splat_weights(OScene *sc, const mi355_render_params *rp, float *weight_hw) {
    orc_select_sensor(sc, rp->sensor);
    uint32_t p0, pn, s0, sn; shard_bounds(sc, rp, &p0, &pn, &s0, &sn);
    for (uint32_t q = 0; q < pn; ++q) for (uint32_t s = s0; s < s0 + sn; ++s) {
        uint32_t lane = shard_pixel(rp, q) * rp->spp + s; pcg32 rng; sampler_seed(&rng, rp->seed, lane);
        int32_t px, py; lane_to_pixel(sc, lane, rp->spp, &px, &py);
        v2 sp; sp.x = (float) px + pcg32_float(&rng); sp.y = (float) py + pcg32_float(&rng);
        float one = 1.f; film_put(sc, weight_hw, 1, sp, &one);
    }
}
/* adjoint of splat + develop for one lane: dL_c = sum_taps grad_in[p,c] * w(p) / W[p] */
static v3 delta_L(const OScene *sc, v2 pos, const float *grad_in, const float *weight_hw) {
    int32_t x0, y0; float wx[16], wy[16]; int count = splat_weights(sc, pos, &x0, &y0, wx, wy);
    int32_t W = (int32_t) sc->sensor.crop_width, H = (int32_t) sc->sensor.crop_height;
    v3 r = V3(0, 0, 0);
    for (int ys = 0; ys < count; ++ys) for (int xs = 0; xs < count; ++xs) {
        int32_t x = x0 + xs, y = y0 + ys; if (x < 0 || y < 0 || x >= W || y >= H) continue;
        size_t i = (size_t) y * W + x; float wv = weight_hw[i]; float dv = wv == 0.f ? 1.f : wv;
        float w = wy[ys] * wx[xs] / dv;
        r.x += grad_in[3*i] * w; r.y += grad_in[3*i+1] * w; r.z += grad_in[3*i+2] * w;
    }
    return r;
}
/* RBIntegrator.render_backward steps 1b-3 (common.py:741-796); gradients are ACCUMULATED into grad */
ORC_API void orc_render_backward(OScene *sc, const mi355_render_params *rp, const float *grad_in,
                                 const float *weight_hw, float *grad, int n_threads) {
    orc_select_sensor(sc, rp->sensor);
    uint32_t p0, pn, s0, sn; shard_bounds(sc, rp, &p0, &pn, &s0, &sn);
    mi355_param_info layout[4096]; uint64_t total = 0; int np = orc_param_layout(sc, layout, 4096, &total);
    int32_t *bsdf_slot = (int32_t *) malloc(4 * (sc->n_bsdfs + 1)), *tex_slot = (int32_t *) malloc(4 * (sc->n_textures + 1)), *em_slot = (int32_t *) malloc(4 * (sc->n_emitters + 1));
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) bsdf_slot[i] = -1;
    for (uint32_t i = 0; i < sc->n_textures; ++i) tex_slot[i] = -1;
    for (uint32_t i = 0; i < sc->n_emitters; ++i) em_slot[i] = -1;
    for (int i = 0; i < np; ++i) {
        if (layout[i].kind == MI355_PARAM_BSDF_REFLECTANCE) bsdf_slot[layout[i].index] = i;
        else if (layout[i].kind == MI355_PARAM_TEXTURE_DATA) tex_slot[layout[i].index] = i;
        else if (layout[i].kind == MI355_PARAM_EMITTER_RADIANCE) em_slot[layout[i].index] = i;
    }
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    float *grads = (float *) calloc((size_t) total * (size_t) n_threads, sizeof(float));
    #pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        float *gl = grads + total * (size_t) omp_get_thread_num();
#else
        float *gl = grads;
#endif
        GradCtx g = { gl, layout, (uint32_t) np, bsdf_slot, tex_slot, em_slot };
        Cnt cnt = { 0, 0, 0, 0 };
        #pragma omp for schedule(dynamic, 64)
        for (int64_t q = 0; q < (int64_t) pn; ++q) for (uint32_t s = s0; s < s0 + sn; ++s) {
            uint32_t lane = shard_pixel(rp, (uint32_t) q) * rp->spp + s;
            pcg32 rng; sampler_seed(&rng, rp->seed, lane);
            v2 sp; Ray ray; float cw = camera_ray(sc, &rng, lane, rp->spp, &sp, &ray);
            if (cw == 0.f) continue;
            v3 dL = scale3(delta_L(sc, sp, grad_in, weight_hw), cw);       /* value = L * weight, common.py:752 */
            pcg32 rng2 = rng;                                            /* sampler.clone(), common.py:767 */
            v3 L; int valid;
            prb_sample(sc, &rng2, ray, rp, 0, V3(0,0,0), V3(0,0,0), NULL, &L, &valid, &cnt);
            v3 L2;
            prb_sample(sc, &rng, ray, rp, 1, dL, L, &g, &L2, &valid, &cnt);
        }
    }
    for (int t = 0; t < n_threads; ++t) for (uint64_t i = 0; i < total; ++i) grad[i] += grads[total * (size_t) t + i];
    free(grads); free(bsdf_slot); free(tex_slot); free(em_slot);
}

/* RBIntegrator.render_forward (common.py:498-629): primal walk (1), forward-mode walk (2) with the same sampler stream, dL splatted
   with weight 1 into an RGBW accumulator whose develop() is the gradient image.  `tangent` = one float per differentiable scene
   float, laid out like the gradient buffer (orc_param_layout): what dr.set_grad() put on the parameters. */
ORC_API uint64_t orc_render_forward(OScene *sc, const mi355_render_params *rp, const float *tangent, float *film_rgbw, int n_threads) {
    orc_select_sensor(sc, rp->sensor);
    uint32_t p0, pn, s0, sn; shard_bounds(sc, rp, &p0, &pn, &s0, &sn);
    mi355_param_info layout[4096]; uint64_t total = 0; int np = orc_param_layout(sc, layout, 4096, &total);
    int32_t *bsdf_slot = (int32_t *) malloc(4 * (sc->n_bsdfs + 1)), *tex_slot = (int32_t *) malloc(4 * (sc->n_textures + 1)), *em_slot = (int32_t *) malloc(4 * (sc->n_emitters + 1));
    for (uint32_t i = 0; i < sc->n_bsdfs; ++i) bsdf_slot[i] = -1;
    for (uint32_t i = 0; i < sc->n_textures; ++i) tex_slot[i] = -1;
    for (uint32_t i = 0; i < sc->n_emitters; ++i) em_slot[i] = -1;
    for (int i = 0; i < np; ++i) {
        if (layout[i].kind == MI355_PARAM_BSDF_REFLECTANCE) bsdf_slot[layout[i].index] = i;
        else if (layout[i].kind == MI355_PARAM_TEXTURE_DATA) tex_slot[layout[i].index] = i;
        else if (layout[i].kind == MI355_PARAM_EMITTER_RADIANCE) em_slot[layout[i].index] = i;
    }
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    size_t fsz = (size_t) sc->sensor.crop_width * sc->sensor.crop_height * 4;
    float *films = (float *) calloc(fsz * (size_t) n_threads, sizeof(float));
    #pragma omp parallel num_threads(n_threads)
    {
#ifdef _OPENMP
        float *film = films + fsz * (size_t) omp_get_thread_num();
#else
        float *film = films;
#endif
        Cnt cnt = { 0, 0, 0, 0 };
        #pragma omp for schedule(dynamic, 64)
        for (int64_t q = 0; q < (int64_t) pn; ++q) for (uint32_t s = s0; s < s0 + sn; ++s) {
            uint32_t lane = shard_pixel(rp, (uint32_t) q) * rp->spp + s;
            pcg32 rng; sampler_seed(&rng, rp->seed, lane);
            v2 sp; Ray ray; float cw = camera_ray(sc, &rng, lane, rp->spp, &sp, &ray);
            v3 dl = V3(0, 0, 0);
            if (cw != 0.f) {
                pcg32 rng2 = rng;                                            /* sampler.clone(), common.py:578 */
                v3 L; int valid;
                prb_sample(sc, &rng2, ray, rp, 0, V3(0,0,0), V3(0,0,0), NULL, &L, &valid, &cnt);
                GradCtx g = { (float *) tangent, layout, (uint32_t) np, bsdf_slot, tex_slot, em_slot, 1, &dl };
                v3 L2;
                prb_sample(sc, &rng, ray, rp, 1, V3(1, 1, 1), L, &g, &L2, &valid, &cnt);
            }
            float vals[4] = { dl.x * cw, dl.y * cw, dl.z * cw, 1.f };       /* value = dL * weight, weight = 1 (common.py:604-611) */
            film_put(sc, film, 4, sp, vals);
        }
    }
    for (int t = 0; t < n_threads; ++t) for (size_t i = 0; i < fsz; ++i) film_rgbw[i] += films[fsz * (size_t) t + i];
    free(films); free(bsdf_slot); free(tex_slot); free(em_slot);
    return (uint64_t) pn * sn;
}

/* ----------------------------------------------------------------------------------------
 * ray-level test hooks (Scene::ray_intersect / ray_test on caller rays)
 * rays: n x 8 (o, d, maxt, pad); hits: n x 8 (t, u, v, prim, shape, occluded, -, -);
 * si_out (optional): n x 24 (p3, n3, uv2, sh_n3, wi3, t, shape (0xffffffff: no hit), sh_s3, sh_t3, pad2) -- mi355_debug_surface_interaction's layout
 * ---------------------------------------------------------------------------------------- */
ORC_API void orc_ray_intersect(OScene *sc, const float *rays, uint32_t n, float *hits, float *si_out) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r; r.o = V3(rays[8*i], rays[8*i+1], rays[8*i+2]); r.d = V3(rays[8*i+3], rays[8*i+4], rays[8*i+5]); r.maxt = rays[8*i+6];
        PI pi = scene_intersect(sc, &r);
        float *h = hits + 8 * i; uint32_t occ = (uint32_t) scene_ray_test(sc, &r);
        h[0] = pi.t; h[1] = pi.prim_uv.x; h[2] = pi.prim_uv.y; memcpy(&h[3], &pi.prim_index, 4); memcpy(&h[4], &pi.shape_index, 4); memcpy(&h[5], &occ, 4); h[6] = h[7] = 0;
        if (si_out) {
            float *o = si_out + 24 * i; uint32_t none = 0xffffffffu;
            for (int k = 0; k < 24; ++k) o[k] = 0.f;
            o[14] = pi.t; memcpy(&o[15], &none, 4);
            if (pi.t == INFINITY) continue;
            SI si = compute_si(sc, &r, &pi);
            o[0] = si.p.x; o[1] = si.p.y; o[2] = si.p.z; o[3] = si.n.x; o[4] = si.n.y; o[5] = si.n.z; o[6] = si.uv.x; o[7] = si.uv.y;
            o[8] = si.sh_n.x; o[9] = si.sh_n.y; o[10] = si.sh_n.z; o[11] = si.wi.x; o[12] = si.wi.y; o[13] = si.wi.z; o[14] = si.t;
            memcpy(&o[15], &si.shape, 4);
            o[16] = si.sh_s.x; o[17] = si.sh_s.y; o[18] = si.sh_s.z; o[19] = si.sh_t.x; o[20] = si.sh_t.y; o[21] = si.sh_t.z;
        }
    }
}
/* BasicPRBIntegrator.trace -- src/python/python/ad/integrators/prb_basic.py:46-140, the fork's geometric ray tracer (design_problem.py:183-203,
   :415-417, :430-472).  Statement for statement: loop state (ray, penultimate_ray, depth, beta, active) :64-77; scene.ray_intersect :88;
   active_next &= (depth + 1 < max_depth) & si.is_valid() :99; bsdf.sample(ctx, si, 1.0, (0, 0), active_next) :106-112 (sample1 = 1.0: "the
   reflective path is never taken" unless the Fresnel reflectance is 1); penultimate_ray = ray :116; ray = si.spawn_ray(to_world(wo)) :119;
   beta *= weight :120; active_next &= any(beta != 0) :123; depth[si.is_valid()] += 1 :126; active_next &= depth < num_lenses :129.
   BSDFContext() is TransportMode::Radiance (the default), so the refracted weight carries eta_ti^2 -- only its being non-zero matters.
   out: n x 12 = penultimate o3, d3, last si.p (0 when the last ray hit nothing), si.is_valid(), depth, 0 (mi355_trace_rays). */
ORC_API void orc_trace(OScene *sc, const float *rays, uint32_t n, uint32_t num_lenses, int32_t max_depth_in, float *out) {
    const uint32_t max_depth = (uint32_t) max_depth_in;
    const int saved_mode = g_importance_mode; g_importance_mode = 0;
    g_watertight = 1;                                                       /* see pluecker_watertight */
    for (uint32_t i = 0; i < n; ++i) {
        Ray ray; ray.o = V3(rays[8*i], rays[8*i+1], rays[8*i+2]); ray.d = V3(rays[8*i+3], rays[8*i+4], rays[8*i+5]); ray.maxt = rays[8*i+6];
        Ray pen = ray; uint32_t depth = 0; v3 beta = V3(1.f, 1.f, 1.f), p_last = V3(0.f, 0.f, 0.f); int active = 1, si_valid = 0;
        while (active) {
            PI pi = scene_intersect(sc, &ray);
            int valid = pi.t != INFINITY;
            int active_next = valid && (depth + 1u < max_depth);
            pen = ray; si_valid = valid; p_last = V3(0.f, 0.f, 0.f);
            if (valid) {
                SI si = compute_si(sc, &ray, &pi);
                p_last = si.p;
                const mi355_bsdf_desc *bsdf = &sc->bsdfs[sc->shapes[si.shape].bsdf];
                v3 refl = bsdf_reflectance(sc, bsdf, &si);
                BSample bs; v3 w; v2 s2 = { 0.f, 0.f };
                bsdf_sample(bsdf, refl, si.wi, 1.0f, s2, active_next, &bs, &w);
                ray = spawn_ray(si.p, si.n, si_to_world(&si, bs.wo));
                beta = mul3(beta, w);
                active_next = active_next && (beta.x != 0.f || beta.y != 0.f || beta.z != 0.f);
                depth += 1u;
                active_next = active_next && depth < num_lenses;
            }
            active = active_next;
        }
        float *o = out + 12 * (size_t) i;
        o[0] = pen.o.x; o[1] = pen.o.y; o[2] = pen.o.z; o[3] = pen.d.x; o[4] = pen.d.y; o[5] = pen.d.z;
        o[6] = p_last.x; o[7] = p_last.y; o[8] = p_last.z; o[9] = si_valid ? 1.f : 0.f; o[10] = (float) depth; o[11] = 0.f;
    }
    g_importance_mode = saved_mode; g_watertight = 0;
}
ORC_API void orc_primary_rays(OScene *sc, const float *sample_pos, uint32_t n, float *rays) {
    for (uint32_t i = 0; i < n; ++i) {
        v2 p = { sample_pos[2*i], sample_pos[2*i+1] }; Ray r = sample_primary_ray(sc, p); float *o = rays + 8 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.maxt; o[7] = 0;
    }
}
/* RealisticLensCamera::sample_ray on caller samples: rays n x 8 (o, d, maxt, weight) */
ORC_API void orc_lens_rays(OScene *sc, const float *sample_pos, const float *aperture, uint32_t n, float *rays) {
    for (uint32_t i = 0; i < n; ++i) {
        v2 p = { sample_pos[2*i], sample_pos[2*i+1] }, a = { aperture[2*i], aperture[2*i+1] }; Ray r; float w = sample_lens_ray(sc, p, a, &r); float *o = rays + 8 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z; o[6] = r.maxt; o[7] = w;
    }
}
ORC_API float orc_shape_surface_area(const OScene *sc, uint32_t shape) { return 1.f / sc->shapes[shape].inv_area; }
ORC_API void orc_texture_eval(const OScene *sc, uint32_t tex, const float *uv, int n, float *out3) {
    for (int i = 0; i < n; ++i) { v2 q = { uv[2*i], uv[2*i+1] }; v3 r = texture_eval(&sc->textures[tex], q, NULL, NULL); out3[3*i] = r.x; out3[3*i+1] = r.y; out3[3*i+2] = r.z; }
}
ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
