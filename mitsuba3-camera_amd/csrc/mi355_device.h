// mi355_device.h -- device-side building blocks of the gfx950 path-tracing / PRB engine.
//
// Written for CDNA4 only (64-lane waves, scalar unit, LDS).  Design notes:
//   * scene records (shapes, triangles, BVH nodes) are indexed by WAVE-UNIFORM loop counters,
//     so the compiler fetches them with s_load into SGPRs and the per-lane work is pure VALU;
//   * per-lane path state lives in VGPRs for the whole path (fused mode) or in SoA streams
//     (wavefront mode, mi355_wavefront.hip);
//   * arithmetic mirrors the reference's Dr.Jit expressions: dot/cross/matrix products are
//     explicit fma chains and the translation unit is compiled with -ffp-contract=off, so
//     a lane computes the same numbers as the CPU oracle up to rcp/sqrt/sincos rounding.
//
// Reference citations are relative to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mi355.h"

#define MI_EPS        5.9604644775390625e-8f
#define MI_RAY_EPS    (MI_EPS * 1500.f)                 // include/mitsuba/core/math.h:17-23
#define MI_SHADOW_EPS (MI_RAY_EPS * 10.f)
#define MI_PI         3.14159265358979323846f
#define MI_INV_PI     0.31830988618379067154f
#define MI_LARGEST    3.402823466e+38f
#define MI_INF        __builtin_huge_valf()

#define MI_DEV static __device__ __forceinline__

// -------------------------------------------------------------------------------------------
// device scene records (host builds them in mi355_engine.hip::build_device_scene)
// -------------------------------------------------------------------------------------------
#define DSHAPE_FLIP      1u
#define DSHAPE_HAS_NRM   2u
#define DSHAPE_HAS_UV    4u
#define DSHAPE_NRM_DEP   8u          // the vertex normals were recomputed from the positions by a parameter update (Mesh::parameters_changed ->
                                     // recompute_vertex_normals, mesh.cpp:138-139): the derivative passes treat them as functions of the positions

struct DShape {                      // 176 B, read through the scalar cache
    uint32_t type, bsdf; int32_t emitter; uint32_t flags;
    float to_object[12];
    float to_world[12];
    float frame_s[3], frame_t[3], frame_n[3]; float inv_area;   // rectangle.cpp:115-126
    uint32_t tri_offset, tri_count;  // into DScene::tris / faces (mesh)
    uint32_t vtx_offset;             // into positions/normals/texcoords
    int32_t  bvh_root;               // >= 0: the mesh's triangles are in the scene BVH; -1: flattened into DScene::prims / boxes
    float bbox_min[3], bbox_max[3];
    uint32_t cdf_offset;             // mesh: into DScene::area_cdf (inclusive prefix sums of the face areas, ORIGINAL face order)
    uint32_t grad_offset;            // mesh: offset of its MI355_PARAM_VERTEX_POSITIONS block in the gradient buffer (0xffffffff: none)
};
struct DTri { float p0[3], e1[3], e2[3]; };               // mesh.h:438: e1 = p1-p0, e2 = p2-p0 precomputed (same arithmetic)
// Flat, wave-uniform primitive: a parallelogram frame P0 + s*Es + t*Et, stored as the affine map world -> (s, t, h)
// (h = signed distance along Es x Et in that basis).  One 64-byte record = one s_load_dwordx16.
//   rectangle shape : m = the shape's to_object, bounds s,t in [-1,1]               (rectangle.cpp:446-468, same formula)
//   two coplanar triangles that form a parallelogram (cube faces, OBJ quads): bounds s,t in [0,1]
//   single triangle : Es = e1, Et = e2, (s,t) are its barycentrics, bounds s,t >= 0, s+t <= 1
// Device order of m[0..7] (MI_PRIM_INTERLEAVE, default): the s-row and the t-row interleaved -- m0 m4 m1 m5 m2 m6 m3 m7 -- so that the pairs
// the packed-math instructions want (v_pk_fma_f32 on (s, t) at once) are ADJACENT, even-aligned SGPRs straight out of the s_load; in row
// order the compiler builds every pair with s_mov (4 per primitive test).  Host code writes rows; prims_for_device() permutes at upload.
#ifndef MI_PRIM_INTERLEAVE
#define MI_PRIM_INTERLEAVE 1
#endif
#if MI_PRIM_INTERLEAVE
#define MI_PM(i) ((i) < 8 ? (((i) & 3) * 2 + ((i) >> 2)) : (i))
#else
#define MI_PM(i) (i)
#endif
struct DPrim {
    float m[12];
    float lo;        // lower bound of s and t (-1 or 0)
    float smax;      // upper bound of s+t (1 = single triangle, 2 = never binding)
    uint32_t shape;
    uint32_t fin;    // index into DScene::fins (mesh) or 0xffffffff (rectangle)
};
// A mesh whose 12 triangles form a parallelepiped (the `cube` plugin under any affine to_world, OBJ boxes) is ONE record:
// the affine map world -> box-local with the box = [0,1]^3; a slab test replaces six parallelogram tests.
// Faces are numbered 2*axis + side (side 0: u_axis = 0, side 1: u_axis = 1); in-face coordinates are the two other
// local coordinates in increasing axis order, mapped to (prim_index, b1, b2) by fins[fin + face].
struct DBox {
    float m[12];
    uint32_t shape, fin;
    uint32_t pad[2];
};
struct DPrimFin {    // (b1, b2) = A (s, t) + c for each of the two triangles of a mesh primitive
    uint32_t tri_a, tri_b;      // ORIGINAL face ids (the reference's prim_index)
    float a[6], b[6];
};
// ONE 4-wide BVH over the triangles of every mesh that is not flattened (128-byte nodes = eight 16-byte loads per lane and step,
// half the dependent steps of a binary tree).  Child boxes are SoA ([axis][child]) so that the four slab tests of a lane share their
// ray operands.  `entry`: 0xffffffff = empty slot; bit 31 clear = index of an inner node; bit 31 set = leaf, bits 29-30 = triangle
// count - 1, bits 0-28 = first triangle (a slot of DScene::tris, the leaf's triangles are contiguous).
struct DBvh4 { float lo[3][4]; float hi[3][4]; uint32_t entry[4]; uint32_t pad[4]; };
#define MI_BVH_EMPTY 0xffffffffu
#define MI_BVH_LEAF 0x80000000u
// What the traversal READS: the same node in 56 bytes (stride 64).  The light tracer's kernels on large meshes are bound by the data-return
// path of the vector memory pipeline (TD busy 88 % of the launch on the lens doublet, profiles/r03_pmc_lens_mem.txt), and 8 of every
// 10 cycles of it went into the 128-byte nodes.  Child boxes are quantised to 8 bits per plane against the box of all children:
// plane = org[axis] + q * 2^e[axis] with a power-of-two step, so decoding is exact (one fma), and conservatively -- low planes rounded
// down, high planes up.  DBvh4 stays as the refit's working copy (full precision bottom-up), DBvhQ is re-encoded from it node by node.
//   dwordx4 #0: org.x org.y org.z exps (three biased exponents, one per byte)
//   dwordx4 #1: qlo.x qlo.y qlo.z qhi.x        (one byte per child)
//   dwordx4 #2: qhi.y qhi.z entry0 entry1
//   dwordx2 #3: entry2 entry3
struct DBvhQ { float org[3]; uint32_t exps; uint32_t qlo[3]; uint32_t qhi[3]; uint32_t entry[4]; uint32_t pad[2]; };
static_assert(sizeof(DBvhQ) == 64, "DBvhQ is one 64-byte line");
static inline __host__ __device__ void bvhq_encode(const DBvh4 &n, DBvhQ &q) {
    for (int a = 0; a < 3; ++a) {
        float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
        for (int c = 0; c < 4; ++c) if (n.entry[c] != MI_BVH_EMPTY) { lo = n.lo[a][c] < lo ? n.lo[a][c] : lo; hi = n.hi[a][c] > hi ? n.hi[a][c] : hi; }
        if (!(hi >= lo)) { lo = 0.f; hi = 0.f; }                            // no child (never traversed) or non-finite boxes (never hit)
        // step 2^e with 254 steps covering the extent (one step of head room for the outward rounding below)
        int e = -126;
        const float ext = hi - lo;
        if (ext > 0.f && ext < __builtin_huge_valf()) { int ex; (void) __builtin_frexpf(ext / 254.f, &ex); e = ex; }   // ext / 254 = m 2^ex, m in [0.5, 1): 2^ex >= ext / 254
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        const float step = __builtin_ldexpf(1.f, e), inv = __builtin_ldexpf(1.f, -e);
        uint32_t wl = 0, wh = 0;
        for (int c = 0; c < 4; ++c) {
            uint32_t ql = 0, qh = 0;
            if (n.entry[c] != MI_BVH_EMPTY) {
                float fl = __builtin_floorf((n.lo[a][c] - lo) * inv), fh = __builtin_ceilf((n.hi[a][c] - lo) * inv);
                fl = fl < 0.f ? 0.f : (fl > 255.f ? 255.f : fl); fh = fh < 0.f ? 0.f : (fh > 255.f ? 255.f : fh);
                ql = (uint32_t) fl; qh = (uint32_t) fh;
                while (ql > 0u && __builtin_fmaf((float) ql, step, lo) > n.lo[a][c]) --ql;          // the DECODED plane must not lie inside the child's box
                while (qh < 255u && __builtin_fmaf((float) qh, step, lo) < n.hi[a][c]) ++qh;
            }
            wl |= ql << (8 * c); wh |= qh << (8 * c);
        }
        q.org[a] = lo; q.qlo[a] = wl; q.qhi[a] = wh;
        if (a == 0) q.exps = 0u;
        q.exps |= (uint32_t) (e + 127) << (8 * a);
    }
    for (int c = 0; c < 4; ++c) q.entry[c] = n.entry[c];
    q.pad[0] = q.pad[1] = 0u;
}
struct DBsdf {                       // 96 B
    uint32_t type; float reflectance[3]; int32_t texture; int32_t grad_slot;
    float alpha_u, alpha_v; uint32_t distribution, sample_visible;      // roughconductor.cpp:163-212
    float eta[3], k[3];
    float rel_eta;                                                      // dielectric: int_ior / ext_ior (dielectric.cpp:222)
    float transmittance[3];
    float ext_ior;                                                      // dielectric: the denominator of rel_eta (IOR gradients / updates)
    int32_t ior_grad_offset;                                            // offset of the (int_ior, ext_ior) pair in the gradient buffer, or -1
    uint32_t pad[2];
};
struct DSphere { float center[3]; float radius; uint32_t shape; uint32_t pad[3]; };   // sphere.cpp:158-159
struct DEmitter { uint32_t type; float radiance[3]; uint32_t shape; int32_t grad_slot; uint32_t pad[2]; };
struct DTexture { uint32_t w, h, c, filter, wrap; int32_t grad_slot; uint64_t data_offset; uint64_t grad_offset; float to_uv[6]; };

struct DScene {
    uint32_t n_shapes, n_bsdfs, n_emitters, n_textures;
    const DShape   *shapes;
    const DTri     *tris;         // traversal order (BVH leaves are contiguous)
    const uint32_t *tri_face;     // traversal slot -> original face id (the reference's prim_index)
    const uint32_t *faces;        // 3 per triangle in ORIGINAL face order, mesh-local vertex ids
    const float    *positions;    // 3 per vertex (all meshes concatenated)
    const float    *normals;      // 3 per vertex (zeros when the mesh has none)
    const float    *texcoords;    // 2 per vertex
    const float    *area_cdf;     // meshes: Mesh::build_pmf table (mesh.cpp:430-470), one float per face
    const DBvhQ *bvh;             // node 0 = root (n_bvh_shapes != 0); the quantised copy of the 4-wide tree
    const uint32_t *tri_shape;    // traversal slot -> shape index
    const DPrim    *prims;        // rectangles + every mesh of <= MI_FLAT_MAX triangles
    const DPrimFin *fins;
    const DBox     *boxes;        // parallelepiped meshes
    const DSphere  *spheres;
    uint32_t n_spheres, features; // features: bit 0 = some BSDF is not `diffuse`, bit 1 = analytic spheres present, bit 2 = realisticlens sensor,
                                  // bit 3 = meshes traversed through the BVH; any bit set -> kernels instantiated with FULL = true
    const uint32_t *bvh_shapes;   // shapes traversed through their BVH
    uint32_t n_prims, n_bvh_shapes, n_boxes;
    const DBsdf    *bsdfs;
    const DEmitter *emitters;
    const DTexture *textures;
    const float    *texdata;      // all texture tensors concatenated
    // sensor + film
    float sample_to_camera[16];
    float cam_to_world[12];
    float near_clip, far_clip;
    uint32_t crop_w, crop_h, crop_x, crop_y;
    uint32_t grid_w, border;      // lane grid of the sampling integrators: crop window + 2 * border wide (Film::sample_border).  Kept next to
                                  // the crop fields on purpose: kernel arguments are fetched with wide scalar loads, and with these three words
                                  // further down the forward kernel carried 120 more SGPR spill moves (10.04 ms against 9.80 ms)
    uint32_t rfilter; float filter_radius; float filter_coeff[10];
    float inv_emitter_count;
    // realisticlens sensor: interface table (8 floats per interface, film -> world), physical half extents of the film
    uint32_t sensor_type, lens_count; const float *lens; float film_px, film_py;
    uint32_t film_w, film_h;
    float camera_to_sample[16];   // perspective sensor, particle tracer: sample_direction (perspective.cpp:282-340)
    float image_rect[4], normalization;   // lo.x, lo.y, hi.x, hi.y of the image plane at z = 1 and 1 / its area (perspective.cpp:190-199)
    float shutter_open_time;      // > 0: one extra sampler draw per sample for the time (integrator.cpp:442-444)
    uint32_t grid_h;
    uint32_t sample_exit_pupil; float film_diagonal; const float *pupil;   // exit-pupil bounds, 64 x (min.x, min.y, max.x, max.y)
};

// Several perspective sensors served by ONE light-traced pass (mi355_render_accumulate_sensors): the light paths of a launch depend on the
// seed and on the film SIZE only, so the fields of a lens-design scene -- same seed, same film size, one camera each -- share them; only the
// sensor connection at every vertex is made once per camera.  The films lie side by side in one virtual film (x_offset, DSensorSet::virt_w is
// the kernel's crop_w), so the deferred splat sorts and accumulates all of them in one go.
#define MI_MULTI_MAX MI355_MAX_SENSORS_PER_PASS                              // (8: the notebooks run 3, 5 and 6 field sources)
struct DSensorView { float cam_to_world[12]; float camera_to_sample[16]; float image_rect[4]; float normalization, near_clip, far_clip, x_offset; };
struct DSensorSet { uint32_t n; float film_w, film_h; uint32_t pad; DSensorView v[MI_MULTI_MAX]; };

// Record tables in LDS (the north star's "BSDF params staged in LDS").  The BASIC kernel variants (FULL == 0: all-diffuse scenes of flat
// primitives -- the headline configurations) stage the shape / bsdf / emitter / primitive-finish records once per workgroup (lt_fill,
// 4.6 KB) and index them per lane with ds_read instead of per-lane global gathers: forward kernel 8.92 -> 8.71 ms on the Cornell box
// (profiles/r03_occupancy_ab.txt).  A scene whose tables exceed the staging area runs the general kernels (features bit 4).
// -DMI_LDS_TABLES=0 restores the global gathers everywhere (A/B).
#ifndef MI_LDS_TABLES
#define MI_LDS_TABLES 1
#endif
#define MI_LT_SHAPES 16
#define MI_LT_BSDFS 8
#define MI_LT_EMITTERS 4
#define MI_LT_FINS 16
#if MI_LDS_TABLES
__shared__ __attribute__((aligned(16))) DShape g_lt_shapes[MI_LT_SHAPES];
__shared__ __attribute__((aligned(16))) DBsdf g_lt_bsdfs[MI_LT_BSDFS];
__shared__ __attribute__((aligned(16))) DEmitter g_lt_emitters[MI_LT_EMITTERS];
__shared__ __attribute__((aligned(16))) DPrimFin g_lt_fins[MI_LT_FINS];
#define MI_SC_SHAPE(sc, i) (*((FULL) == 0 ? &g_lt_shapes[i] : &(sc).shapes[i]))
#define MI_SC_BSDF(sc, i) (*((FULL) == 0 ? &g_lt_bsdfs[i] : &(sc).bsdfs[i]))
#define MI_SC_EMITTER(sc, i) (*((FULL) == 0 ? &g_lt_emitters[i] : &(sc).emitters[i]))
#define MI_SC_FIN(sc, i) (*((FULL) == 0 ? &g_lt_fins[i] : &(sc).fins[i]))
// all threads of the workgroup, at the top of every kernel that has a FULL == 0 instantiation; ends with a barrier.  The primitive-finish
// table is copied whole: the host allocates at least MI_LT_FINS entries.
static __device__ __forceinline__ void lt_fill(const DScene &sc) {
    const uint32_t *src[4] = { (const uint32_t *) sc.shapes, (const uint32_t *) sc.bsdfs, (const uint32_t *) sc.emitters, (const uint32_t *) sc.fins };
    uint32_t *dst[4] = { (uint32_t *) g_lt_shapes, (uint32_t *) g_lt_bsdfs, (uint32_t *) g_lt_emitters, (uint32_t *) g_lt_fins };
    const uint32_t words[4] = { min(sc.n_shapes, (uint32_t) MI_LT_SHAPES) * (uint32_t) (sizeof(DShape) / 4), min(sc.n_bsdfs, (uint32_t) MI_LT_BSDFS) * (uint32_t) (sizeof(DBsdf) / 4),
                                min(sc.n_emitters, (uint32_t) MI_LT_EMITTERS) * (uint32_t) (sizeof(DEmitter) / 4), (uint32_t) MI_LT_FINS * (uint32_t) (sizeof(DPrimFin) / 4) };
    for (int t = 0; t < 4; ++t) for (uint32_t k = threadIdx.x; k < words[t]; k += blockDim.x) dst[t][k] = src[t][k];
    __syncthreads();
}
#define MI_LT_FILL(FULL, sc) do { if ((FULL) == 0) lt_fill(sc); } while (0)
#define MI_LT_BYTES ((uint32_t) (MI_LT_SHAPES * sizeof(DShape) + MI_LT_BSDFS * sizeof(DBsdf) + MI_LT_EMITTERS * sizeof(DEmitter) + MI_LT_FINS * sizeof(DPrimFin)))
#else
#define MI_SC_SHAPE(sc, i) ((sc).shapes[i])
#define MI_SC_BSDF(sc, i) ((sc).bsdfs[i])
#define MI_SC_EMITTER(sc, i) ((sc).emitters[i])
#define MI_SC_FIN(sc, i) ((sc).fins[i])
#define MI_LT_FILL(FULL, sc) do { } while (0)
#define MI_LT_BYTES 0u
#endif

// -------------------------------------------------------------------------------------------
// small vector math with Dr.Jit's operation order
// -------------------------------------------------------------------------------------------
struct v3 { float x, y, z; };
struct v2 { float x, y; };
MI_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
MI_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
MI_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
MI_DEV v3 operator*(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
MI_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
MI_DEV v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
MI_DEV v3 fma3s(v3 a, float s, v3 c) { return V3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }
MI_DEV v3 fma3(v3 a, v3 b, v3 c) { return V3(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)); }
MI_DEV float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
MI_DEV v3 cross3(v3 a, v3 b) {
    return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// dr::rcp / dr::rsqrt / dr::sqrt are approximate-plus-Newton in the reference's LLVM back-end too; here they map to
// the hardware transcendentals (v_rcp_f32 / v_rsq_f32 / v_sqrt_f32, 1 ulp) instead of the ~10-instruction IEEE expansions.
#ifndef MI_EXACT_MATH
MI_DEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
MI_DEV float rsqrt_(float x) { return __builtin_amdgcn_rsqf(x); }
MI_DEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
MI_DEV float div_(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
MI_DEV void sincos_(float x, float &sn, float &cs) { sn = __sinf(x); cs = __cosf(x); }
#else
MI_DEV float rcp_(float x) { return 1.0f / x; }
MI_DEV float rsqrt_(float x) { return 1.0f / sqrtf(x); }
MI_DEV float sqrt_(float x) { return sqrtf(x); }
MI_DEV float div_(float a, float b) { return a / b; }
MI_DEV void sincos_(float x, float &sn, float &cs) { sincosf(x, &sn, &cs); }
#endif
MI_DEV float sqnorm3(v3 a) { return dot3(a, a); }
MI_DEV float norm3(v3 a) { return sqrt_(dot3(a, a)); }
MI_DEV v3 normalize3(v3 a) { return a * rsqrt_(sqnorm3(a)); }
MI_DEV float max3(v3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }
MI_DEV float mulsign(float a, float b) {
    return __uint_as_float(__float_as_uint(a) ^ (__float_as_uint(b) & 0x80000000u));
}
MI_DEV v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }

// include/mitsuba/core/vector.h:116-136
MI_DEV void coordinate_system(v3 n, v3 &s, v3 &t) {
    float sign = copysignf(1.0f, n.z), a = -rcp_(sign + n.z), b = n.x * n.y * a;
    s = V3(mulsign((n.x * n.x) * a, n.z) + 1.0f, mulsign(b, n.z), mulsign(-n.x, n.z));
    t = V3(b, fmaf(n.y, n.y * a, sign), -n.y);
}
// include/mitsuba/core/transform.h:98-141 (row-major 3x4)
MI_DEV v3 xf_point(const float *m, v3 p) {
    v3 r = V3(m[3], m[7], m[11]);
    r = fma3s(V3(m[0], m[4], m[8]), p.x, r);
    r = fma3s(V3(m[1], m[5], m[9]), p.y, r);
    r = fma3s(V3(m[2], m[6], m[10]), p.z, r);
    return r;
}
MI_DEV v3 xf_vector(const float *m, v3 v) {
    v3 r = V3(m[0], m[4], m[8]) * v.x;
    r = fma3s(V3(m[1], m[5], m[9]), v.y, r);
    r = fma3s(V3(m[2], m[6], m[10]), v.z, r);
    return r;
}

// -------------------------------------------------------------------------------------------
// RNG: TEA-seeded PCG32 per lane (random.h:76-91, sampler.cpp:116-135, independent.cpp:88-97)
// -------------------------------------------------------------------------------------------
struct Pcg32 { uint64_t state, inc; };
MI_DEV void tea32(uint32_t &v0, uint32_t &v1) {
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
}
MI_DEV uint32_t pcg32_next(Pcg32 &r) {
    uint64_t old = r.state;
    r.state = old * 0x5851f42d4c957f2dULL + r.inc;
    uint32_t xorshift = (uint32_t) (((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t) (old >> 59u);
    return (xorshift >> rot) | (xorshift << ((~rot + 1u) & 31u));
}
MI_DEV void sampler_seed(Pcg32 &r, uint32_t seed, uint32_t lane) {
    uint32_t v0 = seed, v1 = lane; tea32(v0, v1);
    r.state = 0; r.inc = ((uint64_t) v1 << 1u) | 1u;
    pcg32_next(r); r.state += (uint64_t) v0; pcg32_next(r);
}
MI_DEV float next_1d(Pcg32 &r) { return __uint_as_float((pcg32_next(r) >> 9) | 0x3f800000u) - 1.0f; }

// -------------------------------------------------------------------------------------------
// warps and filters
// -------------------------------------------------------------------------------------------
// include/mitsuba/core/warp.h:54-90, 412-420
MI_DEV v3 square_to_cosine_hemisphere(float sx, float sy) {
    float x = fmaf(2.f, sx, -1.f), y = fmaf(2.f, sy, -1.f);
    bool is_zero = (x == 0.f && y == 0.f), q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = div_(0.25f * MI_PI * rp, r);
    if (q13) phi = 0.5f * MI_PI - phi;
    if (is_zero) phi = 0.f;
    float sn, cs; sincos_(phi, sn, cs);
    float px = r * cs, py = r * sn;
    float sq = fmaf(py, py, px * px);
    return V3(px, py, sqrt_(fmaxf(1.f - sq, 0.f)));
}
// Dr.Jit estrin_impl for 10 coefficients; src/rfilters/gaussian.cpp:88-98
MI_DEV float estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fmaf(x, c[1], c[0]), a1 = fmaf(x, c[3], c[2]), a2 = fmaf(x, c[5], c[4]),
          a3 = fmaf(x, c[7], c[6]), a4 = fmaf(x, c[9], c[8]);
    float b0 = fmaf(x2, a1, a0), b1 = fmaf(x2, a3, a2);
    float c0 = fmaf(x4, b1, b0);
    return fmaf(x8, a4, c0);
}
MI_DEV float filter_eval(const DScene &sc, float x) {
    if (sc.rfilter == MI355_RFILTER_GAUSSIAN) return fmaxf(estrin10(x * x, sc.filter_coeff), 0.f);
    return fmaxf(0.f, 1.f - fabsf(x * (1.f / sc.filter_radius)));       // tent
}
MI_DEV float mis_weight(float a, float b) {                              // path.cpp:300-305
    a *= a; b *= b; float w = div_(a, a + b); return isfinite(w) ? w : 0.f;
}

// -------------------------------------------------------------------------------------------
// rays, hits
// -------------------------------------------------------------------------------------------
struct Ray { v3 o, d; float maxt; };
struct Hit { float t; float u, v; uint32_t prim, shape; };              // PreliminaryIntersection, interaction.h:570-675
struct SI {                                                             // SurfaceInteraction, interaction.h:171-492
    float t; v3 p, n; v2 uv; v3 sh_s, sh_t, sh_n; v3 wi; uint32_t prim; int32_t shape;
};
MI_DEV v3 ray_at(const Ray &r, float t) { return fma3s(r.d, t, r.o); }

// Mesh::moeller_trumbore, mesh.h:430-453
MI_DEV bool tri_intersect(const DTri &tr, const Ray &ray, float &t, float &u, float &v) {
    v3 p0 = ld3(tr.p0), e1 = ld3(tr.e1), e2 = ld3(tr.e2);
    v3 pvec = cross3(ray.d, e2);
    float inv_det = rcp_(dot3(e1, pvec));
    v3 tvec = ray.o - p0;
    u = dot3(tvec, pvec) * inv_det;
    v3 qvec = cross3(tvec, e1);
    v = dot3(ray.d, qvec) * inv_det;
    t = dot3(e2, qvec) * inv_det;
    return u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t >= 0.f && t <= ray.maxt;
}
// Watertight ray / triangle test, used by the geometric tracer (k_trace) only.  Its callers aim rays EXACTLY at mesh vertices and along
// mesh edges: a field's central ray goes to the entrance-pupil position, which is the pole vertex of the front lens mesh
// (notebooks/scripts/lens_design/design_problem.py:152-153, :183-189), and the tracer check sends its rays in the symmetry plane y = 0 of
// the icosphere (:442-453).  The reference intersects meshes through Embree / OptiX, which do not leak at shared edges; Moeller-Trumbore
// (above, the formula of mesh.h:430-453 that the rendering kernels and the oracle's renderer use) does -- u and v are relative to each
// triangle's own first vertex, so a ray on a shared edge can be marginally outside BOTH neighbours.  Pluecker edge functions (Embree's
// PlueckerIntersector): f(P, Q) = dot(cross(P - Q, P + Q), d) with P, Q relative to the ray origin; the neighbour sees the edge as (Q, P)
// and computes exactly -f (negation and commutation are exact, rounding is sign-symmetric), so a ray is inside at least one of the two.
// The exact vertices are gathered from the mesh buffers (DTri holds rounded edge vectors); `slot` = traversal slot of the triangle.
MI_DEV bool tri_intersect_watertight(const DScene &sc, uint32_t slot, const Ray &ray, float &t, float &u, float &v) {
    const DShape &sh = sc.shapes[sc.tri_shape[slot]];
    const uint32_t *fi = sc.faces + 3 * (size_t) (sh.tri_offset + sc.tri_face[slot]);
    v3 v0 = ld3(sc.positions + 3 * (size_t) (sh.vtx_offset + fi[0])) - ray.o, v1 = ld3(sc.positions + 3 * (size_t) (sh.vtx_offset + fi[1])) - ray.o,
       v2 = ld3(sc.positions + 3 * (size_t) (sh.vtx_offset + fi[2])) - ray.o;
    v3 e0 = v2 - v0, e1 = v0 - v1, e2 = v1 - v2;
    float U = dot3(cross3(e0, v2 + v0), ray.d), V = dot3(cross3(e1, v0 + v1), ray.d), W = dot3(cross3(e2, v1 + v2), ray.d);
    float UVW = U + V + W, eps = MI_EPS * fabsf(UVW);
    float mn = fminf(fminf(U, V), W), mx = fmaxf(fmaxf(U, V), W);
    bool ok = (mn >= -eps || mx <= eps) && UVW != 0.f;
    v3 ng = cross3(v1 - v0, v2 - v0);
    float den = dot3(ng, ray.d);
    t = dot3(v0, ng) / den;
    float inv = 1.f / UVW;
    u = fminf(fmaxf(U * inv, 0.f), 1.f); v = fminf(fmaxf(V * inv, 0.f), 1.f);
    return ok && den != 0.f && t >= 0.f && t <= ray.maxt;
}
template <bool WT> MI_DEV bool tri_hit(const DScene &sc, uint32_t slot, const Ray &ray, float &t, float &u, float &v) {
    if (WT) return tri_intersect_watertight(sc, slot, ray, t, u, v);
    return tri_intersect(sc.tris[slot], ray, t, u, v);
}
// conservative slab test against a (wave-uniform) box; used only to SKIP work, never to decide a hit
MI_DEV bool box_overlap(const float *lo, const float *hi, const Ray &ray, v3 inv_d) {
    // entering / leaving planes by the sign of inv_d; a NaN plane distance (ray inside a face plane, zero direction component) sets no bound
    const bool nx = __float_as_uint(inv_d.x) >> 31, ny = __float_as_uint(inv_d.y) >> 31, nz = __float_as_uint(inv_d.z) >> 31;
    float t0x = ((nx ? hi[0] : lo[0]) - ray.o.x) * inv_d.x, t1x = ((nx ? lo[0] : hi[0]) - ray.o.x) * inv_d.x;
    float t0y = ((ny ? hi[1] : lo[1]) - ray.o.y) * inv_d.y, t1y = ((ny ? lo[1] : hi[1]) - ray.o.y) * inv_d.y;
    float t0z = ((nz ? hi[2] : lo[2]) - ray.o.z) * inv_d.z, t1z = ((nz ? lo[2] : hi[2]) - ray.o.z) * inv_d.z;
    float tn = fmaxf(fmaxf(0.f, t0x), fmaxf(t0y, t0z));
    float tf = fminf(fminf(ray.maxt, t1x), fminf(t1y, t1z));
    return tn <= tf * 1.00001f + 1e-6f;
}

#ifndef MI_BVH_STACK
#define MI_BVH_STACK 32                     // (overridable for tests of the deep-stack fallback: -DMI_BVH_STACK=12 -DMI_BVH_LEVELS=5)
#endif
#ifndef MI_PRIM_UNROLL
#define MI_PRIM_UNROLL 1
#endif
#define MI_STR2(x) #x
#define MI_STR(x) MI_STR2(x)
#define MI_UNROLL_PRIMS _Pragma(MI_STR(unroll MI_PRIM_UNROLL))
#define MI_FLAT_MAX 64      // meshes up to this many triangles are expanded into the flat primitive list
// Per-lane traversal of the scene BVH (DBvh4).  The stack lives in LDS, laid out [level][lane] so that the 64 lanes of a wave touch
// 64 consecutive banks (conflict-free ds_write_b32 / ds_read_b32); `stk` already points at this lane's column.  Its depth bounds the
// tree: a node pushes at most three siblings, the host builds trees of at most MI_BVH_LEVELS levels (build_scene_bvh).
// Closest hit: the hit children of a node are visited near to far (sorted by entry distance) -- leaves at once, so that a hit shrinks
// ray.maxt before anything farther is looked at; inner children go on the stack far first.
#define MI_STK(i) stk[(i) * 64]
#ifndef MI_BVH_LEVELS
#define MI_BVH_LEVELS 11    // 3 * (MI_BVH_LEVELS - 1) <= MI_BVH_STACK
#endif
static_assert(3 * (MI_BVH_LEVELS - 1) <= MI_BVH_STACK, "the traversal stack must hold three siblings per level");
// MI_BVH_STATS (diagnostic builds only, tools/bvh_stats.py): traversal steps per ray -- [0] rays that entered the BVH, [1] nodes fetched, [2] triangles tested
#ifdef MI_BVH_STATS
static __device__ unsigned long long g_bvh_stats[4];
#define BVH_STAT(k, n) atomicAdd(&g_bvh_stats[k], (unsigned long long) (n))
#else
#define BVH_STAT(k, n)
#endif
struct Bvh4Hits { float tn[4]; uint32_t e[4]; };
// slab test of the four children of `node`; a child that is missed (or empty) gets tn = +inf.  Conservative like box_overlap.
// The entering plane of an axis is the box's low face when the ray runs up that axis and its high face otherwise, picked by the SIGN of
// inv_d -- not by min / max of the two plane distances.  Two reasons: (i) a ray that lies exactly IN a face plane with a zero direction
// component gives 0 * inf = NaN for that plane; v_max_f32 / v_min_f32 return the other operand (IEEE minNum / maxNum), so the NaN plane
// simply sets no bound -- the ray is inside the closed slab for every t, which is the right answer -- whereas min(NaN, +inf) = +inf made
// such a ray MISS the box.  The fork's own tracer check sends all its rays in the plane y = 0 (design_problem.py:442-453), a symmetry
// plane of the icosphere lens meshes: edges, and therefore box faces, lie exactly in it.  (ii) it is two VALU instructions per child and
// axis less.
MI_DEV Bvh4Hits bvh4_node(const DScene &sc, uint32_t node, const Ray &ray, v3 inv_d, float tmax) {
    const uint4 *n4 = reinterpret_cast<const uint4 *>(sc.bvh + node);
    const uint4 w0 = n4[0], w1 = n4[1], w2 = n4[2]; const uint2 w3 = *reinterpret_cast<const uint2 *>(n4 + 3);
    Bvh4Hits h;
    float tn[4] = { 0.f, 0.f, 0.f, 0.f }, tf[4] = { tmax, tmax, tmax, tmax };
#define MI_BVHQ_AXIS(ORG, EXPSHIFT, QLO, QHI, O, INV) { \
        const bool neg = __float_as_uint(INV) >> 31; const uint32_t nr = neg ? (QHI) : (QLO), fr = neg ? (QLO) : (QHI); \
        const float step = __uint_as_float(((w0.w >> (EXPSHIFT)) & 255u) << 23), org = __uint_as_float(ORG); \
        tn[0] = fmaxf(tn[0], (fmaf((float) (nr & 255u), step, org) - (O)) * (INV)); tf[0] = fminf(tf[0], (fmaf((float) (fr & 255u), step, org) - (O)) * (INV)); \
        tn[1] = fmaxf(tn[1], (fmaf((float) ((nr >> 8) & 255u), step, org) - (O)) * (INV)); tf[1] = fminf(tf[1], (fmaf((float) ((fr >> 8) & 255u), step, org) - (O)) * (INV)); \
        tn[2] = fmaxf(tn[2], (fmaf((float) ((nr >> 16) & 255u), step, org) - (O)) * (INV)); tf[2] = fminf(tf[2], (fmaf((float) ((fr >> 16) & 255u), step, org) - (O)) * (INV)); \
        tn[3] = fmaxf(tn[3], (fmaf((float) (nr >> 24), step, org) - (O)) * (INV)); tf[3] = fminf(tf[3], (fmaf((float) (fr >> 24), step, org) - (O)) * (INV)); }
    MI_BVHQ_AXIS(w0.x, 0, w1.x, w1.w, ray.o.x, inv_d.x)
    MI_BVHQ_AXIS(w0.y, 8, w1.y, w2.x, ray.o.y, inv_d.y)
    MI_BVHQ_AXIS(w0.z, 16, w1.z, w2.y, ray.o.z, inv_d.z)
#undef MI_BVHQ_AXIS
    h.e[0] = w2.z; h.e[1] = w2.w; h.e[2] = w3.x; h.e[3] = w3.y;
#pragma unroll
    for (int c = 0; c < 4; ++c) h.tn[c] = (h.e[c] != MI_BVH_EMPTY && tn[c] <= tf[c] * 1.00001f + 1e-6f) ? tn[c] : MI_INF;
    return h;
}
MI_DEV void bvh4_sort(Bvh4Hits &h) {                                       // near to far: the 5-exchange network for four keys
#define MI_CSWAP(a, b) { const bool sw = h.tn[b] < h.tn[a]; const float ta = sw ? h.tn[b] : h.tn[a], tb = sw ? h.tn[a] : h.tn[b]; \
                         const uint32_t ea = sw ? h.e[b] : h.e[a], eb = sw ? h.e[a] : h.e[b]; h.tn[a] = ta; h.tn[b] = tb; h.e[a] = ea; h.e[b] = eb; }
    MI_CSWAP(0, 1) MI_CSWAP(2, 3) MI_CSWAP(0, 2) MI_CSWAP(1, 3) MI_CSWAP(1, 2)
#undef MI_CSWAP
}
// PHASED traversal (MI_BVH_PHASED, default): the straightforward loop below tests the triangles of a node's leaf children inside the
// node step, so one lane that has reached the bottom of the tree holds its wave for up to 16 triangle tests per trip while the other
// lanes still walk inner nodes -- on incoherent rays (the light tracer's samples are independent per lane) 27 % of the VALU lanes were
// busy (profiles/r02_pmc_lens_summary.txt).  Here the two kinds of work are separated in TIME, in the manner of Aila & Laine's
// "while-while" loop: phase A walks inner nodes only and POSTPONES every leaf child it accepts into a per-lane FIFO (the top end of the
// lane's LDS stack; the inner-node stack grows from the bottom), and it keeps going -- speculatively, for lanes that already hold a
// leaf -- until every lane that can still walk holds at least one; phase B then tests the postponed triangles, ONE per lane and trip,
// until all FIFOs are empty; then A again.  Closest-hit semantics are unchanged (every accepted leaf is tested, the nearest t wins; the
// test ORDER differs, so an exact tie between two triangles may resolve to the other one).
#ifndef MI_BVH_PHASED
#define MI_BVH_PHASED 1
#endif
#define MI_QSTK(i) stk[(MI_BVH_STACK - 1 - (i)) * 64]
#ifndef MI_BVH_QCULL
#define MI_BVH_QCULL 1                      // closest hit: keep a postponed leaf's entering distance and drop the leaf when a nearer hit was found meanwhile
#endif
#if MI_BVH_PHASED
template <bool WT = false>
MI_DEV void bvh_closest(const DScene &sc, Ray &ray, v3 inv_d, Hit &hit, bool active, uint32_t *stk) {
    int sp = 0, qh = 0, qt = 0;                                            // inner stack [0, sp) from the bottom; leaf FIFO [qh, qt) from the top
    uint32_t node = active ? 0u : MI_BVH_EMPTY, tri = 0u, left = 0u;
    if (active) BVH_STAT(0, 1);
    for (;;) {
        for (;;) {                                                         // ---- phase A: inner nodes
            const bool can = node != MI_BVH_EMPTY, empty = qh == qt;
            if (!__any(can && empty)) break;                               // every lane that can still walk holds a leaf (or none can walk)
#if MI_BVH_QCULL
            const bool room = sp + 2 * qt + 11 <= MI_BVH_STACK;            // 3 pushes + 4 leaves of two words (entry, entering distance)
#else
            const bool room = sp + qt + 7 <= MI_BVH_STACK;                 // 3 pushes + 4 leaves
#endif
            if (can && (room || empty)) {
                Bvh4Hits h = bvh4_node(sc, node, ray, inv_d, ray.maxt);
                BVH_STAT(1, 1);
                bvh4_sort(h);
                uint32_t next = MI_BVH_EMPTY;
#pragma unroll
                for (int c = 3; c >= 0; --c)
                    if (h.tn[c] != MI_INF && !(h.e[c] & MI_BVH_LEAF)) { if (next != MI_BVH_EMPTY && sp + (MI_BVH_QCULL ? 2 : 1) * qt < MI_BVH_STACK) { MI_STK(sp) = next; sp++; } next = h.e[c]; }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (h.tn[c] <= ray.maxt && (h.e[c] & MI_BVH_LEAF)) {
#if MI_BVH_QCULL
                        if (room) { MI_QSTK(2 * qt) = h.e[c]; MI_QSTK(2 * qt + 1) = __float_as_uint(h.tn[c]); qt++; }
#else
                        if (room) { MI_QSTK(qt) = h.e[c]; qt++; }
#endif
                        else {                                             // (a stack this deep is all but unreachable: test the leaf on the spot)
                            const uint32_t first = h.e[c] & 0x1fffffffu, count = ((h.e[c] >> 29) & 3u) + 1u;
                            BVH_STAT(2, count);
                            for (uint32_t k = 0; k < count; ++k) {
                                float t, u, v;
                                if (tri_hit<WT>(sc, first + k, ray, t, u, v) && t < hit.t) {
                                    hit.t = t; hit.u = u; hit.v = v; hit.prim = sc.tri_face[first + k]; hit.shape = sc.tri_shape[first + k]; ray.maxt = t;
                                }
                            }
                        }
                    }
                if (next != MI_BVH_EMPTY) node = next;
                else if (sp > 0) { --sp; node = MI_STK(sp); }
                else node = MI_BVH_EMPTY;
            }
        }
        while (__any(left != 0u || qh != qt)) {                            // ---- phase B: one postponed triangle per lane and trip
            if (left == 0u && qh != qt) {
#if MI_BVH_QCULL
                const uint32_t e = MI_QSTK(2 * qh); const float tn = __uint_as_float(MI_QSTK(2 * qh + 1)); qh++;
                tri = e & 0x1fffffffu; left = tn <= ray.maxt ? ((e >> 29) & 3u) + 1u : 0u;      // a nearer hit was found since the leaf was postponed
#else
                const uint32_t e = MI_QSTK(qh); qh++;
                tri = e & 0x1fffffffu; left = ((e >> 29) & 3u) + 1u;
#endif
                if (qh == qt) qh = qt = 0;
            }
            if (left != 0u) {
                float t, u, v;
                BVH_STAT(2, 1);
                if (tri_hit<WT>(sc, tri, ray, t, u, v) && t < hit.t) {
                    hit.t = t; hit.u = u; hit.v = v; hit.prim = sc.tri_face[tri]; hit.shape = sc.tri_shape[tri]; ray.maxt = t;
                }
                tri++; left--;
            }
        }
        if (!__any(node != MI_BVH_EMPTY)) break;
    }
}
MI_DEV bool bvh_any(const DScene &sc, const Ray &ray, v3 inv_d, bool active, uint32_t *stk) {
    bool occluded = false;
    int sp = 0, qh = 0, qt = 0;
    uint32_t node = active ? 0u : MI_BVH_EMPTY, tri = 0u, left = 0u;
    if (active) BVH_STAT(0, 1);
    for (;;) {
        for (;;) {
            const bool can = node != MI_BVH_EMPTY, empty = qh == qt;
            if (!__any(can && empty)) break;
            const bool room = sp + qt + 7 <= MI_BVH_STACK;
            if (can && (room || empty)) {
                const Bvh4Hits h = bvh4_node(sc, node, ray, inv_d, ray.maxt);
                BVH_STAT(1, 1);
                uint32_t next = MI_BVH_EMPTY;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (h.tn[c] == MI_INF) continue;
                    if (h.e[c] & MI_BVH_LEAF) {
                        if (room) { MI_QSTK(qt) = h.e[c]; qt++; }
                        else {
                            const uint32_t first = h.e[c] & 0x1fffffffu, count = ((h.e[c] >> 29) & 3u) + 1u;
                            BVH_STAT(2, count);
                            for (uint32_t k = 0; k < count; ++k) { float t, u, v; if (tri_intersect(sc.tris[first + k], ray, t, u, v)) occluded = true; }
                        }
                    } else { if (next != MI_BVH_EMPTY && sp + qt < MI_BVH_STACK) { MI_STK(sp) = next; sp++; } next = h.e[c]; }
                }
                if (next != MI_BVH_EMPTY) node = next;
                else if (sp > 0) { --sp; node = MI_STK(sp); }
                else node = MI_BVH_EMPTY;
                if (occluded) { node = MI_BVH_EMPTY; sp = 0; qh = qt = 0; }
            }
        }
        while (__any(left != 0u || qh != qt)) {
            if (left == 0u && qh != qt) {
                const uint32_t e = MI_QSTK(qh); qh++;
                tri = e & 0x1fffffffu; left = ((e >> 29) & 3u) + 1u;
                if (qh == qt) qh = qt = 0;
            }
            if (left != 0u) {
                float t, u, v;
                BVH_STAT(2, 1);
                if (tri_intersect(sc.tris[tri], ray, t, u, v)) { occluded = true; node = MI_BVH_EMPTY; sp = 0; qh = qt = 0; left = 1u; }
                tri++; left--;
            }
        }
        if (!__any(node != MI_BVH_EMPTY)) break;
    }
    return occluded;
}
#else
template <bool WT = false>
MI_DEV void bvh_closest(const DScene &sc, Ray &ray, v3 inv_d, Hit &hit, bool active, uint32_t *stk) {
    int sp = 0; uint32_t node = 0u; bool have = active;
    if (active) BVH_STAT(0, 1);
    while (have) {
        Bvh4Hits h = bvh4_node(sc, node, ray, inv_d, ray.maxt);
        BVH_STAT(1, 1);
        bvh4_sort(h);
        uint32_t next = MI_BVH_EMPTY;
#pragma unroll
        for (int c = 3; c >= 0; --c)                                       // far -> near: the nearest inner child ends up in `next`, the others on the stack
            if (h.tn[c] != MI_INF && !(h.e[c] & MI_BVH_LEAF)) { if (next != MI_BVH_EMPTY && sp < MI_BVH_STACK) { MI_STK(sp) = next; sp++; } next = h.e[c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c)                                        // near -> far: leaves are intersected at once
            if (h.tn[c] <= ray.maxt && (h.e[c] & MI_BVH_LEAF)) {
                const uint32_t first = h.e[c] & 0x1fffffffu, count = ((h.e[c] >> 29) & 3u) + 1u;
                BVH_STAT(2, count);
                for (uint32_t k = 0; k < count; ++k) {
                    float t, u, v;
                    if (tri_hit<WT>(sc, first + k, ray, t, u, v) && t < hit.t) {
                        hit.t = t; hit.u = u; hit.v = v; hit.prim = sc.tri_face[first + k]; hit.shape = sc.tri_shape[first + k]; ray.maxt = t;
                    }
                }
            }
        if (next != MI_BVH_EMPTY) node = next;
        else if (sp > 0) { --sp; node = MI_STK(sp); }
        else have = false;
    }
}
MI_DEV bool bvh_any(const DScene &sc, const Ray &ray, v3 inv_d, bool active, uint32_t *stk) {
    bool occluded = false;
    int sp = 0; uint32_t node = 0u; bool have = active;
    if (active) BVH_STAT(0, 1);
    while (have) {
        const Bvh4Hits h = bvh4_node(sc, node, ray, inv_d, ray.maxt);
        BVH_STAT(1, 1);
        uint32_t next = MI_BVH_EMPTY;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (h.tn[c] == MI_INF) continue;
            if (h.e[c] & MI_BVH_LEAF) {
                const uint32_t first = h.e[c] & 0x1fffffffu, count = ((h.e[c] >> 29) & 3u) + 1u;
                BVH_STAT(2, count);
                for (uint32_t k = 0; k < count; ++k) { float t, u, v; if (tri_intersect(sc.tris[first + k], ray, t, u, v)) occluded = true; }
            } else { if (next != MI_BVH_EMPTY && sp < MI_BVH_STACK) { MI_STK(sp) = next; sp++; } next = h.e[c]; }
        }
        if (occluded) break;
        if (next != MI_BVH_EMPTY) node = next;
        else if (sp > 0) { --sp; node = MI_STK(sp); }
        else have = false;
    }
    return occluded;
}
#endif

// One flat primitive against one ray: 18 fma for the affine map, one v_rcp, two fma, then a branch-free
// bounds test.  `tlim` is an EXCLUSIVE upper bound on t (next float above ray.maxt, or the closest hit so far).
// Tables indexed by a wave-uniform loop counter are read through the constant address space: the loads are then
// scalar (s_load -> SGPR operands) no matter which stores / atomics the surrounding kernel performs.  (Through a plain
// global pointer the compiler falls back to per-lane vector loads as soon as it cannot prove the table unclobbered.)
#define MI_K(T) const __attribute__((address_space(4))) T
template <typename T> MI_DEV MI_K(T) *as_const_table(const T *p) { return (MI_K(T) *) p; }
template <typename P>
MI_DEV bool prim_test(P pr, const Ray &ray, float tlim, float &t, float &s, float &u) {
    auto m = pr->m;
    float os = fmaf(m[MI_PM(2)], ray.o.z, fmaf(m[MI_PM(1)], ray.o.y, fmaf(m[MI_PM(0)], ray.o.x, m[MI_PM(3)])));
    float ot = fmaf(m[MI_PM(6)], ray.o.z, fmaf(m[MI_PM(5)], ray.o.y, fmaf(m[MI_PM(4)], ray.o.x, m[MI_PM(7)])));
    float oh = fmaf(m[10], ray.o.z, fmaf(m[9], ray.o.y, fmaf(m[8], ray.o.x, m[11])));
    float ds = fmaf(m[MI_PM(2)], ray.d.z, fmaf(m[MI_PM(1)], ray.d.y, m[MI_PM(0)] * ray.d.x));
    float dt = fmaf(m[MI_PM(6)], ray.d.z, fmaf(m[MI_PM(5)], ray.d.y, m[MI_PM(4)] * ray.d.x));
    float dh = fmaf(m[10], ray.d.z, fmaf(m[9], ray.d.y, m[8] * ray.d.x));
    t = -oh * rcp_(dh);
    s = fmaf(ds, t, os); u = fmaf(dt, t, ot);
    return ((int) (t >= 0.f) & (int) (t < tlim) & (int) (fminf(s, u) >= pr->lo) & (int) (fmaxf(s, u) <= 1.f) & (int) (s + u <= pr->smax)) != 0;   // branch-free on purpose
}
// Software-pipelined scalar loads for the flat-primitive loops.  Written as `load record k; test record k` the compiler puts an
// s_waitcnt in front of every test, and the scalar-cache latency of each 64-byte record is exposed once per primitive and ray
// (profiles/r02_a: 43 % of the wave-cycles of the forward kernel are spent parked at s_waitcnt).  A value-level rewrite (load k + 1 into a
// second struct, test k) is rotated straight back by the optimiser, so the loads are issued by hand: prim_issue() starts the three
// s_load of a record into a register tuple the compiler cannot sink, prim_wait() is the s_waitcnt that makes them usable.  The loops
// issue record k + 1 before they test record k.  The compiler's own wait-count bookkeeping does not know about these loads, so every
// loop DRAINS them (prim_drain) before it is left: no hand-issued load is ever outstanding in compiler-scheduled code.
// MEASURED (round 2, MI355X, Cornell 512^2 @ 256 spp, profiles/r02_prim_pipeline_on.json vs _off.json): k_render_compact 9.44 ms with the
// pipelined loads against 9.02 ms with the plain loop -- the second register set pushes the kernel past its SGPR budget (30 v_writelane /
// v_readlane spill moves instead of 4) and the scalar cache already answers the 15 records within the shadow of 4-5 resident waves.
// It therefore stays OFF; -DMI_PRIM_PIPELINE=1 rebuilds the experiment.
#ifndef MI_PRIM_PIPELINE
#define MI_PRIM_PIPELINE 0
#endif
typedef float mi_f32x8 __attribute__((ext_vector_type(8)));
typedef float mi_f32x4 __attribute__((ext_vector_type(4)));
typedef float mi_f32x2 __attribute__((ext_vector_type(2)));
struct PrimRegs { mi_f32x8 a; mi_f32x4 b; mi_f32x2 c; };                    // m[0..7] | m[8..11] | lo, smax
MI_DEV PrimRegs prim_issue(MI_K(DPrim) *p) {
    PrimRegs r;
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0x30" : "=&s"(r.a), "=&s"(r.b), "=&s"(r.c) : "s"(p));
    __builtin_amdgcn_sched_barrier(0);                                      // the test of the OTHER record stays behind the issue: that is the overlap
    return r;
}
MI_DEV void prim_wait(PrimRegs &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r.a), "+s"(r.b), "+s"(r.c)); }
// wait for `cur`, then start the next record into `nxt`.  `cur` passes THROUGH the statement, so every instruction of cur's test depends on
// it and stays behind the issue of the next load (as two statements the optimiser sank the issue below the test in one of the two halves).
MI_DEV void prim_wait_issue(PrimRegs &cur, PrimRegs &nxt, MI_K(DPrim) *p) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_load_dwordx8 %0, %6, 0x0\n\ts_load_dwordx4 %1, %6, 0x20\n\ts_load_dwordx2 %2, %6, 0x30"
                 : "=&s"(nxt.a), "=&s"(nxt.b), "=&s"(nxt.c), "+s"(cur.a), "+s"(cur.b), "+s"(cur.c) : "s"(p));
}
MI_DEV void prim_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
MI_DEV bool prim_test_regs(const PrimRegs &r, const Ray &ray, float tlim, float &t, float &s, float &u) {
    float os = fmaf(r.a[MI_PM(2)], ray.o.z, fmaf(r.a[MI_PM(1)], ray.o.y, fmaf(r.a[MI_PM(0)], ray.o.x, r.a[MI_PM(3)])));
    float ot = fmaf(r.a[MI_PM(6)], ray.o.z, fmaf(r.a[MI_PM(5)], ray.o.y, fmaf(r.a[MI_PM(4)], ray.o.x, r.a[MI_PM(7)])));
    float oh = fmaf(r.b[2], ray.o.z, fmaf(r.b[1], ray.o.y, fmaf(r.b[0], ray.o.x, r.b[3])));
    float ds = fmaf(r.a[MI_PM(2)], ray.d.z, fmaf(r.a[MI_PM(1)], ray.d.y, r.a[MI_PM(0)] * ray.d.x));
    float dt = fmaf(r.a[MI_PM(6)], ray.d.z, fmaf(r.a[MI_PM(5)], ray.d.y, r.a[MI_PM(4)] * ray.d.x));
    float dh = fmaf(r.b[2], ray.d.z, fmaf(r.b[1], ray.d.y, r.b[0] * ray.d.x));
    t = -oh * rcp_(dh);
    s = fmaf(ds, t, os); u = fmaf(dt, t, ot);
    return ((int) (t >= 0.f) & (int) (t < tlim) & (int) (fminf(s, u) >= r.c[0]) & (int) (fmaxf(s, u) <= 1.f) & (int) (s + u <= r.c[1])) != 0;
}
MI_DEV float excl_limit(float maxt) { return maxt >= MI_LARGEST ? MI_INF : __uint_as_float(__float_as_uint(maxt) + 1u); }

// Slab test of one oriented box: entering distance if the origin is outside, exiting distance if it is inside
// (the reference's triangles are two-sided for the intersection routine, mesh.h:430-453).
struct BoxLocal { v3 o, d; float tnear, tfar; v3 tn, tf; };
template <typename P>
MI_DEV BoxLocal box_local(P bx, const Ray &ray) {
    auto m = bx->m; BoxLocal b;
    b.o = V3(fmaf(m[2], ray.o.z, fmaf(m[1], ray.o.y, fmaf(m[0], ray.o.x, m[3]))),
             fmaf(m[6], ray.o.z, fmaf(m[5], ray.o.y, fmaf(m[4], ray.o.x, m[7]))),
             fmaf(m[10], ray.o.z, fmaf(m[9], ray.o.y, fmaf(m[8], ray.o.x, m[11]))));
    b.d = V3(fmaf(m[2], ray.d.z, fmaf(m[1], ray.d.y, m[0] * ray.d.x)),
             fmaf(m[6], ray.d.z, fmaf(m[5], ray.d.y, m[4] * ray.d.x)),
             fmaf(m[10], ray.d.z, fmaf(m[9], ray.d.y, m[8] * ray.d.x)));
    v3 inv = V3(rcp_(b.d.x), rcp_(b.d.y), rcp_(b.d.z));
    v3 t0 = V3(-b.o.x * inv.x, -b.o.y * inv.y, -b.o.z * inv.z);
    v3 t1 = V3((1.f - b.o.x) * inv.x, (1.f - b.o.y) * inv.y, (1.f - b.o.z) * inv.z);
    b.tn = V3(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
    b.tf = V3(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
    b.tnear = fmaxf(fmaxf(b.tn.x, b.tn.y), b.tn.z);
    b.tfar = fminf(fminf(b.tf.x, b.tf.y), b.tf.z);
    return b;
}
template <typename P>
MI_DEV bool box_test(P bx, const Ray &ray, float tlim, float &t) {
    BoxLocal b = box_local(bx, ray);
    t = b.tnear < 0.f ? b.tfar : b.tnear;
    return (b.tnear <= b.tfar) & (t >= 0.f) & (t < tlim);
}

// math::solve_quadratic, include/mitsuba/core/math.h:360-402
MI_DEV bool solve_quadratic(float a, float b, float c, float &x0, float &x1) {
    bool linear_case = a == 0.f, valid_linear = linear_case && b != 0.f;
    x0 = x1 = -c / b;
    float discrim = fmaf(b, b, -(4.f * a * c));
    bool valid_quadratic = !linear_case && discrim >= 0.f;
    if (valid_quadratic) {
        float sqrt_discrim = sqrt_(discrim);
        float temp = -0.5f * (b + copysignf(sqrt_discrim, b));
        float x0p = div_(temp, a), x1p = div_(c, temp);
        x0 = fminf(x0p, x1p); x1 = fmaxf(x0p, x1p);
    }
    return valid_linear || valid_quadratic;
}
// The same with IEEE division and square root, whatever MI_EXACT_MATH says: the aspherical lens interface accepts or rejects a ray by
// |dz| < 10 eps after a Newton iteration that STARTS from this root -- a 1-ulp difference of the hardware v_rcp / v_sqrt flips that test for
// a handful of rays, each of which adds or removes a whole path (round 2: asph_singlet gradient 3.6e-3 off; exact: 8.7e-4).  Lens-sensor
// kernels only.
MI_DEV bool solve_quadratic_ieee(float a, float b, float c, float &x0, float &x1) {
    bool linear_case = a == 0.f, valid_linear = linear_case && b != 0.f;
    x0 = x1 = -c / b;
    float discrim = fmaf(b, b, -(4.f * a * c));
    bool valid_quadratic = !linear_case && discrim >= 0.f;
    if (valid_quadratic) {
        float sqrt_discrim = sqrtf(discrim);
        float temp = -0.5f * (b + copysignf(sqrt_discrim, b));
        float x0p = temp / a, x1p = c / temp;
        x0 = fminf(x0p, x1p); x1 = fmaxf(x0p, x1p);
    }
    return valid_linear || valid_quadratic;
}
MI_DEV v3 normalize3_ieee(v3 a) { return a * (1.0f / sqrtf(sqnorm3(a))); }
// Sphere::ray_intersect_preliminary_impl, src/shapes/sphere.cpp:491-563 (closest) and ray_test_impl :565-601 (any)
template <typename P>
MI_DEV bool sphere_intersect(P sp, const Ray &ray, float tlim, float &t) {
    v3 center = V3(sp->center[0], sp->center[1], sp->center[2]);
    v3 l = ray.o - center, d = ray.d;
    float plane_t = div_(dot3(-l, d), norm3(d));
    bool no_hit = plane_t == 0.f && (ray.o.x != center.x && ray.o.y != center.y && ray.o.z != center.z);
    v3 o = ray_at(ray, plane_t) - center;
    no_hit = no_hit && (norm3(o) > sp->radius);
    float A = sqnorm3(d), B = 2.f * dot3(o, d), C = sqnorm3(o) - sp->radius * sp->radius;
    float near_t, far_t; bool found = solve_quadratic(A, B, C, near_t, far_t);
    near_t += plane_t; far_t += plane_t;
    bool out_bounds = !(near_t < tlim && far_t >= 0.f), in_bounds = near_t < 0.f && !(far_t < tlim);
    t = near_t < 0.f ? far_t : near_t;
    return found && !no_hit && !out_bounds && !in_bounds;
}

#define MI_DISK_FIN 0xfffffffeu                                         // DPrim::fin of a disk: the plane test of a rectangle + x^2 + y^2 <= 1
#define MI_BOX_FLAG 0x40000000u
#define MI_SPHERE_FLAG 0x20000000u
// Scene::ray_intersect_preliminary (scene.cpp:192-199): closest hit over all shapes.
template <int FULL, bool WT = false>
MI_DEV Hit scene_intersect(const DScene &sc, Ray ray, bool active, uint32_t *stk) {
    Hit hit; hit.t = MI_INF; hit.u = hit.v = 0.f; hit.prim = 0; hit.shape = 0xffffffffu;
    float tlim = active ? excl_limit(ray.maxt) : 0.f;
    uint32_t best = 0xffffffffu;
    const uint32_t n = sc.n_prims;
    MI_K(DPrim) *prims = as_const_table(sc.prims); MI_K(DBox) *boxes = as_const_table(sc.boxes); MI_K(DSphere) *spheres = as_const_table(sc.spheres);
#if MI_PRIM_PIPELINE
    if (!FULL && n) {                                                       // (FULL scenes test `fin` for disks too: they keep the plain loop)
        PrimRegs A = prim_issue(prims), B;                                  // two register sets, used in turn: a load never waits for the reads of the other set
        for (uint32_t k = 0; k < n;) {
            prim_wait_issue(A, B, prims + (k + 1u < n ? k + 1u : k));          // record k + 1 in flight while record k is tested
            { float t, s, u; const bool h = prim_test_regs(A, ray, tlim, t, s, u); tlim = h ? t : tlim; hit.u = h ? s : hit.u; hit.v = h ? u : hit.v; best = h ? k : best; }
            if (++k >= n) break;
            prim_wait_issue(B, A, prims + (k + 1u < n ? k + 1u : k));
            { float t, s, u; const bool h = prim_test_regs(B, ray, tlim, t, s, u); tlim = h ? t : tlim; hit.u = h ? s : hit.u; hit.v = h ? u : hit.v; best = h ? k : best; }
            ++k;
        }
        prim_drain();
    } else
#endif
    {
    MI_UNROLL_PRIMS
    for (uint32_t k = 0; k < n; ++k) {
        float t, s, u;
        bool h = prim_test(prims + k, ray, tlim, t, s, u);
        if (FULL && prims[k].fin == MI_DISK_FIN) h = h && fmaf(s, s, u * u) <= 1.f;   // Disk::ray_intersect_preliminary_impl, disk.cpp:392-416
        tlim = h ? t : tlim; hit.u = h ? s : hit.u; hit.v = h ? u : hit.v; best = h ? k : best;
    }
    }
    for (uint32_t k = 0; k < sc.n_boxes; ++k) {
        float t;
        bool h = box_test(boxes + k, ray, tlim, t);
        tlim = h ? t : tlim; best = h ? (MI_BOX_FLAG | k) : best;
    }
    if (FULL) for (uint32_t k = 0; k < sc.n_spheres; ++k) {
        float t;
        bool h = sphere_intersect(spheres + k, ray, tlim, t) && t < tlim;
        tlim = h ? t : tlim; best = h ? (MI_SPHERE_FLAG | k) : best;
    }
    if (best != 0xffffffffu) {                                          // per-lane finalize (divergent gather, once per ray)
        hit.t = tlim; ray.maxt = tlim;
        uint32_t fin = 0xffffffffu; float s = hit.u, u = hit.v;
        if (FULL && (best & MI_SPHERE_FLAG)) {
            hit.shape = sc.spheres[best & ~MI_SPHERE_FLAG].shape; hit.prim = 0; hit.u = hit.v = 0.f;
        } else if (best & MI_BOX_FLAG) {
            const DBox &bx = sc.boxes[best & ~MI_BOX_FLAG];
            BoxLocal b = box_local(&bx, ray);                           // same arithmetic as in the loop -> same face decision
            bool inside = b.tnear < 0.f;
            v3 key = inside ? V3(-b.tf.x, -b.tf.y, -b.tf.z) : b.tn;   // entering: arg max tn, exiting: arg min tf
            uint32_t axis = (key.x >= key.y && key.x >= key.z) ? 0u : (key.y >= key.z ? 1u : 2u);
            float da = axis == 0 ? b.d.x : (axis == 1 ? b.d.y : b.d.z);
            uint32_t side = (da > 0.f) == inside ? 1u : 0u;            // entering through the low face iff d > 0
            v3 pl = fma3s(b.d, tlim, b.o);
            s = axis == 0 ? pl.y : pl.x; u = axis == 2 ? pl.y : pl.z;
            s = fminf(fmaxf(s, 0.f), 1.f); u = fminf(fmaxf(u, 0.f), 1.f);
            hit.shape = bx.shape; fin = bx.fin + 2u * axis + side;
        } else {
            const DPrim &pr = sc.prims[best];
            hit.shape = pr.shape; fin = pr.fin;
            if (FULL && fin == MI_DISK_FIN) fin = 0xffffffffu;         // a disk keeps its local (x, y) like a rectangle
        }
        if (fin != 0xffffffffu) {
            const DPrimFin &f = MI_SC_FIN(sc, fin);
            float b1 = fmaf(f.a[0], s, fmaf(f.a[1], u, f.a[2])), b2 = fmaf(f.a[3], s, fmaf(f.a[4], u, f.a[5]));
            hit.prim = f.tri_a;
            if (f.tri_b != f.tri_a && fminf(fminf(b1, b2), 1.f - b1 - b2) < 0.f) {
                b1 = fmaxf(fmaf(f.b[0], s, fmaf(f.b[1], u, f.b[2])), 0.f); b2 = fmaxf(fmaf(f.b[3], s, fmaf(f.b[4], u, f.b[5])), 0.f);
                hit.prim = f.tri_b;
            }
            hit.u = b1; hit.v = b2;
        }
    }
    if (FULL == 1 && sc.n_bvh_shapes) {                                          // (a scene with BVH meshes runs the FULL kernels: features bit 3)
        v3 inv_d = V3(rcp_(ray.d.x), rcp_(ray.d.y), rcp_(ray.d.z));
        bvh_closest<WT>(sc, ray, inv_d, hit, active, stk);
    }
    return hit;
}
// Scene::ray_test (scene.cpp:201-210): any hit.
template <int FULL>
MI_DEV bool scene_ray_test(const DScene &sc, const Ray &ray, bool active, uint32_t *stk) {
    bool occluded = false;
    if (!__any(active)) return false;                                      // nobody in this wave asked
    float tlim = active ? excl_limit(ray.maxt) : 0.f;
    const uint32_t n = sc.n_prims;
    MI_K(DPrim) *prims = as_const_table(sc.prims); MI_K(DBox) *boxes = as_const_table(sc.boxes); MI_K(DSphere) *spheres = as_const_table(sc.spheres);
#if MI_PRIM_PIPELINE
    if (!FULL && n) {
        PrimRegs A = prim_issue(prims), B;
        for (uint32_t k = 0; k < n;) {
            prim_wait_issue(A, B, prims + (k + 1u < n ? k + 1u : k));
            { float t, s, u; if (prim_test_regs(A, ray, tlim, t, s, u)) { occluded = true; tlim = 0.f; } }
            if (++k >= n) break;
            prim_wait_issue(B, A, prims + (k + 1u < n ? k + 1u : k));
            { float t, s, u; if (prim_test_regs(B, ray, tlim, t, s, u)) { occluded = true; tlim = 0.f; } }
            ++k;
        }
        prim_drain();
    } else
#endif
    {
    MI_UNROLL_PRIMS
    for (uint32_t k = 0; k < n; ++k) {
        float t, s, u;
        bool h = prim_test(prims + k, ray, tlim, t, s, u);
        if (FULL && prims[k].fin == MI_DISK_FIN) h = h && fmaf(s, s, u * u) <= 1.f;   // Disk::ray_test_impl, disk.cpp:418-431
        if (h) { occluded = true; tlim = 0.f; }
    }
    }
    for (uint32_t k = 0; k < sc.n_boxes; ++k) {
        float t;
        if (box_test(boxes + k, ray, tlim, t)) { occluded = true; tlim = 0.f; }
    }
    if (FULL) for (uint32_t k = 0; k < sc.n_spheres; ++k) {
        float t;
        if (sphere_intersect(spheres + k, ray, tlim, t)) { occluded = true; tlim = 0.f; }
    }
    if (FULL == 1 && sc.n_bvh_shapes) {
        v3 inv_d = V3(rcp_(ray.d.x), rcp_(ray.d.y), rcp_(ray.d.z));
        occluded = bvh_any(sc, ray, inv_d, active && !occluded, stk) || occluded;
    }
    return occluded;
}

// PreliminaryIntersection::compute_surface_interaction (interaction.h:642-668) ->
// Rectangle / Mesh::compute_surface_interaction (rectangle.cpp:497-567 IsDiff branch, mesh.cpp:1375-1548)
// -> finalize_surface_interaction (interaction.h:464-484).  Caller guarantees hit.t != inf.
template <int FULL>
MI_DEV SI compute_si(const DScene &sc, const Ray &ray, const Hit &hit) {
    SI si; si.t = hit.t; si.shape = (int32_t) hit.shape; si.prim = hit.prim;
    const DShape &sh = MI_SC_SHAPE(sc, hit.shape);                         // per-lane gather (divergent index)
    v3 dp_du, dp_dv;
    if (sh.type == MI355_SHAPE_RECTANGLE) {
        si.p = ray_at(ray, hit.t);
        si.n = ld3(sh.frame_n); si.sh_n = si.n; dp_du = ld3(sh.frame_s); dp_dv = ld3(sh.frame_t);
        si.uv.x = fmaf(hit.u, .5f, .5f); si.uv.y = fmaf(hit.v, .5f, .5f);
    } else if (FULL && sh.type == MI355_SHAPE_DISK) {                     // Disk::compute_surface_interaction, disk.cpp:436-519
        si.p = ray_at(ray, hit.t);
        float r = sqrtf(fmaf(hit.u, hit.u, hit.v * hit.v)), inv_r = 1.f / r;
        float v = atan2f(hit.v, hit.u) * (0.5f * MI_INV_PI); if (v < 0.f) v += 1.f;
        si.uv.x = r; si.uv.y = v;
        float cos_phi = r != 0.f ? hit.u * inv_r : 1.f, sin_phi = r != 0.f ? hit.v * inv_r : 0.f;
        dp_du = xf_vector(sh.to_world, V3(cos_phi, sin_phi, 0.f)); dp_dv = xf_vector(sh.to_world, V3(-sin_phi, cos_phi, 0.f));
        si.n = ld3(sh.frame_n); si.sh_n = si.n;
    } else if (FULL && sh.type == MI355_SHAPE_SPHERE) {                   // sphere.cpp:609-712, IsDiff && !FollowShape
        v3 center = ld3(sh.frame_s);
        si.p = ray_at(ray, hit.t);
        si.sh_n = normalize3(si.p - center);
        v3 local = xf_point(sh.to_object, si.p);
        float rd_2 = local.x * local.x + local.y * local.y;
        v3 dz = V3(local.x, local.y, local.z - mulsign(1.f, local.z));      // dr::unit_angle_z: robust acos(z)
        float temp = 2.f * asinf(.5f * norm3(dz));
        float theta = local.z >= 0.f ? temp : MI_PI - temp;
        float phi = atan2f(local.y, local.x); if (phi < 0.f) phi += 2.f * MI_PI;
        si.uv.x = phi * (0.5f * MI_INV_PI); si.uv.y = theta * MI_INV_PI;
        dp_du = V3(-local.y, local.x, 0.f);
        float rd = sqrt_(rd_2), inv_rd = rcp_(rd);
        dp_dv = V3(local.z * (local.x * inv_rd), local.z * (local.y * inv_rd), -rd);
        if (rd == 0.f) dp_dv = V3(1.f, 0.f, 0.f);
        dp_du = xf_vector(sh.to_world, dp_du) * (2.f * MI_PI); dp_dv = xf_vector(sh.to_world, dp_dv) * MI_PI;
        if (sh.flags & DSHAPE_FLIP) si.sh_n = -si.sh_n;
        si.n = si.sh_n;
    } else {
        const uint32_t *fi = sc.faces + 3 * (size_t) (sh.tri_offset + hit.prim);
        uint32_t i0 = sh.vtx_offset + fi[0], i1 = sh.vtx_offset + fi[1], i2 = sh.vtx_offset + fi[2];
        v3 p0 = ld3(sc.positions + 3 * (size_t) i0), p1 = ld3(sc.positions + 3 * (size_t) i1), p2 = ld3(sc.positions + 3 * (size_t) i2);
        v3 dp0 = p1 - p0, dp1 = p2 - p0;
        float b1 = hit.u, b2 = hit.v, b0 = 1.f - b1 - b2;
        si.p = fma3s(p0, b0, fma3s(p1, b1, p2 * b2));
        si.n = normalize3(cross3(dp0, dp1));
        si.uv.x = b1; si.uv.y = b2;
        coordinate_system(si.n, dp_du, dp_dv);
        if (sh.flags & DSHAPE_HAS_UV) {
            const float *t0 = sc.texcoords + 2 * (size_t) i0, *t1 = sc.texcoords + 2 * (size_t) i1, *t2 = sc.texcoords + 2 * (size_t) i2;
            v2 uv0 = { t0[0], t0[1] }, uv1 = { t1[0], t1[1] }, uv2 = { t2[0], t2[1] };
            si.uv.x = fmaf(uv2.x, b2, fmaf(uv1.x, b1, uv0.x * b0));
            si.uv.y = fmaf(uv2.y, b2, fmaf(uv1.y, b1, uv0.y * b0));
            v2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
            float det = fmaf(duv0.x, duv1.y, -(duv0.y * duv1.x)), inv_det = rcp_(det);
            if (det != 0.f) {
                dp_du = V3(fmaf(duv1.y, dp0.x, -(duv0.y * dp1.x)), fmaf(duv1.y, dp0.y, -(duv0.y * dp1.y)), fmaf(duv1.y, dp0.z, -(duv0.y * dp1.z))) * inv_det;
                dp_dv = V3(fmaf(-duv1.x, dp0.x, duv0.x * dp1.x), fmaf(-duv1.x, dp0.y, duv0.x * dp1.y), fmaf(-duv1.x, dp0.z, duv0.x * dp1.z)) * inv_det;
            }
        }
        if (sh.flags & DSHAPE_HAS_NRM) {
            v3 n0 = ld3(sc.normals + 3 * (size_t) i0), n1 = ld3(sc.normals + 3 * (size_t) i1), n2 = ld3(sc.normals + 3 * (size_t) i2);
            v3 n = fma3s(n2, b2, fma3s(n1, b1, n0 * b0));
            si.sh_n = n * rsqrt_(sqnorm3(n));
        } else si.sh_n = si.n;
        if (sh.flags & DSHAPE_FLIP) { si.n = -si.n; si.sh_n = -si.sh_n; }
    }
    si.sh_s = normalize3(fma3s(si.sh_n, -dot3(si.sh_n, dp_du), dp_du));   // interaction.h:245-255
    if (dp_du.x == 0.f && dp_du.y == 0.f && dp_du.z == 0.f) { v3 tt; coordinate_system(si.sh_n, si.sh_s, tt); }
    si.sh_t = cross3(si.sh_n, si.sh_s);
    v3 md = -ray.d;
    si.wi = V3(dot3(md, si.sh_s), dot3(md, si.sh_t), dot3(md, si.sh_n));
    return si;
}
MI_DEV v3 si_to_world(const SI &si, v3 v) { return fma3s(si.sh_n, v.z, fma3s(si.sh_t, v.y, si.sh_s * v.x)); }
MI_DEV v3 si_to_local(const SI &si, v3 v) { return V3(dot3(v, si.sh_s), dot3(v, si.sh_t), dot3(v, si.sh_n)); }
MI_DEV v3 offset_p(v3 p, v3 n, v3 d) {                                    // interaction.h:158-162
    float mag = (1.f + fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z))) * MI_RAY_EPS;
    mag = mulsign(mag, dot3(n, d));
    return fma3s(n, mag, p);
}
MI_DEV Ray spawn_ray(v3 p, v3 n, v3 d) { Ray r; r.o = offset_p(p, n, d); r.d = d; r.maxt = MI_LARGEST; return r; }
MI_DEV Ray spawn_ray_to(v3 p, v3 n, v3 target) {                          // interaction.h:138-145
    Ray r; r.o = offset_p(p, n, target - p);
    v3 d = target - r.o; float dist = norm3(d);
    r.d = d * rcp_(dist); r.maxt = dist * (1.f - MI_SHADOW_EPS); return r;
}

// -------------------------------------------------------------------------------------------
// textures / BSDF / emitters
// -------------------------------------------------------------------------------------------
MI_DEV int32_t wrap_i(int32_t p, int32_t n, uint32_t mode) {
    if (mode == MI355_WRAP_CLAMP) return p < 0 ? 0 : (p > n - 1 ? n - 1 : p);
    if (mode == MI355_WRAP_REPEAT) { int32_t m = p % n; return m < 0 ? m + n : m; }
    int32_t m = p % (2 * n); if (m < 0) m += 2 * n; return m < n ? m : 2 * n - 1 - m;
}
struct TexTaps { int32_t idx[4]; float w[4]; };
// BitmapTexture::interpolate_3, bitmap.cpp:696-710 (dr::Texture2f::eval: pos = uv*res - 0.5, 4-tap lerp)
MI_DEV v3 texture_eval(const DScene &sc, const DTexture &t, v2 uv_in, TexTaps *taps) {
    float ux = fmaf(t.to_uv[0], uv_in.x, fmaf(t.to_uv[1], uv_in.y, t.to_uv[2]));
    float uy = fmaf(t.to_uv[3], uv_in.x, fmaf(t.to_uv[4], uv_in.y, t.to_uv[5]));
    int32_t W = (int32_t) t.w, H = (int32_t) t.h, C = (int32_t) t.c;
    const float *data = sc.texdata + t.data_offset;
    float out[3];
    if (t.filter == MI355_FILTER_NEAREST) {
        int32_t x = wrap_i((int32_t) floorf(ux * (float) W), W, t.wrap), y = wrap_i((int32_t) floorf(uy * (float) H), H, t.wrap);
        for (int c = 0; c < 3; ++c) out[c] = data[(y * W + x) * C + (C == 3 ? c : 0)];
        if (taps) { taps->idx[0] = y * W + x; taps->w[0] = 1.f; taps->idx[1] = taps->idx[2] = taps->idx[3] = -1; taps->w[1] = taps->w[2] = taps->w[3] = 0.f; }
        return V3(out[0], out[1], out[2]);
    }
    float px = fmaf(ux, (float) W, -.5f), py = fmaf(uy, (float) H, -.5f);
    float fx = floorf(px), fy = floorf(py);
    int32_t ix = (int32_t) fx, iy = (int32_t) fy;
    float w1x = px - fx, w1y = py - fy, w0x = 1.f - w1x, w0y = 1.f - w1y;
    int32_t x0 = wrap_i(ix, W, t.wrap), x1 = wrap_i(ix + 1, W, t.wrap), y0 = wrap_i(iy, H, t.wrap), y1 = wrap_i(iy + 1, H, t.wrap);
    int32_t i00 = y0 * W + x0, i10 = y0 * W + x1, i01 = y1 * W + x0, i11 = y1 * W + x1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int cc = C == 3 ? c : 0;
        float v00 = data[i00 * C + cc], v10 = data[i10 * C + cc], v01 = data[i01 * C + cc], v11 = data[i11 * C + cc];
        out[c] = fmaf(w0y, fmaf(w0x, v00, w1x * v10), w1y * fmaf(w0x, v01, w1x * v11));
    }
    if (taps) { taps->idx[0] = i00; taps->idx[1] = i10; taps->idx[2] = i01; taps->idx[3] = i11;
                taps->w[0] = w0y * w0x; taps->w[1] = w0y * w1x; taps->w[2] = w1y * w0x; taps->w[3] = w1y * w1x; }
    return V3(out[0], out[1], out[2]);
}
// SmoothDiffuse::eval_pdf, diffuse.cpp:160-180
MI_DEV void diffuse_eval_pdf(v3 refl, v3 wi, v3 wo, bool active, v3 &val, float &pdf) {
    active = active && wi.z > 0.f && wo.z > 0.f;
    val = active ? (refl * MI_INV_PI) * wo.z : V3(0.f, 0.f, 0.f);
    pdf = active ? MI_INV_PI * wo.z : 0.f;
}
// SmoothDiffuse::sample, diffuse.cpp:101-125 (JIT semantics: weight masked, fields always written)
MI_DEV void diffuse_sample(v3 refl, v3 wi, float s2x, float s2y, bool active, v3 &wo, float &pdf, v3 &weight) {
    active = active && wi.z > 0.f;
    wo = square_to_cosine_hemisphere(s2x, s2y);
    pdf = MI_INV_PI * wo.z;
    weight = (active && pdf > 0.f) ? refl : V3(0.f, 0.f, 0.f);
}

// -------------------------------------------------------------------------------------------
// microfacet distribution (include/mitsuba/render/microfacet.h), Fresnel (fresnel.h), rough conductor, smooth dielectric
// -------------------------------------------------------------------------------------------
MI_DEV float erfinv_(float x) {                                            // Giles 2010, single precision
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
struct MF { uint32_t type; float au, av; bool sample_visible; };
MI_DEV MF mf_make(const DBsdf &b) { MF m; m.type = b.distribution; m.au = fmaxf(b.alpha_u, 1e-4f); m.av = fmaxf(b.alpha_v, 1e-4f); m.sample_visible = b.sample_visible != 0; return m; }
MI_DEV float sqr_(float x) { return x * x; }
MI_DEV float safe_sqrt_(float x) { return sqrtf(fmaxf(x, 0.f)); }
MI_DEV float mf_eval(const MF &d, v3 m) {                                  // microfacet.h:185-207
    float alpha_uv = d.au * d.av, ct = m.z, ct2 = ct * ct, result;
    if (d.type == 0) result = expf(-(sqr_(m.x / d.au) + sqr_(m.y / d.av)) / ct2) / (MI_PI * alpha_uv * sqr_(ct2));
    else result = 1.0f / (MI_PI * alpha_uv * sqr_(sqr_(m.x / d.au) + sqr_(m.y / d.av) + sqr_(m.z)));
    return result * ct > 1e-20f ? result : 0.f;
}
MI_DEV float mf_smith_g1(const MF &d, v3 v, v3 m) {                        // microfacet.h:341-365
    float xy_alpha_2 = sqr_(d.au * v.x) + sqr_(d.av * v.y), tan_theta_alpha_2 = xy_alpha_2 / sqr_(v.z), result;
    if (d.type == 0) {
        float a = 1.0f / sqrtf(tan_theta_alpha_2), a_sqr = a * a;
        result = a >= 1.6f ? 1.f : (3.535f * a + 2.181f * a_sqr) / (1.f + 2.276f * a + 2.577f * a_sqr);
    } else result = 2.f / (1.f + sqrtf(1.f + tan_theta_alpha_2));
    if (xy_alpha_2 == 0.f) result = 1.f;
    if (dot3(v, m) * v.z <= 0.f) result = 0.f;
    return result;
}
MI_DEV float mf_pdf(const MF &d, v3 wi, v3 m) {                            // microfacet.h:219-228
    float result = mf_eval(d, m);
    if (d.sample_visible) result *= mf_smith_g1(d, wi, m) * fabsf(dot3(wi, m)) / wi.z; else result *= m.z;
    return result;
}
MI_DEV void disk_concentric(float sx, float sy, float &px, float &py) {     // warp.h:54-90
    float x = fmaf(2.f, sx, -1.f), y = fmaf(2.f, sy, -1.f);
    bool is_zero = (x == 0.f && y == 0.f), q13 = fabsf(x) < fabsf(y);
    float r = q13 ? y : x, rp = q13 ? x : y;
    float phi = 0.25f * MI_PI * rp / r;
    if (q13) phi = 0.5f * MI_PI - phi;
    if (is_zero) phi = 0.f;
    float sn, cs; sincosf(phi, &sn, &cs); px = r * cs; py = r * sn;
}
MI_DEV void mf_sample_visible_11(const MF &d, float cos_theta_i, float sx, float sy, float &ox, float &oy) {   // microfacet.h:368-421
    if (d.type == 0) {
        float tan_theta_i = safe_sqrt_(fmaf(-cos_theta_i, cos_theta_i, 1.f)) / cos_theta_i, cot_theta_i = 1.0f / tan_theta_i;
        float maxval = erff(cot_theta_i);
        sx = fmaxf(fminf(sx, 1.f - 1e-6f), 1e-6f); sy = fmaxf(fminf(sy, 1.f - 1e-6f), 1e-6f);
        float x = maxval - (maxval + 1.f) * erff(sqrtf(-logf(sx)));
        const float inv_sqrt_pi = 0.56418958354775628695f;
        sx *= 1.f + maxval + inv_sqrt_pi * tan_theta_i * expf(-sqr_(cot_theta_i));
        for (int i = 0; i < 3; ++i) {
            float slope = erfinv_(x), value = 1.f + x + inv_sqrt_pi * tan_theta_i * expf(-sqr_(slope)) - sx, derivative = 1.f - slope * tan_theta_i;
            x -= value / derivative;
        }
        ox = erfinv_(x); oy = erfinv_(fmaf(2.f, sy, -1.f));
    } else {
        float px, py; disk_concentric(sx, sy, px, py);
        float sfac = 0.5f * (1.f + cos_theta_i), a = safe_sqrt_(1.f - sqr_(px));
        py = fmaf(py - a, sfac, a);
        float z = safe_sqrt_(1.f - fmaf(py, py, px * px));
        float sin_theta_i = safe_sqrt_(1.f - sqr_(cos_theta_i));
        float norm = 1.0f / fmaf(sin_theta_i, py, cos_theta_i * z);
        ox = fmaf(cos_theta_i, py, -(sin_theta_i * z)) * norm; oy = px * norm;
    }
}
MI_DEV v3 mf_sample(const MF &d, v3 wi, float sx, float sy, float &pdf) {   // microfacet.h:244-326
    if (!d.sample_visible) {
        float sin_phi, cos_phi, cos_theta, cos_theta_2, alpha_2;
        if (d.au == d.av) { sincosf((2.f * MI_PI) * sy, &sin_phi, &cos_phi); alpha_2 = d.au * d.au; }
        else {
            float ratio = d.av / d.au, tmp = ratio * tanf((2.f * MI_PI) * sy);
            cos_phi = 1.0f / sqrtf(fmaf(tmp, tmp, 1.f)); cos_phi = mulsign(cos_phi, fabsf(sy - .5f) - .25f);
            sin_phi = cos_phi * tmp;
            alpha_2 = 1.0f / (sqr_(cos_phi / d.au) + sqr_(sin_phi / d.av));
        }
        if (d.type == 0) {
            cos_theta = 1.0f / sqrtf(fmaf(-alpha_2, logf(1.f - sx), 1.f)); cos_theta_2 = cos_theta * cos_theta;
            float cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = (1.f - sx) / (MI_PI * d.au * d.av * cos_theta_3);
        } else {
            float tan_theta_m_2 = alpha_2 * sx / (1.f - sx);
            cos_theta = 1.0f / sqrtf(1.f + tan_theta_m_2); cos_theta_2 = cos_theta * cos_theta;
            float temp = 1.f + tan_theta_m_2 / alpha_2, cos_theta_3 = fmaxf(cos_theta_2 * cos_theta, 1e-20f);
            pdf = 1.0f / (MI_PI * d.au * d.av * cos_theta_3 * sqr_(temp));
        }
        float sin_theta = sqrtf(1.f - cos_theta_2);
        return V3(cos_phi * sin_theta, sin_phi * sin_theta, cos_theta);
    }
    v3 wi_p = normalize3(V3(d.au * wi.x, d.av * wi.y, wi.z));
    float st2 = fmaf(wi_p.x, wi_p.x, wi_p.y * wi_p.y), ist = 1.0f / sqrtf(st2);   // Frame3f::sincos_phi, frame.h:111-122
    float cos_phi = fminf(fmaxf(wi_p.x * ist, -1.f), 1.f), sin_phi = fminf(fmaxf(wi_p.y * ist, -1.f), 1.f);
    if (fabsf(st2) <= 4.f * MI_EPS) { cos_phi = 1.f; sin_phi = 0.f; }
    float slx, sly; mf_sample_visible_11(d, wi_p.z, sx, sy, slx, sly);
    float ax = fmaf(cos_phi, slx, -(sin_phi * sly)) * d.au, ay = fmaf(sin_phi, slx, cos_phi * sly) * d.av;
    v3 m = normalize3(V3(-ax, -ay, 1.f));
    pdf = mf_eval(d, m) * mf_smith_g1(d, wi, m) * fabsf(dot3(wi, m)) / wi.z;
    return m;
}
MI_DEV void fresnel_(float cos_theta_i, float eta, float &r, float &cos_theta_t, float &eta_it, float &eta_ti) {   // fresnel.h:34-72
    bool outside = cos_theta_i >= 0.f;
    float rcp_eta = 1.0f / eta; eta_it = outside ? eta : rcp_eta; eta_ti = outside ? rcp_eta : eta;
    float cos_theta_t_sqr = fmaf(-fmaf(-cos_theta_i, cos_theta_i, 1.f), eta_ti * eta_ti, 1.f);
    float cia = fabsf(cos_theta_i), cta = safe_sqrt_(cos_theta_t_sqr);
    bool index_matched = eta == 1.f, special = index_matched || cia == 0.f;
    float a_s = fmaf(-eta_it, cta, cia) / fmaf(eta_it, cta, cia), a_p = fmaf(-eta_it, cia, cta) / fmaf(eta_it, cia, cta);
    r = 0.5f * (a_s * a_s + a_p * a_p);
    if (special) r = index_matched ? 0.f : 1.f;
    cos_theta_t = mulsign(-cta, cos_theta_i);
}
MI_DEV float fresnel_conductor_(float cos_theta_i, float eta_r, float eta_i) {   // fresnel.h:92-118
    float ci2 = cos_theta_i * cos_theta_i, si2 = 1.f - ci2, si4 = si2 * si2;
    float temp_1 = eta_r * eta_r - eta_i * eta_i - si2, a_2_pb_2 = safe_sqrt_(temp_1 * temp_1 + 4.f * eta_i * eta_i * eta_r * eta_r),
          a = safe_sqrt_(.5f * (a_2_pb_2 + temp_1));
    float term_1 = a_2_pb_2 + ci2, term_2 = 2.f * cos_theta_i * a;
    float r_s = (term_1 - term_2) / (term_1 + term_2);
    float term_3 = a_2_pb_2 * ci2 + si4, term_4 = term_2 * si2;
    float r_p = r_s * (term_3 - term_4) / (term_3 + term_4);
    return 0.5f * (r_s + r_p);
}
struct BSample { v3 wo; float pdf, eta; bool delta; };
MI_DEV void roughconductor_eval_pdf(const DBsdf &b, v3 wi, v3 wo, bool active, v3 &val, float &pdf) {   // roughconductor.cpp:428-517
    val = V3(0.f, 0.f, 0.f); pdf = 0.f;
    v3 H = normalize3(wo + wi);
    active = active && wi.z > 0.f && wo.z > 0.f && dot3(wi, H) > 0.f && dot3(wo, H) > 0.f;
    if (!active) return;
    MF d = mf_make(b);
    float D = mf_eval(d, H);
    if (D == 0.f) return;
    float g1_wi = mf_smith_g1(d, wi, H), G = g1_wi * mf_smith_g1(d, wo, H);
    float value = D * G / (4.f * wi.z), ct = dot3(wi, H);
    v3 F = V3(fresnel_conductor_(ct, b.eta[0], b.k[0]), fresnel_conductor_(ct, b.eta[1], b.k[1]), fresnel_conductor_(ct, b.eta[2], b.k[2]));
    val = F * (ld3(b.reflectance) * value);
    pdf = d.sample_visible ? D * g1_wi / (4.f * wi.z) : mf_pdf(d, wi, H) / (4.f * dot3(wo, H));
}
MI_DEV void roughconductor_sample(const DBsdf &b, v3 wi, float sx, float sy, bool active, BSample &bs, v3 &weight) {   // roughconductor.cpp:226-312
    weight = V3(0.f, 0.f, 0.f);
    active = active && wi.z > 0.f;
    MF d = mf_make(b);
    float pdf; v3 m = mf_sample(d, wi, sx, sy, pdf);
    bs.wo = m * (2.f * dot3(wi, m)) - wi; bs.eta = 1.f; bs.delta = false;
    active = active && pdf != 0.f && bs.wo.z > 0.f;
    float w = d.sample_visible ? mf_smith_g1(d, bs.wo, m) : mf_smith_g1(d, wi, m) * mf_smith_g1(d, bs.wo, m) * dot3(wi, m) / (wi.z * m.z);
    bs.pdf = pdf / (4.f * dot3(bs.wo, m));
    float ct = dot3(wi, m);
    v3 F = V3(fresnel_conductor_(ct, b.eta[0], b.k[0]), fresnel_conductor_(ct, b.eta[1], b.k[1]), fresnel_conductor_(ct, b.eta[2], b.k[2]));
    if (active) weight = F * (ld3(b.reflectance) * w);
}
// `importance` = TransportMode::Importance (particle tracer): no radiance scaling of the transmitted lobe (dielectric.cpp:361-366)
MI_DEV void dielectric_sample(const DBsdf &b, v3 wi, float s1, bool active, BSample &bs, v3 &weight, bool importance = false) {   // dielectric.cpp:246-371
    float r_i, cos_theta_t, eta_it, eta_ti; fresnel_(wi.z, b.rel_eta, r_i, cos_theta_t, eta_it, eta_ti);
    bool selected_r = s1 <= r_i && active;
    bs.pdf = selected_r ? r_i : 1.f - r_i;
    bs.wo = selected_r ? V3(-wi.x, -wi.y, wi.z) : V3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t);
    bs.eta = selected_r ? 1.f : eta_it; bs.delta = true;
    v3 w = selected_r ? ld3(b.reflectance) : (importance ? ld3(b.transmittance) : ld3(b.transmittance) * (eta_ti * eta_ti));
    weight = active ? w : V3(0.f, 0.f, 0.f);
}
// SmoothConductor::sample, conductor.cpp:248-307 (unpolarized); eval / pdf are zero (delta lobe)
MI_DEV void conductor_sample(const DBsdf &b, v3 wi, bool active, BSample &bs, v3 &weight) {
    active = active && wi.z > 0.f;
    bs.wo = active ? V3(-wi.x, -wi.y, wi.z) : V3(0.f, 0.f, 0.f); bs.eta = active ? 1.f : 0.f; bs.pdf = active ? 1.f : 0.f; bs.delta = active;
    v3 F = V3(fresnel_conductor_(wi.z, b.eta[0], b.k[0]), fresnel_conductor_(wi.z, b.eta[1], b.k[1]), fresnel_conductor_(wi.z, b.eta[2], b.k[2]));
    weight = active ? ld3(b.reflectance) * F : V3(0.f, 0.f, 0.f);
}
template <int FULL> MI_DEV void bsdf_eval_pdf(const DBsdf &b, v3 refl, v3 wi, v3 wo, bool active, v3 &val, float &pdf) {
    if (!FULL || b.type == MI355_BSDF_DIFFUSE) diffuse_eval_pdf(refl, wi, wo, active, val, pdf);
    else if (b.type == MI355_BSDF_ROUGHCONDUCTOR) roughconductor_eval_pdf(b, wi, wo, active, val, pdf);
    else { val = V3(0.f, 0.f, 0.f); pdf = 0.f; }
}
template <int FULL> MI_DEV void bsdf_sample(const DBsdf &b, v3 refl, v3 wi, float s1, float s2x, float s2y, bool active, BSample &bs, v3 &weight,
                                             bool importance = false) {
    if (!FULL || b.type == MI355_BSDF_DIFFUSE) { diffuse_sample(refl, wi, s2x, s2y, active, bs.wo, bs.pdf, weight); bs.eta = 1.f; bs.delta = false; }
    else if (b.type == MI355_BSDF_ROUGHCONDUCTOR) roughconductor_sample(b, wi, s2x, s2y, active, bs, weight);
    else if (b.type == MI355_BSDF_CONDUCTOR) conductor_sample(b, wi, active, bs, weight);
    else dielectric_sample(b, wi, s1, active, bs, weight, importance);
}

struct DS { v3 p, n; float pdf; v3 d; float dist; int32_t emitter; };     // DirectionSample, records.h:93-195

// Scene::sample_emitter_direction (scene.cpp:299-353) -> AreaLight::sample_direction (area.cpp:117-168)
// -> Shape::sample_direction (shape.cpp:358-375) -> Rectangle::sample_position (rectangle.cpp:166-180).
// The shadow ray is returned to the caller so that fused and wavefront modes can trace it their own way.
// Mesh::sample_position (mesh.cpp:749-803): face by area (DiscreteDistribution::sample_reuse, distr_1d.h:104-182), then
// square_to_uniform_triangle (warp.h:153-156)
MI_DEV void mesh_sample_position(const DScene &sc, const DShape &sh, float sx, float sy, v3 &p, v3 &n, float &pdf) {
    const float *cdf = sc.area_cdf + sh.cdf_offset;
    const float sum = cdf[sh.tri_count - 1u], norm = 1.f / sum, target = sy * sum;
    uint32_t lo = 0, hi = sh.tri_count - 1u;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] < target) lo = mid + 1u; else hi = mid; }
    const uint32_t f = lo;
    const float c_prev = f > 0u ? cdf[f - 1u] : 0.f;
    sy = (sy - c_prev * norm) / ((cdf[f] - c_prev) * norm);
    const uint32_t *fi = sc.faces + 3 * (size_t) (sh.tri_offset + f);
    uint32_t i0 = sh.vtx_offset + fi[0], i1 = sh.vtx_offset + fi[1], i2 = sh.vtx_offset + fi[2];
    v3 p0 = ld3(sc.positions + 3 * (size_t) i0), p1 = ld3(sc.positions + 3 * (size_t) i1), p2 = ld3(sc.positions + 3 * (size_t) i2);
    v3 e0 = p1 - p0, e1 = p2 - p0;
    float t = safe_sqrt_(1.f - sx), bx = 1.f - t, by = t * sy;
    p = fma3s(e0, bx, fma3s(e1, by, p0));
    v3 n0 = ld3(sc.normals + 3 * (size_t) i0), n1 = ld3(sc.normals + 3 * (size_t) i1), n2 = ld3(sc.normals + 3 * (size_t) i2);
    v3 nn = (sh.flags & DSHAPE_HAS_NRM) ? fma3s(n0, 1.f - bx - by, fma3s(n1, bx, n2 * by)) : cross3(e0, e1);
    nn = normalize3(nn);
    n = (sh.flags & DSHAPE_FLIP) ? -nn : nn; pdf = norm;
}
template <int FULL>
MI_DEV bool sample_emitter_direction(const DScene &sc, const SI &si, float sx, float sy, bool active, DS &ds, v3 &spec, Ray &shadow) {
    ds.p = ds.n = ds.d = V3(0.f, 0.f, 0.f); ds.pdf = 0.f; ds.dist = 0.f; ds.emitter = -1; spec = V3(0.f, 0.f, 0.f);
    if (sc.n_emitters == 0) return false;
    uint32_t index = 0; float emitter_weight = 1.f;
    if (sc.n_emitters > 1) {                                               // scene.cpp:227-250
        float cf = (float) sc.n_emitters, scaled = sx * cf;
        index = min((uint32_t) scaled, sc.n_emitters - 1u);
        emitter_weight = cf; sx = scaled - (float) index;
    }
    const DEmitter &em = MI_SC_EMITTER(sc, index);
    const DShape &sh = MI_SC_SHAPE(sc, em.shape);
    ds.emitter = (int32_t) index;
    if (FULL && em.type != MI355_EMITTER_AREA) return false;               // DirectionalArea::sample_direction fails by design, directionalarea.cpp:126-130
    if (sh.type == MI355_SHAPE_RECTANGLE || sh.type == MI355_SHAPE_MESH || (FULL && sh.type == MI355_SHAPE_DISK)) {   // Shape::sample_direction, shape.cpp:358-375
        if (sh.type == MI355_SHAPE_MESH) mesh_sample_position(sc, sh, sx, sy, ds.p, ds.n, ds.pdf);
        else if (FULL && sh.type == MI355_SHAPE_DISK) {                    // Disk::sample_position, disk.cpp:173-191
            float qx, qy; disk_concentric(sx, sy, qx, qy);
            ds.p = xf_point(sh.to_world, V3(qx, qy, 0.f)); ds.n = ld3(sh.frame_n); ds.pdf = sh.inv_area;
        }
        else { ds.p = xf_point(sh.to_world, V3(sx * 2.f - 1.f, sy * 2.f - 1.f, 0.f)); ds.n = ld3(sh.frame_n); ds.pdf = sh.inv_area; }
        ds.d = ds.p - si.p;
        float dist2 = sqnorm3(ds.d); ds.dist = sqrt_(dist2);
        ds.d = ds.d * rcp_(ds.dist);
        float dp = fabsf(dot3(ds.d, ds.n));
        float x = div_(dist2, dp);
        ds.pdf *= isfinite(x) ? x : 0.f;
    } else if (FULL && sh.type == MI355_SHAPE_SPHERE) {                   // Sphere::sample_direction, sphere.cpp:236-311
        v3 center = ld3(sh.frame_s); float radius = sh.frame_t[0];
        v3 dc_v = center - si.p; float dc_2 = sqnorm3(dc_v);
        float radius_adj = radius * ((sh.flags & DSHAPE_FLIP) ? (1.f + MI_RAY_EPS) : (1.f - MI_RAY_EPS));
        v3 dn;
        if (dc_2 > radius_adj * radius_adj) {
            float inv_dc = rsqrt_(dc_2), sin_theta_max = radius * inv_dc, sin_theta_max_2 = sin_theta_max * sin_theta_max,
                  inv_sin_theta_max = rcp_(sin_theta_max), cos_theta_max = sqrt_(fmaxf(1.f - sin_theta_max_2, 0.f));
            float t1 = fmaf(cos_theta_max - 1.f, sx, 1.f);
            float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - t1 * t1 : sin_theta_max_2 * sx,
                  cos_theta = sqrt_(fmaxf(1.f - sin_theta_2, 0.f));
            float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * sqrt_(fmaxf(fmaf(-sin_theta_2, inv_sin_theta_max * inv_sin_theta_max, 1.f), 0.f)),
                  sin_alpha = sqrt_(fmaxf(fmaf(-cos_alpha, cos_alpha, 1.f), 0.f));
            float sin_phi, cos_phi; sincos_(sy * (2.f * MI_PI), sin_phi, cos_phi);
            v3 fn = dc_v * (-inv_dc), fs, ft; coordinate_system(fn, fs, ft);
            dn = fma3s(fn, cos_alpha, fma3s(ft, sin_phi * sin_alpha, fs * (cos_phi * sin_alpha)));
            ds.p = fma3s(dn, radius, center); ds.n = dn; ds.d = ds.p - si.p;
            float dist2 = sqnorm3(ds.d); ds.dist = sqrt_(dist2); ds.d = ds.d * rcp_(ds.dist);
            ds.pdf = div_(0.5f * MI_INV_PI, 1.f - cos_theta_max);
            if (ds.dist == 0.f) ds.pdf = 0.f;
        } else {
            float z = fmaf(-2.f, sy, 1.f), r = sqrt_(fmaxf(fmaf(-z, z, 1.f), 0.f)), sn, cs; sincos_(2.f * MI_PI * sx, sn, cs);
            dn = V3(r * cs, r * sn, z);
            ds.p = fma3s(dn, radius, center); ds.n = dn; ds.d = ds.p - si.p;
            float dist2 = sqnorm3(ds.d); ds.dist = sqrt_(dist2); ds.d = ds.d * rcp_(ds.dist);
            ds.pdf = div_(sh.inv_area * dist2, fabsf(dot3(ds.d, ds.n)));
        }
        if (sh.flags & DSHAPE_FLIP) ds.n = -ds.n;
    } else return false;
    active = active && dot3(ds.d, ds.n) < 0.f && ds.pdf != 0.f;            // area.cpp:136
    spec = active ? ld3(em.radiance) * rcp_(ds.pdf) : V3(0.f, 0.f, 0.f);
    if (sc.n_emitters > 1) { ds.pdf *= sc.inv_emitter_count; spec = spec * emitter_weight; }
    active = active && ds.pdf != 0.f;
    if (active) shadow = spawn_ray_to(si.p, si.n, ds.p);
    return active;                                                         // true: a shadow ray must be tested
}
// Scene::pdf_emitter_direction (scene.cpp:355-366) -> area.cpp:170-197 -> shape.cpp:377-388
template <int FULL>
MI_DEV float pdf_emitter_direction(const DScene &sc, int32_t emitter, v3 ref_p, v3 d, v3 n, float dist) {
    const DShape &sh = MI_SC_SHAPE(sc, MI_SC_EMITTER(sc, emitter).shape);
    float dp = dot3(d, n);
    if (!(dp < 0.f)) return 0.f;
    float adp = fabsf(dp), pdf;
    if (FULL && sh.type == MI355_SHAPE_SPHERE) {                           // Sphere::pdf_direction, sphere.cpp:313-325
        float sin_alpha = sh.frame_t[0] * rcp_(norm3(ld3(sh.frame_s) - ref_p)), cos_alpha = sqrt_(fmaxf(1.f - sin_alpha * sin_alpha, 0.f));
        pdf = sin_alpha < 0.99999994f ? div_(0.5f * MI_INV_PI, 1.f - cos_alpha) : div_(sh.inv_area * (dist * dist), adp);
    } else pdf = sh.inv_area * (adp != 0.f ? div_(dist * dist, adp) : 0.f);
    return pdf * sc.inv_emitter_count;
}

// -------------------------------------------------------------------------------------------
// primary rays: integrator.cpp:293-310,419-455; perspective.cpp:239-280
// -------------------------------------------------------------------------------------------
MI_DEV Ray sample_primary_ray(const DScene &sc, float pos_x, float pos_y) {
    float sx = 1.f / (float) sc.crop_w, sy = 1.f / (float) sc.crop_h;
    float ox = -(float) sc.crop_x * sx, oy = -(float) sc.crop_y * sy;
    float ax = fmaf(pos_x, sx, ox), ay = fmaf(pos_y, sy, oy);
    const float *m = sc.sample_to_camera;
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { float v = m[i*4+3]; v = fmaf(m[i*4+0], ax, v); v = fmaf(m[i*4+1], ay, v); v = fmaf(m[i*4+2], 0.f, v); r[i] = v; }
    float iw = rcp_(r[3]);
    v3 d = normalize3(V3(r[0] * iw, r[1] * iw, r[2] * iw));
    const float *tw = sc.cam_to_world;
    Ray ray; ray.d = xf_vector(tw, d);
    float inv_z = rcp_(d.z), near_t = sc.near_clip * inv_z, far_t = sc.far_clip * inv_z;
    ray.o = V3(tw[3], tw[7], tw[11]) + ray.d * near_t;
    ray.maxt = far_t - near_t;
    return ray;
}

// -------------------------------------------------------------------------------------------
// RealisticLensCamera::sample_ray -- src/sensors/realisticlens.cpp:2209-2292 (rear-element sampling :1870-1878),
// trace_ray_from_film :1654-1683, LensInterface::compute_interaction :269-297, SpheroidLens::intersect :359-397,
// PlanoLens::intersect :444-468, ApertureStop::compute_interaction :505-517.
// The interface table is indexed by a wave-uniform loop counter (scalar loads); every lane refracts its own ray.
// -------------------------------------------------------------------------------------------
// AsphericalLens (realisticlens.cpp:553-821), per-ray semantics (the reference's JIT variants branch on dr::none / dr::any over
// the whole wavefront there).  Row: L[6] = kappa, L[7] = c (unitless), L[8..14] = a_i * r_scale^(2i+3).
#define MI_LENS_ROW 16
MI_DEV float asph_conic(const float *L, float r2_) {
    float c = L[7], K = L[6];
    float sqr_term = 1.f - (1.f + K) * (c * c) * r2_;
    return c * r2_ * (1.f / (1.f + sqrtf(sqr_term)));
}
MI_DEV float asph_eval(const float *L, float r2) {
    float er = L[2], r2_ = r2 * (1.f / (er * er)), z_ = 0.f;
#pragma unroll
    for (int i = 6; i >= 0; --i) z_ = fmaf(z_, r2_, L[8 + i]);
    z_ *= r2_ * r2_;
    z_ += asph_conic(L, r2_);
    return -z_ * er + L[3];
}
MI_DEV float asph_grad_(const float *L, float r2_) {
    float r_ = sqrtf(r2_), z_ = 0.f;
#pragma unroll
    for (int i = 6; i >= 0; --i) z_ = fmaf(z_, r2_, (2.f * (float) i + 4.f) * L[8 + i]);
    z_ *= r2_ * r_;
    float cr = L[7] * r_, sqr_term = 1.f - (1.f + L[6]) * (cr * cr);
    z_ += cr * (1.f / sqrtf(sqr_term));
    return -z_;
}
MI_DEV v3 asph_normal(const float *L, v3 p) {
    float inv = 1.f / L[2], rx = p.x * inv, ry = p.y * inv, r2_ = rx * rx + ry * ry;
    if (r2_ >= 4.f * MI_EPS) { float s = 1.f / sqrtf(r2_); rx *= s; ry *= s; } else { rx = 0.f; ry = 0.f; }
    float zg = asph_grad_(L, r2_);
    return normalize3_ieee(V3(zg * rx, zg * ry, -1.f));
}
MI_DEV bool asph_conic_hit(const float *L, v3 o, v3 d, float &t_out) {
    float R = L[2], c = L[7], K = L[6], z0 = L[3];
    v3 on = (o - V3(0.f, 0.f, z0)) * (1.f / R);
    float A = c * (1.f + K * (d.z * d.z)), B = 2.f * (c * (dot3(on, d) + K * on.z * d.z) - d.z), C = c * (dot3(on, on) + K * (on.z * on.z)) - 2.f * on.z;
    float t0, t1; if (!solve_quadratic_ieee(A, B, C, t0, t1)) return false;
    t0 *= R; t1 *= R;
    float z0t = c * (o.z + t0 * d.z - z0), z1t = c * (o.z + t1 * d.z - z0), lim = K > -1.f ? R * (1.f / (1.f + K)) : MI_INF;
    bool v0 = z0t > 0.f && z0t <= lim, v1 = z1t > 0.f && z1t <= lim;
    if (!(v0 || v1)) return false;
    t_out = (v0 && v1) ? (t0 > 0.f ? t0 : t1) : (v0 ? t0 : t1);
    return true;
}
MI_DEV bool asph_intersect(const float *L, v3 o, v3 d, float &t, v3 &p, v3 &n) {
    const float TOL = MI_EPS * 10.f;
    if (!asph_conic_hit(L, o, d, t)) t = (L[3] - o.z) * (1.f / d.z);
    p = fma3s(d, t, o); float r2 = p.x * p.x + p.y * p.y, err = MI_INF;
    for (int itr = 0; itr < 10; ++itr) {
        float z_asph = asph_eval(L, r2);
        v3 plane_p = V3(p.x, p.y, z_asph);
        err = fabsf(p.z - z_asph);
        v3 pn = asph_normal(L, p);
        t = dot3(pn, plane_p - o) / dot3(pn, d);
        p = fma3s(d, t, o); r2 = p.x * p.x + p.y * p.y;
        if (!(err > TOL)) break;
    }
    if (!(err < TOL)) return false;
    n = asph_normal(L, p);
    return t == t && fabsf(t) != MI_INF;
}
MI_DEV bool lens_interact(const float *L, v3 &o, v3 &d) {
    uint32_t kind = (uint32_t) L[0]; float R = L[1], er = L[2], z = L[3], n_left = L[4], n_right = L[5];
    v3 p, n; float t;
    if (kind == 3) {
        if (!asph_intersect(L, o, d, t, p, n)) return false;
    } else if (kind == 0) {
        v3 center = V3(0.f, 0.f, z + R);
        v3 pc = center - o;
        float center_proj = dot3(pc, d);
        float perp = sqrtf(sqnorm3(pc - d * center_proj));                  // (IEEE throughout the lens: see solve_quadratic_ieee)
        float disc = R * R - perp * perp;
        if (!(disc >= 0.f)) return false;
        float sq = sqrtf(disc), near_t = center_proj - sq, far_t = center_proj + sq;
        if (!(far_t >= 0.f)) return false;
        bool convex = R > 0.f;
        t = (convex != (d.z < 0.f)) ? (near_t >= 0.f ? near_t : MI_INF) : far_t;
        if (!(t != MI_INF)) return false;
        p = fma3s(d, t, o);
        n = normalize3_ieee(p - center); if (!convex) n = -n;
    } else {
        n = V3(0.f, 0.f, -1.f);
        float n_dot_d = dot3(n, d);
        if (!(fabsf(n_dot_d) >= MI_EPS)) return false;
        t = (n.z * z - dot3(n, o)) / n_dot_d;
        if (!(t >= 0.f)) return false;
        p = fma3s(d, t, o);
    }
    if (!((p.x * p.x + p.y * p.y) < er * er)) return false;
    if (kind == 2) { o = p; return true; }
    v3 wi = -d;
    float cos_theta_i = dot3(wi, n), r, cos_theta_t, eta_it, eta_ti;
    fresnel_(cos_theta_i, n_right / n_left, r, cos_theta_t, eta_it, eta_ti);
    if (!(r <= 0.99999994f)) return false;
    float k = fmaf(dot3(wi, n), eta_ti, cos_theta_t);                       // refract(wi, m, cos_theta_t, eta_ti), fresnel.h:309-313
    d = V3(fmaf(n.x, k, -(wi.x * eta_ti)), fmaf(n.y, k, -(wi.y * eta_ti)), fmaf(n.z, k, -(wi.z * eta_ti)));
    o = p;
    return true;
}
MI_DEV float sample_lens_ray(const DScene &sc, float pos_x, float pos_y, float ap_x, float ap_y, Ray &ray) {
    float sx = 1.f / (float) sc.crop_w, sy = 1.f / (float) sc.crop_h;
    float ax = fmaf(pos_x, sx, -(float) sc.crop_x * sx), ay = fmaf(pos_y, sy, -(float) sc.crop_y * sy);
    float rsx = (float) sc.crop_w / (float) sc.film_w, rsy = (float) sc.crop_h / (float) sc.film_h;
    float rox = (float) sc.crop_x / (float) sc.film_w, roy = (float) sc.crop_y / (float) sc.film_h;
    v3 film_p = V3(sc.film_px * (2.f * fmaf(ax, rsx, rox) - 1.f), sc.film_py * (2.f * fmaf(ay, rsy, roy) - 1.f), 0.f);
    float dkx, dky; disk_concentric(ap_x, ap_y, dkx, dky);
    float bounds_area = 0.f;
    v3 pupil_p = V3(0.f, 0.f, 0.f);
    if (sc.sample_exit_pupil) {                                             // sample_exit_pupil, realisticlens.cpp:1808-1836
        float r_film = sqrtf(film_p.x * film_p.x + film_p.y * film_p.y);
        uint32_t r_idx = (uint32_t) (int32_t) floorf(r_film / sc.film_diagonal * 64.f);
        if (r_idx > 63u) r_idx = 63u;
        const float *b = sc.pupil + 4 * r_idx;
        float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        bounds_area = (b2 - b0) * (b3 - b1);
        float ppx = fmaf(b2, ap_x, fmaf(-b0, ap_x, b0)), ppy = fmaf(b3, ap_y, fmaf(-b1, ap_y, b1));      // dr::lerp
        float sin_t = r_film > 0.f ? film_p.y * (1.f / r_film) : 0.f, cos_t = r_film > 0.f ? film_p.x * (1.f / r_film) : 1.f;
        pupil_p = V3(cos_t * ppx - sin_t * ppy, sin_t * ppx + cos_t * ppy, sc.lens[3]);
    }
    v3 aperture_p = sc.sample_exit_pupil ? pupil_p : V3(sc.lens[2] * dkx, sc.lens[2] * dky, sc.lens[3]);   // sample_rear_element, :1870-1878
    v3 o = film_p, d = normalize3_ieee(aperture_p - film_p);
    bool active = true;
    for (uint32_t i = 0; i < sc.lens_count; ++i) { v3 o2 = o, d2 = d; bool ok = lens_interact(sc.lens + MI_LENS_ROW * i, o2, d2); if (active && ok) { o = o2; d = d2; } active = active && ok; }
    const float *tw = sc.cam_to_world;
    ray.o = xf_point(tw, o); ray.d = xf_vector(tw, d);
    float inv_z = 1.f / d.z, near_t = sc.near_clip * inv_z, far_t = sc.far_clip * inv_z;
    ray.o = ray.o + ray.d * near_t; ray.maxt = far_t - near_t;
    float c2 = d.z * d.z;
    if (sc.sample_exit_pupil)                                               // :2282-2285
        return active ? sc.shutter_open_time * bounds_area * (1.f / (sc.lens[3] * sc.lens[3])) : 0.f;
    return active ? c2 * c2 : 0.f;
}
// RadicalInverse::eval (include/mitsuba/core/qmc.h:67-92), recip = 1.f / base (qmc.cpp:50)
MI_DEV float radical_inverse(uint32_t base, uint32_t index) {
    uint32_t value = 0; float factor = 1.f, recip = 1.f / (float) base;
    while (index != 0u) {
        uint32_t next = index / base;
        factor = factor * recip;
        value = (value - next) * base + index;
        index = next;
    }
    return fminf(0.99999994f, (float) value * factor);
}
// render_sample head (integrator.cpp:433-455): pixel jitter, aperture sample when the sensor needs one, ray + ray weight
template <int FULL>
MI_DEV float camera_ray(const DScene &sc, Pcg32 &rng, int32_t px, int32_t py, float &pos_x, float &pos_y, Ray &ray) {
    pos_x = (float) px + next_1d(rng); pos_y = (float) py + next_1d(rng);                  // integrator.cpp:435
    if (FULL && sc.sensor_type == MI355_SENSOR_REALISTICLENS) {
        float ap_x = next_1d(rng), ap_y = next_1d(rng);                                    // needs_aperture_sample, :438-440
        if (sc.shutter_open_time > 0.f) (void) next_1d(rng);                               // time sample, :442-444
        return sample_lens_ray(sc, pos_x, pos_y, ap_x, ap_y, ray);
    }
    if (sc.shutter_open_time > 0.f) (void) next_1d(rng);                                   // time sample, :442-444
    ray = sample_primary_ray(sc, pos_x, pos_y); return 1.f;
}

// MI_PROF (diagnostic builds only; printed by mi355_get_counters, tools/occupancy_ab.sh builds it): per-wave s_memtime deltas per section of the
// path loop.  The hooks serialise the kernel (78 ms instead of 9: profiles/r02_occupancy_ab.txt) -- section SHARES of the instrumented kernel only.
// Event counters are WAVE totals kept in SGPRs (ballot + s_bcnt1): no VGPR and no VALU instruction is spent on them.
MI_DEV uint32_t wave_count(bool pred) { return (uint32_t) __popcll(__ballot(pred)); }
#ifdef MI_PROF
struct Counters3 { uint32_t isect, shade, shadow; uint32_t prof[8]; unsigned long long t; };
static __device__ unsigned long long g_prof[8];
#define PROF_START(c) { __builtin_amdgcn_sched_barrier(0); (c).t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define PROF(c, k) { __builtin_amdgcn_sched_barrier(0); unsigned long long n_ = __builtin_amdgcn_s_memtime(); (c).prof[k] += (uint32_t) (n_ - (c).t); (c).t = n_; __builtin_amdgcn_sched_barrier(0); }
#else
struct Counters3 { uint32_t isect, shade, shadow; };
#define PROF_START(c)
#define PROF(c, k)
#endif

// Every kernel of this engine takes the DScene BY VALUE AS ITS FIRST ARGUMENT, i.e. it sits at offset 0 of the kernel-argument segment.
// MI_SCENE_RELOAD(name) re-derives a reference to it through a pointer the optimiser cannot see through.  Placed at the top of a bounce
// loop's body it stops the compiler from fetching every scene constant once, before the loop, and carrying ~150 of them through it in
// SGPRs it does not have: the spills go through VGPR lanes (v_writelane / v_readlane) and each costs a VALU issue slot, the scarce
// resource of these kernels.  Re-fetching is a scalar load that hits the constant cache.  Measured on the forward kernel: 151 -> 30 spill
// moves, 9.81 -> 9.08 ms (Cornell box, 512^2 x 256 spp).
#define MI_SCENE_RELOAD(name) \
    const __attribute__((address_space(4))) DScene *name##_k = (const __attribute__((address_space(4))) DScene *) __builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(name##_k)); \
    const DScene &name = *(const DScene *) name##_k

// -------------------------------------------------------------------------------------------
// PathIntegrator::sample, src/integrators/path.cpp:95-287
// -------------------------------------------------------------------------------------------
// Loop-carried state of PathIntegrator::sample (path.cpp:108-121).  For the lanes that are still alive the loop counter
// IS the path depth (depth is only incremented on valid hits, path.cpp:266, and an invalid hit ends the path), and
// prev_bsdf_delta is true exactly at the first vertex -- so neither needs a register.  eta stays 1 (no refraction yet).
// With FULL = true (scenes with glossy / specular BSDFs) eta and prev_bsdf_delta become real state.
struct PathState { Ray ray; v3 throughput, result, prev_p; float prev_bsdf_pdf; Pcg32 rng; float eta; bool prev_delta; };

MI_DEV void path_init(PathState &st, const Ray &ray, const Pcg32 &rng) {
    st.ray = ray; st.throughput = V3(1.f, 1.f, 1.f); st.result = V3(0.f, 0.f, 0.f); st.prev_p = V3(0.f, 0.f, 0.f);
    st.prev_bsdf_pdf = 1.f; st.rng = rng; st.eta = 1.f; st.prev_delta = true;
}
// One iteration of the loop of path.cpp:142-281 at depth `it`.  Returns whether the lane continues; `hit_any` reports
// a valid surface hit (feeds valid_ray, path.cpp:256).
template <int FULL>
MI_DEV bool path_step(const DScene &sc, PathState &st, uint32_t it, uint32_t max_depth, uint32_t rr_depth,
                      bool active, uint32_t *stk, bool &hit_any, Counters3 &cnt) {
    PROF_START(cnt);
    Hit hit = scene_intersect<FULL>(sc, st.ray, active, stk);
    PROF(cnt, 0);
    cnt.isect += wave_count(active);
    bool si_valid = active && hit.t != MI_INF;
    cnt.shade += wave_count(si_valid);                                      // (counted here: ballot(true) inside the branch gets hoisted)
    bool active_next = si_valid && (it + 1 < max_depth);                    // path.cpp:177
    bool alive = false, shadow_req = false;      // counters are only touched where control flow is wave-uniform (they live in SGPRs)
    hit_any = si_valid;
    if (si_valid) {
        SI si = compute_si<FULL>(sc, st.ray, hit);
        const DShape &sh = MI_SC_SHAPE(sc, hit.shape);
        int32_t emitter = sh.emitter;
        if (emitter >= 0) {                                                // path.cpp:158-174
            v3 rel = si.p - st.prev_p; float dist = norm3(rel); v3 d = rel * rcp_(dist);   // records.h:173-180
            bool prev_delta = FULL ? st.prev_delta : (it == 0);
            float em_pdf = prev_delta ? 0.f : pdf_emitter_direction<FULL>(sc, emitter, st.prev_p, d, si.sh_n, dist);
            float mis_bsdf = mis_weight(st.prev_bsdf_pdf, em_pdf);
            v3 le = (si.wi.z > 0.f && st.prev_bsdf_pdf > 0.f) ? ld3(MI_SC_EMITTER(sc, emitter).radiance) : V3(0.f, 0.f, 0.f);  // area.cpp:83-90
            st.result = fma3(st.throughput, le * mis_bsdf, st.result);
        }
        PROF(cnt, 1);
        // At the last vertex (it + 1 == max_depth, uniform over the wave) nothing below can change the result:
        // active_next is false for every lane, so no NEE contribution, no continuation (path.cpp:177-187).
        if (it + 1 < max_depth) {
            const DBsdf &bsdf = MI_SC_BSDF(sc, sh.bsdf);
            v3 refl = bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance);
            // ---- emitter sampling (path.cpp:186-208)
            float ex = next_1d(st.rng), ey = next_1d(st.rng);
            DS ds; v3 em_weight; Ray shadow; shadow.o = shadow.d = V3(0.f, 0.f, 1.f); shadow.maxt = 0.f;
            bool smooth = !FULL || (bsdf.type != MI355_BSDF_DIELECTRIC && bsdf.type != MI355_BSDF_CONDUCTOR);        // BSDFFlags::Smooth, path.cpp:187
            bool need_shadow = sample_emitter_direction<FULL>(sc, si, ex, ey, active_next && smooth, ds, em_weight, shadow);
            PROF(cnt, 2);
            // ---- BSDF eval + sample (path.cpp:212-216, bsdf.cpp:22-33)
            float s1 = next_1d(st.rng), s2x = next_1d(st.rng), s2y = next_1d(st.rng);
            v3 wo = si_to_local(si, ds.d);
            v3 bsdf_val; float bsdf_pdf; bsdf_eval_pdf<FULL>(bsdf, refl, si.wi, wo, true, bsdf_val, bsdf_pdf);
            BSample bsm; v3 bsdf_weight; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, true, bsm, bsdf_weight);
            v3 bs_wo = bsm.wo; float bs_pdf = bsm.pdf;
            v3 contrib = (bsdf_val * em_weight) * mis_weight(ds.pdf, bsdf_pdf);
            PROF(cnt, 3);
            bool occluded = scene_ray_test<FULL>(sc, shadow, need_shadow, stk);  // scene.cpp:341-346
            PROF(cnt, 4);
            shadow_req = need_shadow;
            if (need_shadow && !occluded) st.result = fma3(st.throughput, contrib, st.result);   // path.cpp:220-230
            // ---- BSDF sampling + loop update (path.cpp:234-280)
            st.ray = spawn_ray(si.p, si.n, si_to_world(si, bs_wo));
            st.throughput = st.throughput * bsdf_weight;
            st.prev_p = si.p; st.prev_bsdf_pdf = bs_pdf;
            if (FULL) { st.eta *= bsm.eta; st.prev_delta = bsm.delta; }
            const float eta = FULL ? st.eta : 1.f;
            float tmax = max3(st.throughput);
            float rr_prob = fminf(tmax * (eta * eta), .95f);
            bool rr_active = it + 1 >= rr_depth, rr_continue = next_1d(st.rng) < rr_prob;
            if (rr_active) st.throughput = st.throughput * rcp_(rr_prob);
            alive = active_next && (!rr_active || rr_continue) && tmax != 0.f;
            PROF(cnt, 5);
        }
    }
    cnt.shadow += wave_count(shadow_req);
    return alive;
}
// `keep_stream`: the sampler stream is continued by a further pass (samples_per_pass), so it must end where the reference's does.  Every
// sampler call inside the loop of path.cpp:142-281 is UNMASKED and the loop masks by `active` alone, so a lane draws its six numbers
// (next_2d, next_1d, next_2d, next_1d: :196, :212-213, :272) in EVERY iteration it enters -- also in the one where its ray leaves the
// scene or reaches max_depth, where path_step (rightly, for a single pass) draws nothing.
template <int FULL>
MI_DEV void path_sample(const DScene &sc, Pcg32 &rng, Ray ray, uint32_t max_depth, uint32_t rr_depth,
                        bool active, uint32_t *stk, v3 &result_out, bool &valid_out, Counters3 &cnt, bool keep_stream = false) {
    PathState st; path_init(st, ray, rng);
    bool valid_ray = false;                                                // no environment emitter, path.cpp:115
    for (uint32_t it = 0; it < max_depth && __any(active); ++it) {
        MI_SCENE_RELOAD(scb); const DScene &sc = scb;                       // scene constants are re-fetched per bounce, not carried in spilled SGPRs
        bool hit_any; const bool entered = active;
        active = path_step<FULL>(sc, st, it, max_depth, rr_depth, active, stk, hit_any, cnt);
        if (keep_stream && entered && !(hit_any && it + 1 < max_depth)) for (int k = 0; k < 6; ++k) (void) next_1d(st.rng);
        valid_ray = valid_ray || hit_any;
    }
    rng = st.rng;
    result_out = valid_ray ? st.result : V3(0.f, 0.f, 0.f); valid_out = valid_ray;
}

// -------------------------------------------------------------------------------------------
// PRBIntegrator.sample, src/python/python/ad/integrators/prb.py:59-257
//   BACKWARD = false: primal estimator, returns L
//   BACKWARD = true : replays the same RNG stream with L = state_in and accumulates
//                     dL * d(Le + Lr_dir + Lr_ind)/d(theta) (closed form of prb.py:202-247)
// Gradient sinks: `s_grad` = per-workgroup LDS accumulators for the "small" parameters
// (constant reflectances, emitter radiances), `g_grad` = the global gradient buffer (textures).
// -------------------------------------------------------------------------------------------
// `q` = per-workgroup LDS queue of texture-gradient records (float index of the texel's first channel, d/dr, d/dg, d/db),
// flushed once at the end of the kernel: keeps ~3000-cycle float atomics out of the bounce loop, where every later
// vector load would have to wait for them (vmcnt retires in order).
// One record (8 dwords) per textured vertex: texture id, texel ids of the (0,0) and (1,1) taps, the two lerp weights and
// dLoss/d(albedo rgb); the other two taps and the 12 (tap, channel) atomics are expanded at flush time.
struct GradSink { float *s_grad; float *g_grad; float *q; uint32_t *qn; uint32_t qcap; const uint32_t *small_map; };

// FWD (with BACKWARD): prb.py in dr.ADMode.Forward, driven by RBIntegrator.render_forward (common.py:498-629).  The same second walk
// with L = state_in, but every d(Le + Lr_dir + Lr_ind)/d(theta) is contracted with the parameter TANGENT (sink.g_grad, laid out like the
// gradient buffer; sink.small_map resolves the constant parameters) and summed into the lane's dL, returned in L_out.
template <bool BACKWARD, int FULL, bool FWD = false>
MI_DEV void prb_sample(const DScene &sc, Pcg32 &rng, Ray ray, uint32_t max_depth, uint32_t rr_depth, bool hide_emitters,
                       bool active, uint32_t *stk, v3 dL, v3 state_in, GradSink sink,
                       v3 &L_out, bool &valid_out, Counters3 &cnt) {
    uint32_t depth = 0; v3 L = BACKWARD ? state_in : V3(0.f, 0.f, 0.f); v3 beta = V3(1.f, 1.f, 1.f); float eta = 1.f;
    v3 prev_p = V3(0.f, 0.f, 0.f); float prev_bsdf_pdf = 1.f; bool prev_bsdf_delta = true;
    v3 fwd_acc = V3(0.f, 0.f, 0.f);                                          // FWD: the lane's dL
    for (uint32_t it = 0; it < max_depth && __any(active); ++it) {
        MI_SCENE_RELOAD(scb); const DScene &sc = scb;                       // scene constants are re-fetched per bounce, not carried in spilled SGPRs
        Hit hit = scene_intersect<FULL>(sc, ray, active, stk);
        cnt.isect += wave_count(active);
        bool si_valid = active && hit.t != MI_INF;
        cnt.shade += wave_count(si_valid);                                  // (counted here: ballot(true) inside the branch gets hoisted)
        bool active_next = si_valid && !(hide_emitters && depth == 0 && !si_valid);
        active = false;
        bool shadow_req = false;                 // counted after the divergent region: counters live in SGPRs
        if (si_valid) {
                SI si = compute_si<FULL>(sc, ray, hit);
            const DShape &sh = sc.shapes[hit.shape];
            int32_t emitter = sh.emitter;
            v3 Le = V3(0.f, 0.f, 0.f); float le_mis = 0.f; bool le_front = false;
            if (emitter >= 0) {                                            // prb.py:126-134
                v3 rel = si.p - prev_p; float dist = norm3(rel); v3 d = rel * rcp_(dist);
                float em_pdf = prev_bsdf_delta ? 0.f : pdf_emitter_direction<FULL>(sc, emitter, prev_p, d, si.sh_n, dist);
                le_mis = mis_weight(prev_bsdf_pdf, em_pdf);
                le_front = si.wi.z > 0.f;
                v3 e = le_front ? ld3(sc.emitters[emitter].radiance) : V3(0.f, 0.f, 0.f);
                Le = (beta * le_mis) * e;
            }
            active_next = active_next && (depth + 1 < max_depth);          // prb.py:139
            if (!(it + 1 < max_depth)) {
                // last vertex (uniform over the wave): active_next is false for every lane, so Lr_dir = 0, the BSDF sample is
                // unused and bsdf.eval(..., active_next) = 0 -- only Le and its emitter gradient remain (prb.py:134,173,230).
                L = BACKWARD ? L - Le : L + Le;
                if (FWD) {
                    if (emitter >= 0 && le_front && sc.emitters[emitter].grad_slot >= 0) {
                        const uint32_t *sm = sink.small_map + sc.emitters[emitter].grad_slot;
                        fwd_acc = fwd_acc + (beta * le_mis) * V3(sink.g_grad[sm[0]], sink.g_grad[sm[1]], sink.g_grad[sm[2]]);
                    }
                } else
                if (BACKWARD && emitter >= 0 && le_front && sc.emitters[emitter].grad_slot >= 0) {
                    float *gb = sink.s_grad + sc.emitters[emitter].grad_slot;
                    atomicAdd(gb + 0, dL.x * beta.x * le_mis); atomicAdd(gb + 1, dL.y * beta.y * le_mis); atomicAdd(gb + 2, dL.z * beta.z * le_mis);
                }
                depth += 1;
                continue;
            }
            const DBsdf &bsdf = sc.bsdfs[sh.bsdf];
            TexTaps taps; v3 refl;
            if (bsdf.texture >= 0) refl = texture_eval(sc, sc.textures[bsdf.texture], si.uv, BACKWARD ? &taps : nullptr);
            else refl = ld3(bsdf.reflectance);
            // ---- emitter sampling (prb.py:142-162)
            float ex = next_1d(rng), ey = next_1d(rng);
            DS ds; v3 em_weight; Ray shadow; shadow.o = shadow.d = V3(0.f, 0.f, 1.f); shadow.maxt = 0.f;
            bool smooth = !FULL || (bsdf.type != MI355_BSDF_DIELECTRIC && bsdf.type != MI355_BSDF_CONDUCTOR);
            bool need_shadow = sample_emitter_direction<FULL>(sc, si, ex, ey, active_next && smooth, ds, em_weight, shadow);
            bool occluded = scene_ray_test<FULL>(sc, shadow, need_shadow, stk);
            shadow_req = need_shadow;
            bool active_em = need_shadow && !occluded;
            if (!active_em) { ds.pdf = 0.f; em_weight = V3(0.f, 0.f, 0.f); }
            v3 wo = si_to_local(si, ds.d);
            v3 bsdf_value_em; float bsdf_pdf_em; bsdf_eval_pdf<FULL>(bsdf, refl, si.wi, wo, active_em, bsdf_value_em, bsdf_pdf_em);
            const bool is_diffuse = !FULL || bsdf.type == MI355_BSDF_DIFFUSE;
            float mis_em = mis_weight(ds.pdf, bsdf_pdf_em);
            v3 Lr_dir = ((beta * mis_em) * bsdf_value_em) * em_weight;
            // ---- detached BSDF sampling (prb.py:166-169)
            float s1 = next_1d(rng), s2x = next_1d(rng), s2y = next_1d(rng);
            BSample bsm; v3 bsdf_weight; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, active_next, bsm, bsdf_weight);
            v3 bs_wo = bsm.wo; float bs_pdf = bsm.pdf;
            // ---- update (prb.py:173-182)
            v3 beta_pre = beta;
            L = BACKWARD ? (L - Le) - Lr_dir : (L + Le) + Lr_dir;
            ray = spawn_ray(si.p, si.n, si_to_world(si, bs_wo));
            beta = beta * bsdf_weight;
            prev_p = si.p; prev_bsdf_pdf = bs_pdf; prev_bsdf_delta = FULL ? bsm.delta : false;
            if (FULL) eta *= bsm.eta;
            // ---- stopping criterion (prb.py:186-198)
            float bmax = max3(beta);
            active_next = active_next && bmax != 0.f;
            float rr_prob = fminf(bmax * (eta * eta), .95f);
            bool rr_active = depth >= rr_depth;
            if (rr_active) beta = beta * rcp_(rr_prob);
            bool rr_continue = next_1d(rng) < rr_prob;
            active_next = active_next && (!rr_active || rr_continue);
            if (BACKWARD) {                                                // prb.py:202-247
                float co = bs_wo.z; bool ev_active = active_next && si.wi.z > 0.f && co > 0.f && is_diffuse;
                v3 det = bsdf_weight * bs_pdf;
                v3 inv = V3(det.x != 0.f ? rcp_(det.x) : 0.f, det.y != 0.f ? rcp_(det.y) : 0.f, det.z != 0.f ? rcp_(det.z) : 0.f);
                float dval = ev_active ? MI_INV_PI * co : 0.f;
                v3 dLo = (L * inv) * dval;                                 // d Lr_ind / d albedo
                bool em_act = active_em && si.wi.z > 0.f && wo.z > 0.f && is_diffuse;
                float dval_em = em_act ? MI_INV_PI * wo.z : 0.f;
                dLo = dLo + ((beta_pre * mis_em) * dval_em) * em_weight;   // d Lr_dir / d albedo
                v3 ga = dL * dLo;
                if (FWD) {
                    v3 dalb = V3(0.f, 0.f, 0.f);
                    if (is_diffuse && bsdf.texture >= 0) {
                        const DTexture &tx = sc.textures[bsdf.texture];
                        const float *tb = sink.g_grad + tx.grad_offset;
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (tx.grad_slot >= 0 && taps.idx[k] >= 0) {
                            if (tx.c == 3) { dalb.x = fmaf(taps.w[k], tb[taps.idx[k] * 3 + 0], dalb.x); dalb.y = fmaf(taps.w[k], tb[taps.idx[k] * 3 + 1], dalb.y);
                                             dalb.z = fmaf(taps.w[k], tb[taps.idx[k] * 3 + 2], dalb.z); }
                            else { float v = taps.w[k] * tb[taps.idx[k]]; dalb = dalb + V3(v, v, v); }
                        }
                    } else if (is_diffuse && bsdf.grad_slot >= 0) {
                        const uint32_t *sm = sink.small_map + bsdf.grad_slot;
                        dalb = V3(sink.g_grad[sm[0]], sink.g_grad[sm[1]], sink.g_grad[sm[2]]);
                    }
                    if (is_diffuse) fwd_acc = fwd_acc + dLo * dalb;      // (other lobes carry no parameter; their inv * 0 may be NaN)
                    if (emitter >= 0 && le_front && sc.emitters[emitter].grad_slot >= 0) {          // d Le / d radiance
                        const uint32_t *sm = sink.small_map + sc.emitters[emitter].grad_slot;
                        fwd_acc = fwd_acc + (beta_pre * le_mis) * V3(sink.g_grad[sm[0]], sink.g_grad[sm[1]], sink.g_grad[sm[2]]);
                    }
                    if (active_em && sc.emitters[ds.emitter].grad_slot >= 0) {                       // d Lr_dir / d radiance
                        const uint32_t *sm = sink.small_map + sc.emitters[ds.emitter].grad_slot;
                        v3 w = ((beta_pre * mis_em) * bsdf_value_em) * rcp_(ds.pdf);
                        fwd_acc = fwd_acc + w * V3(sink.g_grad[sm[0]], sink.g_grad[sm[1]], sink.g_grad[sm[2]]);
                    }
                } else {
                bool textured = is_diffuse && bsdf.texture >= 0 && sc.textures[bsdf.texture < 0 ? 0 : bsdf.texture].grad_slot >= 0 && sink.g_grad != nullptr;
                // (nearest-filter lookups have one tap and take the direct path; a slot that is allocated must be written)
                bool queued = textured && sc.textures[bsdf.texture].c == 3 && sink.q != nullptr && taps.idx[3] >= 0;
                {   // wave-aggregated slot allocation: one LDS atomic per wave, one record per textured lane
                    unsigned long long m = __ballot(queued);
                    if (m) {
                        uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                        uint32_t base = 0;
                        if (rank == 0 && queued) base = atomicAdd(sink.qn, (uint32_t) __popcll(m));
                        base = (uint32_t) __shfl((int) base, __ffsll((long long) m) - 1);
                        uint32_t slot = base + rank;
                        if (queued && slot < sink.qcap) {
                            float *r = sink.q + 8u * slot;
                            float w1x = taps.w[1] + taps.w[3], w1y = taps.w[2] + taps.w[3];     // w = (w0y|w1y) * (w0x|w1x), sums to 1
                            r[0] = __uint_as_float((uint32_t) bsdf.texture); r[1] = __uint_as_float((uint32_t) taps.idx[0]);
                            r[2] = __uint_as_float((uint32_t) taps.idx[3]); r[3] = w1x; r[4] = w1y; r[5] = ga.x; r[6] = ga.y; r[7] = ga.z;
                            textured = false;                              // handled
                        }
                    }
                }
                if (textured) {                                            // queue full, or a 1-channel texture: direct atomics
                    const DTexture &tx = sc.textures[bsdf.texture];
                    float *gb = sink.g_grad + tx.grad_offset;
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (taps.idx[k] >= 0 && taps.w[k] != 0.f) {
                        if (tx.c == 3) { atomicAdd(gb + taps.idx[k] * 3 + 0, taps.w[k] * ga.x); atomicAdd(gb + taps.idx[k] * 3 + 1, taps.w[k] * ga.y);
                                         atomicAdd(gb + taps.idx[k] * 3 + 2, taps.w[k] * ga.z); }
                        else atomicAdd(gb + taps.idx[k], taps.w[k] * (ga.x + ga.y + ga.z));
                    }
                }
                if (bsdf.texture >= 0 || !is_diffuse) {
                } else if (bsdf.grad_slot >= 0) {
                    float *gb = sink.s_grad + bsdf.grad_slot;
                    atomicAdd(gb + 0, ga.x); atomicAdd(gb + 1, ga.y); atomicAdd(gb + 2, ga.z);
                }
                if (emitter >= 0 && le_front && sc.emitters[emitter].grad_slot >= 0) {          // d Le / d radiance
                    float *gb = sink.s_grad + sc.emitters[emitter].grad_slot;
                    atomicAdd(gb + 0, dL.x * beta_pre.x * le_mis); atomicAdd(gb + 1, dL.y * beta_pre.y * le_mis);
                    atomicAdd(gb + 2, dL.z * beta_pre.z * le_mis);
                }
                if (active_em && sc.emitters[ds.emitter].grad_slot >= 0) {                       // d Lr_dir / d radiance
                    float *gb = sink.s_grad + sc.emitters[ds.emitter].grad_slot;
                    v3 w = ((beta_pre * mis_em) * bsdf_value_em) * rcp_(ds.pdf);
                    atomicAdd(gb + 0, dL.x * w.x); atomicAdd(gb + 1, dL.y * w.y); atomicAdd(gb + 2, dL.z * w.z);
                }
                }
            }
            depth += 1;                                                    // prb.py:249
            active = active_next;
        }
        cnt.shadow += wave_count(shadow_req);
    }
    L_out = FWD ? fwd_acc : L; valid_out = depth != 0;
}

// One bounce of PRBIntegrator.sample in PRIMAL mode (prb.py:59-257, same arithmetic as prb_sample<false>) on the path
// integrator's state record (throughput = beta, result = L), so that the re-packing kernel serves both integrators.
// For a live path the loop counter is its depth (prb.py:249).
template <int FULL>
MI_DEV bool prb_primal_step(const DScene &sc, PathState &st, uint32_t it, uint32_t max_depth, uint32_t rr_depth,
                            bool active, uint32_t *stk, bool &hit_any, Counters3 &cnt) {
    Hit hit = scene_intersect<FULL>(sc, st.ray, active, stk);
    cnt.isect += wave_count(active);
    bool si_valid = active && hit.t != MI_INF;
    cnt.shade += wave_count(si_valid);
    hit_any = si_valid;
    bool alive = false, shadow_req = false;
    if (si_valid) {
        SI si = compute_si<FULL>(sc, st.ray, hit);
        const DShape &sh = sc.shapes[hit.shape];
        int32_t emitter = sh.emitter;
        v3 Le = V3(0.f, 0.f, 0.f);
        if (emitter >= 0) {                                                // prb.py:126-134
            v3 rel = si.p - st.prev_p; float dist = norm3(rel); v3 d = rel * rcp_(dist);
            bool prev_delta = FULL ? st.prev_delta : (it == 0);
            float em_pdf = prev_delta ? 0.f : pdf_emitter_direction<FULL>(sc, emitter, st.prev_p, d, si.sh_n, dist);
            float le_mis = mis_weight(st.prev_bsdf_pdf, em_pdf);
            v3 e = si.wi.z > 0.f ? ld3(sc.emitters[emitter].radiance) : V3(0.f, 0.f, 0.f);
            Le = (st.throughput * le_mis) * e;
        }
        bool active_next = it + 1 < max_depth;                             // prb.py:139
        if (!active_next) st.result = st.result + Le;                      // last vertex: only Le remains (see prb_sample)
        else {
            const DBsdf &bsdf = sc.bsdfs[sh.bsdf];
            v3 refl = bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance);
            // ---- emitter sampling (prb.py:142-162)
            float ex = next_1d(st.rng), ey = next_1d(st.rng);
            DS ds; v3 em_weight; Ray shadow; shadow.o = shadow.d = V3(0.f, 0.f, 1.f); shadow.maxt = 0.f;
            bool smooth = !FULL || (bsdf.type != MI355_BSDF_DIELECTRIC && bsdf.type != MI355_BSDF_CONDUCTOR);
            bool need_shadow = sample_emitter_direction<FULL>(sc, si, ex, ey, active_next && smooth, ds, em_weight, shadow);
            bool occluded = scene_ray_test<FULL>(sc, shadow, need_shadow, stk);
            shadow_req = need_shadow;
            bool active_em = need_shadow && !occluded;
            if (!active_em) { ds.pdf = 0.f; em_weight = V3(0.f, 0.f, 0.f); }
            v3 wo = si_to_local(si, ds.d);
            v3 bsdf_value_em; float bsdf_pdf_em; bsdf_eval_pdf<FULL>(bsdf, refl, si.wi, wo, active_em, bsdf_value_em, bsdf_pdf_em);
            float mis_em = mis_weight(ds.pdf, bsdf_pdf_em);
            v3 Lr_dir = ((st.throughput * mis_em) * bsdf_value_em) * em_weight;
            // ---- detached BSDF sampling (prb.py:166-169)
            float s1 = next_1d(st.rng), s2x = next_1d(st.rng), s2y = next_1d(st.rng);
            BSample bsm; v3 bsdf_weight; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, active_next, bsm, bsdf_weight);
            // ---- update (prb.py:173-182)
            st.result = (st.result + Le) + Lr_dir;
            st.ray = spawn_ray(si.p, si.n, si_to_world(si, bsm.wo));
            st.throughput = st.throughput * bsdf_weight;
            st.prev_p = si.p; st.prev_bsdf_pdf = bsm.pdf;
            if (FULL) { st.eta *= bsm.eta; st.prev_delta = bsm.delta; }
            // ---- stopping criterion (prb.py:186-198)
            const float eta = FULL ? st.eta : 1.f;
            float bmax = max3(st.throughput);
            active_next = active_next && bmax != 0.f;
            float rr_prob = fminf(bmax * (eta * eta), .95f);
            bool rr_active = it >= rr_depth;
            if (rr_active) st.throughput = st.throughput * rcp_(rr_prob);
            bool rr_continue = next_1d(st.rng) < rr_prob;
            alive = active_next && (!rr_active || rr_continue);
        }
    }
    cnt.shadow += wave_count(shadow_req);
    return alive;
}

// -------------------------------------------------------------------------------------------
// ONE-PASS adjoint.  prb.py needs the primal pass only for L_total: the adjoint walks the path again with
// L_k = L_total - P_k (P_k = everything collected up to and including vertex k) and every albedo gradient is LINEAR in it:
//     ga_k = dL * ((L_k * inv_k) * dval_k + direct_k) = L_total * A'_k + C'_k,
//     A'_k = dL * (inv_k * dval_k),   C'_k = dL * (direct_k - (P_k * inv_k) * dval_k).
// So one walk suffices: each vertex queues (target, A', C') in LDS, the lane publishes L_total = P_last when its path ends,
// and the workgroup resolves the records when all of its paths are done.  Emitter-radiance gradients do not depend on L
// and are accumulated at once, exactly as in the two-pass kernel.  Records live in 16-byte granules of one LDS queue that
// is filled from both ends (one 32-bit counter: low half = granules used at the bottom, high half = records at the top),
// so the resolver walks two dense lists:
//   constant reflectance, from the bottom : 1 DWORD (bsdf << 8 | lane); C' goes to s_grad immediately and, for a diffuse
//                                           BSDF, inv * dval = 1 / (albedo * pdf) * pdf is re-formed at resolve time as 1 / albedo
//   texture, from the top                 : 3 granules (tex << 8 | lane, texel(0,0), texel(1,1), w1x) (w1y, A''rgb) (x1 - x0, C'rgb)
// with A'' = inv * dval (dL is folded into what the lane publishes: L_total * dL).
// A workgroup whose queue overflows raises `overflow`, publishes nothing and is redone by the two-pass kernel, so the result
// never depends on the queue size.
// -------------------------------------------------------------------------------------------
// s_grad holds 2^copy_shift copies of every small-parameter accumulator, copy = lane % copies: the lanes of a wave mostly
// add to the SAME parameter (the one light, the white walls), which on a single LDS word serialises 64-fold.
//
// FIXED-POINT accumulation (round 2).  An LDS fp32 atomic add is executed lane by lane on this chip (~2 cycles per active lane, ~120 per
// full wave instruction) while a 64-bit integer one takes ~9 (profiles/r01_j_ptracer_pmc_summary.txt); compiled out, the fp32 atomics of
// this kernel cost 0.43 ms of 4.76 (profiles/r02_prb_experiments.txt).  So every accumulator has a 64-bit integer twin, `s_fix`, in units
// of  q = 2^-fix_shift,  with the exponent chosen per WORKGROUP from B = max |dL| over its lanes (every addend is dL times path
// quantities):  B * 2^fix_shift < 2^36.  An addend below  fix_limit = 2^50 q  (i.e. 2^14 B) goes to the integer twin -- at most
// 256 lanes x 8 vertices of them per accumulator keep the sum under 2^61 --, anything larger, non-finite, or a workgroup whose B is 0 /
// non-finite uses the fp32 accumulator as before.  The quantum is 2^-36 of the largest dL of the workgroup: finer than the rounding of
// the fp32 sum it replaces.  The two are added when the workgroup publishes.
struct GradSink1 { float *s_grad; uint32_t copy_shift; float4 *q; uint32_t *qn; uint32_t qcap; uint32_t *overflow; bool have_tex_grad;
                   unsigned long long *s_fix; float fix_scale, fix_limit; };
// x (already scaled, |x| < 2^50) -> 64-bit two's complement without a 64-bit conversion.  The MAGNITUDE is split -- hi = floor(|x| / 2^32)
// fits an int, lo = |x| - hi 2^32 is in [0, 2^32) and exact (|x| has 24 significant bits, lo keeps its low ones) -- and the sign is applied in
// integer arithmetic: splitting x itself gave a negative |x| < 2^32 the low word 2^32 + x, which a float only holds to 256 quanta.
MI_DEV unsigned long long fix_from_float(float x) {
    const float ax = fabsf(x);
    const float hi = floorf(ax * 2.3283064365386963e-10f);                  // 2^-32
    const float lo = fmaf(hi, -4294967296.f, ax);
    const unsigned long long mag = ((unsigned long long) (uint32_t) hi << 32) | (unsigned long long) (uint32_t) lo;
    return x < 0.f ? 0ull - mag : mag;
}
MI_DEV float fix_to_float(unsigned long long v, float inv_scale) { return (float) (long long) v * inv_scale; }
MI_DEV void small_grad_add(const GradSink1 &sink, int32_t slot, v3 g) {
    const uint32_t c = threadIdx.x & ((1u << sink.copy_shift) - 1u);
    const uint32_t at = (((uint32_t) slot) << sink.copy_shift) + c, step = 1u << sink.copy_shift;
    if (((int) (fabsf(g.x) < sink.fix_limit) & (int) (fabsf(g.y) < sink.fix_limit) & (int) (fabsf(g.z) < sink.fix_limit)) != 0) {   // (false for NaN, and always when fix_limit = 0)
        unsigned long long *fb = sink.s_fix + at;
        atomicAdd(fb, fix_from_float(g.x * sink.fix_scale)); atomicAdd(fb + step, fix_from_float(g.y * sink.fix_scale)); atomicAdd(fb + 2u * step, fix_from_float(g.z * sink.fix_scale));
    } else {
        float *gb = sink.s_grad + at;
        atomicAdd(gb, g.x); atomicAdd(gb + step, g.y); atomicAdd(gb + 2u * step, g.z);
    }
}
// Sum of v over the 64 lanes of the wave (all lanes must be executing): four DPP adds give every lane the sum of its row of 16,
// the four row sums are read back as scalars.
template <int CTRL> MI_DEV float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
MI_DEV float wave_sum(float v) {
    v = dpp_add<0xB1>(v);                   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);                  // row_half_mirror
    v = dpp_add<0x140>(v);                  // row_mirror
    const int b = __float_as_int(v);
    return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16)))
         + (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}
// The same accumulation from CONVERGED code (every lane of the wave calls it; slot < 0 = nothing to add).  An LDS fp32 atomic add
// is executed lane by lane on this chip (about 2 cycles per active lane, ~120 per full wave instruction, against ~9 for an integer
// one: profiles/r01_j_ptracer_pmc_summary.txt), and a wave's lanes mostly add to the SAME accumulator: the one light's radiance,
// the wall the 64 samples of a pixel see.  So the lanes that share a slot are summed across the wave first (DPP, 11 instructions per
// channel) and ONE lane adds; after MI_SMALL_GROUPS distinct slots the rest falls back to one atomic per lane.
// Used ONCE per walk, for the radiance sums the lanes carried in registers (Prb1State::rad): doing it at every vertex for every
// accumulator costs ~45 VALU instructions per group and was measured at no gain (4.76 -> 4.71 ms on BASELINE configs[2]).
#define MI_SMALL_GROUPS 3
MI_DEV void small_grad_add_wave(const GradSink1 &sink, int32_t slot, v3 g) {
    unsigned long long todo = __ballot(slot >= 0);
    const uint32_t lane_id = threadIdx.x & 63u;
    for (int round = 0; round < MI_SMALL_GROUPS && todo; ++round) {
        const int first = (int) __ffsll((long long) todo) - 1;
        const int32_t s0 = __builtin_amdgcn_readlane(slot, first);
        const bool mine = slot == s0;                                       // (slot >= 0 follows: s0 >= 0)
        const float sx = wave_sum(mine ? g.x : 0.f), sy = wave_sum(mine ? g.y : 0.f), sz = wave_sum(mine ? g.z : 0.f);
        if (lane_id == (uint32_t) first) {
            const uint32_t c = (threadIdx.x >> 6) & ((1u << sink.copy_shift) - 1u);        // one copy per wave
            float *gb = sink.s_grad + (((uint32_t) s0) << sink.copy_shift) + c;
            atomicAdd(gb, sx); atomicAdd(gb + (1u << sink.copy_shift), sy); atomicAdd(gb + (2u << sink.copy_shift), sz);
        }
        todo &= ~__ballot(mine);
        if (mine) slot = -1;
    }
    if (slot >= 0) small_grad_add(sink, slot, g);
}

// One bounce of the one-pass adjoint walk at depth `it` (for a live path the loop counter IS its depth, prb.py:249).
// `rad_slot`, `rad`: the lane's pending d/d(radiance) of ONE emitter.  Both the emitter that is hit and the emitter that is sampled add
// to a radiance accumulator at every vertex, and scenes have few emitters: the sum stays in registers for the whole walk and goes to
// LDS once, summed over the wave (prb_sample_1p); a lane that meets a second emitter flushes the first one on the spot.
struct Prb1State { Ray ray; v3 beta, P, prev_p, dL; float prev_bsdf_pdf, eta; bool prev_delta; Pcg32 rng; int32_t rad_slot; v3 rad; };
MI_DEV void prb1_init(Prb1State &st, const Ray &ray, const Pcg32 &rng, v3 dL) {
    st.ray = ray; st.beta = V3(1.f, 1.f, 1.f); st.P = V3(0.f, 0.f, 0.f); st.prev_p = V3(0.f, 0.f, 0.f); st.dL = dL;
    st.prev_bsdf_pdf = 1.f; st.eta = 1.f; st.prev_delta = true; st.rng = rng; st.rad_slot = -1; st.rad = V3(0.f, 0.f, 0.f);
}
MI_DEV void radiance_grad_add(Prb1State &st, const GradSink1 &sink, int32_t slot, v3 g) {
    if (st.rad_slot != slot) {
        if (st.rad_slot >= 0) small_grad_add(sink, st.rad_slot, st.rad);
        st.rad_slot = slot; st.rad = V3(0.f, 0.f, 0.f);
    }
    st.rad = st.rad + g;
}
// `lane_tag` = the workgroup slot the path STARTED in (records and the published L_total * dL are keyed by it).
template <int FULL>
MI_DEV bool prb1_step(const DScene &sc, Prb1State &st, uint32_t it, uint32_t max_depth, uint32_t rr_depth, bool hide_emitters,
                      bool active, uint32_t *stk, const GradSink1 &sink, uint32_t lane_tag, Counters3 &cnt) {
    Hit hit = scene_intersect<FULL>(sc, st.ray, active, stk);
    cnt.isect += wave_count(active);
    bool si_valid = active && hit.t != MI_INF;
    cnt.shade += wave_count(si_valid);                                  // (counted here: ballot(true) inside the branch gets hoisted)
    bool active_next = si_valid && !(hide_emitters && it == 0 && !si_valid);
    bool alive = false;
    bool shadow_req = false;                 // counted after the divergent region: counters live in SGPRs
    if (si_valid) {
            SI si = compute_si<FULL>(sc, st.ray, hit);
        const DShape &sh = sc.shapes[hit.shape];
        int32_t emitter = sh.emitter;
        v3 Le = V3(0.f, 0.f, 0.f); float le_mis = 0.f; bool le_front = false;
        if (emitter >= 0) {                                            // prb.py:126-134
            v3 rel = si.p - st.prev_p; float dist = norm3(rel); v3 d = rel * rcp_(dist);
            float em_pdf = st.prev_delta ? 0.f : pdf_emitter_direction<FULL>(sc, emitter, st.prev_p, d, si.sh_n, dist);
            le_mis = mis_weight(st.prev_bsdf_pdf, em_pdf);
            le_front = si.wi.z > 0.f;
            v3 e = le_front ? ld3(sc.emitters[emitter].radiance) : V3(0.f, 0.f, 0.f);
            Le = (st.beta * le_mis) * e;
            if (le_front && sc.emitters[emitter].grad_slot >= 0)       // d Le / d radiance
                radiance_grad_add(st, sink, sc.emitters[emitter].grad_slot, V3(st.dL.x * st.beta.x * le_mis, st.dL.y * st.beta.y * le_mis, st.dL.z * st.beta.z * le_mis));
        }
        active_next = active_next && (it + 1 < max_depth);          // prb.py:139
        if (!(it + 1 < max_depth)) st.P = st.P + Le;                   // last vertex: only Le remains (see prb_sample)
        else {
        const DBsdf &bsdf = sc.bsdfs[sh.bsdf];
        TexTaps taps; v3 refl;
        if (bsdf.texture >= 0) refl = texture_eval(sc, sc.textures[bsdf.texture], si.uv, &taps);
        else refl = ld3(bsdf.reflectance);
        // ---- emitter sampling (prb.py:142-162)
        float ex = next_1d(st.rng), ey = next_1d(st.rng);
        DS ds; v3 em_weight; Ray shadow; shadow.o = shadow.d = V3(0.f, 0.f, 1.f); shadow.maxt = 0.f;
        bool smooth = !FULL || (bsdf.type != MI355_BSDF_DIELECTRIC && bsdf.type != MI355_BSDF_CONDUCTOR);
        bool need_shadow = sample_emitter_direction<FULL>(sc, si, ex, ey, active_next && smooth, ds, em_weight, shadow);
        bool occluded = scene_ray_test<FULL>(sc, shadow, need_shadow, stk);
        shadow_req = need_shadow;
        bool active_em = need_shadow && !occluded;
        if (!active_em) { ds.pdf = 0.f; em_weight = V3(0.f, 0.f, 0.f); }
        v3 wo = si_to_local(si, ds.d);
        v3 bsdf_value_em; float bsdf_pdf_em; bsdf_eval_pdf<FULL>(bsdf, refl, si.wi, wo, active_em, bsdf_value_em, bsdf_pdf_em);
        const bool is_diffuse = !FULL || bsdf.type == MI355_BSDF_DIFFUSE;
        float mis_em = mis_weight(ds.pdf, bsdf_pdf_em);
        v3 Lr_dir = ((st.beta * mis_em) * bsdf_value_em) * em_weight;
        // ---- detached BSDF sampling (prb.py:166-169)
        float s1 = next_1d(st.rng), s2x = next_1d(st.rng), s2y = next_1d(st.rng);
        BSample bsm; v3 bsdf_weight; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, active_next, bsm, bsdf_weight);
        v3 bs_wo = bsm.wo; float bs_pdf = bsm.pdf;
        // ---- update (prb.py:173-182)
        v3 beta_pre = st.beta;
        st.P = (st.P + Le) + Lr_dir;
        st.ray = spawn_ray(si.p, si.n, si_to_world(si, bs_wo));
        st.beta = st.beta * bsdf_weight;
        st.prev_p = si.p; st.prev_bsdf_pdf = bs_pdf; st.prev_delta = FULL ? bsm.delta : false;
        if (FULL) st.eta *= bsm.eta;
        // ---- stopping criterion (prb.py:186-198)
        float bmax = max3(st.beta);
        active_next = active_next && bmax != 0.f;
        float rr_prob = fminf(bmax * (st.eta * st.eta), .95f);
        bool rr_active = it >= rr_depth;
        if (rr_active) st.beta = st.beta * rcp_(rr_prob);
        bool rr_continue = next_1d(st.rng) < rr_prob;
        active_next = active_next && (!rr_active || rr_continue);
        // ---- gradients (prb.py:202-247 in closed form, split into the L_total-proportional and the known part)
        {
            float co = bs_wo.z; bool ev_active = active_next && si.wi.z > 0.f && co > 0.f && is_diffuse;
            v3 det = bsdf_weight * bs_pdf;
            v3 inv = V3(det.x != 0.f ? rcp_(det.x) : 0.f, det.y != 0.f ? rcp_(det.y) : 0.f, det.z != 0.f ? rcp_(det.z) : 0.f);
            float dval = ev_active ? MI_INV_PI * co : 0.f;
            bool em_act = active_em && si.wi.z > 0.f && wo.z > 0.f && is_diffuse;
            float dval_em = em_act ? MI_INV_PI * wo.z : 0.f;
            v3 direct = ((beta_pre * mis_em) * dval_em) * em_weight;   // d Lr_dir / d albedo
            v3 A = inv * dval;                                         // the lane publishes L_total * st.dL
            v3 Cc = st.dL * (direct - (st.P * inv) * dval);
            bool want_tex = is_diffuse && bsdf.texture >= 0 && sink.have_tex_grad && sc.textures[bsdf.texture < 0 ? 0 : bsdf.texture].grad_slot >= 0;
            bool want_small = is_diffuse && bsdf.texture < 0 && bsdf.grad_slot >= 0;
            if (want_small) {                                          // the known part needs no queue
                small_grad_add(sink, bsdf.grad_slot, Cc);
                want_small = ev_active;                                // A' == 0 otherwise
            }
            unsigned long long mt = __ballot(want_tex), ms = __ballot(want_small);
            if (mt | ms) {                                             // wave-aggregated granule allocation: one LDS atomic per wave
                uint32_t rt = __builtin_amdgcn_mbcnt_hi((uint32_t) (mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mt, 0u));
                uint32_t rs = __builtin_amdgcn_mbcnt_hi((uint32_t) (ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) ms, 0u));
                const uint32_t nt = (uint32_t) __popcll(mt), ns = (uint32_t) __popcll(ms);
                const uint32_t leader = (uint32_t) __ffsll((long long) (mt | ms)) - 1u;   // lane 0 may be masked off here
                const bool lead = (threadIdx.x & 63u) == leader;
                uint32_t base = 0;
                if (lead) base = atomicAdd(sink.qn, ns | (nt << 16));
                base = (uint32_t) __builtin_amdgcn_readlane((int) base, (int) leader);
                const uint32_t base_s = base & 0xffffu, base_t = base >> 16;
                const uint32_t g = want_tex ? sink.qcap - 3u * (base_t + rt + 1u) : base_s + rs;   // granule (texture) or dword (constant) index
                if ((base_s + ns + 3u) / 4u + 3u * (base_t + nt) > sink.qcap || base_s + ns > 0xffffu || base_t + nt > 0xffffu) { if (lead) *sink.overflow = 1u; }
                else if (want_tex) {
                    int32_t i11 = taps.idx[3] >= 0 ? taps.idx[3] : taps.idx[0];          // nearest filter: one tap of weight 1
                    float w1x = taps.w[1] + taps.w[3], w1y = taps.w[2] + taps.w[3];   // w = (w0y|w1y) * (w0x|w1x), sums to 1
                    sink.q[g] = make_float4(__uint_as_float(((uint32_t) bsdf.texture << 8) | lane_tag), __uint_as_float((uint32_t) taps.idx[0]),
                                            __uint_as_float((uint32_t) i11), w1x);
                    sink.q[g + 1] = make_float4(w1y, A.x, A.y, A.z);
                    const int32_t dx = taps.idx[1] >= 0 ? taps.idx[1] - taps.idx[0] : 0;   // x1 - x0 (negative when the tap wraps)
                    sink.q[g + 2] = make_float4(__uint_as_float((uint32_t) dx), Cc.x, Cc.y, Cc.z);
                } else if (want_small)
                    reinterpret_cast<uint32_t *>(sink.q)[g] = ((uint32_t) sh.bsdf << 8) | lane_tag;
            }
            if (active_em && sc.emitters[ds.emitter].grad_slot >= 0) {                       // d Lr_dir / d radiance
                v3 w = ((beta_pre * mis_em) * bsdf_value_em) * rcp_(ds.pdf);
                radiance_grad_add(st, sink, sc.emitters[ds.emitter].grad_slot, st.dL * w);
            }
        }
        alive = active_next;
        }
    }
    cnt.shadow += wave_count(shadow_req);
    return alive;
}
template <int FULL>
MI_DEV void prb_sample_1p(const DScene &sc, Pcg32 &rng, Ray ray, uint32_t max_depth, uint32_t rr_depth, bool hide_emitters,
                          bool active, uint32_t *stk, v3 dL, GradSink1 sink, v3 &L_out, Counters3 &cnt) {
    Prb1State st; prb1_init(st, ray, rng, dL);
    for (uint32_t it = 0; it < max_depth && __any(active); ++it) {
        MI_SCENE_RELOAD(scb);
        active = prb1_step<FULL>(scb, st, it, max_depth, rr_depth, hide_emitters, active, stk, sink, threadIdx.x, cnt);
    }
    small_grad_add_wave(sink, st.rad_slot, st.rad);                         // (converged: the loop's exit test is wave-uniform)
    rng = st.rng; L_out = st.P;
}

// -------------------------------------------------------------------------------------------
// film: ImageBlock::put coalesced path (imageblock.cpp:418-531)
// -------------------------------------------------------------------------------------------
#define MI_MAX_TAPS 5
struct Splat { int32_t x0, y0; int count; float wx[MI_MAX_TAPS], wy[MI_MAX_TAPS]; };
MI_DEV Splat splat_weights(const DScene &sc, float pos_x, float pos_y) {
    Splat s;
    if (sc.rfilter == MI355_RFILTER_BOX) {                                 // imageblock.cpp:210-233
        s.x0 = (int32_t) floorf(pos_x) - (int32_t) sc.crop_x; s.y0 = (int32_t) floorf(pos_y) - (int32_t) sc.crop_y;
        s.count = 1; s.wx[0] = s.wy[0] = 1.f;
#pragma unroll
        for (int i = 1; i < MI_MAX_TAPS; ++i) s.wx[i] = s.wy[i] = 0.f;
        return s;
    }
    int n = (int) ceilf(sc.filter_radius - .5f); s.count = 2 * n + 1;
    int32_t pix = (int32_t) floorf(pos_x) - n, piy = (int32_t) floorf(pos_y) - n;
    s.x0 = pix - (int32_t) sc.crop_x; s.y0 = piy - (int32_t) sc.crop_y;
    float rx = (float) pix + .5f - pos_x, ry = (float) piy + .5f - pos_y;
#pragma unroll
    for (int i = 0; i < MI_MAX_TAPS; ++i) {
        s.wx[i] = i < s.count ? filter_eval(sc, rx) : 0.f; s.wy[i] = i < s.count ? filter_eval(sc, ry) : 0.f;
        rx += 1.f; ry += 1.f;
    }
    return s;
}

// -------------------------------------------------------------------------------------------
// Particle tracer: src/integrators/ptracer.cpp:77-371 (one lane per light sample; AdjointIntegrator::render drives it)
// -------------------------------------------------------------------------------------------
MI_DEV v3 xf_point_proj(const float *m, v3 p) {                             // Transform4f * Point3f with the homogeneous divide
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fmaf(m[i * 4 + 2], p.z, fmaf(m[i * 4 + 1], p.y, fmaf(m[i * 4 + 0], p.x, m[i * 4 + 3])));
    float iw = 1.f / r[3];
    return V3(r[0] * iw, r[1] * iw, r[2] * iw);
}
// RealisticLensCamera::sample_direction (realisticlens.cpp:2296-2345): a point of the FRONT element is sampled, the ray from the
// reference point through it is traced world -> film (interfaces in reverse order), the film hit gives the pixel.
MI_DEV bool lens_sample_direction(const DScene &sc, v3 ref_p, float sx_, float sy_, v2 &uv, v3 &ap_local, float &weight) {
    const float *Lf = sc.lens + MI_LENS_ROW * (sc.lens_count - 1u);
    float dkx, dky; disk_concentric(sx_, sy_, dkx, dky);
    v3 aperture_p = V3(dkx * Lf[2], dky * Lf[2], Lf[3]);
    v3 dir = ref_p - aperture_p; float dist_ref = sqrtf(sqnorm3(dir)), inv_dist_ref = 1.f / dist_ref;
    dir = dir * inv_dist_ref;
    v3 o = ref_p, d = -dir;
    bool ok = true;
    for (int32_t k = (int32_t) sc.lens_count - 1; k >= 0 && ok; --k) ok = lens_interact(sc.lens + MI_LENS_ROW * k, o, d);
    if (!ok) return false;
    float tt = -o.z / d.z; v3 fp = fma3s(d, tt, o);
    float rsx = (float) sc.crop_w / (float) sc.film_w, rsy = (float) sc.crop_h / (float) sc.film_h;
    float rox = (float) sc.crop_x / (float) sc.film_w, roy = (float) sc.crop_y / (float) sc.film_h;
    float sx = ((fp.x / sc.film_px + 1.f) * 0.5f - rox) / rsx, sy = ((fp.y / sc.film_py + 1.f) * 0.5f - roy) / rsy;
    if (!(sx >= 0.f && sx <= 1.f && sy >= 0.f && sy <= 1.f)) return false;
    float normalization = 1.f / ((2.f * sc.film_px * rsx) * (2.f * sc.film_py * rsy));     // 1 / area of the crop window on the film, :2197-2203
    float ct_film = -d.z, inv_ct = 1.f / ct_film, rear_z = sc.lens[3];
    float value = normalization * (inv_ct * inv_ct) * (inv_ct * inv_ct) * (rear_z * rear_z);
    uv.x = sx * (float) sc.crop_w; uv.y = sy * (float) sc.crop_h;
    ap_local = aperture_p;
    weight = value * inv_dist_ref * inv_dist_ref * dir.z;
    return true;
}
// PerspectiveCamera::sample_direction + importance, perspective.cpp:282-386
template <int FULL>
MI_DEV bool sensor_sample_direction(const DScene &sc, v3 p, float sx_, float sy_, v2 &uv, v3 &ds_p, float &weight) {
    const float *tw = sc.cam_to_world;
    v3 rel = p - V3(tw[3], tw[7], tw[11]);                                  // inverse of the rigid camera frame
    v3 ref_p = V3(tw[0] * rel.x + tw[4] * rel.y + tw[8] * rel.z, tw[1] * rel.x + tw[5] * rel.y + tw[9] * rel.z, tw[2] * rel.x + tw[6] * rel.y + tw[10] * rel.z);
    if (!(ref_p.z >= sc.near_clip && ref_p.z <= sc.far_clip)) return false;
    if (FULL && sc.sensor_type == MI355_SENSOR_REALISTICLENS) {
        v3 ap;
        if (!lens_sample_direction(sc, ref_p, sx_, sy_, uv, ap, weight)) return false;
        ds_p = xf_point(tw, ap);
        return true;
    }
    v3 screen = xf_point_proj(sc.camera_to_sample, ref_p);
    if (!(screen.x >= 0.f && screen.x <= 1.f && screen.y >= 0.f && screen.y <= 1.f)) return false;
    uv.x = screen.x * (float) sc.crop_w; uv.y = screen.y * (float) sc.crop_h;
    float dist = norm3(ref_p), inv_dist = 1.f / dist;
    v3 local_d = ref_p * inv_dist;
    ds_p = V3(tw[3], tw[7], tw[11]);
    float ct = local_d.z, inv_ct = 1.f / ct, qx = local_d.x * inv_ct, qy = local_d.y * inv_ct;
    bool valid = ct > 0.f && qx >= sc.image_rect[0] && qx <= sc.image_rect[2] && qy >= sc.image_rect[1] && qy <= sc.image_rect[3];
    float importance = valid ? sc.normalization * inv_ct * inv_ct * inv_ct : 0.f;
    weight = importance * inv_dist * inv_dist;
    return true;
}
// Film position -> position in the virtual film of a DSensorSet.  With a box filter only the pixel floor(x) matters, and x + offset may round up
// across a pixel edge (x within 2^-16 of it, offsets up to 8 pitches): take the pixel first, so that the sample lands where the camera's own film
// would have put it.  The other filters see the position perturbed by that rounding (< 2e-5 pixel), their weights by as much.
MI_DEV float multi_film_x(const DScene &sc, float x, float offset) { return (sc.rfilter == MI355_RFILTER_BOX ? floorf(x) + .5f : x) + offset; }
// The perspective branch of sensor_sample_direction for one camera of a DSensorSet (uv in that camera's own film, before its x_offset)
MI_DEV bool sensor_view_sample_direction(const DSensorView &sv, float film_w, float film_h, v3 p, v2 &uv, v3 &ds_p, float &weight) {
    const float *tw = sv.cam_to_world;
    v3 rel = p - V3(tw[3], tw[7], tw[11]);
    v3 ref_p = V3(tw[0] * rel.x + tw[4] * rel.y + tw[8] * rel.z, tw[1] * rel.x + tw[5] * rel.y + tw[9] * rel.z, tw[2] * rel.x + tw[6] * rel.y + tw[10] * rel.z);
    if (!(ref_p.z >= sv.near_clip && ref_p.z <= sv.far_clip)) return false;
    v3 screen = xf_point_proj(sv.camera_to_sample, ref_p);
    if (!(screen.x >= 0.f && screen.x <= 1.f && screen.y >= 0.f && screen.y <= 1.f)) return false;
    uv.x = screen.x * film_w; uv.y = screen.y * film_h;
    float dist = norm3(ref_p), inv_dist = 1.f / dist;
    v3 local_d = ref_p * inv_dist;
    ds_p = V3(tw[3], tw[7], tw[11]);
    float ct = local_d.z, inv_ct = 1.f / ct, qx = local_d.x * inv_ct, qy = local_d.y * inv_ct;
    bool valid = ct > 0.f && qx >= sv.image_rect[0] && qx <= sv.image_rect[2] && qy >= sv.image_rect[1] && qy <= sv.image_rect[3];
    float importance = valid ? sv.normalization * inv_ct * inv_ct * inv_ct : 0.f;
    weight = importance * inv_dist * inv_dist;
    return true;
}
// Shape::sample_position: rectangle.cpp:166-180, sphere.cpp:210-229
template <int FULL>
MI_DEV void shape_sample_position(const DScene &sc, const DShape &sh, float sx, float sy, v3 &p, v3 &n, float &pdf) {
    if (sh.type == MI355_SHAPE_MESH) { mesh_sample_position(sc, sh, sx, sy, p, n, pdf); return; }
    if (!FULL || sh.type == MI355_SHAPE_RECTANGLE) {
        p = xf_point(sh.to_world, V3(sx * 2.f - 1.f, sy * 2.f - 1.f, 0.f)); n = ld3(sh.frame_n); pdf = sh.inv_area;
    } else if (sh.type == MI355_SHAPE_DISK) {                              // disk.cpp:173-191
        float qx, qy; disk_concentric(sx, sy, qx, qy);
        p = xf_point(sh.to_world, V3(qx, qy, 0.f)); n = ld3(sh.frame_n); pdf = sh.inv_area;
    } else {
        v3 center = ld3(sh.frame_s); float radius = sh.frame_t[0];
        float z = fmaf(-2.f, sy, 1.f), r = safe_sqrt_(fmaf(-z, z, 1.f)), ph = 2.f * MI_PI * sx;      // square_to_uniform_sphere, warp.h:250-255
        v3 local = V3(r * cosf(ph), r * sinf(ph), z);
        p = fma3s(local, radius, center); n = (sh.flags & DSHAPE_FLIP) ? -local : local; pdf = sh.inv_area;
    }
}
// Deferred splat of the particle tracer (reconstruction filters with more than one tap): a connection's 5 x 5 normalised taps cost
// 75 memory-side float atomics scattered over the film -- the atomic units, not the tracing, then set the pace (Cornell box 512^2 x
// 64 spp: 234 ms against 15 ms with a box filter).  Instead the connection is APPENDED as one 20-byte record (film position, colour)
// with one wave-aggregated counter atomic; k_splat_hist / k_splat_scan / k_splat_scatter sort the records by 16 x 16 film tile (counting
// sort whose histogram and cursors are privatised in LDS per 8192-record chunk: same-address GLOBAL atomics run at about one per 4 ns,
// and a light source seen by the camera sends a quarter of all records to two or three tiles) and k_splat_tiles accumulates runs of
// one tile's records in LDS and writes the tile out once per run.  Records beyond the queue's capacity fall back to the direct
// atomics, so the film is complete whatever the capacity.
#define MI_SPLAT_TILE 16
#define MI_SPLAT_APRON 2                                                    // (MI_MAX_TAPS - 1) / 2
// The queue is cut into MI_SPLAT_SEGS segments, each with its own append counter (a workgroup appends to segment blockIdx % SEGS): one
// counter for the whole launch would take ~1 M same-address atomics, ~4 ns each.  Counters sit 128 B apart.
#define MI_SPLAT_SEGS 256u
#define MI_SPLAT_COUNTER_STRIDE 32u
// (rec_c, rec_d: second half of a DERIVATIVE splat record -- colour and film-position tangent -- or null)
struct SplatQueue { float4 *rec_a; float *rec_b; float4 *rec_c; float *rec_d; uint32_t *count; uint32_t *hist; uint32_t cap, seg_cap, tiles_x, tiles_y; };
MI_DEV uint32_t splat_tile_of(const DScene &sc, const SplatQueue &q, float pos_x, float pos_y) {
    // tile of the centre pixel, clamped into the crop window (taps that stick out of the window are dropped anyway)
    int32_t cx = (int32_t) floorf(pos_x) - (int32_t) sc.crop_x, cy = (int32_t) floorf(pos_y) - (int32_t) sc.crop_y;
    cx = min(max(cx, 0), (int32_t) sc.crop_w - 1); cy = min(max(cy, 0), (int32_t) sc.crop_h - 1);
    return (uint32_t) (cy / MI_SPLAT_TILE) * q.tiles_x + (uint32_t) (cx / MI_SPLAT_TILE);
}
// ImageBlock::put with normalize = true (imageblock.cpp:460-474): per-sample weights divided by their sum; RGB only, the weight channel is
// never touched (ptracer.cpp:366-368).  The direct splat -- the overflow path of the record queue, MI355_FLAG_DIRECT_SPLAT, films too
// large for the tile sort.  The taps are walked by ROLLED loops that re-evaluate the filter per tap: it is bound by the memory-side
// atomics anyway, and unrolled its 25 taps sat in the middle of the tracing loop and cost it a thousand SGPR spill moves.
MI_DEV void film_put_normalized(const DScene &sc, float *film, float pos_x, float pos_y, v3 rgb) {
    const bool box = sc.rfilter == MI355_RFILTER_BOX;
    const int n = box ? 0 : (int) ceilf(sc.filter_radius - .5f), count = 2 * n + 1;
    const int32_t pix = (int32_t) floorf(pos_x) - n, piy = (int32_t) floorf(pos_y) - n;
    const float rx0 = (float) pix + .5f - pos_x, ry0 = (float) piy + .5f - pos_y;
    float sx = 0.f, sy = 0.f;
#pragma unroll 1
    for (int i = 0; i < count; ++i) { sx += box ? 1.f : filter_eval(sc, rx0 + (float) i); sy += box ? 1.f : filter_eval(sc, ry0 + (float) i); }
    float factor = sx * sy; factor = factor != 0.f ? 1.f / factor : 0.f;
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
#pragma unroll 1
    for (int t = 0; t < count * count; ++t) {
        const int ty = t / count, tx = t - ty * count;
        const int32_t x = pix + tx - (int32_t) sc.crop_x, y = piy + ty - (int32_t) sc.crop_y;
        if (x < 0 || y < 0 || x >= W || y >= H) continue;
        const float w = (box ? 1.f : filter_eval(sc, ry0 + (float) ty)) * ((box ? 1.f : filter_eval(sc, rx0 + (float) tx)) * factor);
        if (w == 0.f) continue;
        float *px = film + ((size_t) y * W + x) * 4;
        atomicAdd(px, rgb.x * w); atomicAdd(px + 1, rgb.y * w); atomicAdd(px + 2, rgb.z * w);
    }
}
// Called by every lane of the (possibly partial) wave that reached the connection; `put` = this lane has something to splat.
MI_DEV void film_put_deferred(const DScene &sc, const SplatQueue &q, float *film, float pos_x, float pos_y, v3 rgb, bool put) {
    if (put) {                                                              // nothing of it inside the crop window?
        int n = (int) ceilf(sc.filter_radius - .5f);
        int32_t cx = (int32_t) floorf(pos_x) - (int32_t) sc.crop_x, cy = (int32_t) floorf(pos_y) - (int32_t) sc.crop_y;
        put = cx + n >= 0 && cy + n >= 0 && cx - n < (int32_t) sc.crop_w && cy - n < (int32_t) sc.crop_h;
    }
    const unsigned long long m = __ballot(put);
    if (!put) return;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
    const uint32_t leader = (uint32_t) __ffsll((long long) m) - 1u;
    uint32_t base = 0;
    const uint32_t seg = blockIdx.x % MI_SPLAT_SEGS;
    if ((threadIdx.x & 63u) == leader) base = atomicAdd(q.count + seg * MI_SPLAT_COUNTER_STRIDE, (uint32_t) __popcll(m));
    base = (uint32_t) __builtin_amdgcn_readlane((int) base, (int) leader);
    const uint32_t local = base + rank;
    if (local < q.seg_cap && base <= 0xffffffffu - 64u) {
        const uint32_t idx = seg * q.seg_cap + local;
        q.rec_a[idx] = make_float4(pos_x, pos_y, rgb.x, rgb.y); q.rec_b[idx] = rgb.z;
    } else film_put_normalized(sc, film, pos_x, pos_y, rgb);        // queue full: the slow way, same result
}
// Adjoint of the normalised splat: the film is linear in the splatted colour, so dL/d(colour) = sum over taps of w * dL/dfilm.
MI_DEV v3 film_gather_normalized(const DScene &sc, const float *grad_rgb, float pos_x, float pos_y) {
    Splat sp = splat_weights(sc, pos_x, pos_y);
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int i = 0; i < MI_MAX_TAPS; ++i) { sx += i < sp.count ? sp.wx[i] : 0.f; sy += i < sp.count ? sp.wy[i] : 0.f; }
    float factor = sx * sy; factor = factor != 0.f ? 1.f / factor : 0.f;
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
    v3 g = V3(0.f, 0.f, 0.f);
#pragma unroll
    for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
        for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
            int32_t x = sp.x0 + tx, y = sp.y0 + ty;
            if (ty >= sp.count || tx >= sp.count || x < 0 || y < 0 || x >= W || y >= H) continue;
            float w = sp.wy[ty] * (sp.wx[tx] * factor);
            const float *px = grad_rgb + ((size_t) y * W + x) * 3;
            g.x = fmaf(px[0], w, g.x); g.y = fmaf(px[1], w, g.y); g.z = fmaf(px[2], w, g.z);
        }
    return g;
}
// Gradient mode of the particle tracer (Integrator::render_backward, integrator.cpp:65-84, over ptracer.cpp).
//   emitter radiances: every splat is  radiance_c * unit_c  with unit_c free of the radiance, so the gradient is  sum unit_c * gather_c
//                      (gather = the image adjoint collected at the splat's taps);
//   diffuse albedos (constants and textures), `all` mode: the albedo at light-path vertex j is a linear factor of the connection made
//                      AT j (bsdf eval) and of every later one (bsdf sample weight), so  d/d albedo_j = (1 / albedo_j) * S_j  with
//                      S_j = sum_{k >= j} <gather_k, splat_k>.  Path replay: walk 1 sums S_1 (and does the radiances), walk 2 re-traces
//                      the same path from the same sampler state, emits S_j / albedo_j at vertex j and then subtracts connection j.
// Accumulators for the "small" parameters (constant albedos, radiances): 2^copy_shift bank-spread copies per float in LDS (the lanes
// of a wave mostly add to the SAME parameter), flushed once per workgroup; texture gradients go out as global atomics on the texels.
struct PtGradSink { const float *grad_rgb; float *s_grad; uint32_t copy_shift; float *d_grad; const uint32_t *small_map; bool all; };
MI_DEV void pt_grad_add(const PtGradSink &sink, int32_t slot, v3 g) {
    if (slot < 0) return;
    const uint32_t c = threadIdx.x & ((1u << sink.copy_shift) - 1u);
    float *gb = sink.s_grad + (((uint32_t) slot) << sink.copy_shift) + c;
    atomicAdd(gb, g.x); atomicAdd(gb + (1u << sink.copy_shift), g.y); atomicAdd(gb + (2u << sink.copy_shift), g.z);
}
// connect_sensor, ptracer.cpp:286-371; `bsdf` = nullptr for a point on an emitter.  Returns whether a visibility ray was cast.
// GRAD: `weight_unit` = weight / radiance of `emitter`; nothing is splatted, the gathered image adjoint goes to the sink.
template <int FULL, bool GRAD = false>
MI_DEV bool connect_sensor(const DScene &sc, const SI &si, const DBsdf *bsdf, bool seen, v2 uv, v3 ds_p, v3 weight, float sample_scale,
                           float *film, uint32_t *stk, v3 weight_unit = V3(0.f, 0.f, 0.f), uint32_t emitter = 0u, const PtGradSink *sink = nullptr,
                           const SplatQueue *queue = nullptr, bool add_radiance = true, v3 *dv_out = nullptr) {
    if (GRAD && dv_out) *dv_out = V3(0.f, 0.f, 0.f);
    bool want = seen && (weight.x != 0.f || weight.y != 0.f || weight.z != 0.f);
    Ray sensor_ray = spawn_ray_to(si.p, si.n, ds_p);
    if (!want) { sensor_ray.o = sensor_ray.d = V3(0.f, 0.f, 1.f); sensor_ray.maxt = 0.f; }
    bool occluded = scene_ray_test<FULL>(sc, sensor_ray, want, stk);
    if (want && !occluded) {
        v3 local_d = si_to_local(si, sensor_ray.d);
        v3 surface_weight;
        if (bsdf == nullptr) { float c = fmaxf(0.f, local_d.z); surface_weight = V3(c, c, c); }       // foreshortening of the emitter
        else {
            float wi_dot_geo_n = dot3(si.n, si_to_world(si, si.wi)), wo_dot_geo_n = dot3(si.n, sensor_ray.d);
            bool valid = wi_dot_geo_n * si.wi.z > 0.f && wo_dot_geo_n * local_d.z > 0.f;
            float correction = valid ? fabsf((si.wi.z * wo_dot_geo_n) / (local_d.z * wi_dot_geo_n)) : 0.f;
            v3 refl = bsdf->texture >= 0 ? texture_eval(sc, sc.textures[bsdf->texture], si.uv, nullptr) : ld3(bsdf->reflectance);
            v3 val; float pdf; bsdf_eval_pdf<FULL>(*bsdf, refl, si.wi, local_d, true, val, pdf);
            surface_weight = val * correction;
        }
        if (GRAD) {
            v3 ru = (weight_unit * surface_weight) * sample_scale;
            if (ru.x != 0.f || ru.y != 0.f || ru.z != 0.f) {
                const v3 g = film_gather_normalized(sc, sink->grad_rgb, uv.x + (float) sc.crop_x, uv.y + (float) sc.crop_y);
                if (add_radiance) pt_grad_add(*sink, sc.emitters[emitter].grad_slot, ru * g);
                if (dv_out) *dv_out = ((weight * surface_weight) * sample_scale) * g;      // <gather, splat> of this connection
            }
        } else {
        v3 result = (weight * surface_weight) * sample_scale;
        const bool lit = result.x != 0.f || result.y != 0.f || result.z != 0.f;   // (a black splat changes nothing)
        if (queue && queue->cap) film_put_deferred(sc, *queue, film, uv.x + (float) sc.crop_x, uv.y + (float) sc.crop_y, result, lit);
        else if (lit) film_put_normalized(sc, film, uv.x + (float) sc.crop_x, uv.y + (float) sc.crop_y, result);
        }
    }
    return want;
}
MI_DEV uint32_t pick_emitter(const DScene &sc, float &sample, float &weight) {                      // Scene::sample_emitter, scene.cpp:227-250
    weight = 1.f;
    if (sc.n_emitters < 2u) return 0u;
    float cf = (float) sc.n_emitters, scaled = sample * cf;
    uint32_t index = min((uint32_t) scaled, sc.n_emitters - 1u);
    weight = cf; sample = scaled - (float) index;
    return index;
}
MI_DEV SI si_from_position(v3 p, v3 n, int32_t shape) {                     // SurfaceInteraction(PositionSample): frame from the normal
    SI si; si.t = 0.f; si.p = p; si.n = n; si.sh_n = n; coordinate_system(n, si.sh_s, si.sh_t); si.uv.x = si.uv.y = 0.f;
    si.wi = V3(0.f, 0.f, 1.f); si.prim = 0; si.shape = shape;
    return si;
}
// Where a light sample STARTS, as a sort key (k_pt_keys): the emitter it picks and the cell of the emitter's sample domain -- 2D position
// sample, or position x direction for emitters that scatter their rays -- its ray leaves from.  Replays the sampler calls that precede
// prepare_ray in ptracer_sample below (sample_visible_emitters draws 1 + [2 + 2] + 2 + [1] numbers, [..] for connectable emitters only)
// on a COPY of the stream.  The key only decides which hardware lane walks which sample, never a result: were it out of step with
// ptracer_sample, the render would be the same and merely less coherent.
MI_DEV uint32_t morton_spread2(uint32_t x) { x &= 0xffffu; x = (x | (x << 8)) & 0x00ff00ffu; x = (x | (x << 4)) & 0x0f0f0f0fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u; return x; }
MI_DEV uint32_t morton_spread4(uint32_t x) { x &= 0xffu; x = (x | (x << 12)) & 0x000f000fu; x = (x | (x << 6)) & 0x03030303u; x = (x | (x << 3)) & 0x11111111u; return x; }
template <int FULL>
MI_DEV uint32_t ptracer_emission_cell(const DScene &sc, Pcg32 rng, int32_t max_depth, bool hide_emitters, uint32_t bits, bool four_d) {
    const bool timed = sc.shutter_open_time > 0.f;
    if (max_depth != 0 && !hide_emitters) {
        if (timed) (void) next_1d(rng);
        float es = next_1d(rng), w; const uint32_t e0 = pick_emitter(sc, es, w);
        const bool connect = !FULL || sc.emitters[e0].type == MI355_EMITTER_AREA;
        const int skip = (connect ? 4 : 0) + 2 + (connect ? 1 : 0);
        for (int k = 0; k < skip; ++k) (void) next_1d(rng);
    }
    if (timed) (void) next_1d(rng);
    float wav = next_1d(rng), w; const uint32_t ei = pick_emitter(sc, wav, w);
    const float ax = next_1d(rng), ay = next_1d(rng), bx = next_1d(rng), by = next_1d(rng);
    const float scale = (float) (1u << bits);
    const uint32_t qa = min((uint32_t) (ax * scale), (1u << bits) - 1u), qb = min((uint32_t) (ay * scale), (1u << bits) - 1u);
    uint32_t cell;
    if (four_d) {
        const uint32_t qc = min((uint32_t) (bx * scale), (1u << bits) - 1u), qd = min((uint32_t) (by * scale), (1u << bits) - 1u);
        cell = morton_spread4(qa) | (morton_spread4(qb) << 1) | (morton_spread4(qc) << 2) | (morton_spread4(qd) << 3);
        return (ei << (4u * bits)) | cell;
    }
    cell = morton_spread2(qa) | (morton_spread2(qb) << 1);
    return (ei << (2u * bits)) | cell;
}

template <int FULL, bool GRAD = false, bool MULTI = false>
MI_DEV void ptracer_sample(const DScene &sc, Pcg32 &rng, int32_t max_depth, int32_t rr_depth, bool hide_emitters, bool lane_on,
                           float sample_scale, float *film, uint32_t *stk, Counters3 &cnt, const PtGradSink *sink = nullptr, const SplatQueue *queue = nullptr,
                           int walk = 1, v3 *S = nullptr, const DSensorSet *set = nullptr) {
    // MULTI (forward only): `set` lists the cameras; `sc` carries the virtual film they share (see DSensorSet)
    // GRAD, `all` mode: walk 1 adds every connection's <gather, splat> to *S (and does the radiance gradients); walk 2 starts from that
    // total, emits S / albedo at every diffuse vertex and subtracts the vertex's own connection afterwards.
    const bool timed = sc.shutter_open_time > 0.f;
    // ---- sample_visible_emitters, ptracer.cpp:99-171 (wave-uniform condition)
    if (max_depth != 0 && !hide_emitters) {
        if (timed) (void) next_1d(rng);
        float es = next_1d(rng), idx_weight; uint32_t ei = pick_emitter(sc, es, idx_weight);
        const DEmitter &em = sc.emitters[ei]; const DShape &sh = sc.shapes[em.shape];
        // "Don't connect delta emitters with sensor" (ptracer.cpp:114): the MASKED sampler calls of such a lane (next_2d(active),
        // next_1d(active)) leave its stream where it is -- only the sensor's unmasked next_2d() advances it
        const bool connect = !FULL || em.type == MI355_EMITTER_AREA;
        // 3.a (:121-141): in the JIT variants `dr::any_or<true>(active_e)` holds whatever the mask, so the infinite-emitter branch runs
        // and its `sampler->next_2d(active)` consumes two numbers of every connectable lane (pinned by test_ptracer.py:225's value)
        if (connect) { (void) next_1d(rng); (void) next_1d(rng); }
        float sx = 0.f, sy = 0.f; if (connect) { sx = next_1d(rng); sy = next_1d(rng); }
        v3 p, n; float pdf; shape_sample_position<FULL>(sc, sh, sx, sy, p, n, pdf);
        float pos_weight = (connect && pdf > 0.f) ? 1.f / pdf : 0.f;
        SI si = si_from_position(p, n, (int32_t) em.shape);
        float ssx = next_1d(rng), ssy = next_1d(rng);                       // sensor->sample_direction's 2D sample (front-element point of a lens)
        v2 uv; uv.x = uv.y = 0.f; v3 ds_p = V3(0.f, 0.f, 0.f); float sensor_weight = 0.f;
        if (connect) (void) next_1d(rng);                                   // sample_wavelengths: the radiance itself in RGB mode
        if (MULTI) {
            for (uint32_t k = 0; k < set->n; ++k) {
                const bool seen = lane_on && connect && sensor_view_sample_direction(set->v[k], set->film_w, set->film_h, si.p, uv, ds_p, sensor_weight);
                uv.x = multi_film_x(sc, uv.x, set->v[k].x_offset);
                const bool cast = connect_sensor<FULL, GRAD>(sc, si, nullptr, seen, uv, ds_p, ld3(em.radiance) * (idx_weight * pos_weight * sensor_weight), sample_scale, film, stk,
                                                             V3(0.f, 0.f, 0.f), ei, sink, queue, walk == 1);
                cnt.shadow += wave_count(cast);
            }
        } else {
        bool seen = lane_on && connect && sensor_sample_direction<FULL>(sc, si.p, ssx, ssy, uv, ds_p, sensor_weight);
        v3 weight = ld3(em.radiance) * (idx_weight * pos_weight * sensor_weight);
        const float wu = idx_weight * pos_weight * sensor_weight;
        bool cast = connect_sensor<FULL, GRAD>(sc, si, nullptr, seen, uv, ds_p, weight, sample_scale, film, stk, V3(wu, wu, wu), ei, sink, queue, walk == 1);
        cnt.shadow += wave_count(cast);
        }
    }
    // ---- prepare_ray, ptracer.cpp:174-191 -> Scene::sample_emitter_ray (scene.cpp:262-297) -> AreaLight::sample_ray (area.cpp:92-115)
    if (timed) (void) next_1d(rng);
    float wav = next_1d(rng), idx_weight; uint32_t ei = pick_emitter(sc, wav, idx_weight);
    float ax = next_1d(rng), ay = next_1d(rng), bx = next_1d(rng), by = next_1d(rng);
    const DEmitter &em = sc.emitters[ei]; const DShape &esh = sc.shapes[em.shape];
    v3 ep, en; float epdf; shape_sample_position<FULL>(sc, esh, ax, ay, ep, en, epdf);
    SI esi = si_from_position(ep, en, (int32_t) em.shape);
    Ray ray = spawn_ray(esi.p, esi.n, si_to_world(esi, square_to_cosine_hemisphere(bx, by)));
    v3 throughput = ld3(em.radiance) * ((epdf > 0.f ? 1.f / epdf : 0.f) * MI_PI * idx_weight);
    float tu0 = (epdf > 0.f ? 1.f / epdf : 0.f) * MI_PI * idx_weight;        // throughput / radiance (gradient mode)
    if (FULL && em.type == MI355_EMITTER_DIRECTIONALAREA) tu0 = (1.f / esh.inv_area) * idx_weight;
    v3 tu = V3(tu0, tu0, tu0);
    if (FULL && em.type == MI355_EMITTER_DIRECTIONALAREA) {                 // DirectionalArea::sample_ray, directionalarea.cpp:88-113: along the normal, weight = area * radiance
        ray = spawn_ray(esi.p, esi.n, esi.n);
        throughput = ld3(em.radiance) * ((1.f / esh.inv_area) * idx_weight);
    }
    bool active = lane_on && max3(throughput) != 0.f;
    // ---- trace_light_ray, ptracer.cpp:200-284
    float eta = 1.f; int32_t depth = 1;
    Hit hit = scene_intersect<FULL>(sc, ray, active, stk);
    cnt.isect += wave_count(active);
    active = active && hit.t != MI_INF;
    if (max_depth >= 0) active = active && depth < max_depth;
    while (__any(active)) {
        MI_SCENE_RELOAD(scb); const DScene &sc = scb;                       // scene constants are re-fetched per bounce, not carried in spilled SGPRs
        cnt.shade += wave_count(active);
        bool cast = false, next_active = false; uint32_t cast_mask = 0u;
        Ray next_ray = ray;
        if (active) {
            SI si = compute_si<FULL>(sc, ray, hit);
            const DBsdf &bsdf = sc.bsdfs[sc.shapes[hit.shape].bsdf];
            float ssx = next_1d(rng), ssy = next_1d(rng);
            v2 uv; uv.x = uv.y = 0.f; v3 ds_p = V3(0.f, 0.f, 0.f); float sensor_weight = 0.f;
            bool seen;
            v3 dv = V3(0.f, 0.f, 0.f);
            if (MULTI) {
                for (uint32_t k = 0; k < set->n; ++k) {                     // one connection per camera; the two numbers drawn above are every camera's (a pinhole ignores them)
                    seen = sensor_view_sample_direction(set->v[k], set->film_w, set->film_h, si.p, uv, ds_p, sensor_weight);
                    uv.x = multi_film_x(sc, uv.x, set->v[k].x_offset);
                    MI_SCENE_RELOAD(sc2);
                    const bool c = connect_sensor<FULL, GRAD>(sc2, si, &bsdf, seen, uv, ds_p, throughput * sensor_weight, sample_scale, film, stk, V3(0.f, 0.f, 0.f), ei, sink, queue, walk == 1, nullptr);
                    if (c) cast_mask |= 1u << k;
                }
            } else {
            { MI_SCENE_RELOAD(sc1); seen = sensor_sample_direction<FULL>(sc1, si.p, ssx, ssy, uv, ds_p, sensor_weight); }   // (camera matrices: live here only)
            { MI_SCENE_RELOAD(sc2);
              cast = connect_sensor<FULL, GRAD>(sc2, si, &bsdf, seen, uv, ds_p, throughput * sensor_weight, sample_scale, film, stk, tu * sensor_weight, ei, sink, queue,
                                                walk == 1, (GRAD && S) ? &dv : nullptr); }
            }
            float s1 = next_1d(rng), s2x = next_1d(rng), s2y = next_1d(rng);
            TexTaps taps;
            MI_SCENE_RELOAD(sc3);
            v3 refl = bsdf.texture >= 0 ? texture_eval(sc3, sc3.textures[bsdf.texture], si.uv, (GRAD && S && walk == 2) ? &taps : nullptr) : ld3(bsdf.reflectance);
            if (GRAD && S) {
                if (walk == 1) *S = *S + dv;
                else {
                    if (!FULL || bsdf.type == MI355_BSDF_DIFFUSE) {
                        const v3 ga = V3(refl.x != 0.f ? S->x / refl.x : 0.f, refl.y != 0.f ? S->y / refl.y : 0.f, refl.z != 0.f ? S->z / refl.z : 0.f);
                        if (bsdf.texture < 0) pt_grad_add(*sink, bsdf.grad_slot, ga);
                        else if (sc.textures[bsdf.texture].grad_slot >= 0) {
                            const DTexture &tx = sc.textures[bsdf.texture];
                            float *gb = sink->d_grad + tx.grad_offset;
#pragma unroll
                            for (int k = 0; k < 4; ++k) if (taps.idx[k] >= 0 && taps.w[k] != 0.f) {
                                if (tx.c == 3) { atomicAdd(gb + taps.idx[k] * 3 + 0, taps.w[k] * ga.x); atomicAdd(gb + taps.idx[k] * 3 + 1, taps.w[k] * ga.y);
                                                 atomicAdd(gb + taps.idx[k] * 3 + 2, taps.w[k] * ga.z); }
                                else atomicAdd(gb + taps.idx[k], taps.w[k] * (ga.x + ga.y + ga.z));
                            }
                        }
                    }
                    *S = *S - dv;
                }
            }
            BSample bs; v3 bsdf_val; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, true, bs, bsdf_val, true);
            v3 wo_world = si_to_world(si, bs.wo);
            float wi_dot_geo_n = dot3(si.n, -ray.d), wo_dot_geo_n = dot3(si.n, wo_world);
            next_active = wi_dot_geo_n * si.wi.z > 0.f && wo_dot_geo_n * bs.wo.z > 0.f;
            float correction = fabsf((si.wi.z * wo_dot_geo_n) / (bs.wo.z * wi_dot_geo_n));
            throughput = throughput * (bsdf_val * correction); if (GRAD) tu = tu * (bsdf_val * correction);
            eta *= bs.eta;
            next_active = next_active && (throughput.x != 0.f || throughput.y != 0.f || throughput.z != 0.f);
            next_ray = spawn_ray(si.p, si.n, wo_world);
        }
        if (MULTI) { for (uint32_t k = 0; k < set->n; ++k) cnt.shadow += wave_count(((cast_mask >> k) & 1u) != 0u); }
        else cnt.shadow += wave_count(cast);
        active = next_active; ray = next_ray;
        // ptracer.cpp:264-269 intersects the sampled ray and only THEN masks the lane with `depth < max_depth`: after the last vertex a path may
        // shade, that intersection is computed and never read (it draws no random number either).  It is skipped here -- wave-uniform, `depth`
        // is -- and on the lens doublet it was the one incoherent traversal of every light path (a cosine-distributed bounce off the film).
        if (max_depth >= 0 && depth + 1 >= max_depth) active = false;
        else { hit = scene_intersect<FULL>(sc, ray, active, stk); cnt.isect += wave_count(active); }
        depth++;
        if (max_depth >= 0) active = active && depth < max_depth;
        active = active && hit.t != MI_INF;
        if (depth > rr_depth) {                                             // :271-277 (depth is wave-uniform)
            float q = fminf(max3(throughput) * (eta * eta), 0.95f);
            if (active) active = next_1d(rng) < q;                          // next_1d(active): a masked-out lane keeps its stream (matters for passes > 1)
            throughput = throughput * (1.f / q); if (GRAD) tu = tu * (1.f / q);   // q detached (path.cpp:272 does so; ptracer.cpp:277 does not -- DESIGN.md)
        }
    }
}

