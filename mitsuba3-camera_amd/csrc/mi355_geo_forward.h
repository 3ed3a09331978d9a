// mi355_geo_forward.h -- forward-mode derivative of the particle tracer's film with respect to MESH VERTEX POSITIONS and dielectric IORs.
//
// What the fork's lens-design notebooks differentiate (notebooks/scripts/lens_design/lens.py:505-565: lens surfaces are meshes whose
// vertex_positions are functions of the lens scalars; design_problem.py:403-428: rendered with `ptracer`) goes through Dr.Jit's AD in
// the reference; there is no explicit derivative code to restate.  Here it is a dual-number walk of the SAME light paths: one
// scalar parameter theta, a tangent d(vertex)/d(theta) for every vertex of one mesh, and the derivative of everything that
// depends on geometry carried alongside the primal value:
//     ray origin / direction  ->  hit distance, barycentrics, hit point, geometric and shading normal (Moeller-Trumbore, plane and
//     sphere intersections in dual arithmetic)  ->  refracted / reflected direction (delta lobes) or the diffuse frame that maps the
//     sampled local direction to the world  ->  shading-normal correction of the throughput  ->  at every diffuse vertex the sensor
//     connection: direction, distance, importance, bsdf cosine AND the film position, whose derivative enters through the derivative
//     of the (normalised) reconstruction-filter weights.
// Not differentiated, as in the reference's AD: visibility, the discrete choices (which triangle, which lobe, roulette survival).
// Stated limits of this version: perspective sensor; emitters must not sit on the moving mesh; a rough-conductor vertex keeps its
// sampled local direction and weight fixed (their dependence on the local incident direction is ignored); on spheres and disks the tangent frame
// of a DIFFUSE vertex is taken as locally constant (exact for the refractive vertices of a lens, which need no frame); an albedo
// texture is looked up at the primal uv (its spatial gradient times d(uv) is not carried).
// Checked against central finite differences of the primal film (tests/test_gpu_parity.py: test_particle_tracer_vertex_forward_mode):
// a diffuse floor mesh sliding along its normal, the notebooks' arrangement in small (collimated disk source, refracting mesh with
// smooth shading normals that is scaled, diffuse screen; Gaussian and tent filter) and a gold flat mirror mesh that turns -- relative
// L2 error of the interior of the gradient image 0.3 - 3 %.  One thing the finite difference does NOT agree with, by construction: on a
// curved, faceted mesh with interpolated normals the shading-normal correction of the throughput is a sawtooth over the facets; an
// interior derivative (this pass, and AD alike) follows its in-facet slope, whose jumps at the facet edges cancel it in reality
// (measured on a mirror ball: 29 - 60 % depending on the tessellation; near-normal refraction is hardly affected).
#pragma once
#include "mi355_geo_adjoint.h"

struct Df { float v, d; };
MI_DEV Df DF(float v, float d = 0.f) { Df r; r.v = v; r.d = d; return r; }
MI_DEV Df operator+(Df a, Df b) { return DF(a.v + b.v, a.d + b.d); }
MI_DEV Df operator-(Df a, Df b) { return DF(a.v - b.v, a.d - b.d); }
MI_DEV Df operator-(Df a) { return DF(-a.v, -a.d); }
MI_DEV Df operator*(Df a, Df b) { return DF(a.v * b.v, a.v * b.d + a.d * b.v); }
MI_DEV Df operator*(Df a, float b) { return DF(a.v * b, a.d * b); }
MI_DEV Df operator*(float b, Df a) { return DF(a.v * b, a.d * b); }
MI_DEV Df operator/(Df a, Df b) { float i = 1.f / b.v; float q = a.v * i; return DF(q, (a.d - q * b.d) * i); }
MI_DEV Df dsqrt(Df a) { float s = sqrtf(a.v); return DF(s, s > 0.f ? 0.5f * a.d / s : 0.f); }
MI_DEV Df dabs(Df a) { return a.v < 0.f ? -a : a; }
struct D3 { Df x, y, z; };
MI_DEV D3 D3c(v3 v) { D3 r; r.x = DF(v.x); r.y = DF(v.y); r.z = DF(v.z); return r; }
MI_DEV D3 D3vd(v3 v, v3 d) { D3 r; r.x = DF(v.x, d.x); r.y = DF(v.y, d.y); r.z = DF(v.z, d.z); return r; }
MI_DEV v3 val3(D3 a) { return V3(a.x.v, a.y.v, a.z.v); }
MI_DEV D3 operator+(D3 a, D3 b) { D3 r; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
MI_DEV D3 operator-(D3 a, D3 b) { D3 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
MI_DEV D3 operator-(D3 a) { D3 r; r.x = -a.x; r.y = -a.y; r.z = -a.z; return r; }
MI_DEV D3 operator*(D3 a, Df s) { D3 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
MI_DEV D3 operator*(D3 a, float s) { D3 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
MI_DEV D3 operator*(D3 a, D3 b) { D3 r; r.x = a.x * b.x; r.y = a.y * b.y; r.z = a.z * b.z; return r; }
MI_DEV Df ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MI_DEV D3 dcross(D3 a, D3 b) { D3 r; r.x = a.y * b.z - a.z * b.y; r.y = a.z * b.x - a.x * b.z; r.z = a.x * b.y - a.y * b.x; return r; }
MI_DEV D3 dnormalize(D3 a) { Df l = dsqrt(ddot(a, a)); return a * (DF(1.f) / l); }
// coordinate_system (vector.h:116-136) in dual arithmetic; the sign is a discrete choice
MI_DEV void dcoordinate_system(D3 n, D3 &s, D3 &t) {
    const float sign = copysignf(1.0f, n.z.v);
    Df a = -(DF(1.f) / (DF(sign) + n.z)), b = n.x * n.y * a;
    s.x = (n.x * n.x * a) * sign + DF(1.f); s.y = b * sign; s.z = n.x * (-sign);
    t.x = b; t.y = n.y * n.y * a + DF(sign); t.z = -n.y;
}
MI_DEV Df dfilter_eval(const DScene &sc, Df x) {
    if (sc.rfilter == MI355_RFILTER_GAUSSIAN) {                            // the same polynomial as filter_eval, Horner in x^2
        Df x2 = x * x, r = DF(sc.filter_coeff[9]);
#pragma unroll
        for (int i = 8; i >= 0; --i) r = r * x2 + DF(sc.filter_coeff[i]);
        return r.v > 0.f ? r : DF(0.f);
    }
    if (sc.rfilter == MI355_RFILTER_TENT) { Df r = DF(1.f) - dabs(x) * (1.f / sc.filter_radius); return r.v > 0.f ? r : DF(0.f); }
    return DF(fabsf(x.v) <= 0.5f ? 1.f : 0.f);                             // box: not differentiable in the position (derivative 0)
}

// d(vertex position)/d(theta) of every vertex of `shape` (3 floats each); d(int_ior / ext_ior)/d(theta) of the dielectric `bsdf`
struct GeoTangent { uint32_t shape; const float *vtan; uint32_t bsdf; float d_rel_eta; const float *ntan; };   // ntan: d(vertex normal)/d(theta) when the normals follow the positions (DSHAPE_NRM_DEP), else nullptr

// Dual hit record at a primal hit (hit, si): point, geometric normal, shading normal, tangent frame, all as functions of the dual ray.
struct DualHit { D3 p, ng, ns, sh_s, sh_t; };
template <int FULL>
MI_DEV DualHit dual_hit(const DScene &sc, const GeoTangent &gt, const Hit &hit, const SI &si, D3 o, D3 d) {
    DualHit h;
    const DShape &sh = sc.shapes[hit.shape];
    D3 dp_du = D3c(V3(0.f, 0.f, 0.f)); bool have_dpdu = false;
    if (sh.type == MI355_SHAPE_MESH) {
        const uint32_t *fi = sc.faces + 3 * (size_t) (sh.tri_offset + hit.prim);
        const uint32_t l0 = fi[0], l1 = fi[1], l2 = fi[2];
        const uint32_t i0 = sh.vtx_offset + l0, i1 = sh.vtx_offset + l1, i2 = sh.vtx_offset + l2;
        const bool moving = hit.shape == gt.shape && gt.vtan != nullptr;
        D3 p0 = D3vd(ld3(sc.positions + 3 * (size_t) i0), moving ? ld3(gt.vtan + 3 * (size_t) l0) : V3(0.f, 0.f, 0.f));
        D3 p1 = D3vd(ld3(sc.positions + 3 * (size_t) i1), moving ? ld3(gt.vtan + 3 * (size_t) l1) : V3(0.f, 0.f, 0.f));
        D3 p2 = D3vd(ld3(sc.positions + 3 * (size_t) i2), moving ? ld3(gt.vtan + 3 * (size_t) l2) : V3(0.f, 0.f, 0.f));
        // Mesh::moeller_trumbore (mesh.h:430-453) in dual arithmetic
        D3 e1 = p1 - p0, e2 = p2 - p0, pvec = dcross(d, e2);
        Df inv_det = DF(1.f) / ddot(e1, pvec);
        D3 tvec = o - p0;
        Df b1 = ddot(tvec, pvec) * inv_det;
        D3 qvec = dcross(tvec, e1);
        Df b2 = ddot(d, qvec) * inv_det;
        Df b0 = DF(1.f) - b1 - b2;
        h.p = p0 * b0 + p1 * b1 + p2 * b2;                                  // mesh.cpp:1448
        h.ng = dnormalize(dcross(e1, e2));
        if (sh.flags & DSHAPE_HAS_UV) {
            const float *t0 = sc.texcoords + 2 * (size_t) i0, *t1 = sc.texcoords + 2 * (size_t) i1, *t2 = sc.texcoords + 2 * (size_t) i2;
            const float du0x = t1[0] - t0[0], du0y = t1[1] - t0[1], du1x = t2[0] - t0[0], du1y = t2[1] - t0[1];
            const float det = du0x * du1y - du0y * du1x;
            if (det != 0.f) { dp_du = (e1 * du1y - e2 * du0y) * (1.f / det); have_dpdu = true; }
        }
        if (!have_dpdu) { D3 tt; dcoordinate_system(h.ng, dp_du, tt); have_dpdu = true; }
        if (sh.flags & DSHAPE_HAS_NRM) {
            const bool nmov = moving && gt.ntan != nullptr;
            D3 n0 = D3vd(ld3(sc.normals + 3 * (size_t) i0), nmov ? ld3(gt.ntan + 3 * (size_t) l0) : V3(0.f, 0.f, 0.f));
            D3 n1 = D3vd(ld3(sc.normals + 3 * (size_t) i1), nmov ? ld3(gt.ntan + 3 * (size_t) l1) : V3(0.f, 0.f, 0.f));
            D3 n2 = D3vd(ld3(sc.normals + 3 * (size_t) i2), nmov ? ld3(gt.ntan + 3 * (size_t) l2) : V3(0.f, 0.f, 0.f));
            h.ns = dnormalize(n0 * b0 + n1 * b1 + n2 * b2);
        } else h.ns = h.ng;
        if (sh.flags & DSHAPE_FLIP) { h.ng = -h.ng; h.ns = -h.ns; }
    } else if (FULL && sh.type == MI355_SHAPE_SPHERE) {
        const v3 c = ld3(sh.frame_s); const float r = sh.frame_t[0];
        D3 oc = o - D3c(c);
        Df A = ddot(d, d), B = ddot(oc, d) * 2.f, C = ddot(oc, oc) - DF(r * r);
        Df disc = dsqrt(B * B - A * C * 4.f);
        Df ta = (-B - disc) / (A * 2.f), tb = (-B + disc) / (A * 2.f);
        Df t = fabsf(ta.v - hit.t) <= fabsf(tb.v - hit.t) ? ta : tb;       // the root the primal took
        h.p = o + d * t;
        h.ns = (h.p - D3c(c)) * (1.f / r);
        if (sh.flags & DSHAPE_FLIP) h.ns = -h.ns;
        h.ng = h.ns;
    } else {                                                                // rectangle / disk: a fixed plane
        const v3 n = ld3(sh.frame_n), q = V3(sh.to_world[3], sh.to_world[7], sh.to_world[11]);
        Df t = ddot(D3c(q) - o, D3c(n)) / ddot(d, D3c(n));
        h.p = o + d * t;
        h.ng = h.ns = D3c(si.n);
        if (sh.type == MI355_SHAPE_RECTANGLE) { dp_du = D3c(ld3(sh.frame_s)); have_dpdu = true; }
    }
    if (have_dpdu) {                                                        // interaction.h:245-255
        h.sh_s = dnormalize(dp_du - h.ns * ddot(h.ns, dp_du));
        h.sh_t = dcross(h.ns, h.sh_s);
    } else { h.sh_s = D3c(si.sh_s); h.sh_t = D3c(si.sh_t); }               // sphere / disk: frame taken as locally constant
    return h;
}

// Derivative of one normalised splat: value V (dual colour) at film position (px, py) (dual, in film pixels).
MI_DEV void film_put_dual(const DScene &sc, float *gfilm, Df px, Df py, D3 V) {
    const bool box = sc.rfilter == MI355_RFILTER_BOX;
    const int n = box ? 0 : (int) ceilf(sc.filter_radius - .5f), count = 2 * n + 1;
    const int32_t pix = (int32_t) floorf(px.v) - n, piy = (int32_t) floorf(py.v) - n;
    Df wx[MI_MAX_TAPS], wy[MI_MAX_TAPS], sx = DF(0.f), sy = DF(0.f);
#pragma unroll
    for (int i = 0; i < MI_MAX_TAPS; ++i) {
        wx[i] = wy[i] = DF(0.f);
        if (i < count) {
            wx[i] = box ? DF(1.f) : dfilter_eval(sc, DF((float) (pix + i) + .5f) - px);
            wy[i] = box ? DF(1.f) : dfilter_eval(sc, DF((float) (piy + i) + .5f) - py);
            sx = sx + wx[i]; sy = sy + wy[i];
        }
    }
    Df norm = sx * sy;
    if (norm.v == 0.f) return;
    Df factor = DF(1.f) / norm;
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
#pragma unroll
    for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
        for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
            const int32_t x = pix + tx - (int32_t) sc.crop_x, y = piy + ty - (int32_t) sc.crop_y;
            if (ty >= count || tx >= count || x < 0 || y < 0 || x >= W || y >= H) continue;
            Df w = wy[ty] * wx[tx] * factor;
            const float r = (V.x * w).d, g = (V.y * w).d, b = (V.z * w).d;
            float *p = gfilm + ((size_t) y * W + x) * 4;
            if (r != 0.f) atomicAdd(p, r);
            if (g != 0.f) atomicAdd(p + 1, g);
            if (b != 0.f) atomicAdd(p + 2, b);
        }
}

// The same splat, deferred (film_put_deferred): the record carries what the tile pass needs to re-evaluate the dual weights --
// (x, y, dV.rgb | V.rgb, dx, dy) = 40 bytes in two halves.  One wave-aggregated counter atomic per wave; a full queue falls back to
// the direct atomics above.  Called from inside the connection branch: the ballot covers exactly the lanes that splat.
MI_DEV void film_put_deferred_dual(const DScene &sc, const SplatQueue &q, float *gfilm, Df px, Df py, D3 V) {
    bool put;
    {
        int n = sc.rfilter == MI355_RFILTER_BOX ? 0 : (int) ceilf(sc.filter_radius - .5f);
        int32_t cx = (int32_t) floorf(px.v) - (int32_t) sc.crop_x, cy = (int32_t) floorf(py.v) - (int32_t) sc.crop_y;
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
        q.rec_a[idx] = make_float4(px.v, py.v, V.x.d, V.y.d); q.rec_b[idx] = V.z.d;
        q.rec_c[idx] = make_float4(V.x.v, V.y.v, V.z.v, px.d); q.rec_d[idx] = py.d;
    } else film_put_dual(sc, gfilm, px, py, V);
}

// One light sample: the control flow of ptracer_sample (same sampler calls, same decisions), with the geometry duals alongside.
template <int FULL>
MI_DEV void ptracer_sample_geo(const DScene &sc, const GeoTangent &gt, Pcg32 &rng, int32_t max_depth, int32_t rr_depth, bool hide_emitters, bool lane_on,
                               float sample_scale, float *gfilm, uint32_t *stk, const SplatQueue *queue = nullptr) {
    const bool timed = sc.shutter_open_time > 0.f;
    if (max_depth != 0 && !hide_emitters) {                                // sample_visible_emitters: static emitter, static sensor -> no derivative,
        if (timed) (void) next_1d(rng);                                     // but the sampler stream advances exactly as in ptracer_sample
        float es = next_1d(rng), iw; uint32_t ei = pick_emitter(sc, es, iw);
        const bool connect = !FULL || sc.emitters[ei].type == MI355_EMITTER_AREA;
        if (connect) { (void) next_1d(rng); (void) next_1d(rng); (void) next_1d(rng); (void) next_1d(rng); }
        (void) next_1d(rng); (void) next_1d(rng);
        if (connect) (void) next_1d(rng);
    }
    if (timed) (void) next_1d(rng);
    float wav = next_1d(rng), idx_weight; uint32_t ei = pick_emitter(sc, wav, idx_weight);
    float ax = next_1d(rng), ay = next_1d(rng), bx = next_1d(rng), by = next_1d(rng);
    const DEmitter &em = sc.emitters[ei]; const DShape &esh = sc.shapes[em.shape];
    v3 ep, en; float epdf; shape_sample_position<FULL>(sc, esh, ax, ay, ep, en, epdf);
    SI esi = si_from_position(ep, en, (int32_t) em.shape);
    Ray ray = spawn_ray(esi.p, esi.n, si_to_world(esi, square_to_cosine_hemisphere(bx, by)));
    v3 throughput = ld3(em.radiance) * ((epdf > 0.f ? 1.f / epdf : 0.f) * MI_PI * idx_weight);
    if (FULL && em.type == MI355_EMITTER_DIRECTIONALAREA) {
        ray = spawn_ray(esi.p, esi.n, esi.n);
        throughput = ld3(em.radiance) * ((1.f / esh.inv_area) * idx_weight);
    }
    D3 o = D3c(ray.o), d = D3c(ray.d), T = D3c(throughput);                 // the emitter does not move
    bool active = lane_on && max3(throughput) != 0.f;
    float eta = 1.f; int32_t depth = 1;
    Hit hit = scene_intersect<FULL>(sc, ray, active, stk);
    active = active && hit.t != MI_INF;
    if (max_depth >= 0) active = active && depth < max_depth;
    while (__any(active)) {
        MI_SCENE_RELOAD(scb); const DScene &sc = scb;
        bool next_active = false;
        Ray next_ray = ray;
        if (active) {
            SI si = compute_si<FULL>(sc, ray, hit);
            const DBsdf &bsdf = sc.bsdfs[sc.shapes[hit.shape].bsdf];
            DualHit dh = dual_hit<FULL>(sc, gt, hit, si, o, d);
            D3 md = -d;
            D3 wi = { ddot(md, dh.sh_s), ddot(md, dh.sh_t), ddot(md, dh.ns) };
            const v3 refl = bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance);
            // ---- connection to the sensor (connect_sensor, ptracer.cpp:286-371; PerspectiveCamera::sample_direction / importance)
            float ssx = next_1d(rng), ssy = next_1d(rng);
            v2 uv; uv.x = uv.y = 0.f; v3 ds_p = V3(0.f, 0.f, 0.f); float sensor_weight = 0.f;
            bool seen = sensor_sample_direction<FULL>(sc, si.p, ssx, ssy, uv, ds_p, sensor_weight);
            const bool diffuse = !FULL || bsdf.type == MI355_BSDF_DIFFUSE;
            bool want = seen && diffuse && sensor_weight != 0.f && max3(throughput) != 0.f;
            Ray sensor_ray = spawn_ray_to(si.p, si.n, ds_p);
            if (!want) { sensor_ray.o = sensor_ray.d = V3(0.f, 0.f, 1.f); sensor_ray.maxt = 0.f; }
            bool occluded = scene_ray_test<FULL>(sc, sensor_ray, want, stk);
            if (want && !occluded) {
                const float *tw = sc.cam_to_world;
                D3 rel = dh.p - D3c(V3(tw[3], tw[7], tw[11]));
                D3 ref_p = { rel.x * tw[0] + rel.y * tw[4] + rel.z * tw[8], rel.x * tw[1] + rel.y * tw[5] + rel.z * tw[9], rel.x * tw[2] + rel.y * tw[6] + rel.z * tw[10] };
                const float *m = sc.camera_to_sample;
                Df hx = ref_p.x * m[0] + ref_p.y * m[1] + ref_p.z * m[2] + DF(m[3]), hy = ref_p.x * m[4] + ref_p.y * m[5] + ref_p.z * m[6] + DF(m[7]);
                Df hw = ref_p.x * m[12] + ref_p.y * m[13] + ref_p.z * m[14] + DF(m[15]);
                Df px = hx / hw * (float) sc.crop_w + DF((float) sc.crop_x), py = hy / hw * (float) sc.crop_h + DF((float) sc.crop_y);
                Df dist2 = ddot(ref_p, ref_p), dist = dsqrt(dist2);
                Df ct = ref_p.z / dist;
                Df sw = DF(sc.normalization) / (ct * ct * ct) / dist2;       // importance / dist^2 (the image-rectangle test is a discrete choice)
                D3 to_sensor = dnormalize(D3c(ds_p) - dh.p);
                D3 local_d = { ddot(to_sensor, dh.sh_s), ddot(to_sensor, dh.sh_t), ddot(to_sensor, dh.ns) };
                // Adjoint-BSDF correction for shading normals and the diffuse lobe, as connect_sensor evaluates them
                Df wi_dot_geo_n = ddot(dh.ng, md), wo_dot_geo_n = ddot(dh.ng, to_sensor);
                const bool valid = wi_dot_geo_n.v * wi.z.v > 0.f && wo_dot_geo_n.v * local_d.z.v > 0.f && wi.z.v > 0.f && local_d.z.v > 0.f;
                if (valid) {
                    Df correction = dabs((wi.z * wo_dot_geo_n) / (local_d.z * wi_dot_geo_n));
                    Df lobe = local_d.z * MI_INV_PI * correction * sw * sample_scale;
                    D3 V = (T * D3c(refl)) * lobe;
                    if (queue && queue->cap) film_put_deferred_dual(sc, *queue, gfilm, px, py, V);
                    else film_put_dual(sc, gfilm, px, py, V);
                }
            }
            // ---- BSDF sampling: the primal decides, the duals follow
            float s1 = next_1d(rng), s2x = next_1d(rng), s2y = next_1d(rng);
            BSample bs; v3 bsdf_val; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, true, bs, bsdf_val, true);
            v3 wo_world = si_to_world(si, bs.wo);
            float wi_dot_geo_n = dot3(si.n, -ray.d), wo_dot_geo_n = dot3(si.n, wo_world);
            next_active = wi_dot_geo_n * si.wi.z > 0.f && wo_dot_geo_n * bs.wo.z > 0.f;
            float correction = fabsf((si.wi.z * wo_dot_geo_n) / (bs.wo.z * wi_dot_geo_n));
            D3 wo;                                                          // dual outgoing direction (world)
            if (bs.delta) {
                if (bs.wo.z * si.wi.z > 0.f) wo = d + dh.ns * (wi.z * 2.f);   // reflection: d + 2 (wi . n) n
                else {                                                      // refraction (dielectric.cpp:246-371, fresnel.h:25-81): eta_ti d + n (eta_ti cos_i + cos_t)
                    const float rel = bsdf.rel_eta;
                    const float drel = sc.shapes[hit.shape].bsdf == gt.bsdf ? gt.d_rel_eta : 0.f;   // the IOR ratio is the design scalar's material part
                    Df eta_ti = si.wi.z > 0.f ? DF(1.f / rel, -drel / (rel * rel)) : DF(rel, drel);
                    Df ct2 = DF(1.f) - (DF(1.f) - wi.z * wi.z) * (eta_ti * eta_ti);
                    Df cos_t = dsqrt(ct2); if (si.wi.z > 0.f) cos_t = -cos_t;  // mulsign(sqrt, -cos_theta_i)
                    wo = d * eta_ti + dh.ns * (wi.z * eta_ti + cos_t);
                }
            } else wo = dh.sh_s * bs.wo.x + dh.sh_t * bs.wo.y + dh.ns * bs.wo.z;   // sampled local direction, carried by the dual frame
            Df dcorr = DF(correction);
            if (next_active) {
                Df a = wi.z * ddot(dh.ng, wo), b = (bs.delta ? ddot(wo, dh.ns) : DF(bs.wo.z)) * ddot(dh.ng, md);
                dcorr = dabs(a / b);
            }
            throughput = throughput * (bsdf_val * correction);
            D3 W = D3c(bsdf_val);
            if (FULL && bsdf.type == MI355_BSDF_CONDUCTOR && bs.delta) {    // the mirror's weight is its Fresnel term: d F / d cos(theta_i) by a central difference
                const float c = si.wi.z, hh = 1e-3f;
                v3 Fp = V3(fresnel_conductor_(c + hh, bsdf.eta[0], bsdf.k[0]), fresnel_conductor_(c + hh, bsdf.eta[1], bsdf.k[1]), fresnel_conductor_(c + hh, bsdf.eta[2], bsdf.k[2]));
                v3 Fm = V3(fresnel_conductor_(c - hh, bsdf.eta[0], bsdf.k[0]), fresnel_conductor_(c - hh, bsdf.eta[1], bsdf.k[1]), fresnel_conductor_(c - hh, bsdf.eta[2], bsdf.k[2]));
                v3 dF = ((Fp - Fm) * (0.5f / hh)) * ld3(bsdf.reflectance);
                W = D3vd(bsdf_val, dF * wi.z.d);
            }
            if (FULL && bsdf.type == MI355_BSDF_DIELECTRIC && bs.delta) {  // weight = t_i / detach(t_i) (or r_i / detach(r_i)): value 1, gradient d(t_i) / t_i (dielectric.cpp:343-351)
                const bool refl_lobe = bs.wo.z * si.wi.z > 0.f;
                const float drel = sc.shapes[hit.shape].bsdf == gt.bsdf ? gt.d_rel_eta : 0.f;
                ga::V<float> dWc, dWe; ga::dielectric_weight_partials(si.wi.z, bsdf.rel_eta, refl_lobe, ga::mk<float>(bsdf_val.x, bsdf_val.y, bsdf_val.z), dWc, dWe);
                W = D3vd(bsdf_val, V3(dWc.x, dWc.y, dWc.z) * wi.z.d + V3(dWe.x, dWe.y, dWe.z) * drel);
            }
            T = (T * W) * dcorr;
            eta *= bs.eta;
            next_active = next_active && (throughput.x != 0.f || throughput.y != 0.f || throughput.z != 0.f);
            next_ray = spawn_ray(si.p, si.n, wo_world);
            // spawn_ray: origin = p + n * (+-mag) with the primal's constant offset magnitude
            float mag = (1.f + fmaxf(fmaxf(fabsf(si.p.x), fabsf(si.p.y)), fabsf(si.p.z))) * MI_RAY_EPS;
            mag = mulsign(mag, dot3(si.n, wo_world));
            o = dh.p + dh.ng * mag; d = wo;
        }
        active = next_active; ray = next_ray;
        if (max_depth >= 0 && depth + 1 >= max_depth) active = false;       // the reference's last intersection is never read (see ptracer_sample)
        else hit = scene_intersect<FULL>(sc, ray, active, stk);
        depth++;
        if (max_depth >= 0) active = active && depth < max_depth;
        active = active && hit.t != MI_INF;
        if (depth > rr_depth) {
            float q = fminf(max3(throughput) * (eta * eta), 0.95f);
            if (active) active = next_1d(rng) < q;
            throughput = throughput * (1.f / q); T = T * (1.f / q);         // roulette probability detached
        }
    }
}
