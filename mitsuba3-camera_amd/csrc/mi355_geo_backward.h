// mi355_geo_backward.h -- REVERSE-mode geometry / IOR adjoint of the particle tracer: Integrator::render_backward
// (integrator.cpp:65-84) over ptracer.cpp:200-371 with respect to the `vertex_positions` of every mesh and the int_ior / ext_ior
// of every dielectric -- what BASELINE config 5 (notebooks/doublet_full_optim: design_problem.py:508-549) back-propagates through.
//
// One lane = one light-sample stream, as in k_ptracer.  Per sample:
//   1. FORWARD walk with the control flow, sampler calls and decisions of ptracer_sample; at every surface vertex the state that
//      ENTERS it is appended to the lane's PATH TAPE in HBM: ray (o, d), throughput T, the preliminary hit (t, u, v, prim, shape), the
//      bsdf weight W, the roulette factor that follows, and three flags (lobe taken, connection made, path went on).  The tape is SoA
//      over lanes ([vertex][field][lane], 18 dwords per vertex): consecutive lanes write consecutive addresses.
//   2. BACKWARD sweep over the tape, last vertex first, carrying lambda = adjoint of (o, d, T) leaving the vertex (zero behind the
//      last one).  Per vertex: rebuild the shading record with the primal code (compute_si), gather dLoss/dfilm and its derivative
//      with respect to the film position at the connection's taps, run ga::transport_adjoint and the shape's hit adjoint
//      (mi355_geo_adjoint.h), scatter the triangle's position adjoint into the gradient buffer and its vertex-normal adjoint into a
//      per-vertex scratch image, add the IOR adjoint to the workgroup's LDS accumulators.
// Afterwards k_vertex_normals_bwd carries the vertex-normal adjoints of every mesh whose normals are functions of its positions
// (DSHAPE_NRM_DEP) through the angle-weighted normal computation into the position gradients.
// Limits (stated in mi355.h): perspective sensor; emitters and the sensor do not move; a rough-conductor vertex keeps its sampled
// local direction and weight; textures are looked up at the primal uv; on spheres and disks the tangent frame of a diffuse vertex
// is locally constant -- the same list as the forward mode (mi355_geo_forward.h), against which this pass is tested.
#pragma once
#include "mi355_geo_adjoint.h"

#define MI_TAPE_FIELDS 18
#define MI_TAPE_MAX_VERTS 32
#define MI_TF_LOBE_MASK 3u
#define MI_TF_CONN 4u
#define MI_TF_CONT 8u
#define MI_TF_CONN_K(k) ((k) == 0u ? MI_TF_CONN : (8u << (k)))   // several cameras (DSensorSet): camera k's connection bit (16, 32, ... 1024 for k = 1 ... 7)
#define MI_TF_CONN_ANY (MI_TF_CONN | 0x7f0u)
#define MI_TF_SHIFT 12                                            // flags below, shape index above (< 2^20 shapes)
static_assert(MI_MULTI_MAX <= 8, "one connection bit per camera below MI_TF_SHIFT");

struct GeoTape { float *base; uint32_t stride; };      // stride = lanes of the launch; field f of vertex k of lane i at base[(k * FIELDS + f) * stride + i]
struct GeoBwdSink {
    const float *grad_rgb;       // dLoss / d image (crop_h x crop_w x 3)
    float *d_grad;               // the concatenated parameter-gradient buffer
    float *d_nrm_adj;            // 3 floats per vertex of every mesh (DScene::normals layout): adjoint of the vertex normals
    float *s_ior;                // LDS: 2 floats per bsdf (adjoint of int_ior, ext_ior), or nullptr -> global atomics
    uint32_t *tab;               // LDS: the workgroup's vertex-adjoint table (vertex_adj_add), or nullptr -> global atomics
};
struct GeoGradSet { const float *g[MI_MULTI_MAX]; };   // several cameras in one pass: dLoss / d image of each (nullptr: that camera's image carries no adjoint)


// ---- vertex adjoints, summed per workgroup before they leave the CU ----------------------------------------------------------------
// Every mesh-vertex hit scatters 18 floats (position and normal adjoints of the triangle's three corners).  As device-scope float atomics
// those execute on the memory side of the fabric (8 XCDs, 8 L2s: an L2 cannot own the line), one ~47-byte HBM transaction each: on the lens
// doublet 83 M of them per launch, 3.9 GB written for a 12 MB gradient, and 3.5 of the kernel's 6.8 ms (profiles/r03_lens_ab.txt: the same
// build without the scatter takes 3.3 ms) -- and the sorted light samples make it worse, not better: the lanes of a wave hit the same few
// triangles, i.e. the same addresses.  That is also the way out: a workgroup's 256 paths touch a few dozen distinct vertices, so they are
// summed in an LDS hash table first (key = global vertex index; position adjoint 3 floats, normal adjoint 3 floats, SoA so that the six
// adds of a lane fall into six different 4 KB planes) and one global atomic per (vertex, component) leaves at the end.  The table lives in
// the traversal stack's LDS, which the backward sweep does not use; walk and sweep are separated by barriers (k_ptracer_geo_bwd).  A
// vertex that finds no slot within MI_VTAB_PROBES steps goes out directly, as before.
#define MI_VTAB_LOG2 10
#define MI_VTAB (1u << MI_VTAB_LOG2)        // 1024 entries x 8 words = 32 KB = the four waves' BVH stacks (MI_BVH_STACK = 32)
#define MI_VTAB_PROBES 8
#define MI_VTAB_EMPTY 0xffffffffu
MI_DEV void vertex_adj_add(const GeoBwdSink &sink, uint32_t vid, uint32_t gaddr, ga::V<float> ap, bool with_n, ga::V<float> an) {
    if (sink.tab) {
        uint32_t *keys = sink.tab, *addrs = sink.tab + MI_VTAB; float *vals = reinterpret_cast<float *>(sink.tab + 2u * MI_VTAB);
        uint32_t h = (vid * 2654435761u) >> (32 - MI_VTAB_LOG2);
        for (int probe = 0; probe < MI_VTAB_PROBES; ++probe, h = (h + 1u) & (MI_VTAB - 1u)) {
            const uint32_t prev = atomicCAS(&keys[h], MI_VTAB_EMPTY, vid);
            if (prev == MI_VTAB_EMPTY || prev == vid) {
                if (prev == MI_VTAB_EMPTY) addrs[h] = gaddr;                 // (read only after the barrier in front of the flush)
                atomicAdd(&vals[h], ap.x); atomicAdd(&vals[MI_VTAB + h], ap.y); atomicAdd(&vals[2u * MI_VTAB + h], ap.z);
                if (with_n) { atomicAdd(&vals[3u * MI_VTAB + h], an.x); atomicAdd(&vals[4u * MI_VTAB + h], an.y); atomicAdd(&vals[5u * MI_VTAB + h], an.z); }
                return;
            }
        }
    }
    float *g = sink.d_grad + gaddr;
    atomicAdd(g, ap.x); atomicAdd(g + 1, ap.y); atomicAdd(g + 2, ap.z);
    if (with_n) { float *gn = sink.d_nrm_adj + 3 * (size_t) vid; atomicAdd(gn, an.x); atomicAdd(gn + 1, an.y); atomicAdd(gn + 2, an.z); }
}
MI_DEV void vertex_adj_clear(uint32_t *tab) {                               // all 256 threads
    for (uint32_t k = threadIdx.x; k < MI_VTAB; k += blockDim.x) {
        tab[k] = MI_VTAB_EMPTY;
#pragma unroll
        for (uint32_t c = 0; c < 6u; ++c) tab[(2u + c) * MI_VTAB + k] = 0u;
    }
}
MI_DEV void vertex_adj_flush(const GeoBwdSink &sink) {                      // all 256 threads, after a barrier
    const uint32_t *keys = sink.tab, *addrs = sink.tab + MI_VTAB; const float *vals = reinterpret_cast<const float *>(sink.tab + 2u * MI_VTAB);
    for (uint32_t k = threadIdx.x; k < MI_VTAB; k += blockDim.x) {
        const uint32_t vid = keys[k];
        if (vid == MI_VTAB_EMPTY) continue;
        float *g = sink.d_grad + addrs[k], *gn = sink.d_nrm_adj + 3 * (size_t) vid;
#pragma unroll
        for (uint32_t c = 0; c < 3u; ++c) {
            const float a = vals[c * MI_VTAB + k], b = vals[(3u + c) * MI_VTAB + k];
            if (a != 0.f) atomicAdd(g + c, a);
            if (b != 0.f) atomicAdd(gn + c, b);
        }
    }
}

typedef ga::V<float> gv3;
MI_DEV gv3 G3(v3 a) { return ga::mk<float>(a.x, a.y, a.z); }
MI_DEV v3 U3(gv3 a) { return V3(a.x, a.y, a.z); }

// derivative of the reconstruction filter (same polynomial / tent as filter_eval)
MI_DEV float filter_eval_d(const DScene &sc, float x, float &dfdx) {
    if (sc.rfilter == MI355_RFILTER_GAUSSIAN) {
        const float x2 = x * x; float p = sc.filter_coeff[9], dp = 0.f;
#pragma unroll
        for (int i = 8; i >= 0; --i) { dp = fmaf(dp, x2, p); p = fmaf(p, x2, sc.filter_coeff[i]); }   // Horner with derivative in x^2
        if (!(p > 0.f)) { dfdx = 0.f; return 0.f; }
        dfdx = dp * 2.f * x; return p;
    }
    if (sc.rfilter == MI355_RFILTER_TENT) {
        const float inv_r = 1.f / sc.filter_radius, v = 1.f - fabsf(x) * inv_r;
        if (!(v > 0.f)) { dfdx = 0.f; return 0.f; }
        dfdx = x < 0.f ? inv_r : -inv_r; return v;
    }
    dfdx = 0.f; return 1.f;                                                 // box: one tap of weight 1, no dependence on the position
}
// Adjoint of the normalised splat (imageblock.cpp:460-474) at film position (px, py):  Gv = sum_taps w dL/dfilm[tap] and its
// derivatives with respect to px and py (through the filter weights and their normalisation).
MI_DEV void film_gather_adj(const DScene &sc, const float *grad_rgb, float px, float py, gv3 &Gv, gv3 &Gx, gv3 &Gy) {
    Gv = Gx = Gy = ga::mk<float>(0.f, 0.f, 0.f);
    const bool box = sc.rfilter == MI355_RFILTER_BOX;
    const int n = box ? 0 : (int) ceilf(sc.filter_radius - .5f), count = 2 * n + 1;
    const int32_t pix = (int32_t) floorf(px) - n, piy = (int32_t) floorf(py) - n;
    float wx[MI_MAX_TAPS], wy[MI_MAX_TAPS], dwx[MI_MAX_TAPS], dwy[MI_MAX_TAPS], sx = 0.f, sy = 0.f, dsx = 0.f, dsy = 0.f;
#pragma unroll
    for (int i = 0; i < MI_MAX_TAPS; ++i) {
        wx[i] = wy[i] = dwx[i] = dwy[i] = 0.f;
        if (i < count) {
            float d;
            wx[i] = filter_eval_d(sc, (float) (pix + i) + .5f - px, d); dwx[i] = -d;     // d/dpx f(x_i + .5 - px)
            wy[i] = filter_eval_d(sc, (float) (piy + i) + .5f - py, d); dwy[i] = -d;
            sx += wx[i]; sy += wy[i]; dsx += dwx[i]; dsy += dwy[i];
        }
    }
    if (!(sx * sy != 0.f)) return;
    const float isx = 1.f / sx, isy = 1.f / sy;
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
#pragma unroll
    for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
        for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
            const int32_t x = pix + tx - (int32_t) sc.crop_x, y = piy + ty - (int32_t) sc.crop_y;
            if (ty >= count || tx >= count || x < 0 || y < 0 || x >= W || y >= H) continue;
            const float nx = wx[tx] * isx, ny = wy[ty] * isy;               // normalised separable weights
            const float dnx = (dwx[tx] - nx * dsx) * isx, dny = (dwy[ty] - ny * dsy) * isy;
            const float *g = grad_rgb + ((size_t) y * W + x) * 3;
            const gv3 gg = ga::mk<float>(g[0], g[1], g[2]);
            Gv = Gv + gg * (nx * ny); Gx = Gx + gg * (dnx * ny); Gy = Gy + gg * (nx * dny);
        }
}

MI_DEV void tape_store(const GeoTape &tp, uint32_t lane, uint32_t k, const Ray &ray, v3 T, const Hit &hit, v3 W, float rr, uint32_t flags) {
    float *b = tp.base + (size_t) k * MI_TAPE_FIELDS * tp.stride + lane;
    const size_t s = tp.stride;
    b[0] = ray.o.x; b[s] = ray.o.y; b[2 * s] = ray.o.z; b[3 * s] = ray.d.x; b[4 * s] = ray.d.y; b[5 * s] = ray.d.z;
    b[6 * s] = T.x; b[7 * s] = T.y; b[8 * s] = T.z; b[9 * s] = hit.t; b[10 * s] = hit.u; b[11 * s] = hit.v;
    b[12 * s] = W.x; b[13 * s] = W.y; b[14 * s] = W.z; b[15 * s] = rr;
    b[16 * s] = __uint_as_float(hit.prim); b[17 * s] = __uint_as_float((hit.shape << MI_TF_SHIFT) | flags);
}

// One light sample, forward walk: mirrors ptracer_sample call for call and writes one tape record per vertex.  Returns the number of
// vertices on the tape.
template <int FULL, bool MULTI = false>
MI_DEV uint32_t ptracer_geo_walk(const DScene &sc, Pcg32 &rng, int32_t max_depth, int32_t rr_depth, bool hide_emitters, bool lane_on,
                                 uint32_t *stk, const GeoTape &tp, uint32_t lane, uint32_t &n_sweep, const DSensorSet *set = nullptr) {
    // n_sweep (out): 1 + the index of the LAST vertex that made a sensor connection.  The image adjoint enters a path only at its connections,
    // so every vertex behind the last one has a zero adjoint and the sweep starts there -- for a light path that never reached this sensor's
    // view (the light of the OTHER field sources of a lens-design scene: two paths in three) it does not start at all.
    n_sweep = 0u;
    const bool timed = sc.shutter_open_time > 0.f;
    if (max_depth != 0 && !hide_emitters) {                                // sample_visible_emitters: static emitter and sensor -> no derivative;
        if (timed) (void) next_1d(rng);                                     // the stream advances exactly as in ptracer_sample
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
    bool active = lane_on && max3(throughput) != 0.f;
    float eta = 1.f; int32_t depth = 1; uint32_t nv = 0;
    Hit hit = scene_intersect<FULL>(sc, ray, active, stk);
    active = active && hit.t != MI_INF;
    if (max_depth >= 0) active = active && depth < max_depth;
    // ---------------------------------------------------------------- forward walk
    while (__any(active)) {
        MI_SCENE_RELOAD(scb); const DScene &sc = scb;
        bool next_active = false; Ray next_ray = ray; v3 T_in = throughput; v3 W = V3(0.f, 0.f, 0.f); uint32_t flags = 0u;
        if (active) {
            SI si = compute_si<FULL>(sc, ray, hit);
            const DBsdf &bsdf = sc.bsdfs[sc.shapes[hit.shape].bsdf];
            const v3 refl = bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance);
            float ssx = next_1d(rng), ssy = next_1d(rng);
            v2 uv; uv.x = uv.y = 0.f; v3 ds_p = V3(0.f, 0.f, 0.f); float sensor_weight = 0.f;
            const bool diffuse = !FULL || bsdf.type == MI355_BSDF_DIFFUSE;
            for (uint32_t cam = 0; cam < (MULTI ? set->n : 1u); ++cam) {     // one connection test per camera (one camera unless MULTI)
                bool seen = MULTI ? sensor_view_sample_direction(set->v[cam], set->film_w, set->film_h, si.p, uv, ds_p, sensor_weight)
                                  : sensor_sample_direction<FULL>(sc, si.p, ssx, ssy, uv, ds_p, sensor_weight);
                bool want = seen && diffuse && sensor_weight != 0.f && max3(throughput) != 0.f;
                Ray sensor_ray = spawn_ray_to(si.p, si.n, ds_p);
                if (!want) { sensor_ray.o = sensor_ray.d = V3(0.f, 0.f, 1.f); sensor_ray.maxt = 0.f; }
                bool occluded = scene_ray_test<FULL>(sc, sensor_ray, want, stk);
                if (want && !occluded) {
                    const v3 local_d = si_to_local(si, sensor_ray.d);
                    const float wi_g = dot3(si.n, -ray.d), wo_g = dot3(si.n, sensor_ray.d);
                    if (wi_g * si.wi.z > 0.f && wo_g * local_d.z > 0.f && si.wi.z > 0.f && local_d.z > 0.f) flags |= MI_TF_CONN_K(cam);
                }
            }
            float s1 = next_1d(rng), s2x = next_1d(rng), s2y = next_1d(rng);
            BSample bs; v3 bsdf_val; bsdf_sample<FULL>(bsdf, refl, si.wi, s1, s2x, s2y, true, bs, bsdf_val, true);
            v3 wo_world = si_to_world(si, bs.wo);
            float wi_dot_geo_n = dot3(si.n, -ray.d), wo_dot_geo_n = dot3(si.n, wo_world);
            next_active = wi_dot_geo_n * si.wi.z > 0.f && wo_dot_geo_n * bs.wo.z > 0.f;
            float correction = fabsf((si.wi.z * wo_dot_geo_n) / (bs.wo.z * wi_dot_geo_n));
            throughput = throughput * (bsdf_val * correction);
            W = bsdf_val;
            flags |= bs.delta ? (bs.wo.z * si.wi.z > 0.f ? (uint32_t) ga::LOBE_REFLECT : (uint32_t) ga::LOBE_REFRACT) : (uint32_t) ga::LOBE_FRAME;
            eta *= bs.eta;
            next_active = next_active && (throughput.x != 0.f || throughput.y != 0.f || throughput.z != 0.f);
            if (next_active) flags |= MI_TF_CONT;
            next_ray = spawn_ray(si.p, si.n, wo_world);
        }
        const bool was_active = active; const Ray in_ray = ray; const Hit in_hit = hit;
        active = next_active; ray = next_ray;
        if (max_depth >= 0 && depth + 1 >= max_depth) active = false;       // the reference's last intersection is never read (see ptracer_sample)
        else hit = scene_intersect<FULL>(sc, ray, active, stk);
        depth++;
        if (max_depth >= 0) active = active && depth < max_depth;
        active = active && hit.t != MI_INF;
        float rr = 1.f;
        if (depth > rr_depth) {
            float q = fminf(max3(throughput) * (eta * eta), 0.95f);
            if (active) active = next_1d(rng) < q;
            rr = 1.f / q; throughput = throughput * rr;                     // roulette probability detached
        }
        if (was_active && nv < MI_TAPE_MAX_VERTS) { tape_store(tp, lane, nv, in_ray, T_in, in_hit, W, rr, flags); nv++; if (flags & MI_TF_CONN_ANY) n_sweep = nv; }
    }
    return nv;
}
// Backward sweep over the `nv` tape records of one light sample, last vertex first: gradients into the sink.  No ray is traced here.
template <int FULL, bool MULTI = false>
MI_DEV void ptracer_geo_sweep(const DScene &sc, float sample_scale, const GeoTape &tp, uint32_t lane, uint32_t nv, uint32_t n_sweep, const GeoBwdSink &sink,
                              const DSensorSet *set = nullptr, const GeoGradSet *gset = nullptr) {
    gv3 lo = ga::mk<float>(0.f, 0.f, 0.f), ld = lo, lT = lo;               // lambda: adjoint of the ray / throughput LEAVING the current vertex
    ga::Sensor<float> se;
    {
        const float *tw = sc.cam_to_world;
        se.R[0] = tw[0]; se.R[1] = tw[1]; se.R[2] = tw[2]; se.R[3] = tw[4]; se.R[4] = tw[5]; se.R[5] = tw[6]; se.R[6] = tw[8]; se.R[7] = tw[9]; se.R[8] = tw[10];
        se.cam_o = ga::mk<float>(tw[3], tw[7], tw[11]);
#pragma unroll
        for (int i = 0; i < 16; ++i) se.M[i] = sc.camera_to_sample[i];
        se.crop_w = (float) sc.crop_w; se.crop_h = (float) sc.crop_h; se.crop_x = (float) sc.crop_x; se.crop_y = (float) sc.crop_y;
        se.normalization = sc.normalization;
    }
    for (int32_t k = (int32_t) n_sweep - 1; k >= 0; --k) {                  // (n_sweep <= nv: see ptracer_geo_walk)
        const float *b = tp.base + (size_t) k * MI_TAPE_FIELDS * tp.stride + lane; const size_t s = tp.stride;
        Ray r; r.o = V3(b[0], b[s], b[2 * s]); r.d = V3(b[3 * s], b[4 * s], b[5 * s]); r.maxt = MI_LARGEST;
        const v3 T = V3(b[6 * s], b[7 * s], b[8 * s]);
        Hit h; h.t = b[9 * s]; h.u = b[10 * s]; h.v = b[11 * s];
        const v3 Wv = V3(b[12 * s], b[13 * s], b[14 * s]); const float rr = b[15 * s];
        h.prim = __float_as_uint(b[16 * s]); const uint32_t sf = __float_as_uint(b[17 * s]); h.shape = sf >> MI_TF_SHIFT;
        const uint32_t flags = sf & ((1u << MI_TF_SHIFT) - 1u);
        const DShape &sh = sc.shapes[h.shape];
        const DBsdf &bsdf = sc.bsdfs[sh.bsdf];
        const SI si = compute_si<FULL>(sc, r, h);
        ga::Shade<float> shd; shd.p = G3(si.p); shd.ng = G3(si.n); shd.ns = G3(si.sh_n); shd.sh_s = G3(si.sh_s); shd.sh_t = G3(si.sh_t);
        ga::Vertex<float> vx;
        vx.d = G3(r.d); vx.T = G3(T); vx.lobe = (int) (flags & MI_TF_LOBE_MASK);
        vx.W = G3(Wv); vx.dW = vx.dWe = ga::mk<float>(0.f, 0.f, 0.f); vx.rel_eta = bsdf.rel_eta; vx.rr = rr;
        vx.cont = (flags & MI_TF_CONT) != 0u && k + 1 < (int32_t) nv;       // (nothing downstream of the last vertex: its continuation has no adjoint)
        vx.conn = (flags & MI_TF_CONN) != 0u;
        vx.l = ga::mk<float>(0.f, 0.f, 1.f); vx.mag = 0.f;
        if (k + 1 < (int32_t) nv) {                                         // the direction the path left with = the next vertex's incoming direction
            const float *bn = b + (size_t) MI_TAPE_FIELDS * s;
            const v3 wo_w = V3(bn[3 * s], bn[4 * s], bn[5 * s]);
            vx.l = G3(si_to_local(si, wo_w));
            float mag = (1.f + fmaxf(fmaxf(fabsf(si.p.x), fabsf(si.p.y)), fabsf(si.p.z))) * MI_RAY_EPS;
            vx.mag = mulsign(mag, dot3(si.n, wo_w));
        }
        if (FULL && bsdf.type == MI355_BSDF_CONDUCTOR && vx.lobe == ga::LOBE_REFLECT) {   // the mirror's weight is its Fresnel term: slope by a central difference
            const float c = si.wi.z, hh = 1e-3f;
            v3 Fp = V3(fresnel_conductor_(c + hh, bsdf.eta[0], bsdf.k[0]), fresnel_conductor_(c + hh, bsdf.eta[1], bsdf.k[1]), fresnel_conductor_(c + hh, bsdf.eta[2], bsdf.k[2]));
            v3 Fm = V3(fresnel_conductor_(c - hh, bsdf.eta[0], bsdf.k[0]), fresnel_conductor_(c - hh, bsdf.eta[1], bsdf.k[1]), fresnel_conductor_(c - hh, bsdf.eta[2], bsdf.k[2]));
            vx.dW = G3(((Fp - Fm) * (0.5f / hh)) * ld3(bsdf.reflectance));
        }
        if (FULL && bsdf.type == MI355_BSDF_DIELECTRIC && vx.lobe != ga::LOBE_FRAME)      // weight = t_i / detach(t_i) (or r_i / detach(r_i)), dielectric.cpp:343-351
            ga::dielectric_weight_partials(si.wi.z, bsdf.rel_eta, vx.lobe == ga::LOBE_REFLECT, vx.W, vx.dW, vx.dWe);
        vx.refl = G3(Wv); vx.K = sample_scale * MI_INV_PI;
        vx.Gv = vx.Gx = vx.Gy = ga::mk<float>(0.f, 0.f, 0.f);
        ga::Shade<float> a; a.p = a.ng = a.ns = a.sh_s = a.sh_t = ga::mk<float>(0.f, 0.f, 0.f);
        gv3 ad = ga::mk<float>(0.f, 0.f, 0.f), aT = ad, ao = ad; float arel = 0.f;
        if (MULTI) {
            // transport_adjoint is linear in (lambda, G) and ACCUMULATES: the continuation's share goes in with camera 0, every further camera
            // adds its own connection (lambda = 0).  A camera without an image adjoint (gset->g[k] == nullptr) contributes nothing.
            if (flags & MI_TF_CONN_ANY) vx.refl = G3(bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance));
            const gv3 zero = ga::mk<float>(0.f, 0.f, 0.f);
            for (uint32_t cam = 0; cam < set->n; ++cam) {
                const bool conn = (flags & MI_TF_CONN_K(cam)) != 0u && gset->g[cam] != nullptr;
                if (cam != 0u && !conn) continue;
                const DSensorView &sv = set->v[cam];
                ga::Sensor<float> sk = se;
                sk.R[0] = sv.cam_to_world[0]; sk.R[1] = sv.cam_to_world[1]; sk.R[2] = sv.cam_to_world[2]; sk.R[3] = sv.cam_to_world[4]; sk.R[4] = sv.cam_to_world[5]; sk.R[5] = sv.cam_to_world[6];
                sk.R[6] = sv.cam_to_world[8]; sk.R[7] = sv.cam_to_world[9]; sk.R[8] = sv.cam_to_world[10];
                sk.cam_o = ga::mk<float>(sv.cam_to_world[3], sv.cam_to_world[7], sv.cam_to_world[11]);
#pragma unroll
                for (int i = 0; i < 16; ++i) sk.M[i] = sv.camera_to_sample[i];
                sk.normalization = sv.normalization;
                vx.conn = conn; vx.Gv = vx.Gx = vx.Gy = zero;
                if (conn) { float px, py; ga::film_position(sk, shd.p, px, py); film_gather_adj(sc, gset->g[cam], px, py, vx.Gv, vx.Gx, vx.Gy); }
                if (cam == 0u) ga::transport_adjoint(shd, vx, sk, lo, ld, lT, a, ad, aT, arel);
                else ga::transport_adjoint(shd, vx, sk, zero, zero, zero, a, ad, aT, arel);
            }
        } else {
        if (vx.conn) {
            vx.refl = G3(bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance));
            float px, py; ga::film_position(se, shd.p, px, py);
            film_gather_adj(sc, sink.grad_rgb, px, py, vx.Gv, vx.Gx, vx.Gy);
        }
        ga::transport_adjoint(shd, vx, se, lo, ld, lT, a, ad, aT, arel);
        }
        if (sh.type == MI355_SHAPE_MESH) {
            const uint32_t *fi = sc.faces + 3 * (size_t) (sh.tri_offset + h.prim);
            const uint32_t l0 = fi[0], l1 = fi[1], l2 = fi[2], i0 = sh.vtx_offset + l0, i1 = sh.vtx_offset + l1, i2 = sh.vtx_offset + l2;
            ga::Tri<float> t;
            t.p0 = G3(ld3(sc.positions + 3 * (size_t) i0)); t.p1 = G3(ld3(sc.positions + 3 * (size_t) i1)); t.p2 = G3(ld3(sc.positions + 3 * (size_t) i2));
            t.has_nrm = (sh.flags & DSHAPE_HAS_NRM) != 0u; t.flip = (sh.flags & DSHAPE_FLIP) != 0u; t.has_uv = (sh.flags & DSHAPE_HAS_UV) != 0u;
            t.n0 = t.n1 = t.n2 = ga::mk<float>(0.f, 0.f, 1.f);
            if (t.has_nrm) { t.n0 = G3(ld3(sc.normals + 3 * (size_t) i0)); t.n1 = G3(ld3(sc.normals + 3 * (size_t) i1)); t.n2 = G3(ld3(sc.normals + 3 * (size_t) i2)); }
            t.du0x = t.du0y = t.du1x = t.du1y = 0.f;
            if (t.has_uv) {
                const float *t0 = sc.texcoords + 2 * (size_t) i0, *t1 = sc.texcoords + 2 * (size_t) i1, *t2 = sc.texcoords + 2 * (size_t) i2;
                t.du0x = t1[0] - t0[0]; t.du0y = t1[1] - t0[1]; t.du1x = t2[0] - t0[0]; t.du1y = t2[1] - t0[1];
            }
            ga::TriAdj<float> ta;
            ga::mesh_hit_adjoint(t, G3(r.o), G3(r.d), a, ao, ad, ta);
#ifndef MI_GEO_NO_SCATTER                                                   // (attribution builds: tools/lens_ab.sh)
            if (sh.grad_offset != 0xffffffffu) {
                const bool wn = t.has_nrm && (sh.flags & DSHAPE_NRM_DEP);
                vertex_adj_add(sink, i0, sh.grad_offset + 3u * l0, ta.p0, wn, ta.n0);
                vertex_adj_add(sink, i1, sh.grad_offset + 3u * l1, ta.p1, wn, ta.n1);
                vertex_adj_add(sink, i2, sh.grad_offset + 3u * l2, ta.p2, wn, ta.n2);
            }
#else
            if (ta.p0.x + ta.p1.y + ta.p2.z + ta.n0.x + ta.n1.y + ta.n2.z == 1.2345e33f) atomicAdd(sink.d_grad, 1.f);     // keep the adjoint arithmetic alive
#endif
        } else if (FULL && sh.type == MI355_SHAPE_SPHERE) {
            ga::sphere_hit_adjoint(G3(ld3(sh.frame_s)), sh.frame_t[0], (sh.flags & DSHAPE_FLIP) != 0u, G3(r.o), G3(r.d), h.t, a.p, a.ns + a.ng, ao, ad);
        } else {                                                            // rectangle / disk: a fixed plane (frame taken as constant)
            ga::plane_hit_adjoint(ga::mk<float>(sh.to_world[3], sh.to_world[7], sh.to_world[11]), G3(ld3(sh.frame_n)), G3(r.o), G3(r.d), a.p, ao, ad);
        }
        if (arel != 0.f && bsdf.ior_grad_offset >= 0) {                     // rel_eta = int_ior / ext_ior
            const float ext = bsdf.ext_ior, g_int = arel / ext, g_ext = -arel * bsdf.rel_eta / ext;
            if (sink.s_ior) { atomicAdd(sink.s_ior + 2u * sh.bsdf, g_int); atomicAdd(sink.s_ior + 2u * sh.bsdf + 1u, g_ext); }
            else { atomicAdd(sink.d_grad + bsdf.ior_grad_offset, g_int); atomicAdd(sink.d_grad + bsdf.ior_grad_offset + 1, g_ext); }
        }
        lo = ao; ld = ad; lT = aT;
    }
}

// Walk + sweep in one call (the fused kernel: samples_per_pass > 1 re-uses the tape pass after pass).
template <int FULL>
MI_DEV void ptracer_sample_geo_bwd(const DScene &sc, Pcg32 &rng, int32_t max_depth, int32_t rr_depth, bool hide_emitters, bool lane_on,
                                   float sample_scale, uint32_t *stk, const GeoTape &tp, uint32_t lane, const GeoBwdSink &sink) {
    uint32_t n_sweep;
    const uint32_t nv = ptracer_geo_walk<FULL>(sc, rng, max_depth, rr_depth, hide_emitters, lane_on, stk, tp, lane, n_sweep);
#ifndef MI_GEO_NO_SWEEP
    ptracer_geo_sweep<FULL>(sc, sample_scale, tp, lane, nv, n_sweep, sink);
#endif
}

// ---- vertex normals: Mesh::recompute_vertex_normals (mesh.cpp:376-413, JIT branch), as a per-vertex GATHER over the incident
// (face, corner) pairs in ascending face order -- deterministic, no atomics.  adj_off / adj: CSR of (local face id * 4 + corner).
MI_DEV void vertex_normal_gather(const float *positions, const uint32_t *faces, const uint32_t *adj_off, const uint32_t *adj, uint32_t vtx_offset,
                                 uint32_t tri_offset, uint32_t v_global, v3 &n_out, float &len_out, const float *vtan = nullptr, v3 *dn_out = nullptr) {
    // vtan (optional): d(position)/d(theta) of the mesh's vertices (mesh-local vertex order) -> *dn_out = d(normal)/d(theta)
    gv3 N = ga::mk<float>(0.f, 0.f, 0.f), dN = N;
    for (uint32_t e = adj_off[v_global]; e < adj_off[v_global + 1u]; ++e) {
        const uint32_t f = adj[e] >> 2, corner = adj[e] & 3u;
        const uint32_t *fi = faces + 3 * (size_t) (tri_offset + f);
        const gv3 v0 = G3(ld3(positions + 3 * (size_t) (vtx_offset + fi[0]))), v1 = G3(ld3(positions + 3 * (size_t) (vtx_offset + fi[1]))),
                  v2 = G3(ld3(positions + 3 * (size_t) (vtx_offset + fi[2])));
        if (vtan) {
            gv3 term; const gv3 dt = ga::face_normal_term_jvp(v0, v1, v2, G3(ld3(vtan + 3 * (size_t) fi[0])), G3(ld3(vtan + 3 * (size_t) fi[1])), G3(ld3(vtan + 3 * (size_t) fi[2])), (int) corner, term);
            if (term.x == term.x) { N = N + term; dN = dN + dt; }
        } else {
            gv3 n; float ang[3]; ga::face_normal_terms(v0, v1, v2, n, ang);
            if (n.x == n.x) N = N + n * ang[corner];                        // (a degenerate face has no normal: skipped, as length_sqr > 0 does in the scalar branch)
        }
    }
    const float len = sqrtf(ga::dot(N, N));
    len_out = len;
    n_out = len != 0.f ? V3(N.x / len, N.y / len, N.z / len) : V3(1.f, 0.f, 0.f);
    if (dn_out) { const gv3 nn = G3(n_out); *dn_out = len != 0.f ? U3((dN - nn * ga::dot(nn, dN)) * (1.f / len)) : V3(0.f, 0.f, 0.f); }
}
