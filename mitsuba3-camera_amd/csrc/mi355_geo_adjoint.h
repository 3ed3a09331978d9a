// mi355_geo_adjoint.h -- REVERSE-mode derivative of one light-path vertex of the particle tracer with respect to geometry and IOR.
//
// What the fork's lens-design loop needs (notebooks/scripts/lens_design/design_problem.py:508-549: `mi.render` with the `ptracer`
// integrator, then dr.backward(loss) down to the lens scalars through `vertex_positions` and the IORs of the `dispersive` BSDFs):
// the reference gets it from Dr.Jit's AD of ptracer.cpp:200-371 (Integrator::render_backward, integrator.cpp:65-84); there is no
// derivative code to restate.  Here the adjoint is written out by hand, vertex by vertex:
//
//   state entering vertex k : ray (o, d), throughput T
//   geometry                : the hit triangle (p0, p1, p2), its vertex normals (n0, n1, n2), the dielectric's int_ior / ext_ior
//   outputs                 : the ray leaving the vertex (o', d'), the throughput T', and -- at a diffuse vertex that sees the sensor --
//                             the film contribution  c = sum_taps w(px, py) <dLoss/dfilm[tap], V>   (normalised splat, imageblock.cpp:460-474)
//
// and `transport_adjoint` + `mesh_hit_adjoint` map the adjoint of (o', d', T') and dc = 1 back to the adjoint of (o, d, T) and of the
// geometry.  The kernel walks the path forwards once (recording the state entering every vertex, mi355_engine.hip: k_ptracer_geo_bwd),
// then backwards with these functions.  Differentiated exactly as Dr.Jit would (the same list as mi355_geo_forward.h): Moeller-Trumbore
// barycentrics and hit point, geometric and interpolated shading normal, the Gram-Schmidt shading frame, refraction / reflection,
// the shading-normal correction of the throughput, the sensor connection (direction, distance, importance, cosines, film position
// through the derivative of the filter weights).  Not differentiated: visibility and discrete choices (triangle, lobe, roulette).
//
// The functions are templates over the scalar type and compile on the host as well: tests/geo_adjoint_check.cpp runs them in double
// against central differences of the forward functions below (no GPU needed), the oracle checks the assembled gradient with an
// independent forward-mode (multi-dual) evaluation.
#pragma once
#if defined(__HIPCC__)
#define GA_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define GA_HD inline
#endif

namespace ga {

template <typename F> struct V { F x, y, z; };
template <typename F> GA_HD V<F> mk(F x, F y, F z) { V<F> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename F> GA_HD V<F> operator+(V<F> a, V<F> b) { return mk<F>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename F> GA_HD V<F> operator-(V<F> a, V<F> b) { return mk<F>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename F> GA_HD V<F> operator-(V<F> a) { return mk<F>(-a.x, -a.y, -a.z); }
template <typename F> GA_HD V<F> operator*(V<F> a, F s) { return mk<F>(a.x * s, a.y * s, a.z * s); }
template <typename F> GA_HD V<F> operator*(V<F> a, V<F> b) { return mk<F>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename F> GA_HD F dot(V<F> a, V<F> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename F> GA_HD V<F> cross(V<F> a, V<F> b) { return mk<F>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <typename F> GA_HD F hsum(V<F> a) { return a.x + a.y + a.z; }
template <typename F> GA_HD F sqrt_(F x) { return (F) sqrt((double) x); }
template <> GA_HD float sqrt_<float>(float x) { return sqrtf(x); }
template <typename F> GA_HD F abs_(F x) { return x < (F) 0 ? -x : x; }
// y = x / |x|  ->  x_bar = (y_bar - y (y . y_bar)) / |x|
template <typename F> GA_HD V<F> normalize_adj(V<F> y, F len, V<F> ay) { return (ay - y * dot(y, ay)) * ((F) 1 / len); }

// ---- one mesh triangle -------------------------------------------------------------------------------------------
template <typename F> struct Tri {
    V<F> p0, p1, p2, n0, n1, n2;
    bool has_nrm, flip, has_uv;          // DSHAPE_HAS_NRM, DSHAPE_FLIP; has_uv: dp_du from the texture coordinates (mesh.cpp:1481-1492)
    F du0x, du0y, du1x, du1y;            // uv1 - uv0, uv2 - uv0
};
template <typename F> struct TriAdj { V<F> p0, p1, p2, n0, n1, n2; };
// the shading record the transport step reads: hit point, geometric normal, shading normal, tangent frame (flips applied)
template <typename F> struct Shade { V<F> p, ng, ns, sh_s, sh_t; };

// coordinate_system(n).s (vector.h:116-136) and its adjoint with respect to n
template <typename F> GA_HD V<F> cs_s(V<F> n) {
    const F sign = n.z >= (F) 0 ? (F) 1 : (F) -1, a = -(F) 1 / (sign + n.z), b = n.x * n.y * a;
    return mk<F>((n.x * n.x * a) * sign + (F) 1, b * sign, -sign * n.x);
}
template <typename F> GA_HD V<F> cs_s_adj(V<F> n, V<F> as) {
    const F sign = n.z >= (F) 0 ? (F) 1 : (F) -1, a = -(F) 1 / (sign + n.z);
    const F aa = as.x * sign * n.x * n.x + as.y * sign * n.x * n.y;         // d a / d nz = a^2
    return mk<F>(as.x * sign * (F) 2 * n.x * a + as.y * sign * n.y * a - as.z * sign, as.y * sign * n.x * a, aa * a * a);
}

// Mesh::ray_intersect_triangle (mesh.h:430-453) + Mesh::compute_surface_interaction (mesh.cpp:1375-1548) + the shading frame of
// finalize_surface_interaction (interaction.h:245-255), as functions of the ray and the triangle
template <typename F> GA_HD Shade<F> mesh_hit_forward(const Tri<F> &t, V<F> o, V<F> d) {
    const V<F> e1 = t.p1 - t.p0, e2 = t.p2 - t.p0, pvec = cross(d, e2);
    const F inv = (F) 1 / dot(e1, pvec);
    const V<F> tvec = o - t.p0, qvec = cross(tvec, e1);
    const F b1 = dot(tvec, pvec) * inv, b2 = dot(d, qvec) * inv, b0 = (F) 1 - b1 - b2;
    Shade<F> s;
    s.p = t.p0 * b0 + t.p1 * b1 + t.p2 * b2;
    const V<F> gn = cross(e1, e2), ngu = gn * ((F) 1 / sqrt_(dot(gn, gn)));
    V<F> nsu = ngu;
    if (t.has_nrm) { const V<F> m = t.n0 * b0 + t.n1 * b1 + t.n2 * b2; nsu = m * ((F) 1 / sqrt_(dot(m, m))); }
    V<F> dp_du = cs_s(ngu);
    if (t.has_uv) {
        const F det = t.du0x * t.du1y - t.du0y * t.du1x;
        if (det != (F) 0) dp_du = (e1 * t.du1y - e2 * t.du0y) * ((F) 1 / det);
    }
    const F fs = t.flip ? (F) -1 : (F) 1;
    s.ng = ngu * fs; s.ns = nsu * fs;
    const V<F> tt = dp_du - s.ns * dot(s.ns, dp_du);
    s.sh_s = tt * ((F) 1 / sqrt_(dot(tt, tt)));
    s.sh_t = cross(s.ns, s.sh_s);
    return s;
}
// adjoint of mesh_hit_forward: (p, ng, ns, sh_s, sh_t)_bar -> (o, d)_bar (ACCUMULATED) and the triangle's adjoint (SET)
template <typename F> GA_HD void mesh_hit_adjoint(const Tri<F> &t, V<F> o, V<F> d, const Shade<F> &a, V<F> &ao, V<F> &ad, TriAdj<F> &out) {
    // ---- primal, recomputed
    const V<F> e1 = t.p1 - t.p0, e2 = t.p2 - t.p0, pvec = cross(d, e2);
    const F det = dot(e1, pvec), inv = (F) 1 / det;
    const V<F> tvec = o - t.p0, qvec = cross(tvec, e1);
    const F u = dot(tvec, pvec), v = dot(d, qvec), b1 = u * inv, b2 = v * inv, b0 = (F) 1 - b1 - b2;
    const V<F> gn = cross(e1, e2); const F gl = sqrt_(dot(gn, gn)); const V<F> ngu = gn * ((F) 1 / gl);
    V<F> m = ngu, nsu = ngu; F ml = (F) 1;
    if (t.has_nrm) { m = t.n0 * b0 + t.n1 * b1 + t.n2 * b2; ml = sqrt_(dot(m, m)); nsu = m * ((F) 1 / ml); }
    bool uv_frame = false; F detuv = (F) 0;
    if (t.has_uv) { detuv = t.du0x * t.du1y - t.du0y * t.du1x; uv_frame = detuv != (F) 0; }
    const V<F> dp_du = uv_frame ? (e1 * t.du1y - e2 * t.du0y) * ((F) 1 / detuv) : cs_s(ngu);
    const F fs = t.flip ? (F) -1 : (F) 1;
    const V<F> ns = nsu * fs;
    const F q = dot(ns, dp_du);
    const V<F> tt = dp_du - ns * q; const F tl = sqrt_(dot(tt, tt)); const V<F> sh_s = tt * ((F) 1 / tl);
    // ---- adjoint
    V<F> ans = a.ns, ang = a.ng, ash_s = a.sh_s;
    // sh_t = ns x sh_s
    ans = ans + cross(sh_s, a.sh_t); ash_s = ash_s + cross(a.sh_t, ns);
    // sh_s = tt / |tt|, tt = dp_du - ns (ns . dp_du)
    const V<F> att = normalize_adj(sh_s, tl, ash_s);
    V<F> adp_du = att;
    ans = ans - att * q;
    const F aq = -dot(att, ns);
    ans = ans + dp_du * aq; adp_du = adp_du + ns * aq;
    V<F> ae1 = mk<F>(0, 0, 0), ae2 = mk<F>(0, 0, 0), angu = ang * fs;
    if (uv_frame) { ae1 = ae1 + adp_du * (t.du1y / detuv); ae2 = ae2 - adp_du * (t.du0y / detuv); }
    else angu = angu + cs_s_adj(ngu, adp_du);
    // ns = fs * normalize(m)  (or the geometric normal)
    F ab0 = (F) 0, ab1 = (F) 0, ab2 = (F) 0;
    out.n0 = out.n1 = out.n2 = mk<F>(0, 0, 0);
    const V<F> ansu = ans * fs;
    if (t.has_nrm) {
        const V<F> am = normalize_adj(nsu, ml, ansu);
        out.n0 = am * b0; out.n1 = am * b1; out.n2 = am * b2;
        ab0 += dot(am, t.n0); ab1 += dot(am, t.n1); ab2 += dot(am, t.n2);
    } else angu = angu + ansu;
    // ngu = gn / |gn|, gn = e1 x e2
    const V<F> agn = normalize_adj(ngu, gl, angu);
    ae1 = ae1 + cross(e2, agn); ae2 = ae2 + cross(agn, e1);
    // p = b0 p0 + b1 p1 + b2 p2
    out.p0 = a.p * b0; out.p1 = a.p * b1; out.p2 = a.p * b2;
    ab0 += dot(a.p, t.p0); ab1 += dot(a.p, t.p1); ab2 += dot(a.p, t.p2);
    // b0 = 1 - b1 - b2, b1 = u inv, b2 = v inv, inv = 1 / det
    ab1 -= ab0; ab2 -= ab0;
    const F au = ab1 * inv, av = ab2 * inv, adet = -(ab1 * u + ab2 * v) * inv * inv;
    // v = d . qvec, qvec = tvec x e1; u = tvec . pvec; det = e1 . pvec; pvec = d x e2; tvec = o - p0
    ad = ad + qvec * av;
    const V<F> aqvec = d * av;
    V<F> atvec = cross(e1, aqvec); ae1 = ae1 + cross(aqvec, tvec);
    atvec = atvec + pvec * au;
    const V<F> apvec = tvec * au + e1 * adet;
    ae1 = ae1 + pvec * adet;
    ad = ad + cross(e2, apvec); ae2 = ae2 + cross(apvec, d);
    ao = ao + atvec;
    out.p0 = out.p0 - atvec - ae1 - ae2; out.p1 = out.p1 + ae1; out.p2 = out.p2 + ae2;
}

// A fixed plane through q with unit normal n (rectangle / disk; the shape does not move): p = o + d t.  Only p depends on the ray.
template <typename F> GA_HD V<F> plane_hit_forward(V<F> q, V<F> n, V<F> o, V<F> d) { return o + d * (dot(q - o, n) / dot(d, n)); }
template <typename F> GA_HD void plane_hit_adjoint(V<F> q, V<F> n, V<F> o, V<F> d, V<F> ap, V<F> &ao, V<F> &ad) {
    const F den = dot(d, n), t = dot(q - o, n) / den;
    const F at = dot(ap, d), anum = at / den, aden = -at * t / den;
    ao = ao + ap - n * anum; ad = ad + ap * t + n * aden;
}
// A fixed sphere (centre c, radius r): p = o + d t with t the root the primal took (t_primal picks it), ns = +-(p - c) / r.
template <typename F> GA_HD void sphere_hit_forward(V<F> c, F r, bool flip, V<F> o, V<F> d, F t_primal, V<F> &p, V<F> &ns) {
    const V<F> oc = o - c; const F A = dot(d, d), B = (F) 2 * dot(oc, d), C = dot(oc, oc) - r * r, disc = sqrt_(B * B - (F) 4 * A * C);
    const F ta = (-B - disc) / ((F) 2 * A), tb = (-B + disc) / ((F) 2 * A), t = abs_(ta - t_primal) <= abs_(tb - t_primal) ? ta : tb;
    p = o + d * t; ns = (p - c) * ((flip ? (F) -1 : (F) 1) / r);
}
template <typename F> GA_HD void sphere_hit_adjoint(V<F> c, F r, bool flip, V<F> o, V<F> d, F t_primal, V<F> ap, V<F> ans, V<F> &ao, V<F> &ad) {
    const V<F> oc = o - c; const F A = dot(d, d), B = (F) 2 * dot(oc, d), C = dot(oc, oc) - r * r, disc = sqrt_(B * B - (F) 4 * A * C);
    const F ta = (-B - disc) / ((F) 2 * A), tb = (-B + disc) / ((F) 2 * A), t = abs_(ta - t_primal) <= abs_(tb - t_primal) ? ta : tb;
    ap = ap + ans * ((flip ? (F) -1 : (F) 1) / r);                          // ns = +-(p - c) / r
    // p = o + d t;  A t^2 + B t + C = 0  ->  dt = -(dA t^2 + dB t + dC) / (2 A t + B)
    const F at = dot(ap, d), k = -at / ((F) 2 * A * t + B), aA = k * t * t, aB = k * t, aC = k;
    ao = ao + ap; ad = ad + ap * t;
    ad = ad + d * ((F) 2 * aA) + oc * ((F) 2 * aB);
    ao = ao + d * ((F) 2 * aB) + oc * ((F) 2 * aC);
}

// ---- Fresnel reflectance of a dielectric interface with its two partial derivatives ---------------------------------------------
// SmoothDielectric::sample detaches the lobe choice and carries the Fresnel term's derivative in the weight:
//     weight = select(selected_r, r_i / detach(r_i), t_i / detach(t_i))       (dielectric.cpp:343-351, dispersive_dielectric.cpp:356-364)
// -- value 1, gradient d(r_i) / r_i or d(t_i) / t_i -- so the throughput depends on cos(theta_i) and on the IOR ratio through it.
template <typename F> struct D2 { F v, a, b; };                             // value, d / d cos(theta_i), d / d eta
template <typename F> GA_HD D2<F> d2(F v, F a = (F) 0, F b = (F) 0) { D2<F> r; r.v = v; r.a = a; r.b = b; return r; }
template <typename F> GA_HD D2<F> operator+(D2<F> x, D2<F> y) { return d2<F>(x.v + y.v, x.a + y.a, x.b + y.b); }
template <typename F> GA_HD D2<F> operator-(D2<F> x, D2<F> y) { return d2<F>(x.v - y.v, x.a - y.a, x.b - y.b); }
template <typename F> GA_HD D2<F> operator*(D2<F> x, D2<F> y) { return d2<F>(x.v * y.v, x.a * y.v + x.v * y.a, x.b * y.v + x.v * y.b); }
template <typename F> GA_HD D2<F> operator/(D2<F> x, D2<F> y) { const F i = (F) 1 / y.v, q = x.v * i; return d2<F>(q, (x.a - q * y.a) * i, (x.b - q * y.b) * i); }
template <typename F> GA_HD D2<F> d2sqrt(D2<F> x) { const F s = sqrt_(x.v > (F) 0 ? x.v : (F) 0), k = s > (F) 0 ? (F) 0.5 / s : (F) 0; return d2<F>(s, x.a * k, x.b * k); }
// fresnel(cos_theta_i, eta), include/mitsuba/render/fresnel.h:34-72: r and its partial derivatives
template <typename F> GA_HD void fresnel_partials(F cos_theta_i, F eta_, F &r, F &dr_dc, F &dr_deta) {
    const bool outside = cos_theta_i >= (F) 0;
    const D2<F> c = d2<F>(cos_theta_i, (F) 1, (F) 0), eta = d2<F>(eta_, (F) 0, (F) 1), one = d2<F>((F) 1);
    const D2<F> eta_it = outside ? eta : one / eta, eta_ti = outside ? one / eta : eta;
    const D2<F> ct2 = one - (one - c * c) * eta_ti * eta_ti;
    const D2<F> ca = cos_theta_i < (F) 0 ? d2<F>((F) 0) - c : c, cta = d2sqrt(ct2);
    const D2<F> a_s = (ca - eta_it * cta) / (ca + eta_it * cta), a_p = (cta - eta_it * ca) / (cta + eta_it * ca);
    const D2<F> rr = (a_s * a_s + a_p * a_p) * d2<F>((F) 0.5);
    r = rr.v; dr_dc = rr.a; dr_deta = rr.b;
    if (eta_ == (F) 1) { r = (F) 0; dr_dc = dr_deta = (F) 0; }
    else if (ca.v == (F) 0 || !(ct2.v > (F) 0)) { r = (F) 1; dr_dc = dr_deta = (F) 0; }       // grazing incidence / total internal reflection
}
// d W / d cos(theta_i) and d W / d eta of the weight W of a dielectric's sampled lobe (`reflect`: the reflected lobe was taken)
template <typename F> GA_HD void dielectric_weight_partials(F cos_theta_i, F eta, bool reflect, V<F> W, V<F> &dW, V<F> &dWe) {
    F r, dc, de; fresnel_partials(cos_theta_i, eta, r, dc, de);
    const F p = reflect ? r : (F) 1 - r, sg = reflect ? (F) 1 : (F) -1;
    const F kc = p > (F) 0 ? sg * dc / p : (F) 0, ke = p > (F) 0 ? sg * de / p : (F) 0;
    dW = W * kc; dWe = W * ke;
}

// ---- transport through one vertex ----------------------------------------------------------------------------------
enum { LOBE_FRAME = 0,       // a sampled LOCAL direction carried by the shading frame (diffuse; rough conductor: direction and weight kept)
       LOBE_REFRACT = 1,     // dielectric.cpp:246-371, transmitted lobe
       LOBE_REFLECT = 2 };   // mirror direction (dielectric reflection, conductor)
template <typename F> struct Sensor {      // PerspectiveCamera::sample_direction / importance, perspective.cpp:282-386
    F R[9];                  // camera-to-world rotation, row-major (the rigid frame of cam_to_world)
    V<F> cam_o;              // camera position
    F M[16];                 // camera_to_sample
    F crop_w, crop_h, crop_x, crop_y, normalization;
};
template <typename F> struct Vertex {
    V<F> d, T;               // incoming direction and throughput (the incoming origin only matters to the hit)
    int lobe; V<F> l;        // LOBE_FRAME: the sampled local direction
    V<F> W, dW, dWe;         // bsdf weight, d W / d cos(theta_i) and d W / d rel_eta (the Fresnel term carried by a conductor's / dielectric's weight)
    F rel_eta;               // dielectric: int_ior / ext_ior
    F mag;                   // signed ray-origin offset along ng (spawn_ray, interaction.h:158-162): taken as constant
    F rr;                    // 1 / q of the roulette step that FOLLOWS this vertex (1 if none); q is detached
    bool cont;               // the throughput correction was applied and the path went on (next_active)
    bool conn;               // this vertex made an unoccluded, valid connection to the sensor through a diffuse lobe
    V<F> refl;               // connection: albedo at the vertex
    F K;                     // connection: sample_scale / pi
    V<F> Gv, Gx, Gy;         // connection: sum_taps w dL/dfilm[tap], sum_taps dw/dpx dL/dfilm[tap], sum_taps dw/dpy dL/dfilm[tap]
};
template <typename F> struct TransportOut { V<F> o, d, T; F px, py, c; V<F> Vc; };

// film position of a world point (crop-relative pixels + crop offset), as connect_sensor passes it to ImageBlock::put
template <typename F> GA_HD void film_position(const Sensor<F> &se, V<F> p, F &px, F &py) {
    const V<F> rel = p - se.cam_o;
    const V<F> rp = mk<F>(se.R[0] * rel.x + se.R[3] * rel.y + se.R[6] * rel.z, se.R[1] * rel.x + se.R[4] * rel.y + se.R[7] * rel.z, se.R[2] * rel.x + se.R[5] * rel.y + se.R[8] * rel.z);
    const F hx = se.M[0] * rp.x + se.M[1] * rp.y + se.M[2] * rp.z + se.M[3], hy = se.M[4] * rp.x + se.M[5] * rp.y + se.M[6] * rp.z + se.M[7];
    const F hw = se.M[12] * rp.x + se.M[13] * rp.y + se.M[14] * rp.z + se.M[15];
    px = hx / hw * se.crop_w + se.crop_x; py = hy / hw * se.crop_h + se.crop_y;
}

// forward evaluation (host checks only; on the device the primal comes from ptracer_sample's own code)
template <typename F> GA_HD TransportOut<F> transport_forward(const Shade<F> &s, const Vertex<F> &vx, const Sensor<F> &se) {
    TransportOut<F> out; out.c = (F) 0; out.px = out.py = (F) 0; out.Vc = mk<F>(0, 0, 0);
    const V<F> md = -vx.d; const F wiz = dot(md, s.ns);
    if (vx.conn) {
        film_position(se, s.p, out.px, out.py);
        const V<F> rel = s.p - se.cam_o;
        const V<F> rp = mk<F>(se.R[0] * rel.x + se.R[3] * rel.y + se.R[6] * rel.z, se.R[1] * rel.x + se.R[4] * rel.y + se.R[7] * rel.z, se.R[2] * rel.x + se.R[5] * rel.y + se.R[8] * rel.z);
        const F dist = sqrt_(dot(rp, rp)), sw = se.normalization * dist / (rp.z * rp.z * rp.z);
        const V<F> x = se.cam_o - s.p, ts = x * ((F) 1 / sqrt_(dot(x, x)));
        const F lz = dot(ts, s.ns), wo_g = dot(s.ng, ts), wi_g = dot(s.ng, md), corr = abs_((wiz * wo_g) / (lz * wi_g));
        out.Vc = (vx.T * vx.refl) * (lz * vx.K * corr * sw);
        out.c = dot(vx.Gv, out.Vc);          // (the caller's Gv is evaluated AT (px, py); its dependence on them is Gx, Gy)
    }
    V<F> wo; F wolz;
    if (vx.lobe == LOBE_FRAME) { wo = s.sh_s * vx.l.x + s.sh_t * vx.l.y + s.ns * vx.l.z; wolz = vx.l.z; }
    else if (vx.lobe == LOBE_REFRACT) {
        const F eta_ti = wiz > (F) 0 ? (F) 1 / vx.rel_eta : vx.rel_eta, ct2 = (F) 1 - ((F) 1 - wiz * wiz) * eta_ti * eta_ti;
        const F cos_t = wiz > (F) 0 ? -sqrt_(ct2) : sqrt_(ct2);
        wo = vx.d * eta_ti + s.ns * (wiz * eta_ti + cos_t); wolz = dot(wo, s.ns);
    } else { wo = vx.d + s.ns * ((F) 2 * wiz); wolz = dot(wo, s.ns); }
    F corr = (F) 1;
    if (vx.cont) corr = abs_((wiz * dot(s.ng, wo)) / (wolz * dot(s.ng, md)));
    out.o = s.p + s.ng * vx.mag; out.d = wo;
    out.T = (vx.T * vx.W) * (corr * vx.rr);
    return out;
}

// adjoint: given (o', d', T')_bar and dc = 1, ACCUMULATE the adjoints of the shading record, of the incoming direction and throughput
// and of the IOR ratio.  `a` must be zero-initialised by the caller (or hold earlier contributions).
template <typename F> GA_HD void transport_adjoint(const Shade<F> &s, const Vertex<F> &vx, const Sensor<F> &se, V<F> ao_out, V<F> ad_out, V<F> aT_out,
                                                   Shade<F> &a, V<F> &ad, V<F> &aT, F &arel) {
    const V<F> md = -vx.d; const F wiz = dot(md, s.ns);
    V<F> amd = mk<F>(0, 0, 0); F awiz = (F) 0;
    // ---- continuation
    {
        V<F> wo; F wolz, eta_ti = (F) 1, ct2 = (F) 1, k = (F) 0;
        if (vx.lobe == LOBE_FRAME) { wo = s.sh_s * vx.l.x + s.sh_t * vx.l.y + s.ns * vx.l.z; wolz = vx.l.z; }
        else if (vx.lobe == LOBE_REFRACT) {
            eta_ti = wiz > (F) 0 ? (F) 1 / vx.rel_eta : vx.rel_eta; ct2 = (F) 1 - ((F) 1 - wiz * wiz) * eta_ti * eta_ti;
            const F cos_t = wiz > (F) 0 ? -sqrt_(ct2) : sqrt_(ct2);
            k = wiz * eta_ti + cos_t; wo = vx.d * eta_ti + s.ns * k; wolz = dot(wo, s.ns);
        } else { wo = vx.d + s.ns * ((F) 2 * wiz); wolz = dot(wo, s.ns); }
        a.p = a.p + ao_out; a.ng = a.ng + ao_out * vx.mag;                  // o' = p + ng mag
        V<F> awo = ad_out;                                                  // d' = wo
        F corr = (F) 1, r = (F) 1, g1 = (F) 0, g2 = (F) 0;
        if (vx.cont) { g1 = dot(s.ng, wo); g2 = dot(s.ng, md); r = (wiz * g1) / (wolz * g2); corr = abs_(r); }
        aT = aT + (aT_out * vx.W) * (corr * vx.rr);                         // T' = T W corr rr
        const V<F> tw = aT_out * vx.T;
        awiz += dot(tw, vx.dW) * (corr * vx.rr);                            // W = W(cos theta_i, rel_eta): Fresnel term of a conductor / dielectric
        arel += dot(tw, vx.dWe) * (corr * vx.rr);
        if (vx.cont) {
            const F acorr = dot(tw, vx.W) * vx.rr, ar = r < (F) 0 ? -acorr : acorr;
            const F den = wolz * g2, aa = ar / den, ab = -ar * r / den;     // r = (wiz g1) / (wolz g2)
            awiz += aa * g1; const F ag1 = aa * wiz;
            a.ng = a.ng + wo * ag1; awo = awo + s.ng * ag1;
            const F awolz = ab * g2, ag2 = ab * wolz;
            a.ng = a.ng + md * ag2; amd = amd + s.ng * ag2;
            if (vx.lobe != LOBE_FRAME) { awo = awo + s.ns * awolz; a.ns = a.ns + wo * awolz; }
        }
        if (vx.lobe == LOBE_FRAME) { a.sh_s = a.sh_s + awo * vx.l.x; a.sh_t = a.sh_t + awo * vx.l.y; a.ns = a.ns + awo * vx.l.z; }
        else if (vx.lobe == LOBE_REFRACT) {
            ad = ad + awo * eta_ti; F aeta = dot(awo, vx.d);
            a.ns = a.ns + awo * k; const F ak = dot(awo, s.ns);
            awiz += ak * eta_ti; aeta += ak * wiz;
            const F sq = sqrt_(ct2), act2 = sq > (F) 0 ? (wiz > (F) 0 ? -ak : ak) * (F) 0.5 / sq : (F) 0;
            awiz += act2 * ((F) 2 * wiz * eta_ti * eta_ti); aeta += act2 * (-((F) 1 - wiz * wiz) * (F) 2 * eta_ti);
            arel += wiz > (F) 0 ? -aeta / (vx.rel_eta * vx.rel_eta) : aeta;
        } else { ad = ad + awo; a.ns = a.ns + awo * ((F) 2 * wiz); awiz += (F) 2 * dot(awo, s.ns); }
    }
    // ---- connection
    if (vx.conn) {
        const V<F> rel = s.p - se.cam_o;
        const V<F> rp = mk<F>(se.R[0] * rel.x + se.R[3] * rel.y + se.R[6] * rel.z, se.R[1] * rel.x + se.R[4] * rel.y + se.R[7] * rel.z, se.R[2] * rel.x + se.R[5] * rel.y + se.R[8] * rel.z);
        const F hx = se.M[0] * rp.x + se.M[1] * rp.y + se.M[2] * rp.z + se.M[3], hy = se.M[4] * rp.x + se.M[5] * rp.y + se.M[6] * rp.z + se.M[7];
        const F hw = se.M[12] * rp.x + se.M[13] * rp.y + se.M[14] * rp.z + se.M[15];
        const F dist = sqrt_(dot(rp, rp)), iz = (F) 1 / rp.z, sw = se.normalization * dist * iz * iz * iz;
        const V<F> x = se.cam_o - s.p; const F xl = sqrt_(dot(x, x)); const V<F> ts = x * ((F) 1 / xl);
        const F lz = dot(ts, s.ns), wo_g = dot(s.ng, ts), wi_g = dot(s.ng, md), r = (wiz * wo_g) / (lz * wi_g), corr = abs_(r);
        const F lobe = lz * vx.K * corr * sw;
        const V<F> TR = vx.T * vx.refl, Vc = TR * lobe;
        aT = aT + (vx.Gv * vx.refl) * lobe;                                 // c = <Gv, V>, V = T refl lobe
        const F alobe = dot(vx.Gv, TR), apx = dot(vx.Gx, Vc), apy = dot(vx.Gy, Vc);
        F alz = alobe * vx.K * corr * sw; const F acorr = alobe * lz * vx.K * sw, asw = alobe * lz * vx.K * corr;
        const F ar = r < (F) 0 ? -acorr : acorr, den = lz * wi_g, anum = ar / den, aden = -ar * r / den;
        awiz += anum * wo_g; const F awo_g = anum * wiz; alz += aden * wi_g; const F awi_g = aden * lz;
        V<F> ats = s.ng * awo_g + s.ns * alz;
        a.ng = a.ng + ts * awo_g + md * awi_g; amd = amd + s.ng * awi_g; a.ns = a.ns + ts * alz;
        a.p = a.p - normalize_adj(ts, xl, ats);                             // ts = normalize(cam_o - p)
        // sw = normalization dist / z^3; px = hx / hw crop_w + crop_x
        V<F> arp = rp * (asw * se.normalization * iz * iz * iz / dist);
        arp.z += asw * se.normalization * dist * ((F) -3 * iz * iz * iz * iz);
        const F ahx = apx * se.crop_w / hw, ahy = apy * se.crop_h / hw, ahw = -(apx * se.crop_w * hx + apy * se.crop_h * hy) / (hw * hw);
        arp.x += se.M[0] * ahx + se.M[4] * ahy + se.M[12] * ahw; arp.y += se.M[1] * ahx + se.M[5] * ahy + se.M[13] * ahw; arp.z += se.M[2] * ahx + se.M[6] * ahy + se.M[14] * ahw;
        a.p = a.p + mk<F>(se.R[0] * arp.x + se.R[1] * arp.y + se.R[2] * arp.z, se.R[3] * arp.x + se.R[4] * arp.y + se.R[5] * arp.z, se.R[6] * arp.x + se.R[7] * arp.y + se.R[8] * arp.z);
    }
    // wiz = md . ns, md = -d
    amd = amd + s.ns * awiz; a.ns = a.ns + md * awiz;
    ad = ad - amd;
}

// ---- angle-weighted vertex normals (Mesh::recompute_vertex_normals, mesh.cpp:376-413, the JIT branch) -------------------
// One face's contribution to the un-normalised normals of its three corners: N[corner i] += n * angle_i with
// n = normalize((v1 - v0) x (v2 - v0)) and angle_i = safe_acos(normalize(v[i+1] - v[i]) . normalize(v[i+2] - v[i])).
template <typename F> GA_HD F acos_(F x) { return (F) acos((double) x); }
template <> GA_HD float acos_<float>(float x) { return acosf(x); }
template <typename F> GA_HD void face_normal_terms(V<F> v0, V<F> v1, V<F> v2, V<F> &n, F angle[3]) {
    const V<F> gn = cross(v1 - v0, v2 - v0); n = gn * ((F) 1 / sqrt_(dot(gn, gn)));
    const V<F> v[3] = { v0, v1, v2 };
    for (int i = 0; i < 3; ++i) {
        const V<F> x0 = v[(i + 1) % 3] - v[i], x1 = v[(i + 2) % 3] - v[i];
        const V<F> d0 = x0 * ((F) 1 / sqrt_(dot(x0, x0))), d1 = x1 * ((F) 1 / sqrt_(dot(x1, x1)));
        F c = dot(d0, d1); c = c < (F) -1 ? (F) -1 : (c > (F) 1 ? (F) 1 : c);
        angle[i] = acos_(c);
    }
}
// forward mode: tangents dv of the three positions -> tangent of (n * angle[corner]) for one corner
template <typename F> GA_HD V<F> face_normal_term_jvp(V<F> v0, V<F> v1, V<F> v2, V<F> dv0, V<F> dv1, V<F> dv2, int corner, V<F> &term) {
    const V<F> e1 = v1 - v0, e2 = v2 - v0, gn = cross(e1, e2); const F gl = sqrt_(dot(gn, gn)); const V<F> n = gn * ((F) 1 / gl);
    const V<F> dgn = cross(dv1 - dv0, e2) + cross(e1, dv2 - dv0), dn = (dgn - n * dot(n, dgn)) * ((F) 1 / gl);
    const V<F> v[3] = { v0, v1, v2 }, dv[3] = { dv0, dv1, dv2 };
    const int i = corner, i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    const V<F> x0 = v[i1] - v[i], x1 = v[i2] - v[i], dx0 = dv[i1] - dv[i], dx1 = dv[i2] - dv[i];
    const F l0 = sqrt_(dot(x0, x0)), l1 = sqrt_(dot(x1, x1));
    const V<F> d0 = x0 * ((F) 1 / l0), d1 = x1 * ((F) 1 / l1);
    const V<F> dd0 = (dx0 - d0 * dot(d0, dx0)) * ((F) 1 / l0), dd1 = (dx1 - d1 * dot(d1, dx1)) * ((F) 1 / l1);
    F c = dot(d0, d1); const bool clamped = c <= (F) -1 || c >= (F) 1; c = c < (F) -1 ? (F) -1 : (c > (F) 1 ? (F) 1 : c);
    const F angle = acos_(c), dc = dot(dd0, d1) + dot(d0, dd1), dangle = clamped ? (F) 0 : -dc / sqrt_((F) 1 - c * c);
    term = n * angle;
    return dn * angle + n * dangle;
}
// adjoint: aN[i] = adjoint of the un-normalised normal of corner i  ->  adjoint of the three positions (SET)
template <typename F> GA_HD void face_normal_terms_adj(V<F> v0, V<F> v1, V<F> v2, const V<F> aN[3], V<F> av[3]) {
    const V<F> e1 = v1 - v0, e2 = v2 - v0, gn = cross(e1, e2); const F gl = sqrt_(dot(gn, gn)); const V<F> n = gn * ((F) 1 / gl);
    const V<F> v[3] = { v0, v1, v2 };
    av[0] = av[1] = av[2] = mk<F>(0, 0, 0);
    V<F> an = mk<F>(0, 0, 0);
    for (int i = 0; i < 3; ++i) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        const V<F> x0 = v[i1] - v[i], x1 = v[i2] - v[i];
        const F l0 = sqrt_(dot(x0, x0)), l1 = sqrt_(dot(x1, x1));
        const V<F> d0 = x0 * ((F) 1 / l0), d1 = x1 * ((F) 1 / l1);
        F c = dot(d0, d1); const bool clamped = c <= (F) -1 || c >= (F) 1; c = c < (F) -1 ? (F) -1 : (c > (F) 1 ? (F) 1 : c);
        const F angle = acos_(c);
        an = an + aN[i] * angle;                                            // N_i += n angle_i
        const F aangle = dot(aN[i], n);
        const F ac = clamped ? (F) 0 : -aangle / sqrt_((F) 1 - c * c);     // d acos(c) = -dc / sqrt(1 - c^2)
        const V<F> ax0 = normalize_adj(d0, l0, d1 * ac), ax1 = normalize_adj(d1, l1, d0 * ac);
        av[i1] = av[i1] + ax0; av[i2] = av[i2] + ax1; av[i] = av[i] - ax0 - ax1;
    }
    const V<F> agn = normalize_adj(n, gl, an);
    const V<F> ae1 = cross(e2, agn), ae2 = cross(agn, e1);
    av[1] = av[1] + ae1; av[2] = av[2] + ae2; av[0] = av[0] - ae1 - ae2;
}

}  // namespace ga
