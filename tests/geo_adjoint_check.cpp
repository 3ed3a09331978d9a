// geo_adjoint_check.cpp -- host-side check of mitsuba3-camera_amd/csrc/mi355_geo_adjoint.h (the hand-written reverse mode of one
// light-path vertex): every adjoint function is run in double precision against central finite differences of its forward function
// on random inputs.  Built and run by tests/test_host_cpu.py::test_geo_adjoint_matches_finite_differences (g++, no GPU).
// Prints one line per case and exits non-zero if any relative error exceeds 1e-6.
#include "../mitsuba3-camera_amd/csrc/mi355_geo_adjoint.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>

using namespace ga;
typedef double F;
typedef V<F> V3d;

static unsigned long long g_state = 0x853c49e6748fea9bULL;
static double rnd() { g_state = g_state * 6364136223846793005ULL + 1442695040888963407ULL; return (double) (g_state >> 11) / 9007199254740992.0; }
static double rs(double lo, double hi) { return lo + (hi - lo) * rnd(); }
static V3d rv(double lo, double hi) { return mk<F>(rs(lo, hi), rs(lo, hi), rs(lo, hi)); }
static V3d unit(V3d a) { return a * (1.0 / sqrt(dot(a, a))); }

static int g_fail = 0;
// compare an analytic gradient with central differences of phi over the packed inputs x
static void check(const char *name, std::vector<double> x, const std::function<double(const std::vector<double> &)> &phi, const std::vector<double> &grad, double tol = 1e-6) {
    double worst = 0, scale = 0;
    for (double g : grad) scale = std::max(scale, fabs(g));
    for (size_t i = 0; i < x.size(); ++i) {
        const double h = 1e-6 * std::max(1.0, fabs(x[i])), keep = x[i];
        x[i] = keep + h; const double fp = phi(x);
        x[i] = keep - h; const double fm = phi(x);
        x[i] = keep;
        const double fd = (fp - fm) / (2 * h);
        worst = std::max(worst, fabs(fd - grad[i]) / std::max(scale, 1e-12));
    }
    printf("%-44s inputs %3zu  max |fd - adjoint| / max |adjoint| = %.3e %s\n", name, x.size(), worst, worst < tol ? "ok" : "FAIL");
    if (!(worst < tol)) g_fail = 1;
}

static void put(std::vector<double> &x, V3d v) { x.push_back(v.x); x.push_back(v.y); x.push_back(v.z); }
static V3d get(const std::vector<double> &x, size_t &k) { V3d v = mk<F>(x[k], x[k + 1], x[k + 2]); k += 3; return v; }

static Sensor<F> make_sensor() {
    Sensor<F> se;
    // a rotation about a skew axis (Rodrigues), camera at some point; camera_to_sample of a perspective camera
    V3d ax = unit(mk<F>(0.3, -0.5, 0.8)); double ang = 0.7, c = cos(ang), s = sin(ang);
    double R[9] = { c + ax.x * ax.x * (1 - c), ax.x * ax.y * (1 - c) - ax.z * s, ax.x * ax.z * (1 - c) + ax.y * s,
                    ax.y * ax.x * (1 - c) + ax.z * s, c + ax.y * ax.y * (1 - c), ax.y * ax.z * (1 - c) - ax.x * s,
                    ax.z * ax.x * (1 - c) - ax.y * s, ax.z * ax.y * (1 - c) + ax.x * s, c + ax.z * ax.z * (1 - c) };
    memcpy(se.R, R, sizeof(R));
    se.cam_o = mk<F>(0.2, 0.1, 4.0);
    double M[16] = { -1.4, 0, 0.5, 0, 0, -1.4, 0.5, 0, 0, 0, 1.0001, -0.01, 0, 0, 1, 0 };
    memcpy(se.M, M, sizeof(M));
    se.crop_w = 64; se.crop_h = 48; se.crop_x = 3; se.crop_y = 5; se.normalization = 1.7;
    return se;
}

// ---------------------------------------------------------------------------------------------------------------------
static void test_mesh_hit(bool has_nrm, bool flip, bool has_uv) {
    Tri<F> t; t.has_nrm = has_nrm; t.flip = flip; t.has_uv = has_uv;
    t.p0 = mk<F>(-0.6, -0.5, 0.05); t.p1 = mk<F>(0.7, -0.4, -0.1); t.p2 = mk<F>(-0.1, 0.8, 0.12);
    t.n0 = unit(mk<F>(0.1, -0.2, 1.0)); t.n1 = unit(mk<F>(-0.15, 0.1, 1.0)); t.n2 = unit(mk<F>(0.05, 0.25, 1.0));
    t.du0x = 0.9; t.du0y = 0.1; t.du1x = 0.2; t.du1y = 0.8;
    V3d o = mk<F>(0.1, 0.05, 2.0), d = unit(mk<F>(-0.12, 0.07, -1.0));
    Shade<F> a; a.p = rv(-1, 1); a.ng = rv(-1, 1); a.ns = rv(-1, 1); a.sh_s = rv(-1, 1); a.sh_t = rv(-1, 1);
    std::vector<double> x; put(x, o); put(x, d); put(x, t.p0); put(x, t.p1); put(x, t.p2); put(x, t.n0); put(x, t.n1); put(x, t.n2);
    auto phi = [&](const std::vector<double> &y) {
        size_t k = 0; V3d oo = get(y, k), dd = get(y, k); Tri<F> tt = t;
        tt.p0 = get(y, k); tt.p1 = get(y, k); tt.p2 = get(y, k); tt.n0 = get(y, k); tt.n1 = get(y, k); tt.n2 = get(y, k);
        Shade<F> s = mesh_hit_forward(tt, oo, dd);
        return dot(a.p, s.p) + dot(a.ng, s.ng) + dot(a.ns, s.ns) + dot(a.sh_s, s.sh_s) + dot(a.sh_t, s.sh_t);
    };
    V3d ao = mk<F>(0, 0, 0), ad = mk<F>(0, 0, 0); TriAdj<F> ta;
    mesh_hit_adjoint(t, o, d, a, ao, ad, ta);
    std::vector<double> g; put(g, ao); put(g, ad); put(g, ta.p0); put(g, ta.p1); put(g, ta.p2);
    put(g, has_nrm ? ta.n0 : mk<F>(0, 0, 0)); put(g, has_nrm ? ta.n1 : mk<F>(0, 0, 0)); put(g, has_nrm ? ta.n2 : mk<F>(0, 0, 0));
    char name[96]; snprintf(name, sizeof(name), "mesh_hit (normals %d, flip %d, uv frame %d)", has_nrm, flip, has_uv);
    check(name, x, phi, g);
}

static void test_plane_sphere() {
    V3d q = mk<F>(0.1, -0.2, 0.3), n = unit(mk<F>(0.2, 0.1, 1.0)), o = mk<F>(0.3, 0.2, 2.5), d = unit(mk<F>(0.1, -0.05, -1.0)), ap = rv(-1, 1);
    {
        std::vector<double> x; put(x, o); put(x, d);
        auto phi = [&](const std::vector<double> &y) { size_t k = 0; V3d oo = get(y, k), dd = get(y, k); return dot(ap, plane_hit_forward(q, n, oo, dd)); };
        V3d ao = mk<F>(0, 0, 0), ad = mk<F>(0, 0, 0); plane_hit_adjoint(q, n, o, d, ap, ao, ad);
        std::vector<double> g; put(g, ao); put(g, ad);
        check("plane_hit", x, phi, g);
    }
    for (int far_root = 0; far_root < 2; ++far_root) {
        V3d c = mk<F>(0.1, 0.0, 0.2); double r = 0.7; V3d ans = rv(-1, 1);
        V3d p0, n0; sphere_hit_forward(c, r, false, o, d, far_root ? 1e9 : -1e9, p0, n0);
        const double t_primal = dot(p0 - o, d) / dot(d, d);
        std::vector<double> x; put(x, o); put(x, d);
        auto phi = [&](const std::vector<double> &y) { size_t k = 0; V3d oo = get(y, k), dd = get(y, k); V3d p, ns; sphere_hit_forward(c, r, far_root != 0, oo, dd, t_primal, p, ns); return dot(ap, p) + dot(ans, ns); };
        V3d ao = mk<F>(0, 0, 0), ad = mk<F>(0, 0, 0); sphere_hit_adjoint(c, r, far_root != 0, o, d, t_primal, ap, ans, ao, ad);
        std::vector<double> g; put(g, ao); put(g, ad);
        check(far_root ? "sphere_hit (far root, flipped)" : "sphere_hit (near root)", x, phi, g);
    }
}

static void test_transport(int lobe, bool conn, bool cont, bool inside) {
    Sensor<F> se = make_sensor();
    Shade<F> s;
    s.p = mk<F>(0.15, -0.1, 0.2);
    s.ng = unit(mk<F>(0.05, -0.08, 1.0)); s.ns = unit(mk<F>(-0.1, 0.12, 1.0));
    { V3d t0 = mk<F>(1.0, 0.1, 0.0); V3d tt = t0 - s.ns * dot(s.ns, t0); s.sh_s = unit(tt); s.sh_t = cross(s.ns, s.sh_s); }
    Vertex<F> vx; memset(&vx, 0, sizeof(vx));
    vx.d = unit(mk<F>(0.2, -0.15, inside ? 1.0 : -1.0)); vx.T = mk<F>(0.8, 1.1, 0.6);
    if (inside) { s.ng = -s.ng; }                                          // a ray leaving the glass through a surface whose geometric normal is flipped too
    vx.lobe = lobe; vx.l = unit(mk<F>(0.3, -0.2, 0.9));
    vx.W = mk<F>(0.9, 0.7, 0.8); vx.dW = lobe != LOBE_FRAME ? mk<F>(0.2, -0.1, 0.3) : mk<F>(0, 0, 0);
    vx.dWe = lobe != LOBE_FRAME ? mk<F>(-0.15, 0.25, 0.1) : mk<F>(0, 0, 0);
    vx.rel_eta = 1.5; vx.mag = 1e-4; vx.rr = 1.25; vx.cont = cont; vx.conn = conn;
    vx.refl = mk<F>(0.5, 0.6, 0.7); vx.K = 0.01;
    const V3d A = rv(-1, 1), B = rv(-0.2, 0.2), Cc = rv(-0.2, 0.2);         // model image adjoint: G(px, py) = A + B px + C py
    const V3d ao_out = rv(-1, 1), ad_out = rv(-1, 1), aT_out = rv(-1, 1);
    const double wiz0 = -dot(vx.d, s.ns);
    std::vector<double> x; put(x, s.p); put(x, s.ng); put(x, s.ns); put(x, s.sh_s); put(x, s.sh_t); put(x, vx.d); put(x, vx.T); x.push_back(vx.rel_eta);
    auto phi = [&](const std::vector<double> &y) {
        size_t k = 0; Shade<F> ss; ss.p = get(y, k); ss.ng = get(y, k); ss.ns = get(y, k); ss.sh_s = get(y, k); ss.sh_t = get(y, k);
        Vertex<F> v2 = vx; v2.d = get(y, k); v2.T = get(y, k); v2.rel_eta = y[k];
        if (lobe != LOBE_FRAME) v2.W = vx.W + vx.dW * (-dot(v2.d, ss.ns) - wiz0) + vx.dWe * (v2.rel_eta - vx.rel_eta);   // W(cos theta_i, eta), linearised around the primal
        double px = 0, py = 0; if (conn) film_position(se, ss.p, px, py);
        v2.Gv = A + B * px + Cc * py;
        TransportOut<F> o = transport_forward(ss, v2, se);
        return dot(ao_out, o.o) + dot(ad_out, o.d) + dot(aT_out, o.T) + o.c;
    };
    double px = 0, py = 0; if (conn) film_position(se, s.p, px, py);
    vx.Gv = A + B * px + Cc * py; vx.Gx = B; vx.Gy = Cc;
    Shade<F> a; a.p = a.ng = a.ns = a.sh_s = a.sh_t = mk<F>(0, 0, 0);
    V3d ad = mk<F>(0, 0, 0), aT = mk<F>(0, 0, 0); double arel = 0;
    transport_adjoint(s, vx, se, ao_out, ad_out, aT_out, a, ad, aT, arel);
    std::vector<double> g; put(g, a.p); put(g, a.ng); put(g, a.ns); put(g, a.sh_s); put(g, a.sh_t); put(g, ad); put(g, aT); g.push_back(arel);
    char name[96]; snprintf(name, sizeof(name), "transport (lobe %d, conn %d, cont %d, inside %d)", lobe, conn, cont, inside);
    check(name, x, phi, g);
}

static void test_face_normals() {
    V3d v0 = mk<F>(-0.6, -0.5, 0.05), v1 = mk<F>(0.7, -0.4, -0.1), v2 = mk<F>(-0.1, 0.8, 0.12);
    V3d aN[3] = { rv(-1, 1), rv(-1, 1), rv(-1, 1) };
    std::vector<double> x; put(x, v0); put(x, v1); put(x, v2);
    auto phi = [&](const std::vector<double> &y) {
        size_t k = 0; V3d a = get(y, k), b = get(y, k), c = get(y, k); V3d n; double ang[3];
        face_normal_terms(a, b, c, n, ang);
        return dot(aN[0], n) * ang[0] + dot(aN[1], n) * ang[1] + dot(aN[2], n) * ang[2];
    };
    V3d av[3]; face_normal_terms_adj(v0, v1, v2, aN, av);
    std::vector<double> g; put(g, av[0]); put(g, av[1]); put(g, av[2]);
    check("face_normal_terms (angle-weighted normals)", x, phi, g);
    // forward mode against the reverse mode: <aN, J dv> = <J^T aN, dv>
    V3d dv[3] = { rv(-1, 1), rv(-1, 1), rv(-1, 1) }; double lhs = 0, rhs = 0;
    for (int c = 0; c < 3; ++c) { V3d term; V3d dt = face_normal_term_jvp(v0, v1, v2, dv[0], dv[1], dv[2], c, term); lhs += dot(aN[c], dt); rhs += dot(av[c], dv[c]); }
    const double e = fabs(lhs - rhs) / std::max(fabs(lhs), 1e-12);
    printf("%-44s <aN, J dv> vs <J^T aN, dv>: relative difference %.3e %s\n", "face_normal_term_jvp (transpose identity)", e, e < 1e-10 ? "ok" : "FAIL");
    if (!(e < 1e-10)) g_fail = 1;
}

static void test_fresnel() {
    for (int inside = 0; inside < 2; ++inside) {
        const double c0 = inside ? -0.83 : 0.62, e0 = 1.52;
        double r, dc, de; fresnel_partials(c0, e0, r, dc, de);
        auto fr = [](double c, double e) { double r_, a_, b_; fresnel_partials(c, e, r_, a_, b_); return r_; };
        const double h = 1e-6, fc = (fr(c0 + h, e0) - fr(c0 - h, e0)) / (2 * h), fe = (fr(c0, e0 + h) - fr(c0, e0 - h)) / (2 * h);
        const double err = std::max(fabs(fc - dc), fabs(fe - de)) / std::max(fabs(dc), fabs(de));
        // the value against the closed form for this angle: Fresnel equations for unpolarised light
        const double n1 = inside ? e0 : 1.0, n2 = inside ? 1.0 : e0, ci = fabs(c0), st = n1 / n2 * sqrt(1 - ci * ci), ct = sqrt(1 - st * st);
        const double rs = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct), rp = (n1 * ct - n2 * ci) / (n1 * ct + n2 * ci), ref = 0.5 * (rs * rs + rp * rp);
        printf("%-44s r = %.9f (closed form %.9f), partials vs fd: %.3e %s\n", inside ? "fresnel_partials (from inside)" : "fresnel_partials (from outside)", r, ref, err,
               (err < 1e-6 && fabs(r - ref) < 1e-12) ? "ok" : "FAIL");
        if (!(err < 1e-6 && fabs(r - ref) < 1e-12)) g_fail = 1;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The closed-form AD assertions the REFERENCE's own tests hold for Mesh::compute_surface_interaction -- src/render/tests/test_mesh.py --
// restated for mesh_hit_adjoint (reverse mode) and, through the transpose, for what its forward-mode tests say.  The mesh is
// resources/data/common/meshes/rectangle.obj: the square [-1, 1]^2 in z = 0; the asserted gradient vectors themselves disclose its layout
// (test16, :536-640): vertex 0 = (1, -1, 0), 1 = (-1, 1, 0), 2 = (-1, -1, 0), 3 = (1, 1, 0), and the triangle under the ray at
// (0.99999, 0.99999) is {0, 3, 1} (counter-clockwise seen from +z: its normal is +z).  atol 1e-5 as in the reference.
static void expect(const char *what, const std::vector<double> &got, const std::vector<double> &want, double atol = 1e-5) {
    // dr.allclose(a, b, rtol = 1e-5, atol): |a - b| <= atol + rtol |b|
    double worst = 0; bool ok = true;
    for (size_t i = 0; i < want.size(); ++i) { const double e = fabs(got[i] - want[i]); worst = std::max(worst, e); ok = ok && e <= atol + 1e-5 * fabs(want[i]); }
    printf("%-76s max |got - reference| = %.3e %s\n", what, worst, ok ? "ok" : "FAIL");
    if (!ok) g_fail = 1;
}
static void test_reference_mesh_assertions() {
    const V3d P[4] = { mk<F>(1, -1, 0), mk<F>(-1, 1, 0), mk<F>(-1, -1, 0), mk<F>(1, 1, 0) };
    const int tri[3] = { 0, 3, 1 };
    for (int with_normals = 0; with_normals < 2; ++with_normals) {         // si.n (geometric) and si.sh_frame.n (interpolated vertex normals of the flat mesh)
        Tri<F> t; t.p0 = P[tri[0]]; t.p1 = P[tri[1]]; t.p2 = P[tri[2]];
        t.has_nrm = with_normals != 0; t.flip = false; t.has_uv = false; t.n0 = t.n1 = t.n2 = mk<F>(0, 0, 1); t.du0x = t.du0y = t.du1x = t.du1y = 0;
        auto run = [&](V3d o, V3d d, Shade<F> a, V3d &ao, V3d &ad, std::vector<double> &gv, TriAdj<F> *raw = nullptr) {
            ao = ad = mk<F>(0, 0, 0); TriAdj<F> ta; mesh_hit_adjoint(t, o, d, a, ao, ad, ta);
            gv.assign(12, 0.0);
            const V3d tp[3] = { ta.p0, ta.p1, ta.p2 };
            for (int k = 0; k < 3; ++k) { gv[3 * tri[k]] = tp[k].x; gv[3 * tri[k] + 1] = tp[k].y; gv[3 * tri[k] + 2] = tp[k].z; }
            if (raw) *raw = ta;
        };
        auto zero = []() { Shade<F> a; a.p = a.ng = a.ns = a.sh_s = a.sh_t = mk<F>(0, 0, 0); return a; };
        V3d ao, ad; std::vector<double> gv;
        if (!with_normals) {
            // test14 (:427-452): dr.backward(si.p.x) -> grad(ray.o) = (1, 0, 0)
            { Shade<F> a = zero(); a.p.x = 1; run(mk<F>(-0.3, -0.4, -10), mk<F>(0, 0, 1), a, ao, ad, gv);
              expect("test_mesh.py:444-446  backward(si.p.x) -> grad(ray.o) = (1, 0, 0)", { ao.x, ao.y, ao.z }, { 1, 0, 0 }); 
              // test13 (:380-424), forward: d si.p / d ray.o.x = (1, 0, 0) and d si.p / d ray.d.x = (10, 0, 0)  (= the x-columns of the transposes)
              Shade<F> ay = zero(); ay.p.y = 1; V3d aoy, ady; std::vector<double> g2; run(mk<F>(-0.3, -0.4, -10), mk<F>(0, 0, 1), ay, aoy, ady, g2);
              Shade<F> az = zero(); az.p.z = 1; V3d aoz, adz; run(mk<F>(-0.3, -0.4, -10), mk<F>(0, 0, 1), az, aoz, adz, g2);
              expect("test_mesh.py:402-406  forward(ray.o.x) -> grad(si.p) = (1, 0, 0)", { ao.x, aoy.x, aoz.x }, { 1, 0, 0 });
              expect("test_mesh.py:420-424  forward(ray.d.x) -> grad(si.p) = (10, 0, 0)", { ad.x, ady.x, adz.x }, { 10, 0, 0 });
              // si.t = (si.p - ray.o) . ray.d for the unit direction (0, 0, 1): backward(si.t) -> grad(ray.o) = (0, 0, -1) (:448-452) is d(p.z - o.z) / d o
              expect("test_mesh.py:448-452  backward(si.t) -> grad(ray.o) = (0, 0, -1)", { aoz.x, aoz.y, aoz.z - 1.0 }, { 0, 0, -1 });
              // test15 (:455-531), forward over a rigid translation of ALL vertices: d si.t / d z = 1, d si.p / d z = (0, 0, 1)
              run(mk<F>(-0.2, -0.3, -10), mk<F>(0, 0, 1), az, aoz, adz, g2);
              expect("test_mesh.py:488-499  forward(translate z) -> grad(si.t) = 1, grad(si.p).z = 1", { g2[2] + g2[5] + g2[8] + g2[11] }, { 1 });
              run(mk<F>(-0.2, -0.3, -10), mk<F>(0, 0, 1), a, ao, ad, g2);          // a.p = (1, 0, 0): a translation in x or y leaves p.x alone, in z too
              expect("test_mesh.py:488-499  forward(translate x / y / z) -> grad(si.p).x = 0", { g2[0] + g2[3] + g2[6] + g2[9], g2[1] + g2[4] + g2[7] + g2[10], g2[2] + g2[5] + g2[8] + g2[11] }, { 0, 0, 0 }); }
            // test16 (:536-640), the ray that hits the corner (vertex 3)
            { Shade<F> a = zero(); a.p.z = 1; run(mk<F>(0.99999, 0.99999, -10), mk<F>(0, 0, 1), a, ao, ad, gv);
              expect("test_mesh.py:566-577  backward(si.t) / backward(si.p.z) -> vertex 3 moves along z", gv, { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 }); }
        }
        {   Shade<F> a = zero(); (with_normals ? a.ns : a.ng).x = 1; run(mk<F>(0.99999, 0.99999, -10), mk<F>(0, 0, 1), a, ao, ad, gv);
            TriAdj<F> raw; std::vector<double> g3; V3d o2, d2; run(mk<F>(0.99999, 0.99999, -10), mk<F>(0, 0, 1), a, o2, d2, g3, &raw);
            if (with_normals) {
                // the interpolated normal depends on the positions through the vertex normals: recompute_vertex_normals (mesh.cpp:376-413); vertex 3
                // belongs to this triangle only, vertices 0 and 1 also to {0, 1, 2} whose normal does not change the derivative of a flat mesh's n3
                const V3d aN[3] = { raw.n0, raw.n1, raw.n2 }; V3d av[3]; face_normal_terms_adj(t.p0, t.p1, t.p2, aN, av);
                // n_i = normalize(sum of angle-weighted face normals): for the flat mesh |sum| = total angle at the vertex; vertex 3 (index 1 here) has pi / 2
                const double ang_tot[3] = { M_PI / 2, M_PI / 2, M_PI / 2 };   // vertex 0: pi/4 + pi/4, vertex 3: pi/2, vertex 1: pi/4 + pi/4
                std::vector<double> gn(12, 0.0);
                for (int c = 0; c < 3; ++c) {                              // adjoint of normalize at a unit vector n = (0, 0, 1): (aN - n (n . aN)) / |sum|
                    V3d an = aN[c]; an.z = 0; V3d an_s = an * (1.0 / ang_tot[c]); V3d aNc[3] = { mk<F>(0, 0, 0), mk<F>(0, 0, 0), mk<F>(0, 0, 0) }; aNc[c] = an_s;
                    V3d avc[3]; face_normal_terms_adj(t.p0, t.p1, t.p2, aNc, avc);
                    for (int k = 0; k < 3; ++k) { gn[3 * tri[k]] += avc[k].x; gn[3 * tri[k] + 1] += avc[k].y; gn[3 * tri[k] + 2] += avc[k].z; }
                }
                (void) av;
                for (int i = 0; i < 12; ++i) g3[i] += gn[i];
                expect("test_mesh.py:616-622  backward(si.sh_frame.n.x) -> rotate the right edge about y", g3, { 0, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0, -0.5 }, 2e-5);
            } else
                expect("test_mesh.py:600-606  backward(si.n.x) -> rotate the right edge about y", gv, { 0, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0, -0.5 });
        }
        if (!with_normals) { Shade<F> a = zero(); a.ng.y = 1; run(mk<F>(0.99999, 0.99999, -10), mk<F>(0, 0, 1), a, ao, ad, gv);
            expect("test_mesh.py:608-614  backward(si.n.y) -> rotate the top edge about x", gv, { 0, 0, 0.5, 0, 0, 0, 0, 0, 0, 0, 0, -0.5 }); }
    }
    // src/shapes/tests/test_rectangle.py:116-171 (test06 / test07): the analytic rectangle (the unit square in z = 0) as a fixed plane
    {   const V3d q = mk<F>(0, 0, 0), n = mk<F>(0, 0, 1), o = mk<F>(-0.3, -0.3, -10), d = mk<F>(0, 0, 1);
        V3d ao[3], ad[3];
        for (int c = 0; c < 3; ++c) { V3d ap = mk<F>(c == 0, c == 1, c == 2); ao[c] = ad[c] = mk<F>(0, 0, 0); plane_hit_adjoint(q, n, o, d, ap, ao[c], ad[c]); }
        expect("test_rectangle.py:157-160 backward(si.p.x) -> grad(ray.o) = (1, 0, 0)", { ao[0].x, ao[0].y, ao[0].z }, { 1, 0, 0 });
        expect("test_rectangle.py:162-166 backward(si.t) -> grad(ray.o) = (0, 0, -1)", { ao[2].x, ao[2].y, ao[2].z - 1.0 }, { 0, 0, -1 });
        expect("test_rectangle.py:124-128 forward(ray.o.x) -> grad(si.p) = (1, 0, 0)", { ao[0].x, ao[1].x, ao[2].x }, { 1, 0, 0 });
        expect("test_rectangle.py:130-134 forward(ray.o.y) -> grad(si.p) = (0, 1, 0)", { ao[0].y, ao[1].y, ao[2].y }, { 0, 1, 0 });
        expect("test_rectangle.py:148-152 forward(ray.d.x) -> grad(si.p) = (10, 0, 0)", { ad[0].x, ad[1].x, ad[2].x }, { 10, 0, 0 });
    }
    // test04 (:74-98): the angle weighting of recompute_vertex_normals on the 5-vertex, 2-face mesh
    {   const double a_ = 1.0, b_ = 0.5;
        const V3d v[5] = { mk<F>(0, 0, 0), mk<F>(-a_, 1, 0), mk<F>(a_, 1, 0), mk<F>(-b_, 0, 1), mk<F>(b_, 0, 1) };
        V3d n0, n1; double a0[3], a1[3];
        face_normal_terms(v[0], v[1], v[2], n0, a0); face_normal_terms(v[0], v[3], v[4], n1, a1);
        V3d s = n0 * a0[0] + n1 * a1[0]; s = s * (1.0 / sqrt(dot(s, s)));
        V3d ref = mk<F>(0, 0, -1) * (M_PI / 2.0) + mk<F>(0, 1, 0) * acos(3.0 / 5.0); ref = ref * (1.0 / sqrt(dot(ref, ref)));
        expect("test_mesh.py:74-98    angle-weighted vertex normal of the shared vertex", { s.x, s.y, s.z, n0.x, n0.y, n0.z, n1.x, n1.y, n1.z }, { ref.x, ref.y, ref.z, 0, 0, -1, 0, 1, 0 }, 5e-4);
    }
}

int main() {
    test_reference_mesh_assertions();
    test_fresnel();
    for (int nrm = 0; nrm < 2; ++nrm) for (int flip = 0; flip < 2; ++flip) for (int uv = 0; uv < 2; ++uv) test_mesh_hit(nrm, flip, uv);
    test_plane_sphere();
    for (int lobe = 0; lobe < 3; ++lobe) for (int conn = 0; conn < 2; ++conn) for (int cont = 0; cont < 2; ++cont) test_transport(lobe, conn, cont, false);
    test_transport(LOBE_REFRACT, false, true, true);
    test_face_normals();
    return g_fail;
}
