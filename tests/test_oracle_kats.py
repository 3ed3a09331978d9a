"""Pins the CPU oracle (and the library's host-side helpers) to the known-answer vectors the
reference's own tests hold for the leaf functions of the path (tests/golden/reference_kats.json).
CPU only.  Mirrors: test_random.py, test_sensor.py, test_diffuse.py, test_rectangle.py, test_cube.py,
test_rfilter.py, test_imageblock.py, test_hdrfilm.py, test_area.py of the reference."""
import ctypes as C
import math

import numpy as np
import pytest


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- RNG ------------------------------------------------------------------------------------------
def test_tea_float32(orc, mi, kats):
    l = orc.lib(mi.abi)
    k = kats["tea_float32"]
    for v0, v1, expected in k["cases"]:
        assert l.orc_tea_float32(v0, v1, k["rounds"]) == np.float32(expected)


def test_tea_float64(orc, mi, kats):
    l = orc.lib(mi.abi)
    k = kats["tea_float64"]
    for v0, v1, expected in k["cases"]:
        assert l.orc_tea_float64(v0, v1, k["rounds"]) == expected


def test_tea_host_matches_oracle(orc, mi):
    l = orc.lib(mi.abi)
    out = np.zeros(2, dtype=np.uint32)
    for v0, v1 in [(0, 0), (1, 1), (123456789, 987654321), (0xffffffff, 0xffffffff)]:
        l.orc_tea32(v0, v1, 4, _p(out))
        assert tuple(int(x) for x in out) == mi.sample_tea_32(v0, v1)


def test_pcg32_published_vector(orc, mi, kats):
    l = orc.lib(mi.abi)
    k = kats["pcg32_demo"]
    out = np.zeros(len(k["outputs_hex"]), dtype=np.uint32)
    l.orc_pcg32_raw(k["initstate"], k["initseq"], out.size, _p(out))
    assert [format(int(x), "08x") for x in out] == k["outputs_hex"]


def test_sampler_floats_range_and_replay(orc, mi):
    """IndependentSampler wiring (src/samplers/tests/test_independent.py:17-29): same (seed, lane) -> same stream,
    floats in [0,1), different lanes decorrelated by TEA."""
    l = orc.lib(mi.abi)
    a = np.zeros(64, dtype=np.float32); b = np.zeros(64, dtype=np.float32); c = np.zeros(64, dtype=np.float32)
    l.orc_sampler_floats(7, 1234, 64, _p(a)); l.orc_sampler_floats(7, 1234, 64, _p(b)); l.orc_sampler_floats(7, 1235, 64, _p(c))
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert (a >= 0).all() and (a < 1).all()
    # clone semantics used by PRB: replaying from a copied state reproduces the suffix
    big = np.zeros(200000, dtype=np.float32)
    l.orc_sampler_floats(0, 0, big.size, _p(big))
    assert abs(big.mean() - 0.5) < 5e-3 and abs(big.var() - 1 / 12) < 5e-3


# ---- sensor ---------------------------------------------------------------------------------------
def test_parse_fov(mi, kats):
    for props, aspect, expected in kats["parse_fov"]["cases"]:
        assert math.isclose(mi.parse_fov(props, aspect), expected, rel_tol=1e-5)
    with pytest.raises(mi.PluginError):
        mi.parse_fov({"fov": 35.0, "focal_length": "50mm"}, 1.0)
    with pytest.raises(mi.PluginError):
        mi.parse_fov({"fov": 35.0, "fov_axis": "bogus"}, 1.0)
    with pytest.raises(mi.PluginError):
        mi.parse_fov({"fov": 190.0}, 1.0)


def test_perspective_projection(orc, mi, kats):
    k = kats["perspective_projection"]
    film = np.array(k["film_size"], dtype=np.int32); crop = np.array(k["crop_size"], dtype=np.int32); off = np.array(k["crop_offset"], dtype=np.int32)
    expected = np.array(k["matrix"], dtype=np.float32)
    m = np.zeros(16, dtype=np.float32); inv = np.zeros(16, dtype=np.float32)
    orc.lib(mi.abi).orc_perspective_projection(_p(film), _p(crop), _p(off), k["fov_x"], k["near_clip"], k["far_clip"], _p(m), _p(inv))
    assert np.allclose(m.reshape(4, 4), expected, rtol=1e-5, atol=1e-6)
    assert np.allclose(m.reshape(4, 4).astype(np.float64) @ inv.reshape(4, 4).astype(np.float64), np.eye(4), atol=1e-4)
    # the product library's host helper computes the same matrix
    m2 = np.zeros(16, dtype=np.float32)
    I32 = C.POINTER(C.c_int32)
    mi.abi.lib().mi355_perspective_projection(film.ctypes.data_as(I32), crop.ctypes.data_as(I32), off.ctypes.data_as(I32),
                                             k["fov_x"], k["near_clip"], k["far_clip"], m2.ctypes.data_as(mi.abi.f32p))
    assert np.array_equal(m, m2)
    # and so does the Python Transform4f mirror
    T = mi.ScalarTransform4f
    assert np.allclose(T.perspective(35.0, 10.0, 1000.0).matrix[2], [0, 0, 1.010101079940796, -10.1010103225708], rtol=1e-6)


def test_look_at_and_chaining(mi):
    T = mi.ScalarTransform4f
    t = T.look_at(origin=[0, 0, 3.9], target=[0, 0, 0], up=[0, 1, 0])
    assert np.allclose(t.matrix[:3, 2], [0, 0, -1]) and np.allclose(t.translation(), [0, 0, 3.9])
    assert np.allclose(t.matrix[:3, 0], [-1, 0, 0])            # left = up x dir (transform.h:262)
    a = T.translate([1, 2, 3]).rotate([0, 1, 0], 90).scale([2, 2, 2])
    b = T.translate([1, 2, 3]) @ T.rotate([0, 1, 0], 90) @ T.scale([2, 2, 2])
    assert np.allclose(a.matrix, b.matrix)
    assert np.allclose(a.matrix.astype(np.float64) @ a.inverse().matrix.astype(np.float64), np.eye(4), atol=1e-6)
    assert np.allclose(a.transform_affine([1, 0, 0]), [1, 2, 1], atol=1e-6)


# ---- BSDF -----------------------------------------------------------------------------------------
def test_diffuse_eval_pdf(orc, mi, kats):
    l = orc.lib(mi.abi)
    k = kats["diffuse_eval_pdf"]
    refl = np.full(3, k["reflectance"], dtype=np.float32); wi = np.array([0, 0, 1], dtype=np.float32)
    for i in range(k["n_angles"]):
        theta = i / 19.0 * (math.pi / 2)
        wo = np.array([math.sin(theta), 0, math.cos(theta)], dtype=np.float32)
        val = np.zeros(3, dtype=np.float32); pdf = np.zeros(1, dtype=np.float32)
        l.orc_diffuse_eval_pdf(_p(refl), _p(wi), _p(wo), _p(val), _p(pdf))
        if wo[2] > 0:
            assert np.allclose(pdf[0], wo[2] / math.pi, rtol=1e-6, atol=1e-8)
            assert np.allclose(val[0], 0.5 * wo[2] / math.pi, rtol=1e-6, atol=1e-8)
        else:
            assert pdf[0] == 0 and val[0] == 0
    # one-sided: back-facing wi or wo evaluate to zero (diffuse.cpp:137,172)
    val = np.ones(3, dtype=np.float32); pdf = np.ones(1, dtype=np.float32)
    l.orc_diffuse_eval_pdf(_p(refl), _p(np.array([0, 0, -1], dtype=np.float32)), _p(np.array([0, 0, 1], dtype=np.float32)), _p(val), _p(pdf))
    assert pdf[0] == 0 and (val == 0).all()


def test_diffuse_sample_chi2(orc, mi):
    """chi^2-style check of sample() against pdf() (test_diffuse.py:40-52, chi2.py): cosine-weighted hemisphere."""
    l = orc.lib(mi.abi)
    rng = np.random.default_rng(12345)
    n = 200000
    s = rng.random((n, 2), dtype=np.float32)
    out = np.zeros((n, 3), dtype=np.float32)
    l.orc_square_to_cosine_hemisphere(_p(s), n, _p(out))
    assert np.allclose((out ** 2).sum(1), 1, atol=1e-5) and (out[:, 2] >= 0).all()
    # cos(theta) of a cosine-weighted sample has density 2c on [0,1]: bin counts ~ (b^2 - a^2)
    edges = np.linspace(0, 1, 21)
    hist, _ = np.histogram(out[:, 2], bins=edges)
    expected = n * (edges[1:] ** 2 - edges[:-1] ** 2)
    chi2 = ((hist - expected) ** 2 / expected).sum()
    assert chi2 < 60          # 19 dof: p(chi2 > 60) ~ 4e-6
    phi = np.arctan2(out[:, 1], out[:, 0])
    h2, _ = np.histogram(phi, bins=16, range=(-math.pi, math.pi))
    assert (((h2 - n / 16) ** 2) / (n / 16)).sum() < 55
    # sample weight = albedo where pdf > 0
    refl = np.array([0.2, 0.5, 0.8], dtype=np.float32); wi = np.array([0, 0, 1], dtype=np.float32)
    wo = np.zeros(3, dtype=np.float32); pdf = np.zeros(1, dtype=np.float32); w = np.zeros(3, dtype=np.float32)
    l.orc_diffuse_sample(_p(refl), _p(wi), _p(np.array([0.3, 0.6], dtype=np.float32)), _p(wo), _p(pdf), _p(w))
    assert np.allclose(w, refl) and np.isclose(pdf[0], wo[2] / math.pi, rtol=1e-6)


# ---- shapes ---------------------------------------------------------------------------------------
def _shape_scene(mi, orc, shape_dict):
    sc = mi.load_dict({"type": "scene", "foo": shape_dict})
    return sc, orc.OracleScene(sc, mi.abi)


def test_rectangle_create_area_bbox(orc, mi, kats):
    k = kats["rectangle_area_bbox"]
    T = mi.ScalarTransform4f
    sc, o = _shape_scene(mi, orc, {"type": "rectangle"})
    assert sc.shapes[0].primitive_count() == 1
    assert np.isclose(o.surface_area(0), k["default_area"]) and np.isclose(sc.shapes[0].surface_area(), k["default_area"])
    for sx in [1, 2, 4]:
        for tr in ([1.3, -3.0, 5], [-10000, 3.0, 31]):
            sc, o = _shape_scene(mi, orc, {"type": "rectangle", "to_world": T.translate(tr) @ T.scale((sx, 2.5, 1.0))})
            assert np.isclose(o.surface_area(0), sx * 2.5 * 4, rtol=1e-5)
            lo, hi = sc.shapes[0].bbox()
            assert np.allclose(lo, np.array(tr) - [sx, 2.5, 0]) and np.allclose(hi, np.array(tr) + [sx, 2.5, 0])
    sc, o = _shape_scene(mi, orc, {"type": "rectangle", "to_world": mi.Transform4f([[1, 0, 0, 0], [0, 1, 0, 0], [1, 0, 1, 0], [0, 0, 0, 1]])})
    assert np.isclose(o.surface_area(0), k["sheared_z_area"], rtol=1e-6)
    sc, o = _shape_scene(mi, orc, {"type": "rectangle", "to_world": mi.Transform4f([[2, 0, 0, 0], [0, 2, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])})
    assert np.isclose(o.surface_area(0), k["scaled2_area"], rtol=1e-6)


def test_rectangle_ray_intersect(orc, mi, kats):
    k = kats["rectangle_ray_intersect"]
    sc, o = _shape_scene(mi, orc, {"type": "rectangle", "to_world": mi.ScalarTransform4f.scale(k["scale"])})
    coords = np.linspace(-1, 1, k["n"]).astype(np.float32)
    rays = np.zeros((k["n"], 8), dtype=np.float32)
    rays[:, 0] = coords; rays[:, 1] = coords; rays[:, 2] = 5; rays[:, 5] = -1; rays[:, 6] = np.finfo(np.float32).max
    h = orc.unpack_hits(o.ray_intersect(rays))
    found = np.isfinite(h["t"])
    assert np.array_equal(found, np.abs(coords) <= 0.5)
    assert np.array_equal(h["occluded"].astype(bool), found)
    assert found.sum() == k["valid_count"]
    assert np.allclose(h["t"][found], 5.0)


def test_cube(orc, mi, kats):
    k = kats["cube"]
    T = mi.ScalarTransform4f
    sc, o = _shape_scene(mi, orc, {"type": "cube"})
    assert sc.shapes[0].primitive_count() == k["primitive_count"]
    assert np.isclose(o.surface_area(0), k["surface_area"]) and np.isclose(sc.shapes[0].surface_area(), k["surface_area"])
    for r in [1, 2, 4]:
        sc2, _ = _shape_scene(mi, orc, {"type": "cube", "to_world": T.scale([r, r, r])})
        lo, hi = sc2.shapes[0].bbox()
        assert np.allclose(lo, [-r] * 3) and np.allclose(hi, [r] * 3)
    for scale in k["scales"]:
        sc, o = _shape_scene(mi, orc, {"type": "cube", "to_world": T.scale(scale)})
        cs = k["coords"]
        rays = np.array([[x, y, -8, 0, 0, 1, np.finfo(np.float32).max, 0] for x in cs for y in cs], dtype=np.float32)
        hits, si = o.ray_intersect(rays, with_si=True)
        h = orc.unpack_hits(hits)
        expect = (np.abs(rays[:, 0]) <= scale[0]) & (np.abs(rays[:, 1]) <= scale[1])
        assert np.array_equal(np.isfinite(h["t"]), expect)
        assert np.array_equal(h["occluded"].astype(bool), expect)
        n = si[expect, 3:6]
        assert np.array_equal(n, np.tile(np.array([0, 0, -1], dtype=np.float32), (expect.sum(), 1)))
        oo = (rays[expect, 1] + scale[1]) / (2 * scale[1]); tt = (rays[expect, 0] + scale[0]) / (2 * scale[0])
        assert np.allclose(si[expect, 6], 1.0 - oo, atol=1e-5, rtol=1e-5) and np.allclose(si[expect, 7], tt, atol=1e-5, rtol=1e-5)
        assert np.allclose(h["t"][expect], 7.0, atol=1e-5)


# ---- film -----------------------------------------------------------------------------------------
def test_gaussian_filter(orc, mi, kats):
    k = kats["gaussian_rfilter"]
    l = orc.lib(mi.abi)
    x = np.array(k["x"], dtype=np.float32); out = np.zeros_like(x)
    l.orc_gaussian_eval(0.5, _p(x), x.size, _p(out))
    assert abs(out[0] - k["expected"][0]) < k["atol"] and out[1] == 0
    # polynomial tracks exp(-x^2 / (2 sigma^2)) - exp(-r^2/(2 sigma^2)) over the support and vanishes at the radius
    xs = np.linspace(0, 2, 101).astype(np.float32); o2 = np.zeros_like(xs)
    l.orc_gaussian_eval(0.5, _p(xs), xs.size, _p(o2))
    ref = np.exp(-xs.astype(np.float64) ** 2 / 0.5) - math.exp(-8.0)
    assert np.abs(o2 - ref).max() < 2e-3 and o2[-1] < 1e-5 and (o2 >= 0).all()
    # the product library's coefficient table is identical to the oracle's
    a = np.zeros(10, dtype=np.float32); b = np.zeros(10, dtype=np.float32)
    l.orc_gaussian_coeffs(0.5, _p(a)); mi.abi.lib().mi355_gaussian_coeffs(0.5, b.ctypes.data_as(mi.abi.f32p))
    assert np.array_equal(a, b)


def _film_scene(mi, orc, w, h, rfilter):
    d = {"type": "scene", "r": {"type": "rectangle"},
         "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": w, "height": h, "rfilter": rfilter, "pixel_format": "rgb"}}}
    sc = mi.load_dict(d)
    return sc, orc.OracleScene(sc, mi.abi)


def test_imageblock_put_boundary(orc, mi, kats):
    k = kats["imageblock_put_boundary"]
    sc, o = _film_scene(mi, orc, k["size"], k["size"], {"type": "gaussian"})
    film = o.film_put([k["pos"]], [[1, 1, 1, 1]])[:, :, 3]
    l = orc.lib(mi.abi)
    x = np.array([0, 1], dtype=np.float32); ab = np.zeros(2, dtype=np.float32)
    l.orc_gaussian_eval(0.5, _p(x), 2, _p(ab))
    a, b = ab[0] * ab[0], ab[1]
    # separable weights: centre = f(0)^2, edge = f(0) f(1), corner = f(1)^2 (test uses a = f(0) ~ 1 up to atol 1e-3)
    expect = np.array([[b * b, ab[0] * b, b * b], [ab[0] * b, a, ab[0] * b], [b * b, ab[0] * b, b * b]])
    assert np.allclose(film, expect, atol=1e-6)
    ref = np.array([[b * b, b, b * b], [b, ab[0], b], [b * b, b, b * b]])
    assert np.allclose(film, ref, atol=1e-3)


@pytest.mark.parametrize("filter_name", ["gaussian", "tent", "box"])
def test_imageblock_put_bruteforce(orc, mi, filter_name):
    """test_imageblock.py:37-107 (coalesce=True, border=False, normalize=False): put == brute-force filter footprint."""
    sc, o = _film_scene(mi, orc, 6, 6, {"type": filter_name})
    l = orc.lib(mi.abi)

    def f(x):
        x = np.ascontiguousarray(np.abs(x), dtype=np.float32).ravel()
        if filter_name == "gaussian":
            out = np.zeros_like(x); l.orc_gaussian_eval(0.5, _p(x), x.size, _p(out)); return out
        if filter_name == "tent":
            return np.maximum(0, 1 - x)
        return (x <= 0.5).astype(np.float32)
    for j in range(5):
        for i in range(5):
            pos = np.array([3.3 + 0.25 * i, 3 + 0.25 * j], dtype=np.float32)
            film = o.film_put([pos], [[0, 0, 0, 1]])[:, :, 3]
            cx = np.arange(6, dtype=np.float32) + np.float32(0.5) - pos[0]; cy = np.arange(6, dtype=np.float32) + np.float32(0.5) - pos[1]
            if filter_name == "box":
                ref = np.zeros((6, 6), dtype=np.float32); ref[int(np.floor(pos[1])), int(np.floor(pos[0]))] = 1
            else:
                ref = np.outer(f(cy), f(cx))
            assert np.allclose(film, ref, atol=1e-5), (filter_name, pos)


def test_hdrfilm_develop(orc, mi):
    """test_hdrfilm.py:142-198 (RGB, no AOVs): develop = rgb / weight, and weight 0 -> division by 1."""
    sc, o = _film_scene(mi, orc, 3, 5, {"type": "gaussian"})
    xs, ys = np.meshgrid(np.arange(3), np.arange(5))
    pos = np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1).astype(np.float32)
    vals = np.stack([xs.ravel(), 2 * ys.ravel(), np.full(15, 0.1), np.full(15, 0.5)], 1).astype(np.float32)
    film = o.film_put(pos, vals)
    img = o.develop(film)
    assert np.allclose(img, film[:, :, :3] / film[:, :, 3:4], rtol=1e-6)
    film[2, 1, :] = [3, 4, 5, 0]
    assert np.allclose(o.develop(film)[2, 1], [3, 4, 5])


# ---- emitter --------------------------------------------------------------------------------------
def test_area_emitter_identities(orc, mi):
    """test_area.py:66-124 restated on lanes: a depth-1 path sees radiance iff the emitter's front side is hit;
    the NEE pdf of a sampled direction equals the solid-angle pdf of hitting that point (pdf_direction identity) --
    checked through MIS weights summing to one in test_path_oracle.py.  Here: eval() front/back."""
    T = mi.ScalarTransform4f
    for flip, expect in [(False, 1.0), (True, 0.0)]:
        tw = T.translate([0, 0, -2]) if not flip else T.translate([0, 0, -2]).rotate([0, 1, 0], 180)
        d = {"type": "scene",
             "light": {"type": "rectangle", "to_world": tw, "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [2.0, 3.0, 4.0]}}},
             "sensor": {"type": "perspective", "fov": 20, "to_world": T.look_at([0, 0, 0], [0, 0, -1], [0, 1, 0]),
                        "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}, "pixel_format": "rgb"}}}
        sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
        integ = mi.PathIntegrator({"max_depth": 1})
        img, _ = o.render(o.params(integ, 0, 4))
        assert np.allclose(img, np.array([2.0, 3.0, 4.0]) * expect)


# ---- scope row a19: microfacet / Fresnel / dielectric / sphere -----------------------------------------------
def _mf(orc, mi, type_, au, av, sv, wi, v=None, samples=None):
    l = orc.lib(mi.abi)
    n = len(v) if v is not None else len(samples)
    wi = np.asarray(wi, dtype=np.float32)
    ev = np.zeros(n, dtype=np.float32); pdf = np.zeros(n, dtype=np.float32); g1 = np.zeros(n, dtype=np.float32)
    sm = np.zeros((n, 3), dtype=np.float32); sp = np.zeros(n, dtype=np.float32)
    va = np.ascontiguousarray(v, dtype=np.float32) if v is not None else None
    sa = np.ascontiguousarray(samples, dtype=np.float32) if samples is not None else None
    l.orc_microfacet.argtypes = [C.c_uint32, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 5
    l.orc_microfacet(type_, au, av, sv, _p(wi), _p(va) if va is not None else None, _p(sa) if sa is not None else None, n,
                     _p(ev), _p(pdf), _p(g1), _p(sm), _p(sp))
    return ev, pdf, g1, sm, sp


def _dirs(theta, phi):
    theta = np.asarray(theta, dtype=np.float32); phi = np.asarray(phi, dtype=np.float32)
    return np.stack([np.cos(phi) * np.sin(theta), np.sin(phi) * np.sin(theta), np.cos(theta)], 1).astype(np.float32)


@pytest.mark.parametrize("name,type_", [("microfacet_beckmann", 0), ("microfacet_ggx", 1)])
def test_microfacet_tables(orc, mi, kats, name, type_):
    k = kats[name]; wi = [0, 0, 1]; steps = 20
    if type_ == 0:
        v = _dirs(np.linspace(0, np.pi, steps), np.full(steps, np.pi / 2))
        ev, pdf, _, _, _ = _mf(orc, mi, 0, 0.1, 0.3, 0, wi, v)
        assert np.allclose(ev, k["eval_theta_sweep"], rtol=1e-4, atol=1e-6) and np.allclose(pdf, k["pdf_theta_sweep"], rtol=1e-4, atol=1e-6)
        ev, pdf, _, _, _ = _mf(orc, mi, 0, 0.1, 0.1, 0, wi, v)
        assert np.allclose(ev, k["eval_i_theta_sweep"], rtol=1e-4, atol=1e-6) and np.allclose(pdf, k["pdf_i_theta_sweep"], rtol=1e-4, atol=1e-6)
        v = _dirs(np.full(steps, 0.1), np.linspace(0, 2 * np.pi, steps))
        ev, pdf, _, _, _ = _mf(orc, mi, 0, 0.1, 0.3, 0, wi, v)
        assert np.allclose(ev, k["eval_phi_sweep"], rtol=1e-4) and np.allclose(pdf, np.array(k["eval_phi_sweep"]) * np.cos(0.1), rtol=1e-4)
        ev, pdf, _, _, _ = _mf(orc, mi, 0, 0.1, 0.1, 0, wi, v)
        assert np.allclose(ev, k["eval_i_phi_sweep"], rtol=1e-4) and np.allclose(pdf, k["eval_i_phi_sweep"] * np.cos(0.1), rtol=1e-4)
    v = _dirs(np.linspace(np.pi / 3, np.pi / 2, steps), np.full(steps, np.pi / 2))
    assert np.allclose(_mf(orc, mi, type_, 0.1, 0.3, 0, wi, v)[2], k["g1_theta_sweep"], atol=2e-5)
    assert np.allclose(_mf(orc, mi, type_, 0.1, 0.1, 0, wi, v)[2], k["g1_i_theta_sweep"], atol=2e-5)
    v = _dirs(np.full(steps, np.pi / 2 * 0.98), np.linspace(0, 2 * np.pi, steps))
    assert np.allclose(_mf(orc, mi, type_, 0.1, 0.3, 0, wi, v)[2], k["g1_phi_sweep"], rtol=1e-4)
    assert np.allclose(_mf(orc, mi, type_, 0.1, 0.1, 0, wi, v)[2], k["g1_i_phi_sweep"], rtol=1e-4)
    u = np.linspace(0, 1, 6, endpoint=False)
    u1, u2 = np.meshgrid(u, u)
    samples = np.stack([u1.ravel(), u2.ravel()], 1)
    _, _, _, m, pdf = _mf(orc, mi, type_, 0.1, 0.3, 0, wi, samples=samples)
    assert np.allclose(pdf, k["sample_pdf"], rtol=1e-4, atol=1e-4)
    assert np.allclose(m[:12], k["sample_m_first_rows"], atol=5e-4)
    assert np.allclose((m ** 2).sum(1), 1, atol=1e-5)


@pytest.mark.parametrize("type_,alpha,sv,angle", [(0, 0.125, 1, 15), (0, 0.5, 1, 80), (1, 0.125, 1, 30), (1, 0.5, 1, 80), (0, 0.5, 0, 30), (1, 0.125, 0, 15)])
def test_microfacet_sample_matches_pdf(orc, mi, type_, alpha, sv, angle):
    """test_microfacet.py:303-324 (chi^2) in histogram form: sample() is distributed according to pdf()."""
    rng = np.random.default_rng(3)
    n = 400000
    th = np.radians(angle); wi = [np.sin(th), 0, np.cos(th)]
    s = rng.random((n, 2), dtype=np.float32)
    _, _, _, m, _ = _mf(orc, mi, type_, alpha, alpha, sv, wi, samples=s)
    nb = 24
    ct = np.clip(m[:, 2], -1, 1); ph = np.arctan2(m[:, 1], m[:, 0])
    hist, ce, pe = np.histogram2d(ct, ph, bins=[nb, nb], range=[[0, 1], [-np.pi, np.pi]])
    # integrate pdf over each (cos theta, phi) cell with a 20x20 midpoint rule
    sub = 20
    cc = (ce[:-1, None] + (np.arange(sub) + 0.5)[None, :] * (ce[1] - ce[0]) / sub).ravel()
    pp = (pe[:-1, None] + (np.arange(sub) + 0.5)[None, :] * (pe[1] - pe[0]) / sub).ravel()
    C_, P_ = np.meshgrid(cc, pp, indexing="ij")
    st = np.sqrt(np.maximum(0, 1 - C_ ** 2))
    v = np.stack([st * np.cos(P_), st * np.sin(P_), C_], -1).reshape(-1, 3)
    pdf = _mf(orc, mi, type_, alpha, alpha, sv, wi, v)[1].reshape(nb, sub, nb, sub)
    expect = pdf.mean(axis=(1, 3)) * (ce[1] - ce[0]) * (pe[1] - pe[0]) * n
    assert abs(expect.sum() / n - 1) < 0.02                      # pdf integrates to ~1 over the hemisphere
    big = expect > 50
    z = (hist[big] - expect[big]) / np.sqrt(expect[big])
    assert np.abs(z).max() < 6 and abs(z.mean()) < 0.5 and z.std() < 1.6


def test_fresnel(orc, mi, kats):
    l = orc.lib(mi.abi)
    l.orc_fresnel.argtypes = [C.c_float, C.c_float, C.c_void_p]
    l.orc_fresnel_conductor.restype = C.c_float; l.orc_fresnel_conductor.argtypes = [C.c_float] * 3

    def fr(c, eta):
        out = np.zeros(4, dtype=np.float32); l.orc_fresnel(c, eta, _p(out)); return out
    k = kats["fresnel"]
    for c, eta, expected in k["cases"]:
        assert np.allclose(fr(c, eta), expected, atol=1e-6)
    F, ctt, _, scale = fr(np.cos(np.radians(45)), 1.5)
    a = k["deg45_to_1p5"]
    assert np.isclose(ctt, -np.cos(np.radians(a["cos_theta_t_deg"])), atol=1e-6) and np.isclose(F, 0.5 * (a["a_s"] ** 2 + a["a_p"] ** 2), atol=1e-6)
    assert np.isclose((scale * np.sqrt(1 - np.cos(np.radians(45)) ** 2)) ** 2 + ctt ** 2, 1, atol=1e-6)
    F, ctt, _, _ = fr(np.cos(np.radians(45)), 1 / 1.5)
    assert np.isclose(F, 1) and np.isclose(ctt, 0)                          # total internal reflection
    a = k["deg10_from_1p5"]
    F, ctt, _, _ = fr(np.cos(np.radians(10)), 1 / 1.5)
    assert np.isclose(ctt, -np.cos(np.radians(a["cos_theta_t_deg"])), atol=1e-6) and np.isclose(F, 0.5 * (a["a_s"] ** 2 + a["a_p"] ** 2), atol=1e-6)
    # test_fresnel.py:56-66: conductor with a real IOR == dielectric; Snell's law (:69-75); index matched (:48-53)
    for c in np.cos(np.linspace(0, np.pi / 2, 20)):
        for eta in (1.5, 1 / 1.5):
            assert np.isclose(fr(c, eta)[0], l.orc_fresnel_conductor(c, eta, 0.0), atol=2e-6)
    for th in np.linspace(0, np.pi / 2, 20):
        _, ctt, _, _ = fr(np.cos(th), 1.5)
        assert abs(np.sin(th) - 1.5 * np.sin(np.arccos(-ctt))) < 1e-5
    for c in np.linspace(-1, 1, 20):
        F, ctt, _, _ = fr(c, 1.0)
        assert F == 0 and abs(ctt + c) < 5e-7


def _bsdf_desc(mi, props):
    sc = mi.load_dict({"type": "scene", "r": {"type": "rectangle", "bsdf": props}})
    d = sc.desc()
    return sc, d.bsdfs[0]


def test_dielectric_spot_check(orc, mi, kats):
    k = kats["dielectric_spot_check"]
    l = orc.lib(mi.abi)
    l.orc_bsdf_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    sc, b = _bsdf_desc(mi, {"type": "dielectric", "specular_reflectance": 0.3, "specular_transmittance": 0.6, "int_ior": 1.5, "ext_ior": 1.0})
    ang = np.radians(k["angle_deg"]); wi = np.array([np.sin(ang), 0, np.cos(ang)], dtype=np.float32)

    def sample(wi, s1):
        wo = np.zeros(3, dtype=np.float32); ped = np.zeros(3, dtype=np.float32); w = np.zeros(3, dtype=np.float32); s2 = np.zeros(2, dtype=np.float32)
        l.orc_bsdf_sample(C.byref(b), _p(wi), s1, _p(s2), _p(wo), _p(ped), _p(w)); return wo, ped, w
    wo, ped, w = sample(wi, 0.0)
    assert np.isclose(ped[0], k["reflect_pdf"], atol=1e-6) and np.allclose(wo, [-np.sin(ang), 0, np.cos(ang)], atol=1e-6) and ped[2] == 1
    assert np.allclose(w, 0.3)
    wo, ped, w = sample(wi, 1.0)
    a2 = np.radians(k["refract_angle_deg"])
    assert np.isclose(ped[0], 1 - k["reflect_pdf"], atol=1e-6) and np.allclose(wo, [-np.sin(a2), 0, -np.cos(a2)], atol=1e-6)
    assert np.isclose(ped[1], 1.5) and np.allclose(w, 0.6 / 1.5 ** 2, rtol=1e-5)     # radiance scaling eta_ti^2
    wo2, ped2, _ = sample(wo.copy(), 1.0)
    assert np.isclose(ped2[0], 1 - k["reflect_pdf"], atol=1e-5) and np.allclose(wo2, wi, atol=1e-5)


def test_roughconductor_eval_pdf_sample_consistency(orc, mi):
    """test_rough_conductor.py:98-117 + sampling identity: weight * pdf == eval at the sampled direction."""
    l = orc.lib(mi.abi)
    l.orc_bsdf_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    l.orc_bsdf_eval_pdf.argtypes = [C.c_void_p] * 5
    rng = np.random.default_rng(1)
    for props in ({"type": "roughconductor"}, {"type": "roughconductor", "distribution": "ggx", "alpha_u": 0.05, "alpha_v": 0.3, "eta": [0.2, 0.9, 1.1], "k": [3.9, 2.4, 2.2]},
                  {"type": "roughconductor", "alpha": 0.3, "sample_visible": False, "eta": 1.5, "k": 0.5}):
        sc, b = _bsdf_desc(mi, props)
        for _ in range(50):
            th = rng.uniform(0.05, 1.4); ph = rng.uniform(0, 2 * np.pi)
            wi = np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], dtype=np.float32)
            s2 = rng.random(2).astype(np.float32)
            wo = np.zeros(3, dtype=np.float32); ped = np.zeros(3, dtype=np.float32); w = np.zeros(3, dtype=np.float32)
            l.orc_bsdf_sample(C.byref(b), _p(wi), 0.5, _p(s2), _p(wo), _p(ped), _p(w))
            val = np.zeros(3, dtype=np.float32); pdf = np.zeros(1, dtype=np.float32)
            l.orc_bsdf_eval_pdf(C.byref(b), _p(wi), _p(wo), _p(val), _p(pdf))
            if w.max() > 0:
                assert np.isclose(pdf[0], ped[0], rtol=2e-3, atol=1e-6)
                assert np.allclose(w * ped[0], val, rtol=3e-3, atol=1e-6)
            assert ped[1] == 1 and ped[2] == 0


def test_sphere_shape(orc, mi):
    T = mi.ScalarTransform4f
    rot = T.rotate([1, 0, 0], 35)
    s1 = mi.load_dict({"type": "sphere", "radius": 2.0, "center": [1, 0, 0], "to_world": rot})
    s2 = mi.load_dict({"type": "sphere", "to_world": rot @ T.translate([1, 0, 0]) @ T.scale(2)})
    assert np.allclose(s1.to_world.matrix, s2.to_world.matrix, atol=1e-6) and s1.primitive_count() == 1
    assert np.isclose(mi.load_dict({"type": "sphere"}).surface_area(), 4 * np.pi)
    for r in (1, 2, 4):
        lo, hi = mi.load_dict({"type": "sphere", "radius": r}).bbox()
        assert np.allclose(lo, -r) and np.allclose(hi, r)
    # ray casting against the analytic answer (test_sphere.py:50-85): grid of parallel rays, hit iff x^2+y^2 <= r^2
    for r in (1, 3):
        sc = mi.load_dict({"type": "scene", "s": {"type": "sphere", "radius": r, "to_world": T.translate([0, 1, 0]) @ T.rotate([0, 1, 0], 30.0)}})
        o = orc.OracleScene(sc, mi.abi)
        n = 21
        xs = (np.arange(n) + 0.5) / n * 2 * r - r
        X, Y = np.meshgrid(xs, xs)
        rays = np.zeros((n * n, 8), dtype=np.float32)
        rays[:, 0] = X.ravel(); rays[:, 1] = Y.ravel() + 1; rays[:, 2] = -8; rays[:, 5] = 1; rays[:, 6] = np.finfo(np.float32).max
        hits, si = o.ray_intersect(rays, with_si=True)
        h = orc.unpack_hits(hits)
        inside = (X.ravel() ** 2 + Y.ravel() ** 2) <= r * r
        edge = np.abs(np.sqrt(X.ravel() ** 2 + Y.ravel() ** 2) - r) < 1e-3
        assert np.array_equal(np.isfinite(h["t"])[~edge], inside[~edge])
        ok = inside & ~edge
        assert np.allclose(h["t"][ok], 8 - np.sqrt(r * r - X.ravel()[ok] ** 2 - Y.ravel()[ok] ** 2), atol=1e-4)
        nrm = si[ok, 3:6]; p = si[ok, 0:3]
        assert np.allclose(nrm, (p - [0, 1, 0]) / r, atol=1e-4) and np.allclose(np.linalg.norm(p - [0, 1, 0], axis=1), r, atol=1e-4)
        assert (si[ok, 6] >= 0).all() and (si[ok, 6] <= 1).all() and (si[ok, 7] >= 0).all() and (si[ok, 7] <= 1).all()


def test_disk_shape(orc, mi):
    """The reference's own checks for the disk (src/shapes/tests/test_disk.py): test01 area = pi, test02 area / bbox under scale and
    translation, test03 ray_test on a grid <=> x^2 + y^2 <= r^2 with the tangents agreeing with finite differences of si.p, test05's
    uv = (r, phi / 2pi) at (0.9999999, 0)."""
    T = mi.ScalarTransform4f
    sc, o = _shape_scene(mi, orc, {"type": "disk"})
    assert sc.shapes[0].primitive_count() == 1
    assert np.isclose(o.surface_area(0), np.pi) and np.isclose(sc.shapes[0].surface_area(), np.pi)
    for sx in [1, 2, 4]:
        for tr in ([1.3, -3.0, 5], [-10000, 3.0, 31]):
            sc, o = _shape_scene(mi, orc, {"type": "disk", "to_world": T.translate(tr) @ T.scale((sx, 2.5, 1.0))})
            assert np.isclose(o.surface_area(0), sx * 2.5 * np.pi, rtol=1e-5) and np.isclose(sc.shapes[0].surface_area(), sx * 2.5 * np.pi, rtol=1e-5)
            lo, hi = sc.shapes[0].bbox()
            assert np.allclose(lo, np.array(tr) - [sx, 2.5, 0]) and np.allclose(hi, np.array(tr) + [sx, 2.5, 0])
    for r in (1, 3, 5):
        for tr in ([0.0, 0.0, 0.0], [1.0, -5.0, 0.0]):
            sc, o = _shape_scene(mi, orc, {"type": "disk", "to_world": T.translate(tr) @ T.scale((r, r, 1.0))})
            g = np.linspace(-1, 1, 10)
            X, Y = np.meshgrid(1.1 * r * (g - tr[0]), 1.1 * r * (g - tr[1]))
            rays = np.zeros((100, 8), dtype=np.float32)
            rays[:, 0] = X.ravel(); rays[:, 1] = Y.ravel(); rays[:, 2] = -10; rays[:, 5] = 1; rays[:, 6] = np.finfo(np.float32).max
            hits, si = o.ray_intersect(rays, with_si=True)
            h = orc.unpack_hits(hits)
            inside = (rays[:, 0] - tr[0]) ** 2 + (rays[:, 1] - tr[1]) ** 2 <= r * r
            assert np.array_equal(h["occluded"].astype(bool), inside) and np.array_equal(np.isfinite(h["t"]), inside)
            assert np.allclose(h["t"][inside], 10.0) and np.allclose(si[inside, 3:6], [0, 0, 1])
            rad = np.hypot(rays[inside, 0] - tr[0], rays[inside, 1] - tr[1]) / r
            assert np.allclose(si[inside, 6], rad, atol=1e-5) and (si[inside, 7] >= 0).all() and (si[inside, 7] <= 1).all()
    sc, o = _shape_scene(mi, orc, {"type": "disk"})
    rays = np.array([[0.9999999, 0.0, -10.0, 0, 0, 1, np.finfo(np.float32).max, 0]], dtype=np.float32)
    hits, si = o.ray_intersect(rays, with_si=True)
    assert np.allclose(si[0, 6:8], [0.9999999, 0.0], atol=1e-6)
    # flip_normals is folded into the transform (disk.cpp:97-103)
    d = mi.load_dict({"type": "disk", "flip_normals": True})
    assert np.allclose(d.to_world.matrix[:3, 2], [0, 0, -1])


# ---- scope row (f)-1: realistic lens camera --------------------------------------------------------------------
def test_lens_materials(mi):
    """The reference's only pins for the fork-specific lens code: inline asserts of realisticlens.cpp:2444-2461."""
    from mitsuba3_camera_amd import lens
    m = lens.DispersiveMaterial("BK7", cauchy=(1.5046, 0.00420))
    assert np.allclose([m.compute_ior(w) for w in (2.3254, 0.5893, 0.3126)], [1.505376701, 1.516694179, 1.547580488], rtol=1e-6)
    m2 = lens.DispersiveMaterial("NBK7", sellmeier=[(1.03961212, 0.006000699), (0.231792344, 0.020017914), (1.01046945, 103.560653)])
    assert np.allclose([m2.compute_ior(w) for w in (2.3254, 0.5893, 0.3126)], [1.489211725, 1.516727674, 1.548606931], rtol=1e-6)
    assert 60 < m2.abbe_number() < 68          # N-BK7: V_d ~ 64.2


def _lens_scene(mi, design, res=32, diagonal=30.0, **extra):
    T = mi.ScalarTransform4f
    d = mi.cornell_box()
    d["sensor"] = dict({"type": "realisticlens", "lens_design": design, "object_distance": 3.9, "lens_focal_length": 0.05, "lens_diameter": 0.02,
                        "near_clip": 0.001, "far_clip": 100.0, "to_world": T.look_at([0, 0, 3.9], [0, 0, 0], [0, 1, 0]),
                        "sampler": {"type": "independent", "sample_count": 16},
                        "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}, "pixel_format": "rgb", "diagonal": diagonal}}, **extra)
    return mi.load_dict(d)


def test_asphere_reduces_to_sphere_and_gradient(orc, mi):
    """AsphericalLens (realisticlens.cpp:553-821): with kappa = 0 and all a_i = 0 the conic is a sphere, so the traced ray must
    agree with the SpheroidLens of the same radius; the analytic sag gradient agrees with a finite difference of the sag
    (the reference's own _test_asph_grad self-check, :734-768); the float64 host trace agrees with the float32 oracle."""
    from mitsuba3_camera_amd import lens
    glass = lens.DispersiveMaterial("g", cauchy=(1.5, 0.0))
    R, er, z = 0.05, 0.01, 0.03
    sph = lens.Interface(lens.SPHERE, er, z, lens.AIR, glass, curvature_radius=R)
    asp = lens.Interface(lens.ASPHERE, er, z, lens.AIR, glass, curvature_radius=-1000.0 * R, kappa=0.0, ai=[0.0] * 7)   # SpheroidLens(-0.001*R_mm), :1636
    rng = np.random.default_rng(0)
    for _ in range(50):
        o = np.array([rng.uniform(-0.004, 0.004), rng.uniform(-0.004, 0.004), 0.0]); d = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 1.0]); d /= np.linalg.norm(d)
        a, b = lens.trace([sph], o, d), lens.trace([asp], o, d)
        assert a is not None and b is not None
        assert np.allclose(a[0], b[0], atol=2e-7) and np.allclose(a[1], b[1], atol=2e-6)
    wavy = lens.Interface(lens.ASPHERE, er, z, lens.AIR, glass, curvature_radius=40.0, kappa=-1.3, ai=[-2e-4, 3e-6, 0, 0, 0, 0, 0])
    for r in (0.001, 0.004, 0.008):
        h = 1e-6
        fd = (wavy.asph_eval((r + h) ** 2) - wavy.asph_eval((r - h) ** 2)) / (2 * h)
        n = wavy.asph_normal(np.array([r, 0.0, 0.0]))                      # n ~ (dz/dr, 0, -1)
        assert abs(n[0] / -n[2] - fd) < 1e-5 * max(1.0, abs(fd))


@pytest.mark.parametrize("design", ["singlet", "doublet", "exp1a", "tessar", "helios", "jupiter", "fisheye", "gauss", "exp2a", "exp2b", "exp2c"])
def test_lens_focus_and_trace(orc, mi, design):
    """Paraxial self-check in the spirit of notebooks/scripts/lens_design/design_problem.py:430-472: after the constructor's
    thick-lens fine focus (realisticlens.cpp:899-905, :2161-2171) rays leaving one film point through different points of a
    small central aperture zone meet again at the focus distance; the chief ray of an off-axis film point leaves on the other
    side of the axis (inverting lens); weights are cos^4 and blocked rays carry weight 0."""
    sc = _lens_scene(mi, design, 64)
    o = orc.OracleScene(sc, mi.abi)
    n = 64
    rng = np.random.default_rng(0)
    # paraxial zone of the rear element (exp2: the rear "element" is a filter plate 1.35 mm in front of the film and 27 mm in
    # radius -- without the exit-pupil LUT almost every sample of its disk is vignetted, as in the reference's default mode)
    zone = 0.004 if design.startswith("exp2") else 0.15
    ap = 0.5 + (rng.random((n, 2), dtype=np.float32) - 0.5) * zone
    # (exp2, off-axis film points: every ray towards the centre of that plate is vignetted -- on-axis point only)
    for px in ([32.0, 32.0],) if design.startswith("exp2") else ([32.0, 32.0], [40.0, 28.0]):
        rays = o.lens_rays(np.tile(np.array(px, dtype=np.float32), (n, 1)), ap)
        ok = rays[:, 7] > 0
        assert ok.mean() > 0.9
        r = rays[ok]
        # camera looks down -z from z = 3.9: focus plane is z = 0
        t = (0.0 - r[:, 2]) / r[:, 5]
        hit = r[:, 0:3] + t[:, None] * r[:, 3:6]
        spread = hit[:, :2].std(axis=0).max()
        assert spread < 2e-3, (design, px, spread)                              # << the 2 m scene width
        c4 = (r[:, 3:6] @ np.array([0, 0, -1.0])) ** 4
        assert np.allclose(r[:, 7], c4, rtol=1e-4)
        if px[0] != 32.0:                                                       # +x/-y on the film -> same side in the image (upright mapping, like perspective)
            assert abs(hit[:, 0].mean()) > 1e-3
    wide = o.lens_rays(np.tile(np.array([32.0, 32.0], dtype=np.float32), (256, 1)), rng.random((256, 2), dtype=np.float32))
    assert (wide[:, 7] >= 0).all() and (wide[:, 7] <= 1 + 1e-5).all()
    if design == "singlet":                                                     # unknown names fall back to the singlet (:896-898)
        assert np.array_equal(_lens_scene(mi, "no-such-design", 64).sensor.lens_table, sc.sensor.lens_table)


def test_radical_inverse_and_time_sample(orc, mi):
    """RadicalInverse::eval (qmc.h:67-92) in bases 2 and 3 -- van der Corput values are exact binary fractions, base 3 follows
    the float recurrence -- and the extra sampler draw a sensor with shutter_open_time > 0 costs (integrator.cpp:442-444)."""
    l = orc.lib(mi.abi)
    assert [l.orc_radical_inverse(2, i) for i in range(8)] == [0.0, 0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875]
    third = np.float32(1) / np.float32(3)
    assert l.orc_radical_inverse(3, 1) == float(third) and l.orc_radical_inverse(3, 2) == float(np.float32(2) * third)
    assert abs(l.orc_radical_inverse(3, 5) - (2 / 3 + 1 / 9)) < 1e-7 and l.orc_radical_inverse(2, (1 << 20) - 1) < 1.0
    d = small_cbox_for(mi, 8)
    a = orc.OracleScene(mi.load_dict(d), mi.abi)
    d["sensor"]["shutter_open"] = 0.25; d["sensor"]["shutter_close"] = 0.75
    sc_b = mi.load_dict(d); b = orc.OracleScene(sc_b, mi.abi)
    assert sc_b.sensor.shutter_open_time == 0.5
    integ = mi.PathIntegrator({"max_depth": 3})
    la, lb = a.lane_radiance(a.params(integ, 1, 4), 0, 256), b.lane_radiance(b.params(integ, 1, 4), 0, 256)
    assert not np.allclose(la, lb)                                   # the light paths use a shifted sample stream ...
    wa, wb = a.splat_weights(a.params(integ, 1, 4)), b.splat_weights(b.params(integ, 1, 4))
    assert np.array_equal(wa, wb)                                    # ... the film positions (drawn first) do not move
    with pytest.raises(mi.PluginError):
        d["sensor"]["shutter_close"] = 0.1; mi.load_dict(d)


def test_exit_pupil_bounds(orc, mi):
    """compute_exit_pupil_bounds (realisticlens.cpp:1685-1806) on the singlet: the bounds are symmetric about the meridional
    plane (p_film sits on the +x axis), contain the axis for the central ring, stay inside the sampled (1.5 x rear radius)^2
    square plus its padding, and every ray the sensor then samples starts inside its ring's bounds; the ray weight is
    shutter_open_time * bounds_area / z_rear^2 (:2282-2285)."""
    sc = _lens_scene(mi, "singlet", 32, sample_exit_pupil=True, shutter_close=2.0)
    o = orc.OracleScene(sc, mi.abi)
    b = o.exit_pupil_bounds()
    rr = 1.5 * sc.sensor.interfaces[0].element_radius; pad = 4 * rr * np.sqrt(2 / (1 << 20))
    assert b.shape == (64, 4) and np.allclose(b[:, 1], -b[:, 3], atol=2e-5) and (b[:, 0] >= -rr - pad - 1e-7).all() and (b[:, 2] <= rr + pad + 1e-7).all()
    assert b[0, 0] < 0 < b[0, 2] and (b[:, 2] > b[:, 0]).all()
    assert b[40, 2] - b[40, 0] <= b[0, 2] - b[0, 0] + 1e-6               # vignetting narrows the pupil off-axis
    rng = np.random.default_rng(2)
    pos = rng.uniform(0, 32, (256, 2)).astype(np.float32)
    rays = o.lens_rays(pos, rng.random((256, 2), dtype=np.float32))
    ok = rays[:, 7] > 0
    assert ok.mean() > 0.4                                               # the box over-covers the round pupil, but not by much
    z = sc.sensor.interfaces[0].z
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    got = np.unique(np.round(rays[ok, 7] * z * z / 2.0, 10))
    assert all(np.isclose(areas, g, rtol=1e-4).any() for g in got)


@pytest.mark.parametrize("design", ["asph", "hypercentric"])
def test_lens_designs_without_fine_focus(orc, mi, design):
    """`asph` (13 aspheres, realisticlens.cpp:1373-1449) and `hypercentric` (:1322-1370, film not at z = 0) skip the fine focus:
    check the tables load, paraxial rays get through with cos^4 weights, and the hypercentric camera frame is shifted by
    film_z_position (:2263)."""
    from mitsuba3_camera_amd import lens
    sc = _lens_scene(mi, design, 32, diagonal=6.0 if design == "asph" else 10.0)
    assert np.isnan(sc.sensor.focus_delta) and sc.sensor.lens_table.shape == ((15, 16) if design == "asph" else (3, 16))
    o = orc.OracleScene(sc, mi.abi)
    rng = np.random.default_rng(1)
    zone = 0.01 if design == "asph" else 0.002          # (rear plates sit 0.2 mm / a few mm in front of the film)
    ap = 0.5 + (rng.random((128, 2), dtype=np.float32) - 0.5) * zone
    rays = o.lens_rays(np.tile(np.array([16.0, 16.0], dtype=np.float32), (128, 1)), ap)
    ok = rays[:, 7] > 0
    assert ok.mean() > 0.9
    assert np.allclose(rays[ok, 7], (rays[ok, 3:6] @ np.array([0, 0, -1.0])) ** 4, rtol=1e-4)
    if design == "hypercentric":
        _, fz = lens.build_hypercentric()
        assert abs(fz - 0.001 * (450.0 - (50.0 + 395.0 + 420.0 * 4.0 / 121.0))) < 1e-9
        ref = mi.ScalarTransform4f.look_at([0, 0, 3.9], [0, 0, 0], [0, 1, 0])
        shift = sc.sensor.to_world.transform_affine([0, 0, 0]) - ref.transform_affine([0, 0, 0])
        assert np.allclose(shift, [0, 0, -fz], atol=1e-6)            # camera looks down -z: +film_z along its own axis


def test_lens_render_oracle(orc, mi):
    sc = _lens_scene(mi, "doublet", 32, diagonal=50.0)
    o = orc.OracleScene(sc, mi.abi)
    integ = mi.PathIntegrator({"max_depth": 4})
    img, n = o.render(o.params(integ, 0, 16))
    assert np.isfinite(img).all() and img.mean() > 0.01
    sc_p = mi.load_dict(small_cbox_for(mi, 32))
    ref, _ = orc.OracleScene(sc_p, mi.abi).render(o.params(integ, 0, 16))
    # same framing as the 39.3 deg pinhole would need a matched focal length; just check the gross structure: red left, green right
    assert img[12:20, 2:6, 0].mean() > 2 * img[12:20, 2:6, 1].mean() and img[12:20, 26:30, 1].mean() > 1.5 * img[12:20, 26:30, 0].mean()


def small_cbox_for(mi, res):
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    return d


# ---- the geometry derivative, pinned by the closed-form AD assertions the reference's own tests hold ------------------------------
def _allclose(a, b, atol, rtol=1e-5):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))          # dr.allclose


def test_mesh_hit_derivative_matches_the_references_ad_assertions(orc, mi, kats):
    """src/render/tests/test_mesh.py:380-640 (test13 - test16) on the oracle's multi-dual mesh hit (oracle/mi_oracle_geo.inc: md_mesh_hit),
    the function every geometry-gradient parity test of the device is compared with.  Columns of the Jacobian are the reference's forward
    assertions, rows its backward ones.  Input order: ray origin 0-2, direction 3-5, the triangle's positions 6-14, its normals 15-23."""
    K = kats["mesh_ad"]; l = orc.lib(mi.abi); atol = K["atol"]
    V = np.array(K["vertices"], dtype=np.float32); tri = K["triangle"]; P = V[tri]

    def vertex_grad(row):                                                    # a row of the Jacobian over the 3 positions -> the 12-vector of the mesh
        g = np.zeros(12)
        for k, v in enumerate(tri):
            g[3 * v:3 * v + 3] = row[6 + 3 * k:9 + 3 * k]
        return g
    # test13 (:380-424) forward in the ray; test14 (:427-452) backward to the ray
    c = K["ray_forward"]; val, J = orc.geo_mesh_hit_jacobian(l, P, None, c["o"], c["d"])
    assert _allclose(J[0:3, 0], c["dp_dox"], atol) and _allclose(J[0:3, 3], c["dp_ddx"], atol)
    dt_do = J[2, 0:3] - np.array([0, 0, 1.0])                               # si.t = (si.p - ray.o) . d for d = (0, 0, 1)
    assert _allclose(dt_do[2], c["dt_doz"], atol)
    c = K["ray_backward"]; val, J = orc.geo_mesh_hit_jacobian(l, P, None, c["o"], c["d"])
    assert _allclose(J[0, 0:3], c["grad_o_of_px"], atol) and _allclose(J[2, 0:3] - np.array([0, 0, 1.0]), c["grad_o_of_t"], atol)
    # test15 (:455-531): a rigid translation of all vertices along z moves si.t by 1 and si.p by (0, 0, 1)
    c = K["params_forward"]; val, J = orc.geo_mesh_hit_jacobian(l, P, None, c["o"], c["d"])
    dz = J[0:3, 8] + J[0:3, 11] + J[0:3, 14]
    assert _allclose(dz, c["translate_z"]["dp"], atol) and _allclose(dz[2], c["translate_z"]["dt"], atol)
    # test16 (:536-640): the ray that hits the corner
    c = K["params_backward"]; val, J = orc.geo_mesh_hit_jacobian(l, P, None, c["o"], c["d"])
    assert _allclose(vertex_grad(J[2]), c["grad_of_pz"], atol) and _allclose(vertex_grad(J[2]), c["grad_of_t"], atol)
    assert _allclose(vertex_grad(J[3]), c["grad_of_nx"], atol) and _allclose(vertex_grad(J[4]), c["grad_of_ny"], atol)
    # ... and the interpolated normal of the flat mesh with vertex normals that FOLLOW the positions (recompute_vertex_normals): the chain
    # d sh_n / d n_i . d n_i / d positions; vertex 3 belongs to this triangle only, and for a flat mesh the other triangle's share of
    # n_0 and n_1 drops out of the corner ray's derivative (its weights are 1e-5)
    N = np.tile(np.array([0, 0, 1], dtype=np.float32), (3, 1))
    val, Jn = orc.geo_mesh_hit_jacobian(l, P, N, c["o"], c["d"])
    b = [1.0 - val[0] * 0 - 0, 0, 0]                                        # (weights are read off the Jacobian below)
    # d sh_n / d n_k = b_k (I - n n^T) for unit normals; n_k = normalize(sum_f angle_f n_f): at vertex 3 only this face, angle pi / 2
    w3 = Jn[6, 15 + 3 * 1]                                                  # d sh_n.x / d n_{vertex 3}.x = b_1 (vertex 3 is the triangle's second corner)
    assert abs(w3 - 1.0) < 2e-5
    g_sh_nx = vertex_grad(Jn[6]) + w3 * vertex_grad(J[3])                   # direct part (none: sh_n ignores positions at fixed normals) + through n_3 = the face normal
    g_sh_ny = vertex_grad(Jn[7]) + w3 * vertex_grad(J[4])
    assert _allclose(g_sh_nx, c["grad_of_sh_nx"], 2e-5) and _allclose(g_sh_ny, c["grad_of_sh_ny"], 2e-5)


def test_plane_hit_derivative_matches_the_references_ad_assertions(orc, mi, kats):
    """src/shapes/tests/test_rectangle.py:116-171 (test06 / test07) on the oracle's fixed-plane hit (rectangle / disk)."""
    K = kats["rectangle_ad"]; l = orc.lib(mi.abi)
    val, J = orc.geo_plane_hit_jacobian(l, [0, 0, 0], [0, 0, 1], K["o"], K["d"])
    assert _allclose(J[:, 0], K["dp_dox"], 1e-5) and _allclose(J[:, 1], K["dp_doy"], 1e-5) and _allclose(J[:, 3], K["dp_ddx"], 1e-5)
    assert _allclose(J[0, 0:3], K["grad_o_of_px"], 1e-5) and _allclose(J[2, 0:3] - np.array([0, 0, 1.0]), K["grad_o_of_t"], 1e-5)
    assert _allclose((J[2, 0:3] - np.array([0, 0, 1.0]))[2], K["dt_doz"], 1e-5)


def test_vertex_normal_weighting_scheme(orc, mi, kats):
    """src/render/tests/test_mesh.py:74-98 (test04): angle-weighted vertex normals on the reference's 5-vertex / 2-face mesh -- the oracle's
    recompute_vertex_normals and the host mirror's compute_vertex_normals (the device's k_vertex_normals: tests/test_gpu_geometry.py)."""
    K = kats["mesh_normal_weighting"]; l = orc.lib(mi.abi)
    pos = np.array(K["vertex_positions"], dtype=np.float32); faces = np.array(K["faces"], dtype=np.uint32)
    n0, n1 = np.array(K["n0"], dtype=np.float64), np.array(K["n1"], dtype=np.float64)
    n2 = n0 * (math.pi / 2) + n1 * math.acos(3.0 / 5.0); n2 /= np.linalg.norm(n2)
    want = np.stack([n2, n0, n0, n1, n1])
    for got in (orc.vertex_normals(l, pos, faces), mi.scene.compute_vertex_normals(pos, faces)):
        assert np.allclose(got, want, rtol=K["rtol"], atol=1e-6), got
