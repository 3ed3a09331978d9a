"""Self-consistency of the CPU oracle's path / PRB estimators (CPU only).

The reference's EXR images are absent, so the whole-path behaviour is pinned the way the reference's own
AD tests do it (src/integrators/tests/test_ad_integrators.py:742-828): gradients against finite differences of
the primal, plus exact identities of the estimator:
  * radiance is linear in the emitter's `radiance`  =>  d<gin, image>/d rad_c = <gin_c, image_c> / rad_c   (exact)
  * with Russian roulette disabled the estimator is, for fixed random numbers, a polynomial in every albedo
    =>  PRB gradient == central finite difference with the SAME seeds                                      (exact to O(eps^2))
  * path and prb are two estimators of the same integral (they differ only in when RR starts, path.cpp:266-271 vs
    prb.py:192-198) => image means agree within Monte Carlo error; with RR disabled they are lane-identical.
"""
import numpy as np
import pytest

from conftest import small_cbox


def _scene(mi, orc, res=16, **kw):
    sc = mi.load_dict(small_cbox(mi, res, **kw))
    return sc, orc.OracleScene(sc, mi.abi)


def test_cornell_image_sanity(mi, orc):
    sc, o = _scene(mi, orc, 32)
    integ = mi.PathIntegrator({"max_depth": 6})
    img, n = o.render(o.params(integ, 0, 16))
    assert n == 32 * 32 * 16 and np.isfinite(img).all() and (img >= 0).all()
    # red wall on the left, green on the right (camera flips x: sensor.h:261), light visible at the top centre
    left, right = img[12:20, 1:4].mean((0, 1)), img[12:20, 28:31].mean((0, 1))
    assert left[0] > 3 * left[1] and right[1] > 2 * right[0]
    assert img[3:5, 14:18].max() > 5.0
    # max_depth = 1: only directly visible emitters
    img1, _ = o.render(o.params(mi.PathIntegrator({"max_depth": 1}), 0, 4))
    assert (img1[16:] == 0).all() and img1.max() > 5.0


def test_depth_is_monotone(mi, orc):
    sc, o = _scene(mi, orc, 16)
    means = [o.render(o.params(mi.PathIntegrator({"max_depth": d, "rr_depth": 50}), 1, 32))[0].mean() for d in (1, 2, 3, 5)]
    assert means[0] < means[1] < means[2] < means[3]


def test_path_and_prb_agree(mi, orc):
    sc, o = _scene(mi, orc, 16)
    # RR disabled: identical lanes (same RNG consumption order, integrator.cpp:435 / path.cpp / prb.py)
    a = o.lane_radiance(o.params(mi.PathIntegrator({"max_depth": 5, "rr_depth": 99}), 3, 8), 0, 16 * 16 * 8)
    b = o.lane_radiance(o.params(mi.PRBIntegrator({"max_depth": 5, "rr_depth": 99}), 3, 8), 0, 16 * 16 * 8)
    assert np.allclose(a[:, :3], b[:, :3], rtol=1e-5, atol=1e-6)
    # RR enabled: same expectation
    ia, _ = o.render(o.params(mi.PathIntegrator({"max_depth": 8, "rr_depth": 2}), 3, 256))
    ib, _ = o.render(o.params(mi.PRBIntegrator({"max_depth": 8, "rr_depth": 2}), 3, 256))
    assert abs(ia.mean() - ib.mean()) < 0.02 * ia.mean()


def test_seed_and_shard_semantics(mi, orc):
    """Lanes are a pure function of (seed, global lane id): shards of the sample plane add up to the whole film, and
    the sampler's base seed adds to the render seed (sampler.cpp:120)."""
    sc, o = _scene(mi, orc, 16)
    integ = mi.PathIntegrator({"max_depth": 4})
    whole, n = o.render(o.params(integ, 5, 8), develop=False)
    acc = np.zeros_like(whole)
    for r in range(3):
        f, _ = o.render(o.params(integ, 5, 8, shard=mi.shard_pixels(256, r, 3)), develop=False)
        acc += f
    assert np.allclose(acc, whole, rtol=1e-5, atol=1e-5)
    acc = np.zeros_like(whole)
    for r in range(4):
        f, _ = o.render(o.params(integ, 5, 8, spp_shard=(2 * r, 2)), develop=False)
        acc += f
    assert np.allclose(acc, whole, rtol=1e-5, atol=1e-5)
    acc = np.zeros_like(whole)                  # interleaved row tiles (2 rows of 16 pixels, 3 ranks; last tiles ragged)
    for r in range(3):
        f, _ = o.render(o.params(integ, 5, 8, shard=mi.shard_tiles(16, 16, r, 3, rows=2)), develop=False)
        acc += f
    assert np.allclose(acc, whole, rtol=1e-5, atol=1e-5)
    d = small_cbox(mi, 16); d["sensor"]["sampler"]["seed"] = 7
    sc2 = mi.load_dict(d); o2 = orc.OracleScene(sc2, mi.abi)
    assert o2.params(integ, 5, 8).seed == 12
    a, _ = o2.render(o2.params(integ, 5, 8)); b, _ = o.render(o.params(integ, 12, 8))
    assert np.array_equal(a, b)


def test_emitter_gradient_linearity_identity(mi, orc):
    sc, o = _scene(mi, orc, 16)
    integ = mi.PRBIntegrator({"max_depth": 6, "rr_depth": 3})
    gin = np.random.default_rng(7).normal(size=(16, 16, 3)).astype(np.float32)
    p = o.params(integ, 21, 16)
    g = o.render_backward(p, gin)
    img, _ = o.render(p)
    lay, total = o.param_layout()
    off = [o_ for k, i, o_, s in lay if k == mi.abi.PARAM_EMITTER_RADIANCE][0]
    expect = (gin * img).sum((0, 1)) / sc.emitters[0].radiance
    assert np.allclose(g[off:off + 3], expect, rtol=2e-4, atol=1e-5)


def _objective(o, p, gin):
    img, _ = o.render(p)
    return float((gin.astype(np.float64) * img).sum())


@pytest.mark.parametrize("name", ["white", "green", "red"])
def test_albedo_gradient_matches_finite_differences(mi, orc, name):
    sc, o = _scene(mi, orc, 16)
    integ = mi.PRBIntegrator({"max_depth": 5, "rr_depth": 99})        # RR off => polynomial in the albedo
    gin = np.random.default_rng(3).uniform(0.5, 1.5, size=(16, 16, 3)).astype(np.float32)
    p = o.params(integ, 9, 8)
    g = o.render_backward(p, gin)
    lay, _ = o.param_layout()
    bi = [i for i, b in enumerate(sc.bsdfs) if b.id == name][0]
    off = [o_ for k, i, o_, s in lay if k == mi.abi.PARAM_BSDF_REFLECTANCE and i == bi][0]
    base = sc.bsdfs[bi].reflectance.copy()
    for c in range(3):
        eps = 2e-2
        hi = base.copy(); hi[c] += eps; lo = base.copy(); lo[c] -= eps
        o.set_param(mi.abi.PARAM_BSDF_REFLECTANCE, bi, hi); fp = _objective(o, p, gin)
        o.set_param(mi.abi.PARAM_BSDF_REFLECTANCE, bi, lo); fm = _objective(o, p, gin)
        o.set_param(mi.abi.PARAM_BSDF_REFLECTANCE, bi, base)
        fd = (fp - fm) / (2 * eps)
        assert abs(g[off + c] - fd) < 2e-2 * max(abs(fd), 1.0), (name, c, g[off + c], fd)


def test_texture_gradient_matches_finite_differences(mi, orc):
    d = small_cbox(mi, 16)
    rng = np.random.default_rng(12345)
    tex = rng.uniform(0.2, 0.8, (4, 4, 3)).astype(np.float32)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    d["large-box"]["bsdf"] = {"type": "ref", "id": "tex"}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    integ = mi.PRBIntegrator({"max_depth": 4, "rr_depth": 99})
    gin = np.ones((16, 16, 3), dtype=np.float32)
    p = o.params(integ, 4, 16)
    g = o.render_backward(p, gin)
    lay, _ = o.param_layout()
    off, size = [(o_, s) for k, i, o_, s in lay if k == mi.abi.PARAM_TEXTURE_DATA][0]
    gt = g[off:off + size].reshape(4, 4, 3)
    assert np.abs(gt).max() > 0
    for (y, x, c) in [(0, 0, 0), (1, 2, 1), (3, 3, 2), (2, 1, 0)]:
        eps = 2e-2
        hi = tex.copy(); hi[y, x, c] += eps; lo = tex.copy(); lo[y, x, c] -= eps
        o.set_param(mi.abi.PARAM_TEXTURE_DATA, 0, hi); fp = _objective(o, p, gin)
        o.set_param(mi.abi.PARAM_TEXTURE_DATA, 0, lo); fm = _objective(o, p, gin)
        o.set_param(mi.abi.PARAM_TEXTURE_DATA, 0, tex)
        fd = (fp - fm) / (2 * eps)
        assert abs(gt[y, x, c] - fd) < 2e-2 * max(abs(fd), 0.5), ((y, x, c), gt[y, x, c], fd)


def test_backward_with_rr_is_unbiased_estimate_of_fd(mi, orc):
    """test_ad_integrators.py:792-828 protocol at small scale: uniform grad_in, PRB gradient (RR on) vs the finite
    difference of a high-spp primal, within a generous statistical tolerance."""
    sc, o = _scene(mi, orc, 8)
    integ = mi.PRBIntegrator({"max_depth": 6, "rr_depth": 2})
    gin = np.full((8, 8, 3), 1e-3, dtype=np.float32)
    lay, _ = o.param_layout()
    bi = [i for i, b in enumerate(sc.bsdfs) if b.id == "white"][0]
    off = [o_ for k, i, o_, s in lay if k == mi.abi.PARAM_BSDF_REFLECTANCE and i == bi][0]
    g = np.mean([o.render_backward(o.params(integ, s, 256), gin)[off:off + 3] for s in range(4)], axis=0)
    base = sc.bsdfs[bi].reflectance.copy()
    fd = np.zeros(3)
    eps = 0.03
    for c in range(3):
        hi = base.copy(); hi[c] += eps; lo = base.copy(); lo[c] -= eps
        o.set_param(mi.abi.PARAM_BSDF_REFLECTANCE, bi, hi); fp = np.mean([_objective(o, o.params(integ, 100 + s, 512), gin) for s in range(2)])
        o.set_param(mi.abi.PARAM_BSDF_REFLECTANCE, bi, lo); fm = np.mean([_objective(o, o.params(integ, 100 + s, 512), gin) for s in range(2)])
        o.set_param(mi.abi.PARAM_BSDF_REFLECTANCE, bi, base)
        fd[c] = (fp - fm) / (2 * eps)
    assert np.all(np.abs(g - fd) < 0.15 * np.abs(fd)), (g, fd)


@pytest.mark.parametrize("wrap_mode", ["repeat", "clamp", "mirror"])
def test_texture_wrap_modes_follow_the_reference_tests(mi, orc, wrap_mode):
    """The reference's own asserts for `bitmap`, enforced on the oracle's texture fetch (the device fetch is held to the same asserts and
    to bit-equality with this one in test_gpu_parity.py::test_device_texture_fetch_identities):
      * src/textures/tests/test_bitmap.py:143-157 (test04): uv = ((x + 1/2) / W, (y + 1/2) / H) returns texel (x, y);
      * src/textures/tests/test_bitmap.py:58-130 (test03): a 20 x 20 grid over [0,1]^2 equals the grid shifted by -1 / +1 (repeat), the
        reversed shifted grid (mirror), the border row / column (clamp); atol 1e-4 as there.
    This pins the dr::Texture2f convention (pos = uv * res - 1/2, 4-tap lerp, wrap applied to the integer taps) to the reference's tests;
    the image (noise_8x8.png) is replaced by random 8 x 8 RGB data -- the reference's resources submodule is not checked out."""
    tex = np.random.default_rng(8).uniform(0.0, 1.0, (8, 8, 3)).astype(np.float32)
    d = small_cbox(mi, 8)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True, "wrap_mode": wrap_mode}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)

    def grid(xs, ys):
        xv, yv = np.meshgrid(xs, ys)
        return o.texture_eval(0, np.stack([xv.ravel(), yv.ravel()], -1).astype(np.float32))
    centres = np.array([[(x + 0.5) / 8, (y + 0.5) / 8] for y in range(8) for x in range(8)], dtype=np.float32)
    assert np.allclose(o.texture_eval(0, centres), tex.reshape(-1, 3), atol=1e-6)
    assert np.allclose(o.texture_eval(0, np.array([[(7 + 0.5) / 8, (1 + 0.5) / 8]], dtype=np.float32))[0], tex[1, 7], atol=1e-6)
    n = 20
    lin = lambda a, b: np.linspace(a, b, n, dtype=np.float32)
    ref = grid(lin(0, 1), lin(0, 1))
    if wrap_mode == "repeat":
        assert np.allclose(ref, grid(lin(-1, 0), lin(-1, 0)), atol=1e-4) and np.allclose(ref, grid(lin(1, 2), lin(1, 2)), atol=1e-4)
    elif wrap_mode == "mirror":
        assert np.allclose(ref, grid(lin(-1, 0)[::-1], lin(-1, 0)[::-1]), atol=1e-4) and np.allclose(ref, grid(lin(1, 2)[::-1], lin(1, 2)[::-1]), atol=1e-4)
    else:
        r3 = ref.reshape(n, n, 3)
        assert np.allclose(np.broadcast_to(r3[:1], (n, n, 3)), grid(lin(0, 1), lin(-1, 0)).reshape(n, n, 3), atol=1e-4)
        assert np.allclose(np.broadcast_to(r3[-1:], (n, n, 3)), grid(lin(0, 1), lin(1, 2)).reshape(n, n, 3), atol=1e-4)
        assert np.allclose(np.broadcast_to(r3[:, :1], (n, n, 3)), grid(lin(-1, 0), lin(0, 1)).reshape(n, n, 3), atol=1e-4)
        assert np.allclose(np.broadcast_to(r3[:, -1:], (n, n, 3)), grid(lin(1, 2), lin(0, 1)).reshape(n, n, 3), atol=1e-4)


def test_texture_lookup(mi, orc):
    """dr::Texture2f convention (texel centres at (i+0.5)/res, repeat wrap), hand-computed taps; the reference's asserts for the same
    convention are enforced in test_texture_wrap_modes_follow_the_reference_tests."""
    d = small_cbox(mi, 8)
    tex = np.arange(2 * 4 * 3, dtype=np.float32).reshape(2, 4, 3)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex / 100, "raw": True}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    centres = np.array([[(x + 0.5) / 4, (y + 0.5) / 2] for y in range(2) for x in range(4)], dtype=np.float32)
    assert np.allclose(o.texture_eval(0, centres), (tex / 100).reshape(-1, 3), atol=1e-6)
    mid = o.texture_eval(0, np.array([[0.25, 0.25]], dtype=np.float32))[0]      # halfway between texels (0,0) and (1,0)
    assert np.allclose(mid, (tex[0, 0] + tex[0, 1]) / 200, atol=1e-6)
    wrap = o.texture_eval(0, np.array([[0.0, 0.25]], dtype=np.float32))[0]      # repeat: between texel 3 and texel 0
    assert np.allclose(wrap, (tex[0, 3] + tex[0, 0]) / 200, atol=1e-6)


def test_particle_tracer_agrees_with_path_tracer(mi, orc):
    """Scope row (f)-3, forward part: the oracle's light tracer (ptracer.cpp:77-371 under AdjointIntegrator::render,
    integrator.cpp:545-745) and its path tracer estimate the same image -- same maximum path length for the same max_depth
    (visible emitters = length 1, a connection at light-path depth k = length k + 1).  Reference methodology: the two
    integrators are compared statistically (test_renders.py); here: image means within 1.5 %, block means within 6 %."""
    d = small_cbox(mi, 24); d["sensor"]["film"]["rfilter"] = {"type": "box"}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    for md in (1, 2, 5):
        integ = mi.PathIntegrator({"max_depth": md, "rr_depth": 3})
        a, _ = o.render(o.params(integ, 0, 1024))
        b, lanes = o.ptracer_render(o.params(integ, 0, 1024))
        assert lanes == 24 * 24 * 1024 and np.isfinite(b).all()
        assert abs(b.mean() / a.mean() - 1) < 0.015, (md, b.mean() / a.mean())
        if md == 5:
            ab = a.reshape(4, 6, 4, 6, 3).mean(axis=(1, 3)); bb = b.reshape(4, 6, 4, 6, 3).mean(axis=(1, 3))
            assert (np.abs(bb - ab) < 0.06 * ab + 1e-3).mean() > 0.95
    integ = mi.PathIntegrator({"max_depth": 4})
    f1, _ = o.ptracer_render(o.params(integ, 7, 16), develop=False)
    f2, _ = o.ptracer_render(o.params(integ, 7, 16), threads=3, develop=False)
    assert np.allclose(f1, f2, rtol=1e-4, atol=1e-6) and (f1[..., 3] == 0).all()          # lanes are functions of (seed, lane); W unused
    z, _ = o.ptracer_render(o.params(mi.PathIntegrator({"max_depth": 0}), 0, 4))
    assert z.max() == 0.0
    integ.hide_emitters = True
    h, _ = o.ptracer_render(o.params(integ, 0, 256)); integ.hide_emitters = False
    v, _ = o.ptracer_render(o.params(integ, 0, 256))
    light = v - h                                                                           # the directly visible emitter only
    assert light.max() > 5 * np.median(v) and (light >= -0.05 * v.max()).all()


def ptracer_test_scene(mi, radiance=(1.0, 1.0, 1.0), **integrator):
    """create_test_scene(emitter='area') of src/integrators/tests/test_ptracer.py:8-108, with the reference's own keys."""
    T = mi.ScalarTransform4f
    props = {"type": "ptracer", "samples_per_pass": 16, "rr_depth": 9, "max_depth": 4, "hide_emitters": False}; props.update(integrator)
    return {"type": "scene", "integrator": props,
            "sensor": {"type": "perspective", "to_world": T.look_at(origin=(2, 0, 0), target=(0, 0, 0), up=(0, 1, 0)),
                       "sampler": {"type": "independent"},
                       "film": {"type": "hdrfilm", "width": 8, "height": 8, "rfilter": {"type": "box"}}},
            "emitter": {"type": "rectangle", "to_world": T.look_at(origin=(0, 0, 0), target=(1, 0, 0), up=(0, 1, 0)),
                        "area_emitter": {"type": "area", "radiance": {"type": "rgb", "value": radiance}}}}


def ptracer_test_scene_directional(mi):
    """create_test_scene(emitter='directionalarea') of test_ptracer.py:63-108: a collimated 2 x 2 beam from (4, 1, 0) towards the
    origin and the receiver plane the reference adds for it."""
    T = mi.ScalarTransform4f
    d = ptracer_test_scene(mi)
    d["emitter"] = {"type": "rectangle", "to_world": T.look_at(origin=(4, 1, 0), target=(0, 0, 0), up=(0, 1, 0)),
                    "area_emitter": {"type": "directionalarea", "radiance": {"type": "rgb", "value": (1.0, 0.5, 0.2)}}}
    d["receiver"] = {"type": "rectangle", "to_world": T.look_at(origin=(0, 0, 0), target=(1, 0, 0), up=(0, 1, 0)), "bsdf": {"type": "diffuse"}}
    return d


def test_particle_tracer_reference_test01_directionalarea(mi, orc):
    """test_ptracer.py:113-118 (test01_render_simple) for the `directionalarea` emitter: seed 0, 2 spp, at least 20 % of the image
    values are lit -- light reaches the film only through sample_ray + one diffuse bounce on the receiver."""
    sc = mi.load_dict(ptracer_test_scene_directional(mi)); o = orc.OracleScene(sc, mi.abi)
    integ = mi.make_integrator(sc.integrator_props)
    img, _ = o.ptracer_render(o.params(integ, 0, 2))
    assert (img > 0).sum() >= 0.2 * img.size
    integ.max_depth = 1                                              # only directly visible emitters: a collimated one is never connected
    z, _ = o.ptracer_render(o.params(integ, 0, 2))
    assert (z == 0).all()


def test_particle_tracer_reference_golden_gradient(mi, orc, kats):
    """The one number the reference's tests hard-code for this integrator (test_ptracer.py:225): d mean(image) / d radiance = 0.33647
    (atol 1e-5) for seed 3, 64 spp in 4 passes.  The image is linear in the radiance, so the gradient is sum(image at radiance 1) /
    (8 * 8 * 3) per channel.  Reproducing it pins, end to end: TEA seeding of the sampler streams, PCG32 floats, the pass structure
    of AdjointIntegrator::render (one seeding, every stream continued across the passes), the order and MASKING of every sampler
    call in sample_visible_emitters / prepare_ray (including the two numbers the JIT variants burn in the infinite-emitter branch),
    Rectangle::sample_position, PerspectiveCamera::sample_direction + importance, sample_scale and the normalised box splat."""
    k = kats["ptracer_radiance_gradient"]
    sc = mi.load_dict(ptracer_test_scene(mi)); o = orc.OracleScene(sc, mi.abi)
    integ = mi.make_integrator(sc.integrator_props)
    assert integ.samples_per_pass == k["samples_per_pass"] and integ.rr_depth == k["rr_depth"] and integ.max_depth == k["max_depth"]
    p = o.params(integ, k["seed"], k["spp"])
    assert p.passes == 4
    img, lanes = o.ptracer_render(p)
    assert lanes == 8 * 8 * 16
    g = img.sum(axis=(0, 1)) / img.size
    assert np.allclose(g, k["gradient"], atol=k["atol"]), g
    # the same number through the gradient routine itself (unit-radiance light paths gathering dL/dimage = 1 / (8 * 8 * 3))
    gb = o.ptracer_backward(p, np.full((8, 8, 3), 1.0 / 192.0, dtype=np.float32))
    lay, total = o.param_layout()
    off = [l for l in lay if l[0] == mi.abi.PARAM_EMITTER_RADIANCE][0][2]
    assert np.allclose(gb[off:off + 3], k["gradient"], atol=k["atol"]) and (np.delete(gb, [off, off + 1, off + 2]) == 0).all()   # (no light path reaches a surface here)
    # the pass structure matters: one pass of 64 gives another (equally valid) estimate, off by more than the tolerance
    one, _ = o.ptracer_render(o.params(mi.make_integrator(dict(sc.integrator_props, samples_per_pass=64)), k["seed"], k["spp"]))
    assert abs(one.sum() / one.size - k["gradient"]) > 10 * k["atol"] and abs(one.mean() - 1.0) < 0.05
    # test01 / test03 of the same file: the emitter is seen by >= 20 % of the pixels; hidden emitters -> black
    v, _ = o.ptracer_render(o.params(integ, 0, 2))
    assert (v > 0).mean() >= 0.2
    integ.hide_emitters = True
    h, _ = o.ptracer_render(o.params(integ, 0, 4))
    assert (h == 0).all()
    # test05: "we trace the exact same rays regardless of crops, even the noise pattern should be identical"
    integ.hide_emitters = False
    full, _ = o.ptracer_render(o.params(integ, 0, 4))
    d = ptracer_test_scene(mi); d["sensor"]["film"].update(crop_offset_x=2, crop_offset_y=3, crop_width=6, crop_height=4)
    sc2 = mi.load_dict(d); o2 = orc.OracleScene(sc2, mi.abi)
    crop, _ = o2.ptracer_render(o2.params(integ, 0, 4))
    assert crop.shape == (4, 6, 3) and np.allclose(full[3:7, 2:8], crop, atol=1e-4)


def test_particle_tracer_radiance_gradient_is_exact(mi, orc):
    """The film is linear in every emitter's radiance (with the roulette probability detached), so for ANY image adjoint the gradient
    equals  sum(image_e * grad_in) / radiance_e  where image_e is the film lit by emitter e alone -- with the same seeds, to rounding."""
    T = mi.ScalarTransform4f
    d = small_cbox(mi, 16); d["sensor"]["film"]["rfilter"] = {"type": "gaussian"}
    d["lamp2"] = {"type": "sphere", "to_world": T.translate([0.4, -0.2, 0.3]).scale(0.15), "bsdf": {"type": "ref", "id": "white"},
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [2.0, 0.0, 5.0]}}}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    integ = mi.ParticleTracerIntegrator({"max_depth": 5, "rr_depth": 3})
    gin = np.random.default_rng(1).normal(size=(16, 16, 3)).astype(np.float32)
    p = o.params(integ, 5, 32)
    g = o.ptracer_backward(p, gin)
    lay, _ = o.param_layout()
    rads = [np.array(e.radiance, dtype=np.float32) for e in sc.emitters]
    assert len(rads) == 2
    for e, (kind, index, off, size) in enumerate([l for l in lay if l[0] == mi.abi.PARAM_EMITTER_RADIANCE]):
        for other in range(2):
            o.set_param(mi.abi.PARAM_EMITTER_RADIANCE, other, rads[other] if other == e else np.full(3, 1e-30, dtype=np.float32))
        img, _ = o.ptracer_render(p)                     # (1e-30, not 0: an all-black emitter would stop its lanes from tracing)
        expect = (img.astype(np.float64) * gin).sum(axis=(0, 1))
        nz = rads[e] != 0
        assert np.allclose(g[off:off + 3][nz], (expect / np.where(nz, rads[e], 1))[nz], rtol=2e-3, atol=1e-4), (e, g[off:off + 3], expect)
        assert np.isfinite(g[off:off + 3]).all() and (g[off:off + 3][~nz] != 0).all()   # a zero channel still has a derivative


def test_render_forward_is_the_transpose_of_render_backward(mi, orc):
    """RBIntegrator.render_forward (common.py:498-629; prb.py forward mode) restated in the oracle: for any parameter tangent v and any
    image adjoint g,  <g, J v> == <J^T g, v>  with the same seeds (both walks evaluate the same per-vertex derivatives), and the
    gradient image of a single albedo matches a central finite difference of the primal image (the reference's own check,
    test_ad_integrators.py:742-786) -- exactly so while the roulette is off, because the estimator is then polynomial in the albedo."""
    rng = np.random.default_rng(3)
    d = small_cbox(mi, 16)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "raw": True, "data": rng.uniform(0.2, 0.8, (8, 8, 3)).astype(np.float32)}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    integ = mi.PRBIntegrator({"max_depth": 5, "rr_depth": 3})
    lay, total = o.param_layout()
    v = rng.normal(size=total).astype(np.float32)
    gin = rng.normal(size=(16, 16, 3)).astype(np.float32)
    p = o.params(integ, 4, 64)
    jv = o.render_forward(p, v)
    jtg = o.render_backward(p, gin)
    a, b = float((jv.astype(np.float64) * gin).sum()), float((jtg.astype(np.float64) * v).sum())
    assert abs(a - b) < 1e-4 * max(abs(a), abs(b)), (a, b)
    assert (o.render_forward(p, np.zeros(total, dtype=np.float32)) == 0).all()          # no tangent, zero gradient image
    # finite differences of one wall colour, roulette off
    integ = mi.PRBIntegrator({"max_depth": 4, "rr_depth": 100})
    p = o.params(integ, 9, 128)
    kind, index, off, size = [l for l in lay if l[0] == mi.abi.PARAM_BSDF_REFLECTANCE][0]
    t = np.zeros(total, dtype=np.float32); t[off:off + 3] = (1.0, 0.5, -0.25)
    gimg = o.render_forward(p, t)
    base = np.array(sc.bsdfs[index].reflectance, dtype=np.float32); eps = 1e-2
    o.set_param(kind, index, base + eps * t[off:off + 3]); hi, _ = o.render(p)
    o.set_param(kind, index, base - eps * t[off:off + 3]); lo, _ = o.render(p)
    o.set_param(kind, index, base)
    fd = (hi.astype(np.float64) - lo) / (2 * eps)
    assert np.abs(gimg).max() > 0 and np.allclose(gimg, fd, rtol=5e-3, atol=5e-4 * np.abs(fd).max())


def test_film_sample_border(mi, orc):
    """Film::sample_border (film.cpp:32; integrator.cpp:133-134, :307-308; common.py:255-278): the lane grid grows by the filter's
    border_size = ceil(radius - 0.5 - 2 RayEps) (rfilter.cpp:23) on every side, so that edge pixels collect their full footprint."""
    for rf, b in (({"type": "gaussian"}, 2), ({"type": "gaussian", "stddev": 0.3}, 1), ({"type": "tent"}, 1), ({"type": "box"}, 0)):
        sc = mi.load_dict(small_cbox(mi, 12, rfilter=rf, sample_border=True))
        assert sc.sensor.film.border_size == b
        integ = mi.PathIntegrator({"max_depth": 3})
        assert integ.lane_grid(sc.sensor.film) == (12 + 2 * b, 12 + 2 * b)
        o = orc.OracleScene(sc, mi.abi)
        img, lanes = o.render(o.params(integ, 1, 8))
        assert lanes == (12 + 2 * b) ** 2 * 8 and np.isfinite(img).all()
    on = mi.load_dict(small_cbox(mi, 12, sample_border=True)); off = mi.load_dict(small_cbox(mi, 12))
    integ = mi.PRBIntegrator({"max_depth": 3})
    w_on = orc.OracleScene(on, mi.abi); w_off = orc.OracleScene(off, mi.abi)
    W_on = w_on.splat_weights(w_on.params(integ, 0, 256)); W_off = w_off.splat_weights(w_off.params(integ, 0, 256))
    assert abs(W_on[0, 0] / W_on[6, 6] - 1) < 0.05 and W_off[0, 0] / W_off[6, 6] < 0.8          # corner: full footprint vs the part of it that lies inside the window
    assert np.allclose(W_on[4:8, 4:8], W_off[4:8, 4:8], rtol=0.1)


def test_particle_tracer_albedo_gradients_match_finite_differences(mi, orc):
    """ptracer render_backward for diffuse albedos and albedo textures (path replay of the light paths): with the roulette off the film
    is a polynomial in every albedo, so the gradient equals a central finite difference with the same seeds."""
    rng = np.random.default_rng(3)
    d = small_cbox(mi, 16)
    tex0 = rng.uniform(0.2, 0.8, (4, 4, 3)).astype(np.float32)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "raw": True, "data": tex0}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    sc = mi.load_dict(d); o = orc.OracleScene(sc, mi.abi)
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 100})
    lay, total = o.param_layout()
    gin = rng.normal(size=(16, 16, 3)).astype(np.float32)
    p = o.params(integ, 4, 64)
    g = o.ptracer_backward(p, gin)

    def loss():
        img, _ = o.ptracer_render(p)
        return float((img.astype(np.float64) * gin).sum())
    eps = 1e-2
    kind, index, off, size = [l for l in lay if l[0] == mi.abi.PARAM_BSDF_REFLECTANCE][0]
    base = np.array(sc.bsdfs[index].reflectance, dtype=np.float32)
    for c in range(3):
        e = np.zeros(3, np.float32); e[c] = eps
        o.set_param(kind, index, base + e); hi = loss(); o.set_param(kind, index, base - e); lo = loss(); o.set_param(kind, index, base)
        assert abs(g[off + c] - (hi - lo) / (2 * eps)) < 2e-3 * abs(g[off + c]) + 1e-4
    kind, index, off, size = [l for l in lay if l[0] == mi.abi.PARAM_TEXTURE_DATA][0]
    for t in (0, 17, 40):
        e = np.zeros(48, np.float32); e[t] = eps
        o.set_param(kind, index, tex0.ravel() + e); hi = loss(); o.set_param(kind, index, tex0.ravel() - e); lo = loss(); o.set_param(kind, index, tex0.ravel())
        assert abs(g[off + t] - (hi - lo) / (2 * eps)) < 5e-3 * abs(g[off + t]) + 1e-4
    p.flags |= mi.abi.FLAG_EMITTER_GRADS_ONLY                          # radiances only: the other slots stay untouched
    g1 = o.ptracer_backward(p, gin)
    eoff = [l for l in lay if l[0] == mi.abi.PARAM_EMITTER_RADIANCE][0][2]
    assert np.allclose(g1[eoff:eoff + 3], g[eoff:eoff + 3]) and (np.delete(g1, [eoff, eoff + 1, eoff + 2]) == 0).all()


def test_path_samples_per_pass(mi, orc):
    """samples_per_pass for the path integrator (SamplingIntegrator::render, integrator.cpp:283, :319-329): one stream per pixel and
    pass-sample, `passes` samples from it.  The first pass of a multi-pass render is the single-pass render of the same streams, and the
    whole is an unbiased image; prb (one pass in prb.py) refuses."""
    sc = mi.load_dict(small_cbox(mi, 16, rfilter={"type": "box"})); o = orc.OracleScene(sc, mi.abi)
    integ = mi.PathIntegrator({"max_depth": 5, "rr_depth": 3, "samples_per_pass": 4})
    p = o.params(integ, 2, 16); assert p.passes == 4
    multi, lanes = o.render(p, develop=False)
    assert lanes == 16 * 16 * 16
    one, _ = o.render(o.params(mi.PathIntegrator({"max_depth": 5, "rr_depth": 3}), 2, 4), develop=False)      # the same 4 streams per pixel, first sample only
    assert (multi[..., 3] == 16).all() and (one[..., 3] == 4).all()
    # pass 1 of the multi-pass film = the single-pass film; passes 2-4 continue the streams with fresh numbers
    rest = multi - one
    assert (rest[..., 3] == 12).all() and np.isfinite(rest).all() and rest[..., :3].min() >= -1e-4
    ref, _ = o.render(o.params(mi.PathIntegrator({"max_depth": 5, "rr_depth": 3}), 9, 4096))
    big, _ = o.render(o.params(mi.PathIntegrator({"max_depth": 5, "rr_depth": 3, "samples_per_pass": 256}), 2, 1024), develop=False)
    first, _ = o.render(o.params(mi.PathIntegrator({"max_depth": 5, "rr_depth": 3}), 2, 256), develop=False)
    later = (big - first)[..., :3] / 768                                # passes 2-4 alone: the continued streams are as good as fresh ones
    assert abs(later.mean() / ref.mean() - 1) < 0.025 and abs((big[..., :3] / 1024).mean() / ref.mean() - 1) < 0.02
