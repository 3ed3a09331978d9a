"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle on the same seeded inputs.

Pattern follows the reference's check_vectorization (src/python/python/test/util.py:88-175): evaluate lane by
lane on the CPU, once vectorised on the device, compare.  Tolerances:
  * per-lane radiance: rel 1e-4 / abs 1e-5, with at most 1e-3 of the lanes allowed to diverge (a ray grazing an edge
    can take a different branch after a 1-ulp rcp/sqrt/sincos difference between gfx950 and x86);
  * images / gradients: 1e-3 relative to the image maximum (the BASELINE tolerance for radiance and gradients).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import small_cbox

pytestmark = pytest.mark.gpu


def grad_close(a, b, tol=1e-3, what=""):
    """The north star's gradient bar: max |a - b| <= tol * max |b| (1e-3 unless a test states why not).  The measured ratio of every
    call is appended to gpurun_out/grad_parity.txt so that the committed profiles/r03_grad_parity.txt shows the margin of each test."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max()) / scale
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "grad_parity.txt"), "a") as f:
            f.write("%-90s max|a-b|/max|b| = %.3e  (bar %.0e, n = %d)\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::")[-1] + " " + what, err, tol, a.size))
    except OSError:
        pass
    assert np.isfinite(a).all() and err <= tol, (what, err, tol)
    return True

FMAX = np.finfo(np.float32).max


def _rays_for(scene, o, n=4096, seed=1):
    """Mixed bag of rays: primary rays through the film plus random rays from inside the scene."""
    rng = np.random.default_rng(seed)
    f = scene.sensor.film
    sp = rng.random((n // 2, 2), dtype=np.float32) * np.array([f.crop_width, f.crop_height], dtype=np.float32)
    prim = o.primary_rays(sp)
    rnd = np.zeros((n - n // 2, 8), dtype=np.float32)
    rnd[:, 0:3] = rng.uniform(-0.95, 0.95, (rnd.shape[0], 3))
    d = rng.normal(size=(rnd.shape[0], 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rnd[:, 3:6] = d; rnd[:, 6] = FMAX
    short = rng.random(rnd.shape[0]) < 0.3
    rnd[short, 6] = rng.uniform(0.05, 1.5, short.sum())
    return np.concatenate([prim, rnd]).astype(np.float32)


def _gpu_intersect(mi, torch, scene, rays):
    h = scene.handle()
    r = torch.tensor(rays, device="cuda"); out = torch.empty((rays.shape[0], 8), dtype=torch.float32, device="cuda")
    mi.abi.check(mi.abi.lib().mi355_debug_ray_intersect(h, C.c_void_p(r.data_ptr()), rays.shape[0], C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_ray_intersect_cornell(mi, orc, torch_cuda):
    scene = mi.load_dict(small_cbox(mi, 64))
    o = orc.OracleScene(scene, mi.abi)
    rays = _rays_for(scene, o, 8192)
    ref = orc.unpack_hits(o.ray_intersect(rays)); got = orc.unpack_hits(_gpu_intersect(mi, torch_cuda, scene, rays))
    # the engine tests coplanar triangle pairs as one parallelogram (DPrim), the oracle runs Moeller-Trumbore per
    # triangle: a ray through the shared diagonal may report the sibling triangle (same point, same normal)
    same_shape = (ref["shape"] == got["shape"]) & (ref["occluded"] == got["occluded"])
    # the small box's bottom face is coplanar with the floor (y = -1): rays started inside the box (never produced by a
    # path) hit both at the same distance, and the winner of that tie is arbitrary
    tie = ~same_shape & (ref["occluded"] == got["occluded"]) & np.isclose(ref["t"], got["t"], rtol=1e-5, atol=1e-6)
    assert (same_shape | tie).mean() > 0.9995, "shape/occlusion mismatch %g" % (1 - (same_shape | tie).mean())
    assert tie.mean() < 5e-3
    same = same_shape & (ref["prim"] == got["prim"])
    assert same.mean() > 0.997, "prim mismatch %g" % (1 - same.mean())
    hit_any = same_shape & np.isfinite(ref["t"])
    assert np.allclose(ref["t"][hit_any], got["t"][hit_any], rtol=2e-5, atol=2e-6)
    hit = same & np.isfinite(ref["t"])
    assert hit.sum() > 4000
    miss = same & ~np.isfinite(ref["t"])
    assert np.isinf(got["t"][miss]).all()


def test_box_and_quad_primitives(mi, orc, torch_cuda):
    """Geometry organisation: Cornell = 6 rectangle records + 2 oriented boxes; a sheared cube is still a parallelepiped;
    a cube with one displaced corner is not (falls back to parallelogram / triangle records) -- all three agree with the
    oracle's per-triangle Moeller-Trumbore, including prim_index and barycentrics (through uv)."""
    T = mi.ScalarTransform4f
    scene = mi.load_dict(small_cbox(mi, 32))
    assert scene.info() == {"flat_prims": 6, "boxes": 2, "bvh_meshes": 0, "bvh_nodes": 0, "triangles": 24, "grad_floats": 12 + 2 * 24 * 3,
                            "spheres": 0, "features": 0}
    shear = mi.Transform4f([[0.5, 0.2, 0.0, 0.1], [0.0, 0.4, 0.1, -0.2], [0.1, 0.0, 0.3, 0.2], [0, 0, 0, 1]])
    v = np.array([[1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, 1, 1], [1, 1, 1]], dtype=np.float32)
    f = np.array([[0, 1, 2], [3, 0, 2], [4, 5, 6], [7, 4, 6], [0, 4, 7], [1, 0, 7], [1, 7, 6], [2, 1, 6], [2, 6, 5], [3, 2, 5], [4, 0, 3], [5, 4, 3]], dtype=np.uint32)
    bent = v.copy(); bent[7] += [0.3, 0.2, 0.1]
    cases = [({"type": "cube", "to_world": shear}, {"boxes": 1, "flat_prims": 1}),
             ({"type": "mesh", "vertex_positions": bent, "faces": f, "to_world": T.scale(0.5)}, {"boxes": 0}),
             ({"type": "mesh", "vertex_positions": v, "faces": f[:10], "to_world": T.scale(0.5)}, {"boxes": 0, "flat_prims": 6})]
    for shape, expect in cases:
        d = {"type": "scene", "obj": shape, "floor": {"type": "rectangle", "to_world": T.translate([0, -1, 0]).rotate([1, 0, 0], -90)},
             "sensor": {"type": "perspective", "fov": 60, "to_world": T.look_at([0.3, 0.4, 3], [0, 0, 0], [0, 1, 0]),
                        "film": {"type": "hdrfilm", "width": 64, "height": 64, "pixel_format": "rgb"}}}
        sc = mi.load_dict(d); info = sc.info()
        for k, val in expect.items():
            assert info[k] == val, (k, info)
        o = orc.OracleScene(sc, mi.abi)
        rays = _rays_for(sc, o, 8192, seed=11)
        ref = orc.unpack_hits(o.ray_intersect(rays)); got = orc.unpack_hits(_gpu_intersect(mi, torch_cuda, sc, rays))
        same_shape = (ref["shape"] == got["shape"]) & (ref["occluded"] == got["occluded"])
        assert same_shape.mean() > 0.999
        hit = same_shape & np.isfinite(ref["t"])
        assert (ref["shape"][hit] == 0).sum() > 300
        assert np.allclose(ref["t"][hit], got["t"][hit], rtol=3e-5, atol=3e-6)
        same = hit & (ref["prim"] == got["prim"])
        assert same.sum() > 0.99 * hit.sum()
        assert np.allclose(ref["u"][same], got["u"][same], atol=5e-5) and np.allclose(ref["v"][same], got["v"][same], atol=5e-5)


def _icosphere(sub=3):
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]])
    for _ in range(sub):
        cache = {}; verts = list(v); nf = []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = (verts[a] + verts[b]) / 2; m /= np.linalg.norm(m); cache[k] = len(verts); verts.append(m)
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v = np.array(verts); f = np.array(nf)
    return v.astype(np.float32), f.astype(np.uint32)


def test_ray_intersect_bvh_mesh(mi, orc, torch_cuda):
    """1280-triangle icosphere (with vertex normals) -> exercises the LDS-stack BVH traversal against the oracle's
    linear scan, including prim_index remapping and shadow rays."""
    v, f = _icosphere(3)
    T = mi.ScalarTransform4f
    d = {"type": "scene",
         "ball": {"type": "mesh", "vertex_positions": v, "vertex_normals": v, "faces": f, "to_world": T.translate([0.1, -0.2, 0]).scale(0.6)},
         "floor": {"type": "rectangle", "to_world": T.translate([0, -1, 0]).rotate([1, 0, 0], -90)},
         "sensor": {"type": "perspective", "fov": 45, "to_world": T.look_at([0, 0, 3], [0, 0, 0], [0, 1, 0]),
                    "film": {"type": "hdrfilm", "width": 64, "height": 64, "pixel_format": "rgb"}}}
    scene = mi.load_dict(d)
    assert scene.shapes[0].primitive_count() == 1280
    o = orc.OracleScene(scene, mi.abi)
    rays = _rays_for(scene, o, 8192, seed=3)
    ref = orc.unpack_hits(o.ray_intersect(rays)); got = orc.unpack_hits(_gpu_intersect(mi, torch_cuda, scene, rays))
    same = (ref["shape"] == got["shape"]) & (ref["prim"] == got["prim"]) & (ref["occluded"] == got["occluded"])
    assert same.mean() > 0.998
    hit = same & np.isfinite(ref["t"])
    assert (ref["shape"][hit] == 0).sum() > 500          # the sphere is actually hit
    assert np.allclose(ref["t"][hit], got["t"][hit], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("integrator_name,max_depth,rr_depth", [("path", 6, 5), ("path", 1, 5), ("path", 2, 5), ("path", -1, 2), ("prb", 6, 5), ("prb", 3, 1)])
def test_lane_radiance_matches_oracle(mi, orc, torch_cuda, cbox32, integrator_name, max_depth, rr_depth):
    integ = mi.make_integrator({"type": integrator_name, "max_depth": max_depth, "rr_depth": rr_depth})
    o = orc.OracleScene(cbox32, mi.abi)
    spp, seed = 8, 11
    n = 32 * 32 * spp
    p = o.params(integ, seed, spp)
    ref = o.lane_radiance(p, 0, n)
    got = integ.lane_radiance(cbox32, seed, spp, 0, n).cpu().numpy()
    assert np.isfinite(got).all()
    close = np.isclose(got[:, :3], ref[:, :3], rtol=1e-4, atol=1e-5).all(axis=1) & (got[:, 3] == ref[:, 3])
    assert close.mean() > 0.999, "diverged lanes: %g" % (1 - close.mean())
    assert abs(got[:, :3].mean() - ref[:, :3].mean()) < 1e-3 * max(ref[:, :3].mean(), 1e-3)
    if max_depth != 1:
        assert ref[:, :3].max() > 0


@pytest.mark.parametrize("spp", [1, 3, 4, 16, 64, 128])
def test_render_matches_oracle(mi, orc, torch_cuda, cbox32, spp):
    """Whole forward path incl. the film: spp = 1, 3 use per-lane atomics, 4..32 the segmented LDS reducer,
    >= 64 one pixel per wave."""
    integ = mi.PathIntegrator({"max_depth": 6})
    o = orc.OracleScene(cbox32, mi.abi)
    img = integ.render(cbox32, seed=5, spp=spp).cpu().numpy()
    ref, _ = o.render(o.params(integ, 5, spp))
    assert np.isfinite(img).all()
    assert np.abs(img - ref).max() < 1e-3 * ref.max()
    raw = integ.render(cbox32, seed=5, spp=spp, develop=False).cpu().numpy()
    raw_ref, _ = o.render(o.params(integ, 5, spp), develop=False)
    assert np.allclose(raw[:, :, 3], raw_ref[:, :, 3], rtol=1e-4, atol=1e-4)          # W channel: pure filter weights


@pytest.mark.parametrize("spp,max_depth,rr_depth,crop", [(128, 6, 5, None), (200, -1, 2, None), (256, 3, 5, (5, 7, 20, 11)), (130, 1, 5, None)])
def test_path_regeneration_matches_static(mi, orc, torch_cuda, spp, max_depth, rr_depth, crop):
    """MI355_FLAG_REGEN: a wave hands the lanes of finished paths to its next samples.  Samples are still pure functions of
    (seed, global lane id), so film, event counters and the oracle's image must agree with the static assignment; ragged
    pools (spp not a multiple of 64, pixel count not a multiple of the pixels per wave) and the sharded launch included."""
    torch = torch_cuda
    film = {} if crop is None else {"crop_offset_x": crop[0], "crop_offset_y": crop[1], "crop_width": crop[2], "crop_height": crop[3]}
    scene = mi.load_dict(small_cbox(mi, 32, **film))
    integ = mi.PathIntegrator({"max_depth": max_depth, "rr_depth": rr_depth})
    integ.counters(scene, reset=True)
    static = integ.render(scene, seed=8, spp=spp, develop=False)
    c_static = integ.counters(scene, reset=True)
    integ.regen = True
    regen = integ.render(scene, seed=8, spp=spp, develop=False)
    c_regen = integ.counters(scene, reset=True)
    assert torch.allclose(regen, static, rtol=2e-4, atol=2e-4)
    for k in ("n_isect", "n_shade", "n_shadow"):
        assert c_regen[k] == c_static[k], k
    o = orc.OracleScene(scene, mi.abi)
    ref, _ = o.render(o.params(integ, 8, spp))
    img = integ.render(scene, seed=8, spp=spp).cpu().numpy()
    assert np.abs(img - ref).max() < 1e-3 * ref.max()
    h = scene.handle(); lib = mi.abi.lib()
    acc = torch.zeros_like(static)
    W, H = scene.sensor.film.crop_width, scene.sensor.film.crop_height
    for r in range(3):                                           # tiles of 2 rows over 3 ranks + an spp slice
        p = integ._params(scene, 8, spp, shard=mi.shard_tiles(W, H, r, 3, rows=2))
        mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.allclose(acc, static, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("spp,max_depth,rr_depth", [(16, 6, 5), (64, -1, 2), (3, 4, 5), (128, 2, 5)])
def test_repack_modes_match_fused(mi, orc, torch_cuda, cbox32, spp, max_depth, rr_depth):
    """Re-packing modes: MODE_FUSED (default: from bounce 2 on, when it empties a wave), MODE_REPACK_ALWAYS (after every bounce; called MODE_WAVEFRONT up to ABI v8),
    MODE_STATIC (never).  Lanes are still pure functions of (seed, lane id), so the films must agree (up to float-atomic
    order) with each other and with the oracle's, and the event counts must be identical."""
    integ = mi.PathIntegrator({"max_depth": max_depth, "rr_depth": rr_depth})
    fused = integ.render(cbox32, seed=8, spp=spp, develop=False)
    integ.mode = mi.abi.MODE_STATIC
    integ.counters(cbox32, reset=True)
    static = integ.render(cbox32, seed=8, spp=spp, develop=False)
    c_static = integ.counters(cbox32, reset=True)
    assert torch_cuda.allclose(static, fused, rtol=1e-4, atol=1e-4)
    integ.mode = mi.abi.MODE_REPACK_ALWAYS
    integ.counters(cbox32, reset=True)
    wf = integ.render(cbox32, seed=8, spp=spp, develop=False)
    c_wf = integ.counters(cbox32, reset=True)
    assert torch_cuda.allclose(wf, fused, rtol=1e-4, atol=1e-4)
    for k in ("n_isect", "n_shade", "n_shadow"):
        assert c_wf[k] == c_static[k], k
    o = orc.OracleScene(cbox32, mi.abi)
    o.counters(reset=True)
    ref, _ = o.render(o.params(integ, 8, spp))
    img = integ.render(cbox32, seed=8, spp=spp).cpu().numpy()
    assert np.abs(img - ref).max() < 1e-3 * ref.max()
    r = o.counters()
    for k in ("n_isect", "n_shade", "n_shadow"):
        assert abs(int(c_wf[k]) - r[k]) <= 2e-3 * r[k] + 2, (k, c_wf[k], r[k])
    films = {}
    for mode in (mi.abi.MODE_STATIC, mi.abi.MODE_FUSED, mi.abi.MODE_WAVEFRONT):      # the PRB primal estimator re-packs the same way
        prb = mi.PRBIntegrator({"max_depth": max_depth, "rr_depth": rr_depth}); prb.mode = mode
        prb.counters(cbox32, reset=True)
        films[mode] = (prb.render(cbox32, seed=1, spp=spp, develop=False), prb.counters(cbox32, reset=True))
    for mode in (mi.abi.MODE_FUSED, mi.abi.MODE_WAVEFRONT):
        assert torch_cuda.allclose(films[mode][0], films[mi.abi.MODE_STATIC][0], rtol=1e-4, atol=1e-4)
        for k in ("n_isect", "n_shade", "n_shadow"):
            assert films[mode][1][k] == films[mi.abi.MODE_STATIC][1][k], k


@pytest.mark.parametrize("rfilter", [{"type": "box"}, {"type": "tent"}, {"type": "gaussian", "stddev": 0.4}])
def test_render_filters_and_crop(mi, orc, torch_cuda, rfilter):
    d = small_cbox(mi, 48, rfilter=rfilter, crop_width=24, crop_height=16, crop_offset_x=10, crop_offset_y=20)
    scene = mi.load_dict(d)
    integ = mi.PathIntegrator({"max_depth": 4})
    o = orc.OracleScene(scene, mi.abi)
    img = integ.render(scene, seed=2, spp=16).cpu().numpy()
    ref, _ = o.render(o.params(integ, 2, 16))
    assert img.shape == (16, 24, 3)
    assert np.abs(img - ref).max() < 1e-3 * ref.max()


def test_counters_match_oracle(mi, orc, torch_cuda, cbox32):
    integ = mi.PathIntegrator({"max_depth": 6})
    o = orc.OracleScene(cbox32, mi.abi)
    integ.counters(cbox32, reset=True); o.counters(reset=True)
    integ.render(cbox32, seed=9, spp=16)
    o.render(o.params(integ, 9, 16))
    c = integ.counters(cbox32); r = o.counters()
    assert c["n_gen"] == 32 * 32 * 16 and c["n_splat"] == c["n_gen"]
    for k in ("n_isect", "n_shade", "n_shadow"):
        assert abs(int(c[k]) - r[k]) <= 2e-3 * r[k], (k, c[k], r[k])


def test_sharded_render_equals_whole(mi, orc, torch_cuda, cbox32):
    """Multi-GPU decomposition on one device: pixel bands / spp slices accumulate to the same film because lanes are
    seeded by GLOBAL lane id (sampler.cpp:124-130)."""
    torch = torch_cuda
    integ = mi.PathIntegrator({"max_depth": 5})
    h = cbox32.handle(); lib = mi.abi.lib()
    whole = integ.render(cbox32, seed=4, spp=16, develop=False)
    for split in ("pixels", "tiles", "spp"):
        film = torch.zeros_like(whole)
        for r in range(4):
            if split == "pixels":
                p = integ._params(cbox32, 4, 16, shard=mi.shard_pixels(32 * 32, r, 4))
            elif split == "tiles":
                p = integ._params(cbox32, 4, 16, shard=mi.shard_tiles(32, 32, r, 4, rows=3))     # ragged last tile
            else:
                p = integ._params(cbox32, 4, 16, spp_shard=(4 * r, 4))
            mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(film.data_ptr()), None))
        torch.cuda.synchronize()
        assert torch.allclose(film, whole, rtol=1e-4, atol=1e-4), split


# ---- scope row a19: sphere / roughconductor / dielectric (BASELINE config 4 scene) ------------------------
@pytest.fixture(scope="module")
def veach64(mi):
    return mi.load_dict(mi.veach_mis_scene(64))


def test_veach_scene_intersect(mi, orc, torch_cuda, veach64):
    assert veach64.info()["spheres"] == 5 and veach64.info()["features"] & 15 == 3      # (bit 4: more records than the all-diffuse kernels stage in LDS -- irrelevant here)
    o = orc.OracleScene(veach64, mi.abi)
    rays = _rays_for(veach64, o, 8192, seed=5)
    rays[4096:, 0:3] *= 3.0                                    # random origins spread over the whole scene
    ref = orc.unpack_hits(o.ray_intersect(rays)); got = orc.unpack_hits(_gpu_intersect(mi, torch_cuda, veach64, rays))
    same = (ref["shape"] == got["shape"]) & (ref["occluded"] == got["occluded"])
    assert same.mean() > 0.999
    hit = same & np.isfinite(ref["t"])
    sphere_ids = [i for i, sh in enumerate(veach64.shapes) if sh.type == mi.abi.SHAPE_SPHERE]
    assert np.isin(ref["shape"][hit], sphere_ids).sum() > 100
    assert np.allclose(ref["t"][hit], got["t"][hit], rtol=5e-5, atol=5e-6)


@pytest.mark.parametrize("integrator_name,max_depth,rr_depth", [("path", 8, 5), ("path", 3, 5), ("prb", 6, 2)])
def test_veach_lane_radiance(mi, orc, torch_cuda, veach64, integrator_name, max_depth, rr_depth):
    integ = mi.make_integrator({"type": integrator_name, "max_depth": max_depth, "rr_depth": rr_depth})
    o = orc.OracleScene(veach64, mi.abi)
    spp, seed = 4, 3
    n = 64 * 64 * spp
    ref = o.lane_radiance(o.params(integ, seed, spp), 0, n)
    got = integ.lane_radiance(veach64, seed, spp, 0, n).cpu().numpy()
    assert np.isfinite(got).all()
    # glossy lobes amplify ulp-level differences (v_exp / v_rcp / v_rsq / sincos of gfx950 vs glibc).  Measured (tests/tools/lane_parity.py,
    # profiles/r02_lane_parity.txt): 99.98 % of the lanes within rtol 2e-4, every lane within 2e-3, means equal to 5 digits; a
    # -DMI_EXACT_MATH build is listed next to it.
    same = got[:, 3] == ref[:, 3]
    tight = np.isclose(got[:, :3], ref[:, :3], rtol=2e-4, atol=1e-5).all(axis=1) & same
    loose = np.isclose(got[:, :3], ref[:, :3], rtol=2e-3, atol=1e-4).all(axis=1) & same
    assert tight.mean() > 0.999 and loose.mean() > 0.9999, "diverged lanes: %g / %g" % (1 - tight.mean(), 1 - loose.mean())
    assert np.allclose(got[:, :3].mean(0), ref[:, :3].mean(0), rtol=1e-3)


def test_veach_repacking_modes_agree(mi, orc, torch_cuda, veach64):
    """FULL kernels (glossy / specular BSDFs, spheres): the re-packed state also carries eta and prev_bsdf_delta."""
    films, counts = {}, {}
    for mode in (mi.abi.MODE_STATIC, mi.abi.MODE_FUSED, mi.abi.MODE_WAVEFRONT):
        pi = mi.PathIntegrator({"max_depth": 7, "rr_depth": 3}); pi.mode = mode
        pi.counters(veach64, reset=True)
        films[mode] = pi.render(veach64, seed=4, spp=32, develop=False)
        counts[mode] = pi.counters(veach64, reset=True)
    for mode in (mi.abi.MODE_FUSED, mi.abi.MODE_WAVEFRONT):
        ref = films[mi.abi.MODE_STATIC]
        err = (films[mode] - ref).abs()
        assert float(err.max()) <= 2e-4 * float(ref.abs().max()) + 1e-4
        for k in ("n_isect", "n_shade", "n_shadow"):
            assert counts[mode][k] == counts[mi.abi.MODE_STATIC][k], k


def test_veach_render_and_backward(mi, orc, torch_cuda, veach64):
    torch = torch_cuda
    integ = mi.PRBIntegrator({"max_depth": 6, "rr_depth": 5})
    o = orc.OracleScene(veach64, mi.abi)
    img = integ.render(veach64, seed=2, spp=16).cpu().numpy()
    ref, _ = o.render(o.params(integ, 2, 16))
    # fireflies from the tiny lamp make max() a poor scale; compare robustly
    err = np.abs(img - ref)
    assert np.percentile(err, 99.5) < 2e-3 * np.percentile(ref, 99.5) + 1e-4
    assert abs(img.mean() - ref.mean()) < 5e-3 * ref.mean()
    params = mi.traverse(veach64)
    gin = np.ones((64, 64, 3), dtype=np.float32)
    g = integ.render_backward(veach64, params, torch.tensor(gin), seed=6, spp=16).cpu().numpy()
    g_ref = o.render_backward(o.params(integ, 6, 16), gin)
    lay, total = mi.scene_param_layout(veach64)
    colour = [e for e in lay if e[0] in (mi.abi.PARAM_BSDF_REFLECTANCE, mi.abi.PARAM_TEXTURE_DATA, mi.abi.PARAM_EMITTER_RADIANCE)]
    assert (lay, total) == o.param_layout() and sum(e[3] for e in colour) == 3 + 5 * 3    # grey albedo + 5 emitters; conductors carry none
    assert [e for e in lay if e[0] == mi.abi.PARAM_BSDF_IOR] == [(mi.abi.PARAM_BSDF_IOR, [i for i, b in enumerate(veach64.bsdfs) if b.type == mi.abi.BSDF_DIELECTRIC][0], total - 2, 2)]
    assert grad_close(g, g_ref, 1e-3)
    # linearity identity for every emitter: d<1, image>/d rad_c = sum(image_c contribution of that emitter) -> sum over emitters = image sum / 1
    tot = sum((g[off:off + 3] * veach64.emitters[i].radiance).sum() for k, i, off, sz in lay if k == mi.abi.PARAM_EMITTER_RADIANCE)
    img6 = integ.render(veach64, seed=6, spp=16).cpu().numpy()
    assert abs(tot - img6.sum()) < 2e-3 * img6.sum()


# ---- scope row (f)-1: realistic-lens primary rays -----------------------------------------------------------
def _asph_singlet(mi):
    """A biconvex singlet whose two surfaces are aspheres (conic constant + polynomial terms): the `singlet` geometry of
    realisticlens.cpp:936-955 with AsphericalLens interfaces (SpheroidLens(R) == asphere with R_mm = -1000 R, :1636)."""
    from mitsuba3_camera_amd import lens
    base = lens.build_singlet(3.9, 0.05, 0.01)
    glass = base[0].right
    return [lens.Interface(lens.ASPHERE, 0.01, base[0].z, lens.AIR, glass, curvature_radius=-50.0, kappa=-0.6, ai=[2e-5, -3e-7, 0, 0, 0, 0, 0]),
            lens.Interface(lens.ASPHERE, 0.01, base[1].z, glass, lens.AIR, curvature_radius=50.0, kappa=0.4, ai=[-1e-5, 2e-7, 0, 0, 0, 0, 0])]


def _lens_cbox(mi, design, res=32, **extra):
    T = mi.ScalarTransform4f
    d = mi.cornell_box()
    if design == "asph_singlet":
        design, extra = "custom", dict(extra, lens_interfaces=_asph_singlet(mi))
    d["sensor"] = {"type": "realisticlens", "lens_design": design, **extra, "object_distance": 3.9, "lens_focal_length": 0.05, "lens_diameter": 0.02,
                   "near_clip": 0.001, "far_clip": 100.0, "to_world": T.look_at([0, 0, 3.9], [0, 0, 0], [0, 1, 0]),
                   "sampler": {"type": "independent", "sample_count": 16},
                   "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}, "pixel_format": "rgb", "diagonal": 30.0}}
    return mi.load_dict(d)


@pytest.mark.parametrize("design", ["singlet", "doublet", "exp1a", "tessar", "fisheye", "asph_singlet", "exp2a"])
def test_realisticlens_matches_oracle(mi, orc, torch_cuda, design):
    torch = torch_cuda
    scene = _lens_cbox(mi, design)
    o = orc.OracleScene(scene, mi.abi)
    integ = mi.PRBIntegrator({"max_depth": 5, "rr_depth": 3})
    spp, seed = 8, 13
    n = 32 * 32 * spp
    ref = o.lane_radiance(o.params(integ, seed, spp), 0, n)
    got = integ.lane_radiance(scene, seed, spp, 0, n).cpu().numpy()
    close = np.isclose(got[:, :3], ref[:, :3], rtol=2e-4, atol=1e-5).all(axis=1) & (got[:, 3] == ref[:, 3])
    assert close.mean() > 0.998, "diverged lanes: %g" % (1 - close.mean())
    # some rays are vignetted / miss; exp2 (rear filter plate 27 mm wide, 1.35 mm from the film, no exit-pupil LUT) transmits
    # ~1e-4 of the uniformly sampled rays, as in the reference's default mode: its lanes agree but are all dark at this size
    assert (ref[:, :3].max() > 0 or design == "exp2a") and (ref[:, 3] == 0).any()
    if design == "exp2a":
        # ... so agreeing on darkness proves little: a million lanes (1024 spp) let a hundred-odd rays through the 22 surfaces and the two
        # aspheres -- those lanes carry light, and they must agree too
        n_big = 32 * 32 * 1024
        ref_big = o.lane_radiance(o.params(integ, seed, 1024), 0, n_big)
        got_big = integ.lane_radiance(scene, seed, 1024, 0, n_big).cpu().numpy()
        lit = ref_big[:, :3].max(axis=1) > 0
        assert lit.sum() >= 20, int(lit.sum())
        both = np.isclose(got_big[:, :3], ref_big[:, :3], rtol=2e-4, atol=1e-5).all(axis=1) & (got_big[:, 3] == ref_big[:, 3])
        assert both.mean() > 0.9999 and both[lit].mean() > 0.97, (float(both.mean()), float(both[lit].mean()), int(lit.sum()))
    img = integ.render(scene, seed=3, spp=16).cpu().numpy()
    img_ref, _ = o.render(o.params(integ, 3, 16))
    err = np.abs(img - img_ref)
    # (round 3: the whole lens walk -- film point to front element, the asphere's conic root, Newton iteration and acceptance |dz| < 10 eps --
    #  runs in IEEE arithmetic on the device whatever the rest of the build uses, so the aspherical singlet is held to the same bar as
    #  every other design; with v_rcp / v_rsq in the walk a handful of its rays flipped that acceptance, profiles/r02_grad_parity.txt)
    assert err.max() <= 1e-3 * img_ref.max(), (design, err.max() / img_ref.max())
    params = mi.traverse(scene)
    gin = np.random.default_rng(2).normal(size=(32, 32, 3)).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=9, spp=16).cpu().numpy()
    g_ref = o.render_backward(o.params(integ, 9, 16), gin)
    assert grad_close(g, g_ref, 1e-3, design)
    films = {}
    for mode in (mi.abi.MODE_STATIC, mi.abi.MODE_FUSED, mi.abi.MODE_WAVEFRONT):      # re-packing carries eta / prev_delta and the lens weight too
        pi = mi.PathIntegrator({"max_depth": 5}); pi.mode = mode
        films[mode] = pi.render(scene, seed=1, spp=16, develop=False)
    assert torch.allclose(films[mi.abi.MODE_FUSED], films[mi.abi.MODE_STATIC], rtol=2e-4, atol=2e-4)
    assert torch.allclose(films[mi.abi.MODE_WAVEFRONT], films[mi.abi.MODE_STATIC], rtol=2e-4, atol=2e-4)


# ---- PRB adjoint --------------------------------------------------------------------------------------
def _textured_cbox(mi, res=32):
    d = small_cbox(mi, res)
    rng = np.random.default_rng(12345)          # same seed as the reference's conftest (src/conftest.py:29-31)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": rng.uniform(0.2, 0.8, (16, 16, 3)).astype(np.float32),
                                                     "raw": True, "filter_type": "bilinear", "wrap_mode": "repeat"}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    d["small-box"]["bsdf"] = {"type": "ref", "id": "tex"}     # mesh uv path
    return d


def test_splat_weights_match_oracle(mi, orc, torch_cuda, cbox32):
    torch = torch_cuda
    integ = mi.PRBIntegrator()
    o = orc.OracleScene(cbox32, mi.abi)
    for spp in (1, 4, 64):
        p = integ._params(cbox32, 3, spp)
        w = torch.zeros((32, 32), device="cuda")
        mi.abi.check(mi.abi.lib().mi355_prb_splat_weights(cbox32.handle(), C.byref(p), C.c_void_p(w.data_ptr()), None))
        torch.cuda.synchronize()
        assert np.allclose(w.cpu().numpy(), o.splat_weights(p), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("textured", [False, True])
def test_render_backward_matches_oracle(mi, orc, torch_cuda, textured):
    torch = torch_cuda
    scene = mi.load_dict(_textured_cbox(mi) if textured else small_cbox(mi, 32))
    integ = mi.PRBIntegrator({"max_depth": 6, "rr_depth": 5})
    o = orc.OracleScene(scene, mi.abi)
    params = mi.traverse(scene)
    rng = np.random.default_rng(7)
    gin = rng.normal(size=(32, 32, 3)).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=21, spp=16).cpu().numpy()
    g_ref = o.render_backward(o.params(integ, 21, 16), gin)
    assert np.isfinite(g).all() and np.abs(g_ref).max() > 0
    lay_gpu, total = mi.scene_param_layout(scene); lay_cpu, total_cpu = o.param_layout()
    assert lay_gpu == lay_cpu and total == total_cpu
    for kind, index, off, size in lay_gpu:
        a, b = g[off:off + size], g_ref[off:off + size]
        scale = max(np.abs(b).max(), 1e-6)
        assert grad_close(a, b, 1e-3, "kind %d index %d" % (kind, index))
    # named access
    key = "red.reflectance.value"
    assert np.allclose(params.grad(key).cpu().numpy(), g_ref[[o for k, i, o, s in lay_cpu if k == 0 and scene.bsdfs[i].id == "red"][0]:][:3], rtol=2e-3, atol=1e-5)


def _all_textured_cbox(mi, res=32):
    d = small_cbox(mi, res)
    rng = np.random.default_rng(3)
    for name in ("white", "red", "green"):
        d[name] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": rng.uniform(0.3, 0.9, (8, 8, 3)).astype(np.float32),
                                                      "raw": True, "filter_type": "bilinear" if name != "red" else "nearest", "wrap_mode": "repeat"}}
    return d


@pytest.mark.parametrize("case", ["constant", "textured", "overflow", "veach"])
def test_one_pass_adjoint_equals_two_pass(mi, orc, torch_cuda, case):
    """The default backward kernel walks every path once and resolves gradient records with L_total afterwards;
    MI355_FLAG_TWO_PASS is prb.py's primal + adjoint replay.  Same gradients (the difference is fp32 rounding of
    L_total - P_k), same event counts per walk; a workgroup whose record queue overflows is redone by the two-pass kernel
    ("overflow": every wall textured, long paths -> 3 granules per vertex, more than the queue holds)."""
    torch = torch_cuda
    if case == "veach":
        scene = mi.load_dict(mi.veach_mis_scene(48)); integ_args = {"max_depth": 5, "rr_depth": 3}; res = 48
    else:
        d = {"constant": small_cbox(mi, 32), "textured": _textured_cbox(mi), "overflow": _all_textured_cbox(mi)}[case]
        scene = mi.load_dict(d); res = 32
        integ_args = {"max_depth": 12, "rr_depth": 12} if case == "overflow" else {"max_depth": 6, "rr_depth": 5}
    gin = np.random.default_rng(11).normal(size=(res, res, 3)).astype(np.float32)
    out = {}
    for two_pass in (False, True):
        integ = mi.PRBIntegrator(integ_args); integ.two_pass = two_pass
        params = mi.traverse(scene)
        integ.counters(scene, reset=True)
        out[two_pass] = integ.render_backward(scene, params, torch.tensor(gin), seed=5, spp=32).cpu().numpy().copy()
        c = integ.counters(scene, reset=True)
        out[(two_pass, "c")] = c
    a, b = out[False], out[True]
    assert np.isfinite(a).all() and np.abs(b).max() > 0
    lay, total = mi.scene_param_layout(scene)
    for kind, index, off, size in lay:
        scale = max(np.abs(b[off:off + size]).max(), 1e-6)
        assert np.abs(a[off:off + size] - b[off:off + size]).max() < 1e-3 * scale, (case, kind, index)
    c1, c2 = out[(False, "c")], out[(True, "c")]
    assert c2["n_redo_groups"] == 0
    if case == "overflow":
        assert c1["n_redo_groups"] > 0
    else:
        assert c1["n_redo_groups"] == 0, c1["n_redo_groups"]
        for k in ("n_isect", "n_shade", "n_shadow"):
            assert 2 * c1[k] == c2[k], k                                  # one walk instead of two


@pytest.mark.parametrize("design", ["singlet", "doublet", "exp2a"])
def test_exit_pupil_sampling_matches_oracle(mi, orc, torch_cuda, design):
    """realisticlens `sample_exit_pupil` (realisticlens.cpp:1685-1836, :2282-2285): the 64 x 2^20-ray exit-pupil LUT the engine
    traces at scene creation equals the oracle's (min / max of the same rays: order-independent; a ray whose transmission
    flips between v_rsq and libm can move a bound by one sample spacing), lanes agree, shutter_open_time > 0 costs one
    sampler draw per sample -- and with the LUT the 22-surface Nikon design (exp2a) finally transmits most of its rays."""
    import ctypes
    scene = _lens_cbox(mi, design, sample_exit_pupil=True, shutter_open=0.0, shutter_close=1.0)
    got = np.empty((64, 4), dtype=np.float32)
    mi.abi.check(mi.abi.lib().mi355_lens_exit_pupil_bounds(scene.handle(), 0, got.ctypes.data_as(ctypes.POINTER(ctypes.c_float))))
    integ = mi.PRBIntegrator({"max_depth": 4, "rr_depth": 3})
    spp, seed, n = 8, 5, 32 * 32 * 8
    lanes = integ.lane_radiance(scene, seed, spp, 0, n).cpu().numpy()
    assert np.isfinite(lanes).all()
    if design == "exp2a":
        assert (lanes[:, :3].max(axis=1) > 0).mean() > 0.05             # lit lanes; without the LUT: none of 8192 (test_realisticlens_matches_oracle)
        assert got[0, 2] - got[0, 0] < 0.1 * 2 * 1.5 * scene.sensor.interfaces[0].element_radius   # (outer rings nothing gets through: whole square)
        return                                                          # (the oracle's LUT for 22 surfaces takes a minute on 8 cores)
    o = orc.OracleScene(scene, mi.abi)
    ref_b = o.exit_pupil_bounds()
    spacing = 2 * 1.5 * scene.sensor.interfaces[0].element_radius / 1024
    assert np.abs(got - ref_b).max() <= spacing and (got == ref_b).mean() > 0.9
    ref = o.lane_radiance(o.params(integ, seed, spp), 0, n)
    close = np.isclose(lanes[:, :3], ref[:, :3], rtol=2e-4, atol=1e-5).all(axis=1) & (lanes[:, 3] == ref[:, 3])
    assert close.mean() > (0.998 if (got == ref_b).all() else 0.97), "diverged lanes: %g" % (1 - close.mean())
    off = _lens_cbox(mi, design, sample_exit_pupil=True)                 # shutter closed: the reference's weight is 0 (:2284)
    assert float(integ.render(off, seed=1, spp=4).abs().max()) == 0.0


def _fixture_like_cbox(mi, res=32):
    """The ingredients of the reference's notebooks/scenes/cboxes/cbox_path.xml: mesh luminaire (cbox_luminaire.obj is a
    quad = 2 triangles carrying the area emitter), mirror sphere (`conductor`, material none), glass sphere (`dielectric`)."""
    T = mi.ScalarTransform4f
    d = small_cbox(mi, res)
    del d["large-box"], d["small-box"]
    quad = np.array([[0.25, 0.99, -0.25], [0.25, 0.99, 0.25], [-0.25, 0.99, 0.25], [-0.25, 0.99, -0.25]], dtype=np.float32)
    d["light"] = {"type": "mesh", "vertex_positions": quad, "faces": np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32),
                  "bsdf": {"type": "ref", "id": "white"}, "emitter": d["light"]["emitter"]}
    d["mirror"] = {"type": "conductor"}; d["glass"] = {"type": "dielectric"}
    d["mirrorsphere"] = {"type": "sphere", "to_world": T.translate([-0.3, -0.5, 0.2]).scale(0.5), "bsdf": {"type": "ref", "id": "mirror"}}
    d["glasssphere"] = {"type": "sphere", "to_world": T.translate([0.5, -0.75, -0.2]).scale(0.25), "bsdf": {"type": "ref", "id": "glass"}}
    return d


@pytest.mark.parametrize("integrator_name", ["path", "prb", "ptracer"])
def test_mesh_emitter_and_smooth_conductor(mi, orc, torch_cuda, integrator_name):
    """Mesh::sample_position (mesh.cpp:749-803, area-proportional face choice with sample reuse) as the NEE / light-tracing
    source, SmoothConductor::sample (conductor.cpp:248-307): lanes (path, prb), film (ptracer) and gradients against the oracle."""
    torch = torch_cuda
    scene = mi.load_dict(_fixture_like_cbox(mi))
    assert scene.info()["features"] & 3 == 3
    o = orc.OracleScene(scene, mi.abi)
    integ = mi.make_integrator({"type": integrator_name, "max_depth": 6, "rr_depth": 3})
    if integrator_name == "ptracer":
        f = integ.render(scene, seed=2, spp=16, develop=False).cpu().numpy()
        ref, _ = o.ptracer_render(o.params(integ, 2, 16), develop=False)
        scale = np.percentile(ref[..., :3], 99)
        assert np.percentile(np.abs(f - ref), 99.5) < 2e-3 * scale and abs(f.sum() / ref.sum() - 1) < 2e-3
        return
    spp, seed, n = 8, 3, 32 * 32 * 8
    ref = o.lane_radiance(o.params(integ, seed, spp), 0, n)
    got = integ.lane_radiance(scene, seed, spp, 0, n).cpu().numpy()
    close = np.isclose(got[:, :3], ref[:, :3], rtol=5e-4, atol=1e-5).all(axis=1) & (got[:, 3] == ref[:, 3])
    assert close.mean() > 0.995, "diverged lanes: %g" % (1 - close.mean())
    img = integ.render(scene, seed=5, spp=16).cpu().numpy()
    img_ref, _ = o.render(o.params(integ, 5, 16))
    err = np.abs(img - img_ref)
    assert np.percentile(err, 99.5) < 1e-3 * img_ref.max() and abs(img.mean() / img_ref.mean() - 1) < 2e-3
    params = mi.traverse(scene)
    gin = np.random.default_rng(4).normal(size=(32, 32, 3)).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=7, spp=16).cpu().numpy()
    g_ref = o.render_backward(o.params(integ, 7, 16), gin)
    assert grad_close(g, g_ref, 1e-3)


@pytest.mark.parametrize("scene_name", ["cbox-textured", "veach"])
def test_render_forward_matches_oracle(mi, orc, torch_cuda, scene_name):
    """Integrator::render_forward (k_render_forward): the gradient image against the oracle's for a random tangent over ALL parameters,
    the transpose identity <g, J v> == <J^T g, v> against the device's own render_backward, the reference's finite-difference check
    (test_ad_integrators.py:742-786) against the device's own primal renders, shards adding up, and zero tangents -> zero image."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    if scene_name == "veach":
        scene = mi.load_dict(mi.veach_mis_scene(32)); props = {"max_depth": 5, "rr_depth": 2}
    else:
        scene = mi.load_dict(_textured_cbox(mi)); props = {"max_depth": 6, "rr_depth": 3}
    integ = mi.PRBIntegrator(props)
    o = orc.OracleScene(scene, mi.abi)
    params = mi.traverse(scene)
    lay, total = o.param_layout()
    v = rng.normal(size=total).astype(np.float32)
    for key in params._entries:
        kind, index, shape = params._entries[key]; off, size = params._slot(key)
        params.set_tangent(key, torch.tensor(v[off:off + size].reshape(shape)))
    gimg = integ.render_forward(scene, params, seed=5, spp=32).cpu().numpy()
    ref = o.render_forward(o.params(integ, 5, 32), v)
    scale = np.abs(ref).max()
    assert scale > 0 and np.percentile(np.abs(gimg - ref), 99.5) < 2e-3 * scale and abs(gimg.sum() / ref.sum() - 1) < 5e-3
    gin = rng.normal(size=gimg.shape).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=5, spp=32).cpu().numpy()
    a, b = float((gimg.astype(np.float64) * gin).sum()), float((g.astype(np.float64) * v).sum())
    assert abs(a - b) < 2e-3 * max(abs(a), abs(b)), (a, b)
    # shards of the sample plane: the accumulators add up
    f = scene.sensor.film; acc = torch.zeros((f.crop_height, f.crop_width, 4), device="cuda")
    for r in range(2):
        p = integ._params(scene, 5, 32, shard=mi.shard_tiles(f.crop_width, f.crop_height, r, 2, rows=8)); p.integrator = mi.abi.INTEGRATOR_PRB
        mi.abi.check(mi.abi.lib().mi355_render_forward(scene.handle(), C.byref(p), C.c_void_p(params.tangent_buffer(acc.device).data_ptr()), C.c_void_p(acc.data_ptr()), None))
    img2 = torch.empty((f.crop_height, f.crop_width, 3), device="cuda")
    mi.abi.check(mi.abi.lib().mi355_film_develop(scene.handle(), 0, C.c_void_p(acc.data_ptr()), C.c_void_p(img2.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.allclose(img2.cpu().numpy(), gimg, rtol=1e-3, atol=1e-4 * scale)
    params.clear_tangents()
    assert (integ.render_forward(scene, params, seed=5, spp=8).cpu().numpy() == 0).all()
    if scene_name == "cbox-textured":       # finite differences of one wall colour with the roulette off: the estimator is polynomial in it
        integ = mi.PRBIntegrator({"max_depth": 4, "rr_depth": 100})
        key = [k for k in params._entries if params._entries[k][0] == mi.abi.PARAM_BSDF_REFLECTANCE][0]
        t = torch.tensor([1.0, 0.5, -0.25], device="cuda")
        params.set_tangent(key, t)
        gimg = integ.render_forward(scene, params, seed=9, spp=256).cpu().numpy()
        base = params[key].clone(); eps = 1e-2
        params[key] = base + eps * t; params.update(); hi = integ.render(scene, seed=9, spp=256).cpu().numpy().astype(np.float64)
        params[key] = base - eps * t; params.update(); lo = integ.render(scene, seed=9, spp=256).cpu().numpy().astype(np.float64)
        params[key] = base; params.update()
        fd = (hi - lo) / (2 * eps)
        assert np.abs(fd).max() > 0 and np.allclose(gimg, fd, rtol=1e-2, atol=2e-3 * np.abs(fd).max())
        with pytest.raises(mi.PluginError):
            mi.ParticleTracerIntegrator({}).render_forward(scene, params)


@pytest.mark.parametrize("scene_name", ["cbox", "veach"])
def test_path_samples_per_pass_matches_oracle(mi, orc, torch_cuda, scene_name):
    """samples_per_pass for the path integrator on the device (the static kernel walks the passes of a stream and burns the draws of the
    iterations whose body it skips): film against the oracle's, first pass == single-pass render, shards add up, prb refuses."""
    torch = torch_cuda
    scene = mi.load_dict(small_cbox(mi, 32) if scene_name == "cbox" else mi.veach_mis_scene(32))
    o = orc.OracleScene(scene, mi.abi)
    props = {"max_depth": 6, "rr_depth": 3, "samples_per_pass": 16}
    integ = mi.PathIntegrator(props)
    film = integ.render(scene, seed=4, spp=64, develop=False)
    ref, lanes = o.render(o.params(integ, 4, 64), develop=False)
    assert lanes == 32 * 32 * 64
    f = film.cpu().numpy()
    scale = np.percentile(ref[..., :3], 99)
    if scene_name == "veach":
        assert np.percentile(np.abs(f - ref), 99) < 5e-3 * scale * 64 and abs(f[..., :3].sum() / ref[..., :3].sum() - 1) < 5e-3
    else:
        assert np.abs(f - ref).max() < 1e-3 * np.abs(ref).max()
    one = mi.PathIntegrator(dict(props, samples_per_pass=-1)).render(scene, seed=4, spp=16, develop=False).cpu().numpy()
    rest = f - one
    # the first pass IS the single-pass render of the same streams: what remains are three more passes' worth of (non-negative) splats
    assert abs(rest[..., 3].sum() / f[..., 3].sum() - 0.75) < 0.01 and rest[..., 3].min() > 0 and rest[..., :3].min() > -1e-3 * scale
    acc = torch.zeros_like(film)
    for r in range(2):
        p = integ._params(scene, 4, 64, shard=mi.shard_tiles(32, 32, r, 2, rows=8))
        mi.abi.check(mi.abi.lib().mi355_render_accumulate(scene.handle(), C.byref(p), C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.allclose(acc, film, rtol=1e-3, atol=1e-3 * float(scale))
    with pytest.raises(mi.Mi355Error):
        mi.PRBIntegrator(props).render(scene, seed=4, spp=64)


def test_film_sample_border_matches_oracle(mi, orc, torch_cuda):
    """Film::sample_border on the device: lanes over the grown grid (all forward kernels and the adjoint) against the oracle."""
    torch = torch_cuda
    scene = mi.load_dict(small_cbox(mi, 24, sample_border=True, crop_offset_x=3, crop_offset_y=2, crop_width=16, crop_height=20))
    o = orc.OracleScene(scene, mi.abi)
    params = mi.traverse(scene)
    gin = np.random.default_rng(8).normal(size=(20, 16, 3)).astype(np.float32)
    for name in ("path", "prb"):
        integ = mi.make_integrator({"type": name, "max_depth": 5, "rr_depth": 3})
        n = 20 * 24 * 8
        ref = o.lane_radiance(o.params(integ, 3, 8), 0, n); got = integ.lane_radiance(scene, 3, 8, 0, n).cpu().numpy()
        assert (np.isclose(got[:, :3], ref[:, :3], rtol=5e-4, atol=1e-5).all(axis=1) & (got[:, 3] == ref[:, 3])).mean() > 0.998
        for spp in (4, 64, 256):
            for mode in (mi.abi.MODE_FUSED, mi.abi.MODE_STATIC, mi.abi.MODE_WAVEFRONT):
                integ.mode = mode
                img = integ.render(scene, seed=5, spp=spp).cpu().numpy()
                img_ref, lanes = o.render(o.params(integ, 5, spp))
                assert lanes == 20 * 24 * spp and np.abs(img - img_ref).max() < 1e-3 * img_ref.max()
        integ.mode = mi.abi.MODE_FUSED
        if name == "path":
            integ.regen = True
            assert np.abs(integ.render(scene, seed=5, spp=256).cpu().numpy() - img_ref).max() < 1e-3 * img_ref.max()
            integ.regen = False
        params.zero_grad()
        g = integ.render_backward(scene, params, torch.tensor(gin), seed=7, spp=16).cpu().numpy()
        g_ref = o.render_backward(o.params(integ, 7, 16), gin)
        assert grad_close(g, g_ref, 1e-3)


@pytest.mark.parametrize("config", ["area_light_radiance", "directly_visible_area_light_radiance"])
@pytest.mark.parametrize("integrator_name", ["path", "prb"])
def test_reference_ad_protocol(mi, torch_cuda, config, integrator_name):
    """test02_rendering_forward / test03_rendering_backward of src/integrators/tests/test_ad_integrators.py:739-828 on the two of
    its configurations whose plugins are on this path (AreaLightRadianceConfig :164-189, DirectlyVisibleAreaLightRadianceConfig
    :192-205; ConfigBase :44-75: 128^2, Gaussian filter, sample_border, max_depth 2, spp 1024, theta added to every channel of
    `light.emitter.radiance.value`).  The reference image of that protocol is a central finite difference of the `path` integrator at
    12000 spp with eps = 1e-3 (:942-962), generated here by the same recipe since the stored EXRs are not in the tree; thresholds
    are the reference's: relative error (denominator clamped at 0.2) mean < 0.05 and max < 0.5 forward, < 0.05 backward."""
    torch = torch_cuda
    T = mi.ScalarTransform4f
    if config == "area_light_radiance":
        d = {"type": "scene",
             "plane": {"type": "rectangle", "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}},
             "sphere": {"type": "sphere", "to_world": T.scale(0.25)},
             "light": {"type": "rectangle", "to_world": T.translate([1.25, 0.0, 1.0]) @ T.rotate([0, 1, 0], -90),
                       "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [3.0, 3.0, 3.0]}}}}
    else:
        d = {"type": "scene", "light": {"type": "rectangle", "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [1.0, 1.0, 1.0]}}}}
    d["sensor"] = {"type": "perspective", "to_world": T.look_at(origin=[0, 0, 4], target=[0, 0, 0], up=[0, 1, 0]),
                   "film": {"type": "hdrfilm", "rfilter": {"type": "gaussian", "stddev": 0.5}, "width": 128, "height": 128,
                            "sample_border": True, "pixel_format": "rgb", "component_format": "float32"}}
    scene = mi.load_dict(d)
    params = mi.traverse(scene)
    key = "light.emitter.radiance.value"
    params.keep([key])
    initial = params[key].clone()
    # the reference image: finite differences of `path`, 12000 spp, eps = 1e-3 (test_ad_integrators.py:942-962)
    ref_integrator = mi.PathIntegrator({"max_depth": 2}); eps = 1e-3
    params[key] = initial - eps; params.update(); lo = ref_integrator.render(scene, seed=0, spp=12000).double()
    params[key] = initial + eps; params.update(); hi = ref_integrator.render(scene, seed=0, spp=12000).double()
    params[key] = initial; params.update()
    image_fwd_ref = ((hi - lo) / (2 * eps)).float()
    assert float(image_fwd_ref.abs().max()) > 0
    integrator = mi.make_integrator({"type": integrator_name, "max_depth": 2})
    # test02: forward
    params.set_tangent(key, torch.ones(3, device="cuda"))                   # d radiance / d theta
    image_fwd = integrator.render_forward(scene, params, seed=0, spp=1024)
    error = (image_fwd - image_fwd_ref).abs() / image_fwd_ref.abs().clamp(min=2e-1)
    assert float(error.mean()) < 0.05 and float(error.max()) < 0.5, (float(error.mean()), float(error.max()))
    # test03: backward with a uniform adjoint
    grad_in = 0.001
    params.zero_grad()
    integrator.render_backward(scene, params, torch.full((128, 128, 3), grad_in, device="cuda"), seed=0, spp=256)
    grad = float(params.grad(key).sum()) / image_fwd_ref.numel()
    grad_ref = float(image_fwd_ref.mean()) * grad_in
    assert abs(grad - grad_ref) / max(abs(grad_ref), 1e-3) < 0.05, (grad, grad_ref)


# ---- scope row (f)-3, forward part: the particle tracer ----------------------------------------------------------
@pytest.mark.parametrize("case", ["cbox-box", "cbox-gaussian-crop", "veach", "hide", "lens"])
def test_particle_tracer_matches_oracle(mi, orc, torch_cuda, case):
    """k_ptracer against the oracle's restatement of ptracer.cpp / AdjointIntegrator::render: same seeds, same lanes (the film
    is a sum over lanes, so it agrees up to the order of the float atomics), same event counts; shards over the FILM grid add up;
    the image agrees statistically with the path tracer's (the reference's own methodology for this integrator)."""
    torch = torch_cuda
    if case == "veach":
        scene = mi.load_dict(mi.veach_mis_scene(32)); props = {"max_depth": 5, "rr_depth": 2}; spp = 16
    elif case == "lens":       # the combination of the reference's notebooks/scenes/cboxes/cbox_real.xml: ptracer + realisticlens
        scene = _lens_cbox(mi, "doublet", shutter_open=0.0, shutter_close=30.0); props = {"max_depth": 5, "rr_depth": 3}; spp = 16
        scene.sensor.film  # (RealisticLensCamera::sample_direction, realisticlens.cpp:2296-2345: its weight carries "TODO: correct??"
                           #  in the reference and is restated as written)
    else:
        film = {"rfilter": {"type": "box"}} if case != "cbox-gaussian-crop" else {"crop_offset_x": 5, "crop_offset_y": 3, "crop_width": 20, "crop_height": 24}
        scene = mi.load_dict(small_cbox(mi, 32, **film)); props = {"max_depth": 6, "rr_depth": 3}; spp = 16
    integ = mi.ParticleTracerIntegrator(props); integ.hide_emitters = case == "hide"
    o = orc.OracleScene(scene, mi.abi)
    integ.counters(scene, reset=True); o.counters(reset=True)
    film = integ.render(scene, seed=4, spp=spp, develop=False)
    c = integ.counters(scene, reset=True)
    ref, lanes = o.ptracer_render(o.params(integ, 4, spp), develop=False)
    r = o.counters()
    f = film.cpu().numpy()
    assert np.isfinite(f).all() and (f[..., 3] == 0).all() and ref[..., :3].max() > 0
    scale = np.percentile(ref[..., :3], 99)
    err = np.abs(f - ref)
    if case in ("veach", "lens"):   # glossy lobes / lens refractions amplify ulp differences (see test_veach_lane_radiance): robust comparison
        assert np.percentile(err, 99) < 5e-3 * scale and abs(f.sum() - ref.sum()) < 5e-3 * ref.sum()
        for k in ("n_isect", "n_shade", "n_shadow"):
            assert abs(int(c[k]) - r[k]) <= 2e-3 * r[k] + 2, (k, c[k], r[k])
    else:
        # (a connection whose film position sits within an ulp of a pixel border may land on the other side: 2 pixels x 3 channels)
        assert (err > 1e-3 * scale + 1e-5).sum() <= 6 and abs(f.sum() - ref.sum()) < 1e-4 * ref.sum(), err.max() / scale
        for k in ("n_isect", "n_shade", "n_shadow"):
            assert int(c[k]) == r[k], (k, c[k], r[k])
    img = integ.render(scene, seed=4, spp=spp).cpu().numpy()
    assert np.allclose(img, f[..., :3], rtol=1e-4, atol=1e-6)                # W = 0: develop divides by 1 (hdrfilm.cpp:396-400)
    # shards of the film grid (bands, tiles, spp slices) add up to the whole accumulator
    h = scene.handle(); lib = mi.abi.lib()
    fw, fh = scene.sensor.film.width, scene.sensor.film.height
    for shards in ([mi.shard_pixels(fw * fh, k, 3) for k in range(3)], [mi.shard_tiles(fw, fh, k, 2, rows=4) for k in range(2)]):
        acc = torch.zeros_like(film)
        for sh in shards:
            p = integ._params(scene, 4, spp, shard=sh)
            mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None))
        torch.cuda.synchronize()
        assert torch.allclose(acc, film, rtol=1e-3, atol=1e-3 * float(scale) + 1e-5)
    if case == "cbox-box":     # light tracing and path tracing estimate the same image
        path = mi.PathIntegrator(props).render(scene, seed=1, spp=1024).cpu().numpy()
        pt = integ.render(scene, seed=1, spp=1024).cpu().numpy()
        assert abs(pt.mean() / path.mean() - 1) < 0.015
        g0 = integ.render_backward(scene, None, torch.zeros((32, 32, 3)), seed=1, spp=4)      # no SceneParameters: a fresh gradient buffer
        assert g0.numel() == o.param_layout()[1] and float(g0.abs().max()) == 0.0


@pytest.mark.parametrize("rfilter", [{"type": "gaussian"}, {"type": "tent"}, {"type": "gaussian", "stddev": 0.3}])
def test_particle_tracer_deferred_splat(mi, orc, torch_cuda, rfilter):
    """The tile-sorted, LDS-accumulated splat (film_put_deferred -> k_splat_scan / k_splat_scatter / k_splat_tiles) against the direct
    atomics and the oracle: same film up to summation order -- on a film that is not a multiple of the tile size, with a crop window,
    with the queue capped far below the number of connections (overflow records take the direct path), and over two shards."""
    torch = torch_cuda
    d = small_cbox(mi, 48, rfilter=rfilter, width=52, height=37, crop_offset_x=3, crop_offset_y=2, crop_width=45, crop_height=33)
    scene = mi.load_dict(d); o = orc.OracleScene(scene, mi.abi)
    integ = mi.ParticleTracerIntegrator({"max_depth": 5, "rr_depth": 3})
    ref, _ = o.ptracer_render(o.params(integ, 6, 32), develop=False)
    scale = np.percentile(ref[..., :3], 99)
    deferred = integ.render(scene, seed=6, spp=32, develop=False).cpu().numpy()
    integ.direct_splat = True
    direct = integ.render(scene, seed=6, spp=32, develop=False).cpu().numpy()
    integ.direct_splat = False
    assert np.abs(direct - ref).max() < 2e-3 * scale and np.abs(deferred - ref).max() < 2e-3 * scale
    assert np.allclose(deferred, direct, rtol=1e-4, atol=1e-5 * scale) and (deferred[..., 3] == 0).all()
    mi.abi.check(mi.abi.lib().mi355_debug_set_splat_capacity(scene.handle(), 5000))
    capped = integ.render(scene, seed=6, spp=32, develop=False).cpu().numpy()
    mi.abi.check(mi.abi.lib().mi355_debug_set_splat_capacity(scene.handle(), 0))
    assert np.allclose(capped, direct, rtol=1e-4, atol=1e-5 * scale)
    acc = torch.zeros((33, 45, 4), device="cuda")
    for r in range(2):
        p = integ._params(scene, 6, 32, shard=mi.shard_tiles(52, 37, r, 2, rows=8))
        mi.abi.check(mi.abi.lib().mi355_render_accumulate(scene.handle(), C.byref(p), C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.allclose(acc.cpu().numpy(), direct, rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("res", [832, 1024, 1408])
def test_particle_tracer_deferred_splat_on_large_films(mi, torch_cuda, res):
    """The scatter pass of the deferred splat keeps three words per film tile plus a 32 KB permutation in dynamic LDS: 2704 tiles (832^2) fit the
    64 KB a launch gets by default, 4096 (1024^2) and 7744 (1408^2, the largest below SPLAT_MAX_TILES) need the raised limit
    (hipFuncSetAttribute in splat_queue_flush).  Same film as the direct atomics on all three."""
    torch = torch_cuda
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    scene = mi.load_dict(d)
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5})
    deferred = integ.render(scene, seed=2, spp=1, develop=False)
    integ.direct_splat = True
    direct = integ.render(scene, seed=2, spp=1, develop=False)
    scale = float(direct[..., :3].max())
    assert scale > 0 and torch.isfinite(deferred).all()
    assert float((deferred - direct).abs().max()) < 1e-4 * scale
    scene.release()


def test_particle_tracer_gradient_reference_test06(mi, orc, torch_cuda, kats):
    """src/integrators/tests/test_ptracer.py:204-225 (test06_ptracer_gradients) as written there, against ITS hard-coded number:
    traverse -> keep the radiance -> set it to 1 -> render (seed 0) -> loss = mean(image) -> render_backward(seed 3, 64 spp in 4 passes)
    -> d loss / d radiance == 0.33647 (atol 1e-5), on the device and through the public API."""
    torch = torch_cuda
    from test_oracle_path import ptracer_test_scene
    k = kats["ptracer_radiance_gradient"]
    scene = mi.load_dict(ptracer_test_scene(mi, radiance=(1.0, 0.5, 0.2)))
    integrator = mi.make_integrator(scene.integrator_props)
    params = mi.traverse(scene)
    key = "emitter.emitter.radiance.value"
    params.keep([key])
    assert list(params.keys()) == [key]
    params[key] = torch.ones(3, device="cuda"); params.update()
    image = integrator.render(scene, seed=0, spp=64).clone().requires_grad_(True)
    image.mean().backward()
    adjoint = image.grad
    g = integrator.render_backward(scene, params, adjoint, seed=3, spp=64)
    torch.cuda.synchronize()
    got = params.grad(key).cpu().numpy()
    assert got.shape == (3,) and np.allclose(got, k["gradient"], atol=k["atol"]), got
    # and the forward film of the same seeds carries the same number (linearity), as the oracle's does
    img = integrator.render(scene, seed=3, spp=64).cpu().numpy()
    assert np.allclose(img.sum(axis=(0, 1)) / img.size, k["gradient"], atol=k["atol"])
    o = orc.OracleScene(scene, mi.abi)
    ref, _ = o.ptracer_render(o.params(integrator, 3, 64))
    assert np.allclose(img, ref, rtol=1e-4, atol=1e-6)
    # prb.py renders in one pass: a multi-pass request is refused instead of rendering something else
    with pytest.raises(mi.Mi355Error):
        mi.PRBIntegrator({"max_depth": 4, "samples_per_pass": 16}).render(scene, seed=3, spp=64)


def test_particle_tracer_reference_test01_to_05(mi, orc, torch_cuda):
    """test_ptracer.py test01 (area and directionalarea: >= 20 % of the image lit at 2 spp), test03 (hide_emitters -> black), test04
    (no emitter -> black, also undeveloped) and test05 (crop window == cropped full render, "even the noise pattern") on the device."""
    torch = torch_cuda
    from test_oracle_path import ptracer_test_scene, ptracer_test_scene_directional
    for d in (ptracer_test_scene(mi, radiance=(1.0, 0.5, 0.2)), ptracer_test_scene_directional(mi)):
        scene = mi.load_dict(d); integ = mi.make_integrator(scene.integrator_props)
        img = integ.render(scene, seed=0, spp=2).cpu().numpy()
        o = orc.OracleScene(scene, mi.abi); ref, _ = o.ptracer_render(o.params(integ, 0, 2))
        assert (img > 0).sum() >= 0.2 * img.size and np.allclose(img, ref, rtol=1e-3, atol=1e-5)
    scene = mi.load_dict(ptracer_test_scene(mi, hide_emitters=True))
    assert (mi.make_integrator(scene.integrator_props).render(scene, seed=0, spp=4).cpu().numpy() == 0).all()
    d = ptracer_test_scene(mi); del d["emitter"]
    d["receiver"] = ptracer_test_scene_directional(mi)["receiver"]
    scene = mi.load_dict(d); integ = mi.make_integrator(scene.integrator_props)
    assert (integ.render(scene, seed=0, spp=2).cpu().numpy() == 0).all() and (integ.render(scene, seed=0, spp=2, develop=False).cpu().numpy() == 0).all()
    full = mi.load_dict(ptracer_test_scene(mi, radiance=(1.0, 0.5, 0.2)))
    integ = mi.make_integrator(full.integrator_props)
    image1 = integ.render(full, seed=0, spp=4).cpu().numpy()
    d = ptracer_test_scene(mi, radiance=(1.0, 0.5, 0.2)); d["sensor"]["film"].update(crop_offset_x=2, crop_offset_y=3, crop_width=6, crop_height=4)
    image3 = integ.render(mi.load_dict(d), seed=0, spp=4).cpu().numpy()
    assert image3.shape == (4, 6, 3) and np.allclose(image1[3:7, 2:8], image3, atol=1e-4)


def test_particle_tracer_radiance_gradients_match_oracle(mi, orc, torch_cuda):
    """k_ptracer<GRAD> against the oracle on a scene with three emitters (rectangle, sphere, a collimated disk beam), a Gaussian film
    filter, Russian roulette and two passes; shards add up; a zero radiance channel still receives its derivative."""
    torch = torch_cuda
    T = mi.ScalarTransform4f
    d = small_cbox(mi, 32)
    d["lamp2"] = {"type": "sphere", "to_world": T.translate([0.4, -0.2, 0.3]).scale(0.15), "bsdf": {"type": "ref", "id": "white"},
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [2.0, 0.0, 5.0]}}}
    d["beam"] = {"type": "disk", "to_world": T.translate([-0.6, 0.5, 0.0]) @ T.rotate([1, 0, 0], 90.0) @ T.scale(0.2),
                 "bsdf": {"type": "ref", "id": "white"}, "emitter": {"type": "directionalarea", "radiance": {"type": "rgb", "value": [30.0, 20.0, 10.0]}}}
    scene = mi.load_dict(d); o = orc.OracleScene(scene, mi.abi)
    integ = mi.ParticleTracerIntegrator({"max_depth": 6, "rr_depth": 3, "samples_per_pass": 8})
    params = mi.traverse(scene); params.keep(r".*\.emitter\.radiance\.value")
    assert len(params) == 3
    gin = np.random.default_rng(2).normal(size=(32, 32, 3)).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=9, spp=16).cpu().numpy()
    p = o.params(integ, 9, 16); assert p.passes == 2
    ref = o.ptracer_backward(p, gin)
    lay, _ = o.param_layout()
    offs = [l[2] for l in lay if l[0] == mi.abi.PARAM_EMITTER_RADIANCE]
    assert len(offs) == 3
    for off in offs:
        assert np.abs(ref[off:off + 3]).min() > 0 and grad_close(g[off:off + 3], ref[off:off + 3], 1e-3)
    rest = np.ones(g.shape[0], dtype=bool)
    for off in offs: rest[off:off + 3] = False
    assert (g[rest] == 0).all()
    # two film-grid shards accumulate to the same buffer
    params.zero_grad()
    for r in range(2):
        p2 = integ._params(scene, 9, 16, shard=mi.shard_tiles(32, 32, r, 2, rows=4)); p2.flags |= mi.abi.FLAG_EMITTER_GRADS_ONLY
        gt = torch.tensor(gin, device="cuda")
        mi.abi.check(mi.abi.lib().mi355_render_backward(scene.handle(), C.byref(p2), C.c_void_p(gt.data_ptr()), None,
                                                        C.c_void_p(params.grad_buffer(gt.device).data_ptr()), None))
    torch.cuda.synchronize()
    g2 = params.grad_buffer(torch.device("cuda", 0)).cpu().numpy()
    assert np.allclose(g2, g, rtol=1e-3, atol=1e-4 * np.abs(g).max())


@pytest.mark.parametrize("scene_name", ["cbox-textured", "veach"])
def test_particle_tracer_all_parameter_gradients(mi, orc, torch_cuda, scene_name):
    """ptracer.render_backward for EVERY differentiable parameter of the build (radiances, constant albedos, albedo textures): the
    two-walk replay of k_ptracer<GRAD> against the oracle's stored-path restatement, the radiance slots against the single-walk
    kernel, and -- with the roulette off, where the film is polynomial in an albedo -- against finite differences of the device's own
    light-traced film."""
    torch = torch_cuda
    rng = np.random.default_rng(5)
    if scene_name == "veach":
        scene = mi.load_dict(mi.veach_mis_scene(32)); props = {"max_depth": 5, "rr_depth": 2, "samples_per_pass": 8}
    else:
        scene = mi.load_dict(_textured_cbox(mi)); props = {"max_depth": 6, "rr_depth": 3}
    integ = mi.ParticleTracerIntegrator(props)
    o = orc.OracleScene(scene, mi.abi)
    params = mi.traverse(scene)
    gin = rng.normal(size=(32, 32, 3)).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=6, spp=16).cpu().numpy()
    ref = o.ptracer_backward(o.params(integ, 6, 16), gin)
    lay, total = o.param_layout()
    for kind, index, off, size in lay:
        a, b = g[off:off + size], ref[off:off + size]
        if kind in (mi.abi.PARAM_VERTEX_POSITIONS, mi.abi.PARAM_BSDF_IOR):    # (the geometry pass was not asked for: tests/test_gpu_geometry.py)
            assert np.abs(a).max() == 0 and np.abs(b).max() == 0
            continue
        assert np.isfinite(a).all() and np.abs(b).max() > 0, (kind, index)
        assert grad_close(a, b, 1e-3, "kind %d index %d" % (kind, index))
    only = mi.traverse(scene); only.keep(r".*\.emitter\.radiance\.value")
    g1 = integ.render_backward(scene, only, torch.tensor(gin), seed=6, spp=16).cpu().numpy()
    for kind, index, off, size in lay:
        if kind == mi.abi.PARAM_EMITTER_RADIANCE:
            assert np.allclose(g1[off:off + size], g[off:off + size], rtol=1e-3, atol=1e-5 * np.abs(g).max())
        else:
            assert (g1[off:off + size] == 0).all()
    if scene_name == "cbox-textured":
        integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 100})
        params.zero_grad()
        g = integ.render_backward(scene, params, torch.tensor(gin), seed=9, spp=64).cpu().numpy()
        gt = torch.tensor(gin, device="cuda").double()
        def loss():
            return float((integ.render(scene, seed=9, spp=64).double() * gt).sum())
        key = [k for k in params._entries if params._entries[k][0] == mi.abi.PARAM_BSDF_REFLECTANCE][0]
        kind, index, shape = params._entries[key]; off, size = params._layout[(kind, index)]
        base = params[key].clone(); eps = 1e-2
        for c in range(3):
            e = torch.zeros(3, device="cuda"); e[c] = eps
            params[key] = base + e; params.update(); hi = loss()
            params[key] = base - e; params.update(); lo = loss()
            assert abs(g[off + c] - (hi - lo) / (2 * eps)) < 2e-2 * abs(g[off + c]) + 1e-3, (c, g[off + c], (hi - lo) / (2 * eps))
        params[key] = base; params.update()


def _beam_scene(mi, res=32):
    """The ingredients of the reference's lens-design notebook scene (notebooks/: a collimated `directionalarea` beam leaving a
    `disk`, a `dispersive` glass element, a diffuse screen), light-traced onto a perspective film."""
    T = mi.ScalarTransform4f
    return {"type": "scene",
            "sensor": {"type": "perspective", "fov": 40.0, "to_world": T.look_at([0, 0.6, 3.2], [0, 0, 0], [0, 1, 0]),
                       "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "box"}}},
            "glass": {"type": "dispersive", "cauchy_a": 1.5046, "cauchy_b": 0.0042},
            "white": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.7, 0.6]}},
            "beam": {"type": "disk", "to_world": T.translate([-1.5, 0, 0]) @ T.rotate([0, 1, 0], 90.0) @ T.scale(0.35),
                     "bsdf": {"type": "ref", "id": "white"},
                     "emitter": {"type": "directionalarea", "radiance": {"type": "rgb", "value": [6.0, 5.0, 4.0]}}},
            "lens": {"type": "sphere", "to_world": T.translate([-0.4, 0, 0]) @ T.scale(0.5), "bsdf": {"type": "ref", "id": "glass"}},
            "screen": {"type": "rectangle", "to_world": T.translate([1.2, 0, 0]) @ T.rotate([0, 1, 0], -90.0) @ T.scale(1.2),
                       "bsdf": {"type": "ref", "id": "white"}},
            "floor": {"type": "disk", "to_world": T.translate([0, -0.6, 0]) @ T.rotate([1, 0, 0], -90.0) @ T.scale((2.0, 1.5, 1.0)),
                      "bsdf": {"type": "ref", "id": "white"}}}


@pytest.mark.parametrize("case", ["beam-ptracer", "beam-path", "disk-light"])
def test_disk_and_directional_area(mi, orc, torch_cuda, case):
    """Disk (disk.cpp:392-519, :173-191) as an intersected surface, as a sampled area light and as the source of a
    `directionalarea` beam (directionalarea.cpp:88-136: sample_ray along the normal with weight area * radiance, never connected,
    eval = 0), plus the `dispersive` material at its nominal wavelength."""
    torch = torch_cuda
    d = _beam_scene(mi)
    if case == "disk-light":       # a plain area light on the disk, no collimation: path / prb / gradients see it
        d["beam"]["emitter"] = {"type": "area", "radiance": {"type": "rgb", "value": [6.0, 5.0, 4.0]}}
    scene = mi.load_dict(d)
    o = orc.OracleScene(scene, mi.abi)
    rays = _rays_for(scene, o, n=8192, seed=5)
    got, ref = _gpu_intersect(mi, torch, scene, rays), o.ray_intersect(rays)
    hg, hr = orc.unpack_hits(got), orc.unpack_hits(ref)
    same = (hg["shape"] == hr["shape"]) & (hg["occluded"] == hr["occluded"])
    assert same.mean() > 0.999 and np.allclose(hg["t"][same & np.isfinite(hr["t"])], hr["t"][same & np.isfinite(hr["t"])], rtol=5e-5, atol=5e-6)
    assert (hr["shape"][np.isfinite(hr["t"])] == 0).any() and (hr["shape"][np.isfinite(hr["t"])] == 3).any()   # both disks are seen
    if case == "beam-ptracer":
        integ = mi.ParticleTracerIntegrator({"max_depth": 6, "rr_depth": 4})
        integ.counters(scene, reset=True); o.counters(reset=True)
        f = integ.render(scene, seed=3, spp=64, develop=False).cpu().numpy()
        c = integ.counters(scene, reset=True)
        ref, _ = o.ptracer_render(o.params(integ, 3, 64), develop=False)
        r = o.counters()
        scale = np.percentile(ref[..., :3], 99.5)
        assert ref[..., :3].max() > 0 and np.isfinite(f).all()
        assert np.percentile(np.abs(f - ref), 99.5) < 5e-3 * scale and abs(f.sum() / ref.sum() - 1) < 5e-3
        for k in ("n_isect", "n_shade", "n_shadow"):
            assert abs(int(c[k]) - r[k]) <= 2e-3 * r[k] + 2, (k, c[k], r[k])
        # the reference's loop intersects once more after the last vertex a path may shade and never reads the result (ptracer.cpp:264-269): the
        # oracle performs it as written and counts it apart, the device skips it -- so the live counts above agree and this one is not zero
        assert r["n_isect_dead"] > 0
        return
    for name in ("path", "prb"):
        integ = mi.make_integrator({"type": name, "max_depth": 6, "rr_depth": 3})
        spp, seed, n = 8, 3, 32 * 32 * 8
        ref = o.lane_radiance(o.params(integ, seed, spp), 0, n)
        got = integ.lane_radiance(scene, seed, spp, 0, n).cpu().numpy()
        close = np.isclose(got[:, :3], ref[:, :3], rtol=5e-4, atol=1e-5).all(axis=1) & (got[:, 3] == ref[:, 3])
        assert close.mean() > 0.995, "diverged lanes: %g" % (1 - close.mean())
        if case == "beam-path":    # a collimated emitter is invisible to unidirectional path tracing (no NEE, eval = 0)
            assert (ref[:, :3] == 0).all() and (got[:, :3] == 0).all()
    if case == "disk-light":
        assert ref[:, :3].max() > 0
        params = mi.traverse(scene)
        gin = np.random.default_rng(4).normal(size=(32, 32, 3)).astype(np.float32)
        g = integ.render_backward(scene, params, torch.tensor(gin), seed=7, spp=16).cpu().numpy()
        g_ref = o.render_backward(o.params(integ, 7, 16), gin)
        assert np.abs(g_ref).max() > 0 and grad_close(g, g_ref, 1e-3)


@pytest.mark.parametrize("scene_name,integrator_name", [("cbox", "path"), ("cbox", "prb"), ("veach", "path")])
def test_image_z_test_against_oracle(mi, orc, torch_cuda, scene_name, integrator_name):
    """The reference's render test (src/render/tests/test_renders.py:159-233) with the oracle in the role of the stored reference
    image: per-pixel Z-test of the device image (sample budget 2e6, seed 0) against the reference mean, with the reference's
    per-sample variance clamped at 1e-4, significance 0.01 with the Sidak correction over the pixels, passed when >= 99.75 % of the
    pixels accept.  Different seeds on the two sides: this is the statistical statement (same estimator), not lane parity."""
    import math
    res = 16
    if scene_name == "cbox":
        scene = mi.load_dict(small_cbox(mi, res, rfilter={"type": "box"}))
    else:
        d = mi.veach_mis_scene(res); d["sensor"]["film"]["rfilter"] = {"type": "box"}; scene = mi.load_dict(d)
    integ = mi.make_integrator({"type": integrator_name, "max_depth": 6, "rr_depth": 5})
    o = orc.OracleScene(scene, mi.abi)
    spp_ref = 2048 if scene_name == "cbox" else 8192
    lanes = o.lane_radiance(o.params(integ, 12345, spp_ref), 0, res * res * spp_ref)[:, :3].reshape(res * res, spp_ref, 3).astype(np.float64)
    ref, ref_var = lanes.mean(axis=1).reshape(res, res, 3), lanes.var(axis=1).reshape(res, res, 3)    # box filter: a lane lands in its own pixel
    pixel_count = res * res
    spp = int(2e6) // pixel_count
    img = integ.render(scene, seed=0, spp=spp).cpu().numpy().astype(np.float64)
    got = integ.lane_radiance(scene, 0, spp, 0, res * res * spp).cpu().numpy()[:, :3].reshape(res * res, spp, 3).astype(np.float64)
    assert np.allclose(got.mean(axis=1).reshape(res, res, 3), img, rtol=1e-4, atol=1e-6)           # the image IS the mean of its lanes
    # the reference image itself is an estimate here, so both sample variances enter (the glossy scene's heavy tails make the
    # variance of a 2048-sample reference alone too optimistic for a handful of pixels)
    var_pair = np.maximum(ref_var, 1e-4) / spp_ref + np.maximum(got.var(axis=1).reshape(res, res, 3), 1e-4) / spp
    z = np.abs(img - ref) / np.sqrt(var_pair)
    p_value = 2.0 * (1.0 - 0.5 * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0))))
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / pixel_count)
    success = p_value > alpha
    assert np.count_nonzero(success) / 3 >= 0.9975 * pixel_count, (np.count_nonzero(~success), p_value.min(), alpha)
    if scene_name == "cbox":
        assert abs(img.mean() / ref.mean() - 1) < 0.01
    else:               # rare, very bright samples carry the glossy scene's mean: compare where the estimates are stable
        calm = (ref_var / spp_ref < (0.02 * ref) ** 2) & (ref > 0)
        assert calm.mean() > 0.3 and abs(np.median(img[calm] / ref[calm]) - 1) < 0.01


def test_tune_picks_a_variant_without_changing_the_image(mi, torch_cuda, veach64):
    """PathIntegrator.tune(): times re-packing against path regeneration on the device and keeps the faster; the film is the same."""
    integ = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5})
    before = integ.render(veach64, seed=3, spp=128).cpu().numpy()
    t = integ.tune(veach64, spp=128)
    assert set(t) == {"repack", "regen"} and all(v > 0 for v in t.values()) and integ.regen == (t["regen"] < t["repack"]) and not integ.timing
    after = integ.render(veach64, seed=3, spp=128).cpu().numpy()
    assert np.allclose(before, after, rtol=1e-4, atol=1e-6)
    assert set(integ.tune(veach64, spp=16)) == {"repack"} and integ.regen is False        # regeneration needs >= 128 spp


def _geo_floor_scene(mi, offset, res=32):
    """A large diffuse floor MESH moving along its normal: neither the camera nor the light sees a silhouette of it."""
    T = mi.ScalarTransform4f
    quad = np.array([[-4, 0, 4], [4, 0, 4], [4, 0, -4], [-4, 0, -4]], dtype=np.float32) + np.array([0, offset, 0], dtype=np.float32)
    return {"type": "scene",
            "sensor": {"type": "perspective", "fov": 35.0, "to_world": T.look_at([0.4, 2.5, 1.5], [0, 0, 0], [0, 1, 0]),
                       "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}}},
            "white": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.6, 0.4]}},
            "quad": {"type": "mesh", "vertex_positions": quad, "faces": np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), "bsdf": {"type": "ref", "id": "white"}},
            "light": {"type": "rectangle", "to_world": T.look_at([1.9, 1.3, -0.3], [0, 0, 0], [0, 1, 0]) @ T.scale(0.3),
                      "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [20.0, 20.0, 20.0]}}}}


def _geo_lens_scene(mi, scale, material, res=32, rfilter="gaussian", sub=2, ior=1.5):
    """The arrangement of the fork's lens-design notebooks in small: a collimated `directionalarea` beam from a disk, a refracting (or
    mirroring) MESH with smooth shading normals whose size is the design variable, a diffuse screen watched by a perspective camera."""
    T = mi.ScalarTransform4f
    v, f = _icosphere(sub)
    pos = (v * 0.5 * scale).astype(np.float32) + np.array([-0.3, 0, 0], dtype=np.float32)
    d = {"type": "scene",
         "sensor": {"type": "perspective", "fov": 45.0, "to_world": T.look_at([0.2, 0.5, 3.2], [0.6, 0, 0], [0, 1, 0]),
                    "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": rfilter}}},
         "glass": {"type": "dielectric", "int_ior": ior, "ext_ior": 1.0}, "mirror": {"type": "conductor"},
         "white": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.7, 0.6]}},
         "beam": {"type": "disk", "to_world": T.translate([-1.6, 0, 0]) @ T.rotate([0, 1, 0], 90.0) @ T.scale(0.3), "bsdf": {"type": "ref", "id": "white"},
                  "emitter": {"type": "directionalarea", "radiance": {"type": "rgb", "value": [6.0, 5.0, 4.0]}}},
         "lens": {"type": "mesh", "vertex_positions": pos, "faces": f.astype(np.uint32), "vertex_normals": v.astype(np.float32), "bsdf": {"type": "ref", "id": material}},
         "screen": {"type": "rectangle", "to_world": T.translate([1.4, 0, 0]) @ T.rotate([0, 1, 0], -90.0) @ T.scale(1.2), "bsdf": {"type": "ref", "id": "white"}}}
    return d


def _geo_mirror_scene(mi, angle, res=32):
    """A collimated beam on a large flat MIRROR mesh at 45 degrees that turns about the z axis by `angle` (radians): the mirrored spot
    sweeps over a ceiling screen.  (A curved mirror MESH with interpolated normals is not used here on purpose: the shading-normal
    correction of the throughput is a sawtooth over the facets whose in-facet slope the interior derivative -- this pass, and AD --
    picks up while its jumps at the facet edges cancel it in a finite difference.)"""
    T = mi.ScalarTransform4f
    u = np.array([1.0, 1.0, 0.0]) / np.sqrt(2); w = np.array([0, 0, 1.0])
    quad = np.stack([-2 * u - 2 * w, -2 * u + 2 * w, 2 * u + 2 * w, 2 * u - 2 * w])
    ca, sa = np.cos(angle), np.sin(angle)
    rot = np.array([[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]])
    quad = (quad @ rot.T).astype(np.float32)
    return {"type": "scene",
            "sensor": {"type": "perspective", "fov": 45.0, "to_world": T.look_at([0.0, 0.2, 2.8], [0.0, 2.0, 0], [0, 1, 0]),
                       "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}}},
            "mirror": {"type": "conductor", "eta": {"type": "rgb", "value": [0.143, 0.375, 1.442]}, "k": {"type": "rgb", "value": [3.983, 2.386, 1.603]}}, "white": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.7, 0.6]}},
            "beam": {"type": "disk", "to_world": T.translate([-1.6, 0.0, 0]) @ T.rotate([0, 1, 0], 90.0) @ T.scale(0.3), "bsdf": {"type": "ref", "id": "white"},
                     "emitter": {"type": "directionalarea", "radiance": {"type": "rgb", "value": [6.0, 5.0, 4.0]}}},
            "m": {"type": "mesh", "vertex_positions": quad, "faces": np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), "bsdf": {"type": "ref", "id": "mirror"}},
            "screen": {"type": "rectangle", "to_world": T.translate([0.0, 2.0, 0]) @ T.rotate([1, 0, 0], 90.0) @ T.scale(2.0), "bsdf": {"type": "ref", "id": "white"}}}


@pytest.mark.parametrize("case", ["diffuse-floor", "glass-lens", "glass-lens-tent", "turning-mirror"])
def test_particle_tracer_vertex_forward_mode(mi, torch_cuda, case):
    """mi355_ptracer_forward_geometry (mi355_geo_forward.h): the derivative of the light-traced image with respect to mesh vertex
    positions -- what the fork's lens-design notebooks take from Dr.Jit's AD -- against the reference's own check for gradients, a
    central finite difference of the primal image with the same seeds (test_ad_integrators.py:742-786, :942-962).  The scenes keep
    every silhouette of the moving mesh out of sight of the camera and of the light, because neither AD nor this pass differentiates
    visibility; the film's 4-pixel border is left out for the same reason (splats entering the frame are a boundary term)."""
    torch = torch_cuda
    integ = mi.ParticleTracerIntegrator({"max_depth": 5, "rr_depth": 100})
    eps, spp = 2e-3, 256
    if case == "diffuse-floor":
        make = lambda t: _geo_floor_scene(mi, t); base, shape = 0.0, "quad"
        tangent = np.tile(np.array([0, 1, 0], dtype=np.float32), (4, 1))
    elif case == "turning-mirror":      # d(vertex)/d(angle) = z x position; a gold mirror, so that the Fresnel weight varies with the angle too
        make = lambda t: _geo_mirror_scene(mi, t); base, shape = 0.0, "m"
        q0 = _geo_mirror_scene(mi, 0.0)["m"]["vertex_positions"]
        tangent = np.stack([-q0[:, 1], q0[:, 0], np.zeros(4, dtype=np.float32)], axis=1).astype(np.float32)
    else:
        rfilter = "tent" if case.endswith("tent") else "gaussian"
        make = lambda t: _geo_lens_scene(mi, t, "glass", rfilter=rfilter); base, shape = 1.0, "lens"
        tangent = (_icosphere(2)[0] * 0.5).astype(np.float32)               # d(vertex) / d(scale)
    scene = mi.load_dict(make(base))
    g = integ.render_forward_geometry(scene, shape, tangent, seed=1, spp=spp).cpu().numpy().astype(np.float64)
    hi = integ.render(mi.load_dict(make(base + eps)), seed=1, spp=spp).cpu().numpy().astype(np.float64)
    lo = integ.render(mi.load_dict(make(base - eps)), seed=1, spp=spp).cpu().numpy().astype(np.float64)
    fd = (hi - lo) / (2 * eps)
    g, fd = g[4:-4, 4:-4], fd[4:-4, 4:-4]
    assert np.isfinite(g).all() and np.abs(fd).max() > 0
    rel = np.sqrt(((g - fd) ** 2).sum() / (fd ** 2).sum())
    corr = (g * fd).sum() / np.sqrt((g ** 2).sum() * (fd ** 2).sum())
    assert rel < 0.08 and corr > 0.995, (rel, corr)
    # a zero tangent gives a zero derivative; a lens sensor, a non-mesh or an emitting shape are refused
    z = integ.render_forward_geometry(scene, shape, np.zeros_like(tangent), seed=1, spp=16).cpu().numpy()
    assert (z == 0).all()
    with pytest.raises(mi.Mi355Error):
        integ.render_forward_geometry(scene, "light" if case == "diffuse-floor" else "screen", np.zeros((0, 3), dtype=np.float32), seed=1, spp=4)


@pytest.mark.parametrize("rfilter", ["gaussian", "tent", "box"])
def test_forward_mode_derivative_film_deferred_equals_direct(mi, torch_cuda, rfilter):
    """The derivative film of the vertex / IOR forward mode goes through the deferred, tile-sorted splat as the primal film does
    (40-byte records: position, colour tangent | colour, position tangent; k_splat_tiles<true> re-evaluates the dual filter weights);
    MI355_FLAG_DIRECT_SPLAT keeps the plain float atomics.  Same film up to summation order -- also when the queue overflows and part of
    the records takes the direct path."""
    scene = mi.load_dict(_geo_lens_scene(mi, 1.0, "glass", rfilter=rfilter))
    tangent = (_icosphere(2)[0] * 0.5).astype(np.float32)
    films = {}
    for name, direct, cap in (("direct", True, 0), ("deferred", False, 0), ("overflow", False, 4096)):
        integ = mi.ParticleTracerIntegrator({"max_depth": 5, "rr_depth": 100}); integ.direct_splat = direct
        mi.abi.check(mi.abi.lib().mi355_debug_set_splat_capacity(scene.handle(), cap))
        films[name] = integ.render_forward_geometry(scene, "lens", tangent, seed=1, spp=64).cpu().numpy().astype(np.float64)
        both = integ.render_forward_design(scene, shape="lens", vertex_tangent=tangent, bsdf="glass", d_int_ior=0.3, seed=1, spp=64).cpu().numpy().astype(np.float64)
        films[name + "+ior"] = both
    mi.abi.check(mi.abi.lib().mi355_debug_set_splat_capacity(scene.handle(), 0))
    for suffix in ("", "+ior"):
        ref = films["direct" + suffix]; scale = np.abs(ref).max()
        assert scale > 0 and np.isfinite(ref).all()
        for name in ("deferred", "overflow"):
            assert np.abs(films[name + suffix] - ref).max() < 2e-5 * scale, (name, suffix, np.abs(films[name + suffix] - ref).max() / scale)
    scene.release()


def test_degenerate_scenes_and_budgets(mi, orc, torch_cuda):
    """Edge cases of the domain: a scene without shapes, a scene without emitters, max_depth = 0 and 1, a single sample per
    pixel, an empty shard, and a sample budget beyond 2^32 lanes (integrator.cpp:247-265 splits such renders into passes; the
    C ABI asks the caller to)."""
    torch = torch_cuda
    T = mi.ScalarTransform4f
    sensor = {"type": "perspective", "fov": 40, "to_world": T.look_at([0, 0, 4], [0, 0, 0], [0, 1, 0]),
              "film": {"type": "hdrfilm", "width": 16, "height": 12, "rfilter": {"type": "gaussian"}}}
    empty = mi.load_dict({"type": "scene", "sensor": dict(sensor)})
    dark = mi.load_dict({"type": "scene", "sensor": dict(sensor), "wall": {"type": "rectangle", "bsdf": {"type": "diffuse"}}})
    for scene in (empty, dark):
        o = orc.OracleScene(scene, mi.abi)
        for integ in (mi.PathIntegrator({"max_depth": 4}), mi.PRBIntegrator({"max_depth": 4})):
            film = integ.render(scene, seed=1, spp=4, develop=False)
            assert float(film[..., :3].abs().max()) == 0.0 and float(film[..., 3].min()) > 0.0          # weights are still splatted
            ref, _ = o.render(o.params(integ, 1, 4), develop=False)
            assert np.allclose(film.cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
        prb = mi.PRBIntegrator({"max_depth": 4}); params = mi.traverse(scene)
        g = prb.render_backward(scene, params, torch.ones((12, 16, 3), device="cuda"), seed=2, spp=4)
        assert float(g.abs().max()) == 0.0
    scene = mi.load_dict(small_cbox(mi, 16)); o = orc.OracleScene(scene, mi.abi)
    for md, spp in ((0, 4), (1, 4), (6, 1)):
        for integ in (mi.PathIntegrator({"max_depth": md}), mi.PRBIntegrator({"max_depth": md})):
            img = integ.render(scene, seed=3, spp=spp).cpu().numpy()
            ref, _ = o.render(o.params(integ, 3, spp))
            assert np.abs(img - ref).max() <= 1e-3 * max(ref.max(), 1e-6), (md, spp)
            if md == 0:
                assert img.max() == 0.0
    integ = mi.PathIntegrator({"max_depth": 3})
    film = torch.zeros((16, 16, 4), device="cuda")
    p = integ._params(scene, 0, 4, shard=(256, 0))                       # empty shard: nothing to do, not an error
    mi.abi.check(mi.abi.lib().mi355_render_accumulate(scene.handle(), C.byref(p), C.c_void_p(film.data_ptr()), None))
    torch.cuda.synchronize(); assert float(film.abs().max()) == 0.0
    p = integ._params(scene, 0, 1 << 25)                                  # 256 * 2^25 = 2^33 samples: split into passes as the reference does
    assert p.passes >= 2 and 256 * ((1 << 25) // p.passes) <= 0xffffffff
    p.passes = 0                                                          # ... and one wavefront of that size is refused by the C ABI
    assert mi.abi.lib().mi355_render_accumulate(scene.handle(), C.byref(p), C.c_void_p(film.data_ptr()), None) == mi.abi.ERR_INVALID


def test_full_size_properties(mi, torch_cuda):
    """BASELINE configs[1] and [2] at their FULL sizes (512^2 x 256 spp forward, 512^2 x 64 spp PRB backward with the 256^2
    albedo texture) through properties that need no oracle run of that size:
      * radiance is linear in the emitter radiance and scaling by 2 is exact in fp32, so the film doubles up to the order of
        its float atomics (rel 1e-5);
      * the three shard shapes (bands, interleaved tiles, spp slices) and the three re-packing modes give the same film;
      * W channel = sum of filter weights: every sample deposits the same total weight away from the border;
      * PRB: sum_c radiance_c * dLoss/dradiance_c = <grad_in, image> (the image is linear in the radiance), and the one-pass
        adjoint agrees with the two-pass one."""
    torch = torch_cuda
    d = mi.cornell_box(); d["sensor"]["film"]["width"] = 512; d["sensor"]["film"]["height"] = 512
    scene = mi.load_dict(d)
    integ = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5}); integ.mode = mi.abi.MODE_STATIC
    film = integ.render(scene, seed=1, spp=256, develop=False)
    assert torch.isfinite(film).all()
    w = film[8:-8, 8:-8, 3]
    assert float((w / w.mean() - 1).abs().max()) < 0.2 and abs(float(film[..., 3].sum()) / (512 * 512 * 256) - float(w.mean()) / 256) < 0.02 * float(w.mean()) / 256
    params = mi.traverse(scene)
    key = [k for k in params.keys() if k.endswith("radiance.value")][0]
    rad = params[key].clone()
    params[key] = rad * 2; params.update()
    film2 = integ.render(scene, seed=1, spp=256, develop=False)
    params[key] = rad; params.update()
    assert float(film[..., :3].mean()) > 1.0
    assert torch.allclose(film2[..., :3], film[..., :3] * 2, rtol=1e-5, atol=1e-4) and torch.allclose(film2[..., 3], film[..., 3], rtol=1e-5)
    for mode in (mi.abi.MODE_FUSED, mi.abi.MODE_WAVEFRONT):
        im = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5}); im.mode = mode
        other = im.render(scene, seed=1, spp=256, develop=False)
        assert torch.allclose(other, film, rtol=2e-4, atol=2e-3)
    h = scene.handle(); lib = mi.abi.lib()
    fast = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5})
    for shards in ([mi.shard_tiles(512, 512, r, 8) for r in range(8)], [mi.shard_pixels(512 * 512, r, 3) for r in range(3)]):
        acc = torch.zeros_like(film)
        for sh in shards:
            p = fast._params(scene, 1, 256, shard=sh)
            mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None))
        torch.cuda.synchronize()
        assert torch.allclose(acc, film, rtol=2e-4, atol=2e-3)
    acc = torch.zeros_like(film)
    for s0 in range(0, 256, 64):
        p = fast._params(scene, 1, 256, spp_shard=(s0, 64))
        mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.allclose(acc, film, rtol=2e-4, atol=2e-3)
    scene.release()
    # ---- config 3
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "raw": True, "filter_type": "bilinear", "wrap_mode": "repeat",
                                                     "data": np.random.default_rng(12345).uniform(0.2, 0.8, (256, 256, 3)).astype(np.float32)}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    scene3 = mi.load_dict(d)
    gin = torch.tensor(np.random.default_rng(3).uniform(0.5, 1.5, (512, 512, 3)).astype(np.float32), device="cuda")
    grads = {}
    for two_pass in (False, True):
        prb = mi.PRBIntegrator({"max_depth": 6, "rr_depth": 5}); prb.two_pass = two_pass
        params3 = mi.traverse(scene3)
        grads[two_pass] = prb.render_backward(scene3, params3, gin, seed=5, spp=64).clone()
        if not two_pass:
            assert prb.counters(scene3)["n_redo_groups"] == 0
            key3 = [k for k in params3.keys() if k.endswith("radiance.value")][0]
            lhs = float((params3.grad(key3) * params3[key3]).sum())
            img = prb.render(scene3, seed=5, spp=64)
            rhs = float((gin * img).sum())
            assert abs(lhs - rhs) < 2e-3 * abs(rhs), (lhs, rhs)
    a, b = grads[False], grads[True]
    assert torch.isfinite(a).all() and float((a - b).abs().max()) < 1e-3 * float(b.abs().max())
    assert float((a - b).abs().sum()) < 1e-4 * float(b.abs().sum())


def test_render_op_autograd(mi, orc, torch_cuda, cbox32):
    """mi.render as a differentiable op (util.py:410-523): loss.backward() fills the leaf's .grad with what
    render_backward computes for grad_in = dLoss/dImage, using seed_grad = tea32(seed, 1)[0]."""
    torch = torch_cuda
    scene = mi.load_dict(small_cbox(mi, 32))
    integ = mi.PRBIntegrator({"max_depth": 4})
    params = mi.traverse(scene)
    key = "green.reflectance.value"
    leaf = params[key].clone().requires_grad_(True)
    params[key] = leaf
    img = mi.render(scene, params, integrator=integ, seed=1, spp=8)
    loss = img.mean()
    loss.backward()
    assert leaf.grad is not None and torch.isfinite(leaf.grad).all() and leaf.grad.abs().max() > 0
    o = orc.OracleScene(scene, mi.abi)
    gin = np.full((32, 32, 3), 1.0 / (32 * 32 * 3), dtype=np.float32)
    seed_grad = mi.sample_tea_32(1, 1)[0]
    g_ref = o.render_backward(o.params(integ, seed_grad, 8), gin)
    lay, _ = o.param_layout()
    off = [o_ for k, i, o_, s in lay if k == 0 and scene.bsdfs[i].id == "green"][0]
    assert np.allclose(leaf.grad.cpu().numpy(), g_ref[off:off + 3], rtol=2e-3, atol=1e-7)
    # parameter edits reach the device replica: brighter green wall -> brighter image
    with torch.no_grad():
        leaf.mul_(0.5)
    img2 = mi.render(scene, params, integrator=integ, seed=1, spp=8)
    assert img2.mean() < img.mean()


def test_render_op_autograd_with_the_particle_tracer(mi, orc, torch_cuda):
    """mi.render with integrator = ptracer: the same differentiable op, its backward goes through the light tracer's path replay."""
    torch = torch_cuda
    scene = mi.load_dict(small_cbox(mi, 32))
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 3})
    params = mi.traverse(scene)
    keys = ["green.reflectance.value", "light.emitter.radiance.value"]
    leaves = {}
    for k in keys:
        leaves[k] = params[k].clone().requires_grad_(True); params[k] = leaves[k]
    img = mi.render(scene, params, integrator=integ, seed=1, spp=8)
    (img ** 2).mean().backward()
    o = orc.OracleScene(scene, mi.abi)
    gin = (2.0 * img.detach() / img.numel()).cpu().numpy()
    g_ref = o.ptracer_backward(o.params(integ, mi.sample_tea_32(1, 1)[0], 8), gin)
    lay, _ = o.param_layout()
    offs = {"green.reflectance.value": [o_ for k, i, o_, s in lay if k == 0 and scene.bsdfs[i].id == "green"][0],
            "light.emitter.radiance.value": [o_ for k, i, o_, s in lay if k == mi.abi.PARAM_EMITTER_RADIANCE][0]}
    for k in keys:
        got = leaves[k].grad.cpu().numpy(); ref = g_ref[offs[k]:offs[k] + 3]
        assert np.abs(ref).max() > 0 and grad_close(got, ref, 1e-3, k)


def test_errors_are_loud(mi, torch_cuda, cbox32):
    p = mi.PathIntegrator()._params(cbox32, 0, 0)
    film = torch_cuda.zeros((32, 32, 4), device="cuda")
    with pytest.raises(mi.Mi355Error):
        mi.abi.check(mi.abi.lib().mi355_render_accumulate(cbox32.handle(), C.byref(p), C.c_void_p(film.data_ptr()), None))
    p = mi.PathIntegrator()._params(cbox32, 0, 4, shard=(1000, 100))
    with pytest.raises(mi.Mi355Error):
        mi.abi.check(mi.abi.lib().mi355_render_accumulate(cbox32.handle(), C.byref(p), C.c_void_p(film.data_ptr()), None))


def test_inverse_rendering_loop(mi, torch_cuda):
    """End-to-end: mi.render op + PRB backward + Adam recover the red wall's albedo (examples/optimize_albedo.py)."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "optimize_albedo.py")
    spec = importlib.util.spec_from_file_location("optimize_albedo", path); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    losses, errs = m.main(iterations=30, res=48, spp=16, verbose=False)
    assert errs[-1] < 0.25 * errs[0] and min(losses[-5:]) < 0.7 * losses[0]      # the loss floor is Monte Carlo noise


def test_particle_tracer_ior_forward_mode(mi, torch_cuda):
    """The material part of a design tangent: d(image)/d(int_ior) of the lens glass, and a combined tangent (scale and IOR together),
    against finite differences of the primal image -- the notebooks optimise IOR / Abbe numbers next to the surface shapes."""
    integ = mi.ParticleTracerIntegrator({"max_depth": 5, "rr_depth": 100})
    eps, spp = 2e-3, 256
    tangent = (_icosphere(2)[0] * 0.5).astype(np.float32)
    scene = mi.load_dict(_geo_lens_scene(mi, 1.0, "glass"))
    for name, kw, make in (("ior", dict(bsdf="glass", d_int_ior=1.0), lambda t: _geo_lens_scene(mi, 1.0, "glass", ior=1.5 + t)),
                           ("both", dict(shape="lens", vertex_tangent=tangent, bsdf="glass", d_int_ior=-0.5), lambda t: _geo_lens_scene(mi, 1.0 + t, "glass", ior=1.5 - 0.5 * t))):
        g = integ.render_forward_design(scene, seed=1, spp=spp, **kw).cpu().numpy().astype(np.float64)
        hi = integ.render(mi.load_dict(make(eps)), seed=1, spp=spp).cpu().numpy().astype(np.float64)
        lo = integ.render(mi.load_dict(make(-eps)), seed=1, spp=spp).cpu().numpy().astype(np.float64)
        fd = (hi - lo) / (2 * eps)
        g, fd = g[4:-4, 4:-4], fd[4:-4, 4:-4]
        rel = np.sqrt(((g - fd) ** 2).sum() / (fd ** 2).sum()); corr = (g * fd).sum() / np.sqrt((g ** 2).sum() * (fd ** 2).sum())
        assert np.abs(fd).max() > 0 and rel < 0.08 and corr > 0.995, (name, rel, corr)
    with pytest.raises(mi.Mi355Error):
        integ.render_forward_design(scene, bsdf="white", d_int_ior=1.0, seed=1, spp=4)       # not a dielectric


def test_lens_focusing_loop(mi, torch_cuda):
    """examples/focus_lens.py: the notebooks' design loop with one scalar -- ONE scene whose lens mesh is rewritten through
    params['lens.vertex_positions'] every iteration (device refit), light-traced image, spot-size loss back-propagated through the
    reverse-mode geometry adjoint behind mi.render, gradient descent: the spot shrinks."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("focus_lens", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "focus_lens.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    losses, x = mod.main(iterations=20, res=48, spp=32, verbose=False)
    assert np.isfinite(losses).all() and np.mean(losses[-3:]) < 0.6 * np.mean(losses[:3]), (losses[:3], losses[-3:], x)


def test_full_size_properties_config_4(mi, torch_cuda):
    """BASELINE configs[3] at its FULL size (the Veach-MIS arrangement, 1024^2 x 512 spp = 5.4e8 camera paths through the FULL kernels:
    rough conductors, a dielectric sphere, sphere emitters, MIS) through properties that need no oracle run of that size:
      * the film is linear in the emitter radiances and a factor 2 is exact in fp32: every pixel doubles, up to the order of its atomics;
      * static lanes, in-workgroup re-packing and path regeneration are three schedules of the SAME samples: identical films;
      * 16-row tiles dealt to 8 ranks (the multi-GPU decomposition of bench.py) and spp slices accumulate to the whole film;
      * W channel = sum of filter weights: the same total per sample away from the border."""
    torch = torch_cuda
    scene = mi.load_dict(mi.veach_mis_scene(1024))
    W = H = 1024; spp = 512
    base = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5}); base.mode = mi.abi.MODE_STATIC
    film = base.render(scene, seed=3, spp=spp, develop=False)
    assert film.shape == (H, W, 4) and torch.isfinite(film).all() and float(film[..., :3].mean()) > 0.0
    w = film[8:-8, 8:-8, 3]
    assert float((w / w.mean() - 1).abs().max()) < 0.2
    # ---- the three schedules
    for name, mode, regen in (("repack", mi.abi.MODE_FUSED, False), ("wavefront", mi.abi.MODE_WAVEFRONT, False), ("regen", mi.abi.MODE_FUSED, True)):
        it = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5}); it.mode = mode; it.regen = regen
        other = it.render(scene, seed=3, spp=spp, develop=False)
        err = float((other - film).abs().max() / film[..., :3].abs().max())
        assert torch.allclose(other, film, rtol=3e-4, atol=3e-3 * float(film[..., :3].mean())), (name, err)
    # ---- linearity in the radiances (all emitters scaled by 2)
    params = mi.traverse(scene)
    keys = [k for k in params.keys() if k.endswith("radiance.value")]
    assert len(keys) >= 4
    saved = {k: params[k].clone() for k in keys}
    for k in keys:
        params[k] = saved[k] * 2
    params.update()
    film2 = base.render(scene, seed=3, spp=spp, develop=False)
    for k in keys:
        params[k] = saved[k]
    params.update()
    assert torch.allclose(film2[..., :3], film[..., :3] * 2, rtol=1e-5, atol=1e-4 * float(film[..., :3].mean())) and torch.allclose(film2[..., 3], film[..., 3], rtol=1e-5)
    # ---- shard shapes
    h = scene.handle(); lib = mi.abi.lib()
    fast = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5})
    acc = torch.zeros_like(film)
    for r in range(8):
        p = fast._params(scene, 3, spp, shard=mi.shard_tiles(W, H, r, 8))
        mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.allclose(acc, film, rtol=3e-4, atol=3e-3 * float(film[..., :3].mean()))
    acc = torch.zeros_like(film)
    for s0 in range(0, spp, 128):
        p = fast._params(scene, 3, spp, spp_shard=(s0, 128))
        mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.allclose(acc, film, rtol=3e-4, atol=3e-3 * float(film[..., :3].mean()))
    scene.release()


def test_surface_interaction_fields_match_oracle(mi, orc, torch_cuda):
    """SurfaceInteraction3f field by field -- p, n, uv, sh_frame (s, t, n), wi, t, shape -- of the closest hit of the same rays, device
    (`mi355_debug_surface_interaction`, the compute_si of every kernel) against the oracle (`orc_ray_intersect`'s si_out): rectangles and
    cube meshes (Cornell box), spheres + rough plates + a dielectric sphere (Veach arrangement), an OBJ-style mesh with vertex normals and
    texture coordinates in the scene BVH, a disk.  Rays whose two closest hits tie (edges shared by two triangles) may report either
    primitive and are excluded, as in test_ray_intersect_*."""
    torch = torch_cuda
    lib = mi.abi.lib()
    T = mi.ScalarTransform4f
    # a UV sphere mesh (vertex normals + uvs, > 64 faces: goes to the BVH) and a disk in front of a Cornell box
    nu, nv = 24, 12
    th = np.linspace(0, np.pi, nv + 1)[:, None]; ph = np.linspace(0, 2 * np.pi, nu + 1)[None, :]
    P = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3).astype(np.float32)
    UV = np.stack([np.broadcast_to(ph / (2 * np.pi), (nv + 1, nu + 1)), np.broadcast_to(th / np.pi, (nv + 1, nu + 1))], -1).reshape(-1, 2).astype(np.float32)
    F = []
    for a in range(nv):
        for b in range(nu):
            i0 = a * (nu + 1) + b; i1 = i0 + 1; i2 = i0 + nu + 1; i3 = i2 + 1
            if a > 0: F.append((i0, i2, i1))
            if a < nv - 1: F.append((i1, i2, i3))
    F = np.array(F, dtype=np.uint32)
    scenes = []
    scenes.append(mi.load_dict(small_cbox(mi, 32)))
    scenes.append(mi.load_dict(mi.veach_mis_scene(32)))
    d = small_cbox(mi, 32)
    d["ball"] = {"type": "mesh", "vertex_positions": P * 0.3 + np.array([0.1, -0.2, 0.2], np.float32), "vertex_normals": P / np.linalg.norm(P, axis=1, keepdims=True),
                 "vertex_texcoords": UV, "faces": F, "bsdf": {"type": "diffuse"}}
    d["coin"] = {"type": "disk", "to_world": T.translate([-0.4, 0.3, 0.4]).rotate([0, 1, 0], 30).scale(0.25), "bsdf": {"type": "diffuse"}}
    scenes.append(mi.load_dict(d))
    for scene in scenes:
        o = orc.OracleScene(scene, mi.abi)
        rays = _rays_for(scene, o, 8192)
        hits_ref, si_ref = o.ray_intersect(rays, with_si=True)
        h = scene.handle()
        r = torch.tensor(rays, device="cuda"); out = torch.empty((rays.shape[0], 24), dtype=torch.float32, device="cuda")
        mi.abi.check(lib.mi355_debug_surface_interaction(h, C.c_void_p(r.data_ptr()), rays.shape[0], C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        si = out.cpu().numpy()
        shape_ref = si_ref[:, 15].view(np.uint32); shape = si[:, 15].view(np.uint32)
        valid = (shape_ref != 0xffffffff)
        same = valid & (shape == shape_ref) & np.isclose(si[:, 14], si_ref[:, 14], rtol=3e-5, atol=3e-6)
        assert valid.sum() > 0.5 * len(rays) and same.sum() > 0.995 * valid.sum(), (valid.sum(), same.sum())
        assert ((shape == 0xffffffff) == ~valid).mean() > 0.999
        # a ray through a shared edge may land on either triangle: the geometric normal / uv then differ legitimately; such rays show up as
        # a different normal at (almost) the same t -- they must be rare
        a, b = si[same], si_ref[same]
        scale = max(1.0, float(np.abs(b[:, 0:3]).max()))
        fields = {"p": (slice(0, 3), 2e-5 * scale), "n": (slice(3, 6), 2e-5), "uv": (slice(6, 8), 5e-5), "sh_n": (slice(8, 11), 5e-5), "wi": (slice(11, 14), 1e-4),
                  "sh_s": (slice(16, 19), 1e-4), "sh_t": (slice(19, 22), 1e-4)}
        for name, (sl, tol) in fields.items():
            err = np.abs(a[:, sl] - b[:, sl]).max(axis=1)
            assert (err <= tol).mean() > 0.998, (name, (err <= tol).mean(), err.max())
        # frames are orthonormal, wi is the local direction towards the ray origin
        s_, t_, n_ = a[:, 16:19], a[:, 19:22], a[:, 8:11]
        assert np.abs((s_ * t_).sum(1)).max() < 1e-4 and np.abs((s_ * n_).sum(1)).max() < 1e-4 and np.abs(np.linalg.norm(n_, axis=1) - 1).max() < 1e-4
        dirs = rays[same, 3:6] / np.linalg.norm(rays[same, 3:6], axis=1, keepdims=True)
        assert np.abs(a[:, 13] + (dirs * n_).sum(1)).max() < 2e-4
        scene.release()


@pytest.mark.parametrize("wrap_mode", ["repeat", "clamp", "mirror"])
def test_device_texture_fetch_identities(mi, orc, torch_cuda, wrap_mode):
    """The reference's own asserts for `bitmap` on the DEVICE texture fetch (`mi355_debug_texture_eval` = the texture_eval every kernel
    uses), and device == oracle on the same uv:
      * src/textures/tests/test_bitmap.py:143-157 (test04): uv = ((x + 1/2) / W, (y + 1/2) / H) returns texel (x, y) itself;
      * src/textures/tests/test_bitmap.py:58-130 (test03): a 20 x 20 grid over [0,1]^2 against the same grid shifted by -1 / +1
        (repeat: equal; mirror: equal to the reversed grid; clamp: equal to the border row / column), atol 1e-4 as there."""
    torch = torch_cuda
    rng = np.random.default_rng(8)
    tex = rng.uniform(0.0, 1.0, (8, 8, 3)).astype(np.float32)                 # stands in for noise_8x8.png (the reference's resources are a submodule that is not checked out)
    d = small_cbox(mi, 8)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True, "wrap_mode": wrap_mode}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    scene = mi.load_dict(d); o = orc.OracleScene(scene, mi.abi)
    lib = mi.abi.lib(); h = scene.handle()

    def dev_eval(uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        u = torch.tensor(uv, device="cuda"); out = torch.empty((uv.shape[0], 3), dtype=torch.float32, device="cuda")
        mi.abi.check(lib.mi355_debug_texture_eval(h, 0, C.c_void_p(u.data_ptr()), uv.shape[0], C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got, o.texture_eval(0, uv))                    # same arithmetic, same bits
        return got

    def grid(xs, ys):
        xv, yv = np.meshgrid(xs, ys)                                          # dr.meshgrid: x varies fastest
        return dev_eval(np.stack([xv.ravel(), yv.ravel()], -1))
    # test04: texel centres
    centres = np.array([[(x + 0.5) / 8, (y + 0.5) / 8] for y in range(8) for x in range(8)], dtype=np.float32)
    assert np.allclose(dev_eval(centres), tex.reshape(-1, 3), atol=1e-6)
    assert np.allclose(dev_eval([[(7 + 0.5) / 8, (1 + 0.5) / 8]])[0], tex[1, 7], atol=1e-6)   # "coordinates of green pixel: (7, 1)"
    # test03: wrap modes
    n = 20
    lin = lambda a, b: np.linspace(a, b, n, dtype=np.float32)
    ref = grid(lin(0, 1), lin(0, 1))
    if wrap_mode == "repeat":
        assert np.allclose(ref, grid(lin(-1, 0), lin(-1, 0)), atol=1e-4) and np.allclose(ref, grid(lin(1, 2), lin(1, 2)), atol=1e-4)
    elif wrap_mode == "mirror":
        assert np.allclose(ref, grid(lin(-1, 0)[::-1], lin(-1, 0)[::-1]), atol=1e-4) and np.allclose(ref, grid(lin(1, 2)[::-1], lin(1, 2)[::-1]), atol=1e-4)
    else:
        r3 = ref.reshape(n, n, 3)
        assert np.allclose(np.broadcast_to(r3[:1], (n, n, 3)), grid(lin(0, 1), lin(-1, 0)).reshape(n, n, 3), atol=1e-4)      # top: the first row, everywhere
        assert np.allclose(np.broadcast_to(r3[-1:], (n, n, 3)), grid(lin(0, 1), lin(1, 2)).reshape(n, n, 3), atol=1e-4)     # bottom
        assert np.allclose(np.broadcast_to(r3[:, :1], (n, n, 3)), grid(lin(-1, 0), lin(0, 1)).reshape(n, n, 3), atol=1e-4)   # left
        assert np.allclose(np.broadcast_to(r3[:, -1:], (n, n, 3)), grid(lin(1, 2), lin(0, 1)).reshape(n, n, 3), atol=1e-4)  # right
    scene.release()


def test_reduce_hook(mi, torch_cuda):
    """mi355_set_reduce_hook (SURVEY 8(e) through the C ABI): the hook is called once per exchange with the right buffer sizes -- the RGBW
    film between accumulate and develop, PRB's weight image, the gradient buffer -- on the call's stream; its failure fails the call.
    The hook used here plays "two ranks that rendered the same shard": it doubles the buffer with a torch kernel on the stream the calls
    run on (the null stream), i.e. stream-ordered like ncclAllReduce would be.  The developed image is then unchanged (RGB and W both double), the weight image and the gradients double."""
    torch = torch_cuda
    scene = mi.load_dict(small_cbox(mi, 32))
    integ = mi.PRBIntegrator({"max_depth": 4})
    lib = mi.abi.lib(); h = scene.handle()
    W = H = 32
    calls = []
    bufs = {}
    def double(user, ptr, count, stream):                                   # stream-ordered, like an all-reduce kernel: a torch op on the
        calls.append(int(count))                                            # same (null) stream, on the tensor that owns `ptr`
        t = bufs.get(ptr)
        if t is None or t.numel() != int(count): return 1
        t *= 2
        return 0
    film = torch.zeros((H, W, 4), device="cuda"); img = torch.zeros((H, W, 3), device="cuda"); img2 = torch.zeros_like(img); film2 = torch.zeros_like(film)
    p = integ._params(scene, 3, 8)
    mi.abi.check(lib.mi355_render(h, C.byref(p), C.c_void_p(film.data_ptr()), C.c_void_p(img.data_ptr()), None))
    wgt = torch.zeros((H, W), device="cuda"); mi.abi.check(lib.mi355_prb_splat_weights(h, C.byref(p), C.c_void_p(wgt.data_ptr()), None))
    total = scene.info()["grad_floats"]
    gin = torch.ones((H, W, 3), device="cuda"); g = torch.zeros(total, device="cuda")
    mi.abi.check(lib.mi355_render_backward(h, C.byref(p), C.c_void_p(gin.data_ptr()), C.c_void_p(wgt.data_ptr()), C.c_void_p(g.data_ptr()), None))
    torch.cuda.synchronize()
    scene.set_reduce_hook(double)
    bufs[film2.data_ptr()] = film2
    mi.abi.check(lib.mi355_render(h, C.byref(p), C.c_void_p(film2.data_ptr()), C.c_void_p(img2.data_ptr()), None))
    wgt2 = torch.zeros((H, W), device="cuda"); bufs[wgt2.data_ptr()] = wgt2; mi.abi.check(lib.mi355_prb_splat_weights(h, C.byref(p), C.c_void_p(wgt2.data_ptr()), None))
    g2 = torch.zeros(total, device="cuda"); bufs[g2.data_ptr()] = g2
    mi.abi.check(lib.mi355_render_backward(h, C.byref(p), C.c_void_p(gin.data_ptr()), C.c_void_p(wgt.data_ptr()), C.c_void_p(g2.data_ptr()), None))
    torch.cuda.synchronize()
    assert calls == [W * H * 4, W * H, total]
    assert torch.allclose(film2, 2 * film, rtol=1e-5, atol=1e-6) and torch.allclose(img2, img, rtol=1e-5, atol=1e-6)
    assert torch.allclose(wgt2, 2 * wgt, rtol=1e-5) and float(g.abs().max()) > 0 and torch.allclose(g2, 2 * g, rtol=1e-4, atol=1e-6 * float(g.abs().max()))
    # mi355_render_accumulate is the raw shard and never exchanges; an empty shard still reaches the hook (it is a collective)
    n = len(calls)
    acc = torch.zeros_like(film); mi.abi.check(lib.mi355_render_accumulate(h, C.byref(p), C.c_void_p(acc.data_ptr()), None)); torch.cuda.synchronize()
    assert len(calls) == n and torch.allclose(acc, film, rtol=1e-5, atol=1e-6)
    pe = integ._params(scene, 3, 8, shard=(W * H, 0))
    mi.abi.check(lib.mi355_render(h, C.byref(pe), C.c_void_p(film2.data_ptr()), C.c_void_p(img2.data_ptr()), None)); torch.cuda.synchronize()
    assert len(calls) == n + 1 and float(film2.abs().max()) == 0.0
    # a failing hook fails the call
    scene.set_reduce_hook(lambda user, ptr, count, stream: 7)
    assert lib.mi355_render(h, C.byref(p), C.c_void_p(film2.data_ptr()), C.c_void_p(img2.data_ptr()), None) == mi.abi.ERR_DEVICE
    scene.set_reduce_hook(None)
    mi.abi.check(lib.mi355_render(h, C.byref(p), C.c_void_p(film2.data_ptr()), C.c_void_p(img2.data_ptr()), None)); torch.cuda.synchronize()
    assert torch.allclose(film2, film, rtol=1e-5, atol=1e-6)
    scene.release()


def test_adjoint_fixed_point_accumulators(mi, orc, torch_cuda):
    """The one-pass adjoint sums its small-parameter gradients in 64-bit fixed point per workgroup (exponent from the workgroup's largest
    |dL|, fp32 fallback for outliers and non-finite values).  The result must not depend on that: image adjoints spanning 13 decades,
    an all-zero adjoint, one with huge isolated pixels, and one with a NaN give what the oracle and the two-pass kernel (fp32 sums) give."""
    torch = torch_cuda
    d = small_cbox(mi, 32)
    tex = np.random.default_rng(1).uniform(0.2, 0.8, (8, 8, 3)).astype(np.float32)
    d["tex"] = {"type": "diffuse", "reflectance": {"type": "bitmap", "data": tex, "raw": True}}
    d["back"]["bsdf"] = {"type": "ref", "id": "tex"}
    scene = mi.load_dict(d); o = orc.OracleScene(scene, mi.abi)
    rng = np.random.default_rng(7)
    wide = (np.exp(rng.normal(0, 5, (32, 32, 3))) * rng.choice([-1.0, 1.0], (32, 32, 3))).astype(np.float32)
    spikes = rng.normal(size=(32, 32, 3)).astype(np.float32); spikes[5, 7] = 1e20; spikes[20, 3] = -3e25
    for name, gin in (("wide", wide), ("zero", np.zeros((32, 32, 3), np.float32)), ("spikes", spikes)):
        res = {}
        for two_pass in (False, True):
            integ = mi.PRBIntegrator({"max_depth": 5, "rr_depth": 3}); integ.two_pass = two_pass
            params = mi.traverse(scene)
            res[two_pass] = integ.render_backward(scene, params, torch.tensor(gin), seed=11, spp=16).cpu().numpy().astype(np.float64)
        ref = o.render_backward(o.params(mi.PRBIntegrator({"max_depth": 5, "rr_depth": 3}), 11, 16), gin).astype(np.float64)
        assert np.isfinite(res[False]).all()
        if name == "zero":
            assert (res[False] == 0).all() and (ref == 0).all()
            continue
        lay, _ = o.param_layout()
        for kind, index, off, size in lay:                               # per parameter: the spikes dominate the largest one only
            a, b, c = res[False][off:off + size], res[True][off:off + size], ref[off:off + size]
            if np.abs(c).max() == 0:
                assert np.abs(a).max() == 0
                continue
            assert grad_close(a, c, 1e-3, "%s kind %d index %d vs oracle" % (name, kind, index)) and grad_close(a, b, 1e-3, "%s kind %d index %d vs two-pass" % (name, kind, index))
    bad = rng.normal(size=(32, 32, 3)).astype(np.float32); bad[10, 10, 1] = np.nan
    integ = mi.PRBIntegrator({"max_depth": 5, "rr_depth": 3})
    g = integ.render_backward(scene, mi.traverse(scene), torch.tensor(bad), seed=11, spp=16).cpu().numpy()
    assert np.isnan(g).any()                                              # a NaN adjoint reaches the gradient, as it does through fp32 sums
    scene.release()
