"""GPU tests of the geometry row of SURVEY.md 8(f)-3 / 8(f)-4 -- BASELINE config 5's path:

  * `vertex_positions` / IOR as scene parameters: SceneParameters.update -> MI355_PARAM_VERTEX_POSITIONS / MI355_PARAM_BSDF_IOR -> device-side
    recompute of the vertex normals, triangle rebuild, BVH refit, flat-table rebuild (Mesh::parameters_changed mesh.cpp:96-163,
    Scene::parameters_changed scene.cpp:481-529), checked against a freshly built scene;
  * the reverse-mode geometry / IOR adjoint of the particle tracer (MI355_FLAG_GEOMETRY_GRADS, hand-written: csrc/mi355_geo_adjoint.h)
    against the oracle's independent forward-mode (multi-dual, double precision) derivative of the same paths (oracle/mi_oracle_geo.inc),
    against the engine's own forward mode (transpose identity) and against finite differences of the primal;
  * the fork's lens-design loop (notebooks/scripts/lens_design) on a doublet of three aspheric surface meshes.

Tolerances: 1e-3 of the largest gradient component (the north star's bar) -- measured agreement is 1e-6 .. 3e-5.  One exception is written
where it applies: with the TENT filter the derivative of the splat weights is piecewise constant, and a connection that lands within
float rounding of a pixel centre can take the other side of the kink on the device (fp32) than in the oracle (fp64); such an event
moves the handful of vertices of one triangle by ~1e-3 (seen once in seven seeds), so that case asserts on all but 5 % of the vertices.
"""
import os

import numpy as np
import pytest

from test_gpu_parity import _geo_floor_scene, _geo_lens_scene, _geo_mirror_scene, _icosphere

pytestmark = pytest.mark.gpu


def _cap_lens_scene(mi, res=32, subdiv=2, rfilter="gaussian"):
    """The notebooks' arrangement in small, built with the lens-design helpers: collimated `directionalarea` disk -> ONE refracting
    surface mesh (conic cap, computed vertex normals) with its baffle -> diffuse screen watched by a perspective camera."""
    ld = mi.lens_design; T = mi.ScalarTransform4f
    surf = ld.ConicSurface(radial_extent=0.5, c=1 / 1.2, K=0.0, z0=1.0)
    V, F, Va, Fa, _ = ld.create_surface_geometry(subdiv, 0.5, surf.compute_z_np, c=surf.get_curvature(), baffle_radius=0.7, flip_normals=True)
    return {'type': 'scene', 'integrator': {'type': 'ptracer', 'max_depth': 4},
            'white': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': (0.8, 0.7, 0.6)}},
            'black': {'type': 'diffuse', 'reflectance': 0.0},
            'screen': {'type': 'rectangle', 'to_world': T.look_at(target=[0, 0, 1], origin=[0, 0, 0], up=[0, 1, 0]).scale([2, 2, 1]), 'bsdf': {'type': 'ref', 'id': 'white'}},
            'lens': {'type': 'mesh', 'vertex_positions': V.astype(np.float32), 'faces': F.astype(np.uint32), 'compute_vertex_normals': True,
                     'bsdf': {'type': 'dielectric', 'int_ior': 1.5, 'ext_ior': 1.0, 'specular_reflectance': 0.0}},
            'baffle': {'type': 'mesh', 'vertex_positions': Va.astype(np.float32), 'faces': Fa.astype(np.uint32), 'compute_vertex_normals': True, 'bsdf': {'type': 'ref', 'id': 'black'}},
            'source': {'type': 'disk', 'to_world': T.look_at(target=[0, 0, 1.0], origin=[0, 0, 3.0], up=[0, 1, 0]) @ T.scale([0.45, 0.45, 1]),
                       'bsdf': {'type': 'ref', 'id': 'black'}, 'emitter': {'type': 'directionalarea', 'radiance': 1.0}},
            'sensor': {'type': 'perspective', 'near_clip': 0.01, 'far_clip': 10.0, 'fov': 60.0,
                       'to_world': T.look_at(target=[0.1, 0.05, 0.0], origin=[0.3, 0.2, 0.6], up=[0, 1, 0]),
                       'film': {'type': 'hdrfilm', 'width': res, 'height': res, 'pixel_format': 'rgb', 'rfilter': {'type': rfilter}}}}


def scene_layout(mi, scene):
    """(kind, index, offset, size) of every gradient slot (mi355_scene_param_layout)."""
    import importlib
    return mi.scene_param_layout(scene)[0]


def _shape_index(scene, sid):
    return [i for i, s in enumerate(scene.shapes) if s.id == sid][0]


def _disabled_blocks(mi, params):
    """(kind, index) of the geometry / IOR slots the geometry adjoint leaves alone: when some parameter of `params` asks for a gradient
    (requires_grad), dr.backward()'s semantics apply -- only those are visited (HipBackend.enable_grads)."""
    geo = [k for k, e in params._entries.items() if e[0] in (mi.abi.PARAM_VERTEX_POSITIONS, mi.abi.PARAM_BSDF_IOR)]
    wanted = {params._entries[k][:2] for k in geo if getattr(params[k], "requires_grad", False)}
    return set() if not wanted else {params._entries[k][:2] for k in geo} - wanted


def _compare(mi, g, g_ref, lay, tol=1e-3, frac_ok=1.0, disabled=()):
    """Every geometry / IOR block of the gradient buffer against the oracle's, relative to the block's largest reference component."""
    seen = 0
    for k, i, off, size in lay:
        if k not in (mi.abi.PARAM_VERTEX_POSITIONS, mi.abi.PARAM_BSDF_IOR):
            continue
        if (k, i) in disabled:
            assert np.abs(g[off:off + size]).max() == 0, (k, i)             # a slot without a gradient request stays untouched
            continue
        a, b = g[off:off + size].astype(np.float64), g_ref[off:off + size].astype(np.float64)
        assert np.isfinite(a).all()
        m = np.abs(b).max()
        if m == 0:
            assert np.abs(a).max() == 0, (k, i)
            continue
        seen += 1
        e = np.abs(a - b) / m
        if frac_ok >= 1.0:
            assert e.max() < tol, "kind %d index %d: max rel err %.3e" % (k, i, e.max())
        else:
            _log_partial(k, i, e, a, b, tol, frac_ok)
            assert (e < tol).mean() >= frac_ok and np.linalg.norm(a - b) / np.linalg.norm(b) < 5e-3, "kind %d index %d: %.3e, %.3f ok" % (k, i, e.max(), (e < tol).mean())
    return seen


def _log_partial(kind, index, e, a, b, tol, frac_ok):
    """The tent-filter cases are asserted on a FRACTION of the components (the filter's kink: fp32 on the device, fp64 in the oracle); which
    components miss the bar and by how much goes to gpurun_out/geometry_partial_parity.txt (committed as profiles/r03_tent_filter_vertices.txt)."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(out):
        return
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0]
    bad = np.nonzero(e >= tol)[0]
    with open(os.path.join(out, "geometry_partial_parity.txt"), "a") as f:
        f.write("%s  kind %d index %d: %d components, %d (%.2f %%) at or above %.0e of the block's largest reference component (asserted: >= %.0f %% below); "
                "max %.3e, |a - b|_2 / |b|_2 = %.3e\n" % (test, kind, index, e.size, bad.size, 100.0 * bad.size / e.size, tol, 100 * frac_ok, e.max(), np.linalg.norm(a - b) / np.linalg.norm(b)))
        for j in bad[np.argsort(-e[bad])][:12]:
            f.write("    vertex %6d axis %s: device % .6e  oracle % .6e  error / max|oracle| = %.3e\n" % (j // 3, "xyz"[j % 3], a[j], b[j], e[j]))


@pytest.mark.parametrize("case", ["cap-lens-gaussian", "cap-lens-tent", "cap-lens-box", "cap-lens-fixed-normals", "diffuse-floor", "glass-ball", "glass-ball-updated",
                                  "turning-mirror"])
def test_geometry_backward_matches_oracle(mi, orc, torch_cuda, case):
    """render_backward(geometry=True) of the particle tracer -- gradients of a random linear functional of the image with respect to
    EVERY vertex of every mesh and to the IORs -- against the oracle's forward-mode derivative of the same light paths."""
    torch = torch_cuda
    update = None; rr_depth = 5; max_depth = 4; spp = 16; seed = 2; res = 32; frac_ok = 1.0
    if case.startswith("cap-lens"):
        rf = case.split("-")[2] if case.split("-")[2] in ("gaussian", "tent", "box") else "gaussian"
        d = _cap_lens_scene(mi, res, rfilter=rf)       # (the tent case was asserted on 95 % of the components up to round 2; measured now: every component, max 6e-7 -- profiles/r03_tent_filter_vertices.txt)
        if case != "cap-lens-fixed-normals":      # after a vertex update the normals follow the positions (mesh.cpp:138-139); before it they are data
            update = ("lens", lambda p: np.stack([p[:, 0], p[:, 1], 1.0 + (p[:, 2] - 1.0) * 1.05], axis=1))
    elif case == "diffuse-floor":               # a diffuse quad mesh under an area light: a connection at every vertex, roulette on
        d = _geo_floor_scene(mi, 0.0, res); max_depth = 5; rr_depth = 2
    elif case.startswith("glass-ball"):         # refraction in and out of a closed icosphere mesh with file normals
        d = _geo_lens_scene(mi, 1.0, "glass", res); max_depth = 5
        if case.endswith("updated"):
            update = ("lens", lambda p: (p - np.array([-0.3, 0, 0], dtype=np.float32)) * np.array([1.0, 1.04, 0.97], dtype=np.float32) + np.array([-0.3, 0, 0], dtype=np.float32))
    else:                                       # gold mirror mesh: the bsdf weight is a Fresnel term with a slope in cos(theta_i)
        d = _geo_mirror_scene(mi, 0.05, res); max_depth = 4
    scene = mi.load_dict(d)
    integ = mi.ParticleTracerIntegrator({"max_depth": max_depth, "rr_depth": rr_depth})
    params = mi.traverse(scene)
    o = orc.OracleScene(scene, mi.abi)
    if update is not None:
        key = "%s.vertex_positions" % update[0]
        new = update[1](params[key].cpu().numpy().reshape(-1, 3)).astype(np.float32)
        params[key] = torch.tensor(new.reshape(-1), device="cuda"); params.update()
        o.set_param(mi.abi.PARAM_VERTEX_POSITIONS, _shape_index(scene, update[0]), new.reshape(-1))
    p = o.params(integ, seed, spp)
    img = integ.render(scene, seed=seed, spp=spp).cpu().numpy(); ref, _ = o.ptracer_render(p)
    assert ref.max() > 0 and np.abs(img - ref).max() < 1e-4 * ref.max()
    gin = np.random.default_rng(5).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=seed, spp=spp, geometry=True).cpu().numpy()
    p.flags |= mi.abi.FLAG_GEOMETRY_GRADS
    g_ref = o.ptracer_backward(p, gin)
    lay, _ = o.param_layout()
    assert _compare(mi, g, g_ref, lay, 1e-3, frac_ok) >= 1
    # the colour parameters come out of the same call, unchanged by the geometry pass
    for k, i, off, size in lay:
        if k in (mi.abi.PARAM_BSDF_REFLECTANCE, mi.abi.PARAM_EMITTER_RADIANCE):
            m = max(np.abs(g_ref[off:off + size]).max(), 1e-30)
            assert np.abs(g[off:off + size] - g_ref[off:off + size]).max() < 2e-3 * m + 1e-6


def test_geometry_backward_is_the_transpose_of_the_forward_mode(mi, torch_cuda):
    """<g, J v> = <J^T g, v>: the gradient image of the forward design mode (mi355_ptracer_forward_design) contracted with the image adjoint
    equals the reverse-mode gradient contracted with the tangent -- vertex tangent (with the normals following the positions) and IOR tangent."""
    torch = torch_cuda
    res, spp, seed = 32, 64, 3
    scene = mi.load_dict(_cap_lens_scene(mi, res))
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5})
    params = mi.traverse(scene)
    pos = params["lens.vertex_positions"].reshape(-1, 3).clone(); pos[:, 2] = 1.0 + (pos[:, 2] - 1.0) * 1.05
    params["lens.vertex_positions"] = pos.reshape(-1); params.update()
    rng = np.random.default_rng(11)
    gin = torch.tensor(rng.uniform(0.5, 1.5, (res, res, 3)).astype(np.float32), device="cuda")
    vt = torch.tensor(rng.normal(size=tuple(pos.shape)).astype(np.float32) * 0.1, device="cuda")
    g = integ.render_backward(scene, params, gin, seed=seed, spp=spp, geometry=True)
    gv = params.grad("lens.vertex_positions").reshape(-1, 3)
    jv = integ.render_forward_design(scene, shape="lens", vertex_tangent=vt, seed=seed, spp=spp)
    lhs, rhs = float((gin * jv).sum()), float((gv * vt).sum())
    assert abs(lhs - rhs) < 1e-3 * max(abs(lhs), abs(rhs)), (lhs, rhs)
    ji = integ.render_forward_design(scene, bsdf="lens.bsdf", d_int_ior=1.0, d_ext_ior=-0.5, seed=seed, spp=spp)
    lhs = float((gin * ji).sum()); rhs = float(params.grad("lens.bsdf.int_ior")[0] - 0.5 * params.grad("lens.bsdf.ext_ior")[0])
    assert abs(lhs - rhs) < 1e-3 * max(abs(lhs), abs(rhs)), (lhs, rhs)
    assert g.numel() == params._total
    # geometry / IOR gradients only (MI355_FLAG_SKIP_COLOUR_GRADS, what a lens-design step asks for): the same geometry slots, colour slots untouched
    full = g.clone()
    params.zero_grad()
    only = integ.render_backward(scene, params, gin, seed=seed, spp=spp, geometry=True, colour=False)
    lay = scene_layout(mi, scene)
    colour_full = 0.0
    for kind, index, off, size in lay:
        if kind in (mi.abi.PARAM_VERTEX_POSITIONS, mi.abi.PARAM_BSDF_IOR):
            assert torch.allclose(only[off:off + size], full[off:off + size], rtol=1e-4, atol=1e-6 * float(full[off:off + size].abs().max() + 1e-30))
        else:
            assert float(only[off:off + size].abs().max()) == 0.0
            colour_full = max(colour_full, float(full[off:off + size].abs().max()))
    assert colour_full > 0.0


@pytest.mark.parametrize("integrator_name", ["path", "ptracer"])
def test_vertex_update_matches_a_fresh_scene(mi, orc, torch_cuda, integrator_name):
    """SceneParameters.update of `vertex_positions` (refit + normals on the device) renders the same image as a scene built from the moved
    vertices: a BVH mesh with vertex normals (icosphere, 1280 triangles), a box (`cube`, one slab record) and a small flat mesh."""
    torch = torch_cuda
    T = mi.ScalarTransform4f
    v, f = _icosphere(3)

    def make(scale, lift, shear):
        d = mi.cornell_box(); d["sensor"]["film"]["width"] = 48; d["sensor"]["film"]["height"] = 48
        ball = (v * 0.3 * np.array([1.0, scale, 1.0])).astype(np.float32) + np.array([-0.35, 0.35, 0.3], dtype=np.float32)
        d["ball"] = {"type": "mesh", "vertex_positions": ball, "faces": f.astype(np.uint32), "compute_vertex_normals": True, "bsdf": {"type": "ref", "id": "white"}}
        quad = np.array([[0.2, -0.6 + lift, 0.2], [0.8, -0.6 + lift, 0.2], [0.8, -0.6 + lift + shear, 0.8], [0.2, -0.6 + lift, 0.8]], dtype=np.float32)
        d["shelf"] = {"type": "mesh", "vertex_positions": quad, "faces": np.array([[0, 1, 2], [0, 2, 3]], dtype=np.uint32), "bsdf": {"type": "ref", "id": "green"}}
        return d
    scene = mi.load_dict(make(1.0, 0.0, 0.0)); fresh = mi.load_dict(make(1.25, 0.1, 0.15))
    integ = mi.PathIntegrator({"max_depth": 4}) if integrator_name == "path" else mi.ParticleTracerIntegrator({"max_depth": 4})
    before = integ.render(scene, seed=0, spp=32).cpu().numpy()
    info0 = scene.info()
    params = mi.traverse(scene)
    for sid in ("ball", "shelf"):
        params["%s.vertex_positions" % sid] = torch.tensor(fresh.shapes[_shape_index(fresh, sid)].vertex_positions.reshape(-1), device="cuda")
    # the `cube` is ONE slab record; sheared, it stops being a parallelepiped... keep it a box: move it rigidly
    sb = _shape_index(scene, "small-box")
    moved = scene.shapes[sb].vertex_positions + np.array([0.0, 0.15, 0.0], dtype=np.float32)
    params["small-box.vertex_positions"] = torch.tensor(moved.reshape(-1), device="cuda")
    assert sorted(params.update()) == ["ball.vertex_positions", "shelf.vertex_positions", "small-box.vertex_positions"]
    fresh.shapes[_shape_index(fresh, "small-box")].vertex_positions = np.ascontiguousarray(moved)
    a = integ.render(scene, seed=0, spp=32).cpu().numpy(); b = integ.render(fresh, seed=0, spp=32).cpu().numpy()
    assert np.abs(a - before).max() > 1e-2 * before.max()                  # the update did something
    assert np.abs(a - b).max() < 1e-4 * b.max(), np.abs(a - b).max() / b.max()
    assert scene.info()["bvh_nodes"] == info0["bvh_nodes"] and scene.info()["boxes"] == 2      # topology kept, both cubes still one slab record each
    # ... and against the oracle twin updated the same way (its normals recomputed by its own restatement of mesh.cpp:376-413)
    o = orc.OracleScene(fresh, mi.abi)
    ref = (o.render(o.params(integ, 0, 32))[0] if integrator_name == "path" else o.ptracer_render(o.params(integ, 0, 32))[0])
    assert np.abs(a - ref).max() < 1e-3 * ref.max()
    # shearing the box turns it into twelve flat triangles (the table is rebuilt), an IOR update reaches the dielectric
    sheared = moved.copy(); sheared[moved[:, 1] > moved[:, 1].mean(), 0] += 0.1 * (moved[moved[:, 1] > moved[:, 1].mean(), 2] - 0.3)
    params["small-box.vertex_positions"] = torch.tensor(sheared.reshape(-1), device="cuda"); params.update()
    c = integ.render(scene, seed=0, spp=32).cpu().numpy()
    assert scene.info()["boxes"] == 1 and np.isfinite(c).all() and np.abs(c - a).max() > 0
    with pytest.raises(mi.Mi355Error):                                      # the vertex count is fixed
        mi.abi.check(mi.abi.lib().mi355_scene_set_param(scene.handle(), mi.abi.PARAM_VERTEX_POSITIONS, sb, moved.ctypes.data_as(mi.abi.f32p), 9))


def test_ior_update_and_finite_differences(mi, orc, torch_cuda):
    """MI355_PARAM_BSDF_IOR: the update changes the render like a rebuilt scene, and the IOR and vertex gradients equal central finite
    differences of the primal with identical seeds (the reference's protocol for gradients, test_ad_integrators.py:742-786).
    The finite-difference scene is a tilted FLAT interface (two triangles, no vertex normals) and the image functional is smooth and
    vanishes towards the frame, because two things are differentiated neither by AD nor by this pass: splats that cross the border of
    the film, and -- on a curved mesh with interpolated normals -- the jumps of the shading-normal correction at the facet edges (its
    in-facet slope IS differentiated; with 124 facets on the cap the two differ by 12 %)."""
    torch = torch_cuda
    res, spp = 32, 256
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5})
    scene = mi.load_dict(_cap_lens_scene(mi, res, rfilter="gaussian"))
    params = mi.traverse(scene)
    params["lens.bsdf.int_ior"] = torch.tensor([1.6], device="cuda"); params.update()
    d16 = _cap_lens_scene(mi, res, rfilter="gaussian"); d16["lens"]["bsdf"]["int_ior"] = 1.6
    a, b = integ.render(scene, seed=4, spp=spp), integ.render(mi.load_dict(d16), seed=4, spp=spp)
    assert float((a - b).abs().max()) < 1e-5 * float(b.max())
    # ---- a wedge: the beam enters a tilted plane interface, the refracted spot moves with the IOR and with the tilt
    T = mi.ScalarTransform4f

    def wedge(ior, tilt):
        d = _cap_lens_scene(mi, res, rfilter="gaussian"); del d["baffle"]
        q = np.array([[-0.8, -0.8, 1.0 + 0.8 * tilt], [0.8, -0.8, 1.0 - 0.8 * tilt], [0.8, 0.8, 1.0 - 0.8 * tilt], [-0.8, 0.8, 1.0 + 0.8 * tilt]], dtype=np.float32)
        d["lens"] = {"type": "mesh", "vertex_positions": q, "faces": np.array([[0, 2, 1], [0, 3, 2]], dtype=np.uint32),
                     "bsdf": {"type": "dielectric", "int_ior": ior, "ext_ior": 1.0, "specular_reflectance": 0.0}}
        d["source"]["to_world"] = T.look_at(target=[0, 0, 1.0], origin=[0, 0, 3.0], up=[0, 1, 0]) @ T.scale([0.2, 0.2, 1])
        return d
    yy, xx = np.mgrid[0:res, 0:res]
    window = (np.sin(np.pi * (xx + 0.5) / res) * np.sin(np.pi * (yy + 0.5) / res)) ** 2
    gin = torch.tensor(np.repeat(((0.5 + xx / res + 0.3 * yy / res) * window)[:, :, None], 3, axis=2).astype(np.float32), device="cuda")
    ior0, tilt0 = 1.6, 0.35
    spp = 2048      # (the finite difference sees the lobe choice flip for a few hundred samples; the weight's Fresnel derivative is its expectation)
    sw = mi.load_dict(wedge(ior0, tilt0)); pw = mi.traverse(sw)
    integ.render_backward(sw, pw, gin, seed=4, spp=spp, geometry=True)
    f = lambda ior, tilt: float((integ.render(mi.load_dict(wedge(ior, tilt)), seed=4, spp=spp) * gin).sum())
    eps = 4e-3
    g_ior, fd_ior = float(pw.grad("lens.bsdf.int_ior")[0]), (f(ior0 + eps, tilt0) - f(ior0 - eps, tilt0)) / (2 * eps)
    dq = torch.tensor(np.array([[0, 0, 0.8], [0, 0, -0.8], [0, 0, -0.8], [0, 0, 0.8]], dtype=np.float32), device="cuda")      # d(vertex) / d(tilt)
    g_tilt, fd_tilt = float((pw.grad("lens.vertex_positions").reshape(-1, 3) * dq).sum()), (f(ior0, tilt0 + eps) - f(ior0, tilt0 - eps)) / (2 * eps)
    assert abs(fd_ior) > 0 and abs(g_ior - fd_ior) < 3e-2 * abs(fd_ior), (g_ior, fd_ior)
    assert abs(fd_tilt) > 0 and abs(g_tilt - fd_tilt) < 3e-2 * abs(fd_tilt), (g_tilt, fd_tilt)


def test_lens_design_doublet(mi, orc, torch_cuda):
    """BASELINE config 5 in the notebook's own terms (notebooks/doublet_full_optim.ipynb, cells 5-10): the doublet of three even-aspheric
    surface MESHES (icosphere level 5: 3 x 10.4 k triangles + baffles), `dispersive` interfaces, a collimated `directionalarea` disk, the
    `ptracer` integrator, a perspective sensor in front of the white film plane, RMS-spot loss, mi.Adam over the 30 surface scalars.
      (i)   render_backward over ALL vertices equals the oracle's gradient (small film / spp: the oracle is a CPU program);
      (ii)  contracted with d(vertex)/d(scalar) it equals the forward design mode, scalar by scalar;
      (iii) the Adam loop lowers the spot loss through params.update() -- the device scene is never re-created."""
    torch = torch_cuda
    ld = mi.lens_design
    surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
    mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
    system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=5)
    system.disable_all_materials()
    problem = ld.DesignProblem(system, resolution=(32, 32), spp=32, learning_rate=2e-3, iters=6, init_camera_z=0.1, fov=100.0).prepare()
    scene, params = problem.scene, problem.params
    info = scene.info()
    assert info["bvh_meshes"] == 6 and info["triangles"] >= 3 * 8000
    assert len(problem.optimizer.keys()) == 30 and len(scene.sensors()) == 1 and problem.num_sources == 1      # (the 3-field problem: tests/test_gpu_sensors.py)
    # ---- (i) after one update (normals follow the positions), against the oracle
    system.update(params, problem.optimizer)
    o = orc.OracleScene(scene, mi.abi)
    for e in system.elements:
        for key in (e.lens_key, e.baffle_key):
            o.set_param(mi.abi.PARAM_VERTEX_POSITIONS, _shape_index(scene, key), params["%s.vertex_positions" % key].detach().cpu().numpy())
    integ = scene.integrator()
    spp = 16
    img = integ.render(scene, seed=0, spp=spp).cpu().numpy(); p = o.params(integ, 0, spp); ref, _ = o.ptracer_render(p)
    assert ref.max() > 0 and np.abs(img - ref).max() < 2e-4 * ref.max()
    yy, xx = np.mgrid[0:32, 0:32]
    gin = np.repeat((((xx - 15.5) ** 2 + (yy - 15.5) ** 2) / 256.0)[:, :, None], 3, axis=2).astype(np.float32)       # the spot loss's own shape: r^2
    params.zero_grad()
    g = integ.render_backward(scene, params, torch.tensor(gin), seed=0, spp=spp, geometry=True).cpu().numpy()
    p.flags |= mi.abi.FLAG_GEOMETRY_GRADS
    g_ref = o.ptracer_backward(p, gin)
    lay, _ = o.param_layout()
    off_blocks = _disabled_blocks(mi, params)
    assert (mi.abi.PARAM_VERTEX_POSITIONS, _shape_index(scene, "film_plane")) in off_blocks          # the film plane's own vertices are not optimised
    assert _compare(mi, g, g_ref, lay, 1e-3, 0.98, off_blocks) >= 3        # (the three lens surfaces; tent filter: see the module docstring)
    # ---- (ii) reverse mode . d(vertex)/d(scalar) == forward mode, for curvature, conic constant, position and a polynomial term
    e0 = system.elements[1]
    for name in ("c", "K", "z0", "a4", "a6"):
        var = problem.optimizer["lens1_%s" % name].detach().clone().requires_grad_(True)
        keep = e0.surface.params[name]; e0.surface.params[name] = var
        x0, y0 = e0.initial_lens_vertices[:, 0], e0.initial_lens_vertices[:, 1]
        z = e0.surface.compute_z(x0, y0)
        tangent = torch.autograd.grad(z.sum(), var, retain_graph=True)[0]   # (sanity: the scalar does move the surface)
        assert float(tangent.abs().sum()) > 0
        dz = torch.autograd.functional.jvp(lambda s: _surface_z(e0, name, s, x0, y0), var, torch.ones_like(var))[1]
        e0.surface.params[name] = keep
        vt = torch.stack([torch.zeros_like(dz), torch.zeros_like(dz), dz], dim=1).contiguous()
        jv = integ.render_forward_design(scene, shape=e0.lens_key, vertex_tangent=vt, seed=0, spp=spp)
        lhs = float((torch.tensor(gin, device="cuda") * jv).sum())
        rhs = float((params.grad("%s.vertex_positions" % e0.lens_key).reshape(-1, 3) * vt).sum())
        assert abs(lhs - rhs) < 2e-3 * max(abs(lhs), abs(rhs), 1e-12), (name, lhs, rhs)
    # ---- (iii) the optimisation loop of design_problem.py:508-565
    handle = scene._handle
    losses = [problem.step(it)[0] for it in range(6)]
    assert scene._handle is handle                                          # the device replica was updated in place, never rebuilt
    assert np.isfinite(losses).all() and min(losses[3:]) < losses[0], losses
    moved = [float((problem.optimizer[k].detach() - v0).abs().max()) for k, v0 in [("lens1_c", torch.tensor([-8.0 / 24], device="cuda")), ("lens2_c", torch.tensor([8.0 / 24], device="cuda"))]]
    assert max(moved) > 0


def _surface_z(element, name, value, x0, y0):
    keep = element.surface.params[name]; element.surface.params[name] = value
    try:
        return element.surface.compute_z(x0, y0)
    finally:
        element.surface.params[name] = keep


def test_parameter_updates_are_validated(mi, torch_cuda):
    """Error behaviour of the geometry / IOR parameters at the C ABI (the counterpart of Mesh::parameters_changed refusing a changed vertex
    count, mesh.cpp:96-104): wrong sizes, wrong kinds of target, non-positive indices and non-finite positions fail with MI355_ERR_INVALID
    and leave the scene as it was."""
    import ctypes as C
    torch = torch_cuda
    scene = mi.load_dict(_cap_lens_scene(mi, 32)); lib = mi.abi.lib(); h = scene.handle()
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5})
    before = integ.render(scene, seed=2, spp=16).clone()
    lens = _shape_index(scene, "lens"); screen = [i for i, sh in enumerate(scene.shapes) if sh.type != mi.abi.SHAPE_MESH][0]
    nv = scene.shapes[lens].vertex_positions.shape[0]
    good = np.ascontiguousarray(scene.shapes[lens].vertex_positions, dtype=np.float32).reshape(-1)
    f32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    E = mi.abi.ERR_INVALID
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_VERTEX_POSITIONS, lens, f32(good[:-3].copy()), 3 * (nv - 1)) == E       # vertex count changed
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_VERTEX_POSITIONS, screen, f32(good), good.size) == E                    # not a mesh
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_VERTEX_POSITIONS, 10 ** 6, f32(good), good.size) == E                   # no such shape
    bad = good.copy(); bad[7] = np.nan
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_VERTEX_POSITIONS, lens, f32(bad), bad.size) == E                        # non-finite position
    glass = [i for i, b in enumerate(scene.bsdfs) if b.type == mi.abi.BSDF_DIELECTRIC][0]
    other = [i for i, b in enumerate(scene.bsdfs) if b.type != mi.abi.BSDF_DIELECTRIC][0]
    two = np.array([1.5, 1.0], dtype=np.float32)
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_BSDF_IOR, other, f32(two), 2) == E                                      # not a dielectric
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_BSDF_IOR, glass, f32(two), 3) == E                                      # wrong size
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_BSDF_IOR, glass, f32(np.array([0.0, 1.0], dtype=np.float32)), 2) == E   # non-positive index
    assert b"" != lib.mi355_last_error()
    after = integ.render(scene, seed=2, spp=16)
    assert torch.allclose(after, before, rtol=1e-5, atol=1e-7)                                                               # nothing was applied (float-atomic order aside)
    assert lib.mi355_scene_set_param(h, mi.abi.PARAM_BSDF_IOR, glass, f32(two), 2) == 0 and lib.mi355_scene_set_param(h, mi.abi.PARAM_VERTEX_POSITIONS, lens, f32(good), good.size) == 0
    scene.release()


def test_geometry_backward_shards_add_up(mi, torch_cuda):
    """Multi-GPU decomposition of BASELINE configs[4]: the light-sample grid cut into 16-row tiles dealt to 3 ranks (ragged: 40 rows),
    into pixel bands and into spp slices -- the geometry / IOR gradients of the shards add up to the gradient of the whole job (global
    lane ids: the same light paths whoever traces them)."""
    import ctypes as C
    torch = torch_cuda
    res, spp, seed = 40, 32, 5
    scene = mi.load_dict(_cap_lens_scene(mi, res))
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5})
    params = mi.traverse(scene)
    gin = torch.tensor(np.random.default_rng(2).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32), device="cuda")
    whole = integ.render_backward(scene, params, gin, seed=seed, spp=spp, geometry=True, colour=True).clone()
    assert float(whole.abs().max()) > 0
    lib = mi.abi.lib(); h = scene.handle()
    fw, fh = integ.lane_grid(scene.sensor.film)
    decompositions = {"tiles": [dict(shard=mi.shard_tiles(fw, fh, r, 3)) for r in range(3)],
                      "bands": [dict(shard=mi.shard_pixels(fw * fh, r, 4)) for r in range(4)],
                      "spp": [dict(spp_shard=(s0, 8)) for s0 in range(0, spp, 8)]}
    for name, parts in decompositions.items():
        acc = torch.zeros_like(whole)
        for kw in parts:
            p = integ._params(scene, seed, spp, **kw); p.flags |= mi.abi.FLAG_GEOMETRY_GRADS
            mi.abi.check(lib.mi355_render_backward(h, C.byref(p), C.c_void_p(gin.data_ptr()), None, C.c_void_p(acc.data_ptr()), None))
        torch.cuda.synchronize()
        err = float((acc - whole).abs().max()) / float(whole.abs().max())
        assert err < 2e-4, (name, err)
    # the path tape bounds max_depth: an unbounded light path cannot be differentiated this way
    p = mi.ParticleTracerIntegrator({"max_depth": -1, "rr_depth": 5})._params(scene, seed, spp); p.flags |= mi.abi.FLAG_GEOMETRY_GRADS
    acc = torch.zeros_like(whole)
    assert lib.mi355_render_backward(h, C.byref(p), C.c_void_p(gin.data_ptr()), None, C.c_void_p(acc.data_ptr()), None) == mi.abi.ERR_UNSUPPORTED
    scene.release()


def test_geometry_adjoint_with_samples_per_pass(mi, torch_cuda):
    """`samples_per_pass` (AdjointIntegrator::render, integrator.cpp:700-726): one seeding, every stream continued over the passes.  The
    geometry adjoint walks the passes of a stream in one launch and re-uses its tape slot pass after pass; the transpose identity with the
    forward mode (which walks the same passes) must hold, and the gradient must differ from the one-pass gradient (different streams)."""
    torch = torch_cuda
    res, spp, seed = 32, 64, 7
    scene = mi.load_dict(_cap_lens_scene(mi, res))
    params = mi.traverse(scene)
    pos = params["lens.vertex_positions"].reshape(-1, 3).clone(); pos[:, 2] = 1.0 + (pos[:, 2] - 1.0) * 1.03
    params["lens.vertex_positions"] = pos.reshape(-1); params.update()
    rng = np.random.default_rng(3)
    gin = torch.tensor(rng.uniform(0.5, 1.5, (res, res, 3)).astype(np.float32), device="cuda")
    vt = torch.tensor(rng.normal(size=tuple(pos.shape)).astype(np.float32) * 0.1, device="cuda")
    grads = {}
    for per in (-1, 16):
        integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 5, "samples_per_pass": per})
        params.zero_grad()
        integ.render_backward(scene, params, gin, seed=seed, spp=spp, geometry=True)
        gv = params.grad("lens.vertex_positions").reshape(-1, 3).clone()
        jv = integ.render_forward_design(scene, shape="lens", vertex_tangent=vt, seed=seed, spp=spp)
        lhs, rhs = float((gin * jv).sum()), float((gv * vt).sum())
        assert abs(lhs - rhs) < 1e-3 * max(abs(lhs), abs(rhs)), (per, lhs, rhs)
        grads[per] = gv
    assert float((grads[-1] - grads[16]).abs().max()) > 1e-3 * float(grads[-1].abs().max())       # 4 passes of 16 are other streams than 1 pass of 64
    rel = float((grads[-1] - grads[16]).norm() / grads[-1].norm())
    assert rel < 0.5, rel                                                                          # ... of the same estimator
    scene.release()


def test_device_vertex_normals_follow_the_references_weighting_scheme(mi, orc, kats, torch_cuda):
    """src/render/tests/test_mesh.py:74-98 (test04_normal_weighting_scheme) on the DEVICE: after a `vertex_positions` update k_vertex_normals
    recomputes the normals (Mesh::parameters_changed -> recompute_vertex_normals, mesh.cpp:138-139, :376-413); the shading normal of a hit
    next to each of the five vertices is the reference's angle-weighted normal (rtol 5e-4 as there)."""
    import ctypes as C, math
    torch = torch_cuda
    K = kats["mesh_normal_weighting"]
    pos = np.array(K["vertex_positions"], dtype=np.float32); faces = np.array(K["faces"], dtype=np.uint32)
    d = {'type': 'scene', 'integrator': {'type': 'path', 'max_depth': 2},
         'm': {'type': 'mesh', 'vertex_positions': pos + np.float32(0.25), 'faces': faces, 'compute_vertex_normals': True, 'bsdf': {'type': 'diffuse'}},
         'sensor': {'type': 'perspective', 'film': {'type': 'hdrfilm', 'width': 4, 'height': 4}}}
    scene = mi.load_dict(d); params = mi.traverse(scene)
    params['m.vertex_positions'] = torch.tensor(pos.reshape(-1), device="cuda"); params.update()      # (loaded displaced: the update is what is tested)
    n0, n1 = np.array(K["n0"], dtype=np.float64), np.array(K["n1"], dtype=np.float64)
    n2 = n0 * (math.pi / 2) + n1 * math.acos(3.0 / 5.0); n2 /= np.linalg.norm(n2)
    want = np.stack([n2, n0, n0, n1, n1])
    rays = np.zeros((5, 8), dtype=np.float32)
    for v in range(5):
        f = 0 if v in (0, 1, 2) else 1
        centroid = pos[faces[f]].mean(axis=0); target = pos[v] + 2e-5 * (centroid - pos[v])
        nf = n0 if f == 0 else n1
        rays[v, :3] = target + nf; rays[v, 3:6] = -nf; rays[v, 6] = 10.0
    out = torch.empty((5, 24), device="cuda"); d_rays = torch.tensor(rays, device="cuda")
    mi.abi.check(mi.abi.lib().mi355_debug_surface_interaction(scene.handle(), C.c_void_p(d_rays.data_ptr()), 5, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize(); si = out.cpu().numpy()
    assert (si[:, 15].view(np.uint32) == 0).all()                          # every ray hit the mesh
    assert np.allclose(si[:, 8:11], want, rtol=K["rtol"], atol=1e-4), si[:, 8:11]
    o = orc.OracleScene(scene, mi.abi); o.set_param(mi.abi.PARAM_VERTEX_POSITIONS, 0, pos.reshape(-1))
    _, si_ref = o.ray_intersect(rays, with_si=True)
    assert np.allclose(si[:, 8:11], si_ref[:, 8:11], atol=2e-6)
    scene.release()


def test_lane_sort_changes_nothing_but_the_order(mi, torch_cuda):
    """pt_sort_lanes: listing the light samples cell by cell of their emitter's sample domain only changes WHICH hardware lane walks which
    sample -- film, colour gradients and geometry gradients equal the natural order's up to the order of the float atomics, for one pass
    and for samples_per_pass, for a collimated source (2D cells) and an area light (4D cells)."""
    torch = torch_cuda
    res, spp, seed = 32, 64, 3
    for name, d in (("collimated", _cap_lens_scene(mi, res)), ("area", _geo_lens_scene(mi, 1.0, "glass", res))):
        scene = mi.load_dict(d)
        assert scene.info()["bvh_meshes"] >= 1
        gin = torch.tensor(np.random.default_rng(4).uniform(0.5, 1.5, (res, res, 3)).astype(np.float32), device="cuda")
        out = {}
        for per in (-1, 16):
            for sort in (True, False):
                integ = mi.ParticleTracerIntegrator({"max_depth": 5, "rr_depth": 5, "samples_per_pass": per}); integ.no_lane_sort = not sort
                params = mi.traverse(scene)
                img = integ.render(scene, seed=seed, spp=spp).clone()
                g = integ.render_backward(scene, params, gin, seed=seed + 1, spp=spp, geometry=True, colour=True).clone()
                out[(per, sort)] = (img, g)
            (ia, ga), (ib, gb) = out[(per, True)], out[(per, False)]
            assert float(ia.max()) > 0 and float((ia - ib).abs().max()) < 1e-5 * float(ib.max()), (name, per)
            assert float((ga - gb).abs().max()) < 2e-4 * float(gb.abs().max()), (name, per, float((ga - gb).abs().max()) / float(gb.abs().max()))
        scene.release()


def test_lens_sag_kernels_match_the_torch_expression(mi, torch_cuda):
    """mi355_lens_sag / mi355_lens_sag_backward -- EvenAsphericSurface.compute_z_dr of the fork's scripts (lens.py:166-249) and its transpose
    as two kernels -- against the same expression in torch (surface.compute_z) and torch's autograd: positions to 2e-6 of the element
    radius, parameter gradients to 2e-5 of their largest magnitude, for a lens mesh (own x, y) and a baffle (rim point)."""
    torch = torch_cuda
    ld = mi.lens_design
    surf = ld.EvenAsphericSurface(radial_extent=8.0, c=-1 / 24, K=-0.3, z0=49.0, a4=2e-5, a6=-3e-7, a8=1e-9, a16=1e-18)
    V, F, Va, Fa, _ = ld.create_surface_geometry(4, 8.0, surf.compute_z_np, c=surf.get_curvature(), baffle_radius=8.8, flip_normals=True)
    base = torch.tensor(V, dtype=torch.float32, device="cuda"); base_b = torch.tensor(Va, dtype=torch.float32, device="cuda")
    rim_xy = torch.tensor([8.0, 0.0], device="cuda").repeat(base_b.shape[0], 1)
    w = torch.tensor(np.random.default_rng(0).normal(size=V.shape).astype(np.float32), device="cuda"); wb = torch.tensor(np.random.default_rng(1).normal(size=Va.shape).astype(np.float32), device="cuda")
    keys = list(surf.params.keys())
    results = {}
    for use_kernel in (True, False):
        leaves = {k: torch.tensor([float(surf.params[k])], dtype=torch.float32, device="cuda", requires_grad=True) for k in keys}
        saved = dict(surf.params); surf.params.update(leaves)
        if use_kernel:
            pos = ld.surface_positions(surf, base); pos_b = ld.surface_positions(surf, base_b, rim_xy)
        else:
            pos = torch.stack([base[:, 0], base[:, 1], surf.compute_z(base[:, 0], base[:, 1])], dim=1)
            pos_b = torch.stack([base_b[:, 0], base_b[:, 1], surf.compute_z(rim_xy[:1, 0], rim_xy[:1, 1]).expand(base_b.shape[0])], dim=1)
        ((pos * w).sum() + (pos_b * wb).sum()).backward()
        results[use_kernel] = (pos.detach(), pos_b.detach(), torch.cat([leaves[k].grad for k in keys]))
        surf.params.clear(); surf.params.update(saved)
    (pa, pba, ga), (pb, pbb, gb) = results[True], results[False]
    assert float((pa - pb).abs().max()) < 2e-6 * 8.0 and float((pba - pbb).abs().max()) < 2e-6 * 8.0
    assert torch.equal(pa[:, :2], base[:, :2]) and torch.equal(pba[:, :2], base_b[:, :2])
    rel = (ga - gb).abs() / gb.abs().max()
    assert float(gb.abs().max()) > 0 and float(rel.max()) < 2e-5, (ga, gb)
    for k, a_, b_ in zip(keys, ga.tolist(), gb.tolist()):                      # every scalar on its own scale too
        assert abs(a_ - b_) <= 1e-3 * abs(b_) + 1e-6 * float(gb.abs().max()), (k, a_, b_)


@pytest.mark.gpu
def test_spot_loss_kernel_matches_the_torch_expression(mi, torch_cuda):
    """mi355_spot_loss / _backward (losses.py:31-41, rms_loss_and_center) against the same expression written with torch operations on a CPU
    copy in float64: loss, centroid and the image gradient, for a blurred off-centre spot with a coloured background."""
    torch = torch_cuda
    ld = mi.lens_design
    rng = np.random.default_rng(5)
    H, W = 48, 64
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    img = (np.exp(-((ii - 30.3) ** 2 / 18 + (jj - 17.8) ** 2 / 40))[:, :, None] * np.array([1.0, 0.7, 0.4]) + 0.01 * rng.random((H, W, 3))).astype(np.float32)
    a = torch.tensor(img, device="cuda", requires_grad=True)
    la, ia, ja = ld.rms_loss_and_center(a)
    (3.0 * la).backward()
    b = torch.tensor(img.astype(np.float64), requires_grad=True)                # the CPU branch of the same function = the torch expression
    lb, ib, jb = ld.rms_loss_and_center(b)
    (3.0 * lb).backward()
    assert abs(float(la) - float(lb)) < 1e-5 * float(lb) and abs(float(ia) - float(ib)) < 1e-4 and abs(float(ja) - float(jb)) < 1e-4
    ga, gb = a.grad.cpu().double(), b.grad
    assert float((ga - gb).abs().max()) < 1e-4 * float(gb.abs().max())
    assert not ia.requires_grad and not ja.requires_grad


@pytest.mark.gpu
def test_adam_kernel_matches_the_per_variable_update(mi, torch_cuda):
    """mi355_adam_step (all variables in one launch) against the statement-for-statement update of optimizers.py:263-302 that the optimiser
    runs for few variables: three steps, 9 variables of mixed sizes with per-variable learning rates; values equal to float rounding, and
    a variable without a gradient is left alone."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    init = {"v%d" % k: rng.normal(size=(1 if k % 3 else 5,)).astype(np.float32) for k in range(9)}
    grads = [{k: rng.normal(size=v.shape).astype(np.float32) for k, v in init.items()} for _ in range(3)]
    outs = []
    for fused in (True, False):
        opt = mi.Adam(lr=0.05)
        for k, v in init.items():
            opt[k] = torch.tensor(v, device="cuda")
        opt.set_learning_rate({"v2": 0.2, "v7": 0.01})
        for step, g in enumerate(grads):
            for k in init:
                if k == "v4" and step == 1:
                    continue                                                        # no gradient this step: t advances, the value stays
                opt[k].grad = torch.tensor(g[k], device="cuda")
            if fused:
                opt.step()
            else:
                real = dict(opt.variables)                                          # the per-variable branch is taken for <= 4 variables: one at a time
                for k in list(real):
                    opt.variables = {k: real[k]}
                    opt.step()
                    real[k] = opt.variables[k]
                opt.variables = real
        outs.append({k: opt[k].detach().cpu().numpy().copy() for k in init})
    for k in init:
        assert np.allclose(outs[0][k], outs[1][k], rtol=2e-6, atol=1e-7), (k, outs[0][k], outs[1][k])
        assert not np.allclose(outs[0][k], init[k])


@pytest.mark.gpu
@pytest.mark.parametrize("rfilter,n_cams", [("gaussian", 3), ("box", 3), ("gaussian", 7), ("box", 8)])
def test_geometry_adjoint_of_several_sensors_in_one_pass(mi, torch_cuda, rfilter, n_cams):
    """mi355_render_backward_sensors on the cap-lens arrangement watched by three, seven and eight (MI355_MAX_SENSORS_PER_PASS) perspective
    cameras from different sides (Gaussian and box filters; the lens-design test covers the tent filter): gradient buffer == sum of the
    mi355_render_backward calls, block by block, and the images of the one-pass render == the separate renders.  Plain tensors, explicit
    geometry request: every vertex / IOR slot is filled.  A ninth camera is refused."""
    torch = torch_cuda
    T = mi.ScalarTransform4f
    d = _cap_lens_scene(mi, 24, rfilter=rfilter)
    origins = ([-0.25, 0.3, 0.7], [0.1, -0.35, 0.5], [0.3, 0.3, 0.6], [-0.3, -0.2, 0.65], [0.0, 0.4, 0.55], [0.35, -0.1, 0.7], [-0.1, 0.1, 0.8], [0.2, 0.2, 0.9])
    for k, origin in enumerate(origins[:n_cams]):
        se = dict(d['sensor']); se['to_world'] = T.look_at(target=[0.0, 0.0, 0.0], origin=origin, up=[0, 1, 0]); se['fov'] = 50.0 + 5.0 * k
        se['film'] = dict(d['sensor']['film'])
        d['sensor_%d' % (k + 1)] = se
    scene = mi.load_dict(d)
    integ = scene.integrator()
    cams = list(range(n_cams))
    assert len(scene.sensors()) == n_cams + 1 and integ.sensors_share_light_paths(scene, cams)
    assert not integ.sensors_share_light_paths(scene, list(range(9)) if n_cams == 8 else cams + [0] * (9 - n_cams))
    one = integ.render_sensors(scene, cams, seed=3, spp=64)
    for k in cams:
        sep = integ.render(scene, sensor=k, seed=3, spp=64)
        assert float(sep.max()) > 0 and float((one[k] - sep).abs().max()) < 1e-5 * float(sep.max()), k
    params = mi.traverse(scene)
    for key in ("lens.vertex_positions", "lens.bsdf.int_ior"):                # ask for the geometry / IOR gradients (and for nothing else)
        params[key] = params[key].detach().clone().requires_grad_(True)
    params.update()
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    gins = [torch.rand((24, 24, 3), device="cuda", generator=gen) for _ in cams]
    if n_cams > 3:
        gins[2] = None                                                        # a camera whose image carries no adjoint
    params.zero_grad(); g_one = integ.render_backward_sensors(scene, params, gins, cams, seed=8, spp=32).clone()
    params.zero_grad()
    for k in cams:
        if gins[k] is not None:
            integ.render_backward(scene, params, gins[k], sensor=k, seed=8, spp=32)
    g_sum = params.grad_buffer(g_one.device).clone()
    seen = 0
    for kind, index, off, size in scene_layout(mi, scene):
        if kind not in (mi.abi.PARAM_VERTEX_POSITIONS, mi.abi.PARAM_BSDF_IOR):
            continue
        a, b = g_one[off:off + size], g_sum[off:off + size]
        m = float(b.abs().max())
        if m == 0:
            assert float(a.abs().max()) == 0; continue
        seen += 1
        assert float((a - b).abs().max()) < 1e-3 * m, (kind, index, float((a - b).abs().max()) / m)
    assert seen >= 2
    scene.release()
