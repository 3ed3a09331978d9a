"""Development check of the geometry adjoint (GPU): device render_backward(geometry=True) against the oracle's multi-dual derivative."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, load_oracle
mi = load_package(); orc = load_oracle()
ld = mi.lens_design
T = mi.ScalarTransform4f

def lens_scene(res=32, subdiv=2, rfilter="tent"):
    surf = ld.ConicSurface(radial_extent=0.5, c=1 / 1.2, K=0.0, z0=1.0)
    V, F, Va, Fa, _ = ld.create_surface_geometry(subdiv, 0.5, surf.compute_z_np, c=surf.get_curvature(), baffle_radius=0.7, flip_normals=True)
    d = {'type': 'scene', 'integrator': {'type': 'ptracer', 'max_depth': 4},
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
    return d

def main(rfilter="tent", spp=16, update=True, seed=1):
    res = 32
    print("==== rfilter", rfilter, "spp", spp, "update", update)
    d = lens_scene(res, rfilter=rfilter)
    sc = mi.load_dict(d)
    print("scene info", sc.info())
    integ = mi.ParticleTracerIntegrator({'max_depth': 4, 'rr_depth': 5})
    params = mi.traverse(sc)
    li = [i for i, s in enumerate(sc.shapes) if s.id == 'lens'][0]
    # a parameter update (makes the normals follow the positions): scale z slightly
    pos = params['lens.vertex_positions'].reshape(-1, 3).clone(); pos[:, 2] = 1.0 + (pos[:, 2] - 1.0) * 1.05
    o = orc.OracleScene(sc, mi.abi)
    if update:
        params['lens.vertex_positions'] = pos.reshape(-1); params.update()
        o.set_param(mi.abi.PARAM_VERTEX_POSITIONS, li, pos.cpu().numpy().reshape(-1))
    img = integ.render(sc, seed=seed, spp=spp).cpu().numpy()
    p = o.params(integ, seed, spp)
    ref, _ = o.ptracer_render(p)
    print("primal: max", ref.max(), "err", np.abs(img - ref).max() / ref.max())
    rng = np.random.default_rng(5)
    gin = rng.uniform(0.5, 1.5, (res, res, 3)).astype(np.float32)
    params.zero_grad()
    g = integ.render_backward(sc, params, torch.tensor(gin), seed=seed, spp=spp, geometry=True).cpu().numpy()
    p.flags |= mi.abi.FLAG_GEOMETRY_GRADS
    t0 = time.time(); g_ref = o.ptracer_backward(p, gin); print("oracle backward %.1f s" % (time.time() - t0))
    lay, tot = o.param_layout()
    for k, i, off, size in lay:
        a, b = g[off:off + size], g_ref[off:off + size]
        m = np.abs(b).max()
        print("kind %d index %d size %6d  max|ref| %.4e  max|dev-ref| %.4e  rel %.3e" % (k, i, size, m, np.abs(a - b).max(), np.abs(a - b).max() / max(m, 1e-30)))
        if k == 3 and m > 0:
            e = np.abs(a - b).reshape(-1, 3).max(axis=1) / m
            print("   per-vertex rel err: median %.2e  p90 %.2e  max %.2e  (#>1e-3: %d of %d)  rel L2 %.3e" % (np.median(e), np.percentile(e, 90), e.max(), (e > 1e-3).sum(), e.size, np.linalg.norm(a - b) / np.linalg.norm(b)))

for sd in range(2, 8):
    main("tent", 16, True, sd)
