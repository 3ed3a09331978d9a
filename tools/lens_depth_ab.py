"""Cost of each path segment of the lens doublet's light tracer: render / geometry adjoint kernel time against max_depth (the design problem
uses surfaces + 2 = 5).  Usage (GPU box): python tools/lens_depth_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design
surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=7); system.disable_all_materials()
s1 = system.get_rear_surface_params(50.0); system.elements[0].surface.params['c'] = float(s1['c']); system.elements[0].surface.params['z0'] = float(s1['z0']); system.compute_paraxial_quantities()
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=300, num_sources=3, max_field_angle=5.0, init_camera_z=0.1, fov=100.0).prepare()
sc, pl = prob.scene, prob.params
il = sc.integrator(); il.timing = True
system.update(pl, prob.optimizer); pl.update()
gin = torch.ones((128, 128, 3), dtype=torch.float32, device="cuda")
seed = [100]
def timed(fn, n=3):
    fn(); torch.cuda.synchronize(); il.counters(sc, reset=True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); c = il.counters(sc, reset=True)
    return c["kernel_ms"] / max(c["kernel_launches"], 1), {k: c[k] / max(c["n_gen"], 1) for k in ("n_isect", "n_shade", "n_shadow")}
def nxt():
    seed[0] += 1; return seed[0]
for md in (1, 2, 3, 4, 5, 6):
    il.max_depth = md
    r, ev = timed(lambda: il.render(sc, sensor=0, seed=nxt(), spp=256))
    b, _ = timed(lambda: il.render_backward(sc, pl, gin, sensor=0, seed=nxt(), spp=128, geometry=True, colour=False))
    print("max_depth %d: render %.3f ms  geometry adjoint %.3f ms   per light path: %.2f intersections, %.2f shading events, %.2f shadow rays" % (md, r, b, ev["n_isect"], ev["n_shade"], ev["n_shadow"]))
