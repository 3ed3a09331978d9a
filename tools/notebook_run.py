"""notebooks/doublet_full_optim.ipynb, cells 5-15, end to end: the constrained doublet, DesignProblem(num_sources=3, init_camera_z=0.01), 300 iterations
of optimize(), final surface parameters next to the ones the notebook stores (cell 15; it ran the spectral `cuda_ad_dispersion` variant, this
is an RGB build at 589.3 nm: same loop, a monochromatic instead of a polychromatic spot).  Usage (GPU box): python tools/notebook_run.py [iters]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=7); system.disable_all_materials()
s1 = system.get_rear_surface_params(50.0); system.elements[0].surface.params['c'] = float(s1['c']); system.elements[0].surface.params['z0'] = float(s1['z0']); system.compute_paraxial_quantities()
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=iters, num_sources=3, max_field_angle=5.0, init_camera_z=0.01).prepare()
print("tracer check:", prob.geo_check)
torch.cuda.synchronize(); t0 = time.perf_counter()
losses = prob.optimize()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d iterations in %.2f s = %.2f ms per iteration; upsample steps %d" % (iters, dt, dt / iters * 1e3, prob.upsample_steps))
print("loss: first %.6g  mean of first 10 %.6g  mean of last 10 %.6g  min %.6g" % (losses[0], sum(losses[:10]) / 10, sum(losses[-10:]) / 10, min(losses)))
for fs in prob.field_sources:
    print("  %s: spot loss (mm^2) first %.4e last %.4e" % (fs.sensor_key, fs.losses[0], fs.losses[-1]))
ref = [{'c': -0.04504023864865303, 'K': -0.0009244396933354437, 'z0': 46.74386215209961, 'a4': 0.0009508539224043489, 'a6': 0.00010412279516458511, 'a8': -0.00014858432405162603,
        'a10': -0.00010279302659910172, 'a12': 5.96465433773119e-05, 'a14': 0.0002553249360062182, 'a16': 0.0004504272947087884},
       {'c': -0.3356727659702301, 'K': 0.0006883303285576403, 'z0': 49.002906799316406, 'a4': -0.0007925858371891081, 'a6': 9.541663166601211e-05, 'a8': 0.00038140988908708096,
        'a10': 0.0003720808308571577, 'a12': 0.0002486447338014841, 'a14': 9.209151903633028e-05, 'a16': -6.504303746623918e-05}]
init = [{'c': -0.04761601984500885, 'z0': 46.7470817565918}, {'c': -0.3333333432674408, 'z0': 49.0}]
for e in range(2):
    p = prob.lens_system.elements[e].surface.params
    print("surface %d   %-4s %14s %14s %14s %14s" % (e, "key", "initial", "here", "notebook", "moved (here / notebook)"))
    for k in ('c', 'K', 'z0', 'a4', 'a6', 'a8', 'a10', 'a12', 'a14', 'a16'):
        v = float(torch.as_tensor(p[k]).reshape(-1)[0]); i0 = init[e].get(k, 0.0)
        print("            %-4s % .7e % .7e % .7e   % .3e / % .3e" % (k, i0, v, ref[e][k], v - i0, ref[e][k] - i0))
