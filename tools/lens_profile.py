"""cProfile of one design iteration of the lens doublet (host overhead around the kernels)."""
import cProfile, pstats, sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design
surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=7); system.disable_all_materials()
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=300, init_camera_z=0.1, fov=100.0).prepare()
for it in range(2): prob.step(it)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for it in range(2, 5): prob.step(it)
torch.cuda.synchronize(); print("ms per iteration", (time.perf_counter() - t0) / 3 * 1e3)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
