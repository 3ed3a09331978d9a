import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design
surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=7); system.disable_all_materials()
s1 = system.get_rear_surface_params(50.0); system.elements[0].surface.params['c'] = float(s1['c']); system.elements[0].surface.params['z0'] = float(s1['z0']); system.compute_paraxial_quantities()
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=300, num_sources=3, max_field_angle=5.0, init_camera_z=0.1, fov=100.0).prepare()
for it in range(3): prob.step(it)
torch.cuda.synchronize()
pc = time.perf_counter
for it in range(3, 6):
    T = {}
    t0 = pc(); system.update(prob.params, prob.optimizer); T['update'] = pc() - t0
    t0 = pc()
    for fs in prob.field_sources: fs.update(prob.params)
    T['fs.update'] = pc() - t0
    loss = None; pend = []
    for i in range(3):
        t0 = pc(); img = mi.render(prob.scene, prob.params, seed=it, spp=256, spp_grad=128, sensor=i); T['render%d' % i] = pc() - t0
        t0 = pc(); l, st = prob.field_sources[i].spot_loss(img); pend.append(st); loss = 10000.0 * l if loss is None else loss + 10000.0 * l; T['loss%d' % i] = pc() - t0
    t0 = pc(); loss.backward(); T['backward'] = pc() - t0
    t0 = pc(); prob.optimizer.step(); T['step'] = pc() - t0
    t0 = pc(); torch.cuda.synchronize(); T['final sync (GPU still busy)'] = pc() - t0
    prob.params.update()
    print("iteration %d host-side call times (ms, no sync inside):" % it, {k: round(v * 1e3, 3) for k, v in T.items()})
