"""Where a 3-field design iteration spends its host time (BASELINE configs[4]).  Usage (GPU box): python tools/iter_profile.py [subdiv]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design
subdiv = int(sys.argv[1]) if len(sys.argv) > 1 else 7
surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=subdiv); system.disable_all_materials()
s1 = system.get_rear_surface_params(50.0); system.elements[0].surface.params['c'] = float(s1['c']); system.elements[0].surface.params['z0'] = float(s1['z0']); system.compute_paraxial_quantities()
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=300, num_sources=3, max_field_angle=5.0, init_camera_z=0.1, fov=100.0).prepare()
sync = torch.cuda.synchronize
acc = {}
def T(name, fn):
    sync(); t = time.perf_counter(); r = fn(); sync(); acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3; return r
for it in range(8):
    if it == 3:
        acc.clear()
    T("lens_system.update (surfaces -> vertices, push, refit)", lambda: system.update(prob.params, prob.optimizer))
    for fs in prob.field_sources:
        fs.update(prob.params)
    losses = []
    imgs = T("render, 3 fields in one pass", lambda: mi.render_fields(prob.scene, prob.params, [0, 1, 2], seed=it, spp=256, spp_grad=128))
    for i in range(3):
        losses.append(T("loss x3", lambda: 10000.0 * prob.field_sources[i].evaluate_rms_loss(imgs[i], recenter=True)[0]))
    loss = losses[0] + losses[1] + losses[2]
    T("loss.backward (one walk + sweep for 3 fields, chain rule)", lambda: loss.backward())
    T("optimizer.step", lambda: prob.optimizer.step())
    T("params.update", lambda: prob.params.update())
n = 5
tot = sum(acc.values()) / n
for k, v in acc.items():
    print("%-60s %7.3f ms" % (k, v / n))
print("%-60s %7.3f ms" % ("sum (every phase followed by a device sync)", tot))
