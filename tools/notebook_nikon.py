"""notebooks/report_nikon_exp.ipynb, cells 12-19, on the GPU: the 22-surface wide-angle design (aperture stop, two aspheres zeroed), DesignProblem(num_sources=6,
max_field_angle=32.45, init_camera_z=0.99), a few iterations.  3.6 M triangles at subdivision 7.  Usage (GPU box): python tools/notebook_nikon.py [subdiv] [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design
from test_lens_design_cpu import _nikon
subdiv = int(sys.argv[1]) if len(sys.argv) > 1 else 7; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.perf_counter()
system, ai = _nikon(ld, subdiv)
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=0.05e-5 * 1000 * 0.6, iters=500, num_sources=6, max_field_angle=0.5 * 64.9, init_camera_z=0.99).prepare()
torch.cuda.synchronize(); print("prepare: %.1f s; scene info %s; tracer check %s (the notebook: 9.68e-5, 4.52e-5)" % (time.perf_counter() - t0, prob.scene.info(), prob.geo_check))
print("optimizer variables:", len(prob.optimizer.keys()))
losses = []
for it in range(iters):
    t1 = time.perf_counter(); l, up = prob.step(it); torch.cuda.synchronize(); losses.append(l)
    print("iteration %d: loss %.6g  upsample %s  %.1f ms" % (it, l, up, (time.perf_counter() - t1) * 1e3))
