import sys, importlib, numpy as np, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mi = importlib.import_module("mitsuba3-camera_amd")
for name, d in (("cbox1024", (lambda d: (d["sensor"]["film"].update(width=1024, height=1024), d)[1])(mi.cornell_box())), ("veach512", mi.veach_mis_scene(512))):
    sc = mi.load_dict(d)
    integ = mi.ParticleTracerIntegrator({"max_depth": -1, "rr_depth": 2})
    t0 = time.time(); a = integ.render(sc, seed=1, spp=16); torch.cuda.synchronize(); t1 = time.time()
    integ.direct_splat = True
    b = integ.render(sc, seed=1, spp=16); torch.cuda.synchronize(); t2 = time.time()
    print(name, "deferred %.3fs direct %.3fs" % (t1 - t0, t2 - t1), "max rel diff %.2e" % float((a - b).abs().max() / b.abs().max()), "mean", float(a.mean()), bool(torch.isfinite(a).all()))
