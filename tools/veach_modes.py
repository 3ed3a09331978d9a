import sys, importlib, time, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
mi = importlib.import_module("mitsuba3-camera_amd")
sc = mi.load_dict(mi.veach_mis_scene(1024))
for name, mode, regen in (("fused", mi.abi.MODE_FUSED, False), ("static", mi.abi.MODE_STATIC, False), ("wavefront", mi.abi.MODE_WAVEFRONT, False), ("regen", mi.abi.MODE_FUSED, True)):
    integ = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5}); integ.mode = mode; integ.regen = regen; integ.timing = True
    integ.render(sc, seed=0, spp=512); integ.counters(sc, reset=True)
    for i in range(2): img = integ.render(sc, seed=i, spp=512)
    c = integ.counters(sc, reset=True)
    print(name, "kernel_ms %.2f" % (c["kernel_ms"] / c["kernel_launches"]), float(img.mean()))
