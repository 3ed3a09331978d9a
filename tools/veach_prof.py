"""Section profile (build with -DMI_PROF, MI355_LIB=that build): wave cycles per section of path_step on the Veach scene."""
import sys, importlib
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mi = importlib.import_module("mitsuba3-camera_amd")
for name, sc in (("veach", mi.load_dict(mi.veach_mis_scene(1024))), ("cbox", mi.load_dict((lambda d: (d["sensor"]["film"].update(width=512, height=512), d)[1])(mi.cornell_box())))):
    integ = mi.PathIntegrator({"max_depth": 6, "rr_depth": 5}); integ.mode = mi.abi.MODE_STATIC; integ.timing = True
    integ.render(sc, seed=0, spp=256); integ.counters(sc, reset=True)
    integ.render(sc, seed=1, spp=256)
    print(name, file=sys.stderr); c = integ.counters(sc, reset=True); print(name, c["kernel_ms"], {k: c[k] / c["n_gen"] for k in ("n_isect", "n_shade", "n_shadow")}, file=sys.stderr)
