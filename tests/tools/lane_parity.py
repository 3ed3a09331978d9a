"""Lane-level agreement with the oracle on the glossy scene (test_veach_lane_radiance's lanes) at several tolerances, for the library
named by MI355_LIB -- run once with the shipped build and once with a -DMI_EXACT_MATH build to see what the hardware approximations
(v_rcp / v_rsq / v_sqrt / v_exp / v_sin) buy.  Usage (GPU box):  [MI355_LIB=build/libmi355_exact.so] python tests/tools/lane_parity.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package, load_oracle
mi = load_package(); orc = load_oracle(); orc.lib(mi.abi)
scene = mi.load_dict(mi.veach_mis_scene(64)); o = orc.OracleScene(scene, mi.abi)
print("library:", os.environ.get("MI355_LIB", "mitsuba3-camera_amd/libmi355.so"))
for name, md, rr in (("path", 8, 5), ("path", 3, 5), ("prb", 6, 2)):
    integ = mi.make_integrator({"type": name, "max_depth": md, "rr_depth": rr})
    n = 64 * 64 * 4
    ref = o.lane_radiance(o.params(integ, 3, 4), 0, n); got = integ.lane_radiance(scene, 3, 4, 0, n).cpu().numpy()
    same_valid = got[:, 3] == ref[:, 3]
    line = "%-4s max_depth %d rr_depth %d:" % (name, md, rr)
    for rtol in (1e-5, 1e-4, 2e-4, 2e-3):
        close = np.isclose(got[:, :3], ref[:, :3], rtol=rtol, atol=1e-4 if rtol >= 2e-3 else 1e-5).all(axis=1) & same_valid
        line += "  rtol %.0e: %.4f %%" % (rtol, 100 * close.mean())
    line += "  mean ratio %s" % np.round(got[:, :3].mean(0) / ref[:, :3].mean(0), 5)
    print(line)
