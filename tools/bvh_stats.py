"""Traversal steps per ray of the BVH path (diagnostic build -DMI_BVH_STATS) on the lens doublet of BASELINE configs[4] and on a
Cornell box with a 327 k-triangle icosphere: nodes fetched and triangles tested per ray, kernel ms.
  python mitsuba3-camera_amd/build.py --out=build/libmi355_stats.so -DMI_BVH_STATS && MI355_LIB=build/libmi355_stats.so python tools/bvh_stats.py"""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
mi = load_package(); ld = mi.lens_design

def stats(scene, reset=True):
    out = (C.c_uint64 * 4)(); mi.abi.check(mi.abi.lib().mi355_debug_bvh_stats(scene.handle(), out, 1 if reset else 0)); return list(out)

def run(name, scene, integ, spp):
    integ.timing = True
    integ.render(scene, seed=0, spp=spp); torch.cuda.synchronize(); integ.counters(scene, reset=True); stats(scene)
    integ.render(scene, seed=1, spp=spp); torch.cuda.synchronize()
    c = integ.counters(scene, reset=True); st = stats(scene)
    rays = max(st[0], 1)
    return {"scene": name, "info": scene.info(), "kernel_ms": round(c["kernel_ms"], 3), "bvh_rays": st[0], "nodes_per_ray": round(st[1] / rays, 2), "tris_per_ray": round(st[2] / rays, 2),
            "n_isect_plus_shadow": c["n_isect"] + c["n_shadow"]}

out = []
surfs = [{'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 140, 'z0': 48}, {'radial_extent': 8.0, 'K': 0.0, 'c': -1 / 24, 'z0': 49}, {'radial_extent': 8.0, 'K': 0.0, 'c': 1 / 24, 'z0': 52}]
mats = [ld.LensMaterial("sf11", 1.6447, 0.2 * 64.17), ld.LensMaterial("nbk7", 1.5168, 0.2 * 64.17)]
system = ld.LensSystem([ld.EvenAsphericSurface(**s) for s in surfs], mats, subdiv_levels=7)
prob = ld.DesignProblem(system, resolution=(128, 128), spp=128, learning_rate=3e-5, iters=1, init_camera_z=0.1, fov=100.0).prepare()
out.append(run("lens doublet, ptracer 128^2 x 256 spp", prob.scene, prob.scene.integrator(), 256))
d = mi.cornell_box(); d["sensor"]["film"]["width"] = 512; d["sensor"]["film"]["height"] = 512
V, F = ld.icosphere(7)
d["ball"] = {"type": "mesh", "vertex_positions": (V * 0.35 + np.array([-0.3, 0.45, 0.2])).astype(np.float32), "faces": F.astype(np.uint32), "compute_vertex_normals": True, "bsdf": {"type": "ref", "id": "white"}}
sc = mi.load_dict(d)
out.append(run("cornell box + 327k-triangle icosphere, path 512^2 x 16 spp", sc, mi.PathIntegrator({"max_depth": 6}), 16))
print(json.dumps(out, indent=1))
