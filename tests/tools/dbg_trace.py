import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from __graft_entry__ import load_package, load_oracle
mi = load_package(); orc = load_oracle(); orc.lib(mi.abi)
from test_lens_design_cpu import _doublet
ld = mi.lens_design
system = _doublet(ld, 5)
problem = ld.DesignProblem(system, resolution=(32, 32), spp=16, learning_rate=1e-3, iters=1, num_sources=3, max_field_angle=5.0, init_camera_z=0.1, fov=100.0)
geo = mi.load_dict(problem._build_preliminary_scene("prb_basic"))
print(geo.info(), [(s.id, s.type) for s in geo.shapes])
o = orc.OracleScene(geo, mi.abi)
rays = np.zeros((4, 8), dtype=np.float32)
rays[:, :3] = [[0.001, 0, 57.2], [0.5, 0.2, 57.2], [3.0, 1.0, 57.2], [0, 0, 57.2]]; rays[:, 3:6] = [0, 0, -1]; rays[:, 6] = 3.4e38
for depth in (1, 2, 3, 4):
    ref = o.trace(rays, depth, 5)
    out = torch.empty((4, 12), device="cuda"); d_rays = torch.tensor(rays, device="cuda")
    mi.abi.check(mi.abi.lib().mi355_trace_rays(geo.handle(), C.c_void_p(d_rays.data_ptr()), 4, depth, 5, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    print("depth", depth); print(np.round(ref, 5)); print(np.round(out.cpu().numpy(), 5))
hits = torch.empty((4, 8), device="cuda")
mi.abi.check(mi.abi.lib().mi355_debug_ray_intersect(geo.handle(), C.c_void_p(d_rays.data_ptr()), 4, C.c_void_p(hits.data_ptr()), None)); torch.cuda.synchronize()
print(orc.unpack_hits(hits.cpu().numpy())); print(orc.unpack_hits(o.ray_intersect(rays)))
