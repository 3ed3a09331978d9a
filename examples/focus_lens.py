#!/usr/bin/env python3
"""Lens design in miniature with the MI355X engine: slide a glass ball along a collimated beam until its spot on the screen is smallest.

The fork's notebooks (notebooks/doublet_full_optim.ipynb, scripts/lens_design/design_problem.py) optimise lens scalars by rendering
`directionalarea` sources through lens MESHES with `ptracer` and differentiating the image with respect to the mesh vertices.  This is
that loop with one scalar: the scene is created once, `params['lens.vertex_positions']` is rewritten every iteration (BVH refit on the
device), the gradient comes from the reverse-mode geometry adjoint behind mi.render, the loss is the spot's second moment, the update is
plain gradient descent.  (mitsuba3-camera_amd/lens_design.py is the full mirror of the notebooks' DesignProblem: ~30 scalars, Adam.)
    python examples/focus_lens.py            # needs the GPU
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def icosphere(sub):
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(sub):
        cache, nf = {}, []
        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]; v.append(m / np.linalg.norm(m)); cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float32), np.array(f, dtype=np.uint32)


def scene_dict(mi, x, res):
    T = mi.ScalarTransform4f
    v, f = icosphere(3)
    return {"type": "scene",
            "sensor": {"type": "perspective", "fov": 30.0, "to_world": T.look_at([0.4, 0.0, 3.0], [1.4, 0, 0], [0, 1, 0]),
                       "film": {"type": "hdrfilm", "width": res, "height": res, "rfilter": {"type": "gaussian"}}},
            "glass": {"type": "dielectric", "int_ior": 1.5, "ext_ior": 1.0},
            "white": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.8, 0.8, 0.8]}},
            "beam": {"type": "disk", "to_world": T.translate([-1.8, 0, 0]) @ T.rotate([0, 1, 0], 90.0) @ T.scale(0.2), "bsdf": {"type": "ref", "id": "white"},
                     "emitter": {"type": "directionalarea", "radiance": {"type": "rgb", "value": [5.0, 5.0, 5.0]}}},
            "lens": {"type": "mesh", "vertex_positions": (v * 0.5 + np.array([x, 0, 0], dtype=np.float32)).astype(np.float32), "faces": f,
                     "vertex_normals": v, "bsdf": {"type": "ref", "id": "glass"}},
            "screen": {"type": "rectangle", "to_world": T.translate([1.4, 0, 0]) @ T.rotate([0, 1, 0], -90.0) @ T.scale(1.0), "bsdf": {"type": "ref", "id": "white"}}}


def spot_loss(img):
    """Second moment of the image about its centroid (the RMS spot size squared, in pixels^2), differentiable in torch."""
    import torch
    lum = img.sum(dim=2)
    h, w = lum.shape
    ys, xs = torch.meshgrid(torch.arange(h, device=img.device, dtype=torch.float32), torch.arange(w, device=img.device, dtype=torch.float32), indexing="ij")
    tot = lum.sum()
    cx, cy = (lum * xs).sum() / tot, (lum * ys).sum() / tot
    return (lum * ((xs - cx) ** 2 + (ys - cy) ** 2)).sum() / tot


def main(iterations=25, res=64, spp=64, x0=-0.9, lr=0.01, verbose=True):
    """The notebooks' loop: ONE scene; every iteration writes the lens mesh's `vertex_positions` (a function of the design scalar x) into
    the scene parameters -- the device replica refits its BVH and recomputes the vertex normals, it is not rebuilt -- renders with the
    particle tracer and back-propagates the spot loss through the reverse-mode geometry adjoint (mi.render is a torch.autograd op)."""
    import torch
    mi = load_package()
    integ = mi.ParticleTracerIntegrator({"max_depth": 4, "rr_depth": 100})
    scene = mi.load_dict(scene_dict(mi, x0, res))
    params = mi.traverse(scene)
    key = "lens.vertex_positions"
    v, _ = icosphere(3)
    base = torch.tensor(v * 0.5, device="cuda")                                 # the ball about the origin
    axis = torch.tensor([1.0, 0.0, 0.0], device="cuda")                         # d(vertex) / d(x): it slides along the beam
    x = torch.tensor(float(x0), device="cuda", requires_grad=True)
    losses = []
    for it in range(iterations):
        params[key] = (base + x * axis).reshape(-1)                              # differentiable w.r.t. x; pushed by mi.render
        img = mi.render(scene, params, integrator=integ, seed=it, spp=spp)
        loss = spot_loss(img)
        loss.backward()
        grad = float(x.grad)
        with torch.no_grad():
            x -= lr * max(-10.0, min(10.0, grad))
            x.grad.zero_()
        losses.append(float(loss.detach()))
        if verbose:
            print("it %2d  x = %+.4f  spot second moment = %8.3f px^2  d/dx = %+8.3f" % (it, float(x.detach()), losses[-1], grad))
    scene.release()
    return losses, float(x.detach())


if __name__ == "__main__":
    main()
