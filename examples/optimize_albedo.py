#!/usr/bin/env python3
"""Inverse rendering with the MI355X engine: recover the colour of the Cornell box's red wall from a target image.

Mirrors the reference's optimisation tutorials (mi.render as a differentiable op + mi.ad.Adam):
    python examples/optimize_albedo.py            # needs the GPU
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main(iterations=40, res=128, spp=32, verbose=True):
    import torch
    mi = load_package()
    d = mi.cornell_box()
    d["sensor"]["film"]["width"] = res; d["sensor"]["film"]["height"] = res
    scene = mi.load_dict(d)
    integ = mi.PRBIntegrator({"max_depth": 6})
    params = mi.traverse(scene)
    key = "red.reflectance.value"
    target_value = params[key].clone()
    img_ref = mi.render(scene, params, integrator=integ, seed=0, spp=4 * spp)

    params[key] = torch.tensor([0.1, 0.4, 0.7], device=img_ref.device)       # wrong starting colour
    params.update()
    opt = mi.Adam(lr=0.05)
    opt[key] = params[key]
    params.update(opt)
    losses, errs = [], []
    for it in range(iterations):
        img = mi.render(scene, params, integrator=integ, seed=it + 1, spp=spp)
        loss = ((img - img_ref) ** 2).mean()
        loss.backward()
        opt.step()
        opt[key] = opt[key].detach().clamp(0.0, 1.0)
        params.update(opt)
        losses.append(float(loss.detach())); errs.append(float((params[key].detach() - target_value).abs().sum()))
        if verbose and it % 5 == 0:
            print("it %3d  loss %.6f  param error %.4f  value %s" % (it, losses[-1], errs[-1], params[key].detach().cpu().numpy().round(3)))
    return losses, errs


if __name__ == "__main__":
    losses, errs = main()
    print("final parameter error: %.4f (started at %.4f)" % (errs[-1], errs[0]))
