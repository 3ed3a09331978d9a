"""mitsuba3-camera_amd -- MI355X-native path tracing (the reference's wavefront megakernel as fused per-lane kernels) + Path-Replay-Backprop.

Host-side mirror of the slice of Mitsuba 3's Python API that drives the `path` / `prb` hot
path (load_dict, cornell_box, traverse, render, Integrator.render/render_backward); all
sampling work happens in libmi355.so (hand-written HIP for gfx950) behind the C ABI of
include/mi355.h.  There is no CPU fallback: importing works anywhere, rendering needs the GPU.

The directory name contains a hyphen, so the package is loaded under the module name
`mitsuba3_camera_amd` by __graft_entry__.load_package().
"""
from . import _abi as abi
from ._abi import Mi355Error
from .transforms import Transform4f, ScalarTransform4f
from .scene import (Scene, Shape, BSDF, Emitter, Texture, Sensor, Film, PluginError, load_dict, cornell_box, veach_mis_scene, parse_fov)
from .integrators import (Integrator, PathIntegrator, PRBIntegrator, ParticleTracerIntegrator, make_integrator, SceneParameters, traverse, render, render_fields,
                          sample_tea_32, shard_pixels, shard_tiles, shard_of, scene_param_layout)

from .optimizers import Adam, SGD, Optimizer
from . import lens_design
from .xml_loader import load_file, parse_file

__all__ = [n for n in dir() if not n.startswith("_")]
_variant = "hip_ad_rgb"


def variant():
    return _variant


def set_variant(name):
    """Only one variant exists: the hand-written counterpart of llvm_ad_rgb / cuda_ad_rgb."""
    if name not in ("hip_ad_rgb",):
        raise PluginError('variant "%s" is not available (hip_ad_rgb)' % name)
