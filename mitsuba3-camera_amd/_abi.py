"""ctypes mirror of include/mi355.h (field for field) and the loader of libmi355.so.

The product path has no CPU fallback: `lib()` raises if the HIP library is missing or
cannot be loaded, and every entry point raises `Mi355Error` on a non-zero status.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_LIB", os.path.join(HERE, "libmi355.so"))   # MI355_LIB: A/B builds of the same ABI

ABI_VERSION = 9

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_DEVICE, ERR_NO_DEVICE = 0, -1, -2, -3, -4

SHAPE_RECTANGLE, SHAPE_MESH, SHAPE_SPHERE, SHAPE_DISK = 0, 1, 2, 3
BSDF_DIFFUSE, BSDF_ROUGHCONDUCTOR, BSDF_DIELECTRIC, BSDF_CONDUCTOR = 0, 1, 2, 3
EMITTER_AREA, EMITTER_DIRECTIONALAREA = 0, 1
SENSOR_PERSPECTIVE, SENSOR_REALISTICLENS = 0, 1
RFILTER_BOX, RFILTER_GAUSSIAN, RFILTER_TENT = 0, 1, 2
FILTER_NEAREST, FILTER_BILINEAR = 0, 1
WRAP_REPEAT, WRAP_MIRROR, WRAP_CLAMP = 0, 1, 2
FLAG_HIDE_EMITTERS, FLAG_TIMING, FLAG_REGEN, FLAG_TWO_PASS, FLAG_EMITTER_GRADS_ONLY, FLAG_DIRECT_SPLAT, FLAG_GEOMETRY_GRADS, FLAG_SKIP_COLOUR_GRADS = 1, 2, 4, 8, 16, 32, 64, 128
FLAG_NO_LANE_SORT = 256
MODE_FUSED, MODE_REPACK_ALWAYS, MODE_STATIC = 0, 1, 2
MAX_SENSORS_PER_PASS = 8                     # MI355_MAX_SENSORS_PER_PASS
MODE_WAVEFRONT = MODE_REPACK_ALWAYS          # name up to ABI v8 (one fused launch either way: nothing wavefront-formulated about it)
INTEGRATOR_PATH, INTEGRATOR_PRB, INTEGRATOR_PTRACER = 0, 1, 2
PARAM_BSDF_REFLECTANCE, PARAM_TEXTURE_DATA, PARAM_EMITTER_RADIANCE, PARAM_VERTEX_POSITIONS, PARAM_BSDF_IOR, PARAM_SENSOR_TO_WORLD, PARAM_SENSOR_X_FOV = 0, 1, 2, 3, 4, 5, 6

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class TextureDesc(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32),
                ("filter", C.c_uint32), ("wrap", C.c_uint32), ("to_uv", C.c_float * 6),
                ("data", f32p)]


class BsdfDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("reflectance", C.c_float * 3), ("reflectance_texture", C.c_int32),
                ("alpha_u", C.c_float), ("alpha_v", C.c_float), ("distribution", C.c_uint32),
                ("eta", C.c_float * 3), ("k", C.c_float * 3), ("int_ior", C.c_float), ("ext_ior", C.c_float),
                ("sample_visible", C.c_uint32), ("transmittance", C.c_float * 3)]


class EmitterDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("radiance", C.c_float * 3), ("shape", C.c_uint32)]


class ShapeDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("bsdf", C.c_uint32), ("emitter", C.c_int32), ("flip_normals", C.c_uint32),
                ("to_world", C.c_float * 12), ("to_object", C.c_float * 12),
                ("vertex_count", C.c_uint32), ("face_count", C.c_uint32),
                ("vertex_positions", f32p), ("vertex_normals", f32p), ("vertex_texcoords", f32p),
                ("faces", u32p)]


class SensorDesc(C.Structure):
    _fields_ = [("type", C.c_uint32), ("to_world", C.c_float * 12), ("x_fov", C.c_float),
                ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("film_width", C.c_uint32), ("film_height", C.c_uint32),
                ("crop_offset_x", C.c_uint32), ("crop_offset_y", C.c_uint32),
                ("crop_width", C.c_uint32), ("crop_height", C.c_uint32),
                ("rfilter", C.c_uint32), ("rfilter_param", C.c_float),
                ("lens_count", C.c_uint32), ("lens_data", f32p),
                ("film_diagonal", C.c_float), ("focus_distance", C.c_float),
                ("shutter_open_time", C.c_float), ("sample_exit_pupil", C.c_uint32), ("sample_border", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_shapes", C.c_uint32), ("n_bsdfs", C.c_uint32),
                ("n_emitters", C.c_uint32), ("n_textures", C.c_uint32),
                ("shapes", C.POINTER(ShapeDesc)), ("bsdfs", C.POINTER(BsdfDesc)),
                ("emitters", C.POINTER(EmitterDesc)), ("textures", C.POINTER(TextureDesc)),
                ("n_sensors", C.c_uint32), ("sensors", C.POINTER(SensorDesc))]


class RenderParams(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_int32), ("rr_depth", C.c_int32),
                ("flags", C.c_uint32), ("mode", C.c_uint32), ("integrator", C.c_uint32),
                ("pixel_begin", C.c_uint32), ("pixel_count", C.c_uint32),
                ("spp_begin", C.c_uint32), ("spp_count", C.c_uint32),
                ("pixel_tile", C.c_uint32), ("pixel_tile_stride", C.c_uint32), ("passes", C.c_uint32), ("sensor", C.c_uint32)]


class DesignTangent(C.Structure):
    _fields_ = [("shape", C.c_uint32), ("d_vertex_tangent", C.c_void_p), ("bsdf", C.c_uint32), ("d_int_ior", C.c_float), ("d_ext_ior", C.c_float)]


NONE = 0xffffffff


class Counters(C.Structure):
    _fields_ = [("n_gen", C.c_uint64), ("n_isect", C.c_uint64), ("n_shade", C.c_uint64),
                ("n_shadow", C.c_uint64), ("n_splat", C.c_uint64),
                ("kernel_ms", C.c_float), ("kernel_launches", C.c_uint32), ("n_redo_groups", C.c_uint64)]


class ParamInfo(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("index", C.c_uint32), ("offset", C.c_uint64), ("size", C.c_uint64)]


class AdamSlot(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("param_out", C.c_void_p), ("lr_t", C.c_float), ("count", C.c_uint32)]


# every symbol include/mi355.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("mi355_last_error", C.c_char_p, []),
    ("mi355_abi_version", C.c_uint32, []),
    ("mi355_scene_create", C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    ("mi355_scene_destroy", C.c_int, [C.c_void_p]),
    ("mi355_scene_set_param", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, f32p, C.c_uint64]),
    ("mi355_scene_set_param_device", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("mi355_scene_param_enable_grad", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int]),
    ("mi355_scene_param_layout", C.c_int, [C.c_void_p, C.POINTER(ParamInfo), C.c_uint32, C.POINTER(C.c_uint64)]),
    ("mi355_render_accumulate", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    ("mi355_film_develop", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_scene_set_sensor", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(SensorDesc)]),
    ("mi355_scene_sensor_count", C.c_int, [C.c_void_p]),
    ("mi355_lens_sag", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_lens_sag_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_multi_film_layout", C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("mi355_render_accumulate_sensors", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    ("mi355_render_backward_sensors", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]),
    ("mi355_spot_loss", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mi355_spot_loss_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mi355_adam_step", C.c_int, [C.c_void_p, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    ("mi355_trace_rays", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]),
    ("mi355_render", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_prb_splat_weights", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p]),
    ("mi355_render_backward", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_get_counters", C.c_int, [C.c_void_p, C.POINTER(Counters), C.c_int]),
    ("mi355_scene_info", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("mi355_lens_exit_pupil_bounds", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_float)]),
    ("mi355_debug_lane_radiance", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mi355_ptracer_forward_design", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.POINTER(DesignTangent), C.c_void_p, C.c_void_p]),
    ("mi355_render_forward", C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_debug_ray_intersect", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mi355_set_reduce_hook", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("mi355_debug_surface_interaction", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mi355_debug_texture_eval", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("mi355_debug_set_splat_capacity", C.c_int, [C.c_void_p, C.c_uint32]),
    ("mi355_debug_bvh_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]),
    ("mi355_perspective_projection", None, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.c_float, C.c_float, C.c_float, f32p]),
    ("mi355_gaussian_coeffs", None, [C.c_float, f32p]),
]


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)     # mi355_reduce_fn


class Mi355Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mi355 error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    """Load libmi355.so (built by build.py).  Raises if it is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmi355.so not built: run `python __graft_entry__.py build` "
                          "(this engine has no CPU fallback)")
    l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(l, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if l.mi355_abi_version() != ABI_VERSION:
        raise ImportError("libmi355.so ABI version mismatch")
    _lib = l
    return l


def check(code):
    if code < 0:
        raise Mi355Error(code, lib().mi355_last_error().decode("utf-8", "replace"))
    return code
