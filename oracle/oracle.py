"""ctypes wrapper of the CPU oracle (oracle/mi_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")


def build(force=False):
    deps = [os.path.join(HERE, "mi_oracle.c"), os.path.join(HERE, "mi_oracle_geo.inc"), os.path.join(os.path.dirname(HERE), "include", "mi355.h")]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= max(os.path.getmtime(f) for f in deps):
        return LIB
    r = subprocess.run(["make", "-C", HERE, "-B" if force else "-s", "liboracle.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return LIB


_lib = None
f32p = C.POINTER(C.c_float)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def lib(abi):
    """`abi` = the package's _abi module (ctypes mirror of include/mi355.h: the oracle consumes the same POD records)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    # MI355_ORACLE_LIB: another build of the same source, e.g. oracle/liboracle_asan.so (`make -C oracle asan`; run the CPU tests with
    # LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 MI355_ORACLE_LIB=oracle/liboracle_asan.so)
    l = C.CDLL(os.environ.get("MI355_ORACLE_LIB", LIB))
    l.orc_scene_create.restype = C.c_void_p
    l.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
    l.orc_scene_destroy.argtypes = [C.c_void_p]
    l.orc_scene_set_param.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    l.orc_tea_float32.restype = C.c_float
    l.orc_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    l.orc_tea_float64.restype = C.c_double
    l.orc_tea_float64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    l.orc_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    l.orc_pcg32_raw.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
    l.orc_sampler_floats.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    l.orc_perspective_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    l.orc_gaussian_coeffs.argtypes = [C.c_float, C.c_void_p]
    l.orc_gaussian_eval.argtypes = [C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    l.orc_square_to_cosine_hemisphere.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    l.orc_diffuse_eval_pdf.argtypes = [C.c_void_p] * 5
    l.orc_diffuse_sample.argtypes = [C.c_void_p] * 6
    l.orc_lane_radiance.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_uint32, C.c_uint32, C.c_void_p]
    l.orc_render_accumulate.restype = C.c_uint64
    l.orc_render_accumulate.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_int]
    l.orc_film_develop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.orc_film_put.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    l.orc_param_layout.restype = C.c_int
    l.orc_param_layout.argtypes = [C.c_void_p, C.POINTER(abi.ParamInfo), C.c_uint32, C.POINTER(C.c_uint64)]
    l.orc_prb_splat_weights.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p]
    l.orc_render_backward.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    l.orc_render_forward.restype = C.c_uint64
    l.orc_render_forward.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_int]
    l.orc_ptracer_backward.restype = C.c_uint64
    l.orc_ptracer_backward.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_int]
    l.orc_ray_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    l.orc_primary_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    l.orc_lens_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    l.orc_shape_surface_area.restype = C.c_float
    l.orc_shape_surface_area.argtypes = [C.c_void_p, C.c_uint32]
    l.orc_texture_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
    l.orc_get_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    l.orc_dead_intersections.restype = C.c_uint64; l.orc_dead_intersections.argtypes = [C.c_void_p, C.c_int]
    l.orc_num_threads.restype = C.c_int
    l.orc_exit_pupil_bounds.argtypes = [C.c_void_p, C.c_void_p]
    l.orc_ptracer_accumulate.restype = C.c_uint64
    l.orc_ptracer_accumulate.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_void_p, C.c_int]
    l.orc_radical_inverse.restype = C.c_float
    l.orc_radical_inverse.argtypes = [C.c_uint32, C.c_uint64]
    l.orc_geo_mesh_hit_jacobian.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    l.orc_geo_plane_hit_jacobian.argtypes = [C.c_void_p] * 5
    l.orc_vertex_normals.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    l.orc_select_sensor.restype = C.c_int
    l.orc_select_sensor.argtypes = [C.c_void_p, C.c_uint32]
    l.orc_scene_set_sensor.restype = C.c_int
    l.orc_scene_set_sensor.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(abi.SensorDesc)]
    l.orc_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p]
    _lib = l
    return l


class OracleScene:
    """CPU twin of a package Scene: built from the same mi355_scene_desc."""

    def __init__(self, scene, abi):
        self.abi = abi
        self.l = lib(abi)
        self.scene = scene
        d = scene.desc()
        self.h = self.l.orc_scene_create(C.byref(d))
        if not self.h:
            raise RuntimeError("oracle rejected the scene description")
        self.select_sensor(0)

    def select_sensor(self, index):
        """The `Sensor *` argument of Integrator::render: the film-shaped helpers below (render, develop, film_put, primary_rays ...) work on
        this sensor; the render entry points pick theirs from mi355_render_params.sensor."""
        assert self.l.orc_select_sensor(self.h, int(index)) == 0
        f = self.scene.sensors()[int(index)].film
        self.W, self.H = f.crop_width, f.crop_height
        self.sensor = int(index)

    def set_sensor(self, index):
        """Re-read sensor `index` from the host scene (after its to_world / x_fov / film changed)."""
        d = self.scene.sensor_desc(int(index))
        assert self.l.orc_scene_set_sensor(self.h, int(index), C.byref(d)) == 0
        if getattr(self, "sensor", None) == int(index):
            f = self.scene.sensors()[int(index)].film
            self.W, self.H = f.crop_width, f.crop_height

    def trace(self, rays, num_lenses, max_depth):
        """BasicPRBIntegrator.trace (prb_basic.py:46-140) on caller rays (n x 8): n x 12, the layout of mi355_trace_rays."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        out = np.empty((rays.shape[0], 12), dtype=np.float32)
        self.l.orc_trace(self.h, _p(rays), rays.shape[0], int(num_lenses), int(max_depth), _p(out))
        return out

    def __del__(self):
        try:
            if self.h:
                self.l.orc_scene_destroy(self.h); self.h = None
        except Exception:
            pass

    def params(self, integrator, seed, spp, shard=None, spp_shard=None, sensor=None):
        s = self.sensor if sensor is None else int(sensor)
        if s != self.sensor:
            self.select_sensor(s)
        return integrator._params(self.scene, seed, spp, shard=shard, spp_shard=spp_shard, sensor=s)

    def set_param(self, kind, index, values):
        v = np.ascontiguousarray(values, dtype=np.float32)
        assert self.l.orc_scene_set_param(self.h, kind, index, _p(v), v.size) == 0

    def lane_radiance(self, p, lane_begin, count):
        out = np.empty((count, 4), dtype=np.float32)
        self.l.orc_lane_radiance(self.h, C.byref(p), lane_begin, count, _p(out))
        return out

    def render(self, p, threads=0, develop=True):
        film = np.zeros((self.H, self.W, 4), dtype=np.float32)
        n = self.l.orc_render_accumulate(self.h, C.byref(p), _p(film), threads)
        if not develop:
            return film, n
        img = np.empty((self.H, self.W, 3), dtype=np.float32)
        self.l.orc_film_develop(self.h, _p(film), _p(img))
        return img, n

    def ptracer_render(self, p, threads=0, develop=True):
        """AdjointIntegrator::render with the particle tracer (ptracer.cpp): RGBW accumulator whose W stays 0."""
        film = np.zeros((self.H, self.W, 4), dtype=np.float32)
        n = self.l.orc_ptracer_accumulate(self.h, C.byref(p), _p(film), threads)
        return (self.develop(film) if develop else film), n

    def develop(self, film):
        film = np.ascontiguousarray(film, dtype=np.float32)
        img = np.empty((self.H, self.W, 3), dtype=np.float32)
        self.l.orc_film_develop(self.h, _p(film), _p(img))
        return img

    def film_put(self, pos, values):
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 2); values = np.ascontiguousarray(values, dtype=np.float32).reshape(-1, 4)
        film = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self.l.orc_film_put(self.h, _p(film), _p(pos), _p(values), pos.shape[0])
        return film

    def param_layout(self):
        arr = (self.abi.ParamInfo * 4096)(); total = C.c_uint64()
        n = self.l.orc_param_layout(self.h, arr, 4096, C.byref(total))
        return [(arr[i].kind, arr[i].index, arr[i].offset, arr[i].size) for i in range(n)], total.value

    def splat_weights(self, p):
        w = np.zeros((self.H, self.W), dtype=np.float32)
        self.l.orc_prb_splat_weights(self.h, C.byref(p), _p(w))
        return w

    def render_backward(self, p, grad_in, weight=None, threads=0):
        if weight is None:
            weight = self.splat_weights(p)
        _, total = self.param_layout()
        g = np.zeros(max(total, 1), dtype=np.float32)
        grad_in = np.ascontiguousarray(grad_in, dtype=np.float32); weight = np.ascontiguousarray(weight, dtype=np.float32)
        self.l.orc_render_backward(self.h, C.byref(p), _p(grad_in), _p(weight), _p(g), threads)
        return g

    def render_forward(self, p, tangent, threads=0, develop=True):
        """RBIntegrator.render_forward (common.py:498-629): gradient image for the parameter tangent `tangent` (gradient-buffer layout)."""
        _, total = self.param_layout()
        tangent = np.ascontiguousarray(tangent, dtype=np.float32); assert tangent.size == max(total, 1)
        film = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self.l.orc_render_forward(self.h, C.byref(p), _p(tangent), _p(film), threads)
        return self.develop(film) if develop else film

    def ptracer_backward(self, p, grad_in, threads=0):
        """Integrator::render_backward (integrator.cpp:65-84) of the particle tracer, emitter radiances only."""
        _, total = self.param_layout()
        g = np.zeros(max(total, 1), dtype=np.float32)
        grad_in = np.ascontiguousarray(grad_in, dtype=np.float32)
        assert grad_in.shape == (self.H, self.W, 3)
        self.l.orc_ptracer_backward(self.h, C.byref(p), _p(grad_in), _p(g), threads)
        return g

    def ray_intersect(self, rays, with_si=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        hits = np.empty((rays.shape[0], 8), dtype=np.float32)
        si = np.empty((rays.shape[0], 24), dtype=np.float32) if with_si else None
        self.l.orc_ray_intersect(self.h, _p(rays), rays.shape[0], _p(hits), _p(si) if with_si else None)
        return (hits, si) if with_si else hits

    def primary_rays(self, sample_pos):
        sp = np.ascontiguousarray(sample_pos, dtype=np.float32).reshape(-1, 2)
        rays = np.empty((sp.shape[0], 8), dtype=np.float32)
        self.l.orc_primary_rays(self.h, _p(sp), sp.shape[0], _p(rays))
        return rays

    def lens_rays(self, sample_pos, aperture):
        sp = np.ascontiguousarray(sample_pos, dtype=np.float32).reshape(-1, 2); ap = np.ascontiguousarray(aperture, dtype=np.float32).reshape(-1, 2)
        rays = np.empty((sp.shape[0], 8), dtype=np.float32)
        self.l.orc_lens_rays(self.h, _p(sp), _p(ap), sp.shape[0], _p(rays))
        return rays

    def exit_pupil_bounds(self):
        out = np.empty((64, 4), dtype=np.float32)
        self.l.orc_exit_pupil_bounds(self.h, _p(out))
        return out

    def surface_area(self, shape):
        return float(self.l.orc_shape_surface_area(self.h, shape))

    def texture_eval(self, tex, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        out = np.empty((uv.shape[0], 3), dtype=np.float32)
        self.l.orc_texture_eval(self.h, tex, _p(uv), uv.shape[0], _p(out))
        return out

    def counters(self, reset=False):
        c = np.zeros(3, dtype=np.uint64)
        self.l.orc_get_counters(self.h, _p(c), 1 if reset else 0)
        dead = int(self.l.orc_dead_intersections(self.h, 1 if reset else 0))     # light tracer: computed by the reference, never read (ptracer.cpp:264-269)
        return {"n_isect": int(c[0]), "n_shade": int(c[1]), "n_shadow": int(c[2]), "n_isect_dead": dead}


def geo_mesh_hit_jacobian(l, tri, normals, o, d, flip=False):
    """Value (15: p, ng, ns, sh_s, sh_t) and Jacobian (15 x 24: ray origin 3, direction 3, positions 9, normals 9) of the oracle's multi-dual
    mesh hit (mi_oracle_geo.inc: md_mesh_hit)."""
    tri = np.ascontiguousarray(tri, dtype=np.float32).reshape(9); o = np.ascontiguousarray(o, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32)
    nrm = None if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(9)
    out = np.zeros(15 + 15 * 24, dtype=np.float64)
    l.orc_geo_mesh_hit_jacobian(_p(tri), _p(nrm) if nrm is not None else None, 1 if flip else 0, _p(o), _p(d), _p(out))
    return out[:15], out[15:].reshape(15, 24)


def geo_plane_hit_jacobian(l, q, n, o, d):
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (q, n, o, d)]
    out = np.zeros(3 + 18, dtype=np.float64)
    l.orc_geo_plane_hit_jacobian(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return out[:3], out[3:].reshape(3, 6)


def vertex_normals(l, positions, faces):
    """Mesh::recompute_vertex_normals (mesh.cpp:321-413) of the oracle on caller buffers."""
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3); f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
    out = np.zeros_like(pos)
    l.orc_vertex_normals(_p(pos), pos.shape[0], _p(f), f.shape[0], _p(out))
    return out


def unpack_hits(hits):
    """n x 8 float32 -> dict with integer fields reinterpreted."""
    u = hits.view(np.uint32)
    return {"t": hits[:, 0], "u": hits[:, 1], "v": hits[:, 2], "prim": u[:, 3], "shape": u[:, 4], "occluded": u[:, 5]}
