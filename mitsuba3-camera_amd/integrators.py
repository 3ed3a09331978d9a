"""Host-side mirror of the reference's Integrator interface for the `path` and `prb` plugins.

  Integrator.render / render_backward      include/mitsuba/render/integrator.h:74-215
  SamplingIntegrator::render (JIT branch)  src/render/integrator.cpp:246-360
  RBIntegrator.render_backward             src/python/python/ad/integrators/common.py:631-796
  mi.render + _RenderOp                    src/python/python/util.py:356-523
  mi.traverse / SceneParameters            src/python/python/util.py:13-350

Device memory, streams and the multi-GPU collectives come from torch (plumbing); every sample is
traced by libmi355.so through the C ABI.  One process drives one GPU; a multi-GPU job passes
`shard=(rank, world_size)` and the film / gradient buffers are summed with torch.distributed
(RCCL over xGMI) -- the only exchange the path has.
"""
import ctypes as C
import numpy as np

from . import _abi as abi
from .scene import Scene, PluginError


def _torch():
    import torch
    return torch


def sample_tea_32(v0, v1, rounds=4):
    """include/mitsuba/core/random.h:76-91 (host copy used for seed_grad, util.py:517)."""
    v0 &= 0xffffffff; v1 &= 0xffffffff; s = 0
    for _ in range(rounds):
        s = (s + 0x9e3779b9) & 0xffffffff
        v0 = (v0 + ((((v1 << 4) & 0xffffffff) + 0xa341316c) ^ ((v1 + s) & 0xffffffff) ^ ((v1 >> 5) + 0xc8013ea4))) & 0xffffffff
        v1 = (v1 + ((((v0 << 4) & 0xffffffff) + 0xad90777d) ^ ((v0 + s) & 0xffffffff) ^ ((v0 >> 5) + 0x7e95761e))) & 0xffffffff
    return v0, v1


def torch_float32():
    return _torch().float32


def _stream_ptr(device):
    torch = _torch()
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def shard_pixels(n_pixels, rank, world):
    """Contiguous pixel band of rank `rank` (row-major lanes => contiguous global lane ids)."""
    per = (n_pixels + world - 1) // world
    b = min(rank * per, n_pixels); e = min(b + per, n_pixels)
    return b, e - b


SHARD_TILE_ROWS = 16


def shard_tiles(width, height, rank, world, rows=SHARD_TILE_ROWS):
    """Interleaved row tiles ("16-row tiles round-robin" for load balance): tile t belongs to rank
    t % world.  Returns (pixel_begin, pixel_count, pixel_tile, pixel_tile_stride) for mi355_render_params; every tile is
    a contiguous range of global lane ids, and one launch covers all of a rank's tiles."""
    tile = rows * width
    npix = width * height
    n_tiles = (npix + tile - 1) // tile
    mine = range(rank, n_tiles, world)
    count = sum(min(tile, npix - t * tile) for t in mine)
    return (min(rank * tile, npix), count, tile, world * tile)


def shard_of(film, rank, world, grid=None):
    """The shard a rank renders: interleaved 16-row tiles when every rank gets at least one, else contiguous bands."""
    w, h = grid if grid is not None else (film.crop_width, film.crop_height)
    if world > 1 and (h + SHARD_TILE_ROWS - 1) // SHARD_TILE_ROWS >= world:
        return shard_tiles(w, h, rank, world)
    return shard_pixels(w * h, rank, world)


class HipBackend:
    """The product back-end: every call goes to libmi355.so through the C ABI (no fallback).  Tests may
    substitute an object with the same five methods to exercise the sharding / collective logic on CPU."""
    name = "hip"

    def device(self, scene):
        torch = _torch()
        scene.handle()
        return torch.device("cuda", scene._device)

    def accumulate(self, scene, p, film):
        abi.check(abi.lib().mi355_render_accumulate(scene.handle(), C.byref(p), C.c_void_p(film.data_ptr()), _stream_ptr(film.device)))

    def accumulate_sensors(self, scene, p, sensors, film):
        arr = (C.c_uint32 * len(sensors))(*[int(k) for k in sensors])
        abi.check(abi.lib().mi355_render_accumulate_sensors(scene.handle(), C.byref(p), len(sensors), arr, C.c_void_p(film.data_ptr()), _stream_ptr(film.device)))

    def backward_sensors(self, scene, p, sensors, grad_ins, grads):
        arr = (C.c_uint32 * len(sensors))(*[int(k) for k in sensors])
        ptrs = (C.c_void_p * len(sensors))(*[None if g is None else g.data_ptr() for g in grad_ins])
        abi.check(abi.lib().mi355_render_backward_sensors(scene.handle(), C.byref(p), len(sensors), arr, ptrs, C.c_void_p(grads.data_ptr()), _stream_ptr(grads.device)))

    def develop(self, scene, film, img, sensor=0):
        abi.check(abi.lib().mi355_film_develop(scene.handle(), int(sensor), C.c_void_p(film.data_ptr()), C.c_void_p(img.data_ptr()), _stream_ptr(film.device)))

    def splat_weights(self, scene, p, weight):
        abi.check(abi.lib().mi355_prb_splat_weights(scene.handle(), C.byref(p), C.c_void_p(weight.data_ptr()), _stream_ptr(weight.device)))

    def backward(self, scene, p, grad_in, weight, grads):
        abi.check(abi.lib().mi355_render_backward(scene.handle(), C.byref(p), C.c_void_p(grad_in.data_ptr()),
                                                  C.c_void_p(weight.data_ptr()) if weight is not None else None,
                                                  C.c_void_p(grads.data_ptr()), _stream_ptr(grads.device)))

    def param_total(self, scene):
        return scene_param_total(scene)

    def enable_grads(self, scene, params):
        """dr.backward() only visits the parameters whose gradients are enabled: when some geometry / IOR parameter of `params` carries a
        gradient request (a tensor with requires_grad), the geometry adjoint fills the slots of exactly those and skips the others -- the
        film plane and the housing of a lens-design scene are meshes too, and every light path ends on the film plane's two triangles
        (166 ms of same-address float atomics per pass on the doublet).  With no request at all every slot is filled (explicit
        `geometry=True` on plain tensors)."""
        geo = [k for k, e in params._entries.items() if e[0] in (abi.PARAM_VERTEX_POSITIONS, abi.PARAM_BSDF_IOR)]
        wanted = {k for k in geo if getattr(params._values[k], "requires_grad", False)}
        state = getattr(scene, "_grad_enabled", None)
        if state is None:
            state = scene._grad_enabled = {}
        for k in geo:
            kind, index, _ = params._entries[k]
            en = (not wanted) or any(getattr(params._values[q], "requires_grad", False) for q in geo if params._entries[q][:2] == (kind, index))
            if state.get((kind, index), True) != en:
                abi.check(abi.lib().mi355_scene_param_enable_grad(scene.handle(), kind, index, 1 if en else 0))
                state[(kind, index)] = en

    def synchronize(self, device):
        _torch().cuda.current_stream(device).synchronize()


class Integrator:
    """Common part of MonteCarloIntegrator (integrator.cpp:505-522) and ADIntegrator (common.py:20-60)."""
    integrator_id = abi.INTEGRATOR_PATH
    name = "path"

    def __init__(self, props=None):
        props = dict(props or {})
        md = int(props.get("max_depth", self.default_max_depth))
        if md < 0 and md != -1:
            raise PluginError('"max_depth" must be set to -1 (infinite) or a value >= 0')
        self.max_depth = md
        self.rr_depth = int(props.get("rr_depth", 5))
        if self.rr_depth <= 0:
            raise PluginError('"rr_depth" must be set to a value greater than zero!')
        self.hide_emitters = bool(props.get("hide_emitters", False))
        self.samples_per_pass = int(props.get("samples_per_pass", -1))
        self.mode = abi.MODE_FUSED
        self.timing = False
        self.backend = HipBackend()

    default_max_depth = -1

    def aov_names(self):
        return []

    def cancel(self):
        """Integrator::cancel (integrator.h:293-297, integrator.cpp:90-92).  A render is one kernel launch here, so there is nothing to
        interrupt half-way; the flag is honoured between the launches of host-side loops (tune(), optimisation examples)."""
        self._stop = True

    def should_stop(self):
        return getattr(self, "_stop", False)

    def lane_grid(self, film):
        """Pixel grid the lanes are laid over: lane = pixel * spp + s -- the crop window of a sampling integrator, grown by the
        filter's border on every side when the film samples its border (integrator.cpp:133-134)."""
        b = film.border_size if film.sample_border else 0
        return film.crop_width + 2 * b, film.crop_height + 2 * b

    # ---- parameter block for the C ABI -----------------------------------------------------------
    def _params(self, scene, seed, spp, shard=None, spp_shard=None, sensor=0):
        sensor = scene.sensor_index(sensor)
        se = scene.sensors()[sensor]; f = se.film
        p = abi.RenderParams()
        p.sensor = sensor                                               # the `Sensor *` argument of Integrator::render (integrator.h:74-97)
        p.seed = (int(seed) + se.base_seed) & 0xffffffff                # sampler.cpp:120: m_base_seed + seed
        p.spp = int(spp)
        p.max_depth = self.max_depth; p.rr_depth = self.rr_depth
        p.flags = (abi.FLAG_HIDE_EMITTERS if self.hide_emitters else 0) | (abi.FLAG_TIMING if self.timing else 0) | (abi.FLAG_REGEN if getattr(self, "regen", False) else 0) | (abi.FLAG_TWO_PASS if getattr(self, "two_pass", False) else 0) | (abi.FLAG_DIRECT_SPLAT if getattr(self, "direct_splat", False) else 0) | (abi.FLAG_NO_LANE_SORT if getattr(self, "no_lane_sort", False) else 0)
        p.mode = self.mode; p.integrator = self.integrator_id
        spass = int(getattr(self, "samples_per_pass", -1))
        if spass not in (-1, 0, 0xffffffff) and int(spp) > 0:           # integrator.cpp:565-573
            per = min(spass, int(spp))
            if int(spp) % per != 0:
                raise PluginError("sample_count (%d) must be a multiple of samples_per_pass (%d)." % (int(spp), per))
            p.passes = int(spp) // per
        gw, gh = self.lane_grid(f); npix = gw * gh
        # more than 2^32 samples in one wavefront: the reference splits the job into passes by itself (integrator.cpp:247-265, :681-698)
        per_pass = int(spp) // max(int(p.passes), 1)
        if npix * per_pass > 0xffffffff and self.integrator_id != abi.INTEGRATOR_PRB:
            per_pass //= (npix * per_pass + 0xffffffff - 1) // 0xffffffff
            while per_pass > 1 and int(spp) % per_pass:
                per_pass -= 1
            p.passes = int(spp) // max(per_pass, 1)
        if shard is not None:
            p.pixel_begin, p.pixel_count = shard[0], shard[1]
            if len(shard) == 4:
                p.pixel_tile, p.pixel_tile_stride = shard[2], shard[3]
            if p.pixel_count == 0:
                p.pixel_begin, p.pixel_count = npix, 0
                p.pixel_tile = p.pixel_tile_stride = 0
        if spp_shard is not None:
            p.spp_begin, p.spp_count = spp_shard
        return p

    def _resolve_spp(self, scene, spp, sensor=0):
        return int(spp) if spp else scene.sensors()[scene.sensor_index(sensor)].sample_count      # integrator.cpp:140-141: the sensor's sampler

    # ---- Integrator::render -----------------------------------------------------------------------
    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True, shard=None, group=None):
        """Returns the developed image (H x W x 3, torch, on the scene's device); with develop=False the raw
        RGBW accumulator.  `shard=(rank, world)`: this process renders its pixel band and the accumulators are
        summed over `group` (torch.distributed) before develop."""
        torch = _torch()
        be = self.backend
        dev = be.device(scene)
        sensor = scene.sensor_index(sensor)
        f = scene.sensors()[sensor].film
        spp = self._resolve_spp(scene, spp, sensor)
        film = torch.zeros((f.crop_height, f.crop_width, 4), dtype=torch.float32, device=dev)
        band = None
        if shard is not None:
            band = shard_of(f, shard[0], shard[1], grid=self.lane_grid(f))
        p = self._params(scene, seed, spp, shard=band, sensor=sensor)
        if band is None or band[1] > 0:
            be.accumulate(scene, p, film)
        if shard is not None and shard[1] > 1:
            import torch.distributed as dist
            dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
        if not develop:
            return film
        img = torch.empty((f.crop_height, f.crop_width, 3), dtype=torch.float32, device=dev)
        be.develop(scene, film, img, sensor)
        if evaluate:
            be.synchronize(dev)
        return img

    # ---- Integrator::render_backward ---------------------------------------------------------------
    def render_backward(self, scene, params, grad_in, sensor=0, seed=0, spp=0, shard=None, group=None):
        """Accumulates d(loss)/d(theta) for every differentiable scene parameter into `params.grad_buffer`
        (and returns it).  Default implementation of the reference (integrator.cpp:65-84) records the whole
        primal with AD; the path integrator here shares PRB's adjoint kernels instead."""
        torch = _torch()
        be = self.backend
        dev = be.device(scene)
        sensor = scene.sensor_index(sensor)
        f = scene.sensors()[sensor].film
        spp = self._resolve_spp(scene, spp, sensor)
        grad_in = torch.as_tensor(grad_in, dtype=torch.float32, device=dev).contiguous()
        if tuple(grad_in.shape) != (f.crop_height, f.crop_width, 3):
            raise ValueError("grad_in must have shape (H, W, 3)")
        band = shard_of(f, shard[0], shard[1], grid=self.lane_grid(f)) if shard is not None else None
        p = self._params(scene, seed, spp, shard=band, sensor=sensor)
        p.integrator = abi.INTEGRATOR_PRB
        weight = torch.zeros((f.crop_height, f.crop_width), dtype=torch.float32, device=dev)
        active = band is None or band[1] > 0
        multi = shard is not None and shard[1] > 1
        if active:
            be.splat_weights(scene, p, weight)
        if multi:
            import torch.distributed as dist
            dist.all_reduce(weight, op=dist.ReduceOp.SUM, group=group)      # every rank needs the same W[p] (common.py:741-758)
        grads = params.grad_buffer(dev) if params is not None else torch.zeros(max(be.param_total(scene), 1), dtype=torch.float32, device=dev)
        local = torch.zeros_like(grads) if multi else grads
        if active:
            be.backward(scene, p, grad_in, weight, local)
        if multi:
            import torch.distributed as dist
            dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group)       # the ONE collective of the adjoint data path
            grads += local
        return grads

    # ---- Integrator::render_forward -----------------------------------------------------------------
    def render_forward(self, scene, params, sensor=0, seed=0, spp=0, shard=None, group=None):
        """Forward-mode derivative of the render (RBIntegrator.render_forward, common.py:498-629): returns the gradient image
        (H x W x 3) for the parameter tangents set with `params.set_tangent(key, value)` -- the counterpart of dr.set_grad() on a
        scene parameter.  With no tangent set the image is zero, as in the reference."""
        torch = _torch()
        be = self.backend
        dev = be.device(scene)
        sensor = scene.sensor_index(sensor)
        f = scene.sensors()[sensor].film
        spp = self._resolve_spp(scene, spp, sensor)
        if params is None:
            raise PluginError("render_forward needs the SceneParameters carrying the tangents")
        tangent = params.tangent_buffer(dev)
        film = torch.zeros((f.crop_height, f.crop_width, 4), dtype=torch.float32, device=dev)
        band = shard_of(f, shard[0], shard[1], grid=self.lane_grid(f)) if shard is not None else None
        p = self._params(scene, seed, spp, shard=band, sensor=sensor)
        p.integrator = abi.INTEGRATOR_PRB
        if band is None or band[1] > 0:
            abi.check(abi.lib().mi355_render_forward(scene.handle(), C.byref(p), C.c_void_p(tangent.data_ptr()), C.c_void_p(film.data_ptr()), _stream_ptr(dev)))
        if shard is not None and shard[1] > 1:
            import torch.distributed as dist
            dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group)
        img = torch.empty((f.crop_height, f.crop_width, 3), dtype=torch.float32, device=dev)
        be.develop(scene, film, img, sensor)
        return img

    def counters(self, scene, reset=False):
        c = abi.Counters()
        abi.check(abi.lib().mi355_get_counters(scene.handle(), C.byref(c), 1 if reset else 0))
        return {k: getattr(c, k) for k, _ in abi.Counters._fields_}

    def lane_radiance(self, scene, seed, spp, lane_begin, count, sensor=0):
        """Per-lane (r,g,b,valid) of global lanes [lane_begin, lane_begin+count) -- check_vectorization hook."""
        torch = _torch()
        h = scene.handle(); dev = torch.device("cuda", scene._device)
        out = torch.empty((count, 4), dtype=torch.float32, device=dev)
        p = self._params(scene, seed, self._resolve_spp(scene, spp, sensor), sensor=sensor)
        abi.check(abi.lib().mi355_debug_lane_radiance(h, C.byref(p), int(lane_begin), int(count), C.c_void_p(out.data_ptr()), _stream_ptr(dev)))
        torch.cuda.current_stream(dev).synchronize()
        return out


class PathIntegrator(Integrator):
    """src/integrators/path.cpp"""
    integrator_id = abi.INTEGRATOR_PATH
    name = "path"
    default_max_depth = -1


    def tune(self, scene, spp=0, seed=0, frames=2):
        """Picks the forward kernel variant for THIS scene and sample count by timing them on the device (HIP events around the
        kernel): in-workgroup re-packing (the default; best when most paths live long, e.g. the Cornell box: 9.7 ms against 11.2 ms)
        or path regeneration (`regen`, needs >= 128 spp; best when paths die early and unevenly, e.g. the open Veach scene: 105 ms
        against 123 ms).  Same samples and same film either way.  Returns {variant: kernel ms} and leaves the faster one selected."""
        spp = self._resolve_spp(scene, spp)
        saved = (self.timing, getattr(self, "regen", False), self.mode)
        result = {}
        for name, regen in (("repack", False), ("regen", True)):
            if regen and spp < 128:
                continue
            self.timing, self.regen, self.mode = True, regen, abi.MODE_FUSED
            self.render(scene, seed=seed, spp=spp); self.counters(scene, reset=True)
            for i in range(frames):
                self.render(scene, seed=seed + i, spp=spp)
            c = self.counters(scene, reset=True)
            result[name] = c["kernel_ms"] / max(c["kernel_launches"], 1)
        self.timing, self.regen, self.mode = saved[0], min(result, key=result.get) == "regen", saved[2]
        return result


class PRBIntegrator(Integrator):
    """src/python/python/ad/integrators/prb.py (max_depth default 6: common.py:31)"""
    integrator_id = abi.INTEGRATOR_PRB
    name = "prb"
    default_max_depth = 6


class ParticleTracerIntegrator(Integrator):
    """src/integrators/ptracer.cpp under AdjointIntegrator::render (integrator.cpp:545-745): light paths connected to the
    (perspective) sensor.  Lanes cover the whole FILM (not the crop window); rr_depth default 5, max_depth -1."""
    integrator_id = abi.INTEGRATOR_PTRACER
    name = "ptracer"
    default_max_depth = -1

    def lane_grid(self, film):
        return film.width, film.height

    def render(self, scene, sensor=0, seed=0, spp=0, develop=True, evaluate=True, shard=None, group=None):
        return super().render(scene, sensor, seed, spp, develop, evaluate, shard, group)

    def sensors_share_light_paths(self, scene, sensors):
        """True when mi355_render_accumulate_sensors can serve `sensors` with ONE light-traced pass: perspective cameras, one film size without a
        crop window, one reconstruction filter (the light paths of AdjointIntegrator::render depend on the seed and the film size only)."""
        if not hasattr(self.backend, "accumulate_sensors") or not (1 <= len(sensors) <= abi.MAX_SENSORS_PER_PASS):
            return False
        ss = [scene.sensors()[scene.sensor_index(k)] for k in sensors]
        f0 = ss[0].film
        for se in ss:
            f = se.film
            if se.type != abi.SENSOR_PERSPECTIVE or (f.width, f.height) != (f0.width, f0.height) or (f.crop_width, f.crop_height, f.crop_offset_x, f.crop_offset_y) != (f.width, f.height, 0, 0):
                return False
            if (f.rfilter, f.rfilter_param) != (f0.rfilter, f0.rfilter_param) or (se.shutter_open_time > 0) != (ss[0].shutter_open_time > 0):
                return False
        return True

    def render_sensors(self, scene, sensors, seed=0, spp=0, develop=True, evaluate=True, shard=None, group=None):
        """Integrator.render(scene, sensor=k, seed, spp) for every k of `sensors` from ONE pass over the light paths (mi355_render_accumulate_sensors):
        design_problem.py:525-537 renders each field with the same seed, so the fields share their light paths and differ in the sensor connection
        only.  Returns the list of images (or of RGBW accumulators), equal to the separate renders up to the order of the float additions."""
        torch = _torch()
        be = self.backend
        dev = be.device(scene)
        sensors = [scene.sensor_index(k) for k in sensors]
        if not self.sensors_share_light_paths(scene, sensors):
            return [self.render(scene, k, seed, spp, develop, evaluate, shard, group) for k in sensors]
        f = scene.sensors()[sensors[0]].film
        spp = self._resolve_spp(scene, spp, sensors[0])
        pitch, x0, virt_w = C.c_uint32(), C.c_uint32(), C.c_uint32()
        abi.check(abi.lib().mi355_multi_film_layout(f.width, len(sensors), C.byref(pitch), C.byref(x0), C.byref(virt_w)))
        stack = torch.zeros((f.height, virt_w.value, 4), dtype=torch.float32, device=dev)
        band = shard_of(f, shard[0], shard[1], grid=self.lane_grid(f)) if shard is not None else None
        if band is None or band[1] > 0:
            be.accumulate_sensors(scene, self._params(scene, seed, spp, shard=band, sensor=sensors[0]), sensors, stack)
        if shard is not None and shard[1] > 1:                               # the ranks' partial films add up in the virtual film
            import torch.distributed as dist
            dist.all_reduce(stack, op=dist.ReduceOp.SUM, group=group)
        out = []
        for k, sensor in enumerate(sensors):
            film = stack[:, k * pitch.value + x0.value: k * pitch.value + x0.value + f.width, :].contiguous()
            if not develop:
                out.append(film); continue
            img = torch.empty((f.height, f.width, 3), dtype=torch.float32, device=dev)
            be.develop(scene, film, img, sensor)
            out.append(img)
        if evaluate:
            be.synchronize(dev)
        return out

    def render_backward_sensors(self, scene, params, grad_ins, sensors, seed=0, spp=0, shard=None, group=None):
        """render_backward(scene, params, grad_ins[k], sensor=sensors[k], seed, spp) summed over k from ONE walk and ONE sweep over the light paths
        (mi355_render_backward_sensors) -- when the sensors share their light paths and only geometry / IOR parameters ask for a gradient (the
        lens-design step); otherwise the calls are made one after the other.  `grad_ins[k]` may be None."""
        torch = _torch()
        be = self.backend
        sensors = [scene.sensor_index(k) for k in sensors]
        one_pass = (hasattr(be, "backward_sensors") and self.sensors_share_light_paths(scene, sensors) and params is not None
                    and params.wants_geometry() and not params.wants_colour())
        if not one_pass:
            grads = None
            for k, g in zip(sensors, grad_ins):
                if g is not None:
                    grads = self.render_backward(scene, params, g, k, seed=seed, spp=spp, shard=shard, group=group)
            return grads if grads is not None else params.grad_buffer(be.device(scene))
        dev = be.device(scene)
        f = scene.sensors()[sensors[0]].film
        spp = self._resolve_spp(scene, spp, sensors[0])
        gs = []
        for g in grad_ins:
            if g is None:
                gs.append(None); continue
            g = torch.as_tensor(g, dtype=torch.float32, device=dev).contiguous()
            if tuple(g.shape) != (f.crop_height, f.crop_width, 3):
                raise ValueError("grad_in must have shape (H, W, 3)")
            gs.append(g)
        band = shard_of(f, shard[0], shard[1], grid=self.lane_grid(f)) if shard is not None else None
        p = self._params(scene, seed, spp, shard=band, sensor=sensors[0])
        p.flags |= abi.FLAG_GEOMETRY_GRADS | abi.FLAG_SKIP_COLOUR_GRADS
        if hasattr(be, "enable_grads"):
            be.enable_grads(scene, params)
        grads = params.grad_buffer(dev)
        multi = shard is not None and shard[1] > 1
        local = torch.zeros_like(grads) if multi else grads
        if band is None or band[1] > 0:
            be.backward_sensors(scene, p, sensors, gs, local)
        if multi:
            import torch.distributed as dist
            dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group)
            grads += local
        return grads

    def render_forward(self, *a, **k):
        raise PluginError("ptracer: render_forward over the colour parameters is not built (render_backward is; for mesh vertices see render_forward_geometry)")

    def render_forward_design(self, scene, shape=None, vertex_tangent=None, bsdf=None, d_int_ior=0.0, d_ext_ior=0.0, sensor=0, seed=0, spp=0):
        """Forward-mode derivative of the light-traced image with respect to ONE design scalar of a lens -- what the fork's lens-design
        notebooks obtain from Dr.Jit's AD through `ptracer` (lens.py:505-565, design_problem.py:403-428).  The scalar's tangent can have a
        geometric part (`shape`: id or index of a mesh, `vertex_tangent`: V x 3 = d(position)/d(theta)) and a material part (`bsdf`: id or
        index of a dielectric, `d_int_ior` / `d_ext_ior`).  Returns the gradient image (H x W x 3)."""
        torch = _torch()
        be = self.backend
        dev = be.device(scene)
        sensor = scene.sensor_index(sensor)
        f = scene.sensors()[sensor].film
        spp = self._resolve_spp(scene, spp, sensor)
        t = abi.DesignTangent(); t.shape = abi.NONE; t.bsdf = abi.NONE; t.d_vertex_tangent = None
        vt = None
        if shape is not None:
            index = shape if isinstance(shape, int) else [i for i, sh in enumerate(scene.shapes) if sh.id == shape][0]
            nv = scene.shapes[index].vertex_positions.shape[0] if scene.shapes[index].vertex_positions is not None else 0
            vt = torch.as_tensor(vertex_tangent, dtype=torch.float32, device=dev).contiguous()
            if tuple(vt.shape) != (nv, 3):
                raise ValueError("vertex_tangent must have shape (%d, 3)" % nv)
            t.shape = index; t.d_vertex_tangent = vt.data_ptr()
        if bsdf is not None:
            t.bsdf = bsdf if isinstance(bsdf, int) else [i for i, b in enumerate(scene.bsdfs) if b.id == bsdf][0]
            t.d_int_ior = float(d_int_ior); t.d_ext_ior = float(d_ext_ior)
        film = torch.zeros((f.crop_height, f.crop_width, 4), dtype=torch.float32, device=dev)
        p = self._params(scene, seed, spp, sensor=sensor)
        abi.check(abi.lib().mi355_ptracer_forward_design(scene.handle(), C.byref(p), C.byref(t), C.c_void_p(film.data_ptr()), _stream_ptr(dev)))
        img = torch.empty((f.crop_height, f.crop_width, 3), dtype=torch.float32, device=dev)
        be.develop(scene, film, img, sensor)
        return img

    def render_forward_geometry(self, scene, shape, vertex_tangent, sensor=0, seed=0, spp=0):
        """render_forward_design with a vertex tangent only."""
        return self.render_forward_design(scene, shape=shape, vertex_tangent=vertex_tangent, sensor=sensor, seed=seed, spp=spp)

    def render_backward(self, scene, params, grad_in, sensor=0, seed=0, spp=0, shard=None, group=None, geometry=None, colour=None):
        """Integrator::render_backward (integrator.cpp:65-84) over ptracer.cpp: gradients of every differentiable parameter of this build
        (emitter radiances, diffuse reflectances, reflectance textures) by path replay of the light paths; the reference's own test
        value for the radiance (test_ptracer.py:204-225) is reproduced.  When `params` holds radiance keys only (params.keep), the
        cheaper single-walk kernel runs.  `geometry`: also fill the `vertex_positions` and IOR slots (the reverse-mode geometry adjoint,
        MI355_FLAG_GEOMETRY_GRADS); None = when one of those parameters asks for a gradient (a tensor with requires_grad).  `colour`:
        fill the reflectance / texture / radiance slots; None = yes, unless gradients are requested for geometry / IOR parameters ONLY
        (the lens-design step: dr.backward() would not visit the colours either) -- then the colour walk is skipped."""
        torch = _torch()
        if geometry is None:
            geometry = params is not None and params.wants_geometry()
        if colour is None:
            colour = not (geometry and params is not None and params.wants_geometry() and not params.wants_colour())
        radiance_only = (not geometry) and params is not None and all(e[0] == abi.PARAM_EMITTER_RADIANCE for e in params._entries.values())
        be = self.backend
        dev = be.device(scene)
        sensor = scene.sensor_index(sensor)
        f = scene.sensors()[sensor].film
        spp = self._resolve_spp(scene, spp, sensor)
        grad_in = torch.as_tensor(grad_in, dtype=torch.float32, device=dev).contiguous()
        if tuple(grad_in.shape) != (f.crop_height, f.crop_width, 3):
            raise ValueError("grad_in must have shape (H, W, 3)")
        band = shard_of(f, shard[0], shard[1], grid=self.lane_grid(f)) if shard is not None else None
        p = self._params(scene, seed, spp, shard=band, sensor=sensor)
        if radiance_only:
            p.flags |= abi.FLAG_EMITTER_GRADS_ONLY
        if geometry:
            p.flags |= abi.FLAG_GEOMETRY_GRADS
            if not colour:
                p.flags |= abi.FLAG_SKIP_COLOUR_GRADS
            if params is not None and hasattr(be, "enable_grads"):
                be.enable_grads(scene, params)
        grads = params.grad_buffer(dev) if params is not None else torch.zeros(max(be.param_total(scene), 1), dtype=torch.float32, device=dev)
        multi = shard is not None and shard[1] > 1
        local = torch.zeros_like(grads) if multi else grads
        if band is None or band[1] > 0:
            be.backward(scene, p, grad_in, None, local)                    # (the normalised splat has no weight image)
        if multi:
            import torch.distributed as dist
            dist.all_reduce(local, op=dist.ReduceOp.SUM, group=group)
            grads += local
        return grads


class BasicPRBIntegrator(Integrator):
    """The fork's `prb_basic` (src/python/python/ad/integrators/prb_basic.py) as far as the path uses it: `trace()`, the deterministic
    geometric ray tracer the lens-design scripts place their cameras with (design_problem.py:183-203, :415-417) and check their lens
    meshes against (:430-472).  Its `sample()` -- a path tracer without next-event estimation -- is not on the path: the scripts render
    with `ptracer` (design_problem.py:405-422)."""
    integrator_id = abi.INTEGRATOR_PRB
    name = "prb_basic"
    default_max_depth = 6

    def render(self, *a, **k):
        raise PluginError("prb_basic: only trace() is built (the lens-design scripts render with `ptracer`, design_problem.py:405-422)")

    render_backward = render_forward = render

    def trace(self, scene, ray, num_lenses):
        """BasicPRBIntegrator.trace(scene, ray, num_lenses) (prb_basic.py:46-140).  `ray`: a Ray-like object with `.o` and `.d` (n x 3 or 3)
        or an n x 6 array (origin, direction); rays start with maxt = Largest like mi.Ray3f(o, d).  Returns (penultimate_o, penultimate_d,
        valid, p) as torch tensors on the scene's device: origin / direction of the ray that produced the last interaction, the reference's
        validity flag -- si.is_valid() & dr.allclose(0, si.p.z): as in the reference ONE verdict of the closeness test for the whole batch
        (:138), ANDed with each ray's own hit flag -- and the last interaction's position (the film point)."""
        torch = _torch()
        dev = self.backend.device(scene)
        if hasattr(ray, "o") and hasattr(ray, "d"):
            o = torch.as_tensor(np.asarray(ray.o, dtype=np.float32)).reshape(-1, 3); d = torch.as_tensor(np.asarray(ray.d, dtype=np.float32)).reshape(-1, 3)
        else:
            a = torch.as_tensor(np.asarray(ray, dtype=np.float32)).reshape(-1, 6); o, d = a[:, :3], a[:, 3:]
        n = o.shape[0]
        rays = torch.zeros((n, 8), dtype=torch.float32)
        rays[:, 0:3] = o; rays[:, 3:6] = d; rays[:, 6] = 3.4028234663852886e38       # dr::Largest<Float>: Ray3f's default maxt (ray.h:35)
        rays = rays.to(dev)
        out = torch.empty((n, 12), dtype=torch.float32, device=dev)
        abi.check(abi.lib().mi355_trace_rays(scene.handle(), C.c_void_p(rays.data_ptr()), n, int(num_lenses), int(self.max_depth),
                                              C.c_void_p(out.data_ptr()), _stream_ptr(dev)))
        torch.cuda.current_stream(dev).synchronize()
        p = out[:, 6:9]
        hit = out[:, 9] != 0
        z = p[:, 2].abs()
        close = bool((z <= 1e-8 + 1e-5 * z).all().item())          # dr.allclose(0.0, si.p.z): |0 - z| <= atol + rtol |z|, over ALL lanes
        self.penultimate_o, self.penultimate_d, self.valid = out[:, 0:3], out[:, 3:6], hit & close
        return self.penultimate_o, self.penultimate_d, self.valid, p


def make_integrator(props):
    t = props.get("type")
    if t == "prb_basic":
        return BasicPRBIntegrator(props)
    if t == "path":
        return PathIntegrator(props)
    if t == "prb":
        return PRBIntegrator(props)
    if t == "ptracer":
        return ParticleTracerIntegrator(props)
    raise PluginError('integrator plugin "%s" is outside the path (path, prb, ptracer, prb_basic)' % t)


# ------------------------------------------------------------------------------------------------
# traverse / SceneParameters: util.py:13-350
# ------------------------------------------------------------------------------------------------
def scene_param_layout(scene):
    h = scene.handle()
    cap = 4096
    arr = (abi.ParamInfo * cap)(); total = C.c_uint64()
    n = abi.check(abi.lib().mi355_scene_param_layout(h, arr, cap, C.byref(total)))
    return [(arr[i].kind, arr[i].index, arr[i].offset, arr[i].size) for i in range(n)], total.value


def scene_param_total(scene):
    return scene_param_layout(scene)[1]


class SceneParameters:
    """Dictionary of differentiable scene parameters; `update()` pushes edits to the device replica
    (SceneParameters.update -> parameters_changed, util.py:185-232).  Values are torch tensors on the
    scene's device; gradients written by render_backward are read with `grad(key)`."""

    def __init__(self, scene):
        torch = _torch()
        self.scene = scene
        self._entries = {}     # key -> (kind, index, shape)
        self._comp = {}        # key -> component of a shared engine parameter: 0 / 1 = int_ior / ext_ior of MI355_PARAM_BSDF_IOR, None = no engine slot (V_d)
        self._values = {}
        self._dirty = set()
        self._pushed = {}      # key -> (tensor object, its in-place version) of the last upload
        self._grads = None
        self._tangents = None
        scene.handle()
        dev = torch.device("cuda", scene._device)
        layout, total = scene_param_layout(scene)
        self._layout = {(k, i): (off, size) for k, i, off, size in layout}
        self._total = total
        for i, b in enumerate(scene.bsdfs):
            if b.texture is None and (abi.PARAM_BSDF_REFLECTANCE, i) in self._layout:     # diffuse reflectances (the engine's layout)
                self._add("%s.reflectance.value" % b.id, abi.PARAM_BSDF_REFLECTANCE, i, torch.tensor(b.reflectance, device=dev))
        for i, t in enumerate(scene.textures):
            self._add("%s.reflectance.data" % t.id, abi.PARAM_TEXTURE_DATA, i, torch.tensor(t.data, device=dev))
        for i, e in enumerate(scene.emitters):
            self._add("%s.emitter.radiance.value" % e.shape.id, abi.PARAM_EMITTER_RADIANCE, i, torch.tensor(e.radiance, device=dev))
        for i, sh in enumerate(scene.shapes):          # Mesh::traverse, mesh.cpp:1602-1611: `vertex_positions` as a flat array
            if sh.type == abi.SHAPE_MESH and (abi.PARAM_VERTEX_POSITIONS, i) in self._layout:
                self._add("%s.vertex_positions" % sh.id, abi.PARAM_VERTEX_POSITIONS, i, torch.tensor(sh.vertex_positions.reshape(-1), device=dev))
        for i, b in enumerate(scene.bsdfs):
            if b.type != abi.BSDF_DIELECTRIC or (abi.PARAM_BSDF_IOR, i) not in self._layout:
                continue
            if b.dispersive is not None:               # dispersive_dielectric.cpp:236-239
                for comp, name in ((0, "int_ior_d"), (1, "ext_ior_d"), (None, "int_V_d"), (None, "ext_V_d")):
                    self._add("%s.%s" % (b.id, name), abi.PARAM_BSDF_IOR, i, torch.tensor([b.dispersive[name]], dtype=torch.float32, device=dev), comp=comp)
            else:                                      # dielectric.cpp:236-239 exposes eta (non-differentiable); the two indices are this build's parameters
                self._add("%s.int_ior" % b.id, abi.PARAM_BSDF_IOR, i, torch.tensor([b.int_ior], dtype=torch.float32, device=dev), comp=0)
                self._add("%s.ext_ior" % b.id, abi.PARAM_BSDF_IOR, i, torch.tensor([b.ext_ior], dtype=torch.float32, device=dev), comp=1)
        # Sensor parameters (PerspectiveCamera::traverse, perspective.cpp:155-161; Film::traverse, film.cpp:100-104): host values, no gradient
        # slot -- the lens-design loop moves and zooms its cameras under dr.suspend_grad (design_problem.py:113-132)
        self._host = {}        # key -> (what, sensor index)
        for i, se in enumerate(scene.sensors()):
            self._host["%s.to_world" % se.id] = ("to_world", i)
            if se.type == abi.SENSOR_PERSPECTIVE:
                self._host["%s.x_fov" % se.id] = ("x_fov", i)
            for what in ("size", "crop_size", "crop_offset"):
                self._host["%s.film.%s" % (se.id, what)] = (what, i)

    def _add(self, key, kind, index, value, comp=-1):
        self._entries[key] = (kind, index, tuple(value.shape))
        self._values[key] = value
        if comp != -1:
            self._comp[key] = comp

    def wants_geometry(self):
        """True when a geometry / IOR parameter of this set carries a gradient request (a tensor with requires_grad), i.e. when the
        particle tracer's render_backward must run its geometry adjoint too."""
        return any(self._entries[k][0] in (abi.PARAM_VERTEX_POSITIONS, abi.PARAM_BSDF_IOR) and getattr(self._values[k], "requires_grad", False)
                   for k in self._entries)

    def wants_colour(self):
        """True when a reflectance / texture / radiance parameter of this set carries a gradient request (a tensor with requires_grad)."""
        return any(self._entries[k][0] in (abi.PARAM_BSDF_REFLECTANCE, abi.PARAM_TEXTURE_DATA, abi.PARAM_EMITTER_RADIANCE) and getattr(self._values[k], "requires_grad", False)
                   for k in self._entries)

    def keep(self, keys):
        """SceneParameters.keep (util.py:234-258): keep the parameters whose key matches one of `keys` (regular expressions, matched
        from the start of the key as re.match does)."""
        import re
        if not isinstance(keys, list):
            keys = [keys]
        regexps = [re.compile(k).match for k in keys]
        wanted = [k for k in self.keys() if any(r(k) for r in regexps)]
        for k in list(self._entries):
            if k not in wanted:
                del self._entries[k]; del self._values[k]; self._dirty.discard(k); self._comp.pop(k, None)
        for k in list(self._host):
            if k not in wanted:
                del self._host[k]; self._dirty.discard(k)

    def keys(self): return list(self._entries.keys()) + list(self._host.keys())
    def __contains__(self, k): return k in self._entries or k in self._host
    def __len__(self): return len(self._entries) + len(self._host)
    def __iter__(self): return iter(self.keys())
    def items(self): return [(k, self[k]) for k in self.keys()]

    def _get_host(self, key):
        what, i = self._host[key]; se = self.scene.sensors()[i]; f = se.film
        if what == "to_world":
            return se.to_world
        if what == "x_fov":
            return se.x_fov
        return {"size": (f.width, f.height), "crop_size": (f.crop_width, f.crop_height), "crop_offset": (f.crop_offset_x, f.crop_offset_y)}[what]

    def _set_host(self, key, value):
        """PerspectiveCamera::parameters_changed (perspective.cpp:163-171) / Film::parameters_changed (film.cpp:59-98), applied to the host
        object at once; the device replica follows at update()."""
        from .transforms import Transform4f
        what, i = self._host[key]; se = self.scene.sensors()[i]; f = se.film
        if what == "to_world":
            tw = value if isinstance(value, Transform4f) else Transform4f(np.asarray(value.detach().cpu() if hasattr(value, "detach") else value, dtype=np.float64).reshape(4, 4))
            if tw.has_scale():
                raise PluginError("Scale factors in the camera-to-world transformation are not allowed!")
            se.to_world = tw
        elif what == "x_fov":
            v = float(value.detach().reshape(-1)[0]) if hasattr(value, "detach") else float(np.asarray(value).reshape(-1)[0])
            if not 0.0 < v < 180.0:
                raise PluginError("The horizontal field of view must be in the range [0, 180]!")
            se.x_fov = float(np.float32(v))
        else:
            a, b = (int(x) for x in value)
            if what == "size":                       # Film::set_size (film.cpp:59-65): the crop window is reset to the whole film
                f.width, f.height = a, b; f.crop_width, f.crop_height, f.crop_offset_x, f.crop_offset_y = a, b, 0, 0
            elif what == "crop_size":
                f.crop_width, f.crop_height = a, b
            else:
                f.crop_offset_x, f.crop_offset_y = a, b
        self._dirty.add(key)

    def __getitem__(self, key):
        if key in self._host:
            return self._get_host(key)
        return self._values[key]

    def __setitem__(self, key, value):
        torch = _torch()
        if key in self._host:
            return self._set_host(key, value)
        if key not in self._entries:
            raise KeyError(key)
        kind, index, shape = self._entries[key]
        v = torch.as_tensor(value, dtype=torch.float32, device=self._values[key].device)
        if tuple(v.shape) != shape:
            raise ValueError("parameter %s must keep its shape %s" % (key, shape))
        self._values[key] = v
        self._dirty.add(key)

    def update(self, values=None, skip=()):
        """SceneParameters.update (util.py:185-232): push every assigned parameter to the device replica; returns the keys pushed.
        `skip`: keys the caller pushes itself (mi.render's differentiable parameters)."""
        if values is not None:
            for k, v in values.items():
                if k in self._entries or k in self._host:
                    self[k] = v
        out = sorted(k for k in self._dirty if k not in skip)
        sensors = set()
        for key in out:
            if key in self._host:
                sensors.add(self._host[key][1])
            else:
                self._push(key, self._values[key])
        for i in sorted(sensors):                      # one description per changed sensor, whatever combination of its keys was set
            self._push_sensor(i, [k for k in out if k in self._host and self._host[k][1] == i])
        self._dirty.difference_update(out)
        return out

    def _push_sensor(self, index, keys):
        """Hand a changed sensor to the device replica: `to_world` / `x_fov` alone go as parameters (MI355_PARAM_SENSOR_*), a film change
        replaces the sensor's description (mi355_scene_set_sensor).  Either way the replica -- geometry, BVH, scratch -- stays."""
        se = self.scene.sensors()[index]
        whats = {self._host[k][0] for k in keys}
        if self.scene._handle is None:
            return
        if whats <= {"to_world", "x_fov"}:
            if "to_world" in whats:
                m = np.ascontiguousarray(se.to_world.affine12(), dtype=np.float32)
                abi.check(abi.lib().mi355_scene_set_param(self.scene.handle(), abi.PARAM_SENSOR_TO_WORLD, index, m.ctypes.data_as(abi.f32p), 12))
            if "x_fov" in whats:
                v = np.array([se.x_fov], dtype=np.float32)
                abi.check(abi.lib().mi355_scene_set_param(self.scene.handle(), abi.PARAM_SENSOR_X_FOV, index, v.ctypes.data_as(abi.f32p), 1))
        else:
            self.scene.push_sensor(index)

    def _push(self, key, tensor):
        """Upload one parameter to the device replica and keep the host description coherent."""
        kind, index, _ = self._entries[key]
        # The same tensor object at the same in-place version was pushed last time: nothing to do.  (mi.render pushes its differentiable
        # parameters at every call -- their tensors may have been edited in place -- and a design iteration renders once per sensor:
        # without this the six lens meshes would be uploaded and the BVH refitted three times per iteration.)  The reference is kept so
        # that the identity test cannot be fooled by a recycled object.
        last = self._pushed.get(key)
        if last is not None and last[0] is tensor and last[1] == tensor._version:
            return
        self._pushed[key] = (tensor, tensor._version)
        v = tensor.detach().to(torch_float32()).contiguous()
        if kind == abi.PARAM_BSDF_IOR:                 # two keys (or four, `dispersive`) share the engine's (int_ior, ext_ior) pair
            b = self.scene.bsdfs[index]; name = key.rsplit(".", 1)[1]; val = float(v.reshape(-1)[0].item())
            if b.dispersive is not None:
                b.dispersive[name] = val
                b.int_ior = float(b.eval_ior(b.dispersive["int_ior_d"], b.dispersive["int_V_d"])); b.ext_ior = float(b.eval_ior(b.dispersive["ext_ior_d"], b.dispersive["ext_V_d"]))
            elif name == "int_ior":
                b.int_ior = val
            else:
                b.ext_ior = val
            pair = np.array([b.int_ior, b.ext_ior], dtype=np.float32)
            abi.check(abi.lib().mi355_scene_set_param(self.scene.handle(), kind, index, pair.ctypes.data_as(abi.f32p), 2))
            return
        abi.check(abi.lib().mi355_scene_set_param_device(self.scene.handle(), kind, index, C.c_void_p(v.data_ptr()), v.numel(),
                                                          _stream_ptr(v.device)))
        if kind == abi.PARAM_VERTEX_POSITIONS:         # Mesh::parameters_changed: the engine refits and recomputes the normals; the host copy follows lazily
            sh = self.scene.shapes[index]
            if v.data_ptr() == tensor.data_ptr():      # keep what was PUSHED: an in-place edit of the caller's tensor must not leak into the host mirror
                v = v.clone()
            sh._positions_device = v; sh._normals_stale = sh._vn is not None      # (no download here: the host copies catch up when somebody reads them)
            return
        host = v.cpu().numpy()
        if kind == abi.PARAM_BSDF_REFLECTANCE:
            self.scene.bsdfs[index].reflectance = host.copy()
        elif kind == abi.PARAM_EMITTER_RADIANCE:
            self.scene.emitters[index].radiance = host.copy()
        else:
            self.scene.textures[index].data = np.ascontiguousarray(host.reshape(self.scene.textures[index].data.shape))

    # ---- gradients ---------------------------------------------------------------------------------
    def grad_buffer(self, device):
        torch = _torch()
        if self._grads is None or self._grads.device != device:
            self._grads = torch.zeros(max(self._total, 1), dtype=torch.float32, device=device)
        return self._grads

    def _slot(self, key):
        """(offset, size) of a key inside the concatenated gradient buffer (None for a key without an engine slot)."""
        kind, index, _ = self._entries[key]
        off, size = self._layout[(kind, index)]
        if key in self._comp:
            return None if self._comp[key] is None else (off + self._comp[key], 1)
        return off, size

    def set_tangent(self, key, value):
        """dr.set_grad(params[key], value) for render_forward: the direction in parameter space along which the image is differentiated."""
        torch = _torch()
        kind, index, shape = self._entries[key]
        off, size = self._slot(key)
        if self._tangents is None:
            self._tangents = torch.zeros(max(self._total, 1), dtype=torch.float32, device=self._values[key].device)
        v = torch.as_tensor(value, dtype=torch.float32, device=self._tangents.device)
        self._tangents[off:off + size] = v.expand(shape).reshape(-1) if v.numel() != size else v.reshape(-1)

    def clear_tangents(self):
        self._tangents = None

    def tangent_buffer(self, device):
        torch = _torch()
        if self._tangents is None or self._tangents.device != device:
            self._tangents = torch.zeros(max(self._total, 1), dtype=torch.float32, device=device)
        return self._tangents

    def zero_grad(self):
        if self._grads is not None:
            self._grads.zero_()

    def grad(self, key):
        kind, index, shape = self._entries[key]
        torch = _torch()
        slot = self._slot(key)
        if self._grads is None or slot is None:        # (an Abbe number has no effect at the nominal 589.3 nm of an RGB build: gradient zero)
            return torch.zeros(shape, device=self._values[key].device)
        off, size = slot
        return self._grads[off:off + size].reshape(shape)


def traverse(scene):
    return SceneParameters(scene)


# ------------------------------------------------------------------------------------------------
# mi.render: util.py:410-523, as a torch.autograd.Function instead of dr.CustomOp
# ------------------------------------------------------------------------------------------------
def render(scene, params=None, sensor=0, integrator=None, seed=0, seed_grad=0, spp=0, spp_grad=0, shard=None, group=None):
    torch = _torch()
    if params is not None and not isinstance(params, SceneParameters):
        raise Exception('params should be an instance of mi.SceneParameter!')
    assert isinstance(scene, Scene)
    if integrator is None:
        integrator = scene.integrator()
    if integrator is None:
        raise Exception('No integrator specified! Add an integrator in the scene description or provide an integrator directly as argument.')
    if spp_grad == 0:
        spp_grad = spp
    if seed_grad == 0:
        seed_grad = sample_tea_32(seed, 1)[0]          # util.py:517
    elif seed_grad == seed:
        raise Exception('The primal and differential seed should be different to ensure unbiased gradient computation!')

    diff_keys = [k for k in (params.keys() if params is not None else []) if getattr(params[k], "requires_grad", False)]
    if params is not None:
        params.update(skip=diff_keys)      # (the differentiable ones are pushed by the render op itself, every time: their tensors may be edited in place)
    if not diff_keys:
        return integrator.render(scene, sensor, seed=seed, spp=spp, evaluate=False, shard=shard, group=group)     # util.py:376-384: develop=True, evaluate=False
    sensor = scene.sensor_index(sensor)

    class _RenderOp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *tensors):
            for k, t in zip(diff_keys, tensors):
                params._push(k, t); params._dirty.discard(k)
            # util.py:376-384: _RenderOp.eval renders with develop=True, evaluate=False -- the image is queued, not waited for, so the host goes
            # on to queue the loss and the next field's render while this one runs (Integrator.render's own default stays evaluate=True)
            return integrator.render(scene, sensor, seed=seed, spp=spp, evaluate=False, shard=shard, group=group)

        @staticmethod
        def backward(ctx, grad_out):
            params.zero_grad()
            integrator.render_backward(scene, params, grad_out.contiguous(), sensor, seed=seed_grad, spp=spp_grad,
                                       shard=shard, group=group)
            return tuple(params.grad(k).clone() for k in diff_keys)

    return _RenderOp.apply(*[params[k] for k in diff_keys])


def render_fields(scene, params, sensors, integrator=None, seed=0, seed_grad=0, spp=0, spp_grad=0, shard=None, group=None):
    """mi.render(scene, params, sensor=k, seed=seed, ...) for every k of `sensors`, as ONE differentiable operation: the primal images come from
    one pass over the light paths when the sensors allow it (ParticleTracerIntegrator.render_sensors), the adjoint is the sum of the sensors'
    render_backward calls.  Returns the tuple of images.  (design_problem.py:525-537 renders the fields of an iteration one after the other with
    one seed; the results are those of that loop.)"""
    torch = _torch()
    if integrator is None:
        integrator = scene.integrator()
    sensors = [scene.sensor_index(k) for k in sensors]
    if not hasattr(integrator, "render_sensors") or not integrator.sensors_share_light_paths(scene, sensors):
        return tuple(render(scene, params, sensor=k, integrator=integrator, seed=seed, seed_grad=seed_grad, spp=spp, spp_grad=spp_grad, shard=shard, group=group) for k in sensors)
    if spp_grad == 0:
        spp_grad = spp
    if seed_grad == 0:
        seed_grad = sample_tea_32(seed, 1)[0]          # util.py:517
    elif seed_grad == seed:
        raise Exception('The primal and differential seed should be different to ensure unbiased gradient computation!')
    diff_keys = [k for k in (params.keys() if params is not None else []) if getattr(params[k], "requires_grad", False)]
    if params is not None:
        params.update(skip=diff_keys)
    if not diff_keys:
        return tuple(integrator.render_sensors(scene, sensors, seed=seed, spp=spp, evaluate=False, shard=shard, group=group))

    class _FieldsOp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, *tensors):
            for k, t in zip(diff_keys, tensors):
                params._push(k, t); params._dirty.discard(k)
            return tuple(integrator.render_sensors(scene, sensors, seed=seed, spp=spp, evaluate=False, shard=shard, group=group))

        @staticmethod
        def backward(ctx, *grad_outs):
            params.zero_grad()
            # the fields differentiate with ONE seed as well (util.py:517): one walk and one sweep over the light paths when only geometry / IOR
            # parameters ask for gradients, else one render_backward per field -- either way the gradient buffer ends up with the sum over the fields
            integrator.render_backward_sensors(scene, params, [None if g is None else g.contiguous() for g in grad_outs], sensors, seed=seed_grad, spp=spp_grad, shard=shard, group=group)
            return tuple(params.grad(k).clone() for k in diff_keys)

    return _FieldsOp.apply(*[params[k] for k in diff_keys])
