/*
 * mi355.h -- C ABI of the MI355X-native path-tracing / Path-Replay-Backprop engine.
 *
 * This is the drop-in boundary for ONE hot path of Mitsuba 3 (gitmon/mitsuba3-camera):
 * the body of Integrator::render / render_forward / render_backward for the `path`, `prb`
 * and `ptracer` integrators on surface-only RGB scenes.  A Mitsuba-side shim
 * flattens Scene/Shape/BSDF/Emitter/Sensor objects into the POD records below and
 * forwards to these entry points.  All paths cited are relative to the reference tree.
 *
 *   replaced interface                                   reference location
 *   ---------------------------------------------------  -------------------------------------------
 *   Integrator::render(scene,sensor,seed,spp,develop,..)  include/mitsuba/render/integrator.h:74-120,
 *                                                         src/render/integrator.cpp:120-367 (JIT branch :246-360)
 *   Integrator::render_backward(scene,params,grad_in,..)  include/mitsuba/render/integrator.h:161-215,
 *                                                         src/python/python/ad/integrators/common.py:631-796
 *   PathIntegrator::sample                                src/integrators/path.cpp:95-287
 *   PRBIntegrator.sample                                  src/python/python/ad/integrators/prb.py:59-257
 *   Scene::ray_intersect / ray_test                       src/render/scene.cpp:181-210
 *   Film::prepare / put_block / develop (hdrfilm, RGBW)   src/films/hdrfilm.cpp:245-409
 *   SceneParameters.update -> parameters_changed          src/python/python/util.py:185-232
 *   Integrator::render_forward(scene,params,..)           include/mitsuba/render/integrator.h:104-190,
 *                                                         src/python/python/ad/integrators/common.py:498-629
 *   ParticleTracerIntegrator::sample under                src/integrators/ptracer.cpp:77-371,
 *     AdjointIntegrator::render (+ render_backward)       src/render/integrator.cpp:545-745, :65-84
 *   Film::sample_border, samples_per_pass                 src/render/film.cpp:32, src/render/integrator.cpp:133-150,319-329
 *   Mesh::parameters_changed (vertex_positions: bbox,     src/render/mesh.cpp:96-163, :321-413,
 *     vertex normals, area pmf) + Scene::parameters_changed src/render/scene.cpp:481-529  -> device-side refit, no rebuild
 *   dr.backward() through ptracer onto vertex_positions   notebooks/scripts/lens_design/lens.py:505-565, design_problem.py:403-428, :481-584
 *     and IORs (reverse-mode geometry adjoint)              (AD in the reference; src/bsdfs/dielectric.cpp:236-239, :343-351)
 *   the film / weight / gradient all-reduce of a           -- (the reference is single-GPU): mi355_set_reduce_hook
 *     multi-GPU job
 *   ProfilerPhase ranges                                   include/mitsuba/core/profiler.h:15-113 -> roctx ranges of the same names
 *   the `Sensor *sensor` argument of render / render_forward /  include/mitsuba/render/integrator.h:74-97, :104-215; Scene::sensors();
 *     render_backward (any sensor of the scene per call)     src/python/python/util.py:410-523 (`sensor=` of mi.render)
 *   PerspectiveCamera::traverse / parameters_changed /      src/sensors/perspective.cpp:155-199 (`x_fov`, `to_world`), Film size / crop
 *     update_camera_transforms                                (src/render/film.cpp:59-98)  -> mi355_scene_set_param / mi355_scene_set_sensor
 *   BasicPRBIntegrator.trace (the fork's geometric tracer)   src/python/python/ad/integrators/prb_basic.py:46-140 -> mi355_trace_rays
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; `void *stream` is a hipStream_t
 *     (NULL = the default stream); pointers prefixed `d_` are DEVICE pointers owned by
 *     the caller (e.g. torch tensors), everything else is HOST memory owned by the caller
 *     and copied during the call.
 *   - every function returns MI355_OK (0) or a negative error code; the message is
 *     available through mi355_last_error() (thread-local).
 *   - matrices are row-major 3x4 affine blocks (float[12]) = the first three rows of
 *     Mitsuba's Transform4f::matrix.
 *   - all arithmetic is fp32; indices u32; RNG state u64 (TEA-seeded PCG32 per lane,
 *     src/render/sampler.cpp:116-135).
 *   - ONE call in flight per scene handle: a handle owns scratch buffers (the grad_in / W image, redo flags, the particle
 *     tracer's record queue and path tape, event slots, counters) that every entry point re-uses, so two calls on one handle must be
 *     ordered -- same stream, or the first one synchronised -- exactly as Integrator::render is called from one thread
 *     (SURVEY.md 8b "Threading").  Different handles are independent.
 *   - every `d_` buffer must be ordinary (coarse-grained) device memory -- hipMalloc, or a torch CUDA tensor: the library is
 *     built with -munsafe-fp-atomics (hardware float atomics), which are undefined on fine-grained / managed / host-pinned memory.
 */
#ifndef MI355_H
#define MI355_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 9u

/* ---- status codes ------------------------------------------------------------------ */
#define MI355_OK                 0
#define MI355_ERR_INVALID       -1   /* bad argument / inconsistent description          */
#define MI355_ERR_UNSUPPORTED   -2   /* plugin type outside the path's scope             */
#define MI355_ERR_DEVICE        -3   /* HIP runtime error (message has hipGetErrorString) */
#define MI355_ERR_NO_DEVICE     -4   /* no gfx950 device visible; there is NO CPU fallback */

/* ---- plugin type tags (names follow the reference's plugin names) -------------------- */
enum { MI355_SHAPE_RECTANGLE = 0,   /* src/shapes/rectangle.cpp */
       MI355_SHAPE_MESH      = 1,   /* src/render/mesh.cpp (cube/obj/ply all arrive as meshes) */
       MI355_SHAPE_SPHERE    = 2,   /* src/shapes/sphere.cpp */
       MI355_SHAPE_DISK      = 3 }; /* src/shapes/disk.cpp (unit disk in the xy plane of to_world; flip_normals is folded into to_world) */

enum { MI355_BSDF_DIFFUSE        = 0,   /* src/bsdfs/diffuse.cpp */
       MI355_BSDF_ROUGHCONDUCTOR = 1,   /* src/bsdfs/roughconductor.cpp */
       MI355_BSDF_DIELECTRIC     = 2,   /* src/bsdfs/dielectric.cpp */
       MI355_BSDF_CONDUCTOR      = 3 }; /* src/bsdfs/conductor.cpp: delta reflection, reflectance * fresnel_conductor(eta, k) */

enum { MI355_EMITTER_AREA = 0,          /* src/emitters/area.cpp */
       MI355_EMITTER_DIRECTIONALAREA = 1 }; /* src/emitters/directionalarea.cpp: emits along the surface normal only (collimated beam).
                                               Invisible to path / prb (eval = 0, sample_direction fails); a light source for ptracer */

enum { MI355_SENSOR_PERSPECTIVE = 0,    /* src/sensors/perspective.cpp */
       MI355_SENSOR_REALISTICLENS = 1 };/* src/sensors/realisticlens.cpp */

enum { MI355_RFILTER_BOX = 0,           /* src/rfilters/box.cpp      (radius 0.5) */
       MI355_RFILTER_GAUSSIAN = 1,      /* src/rfilters/gaussian.cpp (radius 4*stddev) */
       MI355_RFILTER_TENT = 2 };        /* src/rfilters/tent.cpp */

enum { MI355_FILTER_NEAREST = 0, MI355_FILTER_BILINEAR = 1 };           /* bitmap.cpp:154-163 */
enum { MI355_WRAP_REPEAT = 0, MI355_WRAP_MIRROR = 1, MI355_WRAP_CLAMP = 2 }; /* bitmap.cpp:165-176 */

/* ---- scene description --------------------------------------------------------------- */

/* `bitmap` texture (src/textures/bitmap.cpp): H x W x C float32 tensor, raw=true semantics. */
typedef struct mi355_texture_desc {
    uint32_t width, height, channels;   /* channels: 1 or 3 */
    uint32_t filter;                    /* MI355_FILTER_* */
    uint32_t wrap;                      /* MI355_WRAP_*   */
    float    to_uv[6];                  /* row-major 2x3 affine uv transform (bitmap.cpp:701) */
    const float *data;                  /* host, height*width*channels                        */
} mi355_texture_desc;

typedef struct mi355_bsdf_desc {
    uint32_t type;                      /* MI355_BSDF_* */
    float    reflectance[3];            /* diffuse: `reflectance.value` (srgb.cpp:70-72); conductor: specular_reflectance */
    int32_t  reflectance_texture;       /* index into textures[] or -1 (constant colour) */
    float    alpha_u, alpha_v;          /* roughconductor */
    uint32_t distribution;              /* 0 beckmann, 1 ggx */
    float    eta[3], k[3];              /* roughconductor complex IOR */
    float    int_ior, ext_ior;          /* dielectric */
    uint32_t sample_visible;
    float    transmittance[3];          /* dielectric: specular_transmittance (reflectance[] = specular_reflectance) */
} mi355_bsdf_desc;

typedef struct mi355_emitter_desc {
    uint32_t type;                      /* MI355_EMITTER_AREA */
    float    radiance[3];               /* `radiance.value` (area.cpp:78-81) */
    uint32_t shape;                     /* index of the shape this emitter is attached to */
} mi355_emitter_desc;

typedef struct mi355_shape_desc {
    uint32_t type;                      /* MI355_SHAPE_* */
    uint32_t bsdf;                      /* index into bsdfs[] */
    int32_t  emitter;                   /* index into emitters[] or -1 */
    uint32_t flip_normals;
    float    to_world[12];              /* rectangle / sphere: object->world (rectangle.cpp:115-126) */
    float    to_object[12];             /* its inverse as held by the shape (m_to_object) */
    /* mesh only: buffers exactly as Mesh exposes them (include/mitsuba/render/mesh.h:523-527),
       already in world space (cube.cpp:139-152) */
    uint32_t vertex_count, face_count;
    const float    *vertex_positions;   /* 3*vertex_count */
    const float    *vertex_normals;     /* 3*vertex_count or NULL */
    const float    *vertex_texcoords;   /* 2*vertex_count or NULL */
    const uint32_t *faces;              /* 3*face_count */
} mi355_shape_desc;

typedef struct mi355_sensor_desc {
    uint32_t type;                      /* MI355_SENSOR_* */
    float    to_world[12];              /* camera->world, no scale (perspective.cpp:144-145) */
    float    x_fov;                     /* degrees, after parse_fov (sensor.cpp:149-199) */
    float    near_clip, far_clip;
    uint32_t film_width, film_height;   /* Film::size */
    uint32_t crop_offset_x, crop_offset_y, crop_width, crop_height; /* Film::crop_* */
    uint32_t rfilter;                   /* MI355_RFILTER_* */
    float    rfilter_param;             /* gaussian: stddev (default 0.5); tent: radius */
    /* realistic lens (src/sensors/realisticlens.cpp): interface table ordered from the film towards the world,
       AFTER the constructor's fine focus (:899-905).  lens_count x 16 floats, metres:
       [0] type (0 sphere :345-424, 1 plane :427-493, 2 aperture stop :495-550, 3 asphere :553-821), [1] signed curvature
       radius (sphere), [2] element radius, [3] z intercept, [4] IOR on the film side, [5] IOR on the world side (both at
       589.3 nm, :169-179); asphere only: [6] conic constant kappa, [7] unitless curvature c = r_scale / R with
       r_scale = 1000 * element radius, [8..14] a_i * r_scale^(2i+3), i = 0..6 (ctor :559-572); [15] 0 */
    uint32_t lens_count;
    const float *lens_data;
    float    film_diagonal;             /* physical film diagonal [m] (film.cpp:46-48, film.h:193-198) */
    float    focus_distance;            /* unused by the engine (focus is already baked into the table) */
    float    shutter_open_time;         /* Sensor::shutter_open_time (sensor.cpp: shutter_close - shutter_open).  > 0 costs every sample one
                                           extra sampler draw for the time (integrator.cpp:442-444, common.py:293-295), any sensor type */
    uint32_t sample_exit_pupil;         /* realisticlens `sample_exit_pupil` (:851): aim at the exit-pupil bounds of the film point's radius
                                           (LUT of 64 x 2^20 traced rays, :1685-1806, built at scene creation) instead of the whole rear
                                           element; ray weight = shutter_open_time * bounds_area / z_rear^2 (:2282-2285) */
    uint32_t sample_border;             /* Film::sample_border (film.cpp:32): the lanes of the sampling integrators cover the crop window grown by
                                           the filter's border_size = ceil(radius - 0.5 - 2 RayEps) (rfilter.cpp:23) on every side
                                           (integrator.cpp:133-134, :307-308; common.py:255-278), so edge pixels receive their full footprint */
} mi355_sensor_desc;

typedef struct mi355_scene_desc {
    uint32_t abi_version;               /* MI355_ABI_VERSION */
    uint32_t n_shapes, n_bsdfs, n_emitters, n_textures;
    const mi355_shape_desc   *shapes;
    const mi355_bsdf_desc    *bsdfs;
    const mi355_emitter_desc *emitters;
    const mi355_texture_desc *textures;
    /* Scene::sensors() (include/mitsuba/render/scene.h): every sensor of the scene, in the order the scene lists them.  A call picks one
       with mi355_render_params.sensor -- the `Sensor *sensor` argument of Integrator::render / render_forward / render_backward
       (integrator.h:74-97; mi.render(scene, params, sensor=i), util.py:410-523).  Sensors share the scene replica: choosing or editing one
       touches no geometry, acceleration structure or scratch buffer.  At least one. */
    uint32_t n_sensors;
    const mi355_sensor_desc  *sensors;
} mi355_scene_desc;

/* ---- render parameters ---------------------------------------------------------------- */

#define MI355_FLAG_HIDE_EMITTERS  1u   /* MonteCarloIntegrator::m_hide_emitters */
#define MI355_FLAG_TIMING         2u   /* bracket the path kernel with HIP events on `stream` */
#define MI355_FLAG_TWO_PASS       8u   /* render_backward: primal walk + adjoint walk per path as prb.py does (common.py:760-796)
                                          instead of the default single walk with deferred L_total ("one-pass adjoint") */
#define MI355_FLAG_REGEN          4u   /* path integrator, >= 128 spp: a wave re-uses the lanes of finished paths for its next samples
                                          (path regeneration).  Same samples, same film; opt-in because on the Cornell box it
                                          measured 11.3 ms against 10.8 ms for the static assignment */

#define MI355_FLAG_EMITTER_GRADS_ONLY 16u /* render_backward with MI355_INTEGRATOR_PTRACER: only the MI355_PARAM_EMITTER_RADIANCE slots are
                                          wanted -- one walk per light path instead of the two that albedo / texture gradients need */

#define MI355_FLAG_GEOMETRY_GRADS 64u  /* render_backward with MI355_INTEGRATOR_PTRACER: ALSO fill the MI355_PARAM_VERTEX_POSITIONS and MI355_PARAM_BSDF_IOR
                                          slots -- the reverse-mode geometry / IOR adjoint of the light paths (one extra walk per light path with a
                                          path tape in HBM; DESIGN.md "geometry adjoint").  Perspective sensor, finite max_depth <= 32 */

#define MI355_FLAG_SKIP_COLOUR_GRADS 128u /* with MI355_FLAG_GEOMETRY_GRADS: leave the reflectance / texture / radiance slots untouched -- the colour
                                          walk of the light tracer's render_backward is skipped (a lens-design step differentiates geometry and IORs
                                          only: dr.backward() reaches only the parameters whose gradients are enabled) */

#define MI355_FLAG_DIRECT_SPLAT   32u  /* particle tracer: splat every connection with direct float atomics instead of the default deferred,
                                          tile-sorted splat (same film up to summation order; a debugging / comparison switch -- Gaussian
                                          filter 234.6 ms against 7.1 ms on the Cornell box 512^2 x 64 spp) */

#define MI355_FLAG_NO_LANE_SORT   256u /* particle tracer on scenes with BVH meshes: keep the natural order of the light samples instead of listing them
                                          cell by cell of their emitter's sample domain (DESIGN.md "coherent lanes"; same samples, same film and gradients up
                                          to the order of the float atomics -- an A/B switch) */

enum { MI355_MODE_FUSED = 0,           /* default: one launch per pass, path state in registers; the path integrator re-packs the
                                          live paths of a workgroup (wave ballot + prefix sum, SoA state exchanged through LDS) from
                                          bounce 2 on whenever that empties a whole wave */
       MI355_MODE_REPACK_ALWAYS = 1,   /* re-pack after EVERY bounce, unconditionally; path integrator only.  Still ONE fused launch with the path
                                          state in registers: no per-stage kernels, no SoA streams in HBM (that "wavefront" formulation is not built:
                                          DESIGN.md section 2).  An A/B switch against the adaptive default: same film, 9.7-9.9 against 9.7 ms in round 1 */
       MI355_MODE_WAVEFRONT = MI355_MODE_REPACK_ALWAYS,   /* name used up to ABI v8; misleading (see above), kept so old callers compile */
       MI355_MODE_STATIC = 2 };        /* never re-pack: a lane keeps its path until the longest path of its wave ends */

enum { MI355_INTEGRATOR_PATH = 0,      /* src/integrators/path.cpp */
       MI355_INTEGRATOR_PRB  = 1,      /* src/python/python/ad/integrators/prb.py (primal estimator differs in RR timing) */
       MI355_INTEGRATOR_PTRACER = 2 }; /* src/integrators/ptracer.cpp under AdjointIntegrator::render (integrator.cpp:545-745): light paths
                                          connected to a perspective sensor.  mi355_render_accumulate only; lanes = FILM pixels x spp (the
                                          shard fields count film pixels, not crop pixels), normalised splats into RGB, W stays 0 */

typedef struct mi355_render_params {
    uint32_t seed;                      /* the `seed` argument of Integrator::render */
    uint32_t spp;                       /* samples per pixel of the WHOLE job (defines global lane ids) */
    int32_t  max_depth;                 /* -1 = unbounded (integrator.cpp:510-514) */
    int32_t  rr_depth;                  /* default 5 */
    uint32_t flags;                     /* MI355_FLAG_* */
    uint32_t mode;                      /* MI355_MODE_* */
    uint32_t integrator;                /* MI355_INTEGRATOR_* (selects which primal estimator is replayed) */
    /* shard of the sample plane processed by THIS call (one rank of a multi-GPU job).
       Global lane id = pixel*spp + s  (integrator.cpp:293-305); the shard is the set
       { pixel in [pixel_begin, pixel_begin+pixel_count) } x { s in [spp_begin, spp_begin+spp_count) }.
       Zero counts mean "everything". */
    uint32_t pixel_begin, pixel_count;
    uint32_t spp_begin, spp_count;
    /* Interleaved pixel tiles (load balance across ranks, SURVEY.md 8(e) "16-row tiles round-robin") in ONE launch:
       with pixel_tile != 0 the q-th pixel of the shard, q in [0, pixel_count), is
           pixel_begin + (q / pixel_tile) * pixel_tile_stride + q % pixel_tile
       (rank r of G with tiles of T pixels: pixel_begin = r*T, pixel_tile = T, pixel_tile_stride = G*T).
       pixel_tile == 0: the contiguous range above. */
    uint32_t pixel_tile, pixel_tile_stride;
    /* Passes of the JIT render loops (`samples_per_pass`, integrator.cpp:565-573 and :700-726): the sampler is seeded ONCE for a
       wavefront of  pixels * spp / passes  streams and every stream draws `passes` consecutive samples, so the stream grid is
       spp / passes  per pixel (global stream id = pixel * (spp / passes) + s; spp_begin / spp_count then count streams).
       0 or 1 = one pass.  Honoured by MI355_INTEGRATOR_PTRACER (whose reference test fixture renders in 4 passes) and by
       MI355_INTEGRATOR_PATH with the perspective sensor (SamplingIntegrator::render, integrator.cpp:283, :319-329; every lane draws the
       six numbers of every loop iteration it enters, path.cpp:196, :212-213, :272, so that the next pass starts where the reference's
       does).  prb.py renders in one pass: MI355_ERR_UNSUPPORTED. */
    uint32_t passes;
    uint32_t sensor;                    /* index into mi355_scene_desc.sensors: the `Sensor *` argument of Integrator::render / render_forward /
                                           render_backward (integrator.h:74-97, :104-215).  Film size, crop window, filter and lane grid are this sensor's */
} mi355_render_params;

/* Per-event device counters behind the algorithmic-byte formula of SURVEY.md 8(d). */
typedef struct mi355_counters {
    uint64_t n_gen, n_isect, n_shade, n_shadow, n_splat;
    float    kernel_ms;                 /* HIP-event duration of the last timed path kernel (MI355_FLAG_TIMING) */
    uint32_t kernel_launches;           /* launches covered by kernel_ms */
    uint64_t n_redo_groups;             /* render_backward: workgroups whose one-pass record queue overflowed and that the
                                           two-pass kernel redid (0 on the benchmark scenes) */
} mi355_counters;

/* One differentiable parameter (ParamFlags::Differentiable in traverse()).  The gradient buffer concatenates, in this order: the
   constant diffuse reflectances, the textures, the emitter radiances, the vertex positions of every mesh, the IOR pairs of every
   dielectric.  path / prb write the first three groups (prb.py detaches the intersection, so its geometry gradient is zero); the
   particle tracer writes all of them when MI355_FLAG_GEOMETRY_GRADS is set. */
enum { MI355_PARAM_BSDF_REFLECTANCE = 0,   /* 3 floats      diffuse.cpp:97-99 / srgb.cpp:70-72 */
       MI355_PARAM_TEXTURE_DATA     = 1,   /* H*W*C floats  bitmap.cpp:308-311 */
       MI355_PARAM_EMITTER_RADIANCE = 2,   /* 3 floats      area.cpp:78-81 */
       MI355_PARAM_VERTEX_POSITIONS = 3,   /* 3*vertex_count floats, index = shape: `vertex_positions` of a mesh, world space (mesh.cpp:1602-1611,
                                              mesh.h:523).  Setting it is Mesh::parameters_changed (mesh.cpp:96-163) + Scene::parameters_changed
                                              (scene.cpp:481-529): bounding boxes and the acceleration structure are REFITTED on the device (topology
                                              kept), and a mesh that has vertex normals gets them recomputed from the new positions
                                              (recompute_vertex_normals, mesh.cpp:321-413, angle-weighted) -- from then on its shading normals are
                                              functions of the positions for the derivative passes as well, as under Dr.Jit's AD */
       MI355_PARAM_BSDF_IOR         = 4,   /* 2 floats (int_ior, ext_ior), index = bsdf: a `dielectric`'s indices (dielectric.cpp:236-239), or a
                                              `dispersive` material's indices at 589.3 nm (dispersive_dielectric.cpp:236-239, :411-432 in RGB builds) */
       /* Sensor parameters (PerspectiveCamera::traverse, perspective.cpp:155-161).  Settable, NOT part of the gradient layout: the reference
          flags them Differentiable | Discontinuous, but no configuration on this path differentiates a camera (the lens-design loop moves and
          zooms its cameras under dr.suspend_grad, design_problem.py:113-132).  Setting one is PerspectiveCamera::parameters_changed
          (perspective.cpp:163-171 -> update_camera_transforms :173-199): host-side matrices only, no device work, no synchronisation. */
       MI355_PARAM_SENSOR_TO_WORLD  = 5,   /* 12 floats (row-major 3x4 camera->world), index = sensor; a transform with scale is refused (:166-167) */
       MI355_PARAM_SENSOR_X_FOV     = 6 }; /* 1 float, degrees, index = sensor (perspective only) */

typedef struct mi355_param_info {
    uint32_t kind;                      /* MI355_PARAM_* */
    uint32_t index;                     /* bsdf / texture / emitter index */
    uint64_t offset;                    /* offset (in floats) inside the concatenated gradient buffer */
    uint64_t size;                      /* number of floats */
} mi355_param_info;

typedef struct mi355_scene mi355_scene;

/* ---- lifecycle ---------------------------------------------------------------------------- */
const char *mi355_last_error(void);
uint32_t    mi355_abi_version(void);

/* Flatten + upload a scene replica onto HIP device `device`.  (Scene ctor + accel build,
   src/render/scene.cpp:30-140) */
int mi355_scene_create(const mi355_scene_desc *desc, int device, mi355_scene **out);
int mi355_scene_destroy(mi355_scene *scene);

/* SceneParameters.update for one parameter (util.py:185-232).  `values` is HOST memory.  Geometry updates
   (MI355_PARAM_VERTEX_POSITIONS) only mark the replica dirty; the refit runs once, on the stream of the next call that traces rays. */
int mi355_scene_set_param(mi355_scene *scene, uint32_t kind, uint32_t index,
                          const float *values, uint64_t count);
/* Same, `d_values` is DEVICE memory (optimiser state that already lives in HBM). */
int mi355_scene_set_param_device(mi355_scene *scene, uint32_t kind, uint32_t index,
                                 const float *d_values, uint64_t count, void *stream);

/* Replace the whole description of sensor `index` (same or another type, film size, crop window, filter): what a SceneParameters.update()
   on `<sensor>.film.size` / `crop_size` / `crop_offset` (film.cpp:59-98; design_problem.py:594-617 re-renders at another resolution) or a
   new `Sensor*` handed to Integrator::render amounts to.  Host-side only for a perspective camera; a realisticlens with sample_exit_pupil
   re-traces its exit-pupil table (one kernel).  `index == n_sensors` appends a sensor.  Geometry, BVH and scratch buffers are untouched. */
int mi355_scene_set_sensor(mi355_scene *scene, uint32_t index, const mi355_sensor_desc *sensor);
/* Number of sensors of the replica. */
int mi355_scene_sensor_count(const mi355_scene *scene);

/* dr.enable_grad on / off for one geometry parameter (MI355_PARAM_VERTEX_POSITIONS: index = shape; MI355_PARAM_BSDF_IOR: index = bsdf).
   Dr.Jit's backward pass only visits the parameters whose gradients are enabled (the lens-design loop differentiates the lens and baffle
   meshes, not the film plane or the housing, design_problem.py:481-492 / lens.py:473-498); with a slot disabled the geometry adjoint skips
   its scatter and leaves the slot untouched.  The layout does not change.  All slots are enabled at creation. */
int mi355_scene_param_enable_grad(mi355_scene *scene, uint32_t kind, uint32_t index, int enable);

/* Layout of the concatenated parameter-gradient buffer.  Returns the number of
   parameters; fills at most `capacity` entries. */
int mi355_scene_param_layout(const mi355_scene *scene, mi355_param_info *out, uint32_t capacity,
                             uint64_t *total_floats);

/* ---- forward path (Integrator::render) ----------------------------------------------------- */

/* Accumulate this shard's samples into the RGBW film accumulator (crop_h x crop_w x 4 fp32,
   DEVICE).  Equivalent to render_sample + ImageBlock::put for every lane of the shard
   (integrator.cpp:419-491, imageblock.cpp:418-531).  The accumulator is NOT cleared. */
int mi355_render_accumulate(mi355_scene *scene, const mi355_render_params *p,
                            float *d_film_rgbw, void *stream);

/* HDRFilm::develop (hdrfilm.cpp:304-409) of sensor `sensor`'s film: rgb / (W==0 ? 1 : W) -> crop_h x crop_w x 3 (DEVICE). */
int mi355_film_develop(mi355_scene *scene, uint32_t sensor, const float *d_film_rgbw, float *d_image_rgb, void *stream);

/* clear + accumulate + develop in one call: what Integrator::render returns.  d_image_rgb may be NULL: Integrator::render with
   develop = false (integrator.h:74-120) -- the film is accumulated and left undeveloped in d_film_rgbw. */
int mi355_render(mi355_scene *scene, const mi355_render_params *p,
                 float *d_film_rgbw, float *d_image_rgb, void *stream);

/* ---- PRB adjoint path (RBIntegrator.render_backward) ------------------------------------------- */

/* Step 1a (common.py:741-758): splat weight 1 for every lane of the shard ->
   d_weight (crop_h x crop_w fp32, DEVICE, accumulated).  Multi-GPU jobs all-reduce this
   buffer before step 2 so that every rank sees the same W[p]. */
int mi355_prb_splat_weights(mi355_scene *scene, const mi355_render_params *p,
                            float *d_weight, void *stream);

/* Steps 1b-3 (common.py:741-796, prb.py:59-257): deltaL = adjoint of splat/develop from
   d_grad_in (crop_h x crop_w x 3) and d_weight; primal replay; adjoint replay with the same
   RNG streams; parameter gradients are ACCUMULATED into d_grad_params (layout:
   mi355_scene_param_layout).  Multi-GPU jobs all-reduce d_grad_params afterwards.
   MI355_INTEGRATOR_PTRACER (Integrator::render_backward, integrator.cpp:65-84, over ptracer.cpp; pinned by the hard-coded gradient
   of test_ptracer.py:204-225): the light paths are replayed and every sensor connection gathers d_grad_in at its splat taps.
   Emitter radiances, constant diffuse reflectances and reflectance textures -- every parameter of mi355_scene_param_layout -- receive
   their gradients (path replay: two walks per light path; MI355_FLAG_EMITTER_GRADS_ONLY: radiances only, one walk).  d_weight is
   unused and may be NULL: the normalised splat has no weight channel.  The roulette probability is treated as detached (path.cpp:272
   does so explicitly, ptracer.cpp:277 does not; identical while depth <= rr_depth).  With MI355_FLAG_GEOMETRY_GRADS the vertex positions
   of every mesh and the IORs of every dielectric receive their gradients too (what notebooks/scripts/lens_design/design_problem.py:508-549
   back-propagates through): a hand-written reverse mode of ptracer.cpp:200-371 over a recorded path, mitsuba3-camera_amd/csrc/
   mi355_geo_adjoint.h.  Visibility and discrete choices are not differentiated, as in AD. */
int mi355_render_backward(mi355_scene *scene, const mi355_render_params *p,
                          const float *d_grad_in, const float *d_weight,
                          float *d_grad_params, void *stream);

/* Integrator::render_forward (include/mitsuba/render/integrator.h:104-190; RBIntegrator.render_forward, common.py:498-629; prb.py in
   dr.ADMode.Forward): forward-mode derivative of the render.  d_tangent_params holds one float per differentiable scene float,
   laid out like the gradient buffer (mi355_scene_param_layout) -- what dr.set_grad() put on the parameters.  Every sample's
   dL = sum over its vertices of d(Le + Lr_dir + Lr_ind)/d(theta) . tangent is splatted with weight 1 into d_film_rgbw
   (ACCUMULATED, like mi355_render_accumulate); mi355_film_develop of that accumulator is the gradient image.  path and prb
   share it (the path integrator differentiates through the same estimator). */
int mi355_render_forward(mi355_scene *scene, const mi355_render_params *p,
                         const float *d_tangent_params, float *d_film_rgbw, void *stream);

/* Forward-mode derivative of the LIGHT-TRACED film (MI355_INTEGRATOR_PTRACER) with respect to one design scalar theta of a lens --
   what the fork's lens-design notebooks differentiate through Dr.Jit's AD (notebooks/scripts/lens_design/lens.py:505-565: lens
   surfaces are meshes whose vertex_positions depend on the lens scalars, materials carry IOR parameters; design_problem.py:403-428:
   rendered with `ptracer`; the reference path is Integrator::render_forward, integrator.cpp:44-63, over ptracer.cpp).
   The tangent of theta may have two parts, either of which can be absent:
     - geometry: d(position)/d(theta) of every vertex of the mesh `shape` (3 floats per vertex, the mesh's own vertex order);
     - material: d(int_ior)/d(theta) and d(ext_ior)/d(theta) of the dielectric `bsdf` (the refracted direction depends on their ratio).
   The derivative film d(RGB)/d(theta) is ACCUMULATED into d_film_rgbw (W untouched), so that mi355_film_develop of it is the
   gradient image.  A dual-number walk of the same light paths and sampler streams (mitsuba3-camera_amd/csrc/mi355_geo_forward.h lists
   what is and is not differentiated); reverse mode over a handful of lens scalars = one call per scalar.  Perspective sensor; the
   mesh must not carry an emitter. */
#define MI355_NONE 0xffffffffu
typedef struct mi355_design_tangent {
    uint32_t shape;                     /* mesh whose vertices move, or MI355_NONE */
    const float *d_vertex_tangent;      /* DEVICE, 3 floats per vertex of `shape` (NULL iff shape == MI355_NONE) */
    uint32_t bsdf;                      /* dielectric whose IORs change, or MI355_NONE */
    float    d_int_ior, d_ext_ior;
} mi355_design_tangent;
int mi355_ptracer_forward_design(mi355_scene *scene, const mi355_render_params *p, const mi355_design_tangent *tangent,
                                 float *d_film_rgbw, void *stream);

/* BasicPRBIntegrator.trace(scene, ray, num_lenses) of the fork (src/python/python/ad/integrators/prb_basic.py:46-140): the deterministic
   geometric ray tracer the lens-design scripts use to place each field's camera on its central ray (design_problem.py:183-203) and to
   check the lens meshes against the paraxial matrices (:430-472).  Every caller ray is walked through the scene taking, at each
   interaction, the lobe bsdf.sample() returns for sample1 = 1.0, sample2 = (0, 0) -- a dielectric interface always refracts (it mirrors
   only under total internal reflection), a black surface ends the ray -- until `num_lenses` surfaces were hit, max_depth is reached
   (the integrator's `max_depth`: a vertex continues while depth + 1 < max_depth) or the ray leaves the scene.
     d_rays = n x 8 (o.xyz, d.xyz, maxt, pad), as for mi355_debug_ray_intersect
     d_out  = n x 12: [0..2] origin and [3..5] direction of the PENULTIMATE ray (the one that produced the last interaction),
              [6..8] position of the last interaction (0 when the last ray hit nothing), [9] 1.0 iff that interaction is valid,
              [10] number of surfaces hit, [11] 0
   The reference's `valid` = si.is_valid() & dr.allclose(0, si.p.z) -- one verdict for the whole wavefront -- is formed by the caller from [8], [9]. */
int mi355_trace_rays(mi355_scene *scene, const float *d_rays, uint32_t n, uint32_t num_lenses, int32_t max_depth, float *d_out, void *stream);

/* The lens-design loop's surface update (notebooks/scripts/lens_design/lens.py: EvenAsphericSurface.compute_z_dr :166-249, ConicSurface :55-97,
   LensElement.update :509-568 -- Dr.Jit arithmetic in the reference, part of the AD graph of every design iteration): the sag of a conic / even
   asphere in the scripts' dimensionless form, for every vertex of a surface mesh, and its transpose.
     d_params          10 floats: c, K, z0, a4, a6, a8, a10, a12, a14, a16 (c and the a_i already multiplied by the powers of radial_extent the scripts use)
     d_base_positions  n x 3: the mesh as it was created; x and y are kept
     d_eval_xy         n x 2 points the sag is evaluated at, or NULL: the vertices' own (x, y).  (A baffle follows the surface's rim: every vertex
                       evaluates at (radial_extent, 0), lens.py:553-565.)
     mi355_lens_sag          d_positions (n x 3) = (x, y, z(eval point))
     mi355_lens_sag_backward d_grad_params (10 floats, ACCUMULATED) += sum_i d_grad_positions[i].z * dz_i / d_params
   Pure functions of their arguments on the current HIP device; no scene handle. */
int mi355_lens_sag(const float *d_eval_xy, const float *d_base_positions, uint32_t n, float radial_extent, const float *d_params, float *d_positions, void *stream);
int mi355_lens_sag_backward(const float *d_eval_xy, const float *d_base_positions, uint32_t n, float radial_extent, const float *d_params,
                            const float *d_grad_positions, float *d_grad_params, void *stream);

/* Several sensors, ONE light-traced pass (particle tracer, forward).  notebooks/scripts/lens_design/design_problem.py:525-537 renders every field of a
   design iteration with the SAME seed (`mi.render(scene, params, seed=it, ..., sensor=sensor_idx)`), and AdjointIntegrator::render's light paths depend
   on the seed and on the film SIZE only (integrator.cpp:575-600: one sampler stream per film pixel and sample): the fields' cameras therefore see the same
   light paths, and only the sensor connection at every vertex (ptracer.cpp:232-236, perspective.cpp:282-340) is per camera.  This call traces the paths
   once and connects every vertex to each of the `n_sensors` cameras: the same films as `n_sensors` calls of mi355_render_accumulate with that seed, up to
   the order of the float additions.  Requirements (else MI355_ERR_UNSUPPORTED): perspective sensors, one film size without a crop window, one
   reconstruction filter, one pass (no samples_per_pass), 1 <= n_sensors <= MI355_MAX_SENSORS_PER_PASS (else MI355_ERR_INVALID).  A shard of the light-sample grid (pixel_* / spp_* of `params`) applies to every camera alike: the
   partial films of the ranks add up in the virtual film.
     d_film   H x virt_w x 4 floats (RGBW), zeroed by the caller: sensor k's film occupies the columns [k * pitch + x0, k * pitch + x0 + W) of every row
              (mi355_multi_film_layout: pitch = roundup(W, 16) + 32, x0 = 16, virt_w = n_sensors * pitch); the columns in between receive the filter
              taps that leave a film and are to be ignored.  Develop a film by copying its columns out and calling mi355_film_develop on the copy. */
#define MI355_MAX_SENSORS_PER_PASS 8u
int mi355_multi_film_layout(uint32_t film_w, uint32_t n_sensors, uint32_t *pitch, uint32_t *x0, uint32_t *virt_w);
int mi355_render_accumulate_sensors(mi355_scene *scene, const mi355_render_params *params, uint32_t n_sensors, const uint32_t *sensors, float *d_film, void *stream);
/* ... and its adjoint with respect to mesh vertex positions and IORs (mi355_render_backward with MI355_FLAG_GEOMETRY_GRADS | MI355_FLAG_SKIP_COLOUR_GRADS,
   which `params->flags` must carry): util.py:517 derives ONE differential seed from the primal seed, so the fields of an iteration differentiate over the
   same light paths too -- one forward walk onto the path tape, one backward sweep in which a vertex gathers the image adjoint of every camera it connected
   to; the sum of `n_sensors` calls of mi355_render_backward, up to the order of the float additions.  d_grad_in: HOST array of `n_sensors` device pointers,
   H x W x 3 each, or NULL for a sensor whose image carries no adjoint.  d_grad_params is accumulated into, as by mi355_render_backward. */
int mi355_render_backward_sensors(mi355_scene *scene, const mi355_render_params *params, uint32_t n_sensors, const uint32_t *sensors, const float *const *d_grad_in,
                                  float *d_grad_params, void *stream);

/* The rest of a design iteration that the reference runs as Dr.Jit kernels: the spot loss of one field (notebooks/scripts/lens_design/losses.py:31-41,
   rms_loss_and_center: squared RMS radius of the spot in pixels about its detached centroid, the image normalised by its detached mean) and the
   optimiser step (src/python/python/ad/optimizers.py:263-302, Adam.step without mask_updates) for all design scalars in one launch.
     mi355_spot_loss           d_image height x width x 3 -> d_out[5] = { loss, ibar (row), jbar (column), mean(image), sum(I) }
     mi355_spot_loss_backward  d_grad_image (height x width x 3, overwritten) = d_grad_loss[0] * d loss / d image, from the d_out of the forward call
     mi355_adam_step           per slot: m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2 (both in place); param_out = param - lr_t m / (sqrt(v) + eps),
                               lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) formed by the caller (t is per variable); b1, b2, eps are doubles because
                               1 - b is formed in double before it is rounded to the tensors' float, as with the reference's Python scalars.  `slots` is a HOST array of DEVICE pointers.
   Pure functions of their arguments on the current HIP device; no scene handle. */
typedef struct mi355_adam_slot { const float *param, *grad; float *m, *v, *param_out; float lr_t; uint32_t count; } mi355_adam_slot;
int mi355_spot_loss(const float *d_image, uint32_t height, uint32_t width, float *d_out, void *stream);
int mi355_spot_loss_backward(const float *d_out, const float *d_grad_loss, uint32_t height, uint32_t width, float *d_grad_image, void *stream);
int mi355_adam_step(const mi355_adam_slot *slots, uint32_t n, double beta_1, double beta_2, double epsilon, void *stream);

/* ---- introspection -------------------------------------------------------------------------- */
int mi355_get_counters(mi355_scene *scene, mi355_counters *out, int reset);

/* How the replica's geometry was organised (cf. the accel statistics Scene logs, scene.cpp:108-118):
   out[0] flat parallelogram records, out[1] oriented-box records, out[2] meshes traversed through a BVH,
   out[3] BVH nodes, out[4] triangles, out[5] differentiable floats, out[6] analytic spheres,
   out[7] feature bits (1: glossy / specular BSDFs, 2: analytic spheres / disks / directional emitters, 4: realisticlens sensor (of the sensor
   selected last), 8: meshes traversed through the BVH, 16: more shape / bsdf / emitter records than the all-diffuse kernels stage in LDS;
   any bit -> general kernels). */
int mi355_scene_info(const mi355_scene *scene, uint32_t out[8]);

/* ---- device memory the engine allocates on its own (per scene handle, lazily, kept until mi355_scene_destroy) ---------------------
   Besides the scene replica (vertex / index / texture buffers, BVH: 128 B per 4-wide node, 36 B per triangle) a handle owns scratch
   that is sized by the LARGEST call it has served:
     path / prb         grad_in / W scratch 16 B per pixel; one redo flag (4 B) per workgroup of 256 lanes
     ptracer render     deferred-splat record queue: 40 B per queued connection (unsorted + sorted copy of a 20-byte record), at most
                        min(lanes * min(max_depth, 8), 128 Mi) records = 5.4 GB; connections beyond the capacity are splatted with direct
                        atomics (same film).  mi355_ptracer_forward_design: twice that (records carry a second half).
     ptracer backward   with MI355_FLAG_GEOMETRY_GRADS: the path tape, 72 B per vertex and lane, for chunks of at most 4 Mi lanes:
                        (max_depth - 1) * 72 B * min(lanes, 4 Mi) <= 9.7 GB at max_depth 33; 12 B per mesh vertex of normal adjoints
   Nothing else grows with the job; all of it is coarse-grained device memory of the handle's device. */

/* ---- multi-GPU exchange ------------------------------------------------------------------------------------------
   One process per GPU, every rank holds the whole scene and renders its shard (pixel range, interleaved tiles or spp slice of
   mi355_render_params).  The only exchange of the path is a SUM over the ranks of (a) the RGBW film accumulator before develop,
   (b) PRB's splatted weight image, (c) the parameter-gradient buffer.  This library links no communication library: a caller that wants
   the exchange inside the calls registers a hook that sums `count` floats of device memory over all ranks IN PLACE, ordered on
   `stream` (the shim of INTEGRATION.md section 4 does it with one ncclAllReduce on its RCCL communicator; the Python host layer leaves
   the hook unset and calls torch.distributed itself).  With a hook set:
     mi355_render            accumulate -> hook(film, W*H*4) -> develop   (Integrator::render of the whole job on every rank)
     mi355_prb_splat_weights kernel -> hook(weight, W*H)
     mi355_render_backward   kernels -> hook(grad_params, total_floats)     (d_grad_params must be zero on entry on all ranks but one)
   mi355_render_accumulate never calls it.  The hook is a collective: every rank reaches it, also a rank whose shard is empty.  It
   returns 0 on success; anything else fails the call with MI355_ERR_DEVICE.  fn = NULL removes the hook.
   A call that FAILS before the exchange (an allocation of the tape or the record queue, an invalid argument on one rank only) returns its
   error WITHOUT reaching the hook: the other ranks are then waiting in a collective this rank will never join, so the caller must abort
   the job on any error of a hooked call -- as with any failed collective.  (Argument errors are the same on every rank by construction:
   every rank passes the same parameters but for its shard.) */
typedef int (*mi355_reduce_fn)(void *user, float *d_buffer, uint64_t count, void *stream);
int mi355_set_reduce_hook(mi355_scene *scene, mi355_reduce_fn fn, void *user);

/* realisticlens with sample_exit_pupil: the 64 (min.x, min.y, max.x, max.y) exit-pupil bounds the scene was created with
   (compute_exit_pupil_bounds, realisticlens.cpp:1772-1806).  out = 256 floats of host memory. */
int mi355_lens_exit_pupil_bounds(const mi355_scene *scene, uint32_t sensor, float *out);

/* Per-lane radiance of lanes [lane_begin, lane_begin+count) of the whole job (no film):
   d_out = count x 4 (r,g,b,valid).  Used by the lane-exact parity tests
   (check_vectorization, src/python/python/test/util.py:88-175). */
int mi355_debug_lane_radiance(mi355_scene *scene, const mi355_render_params *p,
                              uint32_t lane_begin, uint32_t count, float *d_out, void *stream);

/* Closest-hit / any-hit queries on caller-provided rays (Scene::ray_intersect_preliminary /
   ray_test, scene.cpp:192-210).  d_rays = n x 8 (o.xyz, d.xyz, maxt, pad);
   d_hits = n x 8 (t, prim_u, prim_v, prim_index, shape_index, occluded, pad, pad) with the
   integer fields bit-cast to float. */
int mi355_debug_ray_intersect(mi355_scene *scene, const float *d_rays, uint32_t n,
                              float *d_hits, void *stream);

/* SurfaceInteraction3f of the closest hit of caller-provided rays (Shape::compute_surface_interaction through
   Scene::ray_intersect, scene.cpp:192-199 + mesh.cpp / rectangle.cpp / sphere.cpp / disk.cpp `compute_surface_interaction`):
   d_si = n x 24 floats (p3, n3, uv2, sh_frame.n3, wi3, t, shape index bit-cast (0xffffffff: no hit), sh_frame.s3, sh_frame.t3, pad2).
   The same layout as the oracle's orc_ray_intersect(si_out): the field-by-field parity test of the two. */
int mi355_debug_surface_interaction(mi355_scene *scene, const float *d_rays, uint32_t n, float *d_si, void *stream);

/* BitmapTexture::eval_3 of texture `texture` (index into the scene's texture table) at caller-provided uv (n x 2 floats):
   d_rgb = n x 3.  The device side of the texel-centre / wrap-mode identities of src/textures/tests/test_bitmap.py:58-157. */
int mi355_debug_texture_eval(mi355_scene *scene, uint32_t texture, const float *d_uv, uint32_t n, float *d_rgb, void *stream);

/* Diagnostic builds (-DMI_BVH_STATS, tools/bvh_stats.py): traversal work since the last reset -- out[0] rays that entered the BVH,
   out[1] nodes fetched, out[2] triangles tested, out[3] unused.  All zero in the shipped build. */
int mi355_debug_bvh_stats(mi355_scene *scene, uint64_t out[4], int reset);

/* Test hook: cap the deferred-splat record queue of the particle tracer at `records` (0 = size it from the launch, the default), so
   that the overflow path -- records beyond the capacity are splatted with direct atomics -- can be exercised on a small scene. */
int mi355_debug_set_splat_capacity(mi355_scene *scene, uint32_t records);

/* ---- host-side pure helpers (no GPU needed; exposed so their KATs run on CPU) ------------------- */

/* perspective_projection (include/mitsuba/render/sensor.h:228-265): 4x4 row-major camera->sample. */
void mi355_perspective_projection(const int32_t film_size[2], const int32_t crop_size[2],
                                  const int32_t crop_offset[2], float fov_x,
                                  float near_clip, float far_clip, float out_matrix[16]);
/* GaussianFilter coefficient table (gaussian.cpp:59-86): 10 Estrin coefficients in x^2. */
void mi355_gaussian_coeffs(float stddev, float out_coeff[10]);

#ifdef __cplusplus
}
#endif
#endif /* MI355_H */
