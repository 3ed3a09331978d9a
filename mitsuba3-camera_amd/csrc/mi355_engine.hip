// mi355_engine.hip -- kernels and host side of the C ABI declared in include/mi355.h.
//
// Kernels (all gfx950, 256-thread workgroups = 4 waves of 64; FULL = scene has non-diffuse BSDFs / spheres / a lens sensor):
//   k_render_compact<INTEGRATOR, FULL>  DEFAULT forward kernel.  One lane per camera path: TEA/PCG32 seed -> pixel jitter ->
//                               primary ray -> PathIntegrator::sample / PRBIntegrator.sample (primal) with the state in
//                               registers; the workgroup re-packs its live paths (wave ballot + mbcnt prefix, SoA state
//                               through LDS) whenever that empties a wave; per-pixel pre-reduced film splat through LDS
//                               (one atomic per tap+channel per wave instead of 100 per lane)
//   k_render<INTEGRATOR, OUT, FULL>     same without re-packing (MI355_MODE_STATIC), and the per-lane debug output
//   k_render_regen<FULL>        opt-in path regeneration variant (MI355_FLAG_REGEN)
//   k_splat_weights             W[p] = sum of filter weights (PRB step 1a)
//   k_prepare_grad              grad_in / W
//   k_prb_backward_1p<FULL, BVH>  DEFAULT adjoint: dL gather -> ONE walk; gradient records wait in an LDS queue for the
//                               lane's L_total, resolved per workgroup (LDS accumulators, per-texel hash, float atomics)
//   k_prb_backward<FULL>        prb.py's two walks (primal replay + adjoint replay): MI355_FLAG_TWO_PASS, and the redo pass
//                               for workgroups whose one-pass queue overflowed
//   k_exit_pupil                realisticlens exit-pupil LUT (64 rings x 2^20 rays) at scene creation
//   k_ptracer<FULL, GRAD>       particle tracer (ptracer.cpp under AdjointIntegrator::render): one lane per light-sample stream;
//                               GRAD = render_backward for the emitter radiances (gathers dL/dimage at the splat taps)
//   k_splat_hist / _scan / _scatter / _tiles   deferred splat of the particle tracer's connections: counting sort by 16 x 16 film
//                               tile (LDS-privatised), then per-tile accumulation in LDS (64-bit fixed point, integer atomics)
//   k_develop                   HDRFilm::develop
//   k_debug_intersect           Scene::ray_intersect_preliminary / ray_test on caller rays
// There is no CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "mi355_device.h"
#include "mi355_geo_forward.h"
#include "mi355_geo_backward.h"
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <numeric>
#include <array>

#include <dlfcn.h>

// ---- roctx ranges named after the reference's ProfilerPhase list (include/mitsuba/core/profiler.h:15-113) ----------------------------
// The marker library is never a link-time dependency: the ranges are emitted when librocprofiler-sdk-roctx.so (or the older
// libroctx64.so) is ALREADY loaded in the process -- which is what `rocprofv3 --marker-trace` arranges -- or when MI355_ROCTX=1 asks for
// it to be loaded.  Otherwise every Phase is two null checks.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr; int (*pop)() = nullptr;
    Roctx() {
        const char *libs[] = { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" };
        const char *want = getenv("MI355_ROCTX");
        void *h = nullptr;
        for (const char *l : libs) if (!h) h = dlopen(l, RTLD_LAZY | RTLD_NOLOAD);
        if (!h && want && want[0] == '1') for (const char *l : libs) if (!h) h = dlopen(l, RTLD_LAZY | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char *)) dlsym(h, "roctxRangePushA"); pop = (int (*)()) dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
Roctx &roctx() { static Roctx r; return r; }
struct Phase {                                                             // ScopedPhase, profiler.h:72-110
    explicit Phase(const char *name) { if (roctx().push) roctx().push(name); }
    ~Phase() { if (roctx().pop) roctx().pop(); }
    Phase(const Phase &) = delete; Phase &operator=(const Phase &) = delete;
};
// profiler_phase_id[] strings of the phases a host-visible call of this engine corresponds to (the finer phases -- RayIntersect,
// BSDFSample, ImageBlockPut ... -- are fused into one kernel here and show up as that kernel in the kernel trace)
constexpr const char *PHASE_INIT_SCENE = "Scene initialization", *PHASE_INIT_ACCEL = "Acceleration data structure creation",
                     *PHASE_RENDER = "Integrator::render()", *PHASE_RENDER_FORWARD = "Integrator::render_forward()",
                     *PHASE_RENDER_BACKWARD = "Integrator::render_backward()", *PHASE_DEVELOP = "Film::develop()",
                     *PHASE_PARAMS = "Scene::parameters_changed()";
}

#ifndef BLOCK
#define BLOCK 256                           // (128 measured: see profiles/r02_occupancy_ab.txt)
#endif
#define WAVES (BLOCK / 64)
#ifndef MI_EXP
#define MI_EXP 0                            // throw-away experiments (tools/*_ab.sh); 0 in every shipped build
#endif
#ifndef MI_MIN_WAVES
#define MI_MIN_WAVES 5                      // k_render*: waves per SIMD the register allocator must leave room for (94 VGPRs; at 6 it spills 20 and gains nothing: profiles/r02_occupancy_ab.txt)
#endif
#ifndef MI_MIN_WAVES_BWD
#define MI_MIN_WAVES_BWD 4                  // k_prb_backward spills at 5 (measured slower)
#endif
#ifndef MI_MIN_WAVES_1P
#define MI_MIN_WAVES_1P 4                   // k_prb_backward_1p: waves per SIMD; its LDS budget per workgroup is 160 KB / this
#endif
#define ROW_STRIDE 68                       // 64 lanes + 4 pad floats: 16-B aligned rows, shifts banks by 4 per row
#define SPLAT_ROWS 14                       // 5 x-weights + 5 y-weights + 4 channel values (separable filter)
#define WAVE_FLOATS (SPLAT_ROWS * ROW_STRIDE)
#define STACK_WORDS (MI_BVH_STACK * 64)     // per wave
#define CSTATE_WORDS(FULL) ((FULL) ? 23 : 21) // per lane, k_render_compact (21 + eta, prev_delta in FULL scenes)
#define MAX_SMALL_GRAD 1024
#define MAX_SPLAT_RECORDS (128ull << 20)   // deferred-splat queue of the particle tracer: 40 B per record (unsorted + sorted copy)
#define GRAD_QUEUE 496                      // texture-gradient records per workgroup (16 KB, 2 per lane); overflow -> direct atomics
// one LDS region per workgroup, reused in turn by: BVH traversal stacks | compaction state exchange | film splat staging
#define CMAX(a, b) ((a) > (b) ? (a) : (b))
// (only the FULL kernels traverse a BVH -- scenes with BVH meshes set features bit 3 -- so only they reserve the traversal stack:
// the all-diffuse flat-primitive kernels keep their 27 KB and five workgroups per CU)
#define REGION_BASIC CMAX(WAVES * WAVE_FLOATS, CSTATE_WORDS(false) * BLOCK)
// FULL is a three-way switch: 0 = BASIC (diffuse, flat primitives), 1 = every feature INCLUDING the BVH traversal (its LDS stack, its
// registers), 2 = every feature but no BVH in the scene (`features` bit 3 clear) -- the traversal costs the regen kernel 52 VGPRs and
// half its occupancy on scenes that have no mesh to traverse (the Veach arrangement: spheres and rectangles only).
#define REGION_FLOATS(FULL) ((FULL) ? CMAX(CMAX(WAVES * WAVE_FLOATS, CSTATE_WORDS(true) * BLOCK), (FULL) == 1 ? WAVES * STACK_WORDS : 0) : REGION_BASIC)
#define STACK_FLOATS(FULL) ((FULL) == 1 ? WAVES * STACK_WORDS : 64)
#define STK_PTR(base, FULL, wave, lane_id) (reinterpret_cast<uint32_t *>(base) + ((FULL) == 1 ? (wave) * STACK_WORDS + (lane_id) : 0u))

struct RenderArgs {
    uint32_t seed, spp_total, max_depth, rr_depth, flags;
    uint32_t pixel_begin, pixel_count, s_begin, s_count;
    uint32_t n_local;          // pixel_count * s_count
    uint32_t seg;              // lanes per pixel inside one wave for the LDS splat (0 = direct atomics)
    uint32_t lane_begin;       // OUT_LANES only
    uint32_t tile, tile_stride; // interleaved pixel tiles (0 = contiguous shard)
    uint32_t ppw;              // k_render_regen: shard pixels per wave
    uint32_t compact_mask;     // k_render_compact: bit `it` set = re-pack the workgroup's live paths after bounce `it`
    uint32_t compact_adaptive; // ... but only when that empties at least one wave
    uint32_t passes;           // k_ptracer: consecutive samples per sampler stream (mi355_render_params::passes)
};
// q-th pixel of the shard -> global pixel index (mi355_render_params::pixel_tile)
MI_DEV uint32_t shard_pixel(const RenderArgs &a, uint32_t q) {
    if (a.tile == 0u) return a.pixel_begin + q;
    uint32_t t = q / a.tile;
    return a.pixel_begin + t * a.tile_stride + (q - t * a.tile);
}

enum { OUT_FILM = 0, OUT_LANES = 1 };

// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD b mod 8), and every XCD has its own 4 MB L2.  When
// consecutive workgroups work on NEIGHBOURING data (the sorted light samples of pt_sort_lanes), giving XCD k the k-th contiguous eighth
// of the work instead of every eighth workgroup lets each L2 hold one eighth of the scene BVH instead of all of it.  The grid must be a
// multiple of 8 workgroups (xcd_grid); logical blocks beyond the work are idle.
#define MI_XCDS 8u
MI_DEV uint32_t xcd_block(uint32_t b, uint32_t n_blocks) { return (b % MI_XCDS) * (n_blocks / MI_XCDS) + b / MI_XCDS; }
static inline uint32_t xcd_grid(uint32_t n_blocks) { return (n_blocks + MI_XCDS - 1u) / MI_XCDS * MI_XCDS; }

// lane bookkeeping shared by all sampling kernels: integrator.cpp:293-310
struct LaneInfo { bool act; uint32_t lane; int32_t px, py; uint32_t pixel; };
MI_DEV LaneInfo lane_info(const DScene &sc, const RenderArgs &a, uint32_t i) {
    LaneInfo li; li.act = i < a.n_local;
    uint32_t ii = li.act ? i : 0;
    li.pixel = shard_pixel(a, ii / a.s_count);
    uint32_t s = a.s_begin + ii % a.s_count;
    li.lane = li.pixel * a.spp_total + s;
    uint32_t y = li.pixel / sc.grid_w;                                     // pos -= border_size, integrator.cpp:307-308
    li.py = (int32_t) (y + sc.crop_y) - (int32_t) sc.border; li.px = (int32_t) (li.pixel - y * sc.grid_w + sc.crop_x) - (int32_t) sc.border;
    return li;
}

// Pre-reduced film splat.  All lanes of a wave that belong to the same pixel share one 5x5 footprint, so
// the wave stages tap weights [25][64] and channel values [CH][64] in LDS and every (pixel, tap, channel)
// output is summed by one lane and added to HBM with ONE atomic.  Lanes whose footprint differs from the
// pixel's nominal one (pos rounds up to the next integer) fall back to their own atomics.
template <int CH>
MI_DEV void film_splat(const DScene &sc, const RenderArgs &a, float *smem_wave, float *film,
                       const LaneInfo &li, uint32_t i_wave_first, const Splat &sp, const float *values) {
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
    const uint32_t lane_id = threadIdx.x & 63;
    int n = (sp.count - 1) / 2;
    int32_t x0n = li.px - n - (int32_t) sc.crop_x, y0n = li.py - n - (int32_t) sc.crop_y;
    bool nominal = li.act && sp.x0 == x0n && sp.y0 == y0n;
    bool lds_path = a.seg != 0;
    if (lds_path) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            smem_wave[k * ROW_STRIDE + lane_id] = nominal ? sp.wx[k] : 0.f;
            smem_wave[(5 + k) * ROW_STRIDE + lane_id] = nominal ? sp.wy[k] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) smem_wave[(10 + c) * ROW_STRIDE + lane_id] = nominal ? values[c] : 0.f;
    }
    __syncthreads();
    if (lds_path) {
        const uint32_t seg = a.seg, groups = 64u / seg, total = groups * 25u * CH;
        for (uint32_t o = lane_id; o < total; o += 64) {
            uint32_t g = o / (25u * CH), r = o - g * (25u * CH), t = r / CH, c = r - t * CH;
            const float *xr = smem_wave + (t % 5u) * ROW_STRIDE + g * seg;
            const float *yr = smem_wave + (5u + t / 5u) * ROW_STRIDE + g * seg;
            const float *vr = smem_wave + (10 + c) * ROW_STRIDE + g * seg;
            float sum = 0.f;
            for (uint32_t l = 0; l < seg; l += 4) {
                float4 xv = *reinterpret_cast<const float4 *>(xr + l);
                float4 yv = *reinterpret_cast<const float4 *>(yr + l);
                float4 vv = *reinterpret_cast<const float4 *>(vr + l);
                sum = fmaf(yv.x * xv.x, vv.x, sum); sum = fmaf(yv.y * xv.y, vv.y, sum);      // weight = wy*wx (imageblock.cpp:486)
                sum = fmaf(yv.z * xv.z, vv.z, sum); sum = fmaf(yv.w * xv.w, vv.w, sum);
            }
            uint32_t ig = i_wave_first + g * seg;
            if (ig < a.n_local && sum != 0.f) {
                uint32_t pixel = shard_pixel(a, ig / a.s_count);
                int32_t py = (int32_t) (pixel / sc.grid_w), px = (int32_t) (pixel - (uint32_t) py * sc.grid_w) - (int32_t) sc.border; py -= (int32_t) sc.border;
                int32_t x = px - n + (int32_t) (t % 5u), y = py - n + (int32_t) (t / 5u);
                if (x >= 0 && y >= 0 && x < W && y < H) atomicAdd(film + ((size_t) y * W + x) * CH + c, sum);
            }
        }
    }
    if (li.act && !(lds_path && nominal)) {                                // imageblock.cpp:478-493
#pragma unroll
        for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
            for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {                      // static indices: keeps Splat in registers
                int32_t x = sp.x0 + tx, y = sp.y0 + ty;
                if (ty >= sp.count || tx >= sp.count || x < 0 || y < 0 || x >= W || y >= H) continue;
                float w = sp.wy[ty] * sp.wx[tx];
#pragma unroll
                for (int c = 0; c < CH; ++c) atomicAdd(film + ((size_t) y * W + x) * CH + c, values[c] * w);
            }
    }
}

MI_DEV void flush_counters(uint32_t *s_cnt, const Counters3 &cnt, unsigned long long *g_counters) {
    // s_cnt zeroed + synced by the caller
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s_cnt[0], cnt.isect); atomicAdd(&s_cnt[1], cnt.shade); atomicAdd(&s_cnt[2], cnt.shadow); }   // wave totals
    __syncthreads();
    if (threadIdx.x < 3 && g_counters) atomicAdd(&g_counters[1 + threadIdx.x], (unsigned long long) s_cnt[threadIdx.x]);
}

template <int INTEGRATOR, int OUT, int FULL>
__global__ __launch_bounds__(BLOCK, FULL ? 4 : MI_MIN_WAVES) void k_render(DScene sc, RenderArgs a, float *film, float *lanes_out,
                                                  unsigned long long *g_counters) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[REGION_FLOATS(FULL)];
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    float *smem_wave = smem + wave * WAVE_FLOATS;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    LaneInfo li;
    if (OUT == OUT_LANES) {   // lanes of the whole job, no sharding
        li.act = i < a.n_local; li.lane = a.lane_begin + (li.act ? i : 0);
        li.pixel = li.lane / a.spp_total;
        uint32_t y = li.pixel / sc.grid_w; li.py = (int32_t) (y + sc.crop_y) - (int32_t) sc.border; li.px = (int32_t) (li.pixel - y * sc.grid_w + sc.crop_x) - (int32_t) sc.border;
    } else li = lane_info(sc, a, i);
    Counters3 cnt = {};
    PROF_START(cnt);
    Pcg32 rng; sampler_seed(rng, a.seed, li.lane);
    float pos_x, pos_y; Ray ray;
    // samples_per_pass (path integrator, film output): one stream per lane, `passes` samples from it (integrator.cpp:283, :319-329);
    // all but the last are traced and splatted here, the last one by the code below
    if (INTEGRATOR == MI355_INTEGRATOR_PATH && OUT == OUT_FILM) for (uint32_t pass = 1; pass < a.passes; ++pass) {
        MI_SCENE_RELOAD(scp);
        float cw = camera_ray<FULL>(scp, rng, li.px, li.py, pos_x, pos_y, ray);
        v3 c; bool v;
        path_sample<FULL>(scp, rng, ray, a.max_depth, a.rr_depth, li.act && cw > 0.f, stk, c, v, cnt, true);
        c = c * cw;
        __syncthreads();
        Splat sp = splat_weights(scp, pos_x, pos_y);
        float values[4] = { c.x, c.y, c.z, 1.f };
        film_splat<4>(scp, a, smem_wave, film, li, blockIdx.x * BLOCK + wave * 64, sp, values);
        __syncthreads();
    }
    float cam_w = camera_ray<FULL>(sc, rng, li.px, li.py, pos_x, pos_y, ray);
    bool trace = li.act && cam_w > 0.f;                                    // rays blocked inside the lens carry weight 0
    PROF(cnt, 6);
    v3 rgb; bool valid;
    if (INTEGRATOR == MI355_INTEGRATOR_PATH)
        path_sample<FULL>(sc, rng, ray, a.max_depth, a.rr_depth, trace, stk, rgb, valid, cnt);
    else {
        GradSink none = { nullptr, nullptr, nullptr, nullptr, 0, nullptr };
        prb_sample<false, FULL>(sc, rng, ray, a.max_depth, a.rr_depth, (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0, trace, stk,
                          V3(0.f, 0.f, 0.f), V3(0.f, 0.f, 0.f), none, rgb, valid, cnt);
    }
    rgb = rgb * cam_w;                                                      // spec_u = ray_weight * spec, integrator.cpp:461
    __syncthreads();                                                        // traversal stack region is reused below
    if (OUT == OUT_LANES) {
        if (li.act) { float4 o = make_float4(rgb.x, rgb.y, rgb.z, valid ? 1.f : 0.f); reinterpret_cast<float4 *>(lanes_out)[i] = o; }
    } else {
        Splat sp = splat_weights(sc, pos_x, pos_y);
        float values[4] = { rgb.x, rgb.y, rgb.z, 1.f };                     // integrator.cpp:477-486 (rgb film, no alpha)
        PROF_START(cnt);
        film_splat<4>(sc, a, smem_wave, film, li, blockIdx.x * BLOCK + wave * 64, sp, values);
        PROF(cnt, 7);
    }
#ifdef MI_PROF
    if (lane_id == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_prof[k], (unsigned long long) cnt.prof[k]);
#endif
    flush_counters(s_cnt, cnt, g_counters);
}

// ---------------------------------------------------------------------------------------------------------------
// k_render_regen: PathIntegrator::sample with PATH REGENERATION.  In k_render a wave lives until its longest path
// ends, so on the Cornell box ~1/3 of the VALU lanes idle (measured lane utilisation 65 %).  Here a wave owns `ppw`
// whole pixels and their samples as a pool: a lane whose path has ended writes (pos, rgb) into the wave's LDS record
// ring and immediately starts the pool's next sample (wave ballot + mbcnt prefix assigns sample indices), so every
// bounce runs with (nearly) all 64 lanes busy.  Whenever 64 records are waiting they are splatted together: records
// with the same 5x5 footprint are pre-reduced through LDS exactly like film_splat does.  Lane <-> sample assignment
// changes, the arithmetic of a sample does not: a sample is still seeded by its GLOBAL lane id (sampler.cpp:124-130).
// Waves run independent trip counts: no workgroup barrier inside the loop, only wave-scope fences.
// ---------------------------------------------------------------------------------------------------------------
#define RING 128                            // records per wave: 63 waiting + 64 new at most
#define RG_WAVE_WORDS(FULL) ((FULL) == 1 ? CMAX(STACK_WORDS, WAVE_FLOATS) : WAVE_FLOATS)   // per wave and private: BVH stack | splat staging (never live together)
MI_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// splat `n` (<= 64) records starting at ring slot `head`
MI_DEV void ring_flush(const DScene &sc, float *stage, const float *ring, uint32_t head, uint32_t n, float *__restrict__ film) {
    const uint32_t lane_id = threadIdx.x & 63;
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
    const bool valid = lane_id < n;
    const uint32_t slot = (head + lane_id) & (RING - 1);
    float pos_x = valid ? ring[slot] : (float) sc.crop_x, pos_y = valid ? ring[RING + slot] : (float) sc.crop_y;
    float v0 = ring[2 * RING + slot], v1 = ring[3 * RING + slot], v2 = ring[4 * RING + slot];
    Splat sp = splat_weights(sc, pos_x, pos_y);
    const uint32_t key = ((uint32_t) (sp.x0 + 16) & 0xffffu) | ((uint32_t) (sp.y0 + 16) << 16);
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const uint32_t first = (uint32_t) __ffsll((long long) todo) - 1u;
        const uint32_t k = (uint32_t) __builtin_amdgcn_readlane((int) key, (int) first);
        const bool m = valid && key == k;
        wave_sync();                                                        // previous pass / BVH stack use is over
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            stage[j * ROW_STRIDE + lane_id] = m ? sp.wx[j] : 0.f;
            stage[(5 + j) * ROW_STRIDE + lane_id] = m ? sp.wy[j] : 0.f;
        }
        stage[10 * ROW_STRIDE + lane_id] = m ? v0 : 0.f; stage[11 * ROW_STRIDE + lane_id] = m ? v1 : 0.f;
        stage[12 * ROW_STRIDE + lane_id] = m ? v2 : 0.f; stage[13 * ROW_STRIDE + lane_id] = m ? 1.f : 0.f;   // integrator.cpp:477-486
        wave_sync();
        const int32_t x0 = (int32_t) (k & 0xffffu) - 16, y0 = (int32_t) (k >> 16) - 16;
        for (uint32_t o = lane_id; o < 100u; o += 64) {
            uint32_t t = o >> 2, c = o & 3u, tx = t % 5u, ty = t / 5u;
            const float *xr = stage + tx * ROW_STRIDE, *yr = stage + (5u + ty) * ROW_STRIDE, *vr = stage + (10u + c) * ROW_STRIDE;
            float sum = 0.f;
#pragma unroll 4
            for (uint32_t l = 0; l < 64; l += 4) {
                float4 xv = *reinterpret_cast<const float4 *>(xr + l);
                float4 yv = *reinterpret_cast<const float4 *>(yr + l);
                float4 vv = *reinterpret_cast<const float4 *>(vr + l);
                sum = fmaf(yv.x * xv.x, vv.x, sum); sum = fmaf(yv.y * xv.y, vv.y, sum);      // weight = wy*wx (imageblock.cpp:486)
                sum = fmaf(yv.z * xv.z, vv.z, sum); sum = fmaf(yv.w * xv.w, vv.w, sum);
            }
            int32_t x = x0 + (int32_t) tx, y = y0 + (int32_t) ty;
            if (sum != 0.f && x >= 0 && y >= 0 && x < W && y < H) atomicAdd(film + ((size_t) y * W + x) * 4 + c, sum);
        }
        todo &= ~__ballot(m);
    }
}

#ifndef MI_REGEN_WAVES_FLAT
#define MI_REGEN_WAVES_FLAT 4
#endif
template <int FULL>
__global__ __launch_bounds__(BLOCK, FULL == 1 ? 2 : (FULL ? MI_REGEN_WAVES_FLAT : MI_MIN_WAVES)) void k_render_regen(DScene sc, RenderArgs a, float *__restrict__ film, unsigned long long *__restrict__ g_counters) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[WAVES * RG_WAVE_WORDS(FULL)];
    __shared__ float s_ring[WAVES * 5 * RING];
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    const uint32_t lane_id = threadIdx.x & 63;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    float *stage = smem + wave * RG_WAVE_WORDS(FULL);
    uint32_t *stk = reinterpret_cast<uint32_t *>(stage) + lane_id;
    float *ring = s_ring + wave * 5 * RING;
    // pool of this wave: shard pixels [q, q_end), all s_count samples of each
    uint32_t q = (blockIdx.x * WAVES + wave) * a.ppw;
    const uint32_t q_end = min(q + a.ppw, a.pixel_count);
    uint32_t s_next = 0, r_head = 0, r_count = 0;
    PathState st; path_init(st, Ray{ V3(0.f, 0.f, 0.f), V3(0.f, 0.f, 1.f), 0.f }, Pcg32{ 0, 1 });
    float pos_x = 0.f, pos_y = 0.f, cam_w = 1.f;
    uint32_t depth = 0;
    bool active = false, pending = false;
    Counters3 cnt = {};
    for (;;) {
        // The sample-generation and splat sections read the scene through a pointer the optimiser cannot see through,
        // re-derived every trip: otherwise LICM hoists their camera / filter constants out of the loop, where they
        // sit in ~80 spilled SGPRs across the (much hotter) bounce code.  DScene is kernel argument 0.
        typedef const __attribute__((address_space(4))) DScene *KScene;
        KScene scp = (KScene) __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(scp));
        const DScene &scl = *(const DScene *) scp;
        // (1) paths that ended in the previous bounce (or were blocked inside the lens) become film records
        unsigned long long fin = __ballot(pending);
        if (fin) {
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (fin >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) fin, 0u));
            if (pending) {
                uint32_t slot = (r_head + r_count + rank) & (RING - 1);
                ring[slot] = pos_x; ring[RING + slot] = pos_y;
                ring[2 * RING + slot] = st.result.x * cam_w; ring[3 * RING + slot] = st.result.y * cam_w;   // spec_u = ray_weight * spec, integrator.cpp:461
                ring[4 * RING + slot] = st.result.z * cam_w;
            }
            r_count += (uint32_t) __popcll(fin); pending = false;
        }
        // (2) 64 records waiting: one pre-reduced splat
        if (r_count >= 64u) { ring_flush(scl, stage, ring, r_head, 64u, film); r_head = (r_head + 64u) & (RING - 1); r_count -= 64u; }
        // (3) idle lanes take the next samples of the current pixel (render_sample head, integrator.cpp:433-455)
        unsigned long long idle = __ballot(!active);
        if (idle && q < q_end) {
            const uint32_t remaining = a.s_count - s_next;
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
            const uint32_t pixel = shard_pixel(a, q);                          // wave-uniform
            const uint32_t y = pixel / scl.grid_w;
            const int32_t py = (int32_t) (y + scl.crop_y) - (int32_t) scl.border, px = (int32_t) (pixel - y * scl.grid_w + scl.crop_x) - (int32_t) scl.border;
            if (!active && rank < remaining) {
                uint32_t lane = pixel * a.spp_total + a.s_begin + s_next + rank;   // global lane id, integrator.cpp:293-305
                Pcg32 rng; sampler_seed(rng, a.seed, lane);
                Ray ray; cam_w = camera_ray<FULL>(scl, rng, px, py, pos_x, pos_y, ray);
                path_init(st, ray, rng);
                depth = 0; active = cam_w > 0.f; pending = !active;          // rays blocked inside the lens carry weight 0
            }
            s_next += min((uint32_t) __popcll(idle), remaining);
            if (s_next == a.s_count) { s_next = 0; ++q; }
        }
        if (!__any(active)) { if (q >= q_end && !__any(pending)) break; continue; }
        // (4) one bounce of every live path (path.cpp:142-281); the depth is per lane now
        bool hit_any;
        bool alive = path_step<FULL>(scl, st, depth, a.max_depth, a.rr_depth, active, stk, hit_any, cnt);
        pending = pending || (active && !alive); active = alive; ++depth;
    }
    if (r_count) ring_flush(sc, stage, ring, r_head, r_count, film);       // r_count < 64 here
    __syncthreads();
    flush_counters(s_cnt, cnt, g_counters);
}

// ---------------------------------------------------------------------------------------------------------------
// k_render_compact: same work as k_render<PATH, FILM>, but after every bounce the workgroup re-packs its surviving
// lanes into the lowest thread slots (wave ballot + popcount prefix, state exchanged through LDS).  Waves whose slots
// are all empty skip the bounce and just meet the others at the barriers, so the VALU time of a workgroup shrinks
// from "6 bounces x 4 waves" to "sum over bounces of ceil(alive / 64)".  Final radiance returns to the lane's ORIGINAL
// slot (`fin`) so that the per-pixel LDS film reduction still sees all samples of a pixel side by side.
// ---------------------------------------------------------------------------------------------------------------
#ifndef MI_COMPACT_WAVES_FLAT
#define MI_COMPACT_WAVES_FLAT 5               // (Veach arrangement, re-packed: 3 / 4 / 5 waves per SIMD = 144 / 117 / 109 ms)
#endif
template <int INTEGRATOR, int FULL>
__global__ __launch_bounds__(BLOCK, FULL == 2 ? MI_COMPACT_WAVES_FLAT : (FULL ? 4 : MI_MIN_WAVES)) void k_render_compact(DScene sc, RenderArgs a, float *film, unsigned long long *g_counters) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[REGION_FLOATS(FULL)];         // BVH stack | compaction state | splat staging
    __shared__ float fin[4 * BLOCK];                                           // final (r, g, b, valid) per ORIGINAL slot
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_wcnt[2 * WAVES];
#ifdef MI_LDS_PAD                                                              // occupancy experiments only (tools/occupancy_ab.sh): extra LDS per workgroup
    __shared__ float s_pad[MI_LDS_PAD];
    if (a.seed == 0xfffffffeu && a.max_depth == 0xffffu) s_pad[threadIdx.x % MI_LDS_PAD] = 0.f;
#endif
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane_id = tid & 63;
    float *smem_wave = smem + wave * WAVE_FLOATS;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    uint32_t i = blockIdx.x * BLOCK + tid;
    Counters3 cnt = {};
    PROF_START(cnt);
    LaneInfo li = lane_info(sc, a, i);
    Pcg32 rng; sampler_seed(rng, a.seed, li.lane);
    float pos_x, pos_y; Ray ray0;
    const float cam_w = camera_ray<FULL>(sc, rng, li.px, li.py, pos_x, pos_y, ray0);   // stays with the ORIGINAL slot, like pos
    PathState st; path_init(st, ray0, rng);
    fin[tid] = 0.f; fin[BLOCK + tid] = 0.f; fin[2 * BLOCK + tid] = 0.f; fin[3 * BLOCK + tid] = 0.f;
    uint32_t orig = tid;                       // which original slot the path in this thread belongs to
    bool valid_ray = false;
    bool alive = li.act && cam_w > 0.f;                                      // this thread currently holds a live path
    for (uint32_t it = 0; it < a.max_depth; ++it) {
        const bool mine = alive;
        alive = false;
        PROF(cnt, 6);
        if (__any(mine)) {                                                  // wave-uniform: empty waves skip the bounce
            bool hit_any;
            MI_SCENE_RELOAD(scb);                                           // re-derived every bounce (mi355_device.h)
            if (INTEGRATOR == MI355_INTEGRATOR_PATH) alive = path_step<FULL>(scb, st, it, a.max_depth, a.rr_depth, mine, stk, hit_any, cnt);
            else alive = prb_primal_step<FULL>(scb, st, it, a.max_depth, a.rr_depth, mine, stk, hit_any, cnt);
            valid_ray = valid_ray || hit_any;
        }
        PROF_START(cnt);                                                    // (section 6: everything between two path steps -- barriers, re-pack)
        if (mine && !alive) {                                               // path over: park its radiance in its original slot
            bool v = valid_ray;
            fin[orig] = v ? st.result.x : 0.f; fin[BLOCK + orig] = v ? st.result.y : 0.f; fin[2 * BLOCK + orig] = v ? st.result.z : 0.f;
            fin[3 * BLOCK + orig] = v ? 1.f : 0.f;
        }
        if (!((a.compact_mask >> (it < 31u ? it : 31u)) & 1u) || it + 1 >= a.max_depth) continue;   // no re-pack after this bounce
        unsigned long long m = __ballot(alive);
        uint32_t *wcnt = s_wcnt + (it & 1u) * WAVES;                        // double-buffered: a fast wave may already be one bounce ahead
        if (lane_id == 0) wcnt[wave] = (uint32_t) __popcll(m);
        __syncthreads();                                                    // everyone is done with the stack / previous state
        uint32_t total = 0, busy = 0;
#pragma unroll
        for (uint32_t w = 0; w < WAVES; ++w) { uint32_t c = wcnt[w]; total += c; busy += c ? 1u : 0u; }
        if (total == 0u) break;                                             // (workgroup-uniform) every path of the workgroup has ended
        const uint32_t needed = (total + 63u) / 64u;                        // waves that stay busy
        if (a.compact_adaptive && needed >= busy) continue;                 // (workgroup-uniform) re-packing would not empty a wave
        // Waves [0, needed) keep their live paths where they are; the live paths of the other waves ("movers") fill the dead
        // slots ("holes") of the kept waves, rank to rank.  total <= 64 * needed guarantees a hole for every mover.
        uint32_t hole_base = 0, mover_base = 0, n_movers = 0;
#pragma unroll
        for (uint32_t w = 0; w < WAVES; ++w) {
            uint32_t c = wcnt[w];
            if (w < needed) hole_base += w < wave ? 64u - c : 0u;
            else { mover_base += w < wave ? c : 0u; n_movers += c; }
        }
        const bool kept = wave < needed;
        if (!kept && alive) {
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
            float *q = smem + mover_base + rank;
            q[0 * BLOCK] = st.ray.o.x; q[1 * BLOCK] = st.ray.o.y; q[2 * BLOCK] = st.ray.o.z;
            q[3 * BLOCK] = st.ray.d.x; q[4 * BLOCK] = st.ray.d.y; q[5 * BLOCK] = st.ray.d.z;
            q[6 * BLOCK] = st.throughput.x; q[7 * BLOCK] = st.throughput.y; q[8 * BLOCK] = st.throughput.z;
            q[9 * BLOCK] = st.result.x; q[10 * BLOCK] = st.result.y; q[11 * BLOCK] = st.result.z;
            q[12 * BLOCK] = st.prev_p.x; q[13 * BLOCK] = st.prev_p.y; q[14 * BLOCK] = st.prev_p.z;
            q[15 * BLOCK] = st.prev_bsdf_pdf;
            q[16 * BLOCK] = __uint_as_float((uint32_t) st.rng.state); q[17 * BLOCK] = __uint_as_float((uint32_t) (st.rng.state >> 32));
            q[18 * BLOCK] = __uint_as_float((uint32_t) st.rng.inc); q[19 * BLOCK] = __uint_as_float((uint32_t) (st.rng.inc >> 32));
            q[20 * BLOCK] = __uint_as_float(orig);
            if (FULL) { q[21 * BLOCK] = st.eta; q[22 * BLOCK] = st.prev_delta ? 1.f : 0.f; }
        }
        __syncthreads();
        if (!kept) alive = false;
        else if (!alive) {
            const unsigned long long dm = ~m;                               // dead slots of this (kept) wave
            uint32_t hr = hole_base + __builtin_amdgcn_mbcnt_hi((uint32_t) (dm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) dm, 0u));
            if (hr < n_movers) {
                const float *q = smem + hr;
                st.ray.o = V3(q[0 * BLOCK], q[1 * BLOCK], q[2 * BLOCK]); st.ray.d = V3(q[3 * BLOCK], q[4 * BLOCK], q[5 * BLOCK]);
                st.ray.maxt = MI_LARGEST;
                st.throughput = V3(q[6 * BLOCK], q[7 * BLOCK], q[8 * BLOCK]); st.result = V3(q[9 * BLOCK], q[10 * BLOCK], q[11 * BLOCK]);
                st.prev_p = V3(q[12 * BLOCK], q[13 * BLOCK], q[14 * BLOCK]); st.prev_bsdf_pdf = q[15 * BLOCK];
                st.rng.state = (uint64_t) __float_as_uint(q[16 * BLOCK]) | ((uint64_t) __float_as_uint(q[17 * BLOCK]) << 32);
                st.rng.inc = (uint64_t) __float_as_uint(q[18 * BLOCK]) | ((uint64_t) __float_as_uint(q[19 * BLOCK]) << 32);
                orig = __float_as_uint(q[20 * BLOCK]);
                if (FULL) { st.eta = q[21 * BLOCK]; st.prev_delta = q[22 * BLOCK] != 0.f; }
                valid_ray = true;                                           // a path that continues has hit a surface
                alive = true;
            }
        }
        if (sc.n_bvh_shapes) __syncthreads();                               // the BVH stack aliases the state just read
    }
    __syncthreads();
    PROF(cnt, 6);
    float values[4] = { fin[tid] * cam_w, fin[BLOCK + tid] * cam_w, fin[2 * BLOCK + tid] * cam_w, 1.f };   // integrator.cpp:461, 477-486
    MI_SCENE_RELOAD(scl);                                                   // the splat's filter table / crop window: fetched here, not before the loop
    Splat sp = splat_weights(scl, pos_x, pos_y);
    __syncthreads();
    film_splat<4>(scl, a, smem_wave, film, li, blockIdx.x * BLOCK + wave * 64, sp, values);
    PROF(cnt, 7);
#ifdef MI_PROF
    if (lane_id == 0) for (int k = 0; k < 8; ++k) atomicAdd(&g_prof[k], (unsigned long long) cnt.prof[k]);
#endif
    flush_counters(s_cnt, cnt, g_counters);
}

// RBIntegrator.render_backward step 1a: common.py:741-758 with L = 1, weight = 1
__global__ __launch_bounds__(BLOCK) void k_splat_weights(DScene sc, RenderArgs a, float *weight) {
    __shared__ __attribute__((aligned(16))) float smem[WAVES * WAVE_FLOATS];   // splat staging only
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    LaneInfo li = lane_info(sc, a, i);
    Pcg32 rng; sampler_seed(rng, a.seed, li.lane);
    float pos_x = (float) li.px + next_1d(rng), pos_y = (float) li.py + next_1d(rng);
    Splat sp = splat_weights(sc, pos_x, pos_y);
    float one[1] = { 1.f };
    film_splat<1>(sc, a, smem + wave * WAVE_FLOATS, weight, li, blockIdx.x * BLOCK + wave * 64, sp, one);
}

// RBIntegrator.render_backward steps 1b-3: common.py:741-796
// grad_in / (W == 0 ? 1 : W), one float4 per pixel: the adjoint of HDRFilm::develop's division (hdrfilm.cpp:396-400)
__global__ void k_prepare_grad(const float *grad_in, const float *weight, float4 *gw, uint32_t n_pixels) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float wv = weight[i]; float inv = 1.0f / (wv == 0.f ? 1.f : wv);
    gw[i] = make_float4(grad_in[3 * i] * inv, grad_in[3 * i + 1] * inv, grad_in[3 * i + 2] * inv, 0.f);
}

// RBIntegrator.render_forward (common.py:498-629): primal walk for L, forward-mode walk (same sampler stream) contracting every
// d(Le + Lr_dir + Lr_ind)/d(theta) with the parameter tangents, dL splatted with weight 1 like a radiance sample.
// (BASIC at 5 waves per SIMD -- 96 VGPRs with a few spilled -- measured 5.94 against 6.17 ms at 4.)
template <int FULL>
__global__ __launch_bounds__(BLOCK, FULL ? 3 : 5) void k_render_forward(DScene sc, RenderArgs a, const float *tangent, const uint32_t *small_map,
                                                                                      float *film, unsigned long long *g_counters) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[REGION_FLOATS(FULL)];
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    float *smem_wave = smem + wave * WAVE_FLOATS;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    LaneInfo li = lane_info(sc, a, i);
    Pcg32 rng; sampler_seed(rng, a.seed, li.lane);
    float pos_x, pos_y; Ray ray;
    float cam_w = camera_ray<FULL>(sc, rng, li.px, li.py, pos_x, pos_y, ray);
    bool trace = li.act && cam_w > 0.f;
    Counters3 cnt = {};
    const bool hide = (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0;
    GradSink none = { nullptr, nullptr, nullptr, nullptr, 0, nullptr }, sink = { nullptr, const_cast<float *>(tangent), nullptr, nullptr, 0, small_map };
    Pcg32 rng2 = rng;                                                       // sampler.clone(), common.py:578
    v3 L, dl; bool valid;
    prb_sample<false, FULL>(sc, rng2, ray, a.max_depth, a.rr_depth, hide, trace, stk, V3(0.f, 0.f, 0.f), V3(0.f, 0.f, 0.f), none, L, valid, cnt);
    prb_sample<true, FULL, true>(sc, rng, ray, a.max_depth, a.rr_depth, hide, trace, stk, V3(1.f, 1.f, 1.f), L, sink, dl, valid, cnt);
    dl = (trace ? dl : V3(0.f, 0.f, 0.f)) * cam_w;                          // value = dL * weight, common.py:604-611
    __syncthreads();                                                        // traversal stack region is reused below
    Splat sp = splat_weights(sc, pos_x, pos_y);
    float values[4] = { dl.x, dl.y, dl.z, 1.f };
    film_splat<4>(sc, a, smem_wave, film, li, blockIdx.x * BLOCK + wave * 64, sp, values);
    flush_counters(s_cnt, cnt, g_counters);
}

template <int FULL>
__global__ __launch_bounds__(BLOCK, FULL ? 3 : MI_MIN_WAVES_BWD) void k_prb_backward(DScene sc, RenderArgs a, const float4 *gw,
                                                        float *g_grad, const uint32_t *small_map, uint32_t n_small,
                                                        unsigned long long *g_counters, const uint32_t *redo) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    if (redo && !redo[blockIdx.x]) return;                                  // second launch: only the workgroups the one-pass kernel gave up on
    __shared__ __attribute__((aligned(16))) float smem[STACK_FLOATS(FULL)];
    __shared__ float s_grad[MAX_SMALL_GRAD];
    __shared__ float s_q[GRAD_QUEUE * 8];
    __shared__ uint32_t s_qn;
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_qn = 0;
    for (uint32_t k = threadIdx.x; k < n_small; k += BLOCK) s_grad[k] = 0.f;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    LaneInfo li = lane_info(sc, a, i);
    Pcg32 rng; sampler_seed(rng, a.seed, li.lane);
    float pos_x, pos_y; Ray ray;
    float cam_w = camera_ray<FULL>(sc, rng, li.px, li.py, pos_x, pos_y, ray);
    bool trace = li.act && cam_w > 0.f;
    // dL = adjoint of splat + develop: sum over taps of w(p) * grad_in[p] / W[p]   (hdrfilm.cpp:400); gw = grad_in / W
    Splat sp = splat_weights(sc, pos_x, pos_y);
    v3 dL = V3(0.f, 0.f, 0.f);
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
    if (li.act) {
#pragma unroll
        for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
            for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
                int32_t x = sp.x0 + tx, y = sp.y0 + ty;
                if (ty >= sp.count || tx >= sp.count || x < 0 || y < 0 || x >= W || y >= H) continue;
                float4 g = gw[(size_t) y * W + x];
                float w = sp.wy[ty] * sp.wx[tx];
                dL.x = fmaf(g.x, w, dL.x); dL.y = fmaf(g.y, w, dL.y); dL.z = fmaf(g.z, w, dL.z);
            }
    }
    dL = dL * cam_w;                                                        // value = L * weight, common.py:752
    Counters3 cnt = {};
    bool hide = (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0;
    GradSink none = { nullptr, nullptr, nullptr, nullptr, 0, nullptr }, sink = { s_grad, g_grad, s_q, &s_qn, GRAD_QUEUE, nullptr };
    Pcg32 rng2 = rng;                                                       // sampler.clone(), common.py:767
    v3 L, L2; bool valid;
    prb_sample<false, FULL>(sc, rng2, ray, a.max_depth, a.rr_depth, hide, trace, stk, V3(0.f, 0.f, 0.f), V3(0.f, 0.f, 0.f), none, L, valid, cnt);
    prb_sample<true, FULL>(sc, rng, ray, a.max_depth, a.rr_depth, hide, trace, stk, dL, L, sink, L2, valid, cnt);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_small; k += BLOCK) { float v = s_grad[k]; if (v != 0.f) atomicAdd(g_grad + small_map[k], v); }
    // flush the texture-gradient queue: 12 consecutive lanes expand one record into its (tap, channel) atomics, so the
    // three channels of a texel (12 contiguous bytes) leave in one wave-instruction and share a 64-B request
    {
        uint32_t n = min(s_qn, (uint32_t) GRAD_QUEUE) * 12u;
        for (uint32_t e = threadIdx.x; e < n; e += BLOCK) {
            uint32_t rec = e / 12u, k = e - rec * 12u, tap = k / 3u, ch = k - tap * 3u;
            const float *r = s_q + 8u * rec;
            const DTexture &tx = sc.textures[__float_as_uint(r[0])];
            uint32_t i00 = __float_as_uint(r[1]), i11 = __float_as_uint(r[2]);
            uint32_t x0 = i00 % tx.w, x1 = i11 % tx.w;
            uint32_t idx = tap == 0 ? i00 : (tap == 3 ? i11 : (tap == 1 ? i00 - x0 + x1 : i11 - x1 + x0));
            float wx = (tap & 1u) ? r[3] : 1.f - r[3], wy = (tap & 2u) ? r[4] : 1.f - r[4];
            float v = (wy * wx) * r[5 + ch];
            if (v != 0.f) atomicAdd(g_grad + tx.grad_offset + (size_t) idx * 3u + ch, v);
        }
    }
    flush_counters(s_cnt, cnt, g_counters);
}

// One-pass adjoint (prb_sample_1p): the path is walked ONCE; gradient records wait in LDS for the lane's L_total.
// BVH = the scene has meshes that need the per-lane traversal stack (20 KB of LDS).  The small-parameter accumulators and
// the record queue live in dynamic LDS: the host hands over whatever is left of 40 KB (four workgroups per CU) --
// ~36 KB of queue on the Cornell box, ~16 KB with a traversal stack.
template <int FULL, bool BVH>
__global__ __launch_bounds__(BLOCK, FULL ? 3 : MI_MIN_WAVES_1P) void k_prb_backward_1p(DScene sc, RenderArgs a, const float4 *gw,
                                                        float *g_grad, const uint32_t *small_map, uint32_t n_small,
                                                        unsigned long long *g_counters, uint32_t *redo, uint32_t q_granules, uint32_t copy_shift) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    extern __shared__ __attribute__((aligned(16))) float4 dyn_lds[];        // [(n_small << copy_shift) rounded up to 4 floats][q_granules x 16 B]
    __shared__ __attribute__((aligned(16))) float smem[BVH ? WAVES * STACK_WORDS : 64];
    __shared__ float s_ltot[3 * BLOCK];
    __shared__ uint32_t s_texinfo[2 * 32];                                   // (grad_offset, channels) of the first 32 textures: keeps a
                                                                            // dependent global load out of the resolve loops
    float *s_grad = reinterpret_cast<float *>(dyn_lds);
    const uint32_t n_acc = n_small << copy_shift;
    unsigned long long *s_fix = reinterpret_cast<unsigned long long *>(dyn_lds + (n_acc + 3u) / 4u);   // the accumulators' 64-bit fixed-point twins
    float4 *s_q = dyn_lds + (n_acc + 3u) / 4u + (n_acc + 1u) / 2u;
    __shared__ uint32_t s_qn, s_over, s_bmax;
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_qn = 0; s_over = 0; s_bmax = 0; }
    for (uint32_t k = threadIdx.x; k < n_acc; k += BLOCK) { s_grad[k] = 0.f; s_fix[k] = 0ull; }
    if (threadIdx.x < 32u && threadIdx.x < sc.n_textures) {
        s_texinfo[2 * threadIdx.x] = (uint32_t) sc.textures[threadIdx.x].grad_offset; s_texinfo[2 * threadIdx.x + 1] = sc.textures[threadIdx.x].c;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    uint32_t *stk = reinterpret_cast<uint32_t *>(smem + (BVH ? wave * STACK_WORDS : 0u)) + (BVH ? lane_id : 0u);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    LaneInfo li = lane_info(sc, a, i);
    Pcg32 rng; sampler_seed(rng, a.seed, li.lane);
    float pos_x, pos_y; Ray ray;
    float cam_w = camera_ray<FULL>(sc, rng, li.px, li.py, pos_x, pos_y, ray);
    bool trace = li.act && cam_w > 0.f;
    // dL = adjoint of splat + develop: sum over taps of w(p) * grad_in[p] / W[p]   (hdrfilm.cpp:400); gw = grad_in / W
    Splat sp = splat_weights(sc, pos_x, pos_y);
    v3 dL = V3(0.f, 0.f, 0.f);
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
    if (li.act) {
#pragma unroll
        for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
            for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
                int32_t x = sp.x0 + tx, y = sp.y0 + ty;
                if (ty >= sp.count || tx >= sp.count || x < 0 || y < 0 || x >= W || y >= H) continue;
                float4 g = gw[(size_t) y * W + x];
                float w = sp.wy[ty] * sp.wx[tx];
                dL.x = fmaf(g.x, w, dL.x); dL.y = fmaf(g.y, w, dL.y); dL.z = fmaf(g.z, w, dL.z);
            }
    }
    dL = dL * cam_w;                                                        // value = L * weight, common.py:752
    // exponent of the fixed-point accumulators: B = max |dL| over the workgroup (bit patterns of non-negative floats order like the values;
    // a NaN sorts above infinity and switches the workgroup to the fp32 accumulators)
    { const float b = fmaxf(fabsf(dL.x), fmaxf(fabsf(dL.y), fabsf(dL.z)));
      const uint32_t bits = (dL.x != dL.x || dL.y != dL.y || dL.z != dL.z) ? 0x7fc00000u : __float_as_uint(b);
      if (bits) atomicMax(&s_bmax, bits); }                                 // (once per lane and launch: the same-address conflict does not matter here)
    __syncthreads();
    float fix_scale = 0.f, fix_limit = 0.f, fix_inv = 0.f;
    { const float B = __uint_as_float(s_bmax);
      int e = 0; (void) frexpf(B, &e);                                      // B < 2^e
      if (B > 0.f && B < MI_INF && e > -90 && e < 100) { fix_scale = ldexpf(1.f, 36 - e); fix_inv = ldexpf(1.f, e - 36); fix_limit = ldexpf(1.f, 14 + e); } }
    Counters3 cnt = {};
    bool hide = (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0;
    GradSink1 sink = { s_grad, copy_shift, s_q, &s_qn, q_granules, &s_over, g_grad != nullptr, s_fix, fix_scale, fix_limit };
    v3 L;
    prb_sample_1p<FULL>(sc, rng, ray, a.max_depth, a.rr_depth, hide, trace, stk, dL, sink, L, cnt);
    s_ltot[threadIdx.x] = L.x * dL.x; s_ltot[BLOCK + threadIdx.x] = L.y * dL.y; s_ltot[2 * BLOCK + threadIdx.x] = L.z * dL.z;   // L_total * dL
    __syncthreads();
    if (s_over) {                                                           // uniform: leave this workgroup to the two-pass kernel
        if (threadIdx.x == 0) { redo[blockIdx.x] = 1u; if (g_counters) atomicAdd(&g_counters[4], 1ull); }
        flush_counters(s_cnt, Counters3{}, g_counters);                     // its events are counted when it is redone
        return;
    }
    // resolve: ga = L_total(lane) * A' + C'
    const uint32_t n_rec_small = s_qn & 0xffffu, n_rec_tex = s_qn >> 16;
    for (uint32_t g = threadIdx.x; g < n_rec_small; g += BLOCK) {           // one dword per record: bsdf << 8 | lane
        const uint32_t tag = reinterpret_cast<const uint32_t *>(s_q)[g], lane = tag & 255u;
        const DBsdf &b = sc.bsdfs[tag >> 8];
        const float i0 = b.reflectance[0] != 0.f ? rcp_(b.reflectance[0]) : 0.f, i1 = b.reflectance[1] != 0.f ? rcp_(b.reflectance[1]) : 0.f,
                    i2 = b.reflectance[2] != 0.f ? rcp_(b.reflectance[2]) : 0.f;           // inv = 0 where bsdf_weight * pdf = 0 (prb.py:213-227)
        small_grad_add(sink, b.grad_slot, V3(s_ltot[lane] * i0, s_ltot[BLOCK + lane] * i1, s_ltot[2 * BLOCK + lane] * i2));
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_small; k += BLOCK) {
        float v = 0.f; unsigned long long f = 0ull;
        for (uint32_t c = 0; c < (1u << copy_shift); ++c) { v += s_grad[(k << copy_shift) + c]; f += s_fix[(k << copy_shift) + c]; }
        v += fix_to_float(f, fix_inv);
        if (v != 0.f) atomicAdd(g_grad + small_map[k], v);
    }
    // Texture records.  Device-scope float atomics are executed on the memory side on this part (the per-XCD L2s are not
    // coherent with each other; PMC: TCC_EA0_ATOMIC = number of atomics, ~40 B of HBM write each), and the 4 x 64 samples of a
    // workgroup hit the same few texels many times, so the (record, tap) contributions are first summed per texel in an LDS
    // hash table (open addressing in the granules the constant-reflectance records no longer need) and only the distinct
    // texels go out: 3.2x fewer atomics on C3.  Whatever does not fit the table takes the direct path.  (Fixed-point sums in the table,
    // as the small-parameter accumulators have them, were measured at no gain -- 4.47 vs 4.50 ms -- and halve its capacity: not done.)
    __syncthreads();
    const uint32_t EMPTY = 0xffffffffu;
    uint32_t T = 0;
    if (n_rec_tex) {                                                        // table of ~2 entries per record, whatever the free granules allow
        const uint32_t free_gran = q_granules - 3u * n_rec_tex;             // (the dword records at the bottom are resolved by now)
        T = 64u; while (T < 2u * n_rec_tex && T < 2048u) T <<= 1;
        while (T > free_gran) T >>= 1;
        if (T < 64u) T = 0u;
    }
    for (uint32_t k = threadIdx.x; k < T; k += BLOCK) s_q[k] = make_float4(__uint_as_float(EMPTY), 0.f, 0.f, 0.f);
    __syncthreads();
    // Records sit in the queue in allocation order, i.e. the 64 first-hit records of one pixel (same texels!) are neighbours.
    // Walking them with a stride that is coprime to their number puts different pixels into the lanes of one LDS atomic
    // instruction, which would otherwise serialise on the shared table entry (PMC: LDS unit 50 % busy before this).
    const uint32_t stride = n_rec_tex % 61u ? 61u : (n_rec_tex % 67u ? 67u : 1u);
#if MI_EXP == 2
    if (a.seed != 0xdeadbeefu) T = 0; else
#endif
    for (uint32_t e = threadIdx.x; e < n_rec_tex * 4u; e += BLOCK) {
        const uint32_t rec = (uint32_t) (((uint64_t) (e >> 2) * stride) % n_rec_tex), tap = e & 3u;
        const uint32_t g = q_granules - 3u * (rec + 1u);
        float4 r0 = s_q[g], r1 = s_q[g + 1], r2 = s_q[g + 2];
        const uint32_t tag = __float_as_uint(r0.x), lane = tag & 255u, tex = tag >> 8;
        const float wx = (tap & 1u) ? r0.w : 1.f - r0.w, wy = (tap & 2u) ? r1.x : 1.f - r1.x, w = wy * wx;
        if (w == 0.f) continue;
        const uint32_t i00 = __float_as_uint(r0.y), i11 = __float_as_uint(r0.z), dx = __float_as_uint(r2.x);
        const uint32_t idx = tap == 0 ? i00 : (tap == 3 ? i11 : (tap == 1 ? i00 + dx : i11 - dx));
        const float v0 = w * fmaf(s_ltot[lane], r1.y, r2.y), v1 = w * fmaf(s_ltot[BLOCK + lane], r1.z, r2.z),
                    v2 = w * fmaf(s_ltot[2 * BLOCK + lane], r1.w, r2.w);
        const uint32_t tx_off = tex < 32u ? s_texinfo[2 * tex] : (uint32_t) sc.textures[tex].grad_offset;
        const uint32_t tx_c = tex < 32u ? s_texinfo[2 * tex + 1] : sc.textures[tex].c;
        bool done = false;
        if (T && tx_c == 3 && tex < 255u && idx < (1u << 24)) {
            const uint32_t key = (tex << 24) | idx;
            uint32_t h = (idx * 2654435761u) >> 8;
            for (int probe = 0; probe < 8 && !done; ++probe, ++h) {
                uint32_t *slot = reinterpret_cast<uint32_t *>(&s_q[h & (T - 1u)]);
                uint32_t old = atomicCAS(slot, EMPTY, key);
                if (old == EMPTY || old == key) {
                    float *acc = reinterpret_cast<float *>(slot);
                    atomicAdd(acc + 1, v0); atomicAdd(acc + 2, v1); atomicAdd(acc + 3, v2);
                    done = true;
                }
            }
        }
#if MI_EXP == 1
        done = true;
#endif
        if (!done) {
            float *gb = g_grad + tx_off;
            if (tx_c == 3) { atomicAdd(gb + (size_t) idx * 3u, v0); atomicAdd(gb + (size_t) idx * 3u + 1, v1); atomicAdd(gb + (size_t) idx * 3u + 2, v2); }
            else atomicAdd(gb + idx, v0 + v1 + v2);                          // 1-channel texture: the three channel gradients add up
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < T * 3u; e += BLOCK) {                 // 3 consecutive lanes = the 12 contiguous bytes of one texel
        const uint32_t ent = e / 3u, ch = e - ent * 3u;
        const uint32_t key = reinterpret_cast<const uint32_t *>(&s_q[ent])[0];
        if (key == EMPTY) continue;
        const float v = reinterpret_cast<const float *>(&s_q[ent])[1 + ch];
        const uint32_t tex = key >> 24, off = tex < 32u ? s_texinfo[2 * tex] : (uint32_t) sc.textures[tex].grad_offset;
#if MI_EXP == 1
        if (v == 12345.678f)
#endif
        if (v != 0.f) atomicAdd(g_grad + off + (size_t) (key & 0xffffffu) * 3u + ch, v);
    }
    flush_counters(s_cnt, cnt, g_counters);
}

// bound_exit_pupil / compute_exit_pupil_bounds (realisticlens.cpp:1685-1806): 64 rings of film radii x 2^20 rays each towards a
// (1.5 x rear radius)^2 square on the rear plane, sampled with radical inverses in bases 2 and 3; the bound of a ring is the
// bounding box of the rays that make it through the lens.  min/max of floats through order-preserving unsigned keys.
MI_DEV uint32_t float_key(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static inline float key_float(uint32_t k) { uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; memcpy(&f, &u, 4); return f; }
#define PUPIL_SEGMENTS 64
#define PUPIL_RAYS (1u << 20)
__global__ __launch_bounds__(256) void k_exit_pupil(const float *lens, uint32_t lens_count, float film_diagonal, uint32_t *keys /* [64][5] */) {
    __shared__ uint32_t s_k[5];
    if (threadIdx.x < 5) s_k[threadIdx.x] = threadIdx.x < 2 ? 0xffffffffu : 0u;
    __syncthreads();
    const uint32_t blocks_per_seg = PUPIL_RAYS / 256u, seg = blockIdx.x / blocks_per_seg, i = (blockIdx.x % blocks_per_seg) * 256u + threadIdx.x;
    const float rear_radius = lens[2] * 1.5f, rear_z = lens[3];
    const float r0 = (float) seg * film_diagonal / (float) PUPIL_SEGMENTS, r1 = (float) (seg + 1u) * film_diagonal / (float) PUPIL_SEGMENTS;
    const float tf = ((float) i + 0.5f) / (float) PUPIL_RAYS;
    v3 p_film = V3(fmaf(r1, tf, fmaf(-r0, tf, r0)), 0.f, 0.f);
    const float u = radical_inverse(2u, i), v = radical_inverse(3u, i);
    v3 p_rear = V3(fmaf(rear_radius, u, fmaf(rear_radius, u, -rear_radius)), fmaf(rear_radius, v, fmaf(rear_radius, v, -rear_radius)), rear_z);
    v3 o = p_film, d = normalize3(p_rear - p_film);
    bool active = true;
    for (uint32_t k = 0; k < lens_count && active; ++k) active = lens_interact(lens + MI_LENS_ROW * k, o, d);
    if (active) {
        atomicMin(&s_k[0], float_key(p_rear.x)); atomicMin(&s_k[1], float_key(p_rear.y));
        atomicMax(&s_k[2], float_key(p_rear.x)); atomicMax(&s_k[3], float_key(p_rear.y));
        atomicAdd(&s_k[4], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 5 && s_k[4] != 0u) {
        uint32_t *g = keys + 5u * seg + threadIdx.x;
        if (threadIdx.x < 2) atomicMin(g, s_k[threadIdx.x]); else if (threadIdx.x < 4) atomicMax(g, s_k[threadIdx.x]); else atomicAdd(g, s_k[4]);
    }
}

// Particle tracer (ptracer.cpp under AdjointIntegrator::render): one lane per sampler stream (a.passes light samples each); a.n_local
// lanes of the shard, global lane id = pixel * (spp / passes) + s over the FILM grid.  A connection's normalised splat is appended to
// the record queue (`queue.cap` != 0; film_put_deferred) and resolved by the k_splat_* kernels below, or -- queue full, or
// MI355_FLAG_DIRECT_SPLAT -- added to the film with plain float atomics.
// GRAD (render_backward): the same lanes, but every connection GATHERS dL/dimage at its splat taps.  Radiance gradients need one walk;
// with `all_params` every sample is walked twice from the same sampler state (see PtGradSink) for the albedo / texture gradients.
// The small-parameter accumulators live in dynamic LDS: (n_small << copy_shift) floats.
#ifndef MI_PT_WAVES_FULL
#define MI_PT_WAVES_FULL 4                  // k_ptracer forward, FULL: 128 VGPRs with 16 spilled beat 143 unspilled at 3 waves (lens doublet render 9.88 -> 8.23 ms)
#endif
#ifndef MI_PT_WAVES_BASIC
#define MI_PT_WAVES_BASIC 4
#endif
#ifndef MI_PT_WAVES_GRAD_FULL
#define MI_PT_WAVES_GRAD_FULL 4             // (same trade as the forward kernel: lens doublet colour backward 10.09 -> 8.37 ms)
#endif
#ifndef MI_PT_WAVES_GRAD_BASIC
#define MI_PT_WAVES_GRAD_BASIC 5            // (Cornell box, all parameters: 10.08 -> 9.54 ms)
#endif
template <int FULL, bool GRAD>
__global__ __launch_bounds__(BLOCK, FULL ? (GRAD ? MI_PT_WAVES_GRAD_FULL : MI_PT_WAVES_FULL) : (GRAD ? MI_PT_WAVES_GRAD_BASIC : MI_PT_WAVES_BASIC)) void k_ptracer(DScene sc, RenderArgs a, float sample_scale, float *film, const float *grad_rgb, float *d_grad,
                                                                 const uint32_t *small_map, unsigned long long *g_counters, SplatQueue queue,
                                                                 uint32_t n_small, uint32_t copy_shift, uint32_t all_params, const uint32_t *order) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[STACK_FLOATS(FULL)];
    __shared__ uint32_t s_cnt[4];
    extern __shared__ float s_grad_dyn[];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    if (GRAD) for (uint32_t k = threadIdx.x; k < (n_small << copy_shift); k += BLOCK) s_grad_dyn[k] = 0.f;
    const PtGradSink sink = { grad_rgb, s_grad_dyn, copy_shift, d_grad, small_map, all_params != 0u };
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    const uint32_t i = (order ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x) * BLOCK + threadIdx.x;     // sorted samples: one contiguous eighth per XCD
    const bool on = i < a.n_local;
    const uint32_t ii = on ? (order ? order[i] : i) : 0u;                  // which light sample this hardware lane walks (pt_sort_lanes)
    const uint32_t lane = shard_pixel(a, ii / a.s_count) * a.spp_total + a.s_begin + ii % a.s_count;
    Pcg32 rng; sampler_seed(rng, a.seed, lane);
    Counters3 cnt = {};
    const bool hide = (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0;
    for (uint32_t pass = 0; pass < a.passes; ++pass) {                      // one stream, `passes` samples (integrator.cpp:719-727)
        if (GRAD && all_params) {
            v3 S = V3(0.f, 0.f, 0.f); Pcg32 replay = rng;
            ptracer_sample<FULL, GRAD>(sc, rng, (int32_t) a.max_depth, (int32_t) a.rr_depth, hide, on, sample_scale, film, stk, cnt, &sink, nullptr, 1, &S);
            ptracer_sample<FULL, GRAD>(sc, replay, (int32_t) a.max_depth, (int32_t) a.rr_depth, hide, on, sample_scale, film, stk, cnt, &sink, nullptr, 2, &S);
        } else
            ptracer_sample<FULL, GRAD>(sc, rng, (int32_t) a.max_depth, (int32_t) a.rr_depth, hide, on, sample_scale, film, stk, cnt, &sink, GRAD ? nullptr : &queue);   // (a compile-time choice: a run-time select of the pointer sends the by-value kernel argument to scratch and its pointers to the flat address space)
    }
    __syncthreads();
    if (GRAD) {
        for (uint32_t k = threadIdx.x; k < n_small; k += BLOCK) {
            float v = 0.f;
            for (uint32_t c = 0; c < (1u << copy_shift); ++c) v += s_grad_dyn[(k << copy_shift) + c];
            if (v != 0.f) atomicAdd(d_grad + small_map[k], v);
        }
    }
    flush_counters(s_cnt, cnt, g_counters);
}

// Forward-mode derivative of the light-traced film with respect to the vertex positions of one mesh (mi355_geo_forward.h): the lanes
// and sampler streams of k_ptracer, dual numbers alongside; the derivative splats are deferred like the primal ones (40-byte records,
// k_splat_tiles<true>) unless MI355_FLAG_DIRECT_SPLAT asks for plain float atomics.
#ifndef MI_GEO_FWD_WAVES
#define MI_GEO_FWD_WAVES 3                 // (forward design mode 12.1 -> 11.0 ms)
#endif
template <int FULL>
__global__ __launch_bounds__(BLOCK, MI_GEO_FWD_WAVES) void k_ptracer_geo(DScene sc, RenderArgs a, float sample_scale, float *gfilm, GeoTangent gt, SplatQueue queue, const uint32_t *order) {
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[STACK_FLOATS(FULL)];
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    const uint32_t i = (order ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x) * BLOCK + threadIdx.x;
    const bool on = i < a.n_local;
    const uint32_t ii = on ? (order ? order[i] : i) : 0u;
    const uint32_t lane = shard_pixel(a, ii / a.s_count) * a.spp_total + a.s_begin + ii % a.s_count;
    Pcg32 rng; sampler_seed(rng, a.seed, lane);
    for (uint32_t pass = 0; pass < a.passes; ++pass)
        ptracer_sample_geo<FULL>(sc, gt, rng, (int32_t) a.max_depth, (int32_t) a.rr_depth, (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0, on, sample_scale, gfilm, stk, &queue);
}

// Several cameras, one light-traced pass (mi355_render_accumulate_sensors; DSensorSet in mi355_device.h).  Forward only.
template <int FULL>
__global__ __launch_bounds__(BLOCK, FULL ? MI_PT_WAVES_FULL : MI_PT_WAVES_BASIC) void k_ptracer_multi(DScene sc, RenderArgs a, float sample_scale, float *film, unsigned long long *g_counters,
                                                                                                     SplatQueue queue, const uint32_t *order, DSensorSet set) {
    MI_LT_FILL(FULL, sc);
    __shared__ __attribute__((aligned(16))) float smem[STACK_FLOATS(FULL)];
    __shared__ uint32_t s_cnt[4];
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    const uint32_t i = (order ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x) * BLOCK + threadIdx.x;
    const bool on = i < a.n_local;
    const uint32_t ii = on ? (order ? order[i] : i) : 0u;
    const uint32_t lane = shard_pixel(a, ii / a.s_count) * a.spp_total + a.s_begin + ii % a.s_count;
    Pcg32 rng; sampler_seed(rng, a.seed, lane);
    Counters3 cnt = {};
    ptracer_sample<FULL, false, true>(sc, rng, (int32_t) a.max_depth, (int32_t) a.rr_depth, (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0, on, sample_scale, film, stk, cnt, nullptr, &queue, 1, nullptr, &set);
    flush_counters(s_cnt, cnt, g_counters);
}

// ---- coherent lanes for the light tracer (pt_sort_lanes) ---------------------------------------------------------------------------
// Light samples are independent streams, so the 64 lanes of a wave start 64 unrelated rays and walk 64 unrelated subtrees of the scene
// BVH (lens doublet: 54 % of the VALU lanes busy, 10-19 GB fetched per launch, profiles/r02_pmc_lens_summary.txt).  The ASSIGNMENT of
// samples to hardware lanes is free, though: k_pt_keys gives every sample the cell of the emitter's sample domain its ray leaves from,
// a counting sort over those cells (k_pt_keys histogram, k_pt_scan, k_pt_scatter) lists the samples cell by cell, and the tracing
// kernels take their sample from that list -- the same samples, the same film and gradients up to the order of the float atomics, but a
// wave now walks one neighbourhood of the BVH.
template <int FULL>
__global__ __launch_bounds__(256) void k_pt_keys(DScene sc, RenderArgs a, uint32_t bits, uint32_t four_d, uint32_t *keys, uint32_t *hist) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n_local) return;
    const uint32_t lane = shard_pixel(a, i / a.s_count) * a.spp_total + a.s_begin + i % a.s_count;
    Pcg32 rng; sampler_seed(rng, a.seed, lane);
    const uint32_t key = ptracer_emission_cell<FULL>(sc, rng, (int32_t) a.max_depth, (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0, bits, four_d != 0u);
    keys[i] = key; atomicAdd(hist + key, 1u);
}
// exclusive prefix sum of `n` bin counts (n <= 64 Ki) in one workgroup; cursor = a second copy of the offsets for the scatter pass
__global__ __launch_bounds__(1024) void k_pt_scan(const uint32_t *hist, uint32_t *cursor, uint32_t n) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u, b = threadIdx.x * per, e = min(b + per, n);
    uint32_t sum = 0; for (uint32_t k = b; k < e; ++k) sum += hist[k];
    part[threadIdx.x] = sum; __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) { const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u; __syncthreads(); part[threadIdx.x] += v; __syncthreads(); }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t k = b; k < e; ++k) { const uint32_t c = hist[k]; cursor[k] = run; run += c; }
}
__global__ __launch_bounds__(256) void k_pt_scatter(const uint32_t *keys, uint32_t *cursor, uint32_t *order, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    order[atomicAdd(cursor + keys[i], 1u)] = i;
}

// ---- deferred splat, steps 2-5 (step 1 = film_put_deferred inside k_ptracer) ------------------------------------------------------
#define SPLAT_CHUNK 8192u                                                   // records per workgroup in the histogram pass
#define SPLAT_RUN 4096u                                                     // records per workgroup in the tile pass
#define SPLAT_MAX_TILES 8000u                                               // LDS of the scatter pass: 2 x 4 B per tile + 80 - 96 KB of records <= 159 KB (films up to ~1430^2; larger ones splat directly)
// Tile histogram of the record queue; one LDS histogram per chunk, flushed with one global atomic per non-empty bin.
__global__ __launch_bounds__(256) void k_splat_hist(DScene sc, SplatQueue q, uint32_t n_tiles) {
    extern __shared__ uint32_t s_bins[];
    const uint32_t cps = (q.seg_cap + SPLAT_CHUNK - 1u) / SPLAT_CHUNK;      // chunks per segment
    for (uint32_t w = blockIdx.x; w < MI_SPLAT_SEGS * cps; w += gridDim.x) {
        const uint32_t seg = w / cps, c0 = seg * q.seg_cap + (w % cps) * SPLAT_CHUNK;
        const uint32_t seg_end = seg * q.seg_cap + min(q.count[seg * MI_SPLAT_COUNTER_STRIDE], q.seg_cap);
        if (c0 >= seg_end) continue;                                        // (uniform over the workgroup)
        for (uint32_t k = threadIdx.x; k < n_tiles; k += 256u) s_bins[k] = 0u;
        __syncthreads();
        const uint32_t c1 = min(c0 + SPLAT_CHUNK, seg_end);
        for (uint32_t i = c0 + threadIdx.x; i < c1; i += 1024u) {            // (four loads in flight per thread: a wave that issues them one at a time spends its life waiting for HBM)
            float4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = i + 256u * u < c1 ? q.rec_a[i + 256u * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i + 256u * u < c1) atomicAdd(s_bins + splat_tile_of(sc, q, a[u].x, a[u].y), 1u);
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n_tiles; k += 256u) { const uint32_t v = s_bins[k]; if (v) atomicAdd(q.hist + k, v); }
        __syncthreads();
    }
}
// Exclusive scan of the tile histogram (one workgroup); clears the cursors; offsets[n_tiles] = number of queued records.
__global__ __launch_bounds__(1024) void k_splat_scan(const uint32_t *hist, uint32_t *offsets, uint32_t *cursor, uint32_t n_tiles) {
    __shared__ uint32_t s_part[1024]; __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_tiles ? hist[i] : 0u;
        s_part[threadIdx.x] = v; __syncthreads();
        for (uint32_t d = 1; d < 1024u; d <<= 1) {                          // Hillis-Steele inclusive scan
            uint32_t t = threadIdx.x >= d ? s_part[threadIdx.x - d] : 0u; __syncthreads();
            s_part[threadIdx.x] += t; __syncthreads();
        }
        const uint32_t incl = s_part[threadIdx.x], carry = s_carry;
        if (i < n_tiles) { offsets[i] = carry + incl - v; cursor[i] = 0u; }
        __syncthreads();
        if (threadIdx.x == 1023u) s_carry = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n_tiles] = s_carry;
}
// Counting sort, scatter step.  A workgroup takes SPLAT_SORT records, SPLAT_SORT_R per thread, and reads them ONCE, into registers: the count
// atomic of a record's tile in LDS returns the record's rank inside that tile; a scan of the counts gives every tile its start inside the
// chunk, one global cursor atomic per non-empty tile reserves the chunk's range in that tile; the records go to LDS at start + rank, i.e.
// sorted, and leave it in that order, so that consecutive lanes store to consecutive slots of a tile's reservation.
// History (profiles/r03_pmc_ptracer_summary.txt): storing each record straight to its slot scattered 16 + 4 bytes per lane -- WRITE_SIZE 3.3 GB
// for 1.3 GB of records, 2.3 ms per 57 M records; sorting a permutation of an 8192-record chunk in LDS and reading the chunk three times from
// global memory cured the writes but not the reads: 768 resident chunks of 160 KB are 120 MB against 8 x 4 MB of L2, FETCH_SIZE 3.5 GB, 1.3 ms.
// Chunk size, light tracer on the Cornell box 512^2 x 64 spp, box / Gaussian filter, ms per render (profiles/r03_splat_scatter_ab.txt): 1024 records
// 7.70 / 9.11, 2048 4.70 / 5.92, 4096 4.18 / 5.45 (as 1024 threads x 4 records; 512 x 8: 4.29 / 5.47); the three-pass 8192-record version 5.12 / 6.21.
// Every chunk costs one returning global atomic per tile it touches, all chunks on the same n_tiles cursor words: the fewer chunks the better, and
// 4096 records x 20 B is what fits the LDS next to the tile tables.
#ifndef SPLAT_SORT_T
#define SPLAT_SORT_T 1024u
#define SPLAT_SORT_R 4u
#endif
#define SPLAT_SORT (SPLAT_SORT_T * SPLAT_SORT_R)
__global__ __launch_bounds__(SPLAT_SORT_T) void k_splat_scatter(DScene sc, SplatQueue q, const uint32_t *offsets, uint32_t *cursor, float4 *out_a, float *out_b,
                                                                float4 *out_c, float *out_d, uint32_t n_tiles) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_sort[];       // [n_tiles] counts -> global bases; [n_tiles] starts inside the chunk; the sorted records: float4[SORT], float[SORT] (, source index[SORT] for the second record half)
    uint32_t *s_bins = s_sort, *s_start = s_sort + n_tiles;
    float4 *s_a = reinterpret_cast<float4 *>(s_sort + ((2u * n_tiles + 3u) & ~3u));
    float *s_b = reinterpret_cast<float *>(s_a + SPLAT_SORT);
    uint32_t *s_src = reinterpret_cast<uint32_t *>(s_b + SPLAT_SORT);
    __shared__ uint32_t s_wave[SPLAT_SORT_T / 64u];
    const uint32_t cps = (q.seg_cap + SPLAT_SORT - 1u) / SPLAT_SORT;
    const uint32_t per = (n_tiles + SPLAT_SORT_T - 1u) / SPLAT_SORT_T;      // tiles per thread in the scan
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t w = blockIdx.x; w < MI_SPLAT_SEGS * cps; w += gridDim.x) {
        const uint32_t seg = w / cps, c0 = seg * q.seg_cap + (w % cps) * SPLAT_SORT;
        const uint32_t seg_end = seg * q.seg_cap + min(q.count[seg * MI_SPLAT_COUNTER_STRIDE], q.seg_cap);
        if (c0 >= seg_end) continue;                                        // (uniform over the workgroup)
        for (uint32_t k = threadIdx.x; k < n_tiles; k += SPLAT_SORT_T) s_bins[k] = 0u;
        __syncthreads();
        const uint32_t n = min(SPLAT_SORT, seg_end - c0);
        float4 a[SPLAT_SORT_R]; float b[SPLAT_SORT_R]; uint32_t tile[SPLAT_SORT_R], rank[SPLAT_SORT_R];
#pragma unroll
        for (uint32_t u = 0; u < SPLAT_SORT_R; ++u) {                       // (all loads of a thread in flight together)
            const uint32_t i = threadIdx.x + SPLAT_SORT_T * u;
            a[u] = i < n ? q.rec_a[c0 + i] : make_float4(0.f, 0.f, 0.f, 0.f); b[u] = i < n ? q.rec_b[c0 + i] : 0.f;
        }
#pragma unroll
        for (uint32_t u = 0; u < SPLAT_SORT_R; ++u) if (threadIdx.x + SPLAT_SORT_T * u < n) {
            tile[u] = splat_tile_of(sc, q, a[u].x, a[u].y); rank[u] = atomicAdd(s_bins + tile[u], 1u);
        }
        __syncthreads();
        // exclusive scan of the counts (thread t owns tiles [t * per, (t + 1) * per)), then the global reservations
        uint32_t sum = 0; const uint32_t kb = min(threadIdx.x * per, n_tiles), ke = min(kb + per, n_tiles);
        for (uint32_t k = kb; k < ke; ++k) sum += s_bins[k];
        uint32_t incl = sum;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t k = 0; k < wave; ++k) run += s_wave[k];
        for (uint32_t k = kb; k < ke; ++k) { const uint32_t v = s_bins[k]; s_start[k] = run; run += v; if (v) s_bins[k] = offsets[k] + atomicAdd(cursor + k, v); }
        __syncthreads();
#pragma unroll
        for (uint32_t u = 0; u < SPLAT_SORT_R; ++u) if (threadIdx.x + SPLAT_SORT_T * u < n) {
            const uint32_t p = s_start[tile[u]] + rank[u];
            s_a[p] = a[u]; s_b[p] = b[u];
            if (out_c) s_src[p] = threadIdx.x + SPLAT_SORT_T * u;
        }
        __syncthreads();
        for (uint32_t pos = threadIdx.x; pos < n; pos += SPLAT_SORT_T) {    // consecutive lanes -> consecutive slots of a tile's reservation
            const float4 ra = s_a[pos];
            const uint32_t t = splat_tile_of(sc, q, ra.x, ra.y), dst = s_bins[t] + (pos - s_start[t]);
            out_a[dst] = ra; out_b[dst] = s_b[pos];
            if (out_c) { const uint32_t src = c0 + s_src[pos]; out_c[dst] = q.rec_c[src]; out_d[dst] = q.rec_d[src]; }
        }
        __syncthreads();
    }
}
// Tile pass: workgroup b owns the sorted records [b * SPLAT_RUN, (b + 1) * SPLAT_RUN) -- equal work whatever the distribution over
// the film.  For every tile that run touches: the tile plus its apron lives in LDS, every record adds its normalised taps there, and
// the tile goes to the film with one float atomic per touched pixel and channel.
// The LDS accumulators are 64-bit FIXED POINT, added with integer atomics: on this chip an LDS fp32 atomic add keeps the LDS busy for
// ~122 cycles per wave instruction (2 cycles per lane; profiles/r01_j_ptracer_pmc_summary.txt) against ~9 for an integer one.  The
// scale is a power of two chosen per run from the largest colour component m of its records: contributions are <= m, a pixel
// receives at most SPLAT_RUN of them, so with m * scale <= 2^50 the sum stays below 2^62, and the quantum is 2^-50 of m -- finer
// than the fp32 additions it replaces.  (Non-finite colours take the float path, so that they reach the film as they are.)
#define MI_SPLAT_TW (MI_SPLAT_TILE + 2 * MI_SPLAT_APRON)
// DUAL (forward-mode derivative film): record = (x, y, dV | V, dx, dy); a tap receives dV w + V dw with w the NORMALISED weight and dw its
// derivative along the film-position tangent (film_put_dual's arithmetic).
template <bool DUAL>
__global__ __launch_bounds__(256) void k_splat_tiles(DScene sc, SplatQueue q, const uint32_t *offsets, const float4 *rec_a, const float *rec_b,
                                                     const float4 *rec_c, const float *rec_d, float *film, uint32_t n_tiles) {
    __shared__ long long s_tile[MI_SPLAT_TW * MI_SPLAT_TW * 3];
    __shared__ uint32_t s_max;
    const uint32_t n = offsets[n_tiles];                                    // records actually queued
    const int W = (int) sc.crop_w, H = (int) sc.crop_h;
    for (uint32_t r0 = blockIdx.x * SPLAT_RUN; r0 < n; r0 += gridDim.x * SPLAT_RUN) {
        const uint32_t r1 = min(r0 + SPLAT_RUN, n);
        uint32_t lo = 0, hi = n_tiles;                                      // last tile whose range starts at or before r0
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) / 2u; if (offsets[mid] <= r0) lo = mid; else hi = mid; }
        uint32_t start = r0;
        for (uint32_t tile = lo; tile < n_tiles && start < r1; ++tile) {
            const uint32_t stop = min(offsets[tile + 1], r1);
            if (stop <= start) continue;                                    // empty tile
            for (uint32_t k = threadIdx.x; k < MI_SPLAT_TW * MI_SPLAT_TW * 3; k += 256u) s_tile[k] = 0ll;
            if (threadIdx.x == 0) s_max = 0u;
            __syncthreads();
            float m = 0.f;
            for (uint32_t j = start + threadIdx.x; j < stop; j += 256u) {
                const float4 a = rec_a[j]; float mj = fmaxf(fabsf(a.z), fmaxf(fabsf(a.w), fabsf(rec_b[j]))), sum = fabsf(a.z) + fabsf(a.w) + fabsf(rec_b[j]);
                if (DUAL) {                                                 // |dV w + V dw| <= |dV| + |V| (|dx| + |dy|) * 8: the normalised weights' slopes stay below 8 per pixel
                    const float4 c = rec_c[j]; const float dd = fabsf(c.w) + fabsf(rec_d[j]);
                    mj += fmaxf(fabsf(c.x), fmaxf(fabsf(c.y), fabsf(c.z))) * dd * 8.f; sum += (fabsf(c.x) + fabsf(c.y) + fabsf(c.z)) * dd;
                }
                m = fmaxf(m, mj); if (!(sum < MI_INF)) m = MI_INF;
            }
            atomicMax(&s_max, __float_as_uint(m));                          // (m >= 0: the bit patterns order like the values)
            __syncthreads();
            m = __uint_as_float(s_max);
            const bool fixed = m < MI_INF;                                  // uniform over the workgroup
            int e = 0; (void) frexpf(m, &e);                                // m < 2^e
            const int se = min(50 - e, 120);
            const float scale = ldexpf(1.f, se), inv_scale = ldexpf(1.f, -se);
            const int32_t ox = (int32_t) (tile % q.tiles_x) * MI_SPLAT_TILE - MI_SPLAT_APRON, oy = (int32_t) (tile / q.tiles_x) * MI_SPLAT_TILE - MI_SPLAT_APRON;
            if (DUAL) {
                if (m != 0.f) for (uint32_t j = start + threadIdx.x; j < stop; j += 256u) {
                    const float4 a = rec_a[j], c = rec_c[j]; const float b = rec_b[j], dy = rec_d[j];
                    const Df px = DF(a.x, c.w), py = DF(a.y, dy);
                    const bool box = sc.rfilter == MI355_RFILTER_BOX;
                    const int n = box ? 0 : (int) ceilf(sc.filter_radius - .5f), count = 2 * n + 1;
                    const int32_t pix = (int32_t) floorf(px.v) - n, piy = (int32_t) floorf(py.v) - n;
                    Df wx[MI_MAX_TAPS], wy[MI_MAX_TAPS], sx = DF(0.f), sy = DF(0.f);
#pragma unroll
                    for (int i = 0; i < MI_MAX_TAPS; ++i) {
                        wx[i] = wy[i] = DF(0.f);
                        if (i < count) {
                            wx[i] = box ? DF(1.f) : dfilter_eval(sc, DF((float) (pix + i) + .5f) - px);
                            wy[i] = box ? DF(1.f) : dfilter_eval(sc, DF((float) (piy + i) + .5f) - py);
                            sx = sx + wx[i]; sy = sy + wy[i];
                        }
                    }
                    const Df norm = sx * sy;
                    if (norm.v == 0.f) continue;
                    const Df factor = DF(1.f) / norm;
#pragma unroll
                    for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
                        for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
                            const int32_t x = pix + tx - (int32_t) sc.crop_x, y = piy + ty - (int32_t) sc.crop_y;
                            if (ty >= count || tx >= count || x < 0 || y < 0 || x >= W || y >= H) continue;
                            const Df w = wy[ty] * wx[tx] * factor;
                            const float r = fmaf(a.z, w.v, c.x * w.d), g = fmaf(a.w, w.v, c.y * w.d), bl = fmaf(b, w.v, c.z * w.d);
                            const int32_t lx = x - ox, ly = y - oy;
                            if (lx < 0 || ly < 0 || lx >= MI_SPLAT_TW || ly >= MI_SPLAT_TW) continue;            // (inside by construction)
                            if (fixed) {
                                unsigned long long *px_ = reinterpret_cast<unsigned long long *>(s_tile) + (ly * MI_SPLAT_TW + lx) * 3;
                                if (r != 0.f) atomicAdd(px_, (unsigned long long) __float2ll_rn(r * scale));
                                if (g != 0.f) atomicAdd(px_ + 1, (unsigned long long) __float2ll_rn(g * scale));
                                if (bl != 0.f) atomicAdd(px_ + 2, (unsigned long long) __float2ll_rn(bl * scale));
                            } else {
                                float *px_ = film + ((size_t) y * W + x) * 4;
                                atomicAdd(px_, r); atomicAdd(px_ + 1, g); atomicAdd(px_ + 2, bl);
                            }
                        }
                }
            } else
            if (m != 0.f) for (uint32_t j = start + threadIdx.x; j < stop; j += 256u) {
                const float4 a = rec_a[j]; const float b = rec_b[j];
                Splat sp = splat_weights(sc, a.x, a.y);
                float sx = 0.f, sy = 0.f;
#pragma unroll
                for (int i = 0; i < MI_MAX_TAPS; ++i) { sx += i < sp.count ? sp.wx[i] : 0.f; sy += i < sp.count ? sp.wy[i] : 0.f; }
                float factor = sx * sy; factor = factor != 0.f ? 1.f / factor : 0.f;
#pragma unroll
                for (int ty = 0; ty < MI_MAX_TAPS; ++ty)
#pragma unroll
                    for (int tx = 0; tx < MI_MAX_TAPS; ++tx) {
                        const int32_t x = sp.x0 + tx, y = sp.y0 + ty;
                        if (ty >= sp.count || tx >= sp.count || x < 0 || y < 0 || x >= W || y >= H) continue;
                        const float w = sp.wy[ty] * (sp.wx[tx] * factor);
                        const int32_t lx = x - ox, ly = y - oy;
                        if (w == 0.f || lx < 0 || ly < 0 || lx >= MI_SPLAT_TW || ly >= MI_SPLAT_TW) continue;   // (inside by construction)
                        if (fixed) {
                            unsigned long long *px = reinterpret_cast<unsigned long long *>(s_tile) + (ly * MI_SPLAT_TW + lx) * 3;
                            atomicAdd(px, (unsigned long long) __float2ll_rn((a.z * w) * scale));
                            atomicAdd(px + 1, (unsigned long long) __float2ll_rn((a.w * w) * scale));
                            atomicAdd(px + 2, (unsigned long long) __float2ll_rn((b * w) * scale));
                        } else {                                            // non-finite colours: straight to the film
                            float *px = film + ((size_t) y * W + x) * 4;
                            atomicAdd(px, a.z * w); atomicAdd(px + 1, a.w * w); atomicAdd(px + 2, b * w);
                        }
                    }
            }
            __syncthreads();
            if (fixed && m != 0.f) for (uint32_t k = threadIdx.x; k < MI_SPLAT_TW * MI_SPLAT_TW; k += 256u) {
                const int32_t x = ox + (int32_t) (k % MI_SPLAT_TW), y = oy + (int32_t) (k / MI_SPLAT_TW);
                if (x < 0 || y < 0 || x >= W || y >= H) continue;
                const long long r = s_tile[3 * k], g = s_tile[3 * k + 1], bl = s_tile[3 * k + 2];
                float *px = film + ((size_t) y * W + x) * 4;
                if (r != 0ll) atomicAdd(px, (float) r * inv_scale);
                if (g != 0ll) atomicAdd(px + 1, (float) g * inv_scale);
                if (bl != 0ll) atomicAdd(px + 2, (float) bl * inv_scale);
            }
            __syncthreads();
            start = stop;
        }
    }
}

// ---- geometry / IOR adjoint of the particle tracer (mi355_geo_backward.h) -----------------------------------------------------------
// Lanes [lane_first, lane_first + n_chunk) of the shard: forward walk onto the HBM path tape, backward sweep into the gradient buffer.
// Dynamic LDS: 2 floats per bsdf for the IOR adjoints (n_ior_lds = 2 * n_bsdfs, or 0 -> global atomics).
#ifndef MI_GEO_BWD_WAVES
#define MI_GEO_BWD_WAVES 3                 // 168 VGPRs + 116 B of scratch at 3 waves beat 219 unspilled at 2 (lens doublet 9.15 -> 7.96 ms)
#endif
template <int FULL, bool MULTI = false>
__global__ __launch_bounds__(BLOCK, MI_GEO_BWD_WAVES) void k_ptracer_geo_bwd(DScene sc, RenderArgs a, float sample_scale, const float *grad_rgb, float *d_grad, float *d_nrm_adj,
                                                               GeoTape tape, uint32_t lane_first, uint32_t n_chunk, uint32_t n_ior_lds, const uint32_t *order,
                                                               DSensorSet set, GeoGradSet gset) {      // (set, gset: read when MULTI -- several cameras, one pass)
    MI_LT_FILL(FULL, sc);                                                  // BASIC variants: record tables staged in LDS
    __shared__ __attribute__((aligned(16))) float smem[STACK_FLOATS(FULL)];
    extern __shared__ float s_ior_dyn[];
    for (uint32_t k = threadIdx.x; k < n_ior_lds; k += BLOCK) s_ior_dyn[k] = 0.f;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane_id = threadIdx.x & 63;
    uint32_t *stk = STK_PTR(smem, FULL, wave, lane_id);
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;                    // slot on the tape
    const uint32_t tl = (order ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x) * BLOCK + threadIdx.x;    // sorted samples: one contiguous eighth of the chunk per XCD
    const uint32_t i = lane_first + tl;
    const bool on = tl < n_chunk && i < a.n_local;
    const uint32_t ii = on ? (order ? order[i] : i) : 0u;
    const uint32_t lane = shard_pixel(a, ii / a.s_count) * a.spp_total + a.s_begin + ii % a.s_count;
    Pcg32 rng; sampler_seed(rng, a.seed, lane);
    // the vertex-adjoint table of the sweep lives in the BVH stacks of the walk (FULL == 1: 4 x 8 KB): barriers keep the two apart
    static_assert(FULL != 1 || STACK_FLOATS(1) >= 8 * (int) MI_VTAB, "the vertex-adjoint table must fit the traversal stacks");
    const GeoBwdSink sink = { grad_rgb, d_grad, d_nrm_adj, n_ior_lds ? s_ior_dyn : nullptr, FULL == 1 ? reinterpret_cast<uint32_t *>(smem) : nullptr };
    const uint32_t slot = min(t, tape.stride - 1u);
    for (uint32_t pass = 0; pass < a.passes; ++pass) {
        uint32_t n_sweep;
        const uint32_t nv = ptracer_geo_walk<FULL, MULTI>(sc, rng, (int32_t) a.max_depth, (int32_t) a.rr_depth, (a.flags & MI355_FLAG_HIDE_EMITTERS) != 0, on, stk, tape, slot, n_sweep, &set);
#ifndef MI_GEO_NO_SWEEP
        if (FULL == 1) {
            // (the barrier that separates the walk's traversal stacks from the table also tells whether ANY path of the workgroup has something to
            //  sweep: sorted by emitter, whole workgroups carry the light of a field source this sensor does not see)
            if (__syncthreads_or(n_sweep != 0u)) {
                vertex_adj_clear(sink.tab); __syncthreads();
                ptracer_geo_sweep<FULL, MULTI>(sc, sample_scale, tape, slot, nv, n_sweep, sink, &set, &gset);
                __syncthreads(); vertex_adj_flush(sink);
            }
            __syncthreads();
        } else ptracer_geo_sweep<FULL, MULTI>(sc, sample_scale, tape, slot, nv, n_sweep, sink, &set, &gset);
#endif
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_ior_lds; k += BLOCK) {
        const float v = s_ior_dyn[k];
        if (v != 0.f) { const int32_t off = sc.bsdfs[k >> 1].ior_grad_offset; if (off >= 0) atomicAdd(d_grad + off + (k & 1u), v); }
    }
}
// Mesh::recompute_vertex_normals for one mesh: one thread per vertex (gather over its incident faces).  With `vtan` the kernel writes the
// TANGENT of the normals instead (forward design mode) and leaves the normals alone.
__global__ __launch_bounds__(256) void k_vertex_normals(const float *positions, const uint32_t *faces, const uint32_t *adj_off, const uint32_t *adj, uint32_t vtx_offset,
                                                        uint32_t tri_offset, uint32_t n_vertices, float *normals, float *nrm_len, const float *vtan, float *ntan) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n_vertices) return;
    v3 n, dn; float len;
    vertex_normal_gather(positions, faces, adj_off, adj, vtx_offset, tri_offset, vtx_offset + v, n, len, vtan, vtan ? &dn : nullptr);
    if (vtan) { ntan[3 * (size_t) v] = dn.x; ntan[3 * (size_t) v + 1] = dn.y; ntan[3 * (size_t) v + 2] = dn.z; return; }
    float *o = normals + 3 * (size_t) (vtx_offset + v);
    o[0] = n.x; o[1] = n.y; o[2] = n.z; nrm_len[vtx_offset + v] = len;
}
// Adjoint of the same computation as a per-vertex GATHER over the incident (face, corner) pairs, like the forward pass: the adjoints of the
// (normalised) vertex normals of a face's corners go through the normalisation (stored lengths) and through face_normal_terms, and the vertex
// keeps its own corner's share.  Every face is evaluated once per corner (3 x the arithmetic of a per-face scatter), but nothing is scattered:
// the per-face version issued 9 memory-side float atomics per face and took 80 us per 166 k-face lens surface.  EIGHT lanes share a vertex --
// each takes every eighth incident face, the shares are summed with three shuffles -- because one thread per vertex left 1.3 waves per SIMD
// to cover three dependent loads per face (62 us); grad_pos is updated with plain loads and stores: no other kernel writes the mesh's slice
// while this one runs (same stream, after the sweep).
#define MI_VNB_LANES 8u
__global__ __launch_bounds__(256) void k_vertex_normals_bwd(const float *positions, const float *normals, const float *nrm_len, const uint32_t *faces, const uint32_t *adj_off,
                                                            const uint32_t *adj, uint32_t vtx_offset, uint32_t tri_offset, uint32_t n_vertices, const float *nrm_adj, float *grad_pos) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, v = t / MI_VNB_LANES, sub = t % MI_VNB_LANES;
    const bool valid = v < n_vertices;
    gv3 acc = ga::mk<float>(0.f, 0.f, 0.f); int touched = 0;
    if (valid) for (uint32_t e = adj_off[vtx_offset + v] + sub; e < adj_off[vtx_offset + v + 1u]; e += MI_VNB_LANES) {
        const uint32_t f = adj[e] >> 2, corner = adj[e] & 3u;
        const uint32_t *fi = faces + 3 * (size_t) (tri_offset + f);
        gv3 p[3], aN[3]; bool any = false;
        for (int c = 0; c < 3; ++c) {
            const size_t g = vtx_offset + fi[c];
            p[c] = G3(ld3(positions + 3 * g));
            const gv3 an = G3(ld3(nrm_adj + 3 * g)), nn = G3(ld3(normals + 3 * g)); const float len = nrm_len[g];
            aN[c] = len != 0.f ? ga::normalize_adj(nn, len, an) : ga::mk<float>(0.f, 0.f, 0.f);
            any = any || aN[c].x != 0.f || aN[c].y != 0.f || aN[c].z != 0.f;
        }
        if (!any) continue;
        gv3 av[3]; ga::face_normal_terms_adj(p[0], p[1], p[2], aN, av);
        const gv3 mine = corner == 0u ? av[0] : (corner == 1u ? av[1] : av[2]);
        if (!(mine.x == mine.x && mine.y == mine.y && mine.z == mine.z)) continue;       // degenerate face
        acc = acc + mine; touched = 1;
    }
#pragma unroll
    for (int off = (int) MI_VNB_LANES / 2; off > 0; off >>= 1) {            // (all 256 threads take part: the groups are aligned runs of 8 lanes of a wave)
        acc.x += __shfl_down(acc.x, off, (int) MI_VNB_LANES); acc.y += __shfl_down(acc.y, off, (int) MI_VNB_LANES); acc.z += __shfl_down(acc.z, off, (int) MI_VNB_LANES);
        touched |= __shfl_down(touched, off, (int) MI_VNB_LANES);
    }
    if (!valid || sub != 0u || !touched) return;
    float *g = grad_pos + 3 * (size_t) v;
    g[0] += acc.x; g[1] += acc.y; g[2] += acc.z;
}
// Scene::parameters_changed for the meshes in the scene BVH: every leaf-ordered triangle from the (new) positions ...
__global__ __launch_bounds__(256) void k_mesh_tris(const float *positions, const uint32_t *faces, const uint32_t *tri_face, const uint32_t *tri_shape, const DShape *shapes,
                                                   uint32_t n, DTri *tris) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const DShape &sh = shapes[tri_shape[k]];
    const uint32_t *fi = faces + 3 * (size_t) (sh.tri_offset + tri_face[k]);
    const v3 a = ld3(positions + 3 * (size_t) (sh.vtx_offset + fi[0])), b = ld3(positions + 3 * (size_t) (sh.vtx_offset + fi[1])), c = ld3(positions + 3 * (size_t) (sh.vtx_offset + fi[2]));
    DTri t; t.p0[0] = a.x; t.p0[1] = a.y; t.p0[2] = a.z; t.e1[0] = b.x - a.x; t.e1[1] = b.y - a.y; t.e1[2] = b.z - a.z; t.e2[0] = c.x - a.x; t.e2[1] = c.y - a.y; t.e2[2] = c.z - a.z;
    tris[k] = t;
}
// ... and the child boxes refitted bottom-up, one launch per tree level (deepest first); topology and leaf contents are kept.
__global__ __launch_bounds__(256) void k_bvh_refit_level(DBvh4 *nodes, DBvhQ *qnodes, const uint32_t *level_nodes, uint32_t n, const DTri *tris) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t node = level_nodes[i];
    DBvh4 nd = nodes[node];
    for (int c = 0; c < 4; ++c) {
        const uint32_t e = nd.entry[c];
        float lo[3] = { MI_INF, MI_INF, MI_INF }, hi[3] = { -MI_INF, -MI_INF, -MI_INF };
        if (e == MI_BVH_EMPTY) { for (int a = 0; a < 3; ++a) lo[a] = hi[a] = 0.f; }
        else if (e & MI_BVH_LEAF) {
            const uint32_t first = e & 0x1fffffffu, count = ((e >> 29) & 3u) + 1u;
            for (uint32_t k = 0; k < count; ++k) {
                const DTri t = tris[first + k];
                for (int a = 0; a < 3; ++a) {
                    const float p = t.p0[a], q = t.p0[a] + t.e1[a], r = t.p0[a] + t.e2[a];
                    lo[a] = fminf(lo[a], fminf(p, fminf(q, r))); hi[a] = fmaxf(hi[a], fmaxf(p, fmaxf(q, r)));
                }
            }
        } else {
            const DBvh4 ch = nodes[e];                                      // (a deeper level: refitted by an earlier launch)
            for (int k = 0; k < 4; ++k) if (ch.entry[k] != MI_BVH_EMPTY) for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], ch.lo[a][k]); hi[a] = fmaxf(hi[a], ch.hi[a][k]); }
        }
        for (int a = 0; a < 3; ++a) { nd.lo[a][c] = lo[a]; nd.hi[a][c] = hi[a]; }
    }
    nodes[node] = nd;
    DBvhQ q; bvhq_encode(nd, q); qnodes[node] = q;                         // what the traversal reads
}

// ---- the lens-design loop's surface update on the device ---------------------------------------------------------------------------
// EvenAsphericSurface.compute_z_dr of the fork's scripts (notebooks/scripts/lens_design/lens.py:166-249; a ConicSurface, :55-97, is the
// same with zero polynomial terms) in its dimensionless form: with r2 = (x^2 + y^2) / R^2,
//     z = z0 - R r2 ( r2 (a4 + a6 r2 + ... + a16 r2^6) + c / (1 + sqrt(max(1 - (1 + K) c^2 r2, 0))) )
// evaluated for every vertex of a surface mesh (LensElement.update, lens.py:509-568: new position = (x0, y0, z(x0, y0))) -- and its
// transpose: d loss / d (c, K, z0, a4 .. a16) from d loss / d positions.  In the reference both are Dr.Jit arithmetic traced into the
// render's AD graph; here they were ~300 tiny torch launches per design iteration (profiles/r03_iter_profile.txt).
// p = (c, K, z0, a4, a6, a8, a10, a12, a14, a16), 10 floats of device memory.
struct SagPoint { float z, d[10]; };
MI_DEV SagPoint sag_eval(const float *p, float R, float x, float y, bool want_grad) {
    const float c = p[0], K = p[1], z0 = p[2], r2 = (x * x + y * y) * (1.f / (R * R));
    float poly = p[9];
#pragma unroll
    for (int k = 8; k >= 3; --k) poly = fmaf(poly, r2, p[k]);
    const float safe = fmaxf(1.f - (1.f + K) * c * c * r2, 0.f), sq = sqrtf(safe), den = 1.f / (1.f + sq);
    const float zz = fmaf(poly, r2, c * den);
    SagPoint o; o.z = fmaf(zz, -R * r2, z0);
    if (want_grad) {
        const float g = -R * r2, ds = safe > 0.f ? 1.f / sq : 0.f;          // d sqrt / d safe * 2 (0 where the clamp is active)
        o.d[0] = g * (den + c * c * (1.f + K) * r2 * ds * den * den);       // d/dc [c / (1 + s)],  ds/dc = -(1 + K) c r2 / s
        o.d[1] = g * (0.5f * c * c * c * r2 * ds * den * den);              // ds/dK = -c^2 r2 / (2 s)
        o.d[2] = 1.f;
        float pw = g * r2;                                                  // d/da_(4 + 2j) = -R r2 * r2^(j + 1)
#pragma unroll
        for (int j = 0; j < 7; ++j) { o.d[3 + j] = pw; pw *= r2; }
    }
    return o;
}
// out[i] = (base[i].x, base[i].y, z(eval[i])): a lens mesh evaluates at its own (x, y); a baffle at the rim point (R, 0) for every vertex
__global__ __launch_bounds__(256) void k_lens_sag(const float *eval_xy, const float *base, uint32_t n, float R, const float *p, float *out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float x = eval_xy ? eval_xy[2 * (size_t) i] : base[3 * (size_t) i], y = eval_xy ? eval_xy[2 * (size_t) i + 1] : base[3 * (size_t) i + 1];
    out[3 * (size_t) i] = base[3 * (size_t) i]; out[3 * (size_t) i + 1] = base[3 * (size_t) i + 1]; out[3 * (size_t) i + 2] = sag_eval(p, R, x, y, false).z;
}
__global__ __launch_bounds__(256) void k_lens_sag_bwd(const float *eval_xy, const float *base, uint32_t n, float R, const float *p, const float *grad_pos, float *grad_p) {
    __shared__ float s_acc[10];
    if (threadIdx.x < 10) s_acc[threadIdx.x] = 0.f;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    float gz = 0.f; SagPoint sp; for (int k = 0; k < 10; ++k) sp.d[k] = 0.f;
    if (i < n) {
        const float x = eval_xy ? eval_xy[2 * (size_t) i] : base[3 * (size_t) i], y = eval_xy ? eval_xy[2 * (size_t) i + 1] : base[3 * (size_t) i + 1];
        sp = sag_eval(p, R, x, y, true); gz = grad_pos[3 * (size_t) i + 2];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) { const float v = wave_sum(sp.d[k] * gz); if ((threadIdx.x & 63u) == 0u && v != 0.f) atomicAdd(&s_acc[k], v); }
    __syncthreads();
    if (threadIdx.x < 10 && s_acc[threadIdx.x] != 0.f) atomicAdd(grad_p + threadIdx.x, s_acc[threadIdx.x]);
}

// ---- the lens-design loop's loss and optimiser step as single launches ----------------------------------------------------------------
// losses.py:31-41 (rms_loss_and_center): I = mean_c(image / mean(image)), centroid (ibar, jbar) = sum(I * (i, j)) / sum(I), loss =
// sum(I * ((i - ibar)^2 + (j - jbar)^2)) / sum(I); mean and centroid are detached.  One workgroup, double accumulators (the film is 128^2),
// out = { loss, ibar, jbar, mean(image), sum(I) }.
MI_DEV double block_sum_f64(double v, double *s_part) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0; for (uint32_t k = 0; k < blockDim.x / 64u; ++k) t += s_part[k];
    return t;
}
__global__ __launch_bounds__(1024) void k_spot_loss(const float *image, uint32_t H, uint32_t W, float *out) {
    __shared__ double s_part[16];
    const uint32_t n = H * W;
    double s = 0.0, si = 0.0, sj = 0.0;
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {
        const double v = (double) image[3 * (size_t) q] + (double) image[3 * (size_t) q + 1] + (double) image[3 * (size_t) q + 2];
        s += v; si += v * (double) (q / W); sj += v * (double) (q % W);
    }
    s = block_sum_f64(s, s_part); si = block_sum_f64(si, s_part); sj = block_sum_f64(sj, s_part);
    const double ibar = si / s, jbar = sj / s;
    double m2 = 0.0;
    for (uint32_t q = threadIdx.x; q < n; q += blockDim.x) {
        const double v = (double) image[3 * (size_t) q] + (double) image[3 * (size_t) q + 1] + (double) image[3 * (size_t) q + 2];
        const double di = (double) (q / W) - ibar, dj = (double) (q % W) - jbar;
        m2 += v * (di * di + dj * dj);
    }
    m2 = block_sum_f64(m2, s_part);
    if (threadIdx.x == 0) {
        const double mean = s / (3.0 * (double) n);
        out[0] = (float) (m2 / s); out[1] = (float) ibar; out[2] = (float) jbar; out[3] = (float) mean; out[4] = (float) (s / (3.0 * mean));
    }
}
// d loss / d image[p][c] = ((i - ibar)^2 + (j - jbar)^2 - loss) / sum(I) / (3 mean), times the incoming adjoint of the loss
__global__ __launch_bounds__(256) void k_spot_loss_bwd(const float *fwd, const float *grad_loss, uint32_t H, uint32_t W, float *grad_image) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= H * W) return;
    const float di = (float) (q / W) - fwd[1], dj = (float) (q % W) - fwd[2];
    const float g = grad_loss[0] * (fmaf(di, di, dj * dj) - fwd[0]) / (fwd[4] * 3.f * fwd[3]);
    grad_image[3 * (size_t) q] = g; grad_image[3 * (size_t) q + 1] = g; grad_image[3 * (size_t) q + 2] = g;
}
// optimizers.py:263-302 (Adam.step without mask_updates) for up to MI_ADAM_SLOTS small tensors per launch; the table travels as a kernel argument
#define MI_ADAM_SLOTS 64u
struct AdamTable { mi355_adam_slot slot[MI_ADAM_SLOTS]; };
__global__ __launch_bounds__(256) void k_adam(AdamTable t, float beta_1, float one_minus_beta_1, float beta_2, float one_minus_beta_2, float epsilon) {
    const mi355_adam_slot &a = t.slot[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < a.count; i += 256u) {
        const float g = a.grad[i];
        const float m = beta_1 * a.m[i] + one_minus_beta_1 * g, v = beta_2 * a.v[i] + one_minus_beta_2 * g * g;
        a.m[i] = m; a.v[i] = v;
        a.param_out[i] = a.param[i] - a.lr_t * m / (sqrtf(v) + epsilon);
    }
}

// HDRFilm::develop, hdrfilm.cpp:304-409
__global__ void k_develop(const float *film, float *image, uint32_t n_pixels) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float4 v = reinterpret_cast<const float4 *>(film)[i];
    float dv = v.w == 0.f ? 1.f : v.w;
    image[3 * i] = v.x / dv; image[3 * i + 1] = v.y / dv; image[3 * i + 2] = v.z / dv;
}

__global__ __launch_bounds__(BLOCK) void k_debug_intersect(DScene sc, const float *rays, uint32_t n, float *hits) {
    __shared__ uint32_t smem[WAVES * STACK_WORDS];
    uint32_t *stk = smem + (threadIdx.x >> 6) * STACK_WORDS + (threadIdx.x & 63);   // (this kernel is the FULL variant)
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x; bool act = i < n; uint32_t ii = act ? i : 0;
    Ray r; r.o = ld3(rays + 8 * (size_t) ii); r.d = ld3(rays + 8 * (size_t) ii + 3); r.maxt = rays[8 * (size_t) ii + 6];
    Hit h = scene_intersect<true>(sc, r, act, stk);
    bool occ = scene_ray_test<true>(sc, r, act, stk);
    if (act) {
        float *o = hits + 8 * (size_t) i;
        o[0] = h.t; o[1] = h.u; o[2] = h.v; o[3] = __uint_as_float(h.prim); o[4] = __uint_as_float(h.shape);
        o[5] = __uint_as_float(occ ? 1u : 0u); o[6] = 0.f; o[7] = 0.f;
    }
}

// compute_surface_interaction of the closest hit (the FULL variant, like k_debug_intersect): 24 floats per ray, mi355.h
__global__ __launch_bounds__(BLOCK) void k_debug_si(DScene sc, const float *rays, uint32_t n, float *out) {
    __shared__ uint32_t smem[WAVES * STACK_WORDS];
    uint32_t *stk = smem + (threadIdx.x >> 6) * STACK_WORDS + (threadIdx.x & 63);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x; bool act = i < n; uint32_t ii = act ? i : 0;
    Ray r; r.o = ld3(rays + 8 * (size_t) ii); r.d = ld3(rays + 8 * (size_t) ii + 3); r.maxt = rays[8 * (size_t) ii + 6];
    Hit h = scene_intersect<true>(sc, r, act, stk);
    if (!act) return;
    float *o = out + 24 * (size_t) i;
    for (int k = 0; k < 24; ++k) o[k] = 0.f;
    o[14] = h.t; o[15] = __uint_as_float(0xffffffffu);
    if (h.t == MI_INF) return;
    SI si = compute_si<true>(sc, r, h);
    o[0] = si.p.x; o[1] = si.p.y; o[2] = si.p.z; o[3] = si.n.x; o[4] = si.n.y; o[5] = si.n.z; o[6] = si.uv.x; o[7] = si.uv.y;
    o[8] = si.sh_n.x; o[9] = si.sh_n.y; o[10] = si.sh_n.z; o[11] = si.wi.x; o[12] = si.wi.y; o[13] = si.wi.z; o[14] = si.t;
    o[15] = __uint_as_float((uint32_t) si.shape);
    o[16] = si.sh_s.x; o[17] = si.sh_s.y; o[18] = si.sh_s.z; o[19] = si.sh_t.x; o[20] = si.sh_t.y; o[21] = si.sh_t.z;
}
// BasicPRBIntegrator.trace (the fork's geometric ray tracer, src/python/python/ad/integrators/prb_basic.py:46-140): one lane per caller
// ray; at every interaction the lobe bsdf.sample() returns for sample1 = 1.0 and sample2 = (0, 0) (:106-112) -- a dielectric refracts unless
// the Fresnel term is 1 (dielectric.cpp:275: selected_r = sample1 <= r_i), a diffuse surface continues with its albedo as the weight, a
// black one ends the ray (:123).  Loop state as in :64-77: ray, penultimate_ray, depth, throughput, active; `si` of the LAST iteration is
// what the function returns (:134-140).  Scene records are read per lane (divergent caller rays); the FULL variant like k_debug_intersect.
__global__ __launch_bounds__(BLOCK) void k_trace(DScene sc, const float *rays, uint32_t n, uint32_t num_lenses, uint32_t max_depth, float *out) {
    __shared__ uint32_t smem[WAVES * STACK_WORDS];
    uint32_t *stk = smem + (threadIdx.x >> 6) * STACK_WORDS + (threadIdx.x & 63);
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x; const bool act = i < n; const uint32_t ii = act ? i : 0;
    Ray ray; ray.o = ld3(rays + 8 * (size_t) ii); ray.d = ld3(rays + 8 * (size_t) ii + 3); ray.maxt = rays[8 * (size_t) ii + 6];
    Ray pen = ray;
    uint32_t depth = 0; v3 beta = V3(1.f, 1.f, 1.f), p_last = V3(0.f, 0.f, 0.f);
    bool active = act, si_valid = false;
    while (__any(active)) {                                                 // every live lane gains a surface or stops: at most num_lenses trips
        Hit hit = scene_intersect<true, true>(sc, ray, active, stk);         // watertight triangle test: see tri_intersect_watertight
        if (active) {
            const bool valid = hit.t != MI_INF;
            bool active_next = valid && (depth + 1u < max_depth);           // :99
            pen = ray; si_valid = valid; p_last = V3(0.f, 0.f, 0.f);        // :116
            if (valid) {
                SI si = compute_si<true>(sc, ray, hit);
                p_last = si.p;
                const DBsdf &bsdf = sc.bsdfs[sc.shapes[hit.shape].bsdf];
                v3 refl = bsdf.texture >= 0 ? texture_eval(sc, sc.textures[bsdf.texture], si.uv, nullptr) : ld3(bsdf.reflectance);
                BSample bs; v3 w; bsdf_sample<true>(bsdf, refl, si.wi, 1.0f, 0.f, 0.f, active_next, bs, w);   // :106-112
                ray = spawn_ray(si.p, si.n, si_to_world(si, bs.wo));         // :119
                beta = beta * w;                                             // :120
                active_next = active_next && (beta.x != 0.f || beta.y != 0.f || beta.z != 0.f);   // :123
                depth += 1u;                                                 // :126
                active_next = active_next && depth < num_lenses;             // :129
            }
            active = active_next;
        }
    }
    if (!act) return;
    float *o = out + 12 * (size_t) i;
    o[0] = pen.o.x; o[1] = pen.o.y; o[2] = pen.o.z; o[3] = pen.d.x; o[4] = pen.d.y; o[5] = pen.d.z;
    o[6] = p_last.x; o[7] = p_last.y; o[8] = p_last.z; o[9] = si_valid ? 1.f : 0.f; o[10] = (float) depth; o[11] = 0.f;
}
__global__ __launch_bounds__(BLOCK) void k_debug_texture(DScene sc, uint32_t tex, const float *uv, uint32_t n, float *out) {
    uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    v2 q; q.x = uv[2 * (size_t) i]; q.y = uv[2 * (size_t) i + 1];
    v3 c = texture_eval(sc, sc.textures[tex], q, nullptr);
    out[3 * (size_t) i] = c.x; out[3 * (size_t) i + 1] = c.y; out[3 * (size_t) i + 2] = c.z;
}

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MI355_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" const char *mi355_last_error(void) { return g_err.c_str(); }
extern "C" uint32_t mi355_abi_version(void) { return MI355_ABI_VERSION; }

// ---- host float helpers (same operation order as the device code) ------------------------------
struct h3 { float x, y, z; };
static inline h3 H3(float x, float y, float z) { h3 r = { x, y, z }; return r; }
static inline float hdot(h3 a, h3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline h3 hcross(h3 a, h3 b) { return H3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))); }
static inline h3 hsub(h3 a, h3 b) { return H3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline h3 hscale(h3 a, float s) { return H3(a.x * s, a.y * s, a.z * s); }
static inline h3 hxf_vector(const float *m, h3 v) {
    h3 r = H3(m[0] * v.x, m[4] * v.x, m[8] * v.x);
    r = H3(fmaf(m[1], v.y, r.x), fmaf(m[5], v.y, r.y), fmaf(m[9], v.y, r.z));
    r = H3(fmaf(m[2], v.z, r.x), fmaf(m[6], v.z, r.y), fmaf(m[10], v.z, r.z));
    return r;
}
static inline h3 hxf_normal(const float *to_object, h3 n) {
    const float *m = to_object;
    h3 r = H3(m[0] * n.x, m[1] * n.x, m[2] * n.x);
    r = H3(fmaf(m[4], n.y, r.x), fmaf(m[5], n.y, r.y), fmaf(m[6], n.y, r.z));
    r = H3(fmaf(m[8], n.z, r.x), fmaf(m[9], n.z, r.y), fmaf(m[10], n.z, r.z));
    return r;
}
static void hmat4_mul(const float *a, const float *b, float *o) {
    float r[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = a[i * 4 + 0] * b[0 * 4 + j];
        for (int k = 1; k < 4; ++k) s = fmaf(a[i * 4 + k], b[k * 4 + j], s);
        r[i * 4 + j] = s;
    }
    memcpy(o, r, sizeof(r));
}
static void hmat4_identity(float *m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.f; }
static void hmat4_scale(float x, float y, float z, float *m) { hmat4_identity(m); m[0] = x; m[5] = y; m[10] = z; }
static void hmat4_translate(float x, float y, float z, float *m) { hmat4_identity(m); m[3] = x; m[7] = y; m[11] = z; }

// perspective_projection (sensor.h:228-265) + Transform::perspective (transform.h:216-233); also returns the
// analytic inverse that Transform carries along (transform.h:62-71)
static void host_perspective_projection(const int32_t *film, const int32_t *crop, const int32_t *off, float fov_x,
                                        float near_, float far_, float *m, float *inv) {
    float fx = (float) film[0], fy = (float) film[1];
    float rsx = (float) crop[0] / fx, rsy = (float) crop[1] / fy, rox = (float) off[0] / fx, roy = (float) off[1] / fy;
    float aspect = fx / fy;
    float recip = 1.f / (far_ - near_);
    float tn = tanf((fov_x * .5f) * (MI_PI / 180.f)), cot = 1.f / tn;
    float P[16], Pi[16];
    memset(P, 0, 64); P[0] = cot; P[5] = cot; P[10] = far_ * recip; P[11] = -near_ * far_ * recip; P[14] = 1.f;
    memset(Pi, 0, 64); Pi[0] = tn; Pi[5] = tn; Pi[15] = 1.f / near_; Pi[11] = 1.f; Pi[14] = (near_ - far_) / (far_ * near_);
    float S1[16], T1[16], S2[16], T2[16], S1i[16], T1i[16], S2i[16], T2i[16], t[16];
    hmat4_scale(1.f / rsx, 1.f / rsy, 1.f, S1);      hmat4_scale(1.f / (1.f / rsx), 1.f / (1.f / rsy), 1.f, S1i);
    hmat4_translate(-rox, -roy, 0.f, T1);            hmat4_translate(rox, roy, 0.f, T1i);
    hmat4_scale(-0.5f, -0.5f * aspect, 1.f, S2);     hmat4_scale(1.f / -0.5f, 1.f / (-0.5f * aspect), 1.f, S2i);
    hmat4_translate(-1.f, -1.f / aspect, 0.f, T2);   hmat4_translate(1.f, 1.f / aspect, 0.f, T2i);
    hmat4_mul(S1, T1, t); hmat4_mul(t, S2, t); hmat4_mul(t, T2, t); hmat4_mul(t, P, m);
    hmat4_mul(Pi, T2i, t); hmat4_mul(t, S2i, t); hmat4_mul(t, T1i, t); hmat4_mul(t, S1i, inv);
}
extern "C" void mi355_perspective_projection(const int32_t film_size[2], const int32_t crop_size[2], const int32_t crop_offset[2],
                                             float fov_x, float near_clip, float far_clip, float out_matrix[16]) {
    float inv[16]; host_perspective_projection(film_size, crop_size, crop_offset, fov_x, near_clip, far_clip, out_matrix, inv);
}
static float host_estrin10(float x, const float *c) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4;
    float a0 = fmaf(x, c[1], c[0]), a1 = fmaf(x, c[3], c[2]), a2 = fmaf(x, c[5], c[4]), a3 = fmaf(x, c[7], c[6]), a4 = fmaf(x, c[9], c[8]);
    float b0 = fmaf(x2, a1, a0), b1 = fmaf(x2, a3, a2);
    return fmaf(x8, a4, fmaf(x4, b1, b0));
}
// GaussianFilter ctor, gaussian.cpp:46-86
extern "C" void mi355_gaussian_coeffs(float stddev, float out_coeff[10]) {
    static const double coeff[10] = { 9.992604880e-1, -4.977025247e-1, 1.222248550e-1, -1.932406282e-2, 2.136713061e-3,
                                      -1.679873860e-4, 9.202145248e-6, -3.329417433e-7, 7.128382794e-9, -6.821193280e-11 };
    double scale = 1; float radius = 4.f * stddev;
    for (int i = 0; i < 10; ++i) { out_coeff[i] = (float) (coeff[i] * scale); scale /= ((double) stddev * (double) stddev); }
    out_coeff[0] -= host_estrin10(radius * radius, out_coeff);
}

// ---- scene object --------------------------------------------------------------------------------
#define MAX_EVENTS 64
// One sensor of the scene (Scene::sensors()): its description and its share of the kernel argument.  `snap` is a DScene of which only the
// sensor / film fields are meaningful (copy_sensor_fields lists them); selecting a sensor copies them into mi355_scene::ds -- DScene keeps
// its field order, which the forward kernel's scalar loads are tuned to (mi355_device.h).
struct SensorSlot {
    mi355_sensor_desc desc;                     // desc.lens_data points into `lens`
    DScene snap;
    std::vector<float> lens, pupil; float *d_lens = nullptr, *d_pupil = nullptr;
};
struct mi355_scene {
    mi355_reduce_fn reduce_fn = nullptr; void *reduce_user = nullptr;      // mi355_set_reduce_hook: in-place SUM over the ranks of a device buffer
    int device = 0;
    DScene ds;                                  // kernel argument (device pointers inside); its sensor fields are those of sensor `cur_sensor`
    std::vector<SensorSlot> sensors; uint32_t cur_sensor = 0xffffffffu; uint32_t base_features = 0;   // feature bits of the scene without the sensor's (bit 2)
    std::vector<DShape> shapes; std::vector<DTri> tris; std::vector<uint32_t> tri_face;
    std::vector<uint32_t> faces; std::vector<float> positions, normals, texcoords;
    std::vector<DPrim> prims; std::vector<DPrimFin> fins; std::vector<uint32_t> bvh_shapes; std::vector<DBox> boxes; std::vector<DSphere> spheres; std::vector<float> area_cdf;
    std::vector<float> ext_iors;                                            // per bsdf (the design forward mode turns IOR tangents into a tangent of rel_eta)
    // geometry updates (MI355_PARAM_VERTEX_POSITIONS): what Mesh::parameters_changed / Scene::parameters_changed need
    std::vector<uint32_t> vcount;                                           // per shape: vertex count (meshes)
    std::vector<uint32_t> vadj_off, vadj;                                   // CSR over ALL mesh vertices: incident (local face * 4 + corner), ascending faces
    std::vector<uint32_t> bvh_levels, bvh_level_off;                        // BVH nodes grouped by depth, root level first (refit runs deepest level first)
    std::vector<uint32_t> tri_shape; uint32_t *d_tri_shape = nullptr; uint64_t n_faces_total = 0;
    std::vector<char> dirty; bool geom_dirty = false;                       // per shape: positions changed since the last refit
    uint32_t *d_vadj_off = nullptr, *d_vadj = nullptr; float *d_nrm_len = nullptr, *d_nrm_adj = nullptr; uint32_t *d_bvh_levels = nullptr;
    float *d_tape = nullptr; size_t tape_floats = 0;                        // path tape of the geometry adjoint
    float *d_ntan = nullptr; size_t ntan_floats = 0;                        // forward design mode: tangent of the vertex normals
    size_t cap_prims = 0, cap_fins = 0, cap_boxes = 0;                      // device capacities of the flat tables (they are rebuilt when a small mesh moves)
    std::vector<DBvh4> bvh; std::vector<DBsdf> bsdfs; std::vector<DEmitter> emitters; std::vector<DTexture> textures;
    std::vector<float> texdata;
    std::vector<mi355_param_info> layout; uint64_t grad_total = 0;
    std::vector<uint32_t> small_map;
    // device buffers
    DShape *d_shapes = nullptr; DTri *d_tris = nullptr; uint32_t *d_faces = nullptr, *d_tri_face = nullptr, *d_bvh_shapes = nullptr;
    DPrim *d_prims = nullptr; DPrimFin *d_fins = nullptr; DBox *d_boxes = nullptr; DSphere *d_spheres = nullptr; float *d_area_cdf = nullptr; float *d_positions = nullptr, *d_normals = nullptr,
          *d_texcoords = nullptr; DBvh4 *d_bvh = nullptr; DBvhQ *d_bvhq = nullptr; DBsdf *d_bsdfs = nullptr; DEmitter *d_emitters = nullptr;
    DTexture *d_textures = nullptr; float *d_texdata = nullptr; uint32_t *d_small_map = nullptr;
    float4 *d_gw = nullptr; size_t gw_pixels = 0;   // grad_in / W scratch (crop_h x crop_w of the largest film served)
    uint32_t *d_redo = nullptr; uint32_t redo_capacity = 0;   // per-workgroup "redo with the two-pass kernel" flags
    // deferred splat of the particle tracer: record queue (unsorted + sorted copy), tile histogram / offsets / cursors, record counter
    float4 *d_rec_a = nullptr, *d_srt_a = nullptr; float *d_rec_b = nullptr, *d_srt_b = nullptr; uint32_t rec_capacity = 0;
    float4 *d_rec_c = nullptr, *d_srt_c = nullptr; float *d_rec_d = nullptr, *d_srt_d = nullptr; uint32_t rec2_capacity = 0;   // derivative splats: second record half
    uint32_t *d_tile_u32 = nullptr; uint32_t tile_capacity = 0; uint32_t splat_cap_override = 0;
    uint32_t *d_pt_keys = nullptr, *d_pt_order = nullptr, *d_pt_bins = nullptr; uint32_t pt_order_capacity = 0;   // pt_sort_lanes: keys, 2 sorted sample lists, 2 x 64 Ki bins
    struct PtOrderTag { uint32_t w[18]; hipStream_t stream; bool valid; } pt_tag[2] = {};   // what each of the two lists was sorted for (pt_sort_lanes)
    uint32_t pt_next = 0;
    unsigned long long *d_counters = nullptr;   // [0] unused, [1] isect, [2] shade, [3] shadow, [4] redone backward workgroups
    uint64_t n_gen = 0, n_splat = 0;
    hipEvent_t ev[MAX_EVENTS][2]; int n_ev = 0; bool ev_created = false;
    float ms_accum = 0.f; uint32_t launches_accum = 0;
};

template <typename T> static int upload(T **dst, const std::vector<T> &src) {
    size_t n = std::max<size_t>(src.size(), 1);
    HIP_TRY(hipMalloc((void **) dst, n * sizeof(T)));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return MI355_OK;
}

// ---- scene BVH: ONE tree over the triangles of every mesh that is not flattened --------------------------------------------------------
// Binary tree by binned SAH (16 bins on each axis, leaves of <= 4 triangles), depth-limited so that the collapsed tree has at most
// MI_BVH_LEVELS levels: where the budget gets tight the split falls back to the object median, which halves the count.  Two binary
// levels collapse into one DBvh4 node.  Topology is fixed after the build; `vertex_positions` updates refit the boxes on the device.
struct BuildRef { float lo[3], hi[3], c[3]; uint32_t shape, face; };
struct BinNode { float lo[3], hi[3]; int32_t left, right; uint32_t first, count; };
static inline float box_area(const float *lo, const float *hi) { float d[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] }; return d[0] < 0.f ? 0.f : 2.f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
static int32_t bin_build(std::vector<BinNode> &nodes, std::vector<BuildRef> &r, uint32_t begin, uint32_t end, int depth, int max_depth) {
    const int32_t id = (int32_t) nodes.size(); nodes.push_back(BinNode());
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY }, clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (uint32_t i = begin; i < end; ++i) for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], r[i].lo[k]); hi[k] = std::max(hi[k], r[i].hi[k]); clo[k] = std::min(clo[k], r[i].c[k]); chi[k] = std::max(chi[k], r[i].c[k]);
    }
    for (int k = 0; k < 3; ++k) { nodes[id].lo[k] = lo[k]; nodes[id].hi[k] = hi[k]; }
    const uint32_t n = end - begin;
    nodes[id].left = nodes[id].right = -1; nodes[id].first = begin; nodes[id].count = n;
    if (n <= 4) return id;
    // levels still available below this node vs. levels a balanced split of n needs: ceil(log2(ceil(n / 4)))
    int need = 0; for (uint32_t m = (n + 3) / 4; m > 1; m = (m + 1) / 2) ++need;
    const bool tight = depth + need >= max_depth;
    uint32_t mid = begin + n / 2; int axis = 0; bool sah_ok = false;
    if (!tight) {
        const int NB = 16; float best = INFINITY; int best_axis = -1, best_bin = -1;
        for (int ax = 0; ax < 3; ++ax) {
            const float ext = chi[ax] - clo[ax]; if (!(ext > 0.f)) continue;
            float blo[NB][3], bhi[NB][3]; uint32_t cnt[NB];
            for (int b = 0; b < NB; ++b) { cnt[b] = 0; for (int k = 0; k < 3; ++k) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; } }
            const float sc = (float) NB / ext;
            for (uint32_t i = begin; i < end; ++i) {
                int b = std::min(NB - 1, std::max(0, (int) ((r[i].c[ax] - clo[ax]) * sc)));
                cnt[b]++; for (int k = 0; k < 3; ++k) { blo[b][k] = std::min(blo[b][k], r[i].lo[k]); bhi[b][k] = std::max(bhi[b][k], r[i].hi[k]); }
            }
            float rarea[NB]; uint32_t rcnt[NB]; float al[3] = { INFINITY, INFINITY, INFINITY }, ah[3] = { -INFINITY, -INFINITY, -INFINITY }; uint32_t c = 0;
            for (int b = NB - 1; b > 0; --b) { for (int k = 0; k < 3; ++k) { al[k] = std::min(al[k], blo[b][k]); ah[k] = std::max(ah[k], bhi[b][k]); } c += cnt[b]; rarea[b] = box_area(al, ah); rcnt[b] = c; }
            for (int k = 0; k < 3; ++k) { al[k] = INFINITY; ah[k] = -INFINITY; } c = 0;
            for (int b = 0; b < NB - 1; ++b) {
                for (int k = 0; k < 3; ++k) { al[k] = std::min(al[k], blo[b][k]); ah[k] = std::max(ah[k], bhi[b][k]); } c += cnt[b];
                if (c == 0 || rcnt[b + 1] == 0) continue;
                const float cost = box_area(al, ah) * (float) ((c + 3) / 4) + rarea[b + 1] * (float) ((rcnt[b + 1] + 3) / 4);
                if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
            }
        }
        if (best_axis >= 0) {
            const float ext = chi[best_axis] - clo[best_axis], sc = 16.f / ext, c0 = clo[best_axis]; const int ax = best_axis, bb = best_bin;
            auto it = std::partition(r.begin() + begin, r.begin() + end, [=](const BuildRef &a) { return std::min(15, std::max(0, (int) ((a.c[ax] - c0) * sc))) <= bb; });
            const uint32_t m = (uint32_t) (it - r.begin());
            // a lopsided split must still leave both sides shallow enough
            auto fits = [&](uint32_t cnt_) { int nd = 0; for (uint32_t q = (cnt_ + 3) / 4; q > 1; q = (q + 1) / 2) ++nd; return depth + 1 + nd <= max_depth; };
            if (m > begin && m < end && fits(m - begin) && fits(end - m)) { mid = m; sah_ok = true; }
        }
    }
    if (!sah_ok) {
        float ext = chi[0] - clo[0]; for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > ext) { ext = chi[k] - clo[k]; axis = k; }
        mid = begin + n / 2;
        std::nth_element(r.begin() + begin, r.begin() + mid, r.begin() + end, [axis](const BuildRef &a, const BuildRef &b) { return a.c[axis] < b.c[axis]; });
    }
    const int32_t l = bin_build(nodes, r, begin, mid, depth + 1, max_depth), rr = bin_build(nodes, r, mid, end, depth + 1, max_depth);
    nodes[id].left = l; nodes[id].right = rr;
    return id;
}
static uint32_t bvh4_collapse(const std::vector<BinNode> &bn, int32_t b, std::vector<DBvh4> &out, std::vector<uint32_t> &depth_of, uint32_t depth) {
    const uint32_t id = (uint32_t) out.size(); out.push_back(DBvh4()); depth_of.push_back(depth);
    int32_t kids[4]; int nk = 0;
    const int32_t two[2] = { bn[b].left, bn[b].right };
    for (int k = 0; k < 2; ++k) {                                           // grandchildren where the child is an inner node
        const BinNode &c = bn[two[k]];
        if (c.left >= 0) { kids[nk++] = c.left; kids[nk++] = c.right; } else kids[nk++] = two[k];
    }
    DBvh4 nd; memset(&nd, 0, sizeof(nd));
    for (int k = 0; k < 4; ++k) nd.entry[k] = MI_BVH_EMPTY;
    for (int k = 0; k < nk; ++k) {
        const BinNode &c = bn[kids[k]];
        for (int a = 0; a < 3; ++a) { nd.lo[a][k] = c.lo[a]; nd.hi[a][k] = c.hi[a]; }
        if (c.left < 0) nd.entry[k] = MI_BVH_LEAF | ((c.count - 1u) << 29) | c.first;
        else nd.entry[k] = bvh4_collapse(bn, kids[k], out, depth_of, depth + 1);
    }
    out[id] = nd;
    return id;
}
// builds s->bvh, s->tris / tri_face / tri_shape (leaf order) and the refit level lists from the host mirrors of the BVH shapes
// DPrim::m in device order (MI_PRIM_INTERLEAVE, mi355_device.h): the host builds rows, the kernels read interleaved (s, t) pairs
static std::vector<DPrim> prims_for_device(const std::vector<DPrim> &in) {
    std::vector<DPrim> out(in);
    for (size_t k = 0; k < in.size(); ++k) for (int i = 0; i < 12; ++i) out[k].m[MI_PM(i)] = in[k].m[i];
    return out;
}
static int build_scene_bvh(mi355_scene *s) {
    Phase ph(PHASE_INIT_ACCEL);
    s->bvh.clear(); s->tris.clear(); s->tri_face.clear(); s->tri_shape.clear(); s->bvh_levels.clear(); s->bvh_level_off.assign(1, 0u);
    if (s->bvh_shapes.empty()) return MI355_OK;
    std::vector<BuildRef> refs;
    for (uint32_t si : s->bvh_shapes) {
        const DShape &sh = s->shapes[si];
        const float *P = s->positions.data() + 3 * (size_t) sh.vtx_offset;
        for (uint32_t f = 0; f < sh.tri_count; ++f) {
            const uint32_t *fi = s->faces.data() + 3 * (size_t) (sh.tri_offset + f);
            BuildRef r; r.shape = si; r.face = f;
            for (int k = 0; k < 3; ++k) {
                const float a = P[3 * (size_t) fi[0] + k], b = P[3 * (size_t) fi[1] + k], c = P[3 * (size_t) fi[2] + k];
                r.lo[k] = std::min(a, std::min(b, c)); r.hi[k] = std::max(a, std::max(b, c)); r.c[k] = (r.lo[k] + r.hi[k]) * .5f;
            }
            refs.push_back(r);
        }
    }
    if (refs.size() >= (1u << 29)) return fail(MI355_ERR_UNSUPPORTED, "more than 2^29 triangles in the scene BVH");
    std::vector<BinNode> bn; bn.reserve(refs.size());
    const int max_bin_depth = 2 * (MI_BVH_LEVELS - 1) + 1;                  // binary leaves at depth <= this collapse into <= MI_BVH_LEVELS levels of DBvh4 nodes
    bin_build(bn, refs, 0, (uint32_t) refs.size(), 0, max_bin_depth);
    std::vector<uint32_t> depth_of;
    if (bn[0].left < 0) return fail(MI355_ERR_INVALID, "scene BVH over fewer than five triangles");
    bvh4_collapse(bn, 0, s->bvh, depth_of, 0);
    uint32_t levels = 0; for (uint32_t d : depth_of) levels = std::max(levels, d + 1);
    if (levels > MI_BVH_LEVELS) return fail(MI355_ERR_UNSUPPORTED, "scene BVH deeper than the traversal stack allows (MI_BVH_LEVELS): too many triangles for this build");
    for (const BuildRef &r : refs) {                                        // leaf order; DTri carries p0, e1, e2 (mesh.h:436)
        const DShape &sh = s->shapes[r.shape];
        const uint32_t *fi = s->faces.data() + 3 * (size_t) (sh.tri_offset + r.face);
        const float *P = s->positions.data() + 3 * (size_t) sh.vtx_offset, *a = P + 3 * (size_t) fi[0], *b = P + 3 * (size_t) fi[1], *c = P + 3 * (size_t) fi[2];
        DTri t; for (int q = 0; q < 3; ++q) { t.p0[q] = a[q]; t.e1[q] = b[q] - a[q]; t.e2[q] = c[q] - a[q]; }
        s->tris.push_back(t); s->tri_face.push_back(r.face); s->tri_shape.push_back(r.shape);
    }
    std::vector<std::vector<uint32_t>> lv(levels);
    for (uint32_t i = 0; i < depth_of.size(); ++i) lv[depth_of[i]].push_back(i);
    for (auto &l : lv) { s->bvh_levels.insert(s->bvh_levels.end(), l.begin(), l.end()); s->bvh_level_off.push_back((uint32_t) s->bvh_levels.size()); }
    return MI355_OK;
}

// ---- flat primitive list ------------------------------------------------------------------------------
// world -> (s, t, h) for the frame P0 + s*Es + t*Et (h along Es x Et), computed in double, stored in float
static bool frame_matrix(const double P0[3], const double Es[3], const double Et[3], float m[12]) {
    double N[3] = { Es[1] * Et[2] - Es[2] * Et[1], Es[2] * Et[0] - Es[0] * Et[2], Es[0] * Et[1] - Es[1] * Et[0] };
    double B[3][3] = { { Es[0], Et[0], N[0] }, { Es[1], Et[1], N[1] }, { Es[2], Et[2], N[2] } };
    double det = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) +
                 B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
    if (!(fabs(det) > 0.0) || !std::isfinite(det)) return false;
    double inv[3][3];
    inv[0][0] = (B[1][1] * B[2][2] - B[1][2] * B[2][1]) / det; inv[0][1] = (B[0][2] * B[2][1] - B[0][1] * B[2][2]) / det; inv[0][2] = (B[0][1] * B[1][2] - B[0][2] * B[1][1]) / det;
    inv[1][0] = (B[1][2] * B[2][0] - B[1][0] * B[2][2]) / det; inv[1][1] = (B[0][0] * B[2][2] - B[0][2] * B[2][0]) / det; inv[1][2] = (B[0][2] * B[1][0] - B[0][0] * B[1][2]) / det;
    inv[2][0] = (B[1][0] * B[2][1] - B[1][1] * B[2][0]) / det; inv[2][1] = (B[0][1] * B[2][0] - B[0][0] * B[2][1]) / det; inv[2][2] = (B[0][0] * B[1][1] - B[0][1] * B[1][0]) / det;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[r * 4 + c] = (float) inv[r][c];
        m[r * 4 + 3] = (float) -(inv[r][0] * P0[0] + inv[r][1] * P0[1] + inv[r][2] * P0[2]);
    }
    return true;
}
// affine map (s,t) -> (b1,b2) of a triangle whose vertices have frame coordinates (sv[i], tv[i])
static bool bary_map(const double sv[3], const double tv[3], float out[6]) {
    double S[3][3] = { { sv[0], tv[0], 1 }, { sv[1], tv[1], 1 }, { sv[2], tv[2], 1 } };
    double det = S[0][0] * (S[1][1] * S[2][2] - S[1][2] * S[2][1]) - S[0][1] * (S[1][0] * S[2][2] - S[1][2] * S[2][0]) +
                 S[0][2] * (S[1][0] * S[2][1] - S[1][1] * S[2][0]);
    if (!(fabs(det) > 1e-300)) return false;
    for (int which = 0; which < 2; ++which) {           // b1: rhs (0,1,0); b2: rhs (0,0,1)
        double rhs[3] = { 0, which == 0 ? 1.0 : 0.0, which == 1 ? 1.0 : 0.0 };
        for (int c = 0; c < 3; ++c) {                   // Cramer
            double M[3][3]; memcpy(M, S, sizeof(M));
            for (int r = 0; r < 3; ++r) M[r][c] = rhs[r];
            double dc = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                        M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
            out[which * 3 + c] = (float) (dc / det);
        }
    }
    return true;
}
static void frame_coords(const float m[12], const float *p, double &s, double &t) {
    s = (double) m[0] * p[0] + (double) m[1] * p[1] + (double) m[2] * p[2] + m[3];
    t = (double) m[4] * p[0] + (double) m[5] * p[1] + (double) m[6] * p[2] + m[7];
}
// Expand a small mesh into flat primitives: coplanar triangle pairs that form a parallelogram become ONE primitive.
static void flatten_mesh(mi355_scene *s, uint32_t shape_index, const mi355_shape_desc &sd) {
    const uint32_t F = sd.face_count; const float *P = sd.vertex_positions;
    std::vector<char> used(F, 0);
    auto V = [&](uint32_t f, int k) { return P + 3 * (size_t) sd.faces[3 * (size_t) f + k]; };
    auto same = [&](const float *a, const float *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
    for (uint32_t f = 0; f < F; ++f) {
        if (used[f]) continue;
        used[f] = 1;
        int partner = -1; int ua = 0, ub = 0;           // indices (within each triangle) of the UNSHARED vertex
        for (uint32_t g = f + 1; g < F && partner < 0; ++g) {
            if (used[g]) continue;
            int shared_f[3] = { 0, 0, 0 }, shared_g[3] = { 0, 0, 0 }, ns = 0;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) if (same(V(f, i), V(g, j))) { shared_f[i] = 1; shared_g[j] = 1; ns++; }
            if (ns != 2) continue;
            int a = !shared_f[0] ? 0 : (!shared_f[1] ? 1 : 2), b = !shared_g[0] ? 0 : (!shared_g[1] ? 1 : 2);
            const float *pa = V(f, a), *pb = V(g, b), *pu = V(f, (a + 1) % 3), *pv = V(f, (a + 2) % 3);
            double scale = 0, err = 0;
            for (int k = 0; k < 3; ++k) { double e = (double) pa[k] + pb[k] - pu[k] - pv[k]; err += e * e; scale += (double) (pa[k] - pu[k]) * (pa[k] - pu[k]) + (double) (pb[k] - pu[k]) * (pb[k] - pu[k]); }
            if (err > 1e-10 * scale) continue;          // not a parallelogram (this also implies coplanarity)
            // same orientation? normals of f and g must agree
            auto nrm = [&](uint32_t q, double n[3]) { const float *x = V(q, 0), *y = V(q, 1), *z = V(q, 2);
                double e1[3] = { (double) y[0] - x[0], (double) y[1] - x[1], (double) y[2] - x[2] }, e2[3] = { (double) z[0] - x[0], (double) z[1] - x[1], (double) z[2] - x[2] };
                n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0]; };
            double nf[3], ng[3]; nrm(f, nf); nrm(g, ng);
            if (nf[0] * ng[0] + nf[1] * ng[1] + nf[2] * ng[2] <= 0) continue;
            partner = (int) g; ua = a; ub = b;
        }
        DPrim pr; memset(&pr, 0, sizeof(pr)); pr.shape = shape_index; pr.fin = (uint32_t) s->fins.size();
        DPrimFin fin; memset(&fin, 0, sizeof(fin));
        bool ok = false;
        if (partner >= 0) {
            const float *pa = V(f, ua), *pb = V((uint32_t) partner, ub), *pu = V(f, (ua + 1) % 3);
            double P0[3] = { pu[0], pu[1], pu[2] }, Es[3] = { (double) pa[0] - pu[0], (double) pa[1] - pu[1], (double) pa[2] - pu[2] },
                   Et[3] = { (double) pb[0] - pu[0], (double) pb[1] - pu[1], (double) pb[2] - pu[2] };
            if (frame_matrix(P0, Es, Et, pr.m)) {
                double sv[3], tv[3], sw[3], tw[3];
                for (int k = 0; k < 3; ++k) { frame_coords(pr.m, V(f, k), sv[k], tv[k]); frame_coords(pr.m, V((uint32_t) partner, k), sw[k], tw[k]); }
                if (bary_map(sv, tv, fin.a) && bary_map(sw, tw, fin.b)) {
                    pr.lo = 0.f; pr.smax = 2.f; fin.tri_a = f; fin.tri_b = (uint32_t) partner; used[partner] = 1; ok = true;
                }
            }
        }
        if (!ok) {                                       // single triangle: (s,t) = barycentrics
            const float *p0 = V(f, 0), *p1 = V(f, 1), *p2 = V(f, 2);
            double P0[3] = { p0[0], p0[1], p0[2] }, Es[3] = { (double) p1[0] - p0[0], (double) p1[1] - p0[1], (double) p1[2] - p0[2] },
                   Et[3] = { (double) p2[0] - p0[0], (double) p2[1] - p0[1], (double) p2[2] - p0[2] };
            if (!frame_matrix(P0, Es, Et, pr.m)) continue;   // degenerate triangle: can never be hit (mesh.h:439 would give inf)
            pr.lo = 0.f; pr.smax = 1.f; fin.tri_a = fin.tri_b = f;
            fin.a[0] = 1; fin.a[4] = 1; fin.b[0] = 1; fin.b[4] = 1;
        }
        s->prims.push_back(pr); s->fins.push_back(fin);
    }
}

// Recognise a parallelepiped: 8 distinct corners P0 + {0,1}E0 + {0,1}E1 + {0,1}E2, every face covered by exactly two triangles.
static bool try_make_box(mi355_scene *s, uint32_t shape_index, const mi355_shape_desc &sd) {
    if (sd.face_count != 12) return false;
    const float *P = sd.vertex_positions;
    std::vector<std::array<double, 3>> pts;
    for (uint32_t v = 0; v < sd.vertex_count; ++v) {
        std::array<double, 3> q = { P[3 * v], P[3 * v + 1], P[3 * v + 2] };
        bool dup = false; for (auto &r : pts) if (r == q) { dup = true; break; }
        if (!dup) pts.push_back(q);
        if (pts.size() > 8) return false;
    }
    if (pts.size() != 8) return false;
    double scale = 0; std::array<double, 3> v[7];
    for (int i = 0; i < 7; ++i) for (int k = 0; k < 3; ++k) { v[i][k] = pts[i + 1][k] - pts[0][k]; scale = std::max(scale, fabs(v[i][k])); }
    auto has = [&](const double q[3]) { for (int i = 0; i < 7; ++i) { double e = 0; for (int k = 0; k < 3; ++k) e = std::max(e, fabs(v[i][k] - q[k])); if (e <= 1e-5 * scale) return true; } return false; };
    int E[3] = { -1, -1, -1 };
    for (int a = 0; a < 7 && E[0] < 0; ++a) for (int b = a + 1; b < 7 && E[0] < 0; ++b) for (int c = b + 1; c < 7 && E[0] < 0; ++c) {
        double ab[3], ac[3], bc[3], abc[3];
        for (int k = 0; k < 3; ++k) { ab[k] = v[a][k] + v[b][k]; ac[k] = v[a][k] + v[c][k]; bc[k] = v[b][k] + v[c][k]; abc[k] = ab[k] + v[c][k]; }
        if (has(ab) && has(ac) && has(bc) && has(abc)) { E[0] = a; E[1] = b; E[2] = c; }
    }
    if (E[0] < 0) return false;
    DBox bx; memset(&bx, 0, sizeof(bx));
    {   // world -> local: rows of [E0 E1 E2]^-1
        double B[3][3]; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[r][c] = v[E[c]][r];
        double det = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1]) - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0]) + B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
        if (!(fabs(det) > 1e-12 * scale * scale * scale)) return false;
        double inv[3][3];
        inv[0][0] = (B[1][1] * B[2][2] - B[1][2] * B[2][1]) / det; inv[0][1] = (B[0][2] * B[2][1] - B[0][1] * B[2][2]) / det; inv[0][2] = (B[0][1] * B[1][2] - B[0][2] * B[1][1]) / det;
        inv[1][0] = (B[1][2] * B[2][0] - B[1][0] * B[2][2]) / det; inv[1][1] = (B[0][0] * B[2][2] - B[0][2] * B[2][0]) / det; inv[1][2] = (B[0][2] * B[1][0] - B[0][0] * B[1][2]) / det;
        inv[2][0] = (B[1][0] * B[2][1] - B[1][1] * B[2][0]) / det; inv[2][1] = (B[0][1] * B[2][0] - B[0][0] * B[2][1]) / det; inv[2][2] = (B[0][0] * B[1][1] - B[0][1] * B[1][0]) / det;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) bx.m[r * 4 + c] = (float) inv[r][c];
            bx.m[r * 4 + 3] = (float) -(inv[r][0] * pts[0][0] + inv[r][1] * pts[0][1] + inv[r][2] * pts[0][2]);
        }
    }
    // classify triangles by face; local coordinates of their vertices must be 0/1
    int face_tris[6][2]; int face_n[6] = { 0, 0, 0, 0, 0, 0 };
    double loc[12][3][3];
    for (uint32_t f = 0; f < 12; ++f) {
        for (int k = 0; k < 3; ++k) {
            const float *q = P + 3 * (size_t) sd.faces[3 * f + k];
            for (int r = 0; r < 3; ++r) {
                double x = (double) bx.m[r * 4] * q[0] + (double) bx.m[r * 4 + 1] * q[1] + (double) bx.m[r * 4 + 2] * q[2] + bx.m[r * 4 + 3];
                double rx = x < 0.5 ? 0.0 : 1.0; if (fabs(x - rx) > 1e-4) return false;
                loc[f][k][r] = rx;
            }
        }
        int face = -1;
        for (int a = 0; a < 3; ++a) if (loc[f][0][a] == loc[f][1][a] && loc[f][1][a] == loc[f][2][a]) face = 2 * a + (int) loc[f][0][a];
        if (face < 0 || face_n[face] >= 2) return false;
        face_tris[face][face_n[face]++] = (int) f;
    }
    for (int k = 0; k < 6; ++k) if (face_n[k] != 2) return false;
    bx.shape = shape_index; bx.fin = (uint32_t) s->fins.size();
    for (int face = 0; face < 6; ++face) {
        int a = face / 2, b = a == 0 ? 1 : 0, c = a == 2 ? 1 : 2;
        DPrimFin fin; memset(&fin, 0, sizeof(fin));
        fin.tri_a = (uint32_t) face_tris[face][0]; fin.tri_b = (uint32_t) face_tris[face][1];
        double sv[3], tv[3], sw[3], tw[3];
        for (int k = 0; k < 3; ++k) { sv[k] = loc[fin.tri_a][k][b]; tv[k] = loc[fin.tri_a][k][c]; sw[k] = loc[fin.tri_b][k][b]; tw[k] = loc[fin.tri_b][k][c]; }
        if (!bary_map(sv, tv, fin.a) || !bary_map(sw, tw, fin.b)) return false;
        s->fins.push_back(fin);
    }
    s->boxes.push_back(bx);
    return true;
}

// The sensor / film share of DScene.  Everything else in DScene belongs to the scene replica and is never touched by a sensor change.
static void copy_sensor_fields(DScene &dst, const DScene &src) {
    memcpy(dst.sample_to_camera, src.sample_to_camera, sizeof(dst.sample_to_camera)); memcpy(dst.cam_to_world, src.cam_to_world, sizeof(dst.cam_to_world));
    dst.near_clip = src.near_clip; dst.far_clip = src.far_clip;
    dst.crop_w = src.crop_w; dst.crop_h = src.crop_h; dst.crop_x = src.crop_x; dst.crop_y = src.crop_y; dst.grid_w = src.grid_w; dst.grid_h = src.grid_h; dst.border = src.border;
    dst.rfilter = src.rfilter; dst.filter_radius = src.filter_radius; memcpy(dst.filter_coeff, src.filter_coeff, sizeof(dst.filter_coeff));
    dst.sensor_type = src.sensor_type; dst.lens_count = src.lens_count; dst.lens = src.lens; dst.film_px = src.film_px; dst.film_py = src.film_py;
    dst.film_w = src.film_w; dst.film_h = src.film_h;
    memcpy(dst.camera_to_sample, src.camera_to_sample, sizeof(dst.camera_to_sample)); memcpy(dst.image_rect, src.image_rect, sizeof(dst.image_rect)); dst.normalization = src.normalization;
    dst.shutter_open_time = src.shutter_open_time; dst.sample_exit_pupil = src.sample_exit_pupil; dst.film_diagonal = src.film_diagonal; dst.pupil = src.pupil;
}
static int validate_sensor(const mi355_sensor_desc &se) {
    if (se.type != MI355_SENSOR_PERSPECTIVE && se.type != MI355_SENSOR_REALISTICLENS)
        return fail(MI355_ERR_UNSUPPORTED, "sensor type not supported by this build (perspective, realisticlens)");
    if (se.type == MI355_SENSOR_REALISTICLENS) {
        if (!se.lens_count || !se.lens_data || se.lens_count > 64) return fail(MI355_ERR_INVALID, "realisticlens needs 1..64 interfaces");
        if (!(se.film_diagonal > 0.f)) return fail(MI355_ERR_INVALID, "realisticlens needs the physical film diagonal");
        for (uint32_t i = 0; i < se.lens_count; ++i) {
            const float *L = se.lens_data + MI_LENS_ROW * i;
            if (!(L[0] == 0.f || L[0] == 1.f || L[0] == 2.f || L[0] == 3.f) || !(L[2] > 0.f) || !(L[4] > 0.f) || !(L[5] > 0.f) || (L[0] == 0.f && L[1] == 0.f))
                return fail(MI355_ERR_INVALID, "malformed lens interface row");
        }
    } else {
        if (!(se.x_fov > 0.f && se.x_fov < 180.f)) return fail(MI355_ERR_INVALID, "The horizontal field of view must be in the range [0, 180]!");   // sensor.cpp:196-197
        if (!(se.near_clip > 0.f && se.far_clip > se.near_clip)) return fail(MI355_ERR_INVALID, "clip planes must satisfy 0 < near_clip < far_clip");
    }
    if (!se.crop_width || !se.crop_height || !se.film_width || !se.film_height) return fail(MI355_ERR_INVALID, "empty film");
    if (se.crop_offset_x + se.crop_width > se.film_width || se.crop_offset_y + se.crop_height > se.film_height) return fail(MI355_ERR_INVALID, "Invalid crop window specification!");   // film.cpp:77-84
    if (se.rfilter != MI355_RFILTER_BOX && se.rfilter != MI355_RFILTER_GAUSSIAN && se.rfilter != MI355_RFILTER_TENT)
        return fail(MI355_ERR_UNSUPPORTED, "reconstruction filter not supported (box, gaussian, tent)");
    const float radius = se.rfilter == MI355_RFILTER_BOX ? 0.5f : (se.rfilter == MI355_RFILTER_GAUSSIAN ? 4.f * se.rfilter_param : se.rfilter_param);
    if (se.rfilter != MI355_RFILTER_BOX && (!(radius > 0.f) || 2 * (int) ceilf(radius - .5f) + 1 > MI_MAX_TAPS))
        return fail(MI355_ERR_UNSUPPORTED, "filter radius > 2.5 pixels is outside this build's splat footprint");
    for (int k = 0; k < 12; ++k) if (!std::isfinite(se.to_world[k])) return fail(MI355_ERR_INVALID, "sensor to_world is not finite");
    // "Scale factors in the camera-to-world transformation are not allowed!" (perspective.cpp:144-145, :166-167); Transform::has_scale,
    // transform.h:325-337: M M^T must be the identity within 1e-3
    for (int i = 0; i < 3; ++i) for (int j = i; j < 3; ++j) {
        float sum = 0.f; for (int k = 0; k < 3; ++k) sum += se.to_world[4 * i + k] * se.to_world[4 * j + k];
        if (fabsf(sum - (i == j ? 1.f : 0.f)) > 1e-3f) return fail(MI355_ERR_INVALID, "Scale factors in the camera-to-world transformation are not allowed!");
    }
    return MI355_OK;
}
// PerspectiveCamera::update_camera_transforms (perspective.cpp:173-199) / the realisticlens constructor's film quantities: fill the slot's
// share of DScene from its description.  The only device work is the exit-pupil table of a realisticlens that samples it.
static int init_sensor_slot(mi355_scene *s, SensorSlot &slot, const mi355_sensor_desc &src) {
    int rc = validate_sensor(src); if (rc) return rc;
    slot.desc = src;
    DScene &ds = slot.snap; memset(&ds, 0, sizeof(ds));
    const mi355_sensor_desc &se = slot.desc;
    int32_t film[2] = { (int32_t) se.film_width, (int32_t) se.film_height }, crop[2] = { (int32_t) se.crop_width, (int32_t) se.crop_height },
            coff[2] = { (int32_t) se.crop_offset_x, (int32_t) se.crop_offset_y };
    float c2s[16];
    if (se.type == MI355_SENSOR_PERSPECTIVE) {
        host_perspective_projection(film, crop, coff, se.x_fov, se.near_clip, se.far_clip, c2s, ds.sample_to_camera);
        memcpy(ds.camera_to_sample, c2s, sizeof(ds.camera_to_sample));
        // m_image_rect / m_normalization, perspective.cpp:190-199: the corners of the sample square on the plane z = 1
        auto proj = [&](float x, float y, float *o) {
            float r[4]; for (int i = 0; i < 4; ++i) r[i] = fmaf(ds.sample_to_camera[i * 4 + 1], y, fmaf(ds.sample_to_camera[i * 4 + 0], x, ds.sample_to_camera[i * 4 + 3]));
            float iw = 1.f / r[3]; o[0] = r[0] * iw; o[1] = r[1] * iw; o[2] = r[2] * iw; };
        float pmin[3], pmax[3]; proj(0.f, 0.f, pmin); proj(1.f, 1.f, pmax);
        float ax = pmin[0] / pmin[2], ay = pmin[1] / pmin[2], bx = pmax[0] / pmax[2], by = pmax[1] / pmax[2];
        ds.image_rect[0] = std::min(ax, bx); ds.image_rect[1] = std::min(ay, by); ds.image_rect[2] = std::max(ax, bx); ds.image_rect[3] = std::max(ay, by);
        ds.normalization = 1.f / ((ds.image_rect[2] - ds.image_rect[0]) * (ds.image_rect[3] - ds.image_rect[1]));
        slot.lens.clear(); slot.desc.lens_data = nullptr; slot.desc.lens_count = 0;
    } else {
        slot.lens.assign(se.lens_data, se.lens_data + MI_LENS_ROW * (size_t) se.lens_count);
        slot.desc.lens_data = slot.lens.data();
        float aspect = (float) se.film_height / (float) se.film_width;       // Film::get_physical_size, film.h:193-198
        float x = sqrtf(se.film_diagonal * se.film_diagonal / (1.f + aspect * aspect));
        ds.film_px = x / 2; ds.film_py = aspect * x / 2; ds.lens_count = se.lens_count;
    }
    ds.sensor_type = se.type; ds.film_w = se.film_width; ds.film_h = se.film_height;
    memcpy(ds.cam_to_world, se.to_world, 48);
    ds.near_clip = se.near_clip; ds.far_clip = se.far_clip;
    ds.crop_w = se.crop_width; ds.crop_h = se.crop_height; ds.crop_x = se.crop_offset_x; ds.crop_y = se.crop_offset_y;
    {   // Film::sample_border: border_size = ceil(radius - 0.5 - 2 RayEps), rfilter.cpp:23
        const float radius = se.rfilter == MI355_RFILTER_BOX ? 0.5f : (se.rfilter == MI355_RFILTER_GAUSSIAN ? 4.f * se.rfilter_param : se.rfilter_param);
        ds.border = se.sample_border ? (uint32_t) std::max(0.f, ceilf(radius - .5f - 2.f * MI_RAY_EPS)) : 0u;
        ds.grid_w = ds.crop_w + 2u * ds.border; ds.grid_h = ds.crop_h + 2u * ds.border;
    }
    ds.rfilter = se.rfilter;
    if (se.rfilter == MI355_RFILTER_GAUSSIAN) { mi355_gaussian_coeffs(se.rfilter_param, ds.filter_coeff); ds.filter_radius = 4.f * se.rfilter_param; }
    else if (se.rfilter == MI355_RFILTER_TENT) ds.filter_radius = se.rfilter_param;
    else ds.filter_radius = .5f;
    ds.shutter_open_time = se.shutter_open_time; ds.film_diagonal = se.film_diagonal;
    ds.sample_exit_pupil = (se.type == MI355_SENSOR_REALISTICLENS && se.sample_exit_pupil) ? 1u : 0u;
    if (slot.d_lens) { HIP_TRY(hipFree(slot.d_lens)); slot.d_lens = nullptr; }
    if (slot.d_pupil) { HIP_TRY(hipFree(slot.d_pupil)); slot.d_pupil = nullptr; }
    slot.pupil.clear();
    if (se.type == MI355_SENSOR_REALISTICLENS) {
        if ((rc = upload(&slot.d_lens, slot.lens))) return rc;
        ds.lens = slot.d_lens;
    }
    if (ds.sample_exit_pupil) {                                             // compute_exit_pupil_bounds, realisticlens.cpp:1772-1806
        uint32_t *d_keys = nullptr, h_keys[PUPIL_SEGMENTS * 5];
        for (uint32_t i = 0; i < PUPIL_SEGMENTS; ++i) { h_keys[5 * i] = h_keys[5 * i + 1] = 0xffffffffu; h_keys[5 * i + 2] = h_keys[5 * i + 3] = h_keys[5 * i + 4] = 0u; }
        HIP_TRY(hipMalloc((void **) &d_keys, sizeof(h_keys)));
        HIP_TRY(hipMemcpy(d_keys, h_keys, sizeof(h_keys), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_exit_pupil, dim3(PUPIL_SEGMENTS * (PUPIL_RAYS / 256u)), dim3(256), 0, 0, slot.d_lens, se.lens_count, se.film_diagonal, d_keys);
        hipError_t le = hipGetLastError(); if (le == hipSuccess) le = hipMemcpy(h_keys, d_keys, sizeof(h_keys), hipMemcpyDeviceToHost);
        (void) hipFree(d_keys);
        HIP_TRY(le);
        const float rear_radius = slot.lens[2] * 1.5f, spacing = 4.f * rear_radius * sqrtf(2.f / (float) PUPIL_RAYS);
        slot.pupil.assign(PUPIL_SEGMENTS * 4, 0.f);
        for (uint32_t i = 0; i < PUPIL_SEGMENTS; ++i) {
            float mnx = -rear_radius, mny = -rear_radius, mxx = rear_radius, mxy = rear_radius;      // nothing transmitted: the whole square
            if (h_keys[5 * i + 4] != 0u) { mnx = key_float(h_keys[5 * i]); mny = key_float(h_keys[5 * i + 1]); mxx = key_float(h_keys[5 * i + 2]); mxy = key_float(h_keys[5 * i + 3]); }
            slot.pupil[4 * i] = mnx - spacing; slot.pupil[4 * i + 1] = mny - spacing; slot.pupil[4 * i + 2] = mxx + spacing; slot.pupil[4 * i + 3] = mxy + spacing;
        }
        if ((rc = upload(&slot.d_pupil, slot.pupil))) return rc;
        ds.pupil = slot.d_pupil;
    }
    return MI355_OK;
}
// Make sensor `index` the one the kernel argument describes (a handful of host-side copies; nothing on the device).
static int select_sensor(mi355_scene *s, uint32_t index) {
    if (index >= s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range");
    if (s->cur_sensor == index) return MI355_OK;
    copy_sensor_fields(s->ds, s->sensors[index].snap);
    s->ds.features = s->base_features | (s->sensors[index].desc.type == MI355_SENSOR_REALISTICLENS ? 4u : 0u);
    s->cur_sensor = index;
    return MI355_OK;
}

static int build_device_scene(mi355_scene *s, const mi355_scene_desc *d) {
    if (!d->n_sensors || !d->sensors) return fail(MI355_ERR_INVALID, "a scene needs at least one sensor");
    for (uint32_t i = 0; i < d->n_sensors; ++i) { int rc_s = validate_sensor(d->sensors[i]); if (rc_s) return rc_s; }
    DScene &ds = s->ds; memset(&ds, 0, sizeof(ds));
    s->spheres.clear();
    // textures
    for (uint32_t i = 0; i < d->n_textures; ++i) {
        const mi355_texture_desc &t = d->textures[i];
        if ((t.channels != 1 && t.channels != 3) || t.width < 2 || t.height < 2 || !t.data)
            return fail(MI355_ERR_INVALID, "bitmap texture must be >= 2x2 with 1 or 3 channels (bitmap.cpp:313-330)");
        DTexture o; memset(&o, 0, sizeof(o)); o.w = t.width; o.h = t.height; o.c = t.channels; o.filter = t.filter; o.wrap = t.wrap;
        o.grad_slot = -1; o.data_offset = s->texdata.size(); memcpy(o.to_uv, t.to_uv, sizeof(o.to_uv));
        s->texdata.insert(s->texdata.end(), t.data, t.data + (size_t) t.width * t.height * t.channels);
        s->textures.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_bsdfs; ++i) {
        const mi355_bsdf_desc &b = d->bsdfs[i];
        if (b.type != MI355_BSDF_DIFFUSE && b.type != MI355_BSDF_ROUGHCONDUCTOR && b.type != MI355_BSDF_DIELECTRIC && b.type != MI355_BSDF_CONDUCTOR)
            return fail(MI355_ERR_UNSUPPORTED, "bsdf type not supported by this build (diffuse, roughconductor, dielectric)");
        if (b.reflectance_texture >= (int32_t) d->n_textures) return fail(MI355_ERR_INVALID, "bsdf references a missing texture");
        if (b.type != MI355_BSDF_DIFFUSE && b.reflectance_texture >= 0) return fail(MI355_ERR_UNSUPPORTED, "textured specular reflectance is not supported");
        if (b.type == MI355_BSDF_DIELECTRIC && !(b.int_ior > 0.f && b.ext_ior > 0.f)) return fail(MI355_ERR_INVALID, "indices of refraction must be positive");
        DBsdf o; memset(&o, 0, sizeof(o)); o.type = b.type; memcpy(o.reflectance, b.reflectance, 12); o.texture = b.reflectance_texture; o.grad_slot = -1;
        o.alpha_u = b.alpha_u; o.alpha_v = b.alpha_v; o.distribution = b.distribution; o.sample_visible = b.sample_visible;
        memcpy(o.eta, b.eta, 12); memcpy(o.k, b.k, 12); memcpy(o.transmittance, b.transmittance, 12);
        o.rel_eta = b.type == MI355_BSDF_DIELECTRIC ? b.int_ior / b.ext_ior : 1.f;
        o.ext_ior = b.type == MI355_BSDF_DIELECTRIC ? b.ext_ior : 1.f; o.ior_grad_offset = -1;
        s->ext_iors.push_back(b.type == MI355_BSDF_DIELECTRIC ? b.ext_ior : 1.f);
        if (b.type != MI355_BSDF_DIFFUSE) ds.features |= 1u;
        s->bsdfs.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_emitters; ++i) {
        const mi355_emitter_desc &e = d->emitters[i];
        if (e.type != MI355_EMITTER_AREA && e.type != MI355_EMITTER_DIRECTIONALAREA) return fail(MI355_ERR_UNSUPPORTED, "emitter type not supported (area, directionalarea)");
        if (e.type != MI355_EMITTER_AREA) ds.features |= 2u;
        if (e.shape >= d->n_shapes) return fail(MI355_ERR_INVALID, "emitter references a missing shape");
        DEmitter o; memset(&o, 0, sizeof(o)); o.type = e.type; memcpy(o.radiance, e.radiance, 12); o.shape = e.shape; o.grad_slot = -1;
        s->emitters.push_back(o);
    }
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const mi355_shape_desc &sd = d->shapes[i];
        if (sd.bsdf >= d->n_bsdfs) return fail(MI355_ERR_INVALID, "shape references a missing bsdf");
        if (sd.emitter >= (int32_t) d->n_emitters) return fail(MI355_ERR_INVALID, "shape references a missing emitter");
        DShape o; memset(&o, 0, sizeof(o)); o.type = sd.type; o.bsdf = sd.bsdf; o.emitter = sd.emitter; o.bvh_root = -1; o.grad_offset = 0xffffffffu;
        s->vcount.push_back(sd.type == MI355_SHAPE_MESH ? sd.vertex_count : 0u);
        // DirectionalArea::eval is zero and its pdf_direction is zero (directionalarea.cpp:126-136,165-168): a path that HITS such a shape
        // sees a plain surface, so the shape record carries no emitter; the emitter table (NEE pick, ptracer) still lists it
        if (sd.emitter >= 0 && d->emitters[sd.emitter].type == MI355_EMITTER_DIRECTIONALAREA) o.emitter = -1;
        o.flags = sd.flip_normals ? DSHAPE_FLIP : 0u;
        memcpy(o.to_world, sd.to_world, 48); memcpy(o.to_object, sd.to_object, 48);
        if (sd.type == MI355_SHAPE_RECTANGLE) {                             // Rectangle::update, rectangle.cpp:115-126
            h3 fs = hxf_vector(o.to_world, H3(2.f, 0.f, 0.f)), ft = hxf_vector(o.to_world, H3(0.f, 2.f, 0.f));
            h3 n = hxf_normal(o.to_object, H3(0.f, 0.f, 1.f)); n = hscale(n, 1.0f / sqrtf(hdot(n, n)));
            o.frame_s[0] = fs.x; o.frame_s[1] = fs.y; o.frame_s[2] = fs.z; o.frame_t[0] = ft.x; o.frame_t[1] = ft.y; o.frame_t[2] = ft.z;
            o.frame_n[0] = n.x; o.frame_n[1] = n.y; o.frame_n[2] = n.z;
            h3 cr = hcross(fs, ft); o.inv_area = 1.0f / sqrtf(hdot(cr, cr));
            DPrim pr; memset(&pr, 0, sizeof(pr)); memcpy(pr.m, o.to_object, 48);   // rectangle.cpp:457-465: local = to_object * ray
            pr.lo = -1.f; pr.smax = 2.f; pr.shape = i; pr.fin = 0xffffffffu;
            s->prims.push_back(pr);
        } else if (sd.type == MI355_SHAPE_MESH) {
            if (!sd.vertex_positions || !sd.faces || !sd.face_count || !sd.vertex_count) return fail(MI355_ERR_INVALID, "mesh without vertices/faces");
            o.tri_offset = (uint32_t) (s->faces.size() / 3); o.tri_count = sd.face_count; o.vtx_offset = (uint32_t) (s->positions.size() / 3);
            if (sd.vertex_normals) o.flags |= DSHAPE_HAS_NRM;
            if (sd.vertex_texcoords) o.flags |= DSHAPE_HAS_UV;
            s->positions.insert(s->positions.end(), sd.vertex_positions, sd.vertex_positions + 3 * (size_t) sd.vertex_count);
            if (sd.vertex_normals) s->normals.insert(s->normals.end(), sd.vertex_normals, sd.vertex_normals + 3 * (size_t) sd.vertex_count);
            else s->normals.insert(s->normals.end(), 3 * (size_t) sd.vertex_count, 0.f);
            if (sd.vertex_texcoords) s->texcoords.insert(s->texcoords.end(), sd.vertex_texcoords, sd.vertex_texcoords + 2 * (size_t) sd.vertex_count);
            else s->texcoords.insert(s->texcoords.end(), 2 * (size_t) sd.vertex_count, 0.f);
            for (int k = 0; k < 3; ++k) { o.bbox_min[k] = INFINITY; o.bbox_max[k] = -INFINITY; }
            float area = 0.f;                                               // Mesh::build_pmf: float prefix sums, surface area = the last one
            o.cdf_offset = (uint32_t) s->area_cdf.size();
            for (uint32_t f = 0; f < sd.face_count; ++f) {
                const uint32_t *fi = sd.faces + 3 * (size_t) f;
                for (int k = 0; k < 3; ++k) if (fi[k] >= sd.vertex_count) return fail(MI355_ERR_INVALID, "face index out of range");
                const float *a = sd.vertex_positions + 3 * (size_t) fi[0], *b = sd.vertex_positions + 3 * (size_t) fi[1], *c = sd.vertex_positions + 3 * (size_t) fi[2];
                for (int k = 0; k < 3; ++k) {
                    o.bbox_min[k] = std::min(o.bbox_min[k], std::min(a[k], std::min(b[k], c[k]))); o.bbox_max[k] = std::max(o.bbox_max[k], std::max(a[k], std::max(b[k], c[k])));
                }
                h3 cr = hcross(hsub(H3(b[0], b[1], b[2]), H3(a[0], a[1], a[2])), hsub(H3(c[0], c[1], c[2]), H3(a[0], a[1], a[2])));
                area += 0.5f * sqrtf(hdot(cr, cr));
                s->area_cdf.push_back(area);
                s->faces.push_back(fi[0]); s->faces.push_back(fi[1]); s->faces.push_back(fi[2]);
            }
            o.inv_area = 1.f / area;
            size_t fins_before = s->fins.size();
            if (try_make_box(s, i, sd)) { }
            else if (sd.face_count <= MI_FLAT_MAX) { s->fins.resize(fins_before); flatten_mesh(s, i, sd); }
            else { s->bvh_shapes.push_back(i); o.bvh_root = 0; ds.features |= 8u; }   // its triangles go into the scene BVH (build_scene_bvh, after this loop)
            s->n_faces_total += sd.face_count;
        } else if (sd.type == MI355_SHAPE_SPHERE) {                         // Sphere::update, sphere.cpp:150-172
            h3 c = H3(o.to_world[3], o.to_world[7], o.to_world[11]);
            h3 ex = hxf_vector(o.to_world, H3(1.f, 0.f, 0.f)); float radius = sqrtf(hdot(ex, ex));
            if (!(radius > 0.f)) return fail(MI355_ERR_INVALID, "sphere radius must be positive");
            o.frame_s[0] = c.x; o.frame_s[1] = c.y; o.frame_s[2] = c.z; o.frame_t[0] = radius;   // DShape: frame_s = centre, frame_t[0] = radius
            o.inv_area = 1.0f / (4.f * MI_PI * (radius * radius));
            DSphere sp; memset(&sp, 0, sizeof(sp)); sp.center[0] = c.x; sp.center[1] = c.y; sp.center[2] = c.z; sp.radius = radius; sp.shape = i;
            s->spheres.push_back(sp); ds.features |= 2u;
        } else if (sd.type == MI355_SHAPE_DISK) {                           // Disk::update, disk.cpp:113-127; surface_area :163-167
            h3 du = hxf_vector(o.to_world, H3(1.f, 0.f, 0.f)), dv = hxf_vector(o.to_world, H3(0.f, 1.f, 0.f));
            float lu = sqrtf(hdot(du, du)), lv = sqrtf(hdot(dv, dv));
            if (!(lu > 0.f) || !(lv > 0.f)) return fail(MI355_ERR_INVALID, "disk transform is singular");
            h3 fs = hscale(du, 1.f / lu), ft = hscale(dv, 1.f / lv);
            h3 n = hxf_normal(o.to_object, H3(0.f, 0.f, 1.f)); n = hscale(n, 1.0f / sqrtf(hdot(n, n)));
            o.frame_s[0] = fs.x; o.frame_s[1] = fs.y; o.frame_s[2] = fs.z; o.frame_t[0] = ft.x; o.frame_t[1] = ft.y; o.frame_t[2] = ft.z;
            o.frame_n[0] = n.x; o.frame_n[1] = n.y; o.frame_n[2] = n.z;
            float dd = hdot(hscale(ft, lv), fs), hgt = sqrtf(lv * lv - dd * dd);
            o.inv_area = 1.0f / (MI_PI * lu * hgt);
            DPrim pr; memset(&pr, 0, sizeof(pr)); memcpy(pr.m, o.to_object, 48);   // disk.cpp:392-416: the rectangle's plane test + x^2 + y^2 <= 1
            pr.lo = -1.f; pr.smax = 2.f; pr.shape = i; pr.fin = MI_DISK_FIN;
            s->prims.push_back(pr); ds.features |= 2u;
        } else return fail(MI355_ERR_UNSUPPORTED, "shape type not supported by this build (rectangle, mesh, sphere, disk)");
        s->shapes.push_back(o);
    }
    // gradient layout: constant reflectances, textures, emitter radiances (mi355_scene_param_layout)
    uint64_t off = 0;
    for (uint32_t i = 0; i < s->bsdfs.size(); ++i) if (s->bsdfs[i].texture < 0 && s->bsdfs[i].type == MI355_BSDF_DIFFUSE) {
        mi355_param_info pi = { MI355_PARAM_BSDF_REFLECTANCE, i, off, 3 }; s->layout.push_back(pi);
        s->bsdfs[i].grad_slot = (int32_t) s->small_map.size();
        for (int k = 0; k < 3; ++k) s->small_map.push_back((uint32_t) (off + k));
        off += 3;
    }
    for (uint32_t i = 0; i < s->textures.size(); ++i) {
        uint64_t sz = (uint64_t) s->textures[i].w * s->textures[i].h * s->textures[i].c;
        mi355_param_info pi = { MI355_PARAM_TEXTURE_DATA, i, off, sz }; s->layout.push_back(pi);
        s->textures[i].grad_slot = (int32_t) i; s->textures[i].grad_offset = off; off += sz;
    }
    for (uint32_t i = 0; i < s->emitters.size(); ++i) {
        mi355_param_info pi = { MI355_PARAM_EMITTER_RADIANCE, i, off, 3 }; s->layout.push_back(pi);
        s->emitters[i].grad_slot = (int32_t) s->small_map.size();
        for (int k = 0; k < 3; ++k) s->small_map.push_back((uint32_t) (off + k));
        off += 3;
    }
    for (uint32_t i = 0; i < s->shapes.size(); ++i) if (s->shapes[i].type == MI355_SHAPE_MESH) {      // `vertex_positions`, mesh.cpp:1602-1611
        mi355_param_info pi = { MI355_PARAM_VERTEX_POSITIONS, i, off, 3ull * s->vcount[i] }; s->layout.push_back(pi);
        if (off <= 0xfffffff0ull) s->shapes[i].grad_offset = (uint32_t) off;
        off += 3ull * s->vcount[i];
    }
    for (uint32_t i = 0; i < s->bsdfs.size(); ++i) if (s->bsdfs[i].type == MI355_BSDF_DIELECTRIC) {   // int_ior, ext_ior: dielectric.cpp:236-239
        mi355_param_info pi = { MI355_PARAM_BSDF_IOR, i, off, 2 }; s->layout.push_back(pi);
        if (off <= 0x7ffffff0ull) s->bsdfs[i].ior_grad_offset = (int32_t) off;
        off += 2;
    }
    s->grad_total = off;
    if (off > 0x7fffffffull) return fail(MI355_ERR_UNSUPPORTED, "more than 2^31 differentiable floats");
    {   // vertex -> incident (face, corner) lists for recompute_vertex_normals (meshes that carry vertex normals)
        const size_t nv_total = s->positions.size() / 3;
        std::vector<uint32_t> deg(nv_total + 1, 0u);
        for (uint32_t i = 0; i < s->shapes.size(); ++i) {
            const DShape &sh = s->shapes[i];
            if (sh.type != MI355_SHAPE_MESH || !(sh.flags & DSHAPE_HAS_NRM)) continue;
            for (uint32_t f = 0; f < sh.tri_count; ++f) for (int c = 0; c < 3; ++c) deg[sh.vtx_offset + s->faces[3 * (size_t) (sh.tri_offset + f) + c] + 1]++;
        }
        for (size_t v = 0; v < nv_total; ++v) deg[v + 1] += deg[v];
        s->vadj_off = deg; s->vadj.assign(deg[nv_total], 0u);
        std::vector<uint32_t> cur(deg.begin(), deg.end() - 1);
        for (uint32_t i = 0; i < s->shapes.size(); ++i) {
            const DShape &sh = s->shapes[i];
            if (sh.type != MI355_SHAPE_MESH || !(sh.flags & DSHAPE_HAS_NRM)) continue;
            for (uint32_t f = 0; f < sh.tri_count; ++f) for (uint32_t c = 0; c < 3; ++c) s->vadj[cur[sh.vtx_offset + s->faces[3 * (size_t) (sh.tri_offset + f) + c]]++] = f * 4u + c;
        }
    }
    { int rc_bvh = build_scene_bvh(s); if (rc_bvh) return rc_bvh; }       // the scene BVH over the large meshes, its leaf-ordered triangles, the refit lists
    s->dirty.assign(s->shapes.size(), 0);
    if (s->small_map.size() > MAX_SMALL_GRAD) return fail(MI355_ERR_UNSUPPORTED, "too many constant differentiable parameters for the LDS reducer");
    ds.n_shapes = (uint32_t) s->shapes.size(); ds.n_bsdfs = (uint32_t) s->bsdfs.size(); ds.n_emitters = (uint32_t) s->emitters.size();
    ds.n_textures = (uint32_t) s->textures.size();
    ds.inv_emitter_count = ds.n_emitters ? 1.f / (float) ds.n_emitters : 0.f;
    // upload
    int rc;
    if ((rc = upload(&s->d_shapes, s->shapes))) return rc;
    if ((rc = upload(&s->d_tris, s->tris))) return rc;
    if ((rc = upload(&s->d_faces, s->faces))) return rc;
    if ((rc = upload(&s->d_tri_face, s->tri_face))) return rc;
    if ((rc = upload(&s->d_tri_shape, s->tri_shape))) return rc;
    if ((rc = upload(&s->d_positions, s->positions))) return rc;
    if ((rc = upload(&s->d_normals, s->normals))) return rc;
    if ((rc = upload(&s->d_texcoords, s->texcoords))) return rc;
    if ((rc = upload(&s->d_bvh, s->bvh))) return rc;
    {   std::vector<DBvhQ> bq(s->bvh.size());                               // the quantised copy the kernels traverse
        for (size_t k = 0; k < s->bvh.size(); ++k) bvhq_encode(s->bvh[k], bq[k]);
        if ((rc = upload(&s->d_bvhq, bq))) return rc; }
    if ((rc = upload(&s->d_prims, prims_for_device(s->prims)))) return rc;
    {   std::vector<DPrimFin> padded(s->fins); padded.resize(std::max<size_t>(padded.size(), MI_LT_FINS));     // lt_fill copies MI_LT_FINS entries whatever the count
        if ((rc = upload(&s->d_fins, padded))) return rc; }
    if ((rc = upload(&s->d_boxes, s->boxes))) return rc;
    if ((rc = upload(&s->d_spheres, s->spheres))) return rc;
    if ((rc = upload(&s->d_area_cdf, s->area_cdf))) return rc;
    ds.area_cdf = s->d_area_cdf;
    if ((rc = upload(&s->d_bvh_shapes, s->bvh_shapes))) return rc;
    if ((rc = upload(&s->d_vadj_off, s->vadj_off))) return rc;
    if ((rc = upload(&s->d_vadj, s->vadj))) return rc;
    if ((rc = upload(&s->d_bvh_levels, s->bvh_levels))) return rc;
    {
        const size_t nv_total = std::max<size_t>(s->positions.size() / 3, 1);
        HIP_TRY(hipMalloc((void **) &s->d_nrm_len, nv_total * sizeof(float))); HIP_TRY(hipMalloc((void **) &s->d_nrm_adj, nv_total * 3 * sizeof(float)));
        HIP_TRY(hipMemset(s->d_nrm_len, 0, nv_total * sizeof(float)));
    }
    s->cap_prims = std::max<size_t>(s->prims.size(), 1); s->cap_fins = std::max<size_t>(s->fins.size(), MI_LT_FINS); s->cap_boxes = std::max<size_t>(s->boxes.size(), 1);
    if ((rc = upload(&s->d_bsdfs, s->bsdfs))) return rc;
    if ((rc = upload(&s->d_emitters, s->emitters))) return rc;
    if ((rc = upload(&s->d_textures, s->textures))) return rc;
    if ((rc = upload(&s->d_texdata, s->texdata))) return rc;
    if ((rc = upload(&s->d_small_map, s->small_map))) return rc;
    HIP_TRY(hipMalloc((void **) &s->d_counters, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->d_counters, 0, 8 * sizeof(unsigned long long)));
    ds.shapes = s->d_shapes; ds.tris = s->d_tris; ds.tri_face = s->d_tri_face; ds.tri_shape = s->d_tri_shape; ds.faces = s->d_faces; ds.positions = s->d_positions; ds.normals = s->d_normals;
    ds.texcoords = s->d_texcoords; ds.bvh = s->d_bvhq; ds.prims = s->d_prims; ds.fins = s->d_fins; ds.bvh_shapes = s->d_bvh_shapes; ds.boxes = s->d_boxes; ds.n_boxes = (uint32_t) s->boxes.size();
    ds.spheres = s->d_spheres; ds.n_spheres = (uint32_t) s->spheres.size();
    ds.n_prims = (uint32_t) s->prims.size(); ds.n_bvh_shapes = (uint32_t) s->bvh_shapes.size(); ds.bsdfs = s->d_bsdfs; ds.emitters = s->d_emitters; ds.textures = s->d_textures;
    ds.texdata = s->d_texdata;
    // record tables larger than the LDS staging area of the BASIC kernels (mi355_device.h, MI_LT_*): the general kernels read them from memory
    if (MI_LDS_TABLES && (s->shapes.size() > MI_LT_SHAPES || s->bsdfs.size() > MI_LT_BSDFS || s->emitters.size() > MI_LT_EMITTERS || s->fins.size() > MI_LT_FINS)) ds.features |= 16u;
    // the sensors: each keeps its own share of the kernel argument; sensor 0 is selected
    s->base_features = ds.features;
    s->sensors.resize(d->n_sensors);
    for (uint32_t i = 0; i < d->n_sensors; ++i) if ((rc = init_sensor_slot(s, s->sensors[i], d->sensors[i]))) return rc;
    return select_sensor(s, 0);
}

extern "C" int mi355_scene_create(const mi355_scene_desc *desc, int device, mi355_scene **out) {
    Phase ph(PHASE_INIT_SCENE);
    if (!desc || !out) return fail(MI355_ERR_INVALID, "null argument");
    if (desc->abi_version != MI355_ABI_VERSION) return fail(MI355_ERR_INVALID, "ABI version mismatch");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
        return fail(MI355_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
    if (device < 0 || device >= n_dev) return fail(MI355_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    mi355_scene *s = new mi355_scene(); s->device = device;
    int rc = build_device_scene(s, desc);
    if (rc != MI355_OK) { std::string keep = g_err; mi355_scene_destroy(s); g_err = keep; return rc; }
    *out = s; return MI355_OK;
}
extern "C" int mi355_scene_destroy(mi355_scene *s) {
    if (!s) return MI355_OK;
    (void) hipSetDevice(s->device);
    void *bufs[] = { s->d_shapes, s->d_tris, s->d_faces, s->d_positions, s->d_normals, s->d_texcoords, s->d_bvh, s->d_bvhq, s->d_bsdfs, s->d_emitters,
                     s->d_textures, s->d_texdata, s->d_small_map, s->d_counters, s->d_tri_face, s->d_prims, s->d_fins, s->d_bvh_shapes, s->d_boxes, s->d_gw, s->d_spheres, s->d_redo, s->d_area_cdf,
                     s->d_rec_a, s->d_srt_a, s->d_rec_b, s->d_srt_b, s->d_rec_c, s->d_srt_c, s->d_rec_d, s->d_srt_d, s->d_tile_u32,
                     s->d_vadj_off, s->d_vadj, s->d_nrm_len, s->d_nrm_adj, s->d_bvh_levels, s->d_tape, s->d_ntan, s->d_tri_shape, s->d_pt_keys, s->d_pt_order, s->d_pt_bins };
    for (void *b : bufs) if (b) (void) hipFree(b);
    for (SensorSlot &sl : s->sensors) { if (sl.d_lens) (void) hipFree(sl.d_lens); if (sl.d_pupil) (void) hipFree(sl.d_pupil); }
    if (s->ev_created) for (int i = 0; i < MAX_EVENTS; ++i) { (void) hipEventDestroy(s->ev[i][0]); (void) hipEventDestroy(s->ev[i][1]); }
    delete s; return MI355_OK;
}

// ---- geometry updates: Mesh::parameters_changed (mesh.cpp:96-163) + Scene::parameters_changed (scene.cpp:481-529) ------------------------
static mi355_shape_desc host_mesh_desc(const mi355_scene *s, uint32_t i) {   // a mesh's buffers as the host mirror holds them
    mi355_shape_desc sd; memset(&sd, 0, sizeof(sd));
    const DShape &sh = s->shapes[i];
    sd.type = MI355_SHAPE_MESH; sd.vertex_count = s->vcount[i]; sd.face_count = sh.tri_count;
    sd.vertex_positions = s->positions.data() + 3 * (size_t) sh.vtx_offset; sd.faces = s->faces.data() + 3 * (size_t) sh.tri_offset;
    return sd;
}
static void push_plane_prim(mi355_scene *s, uint32_t i) {                   // rectangle.cpp:457-465 / disk.cpp:392-416: local = to_object * ray
    const DShape &o = s->shapes[i];
    DPrim pr; memset(&pr, 0, sizeof(pr)); memcpy(pr.m, o.to_object, 48);
    pr.lo = -1.f; pr.smax = 2.f; pr.shape = i; pr.fin = o.type == MI355_SHAPE_DISK ? MI_DISK_FIN : 0xffffffffu;
    s->prims.push_back(pr);
}
template <typename T> static int reupload(T **dst, size_t *cap, const std::vector<T> &src) {
    if (src.size() > *cap) { if (*dst) HIP_TRY(hipFree(*dst)); *dst = nullptr; *cap = 0; HIP_TRY(hipMalloc((void **) dst, src.size() * sizeof(T))); *cap = src.size(); }
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return MI355_OK;
}
// the flat primitive tables (rectangles, disks, meshes of <= MI_FLAT_MAX triangles, boxes) from the host mirror, in shape order
static int rebuild_flat_tables(mi355_scene *s) {
    s->prims.clear(); s->fins.clear(); s->boxes.clear();
    for (uint32_t i = 0; i < s->shapes.size(); ++i) {
        const DShape &sh = s->shapes[i];
        if (sh.type == MI355_SHAPE_RECTANGLE || sh.type == MI355_SHAPE_DISK) push_plane_prim(s, i);
        else if (sh.type == MI355_SHAPE_MESH && sh.bvh_root < 0) {
            const mi355_shape_desc sd = host_mesh_desc(s, i);
            const size_t fins_before = s->fins.size();
            if (!try_make_box(s, i, sd)) { s->fins.resize(fins_before); flatten_mesh(s, i, sd); }
        }
    }
    int rc;
    if ((rc = reupload(&s->d_prims, &s->cap_prims, prims_for_device(s->prims)))) return rc;
    if ((rc = reupload(&s->d_fins, &s->cap_fins, s->fins))) return rc;
    if ((rc = reupload(&s->d_boxes, &s->cap_boxes, s->boxes))) return rc;
    s->ds.prims = s->d_prims; s->ds.fins = s->d_fins; s->ds.boxes = s->d_boxes;
    s->ds.n_prims = (uint32_t) s->prims.size(); s->ds.n_boxes = (uint32_t) s->boxes.size();
    if (MI_LDS_TABLES && s->fins.size() > MI_LT_FINS && !(s->base_features & 16u)) { s->base_features |= 16u; s->ds.features |= 16u; }   // (a moved mesh may pair its triangles differently)
    return MI355_OK;
}
// Runs (once) before any call that traces rays: vertex normals, triangles, BVH boxes, flat tables, emitter tables of every mesh whose
// positions changed.  Work is issued on `st`, the stream of the call that needs it.
static int ensure_geometry(mi355_scene *s, hipStream_t st) {
    if (!s->geom_dirty) return MI355_OK;
    Phase ph(PHASE_PARAMS);                                                 // Mesh::parameters_changed + Scene::parameters_changed: normals, refit
    bool refit = false, reflatten = false, shapes_changed = false, cdf_changed = false;
    for (uint32_t i = 0; i < s->shapes.size(); ++i) {
        if (!s->dirty[i]) continue;
        DShape &sh = s->shapes[i];
        const uint32_t nv = s->vcount[i];
        if (sh.flags & DSHAPE_HAS_NRM) {                                    // recompute_vertex_normals, mesh.cpp:138-139
            hipLaunchKernelGGL(k_vertex_normals, dim3((nv + 255u) / 256u), dim3(256), 0, st, (const float *) s->d_positions, (const uint32_t *) s->d_faces, (const uint32_t *) s->d_vadj_off,
                               (const uint32_t *) s->d_vadj, sh.vtx_offset, sh.tri_offset, nv, s->d_normals, s->d_nrm_len, (const float *) nullptr, (float *) nullptr);
            HIP_TRY(hipGetLastError());
            if (!(sh.flags & DSHAPE_NRM_DEP)) { sh.flags |= DSHAPE_NRM_DEP; shapes_changed = true; }
        }
        if (sh.bvh_root >= 0) refit = true; else reflatten = true;
        bool sampled = sh.emitter >= 0; for (const DEmitter &e : s->emitters) sampled = sampled || e.shape == i;
        if (sampled || sh.bvh_root < 0) {   // recompute_bbox (mesh.cpp:415-428); build_pmf when the mesh is sampled (mesh.cpp:141-142, :430-470): float prefix sums of the face areas
            const float *P = s->positions.data() + 3 * (size_t) sh.vtx_offset;
            for (int k = 0; k < 3; ++k) { sh.bbox_min[k] = INFINITY; sh.bbox_max[k] = -INFINITY; }
            for (uint32_t v = 0; v < nv; ++v) for (int k = 0; k < 3; ++k) { sh.bbox_min[k] = std::min(sh.bbox_min[k], P[3 * v + k]); sh.bbox_max[k] = std::max(sh.bbox_max[k], P[3 * v + k]); }
            float area = 0.f;
            for (uint32_t f = 0; f < sh.tri_count; ++f) {
                const uint32_t *fi = s->faces.data() + 3 * (size_t) (sh.tri_offset + f);
                const float *a = P + 3 * (size_t) fi[0], *b = P + 3 * (size_t) fi[1], *c = P + 3 * (size_t) fi[2];
                h3 cr = hcross(hsub(H3(b[0], b[1], b[2]), H3(a[0], a[1], a[2])), hsub(H3(c[0], c[1], c[2]), H3(a[0], a[1], a[2])));
                area += 0.5f * sqrtf(hdot(cr, cr));
                s->area_cdf[sh.cdf_offset + f] = area;
            }
            sh.inv_area = 1.f / area; shapes_changed = true; cdf_changed = cdf_changed || sampled;
        }
        s->dirty[i] = 0;
    }
    if (refit) {                                                            // the leaf-ordered triangles, then the boxes bottom-up, one launch per level
        const uint32_t nt = (uint32_t) s->tris.size();
        hipLaunchKernelGGL(k_mesh_tris, dim3((nt + 255u) / 256u), dim3(256), 0, st, (const float *) s->d_positions, (const uint32_t *) s->d_faces, (const uint32_t *) s->d_tri_face,
                           (const uint32_t *) s->d_tri_shape, (const DShape *) s->d_shapes, nt, s->d_tris);
        for (size_t l = s->bvh_level_off.size() - 1; l-- > 0;) {
            const uint32_t b = s->bvh_level_off[l], n = s->bvh_level_off[l + 1] - b;
            if (n) hipLaunchKernelGGL(k_bvh_refit_level, dim3((n + 255u) / 256u), dim3(256), 0, st, s->d_bvh, s->d_bvhq, (const uint32_t *) (s->d_bvh_levels + b), n, (const DTri *) s->d_tris);
        }
    }
    HIP_TRY(hipGetLastError());
    // the blocking host copies below must not overtake kernels that still read the old tables; a pure device-side refit (BVH meshes that are
    // not sampled as emitters, normals already position-dependent) issues no copy and does not stall the host
    if (reflatten || shapes_changed || cdf_changed) HIP_TRY(hipStreamSynchronize(st));
    int rc;
    if (reflatten && (rc = rebuild_flat_tables(s))) return rc;
    if (shapes_changed) HIP_TRY(hipMemcpy(s->d_shapes, s->shapes.data(), s->shapes.size() * sizeof(DShape), hipMemcpyHostToDevice));
    if (cdf_changed && !s->area_cdf.empty()) HIP_TRY(hipMemcpy(s->d_area_cdf, s->area_cdf.data(), s->area_cdf.size() * sizeof(float), hipMemcpyHostToDevice));
    s->geom_dirty = false;
    return MI355_OK;
}
static int set_geometry_param(mi355_scene *s, uint32_t kind, uint32_t index, const float *values, uint64_t count, bool device_src, hipStream_t st) {
    HIP_TRY(hipSetDevice(s->device));
    if (kind == MI355_PARAM_VERTEX_POSITIONS) {
        if (index >= s->shapes.size() || s->shapes[index].type != MI355_SHAPE_MESH) return fail(MI355_ERR_INVALID, "vertex_positions: not a mesh");
        if (count != 3ull * s->vcount[index]) return fail(MI355_ERR_INVALID, "vertex_positions update must keep the vertex count (topology changes need a new scene)");
        const DShape &sh = s->shapes[index];
        float *h = s->positions.data() + 3 * (size_t) sh.vtx_offset, *d = s->d_positions + 3 * (size_t) sh.vtx_offset;
        // the host mirror is what the flat tables and the emitter tables are rebuilt from; a mesh in the scene BVH that is not sampled
        // is refitted entirely on the device and needs no host copy (no D2H transfer, no synchronisation)
        bool need_host = sh.bvh_root < 0 || sh.emitter >= 0; for (const DEmitter &e : s->emitters) need_host = need_host || e.shape == index;
        // values that reach the host are validated BEFORE anything is applied (a failed update leaves the scene as it was); a device-to-device
        // update of a BVH-only mesh is not read back -- its values are the caller's responsibility (a non-finite vertex gives its triangles
        // an empty box: they are never hit)
        std::vector<float> staged;
        const float *src_host = device_src ? nullptr : values;
        if (device_src && need_host) {
            staged.resize(count);
            HIP_TRY(hipMemcpyAsync(staged.data(), values, count * 4, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st));
            src_host = staged.data();
        }
        if (src_host) for (uint64_t k = 0; k < count; ++k) if (!std::isfinite(src_host[k])) return fail(MI355_ERR_INVALID, "vertex_positions contains a non-finite value");
        if (device_src) HIP_TRY(hipMemcpyAsync(d, values, count * 4, hipMemcpyDeviceToDevice, st));
        else HIP_TRY(hipMemcpy(d, values, count * 4, hipMemcpyHostToDevice));
        if (src_host) memcpy(h, src_host, count * 4);
        s->dirty[index] = 1; s->geom_dirty = true;
        return MI355_OK;
    }
    // MI355_PARAM_BSDF_IOR
    if (index >= s->bsdfs.size() || s->bsdfs[index].type != MI355_BSDF_DIELECTRIC || count != 2) return fail(MI355_ERR_INVALID, "bad IOR update (2 floats of a dielectric)");
    float v[2];
    if (device_src) { HIP_TRY(hipMemcpyAsync(v, values, 8, hipMemcpyDeviceToHost, st)); HIP_TRY(hipStreamSynchronize(st)); } else memcpy(v, values, 8);
    if (!(v[0] > 0.f && v[1] > 0.f)) return fail(MI355_ERR_INVALID, "indices of refraction must be positive");
    DBsdf &b = s->bsdfs[index];
    b.rel_eta = v[0] / v[1]; b.ext_ior = v[1]; s->ext_iors[index] = v[1];  // dielectric.cpp:222
    HIP_TRY(hipMemcpy(s->d_bsdfs + index, &b, sizeof(DBsdf), hipMemcpyHostToDevice));
    return MI355_OK;
}

static int param_target(mi355_scene *s, uint32_t kind, uint32_t index, uint64_t count, float **host, float **dev) {
    if (kind == MI355_PARAM_BSDF_REFLECTANCE) {
        if (index >= s->bsdfs.size() || count != 3) return fail(MI355_ERR_INVALID, "bad bsdf reflectance update");
        *host = s->bsdfs[index].reflectance; *dev = (float *) ((char *) (s->d_bsdfs + index) + offsetof(DBsdf, reflectance)); return MI355_OK;
    }
    if (kind == MI355_PARAM_EMITTER_RADIANCE) {
        if (index >= s->emitters.size() || count != 3) return fail(MI355_ERR_INVALID, "bad emitter radiance update");
        *host = s->emitters[index].radiance; *dev = (float *) ((char *) (s->d_emitters + index) + offsetof(DEmitter, radiance)); return MI355_OK;
    }
    if (kind == MI355_PARAM_TEXTURE_DATA) {
        if (index >= s->textures.size()) return fail(MI355_ERR_INVALID, "bad texture index");
        const DTexture &t = s->textures[index];
        if (count != (uint64_t) t.w * t.h * t.c) return fail(MI355_ERR_INVALID, "texture update must keep the tensor shape");
        *host = s->texdata.data() + t.data_offset; *dev = s->d_texdata + t.data_offset; return MI355_OK;
    }
    return fail(MI355_ERR_INVALID, "unknown parameter kind");
}
// PerspectiveCamera::parameters_changed (perspective.cpp:163-171): `to_world` / `x_fov` -> update_camera_transforms.  Host-side only.
static int reinit_sensor(mi355_scene *s, uint32_t index, const mi355_sensor_desc &nd_in) {
    mi355_sensor_desc nd = nd_in;
    std::vector<float> lens_copy;                                           // (the description may point into the slot it replaces)
    if (nd.type == MI355_SENSOR_REALISTICLENS && nd.lens_data && nd.lens_count && nd.lens_count <= 64) { lens_copy.assign(nd.lens_data, nd.lens_data + MI_LENS_ROW * (size_t) nd.lens_count); nd.lens_data = lens_copy.data(); }
    int rc = validate_sensor(nd); if (rc) return rc;                         // a refused update leaves the sensor as it was
    const bool device_work = nd.type == MI355_SENSOR_REALISTICLENS || (index < s->sensors.size() && (s->sensors[index].d_lens || s->sensors[index].d_pupil));
    if (device_work) HIP_TRY(hipSetDevice(s->device));                      // (a perspective camera is host-side state only: no runtime call at all)
    if (index == s->sensors.size()) s->sensors.emplace_back();
    if ((rc = init_sensor_slot(s, s->sensors[index], nd))) return rc;
    if (s->cur_sensor == index) s->cur_sensor = 0xffffffffu;                // re-select on the next call
    return MI355_OK;
}
static int set_sensor_param(mi355_scene *s, uint32_t kind, uint32_t index, const float *v, uint64_t count) {
    if (index >= s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range");
    mi355_sensor_desc nd = s->sensors[index].desc;
    if (kind == MI355_PARAM_SENSOR_TO_WORLD) {
        if (count != 12) return fail(MI355_ERR_INVALID, "sensor to_world is 12 floats (row-major 3x4)");
        memcpy(nd.to_world, v, 48);
    } else {
        if (count != 1) return fail(MI355_ERR_INVALID, "sensor x_fov is one float");
        if (nd.type != MI355_SENSOR_PERSPECTIVE) return fail(MI355_ERR_INVALID, "x_fov is a parameter of the perspective sensor");
        nd.x_fov = v[0];
    }
    return reinit_sensor(s, index, nd);
}
extern "C" int mi355_scene_set_sensor(mi355_scene *s, uint32_t index, const mi355_sensor_desc *sensor) {
    Phase ph(PHASE_PARAMS);
    if (!s || !sensor) return fail(MI355_ERR_INVALID, "null argument");
    if (index > s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range (index == count appends)");
    return reinit_sensor(s, index, *sensor);
}
extern "C" int mi355_scene_sensor_count(const mi355_scene *s) {
    if (!s) return fail(MI355_ERR_INVALID, "null scene");
    return (int) s->sensors.size();
}
extern "C" int mi355_scene_set_param(mi355_scene *s, uint32_t kind, uint32_t index, const float *values, uint64_t count) {
    if (!s || !values) return fail(MI355_ERR_INVALID, "null argument");
    if (kind == MI355_PARAM_SENSOR_TO_WORLD || kind == MI355_PARAM_SENSOR_X_FOV) return set_sensor_param(s, kind, index, values, count);
    if (kind == MI355_PARAM_VERTEX_POSITIONS || kind == MI355_PARAM_BSDF_IOR) return set_geometry_param(s, kind, index, values, count, false, nullptr);
    float *h, *d; int rc = param_target(s, kind, index, count, &h, &d); if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    memcpy(h, values, count * 4);
    HIP_TRY(hipMemcpy(d, values, count * 4, hipMemcpyHostToDevice));
    return MI355_OK;
}
extern "C" int mi355_scene_set_param_device(mi355_scene *s, uint32_t kind, uint32_t index, const float *d_values, uint64_t count, void *stream) {
    Phase ph(PHASE_PARAMS);
    if (!s || !d_values) return fail(MI355_ERR_INVALID, "null argument");
    if (kind == MI355_PARAM_SENSOR_TO_WORLD || kind == MI355_PARAM_SENSOR_X_FOV) {      // a sensor is host-side state: fetch the few floats
        float v[12]; if (count > 12) return fail(MI355_ERR_INVALID, "bad sensor parameter size");
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipMemcpyAsync(v, d_values, count * 4, hipMemcpyDeviceToHost, (hipStream_t) stream)); HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
        return set_sensor_param(s, kind, index, v, count);
    }
    if (kind == MI355_PARAM_VERTEX_POSITIONS || kind == MI355_PARAM_BSDF_IOR) return set_geometry_param(s, kind, index, d_values, count, true, (hipStream_t) stream);
    float *h, *d; int rc = param_target(s, kind, index, count, &h, &d); if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(d, d_values, count * 4, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    HIP_TRY(hipMemcpyAsync(h, d_values, count * 4, hipMemcpyDeviceToHost, (hipStream_t) stream));   // keep the host mirror coherent
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    return MI355_OK;
}
extern "C" int mi355_scene_param_enable_grad(mi355_scene *s, uint32_t kind, uint32_t index, int enable) {
    if (!s) return fail(MI355_ERR_INVALID, "null scene");
    uint64_t off = ~0ull;
    for (const mi355_param_info &pi : s->layout) if (pi.kind == kind && pi.index == index) off = pi.offset;
    if (off == ~0ull || (kind != MI355_PARAM_VERTEX_POSITIONS && kind != MI355_PARAM_BSDF_IOR)) return fail(MI355_ERR_INVALID, "no such geometry parameter (vertex_positions of a mesh, IOR pair of a dielectric)");
    HIP_TRY(hipSetDevice(s->device));
    if (kind == MI355_PARAM_VERTEX_POSITIONS) {
        const uint32_t v = enable ? (off <= 0xfffffff0ull ? (uint32_t) off : 0xffffffffu) : 0xffffffffu;
        if (s->shapes[index].grad_offset == v) return MI355_OK;
        s->shapes[index].grad_offset = v;
        HIP_TRY(hipMemcpy((char *) (s->d_shapes + index) + offsetof(DShape, grad_offset), &v, 4, hipMemcpyHostToDevice));
    } else {
        const int32_t v = enable ? (off <= 0x7ffffff0ull ? (int32_t) off : -1) : -1;
        if (s->bsdfs[index].ior_grad_offset == v) return MI355_OK;
        s->bsdfs[index].ior_grad_offset = v;
        HIP_TRY(hipMemcpy((char *) (s->d_bsdfs + index) + offsetof(DBsdf, ior_grad_offset), &v, 4, hipMemcpyHostToDevice));
    }
    return MI355_OK;
}
extern "C" int mi355_scene_param_layout(const mi355_scene *s, mi355_param_info *out, uint32_t capacity, uint64_t *total_floats) {
    if (!s) return fail(MI355_ERR_INVALID, "null scene");
    for (uint32_t i = 0; i < s->layout.size() && i < capacity; ++i) if (out) out[i] = s->layout[i];
    if (total_floats) *total_floats = s->grad_total;
    return (int) s->layout.size();
}

static int make_args(mi355_scene *s, const mi355_render_params *p, RenderArgs *a) {
    if (!s || !p) return fail(MI355_ERR_INVALID, "null argument");
    { int rc_s = select_sensor(s, p->sensor); if (rc_s) return rc_s; }      // the `Sensor *` argument of Integrator::render / render_backward
    // lanes cover the crop window, except for the particle tracer: the whole film (integrator.cpp:575-577)
    uint32_t npix = p->integrator == MI355_INTEGRATOR_PTRACER ? s->ds.film_w * s->ds.film_h : s->ds.grid_w * s->ds.grid_h;
    if (p->spp == 0) return fail(MI355_ERR_INVALID, "spp must be > 0");
    // samples_per_pass (integrator.cpp:565-573, :700-726): spp / passes sampler streams per pixel, `passes` samples from each
    const uint32_t passes = p->passes ? p->passes : 1u;
    if (passes > 1u && p->integrator == MI355_INTEGRATOR_PRB) return fail(MI355_ERR_UNSUPPORTED, "passes > 1 (samples_per_pass): prb.py renders in one pass; built for `path` and `ptracer`");
    if (passes > 1u && p->integrator == MI355_INTEGRATOR_PATH && s->ds.sensor_type != MI355_SENSOR_PERSPECTIVE)
        return fail(MI355_ERR_UNSUPPORTED, "passes > 1 with the realisticlens sensor: a ray blocked in the barrel is still walked by the reference and its draws cannot be reproduced");
    if (p->spp % passes) return fail(MI355_ERR_INVALID, "spp must be a multiple of passes");
    const uint32_t streams = p->spp / passes;
    if ((uint64_t) npix * streams > 0xffffffffull)
        return fail(MI355_ERR_INVALID, "W*H*spp exceeds 2^32 lanes; split into passes (integrator.cpp:247-265)");
    a->passes = passes;
    a->seed = p->seed; a->spp_total = streams; a->max_depth = (uint32_t) p->max_depth; a->rr_depth = (uint32_t) p->rr_depth; a->flags = p->flags;
    a->pixel_begin = p->pixel_begin; a->pixel_count = p->pixel_count ? p->pixel_count : npix - p->pixel_begin;
    a->s_begin = p->spp_begin; a->s_count = p->spp_count ? p->spp_count : streams - p->spp_begin;
    if (p->pixel_begin > npix || a->pixel_count > npix - p->pixel_begin) return fail(MI355_ERR_INVALID, "pixel shard out of range");
    if (p->spp_begin > streams || a->s_count > streams - p->spp_begin) return fail(MI355_ERR_INVALID, "spp shard out of range");
    a->tile = p->pixel_tile; a->tile_stride = p->pixel_tile_stride;
    if (a->tile != 0u && a->pixel_count != 0u) {
        if (a->tile_stride < a->tile) return fail(MI355_ERR_INVALID, "pixel_tile_stride must be >= pixel_tile");
        uint64_t q = a->pixel_count - 1u, last = (uint64_t) p->pixel_begin + (q / a->tile) * a->tile_stride + q % a->tile;
        if (last >= npix) return fail(MI355_ERR_INVALID, "tiled pixel shard out of range");
    }
    uint64_t n = (uint64_t) a->pixel_count * a->s_count;
    if (n > 0xffffffffull - BLOCK) return fail(MI355_ERR_INVALID, "shard too large");
    a->n_local = (uint32_t) n;
    // LDS splat needs every wave to be cut into whole pixels: s_count = 2^k >= 4, or a multiple of 64
    uint32_t sc = a->s_count; a->seg = 0;
    if (sc % 64 == 0) a->seg = 64; else if (sc >= 4 && (sc & (sc - 1)) == 0) a->seg = sc;
    a->lane_begin = 0;
    a->ppw = std::max(1u, 1024u / a->s_count);
    a->compact_mask = 0u; a->compact_adaptive = 0u;
    return MI355_OK;
}
static int ensure_events(mi355_scene *s) {
    if (s->ev_created) return MI355_OK;
    for (int i = 0; i < MAX_EVENTS; ++i) { HIP_TRY(hipEventCreate(&s->ev[i][0])); HIP_TRY(hipEventCreate(&s->ev[i][1])); }
    s->ev_created = true; return MI355_OK;
}
static int drain_events(mi355_scene *s) {
    for (int i = 0; i < s->n_ev; ++i) {
        HIP_TRY(hipEventSynchronize(s->ev[i][1])); float ms = 0.f; HIP_TRY(hipEventElapsedTime(&ms, s->ev[i][0], s->ev[i][1]));
        s->ms_accum += ms; s->launches_accum++;
    }
    s->n_ev = 0; return MI355_OK;
}
struct Timed {
    mi355_scene *s; hipStream_t st; bool on; int slot;
    int begin() {
        if (!on) return MI355_OK;
        int rc = ensure_events(s); if (rc) return rc;
        if (s->n_ev == MAX_EVENTS) { rc = drain_events(s); if (rc) return rc; }
        slot = s->n_ev; HIP_TRY(hipEventRecord(s->ev[slot][0], st)); return MI355_OK;
    }
    int end() { if (!on) return MI355_OK; HIP_TRY(hipEventRecord(s->ev[slot][1], st)); s->n_ev++; return MI355_OK; }
};

// Deferred, tile-sorted splat of the particle tracer (film_put_deferred): queue set-up and the four passes that empty it.  `dual`: the
// records of the forward-mode derivative film carry a second half (colour, film-position tangent) next to (position, colour tangent).
static int splat_queue_setup(mi355_scene *s, const mi355_render_params *p, const RenderArgs &a, hipStream_t st, bool dual, SplatQueue &q) {
    memset(&q, 0, sizeof(q));
    const uint64_t per_lane = (uint64_t) a.passes * (p->max_depth < 0 ? 64u : (uint32_t) p->max_depth);
    const uint32_t all_tiles = ((s->ds.crop_w + MI_SPLAT_TILE - 1) / MI_SPLAT_TILE) * ((s->ds.crop_h + MI_SPLAT_TILE - 1) / MI_SPLAT_TILE);
    if ((p->flags & MI355_FLAG_DIRECT_SPLAT) || (uint64_t) a.n_local * per_lane >= 0xffffffffull || all_tiles > SPLAT_MAX_TILES) return MI355_OK;
    // The queue holds every connection of the launch when max_depth bounds their number; beyond its capacity a record falls back to direct atomics.
    uint64_t want = std::min<uint64_t>((uint64_t) a.n_local * a.passes * (p->max_depth < 0 ? 8u : std::min<uint32_t>((uint32_t) p->max_depth, 8u)), s->splat_cap_override ? s->splat_cap_override : MAX_SPLAT_RECORDS);
    const uint64_t seg_cap = std::max<uint64_t>((want + MI_SPLAT_SEGS - 1u) / MI_SPLAT_SEGS, 1u);   // MI_SPLAT_SEGS segments of equal capacity
    want = seg_cap * MI_SPLAT_SEGS;
    q.tiles_x = (s->ds.crop_w + MI_SPLAT_TILE - 1) / MI_SPLAT_TILE; q.tiles_y = (s->ds.crop_h + MI_SPLAT_TILE - 1) / MI_SPLAT_TILE;
    const uint32_t n_tiles = q.tiles_x * q.tiles_y;
    if (s->rec_capacity < want) {
        for (void *b : { (void *) s->d_rec_a, (void *) s->d_srt_a, (void *) s->d_rec_b, (void *) s->d_srt_b }) if (b) HIP_TRY(hipFree(b));
        s->d_rec_a = s->d_srt_a = nullptr; s->d_rec_b = s->d_srt_b = nullptr; s->rec_capacity = 0;
        HIP_TRY(hipMalloc((void **) &s->d_rec_a, want * sizeof(float4))); HIP_TRY(hipMalloc((void **) &s->d_srt_a, want * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &s->d_rec_b, want * sizeof(float))); HIP_TRY(hipMalloc((void **) &s->d_srt_b, want * sizeof(float)));
        s->rec_capacity = (uint32_t) want;
    }
    if (dual && s->rec2_capacity < want) {
        for (void *b : { (void *) s->d_rec_c, (void *) s->d_srt_c, (void *) s->d_rec_d, (void *) s->d_srt_d }) if (b) HIP_TRY(hipFree(b));
        s->d_rec_c = s->d_srt_c = nullptr; s->d_rec_d = s->d_srt_d = nullptr; s->rec2_capacity = 0;
        HIP_TRY(hipMalloc((void **) &s->d_rec_c, want * sizeof(float4))); HIP_TRY(hipMalloc((void **) &s->d_srt_c, want * sizeof(float4)));
        HIP_TRY(hipMalloc((void **) &s->d_rec_d, want * sizeof(float))); HIP_TRY(hipMalloc((void **) &s->d_srt_d, want * sizeof(float)));
        s->rec2_capacity = (uint32_t) want;
    }
    const size_t n_cnt = (size_t) MI_SPLAT_SEGS * MI_SPLAT_COUNTER_STRIDE;
    if (s->tile_capacity < n_tiles) {                                       // [segment counters | hist n | offsets n + 1 | cursor n]
        if (s->d_tile_u32) HIP_TRY(hipFree(s->d_tile_u32));
        s->d_tile_u32 = nullptr; s->tile_capacity = 0;
        HIP_TRY(hipMalloc((void **) &s->d_tile_u32, (n_cnt + 3 * (size_t) n_tiles + 1) * sizeof(uint32_t))); s->tile_capacity = n_tiles;
    }
    q.rec_a = s->d_rec_a; q.rec_b = s->d_rec_b; q.rec_c = dual ? s->d_rec_c : nullptr; q.rec_d = dual ? s->d_rec_d : nullptr;
    q.cap = (uint32_t) want; q.seg_cap = (uint32_t) seg_cap; q.count = s->d_tile_u32; q.hist = s->d_tile_u32 + n_cnt;
    HIP_TRY(hipMemsetAsync(s->d_tile_u32, 0, (n_cnt + (size_t) n_tiles) * sizeof(uint32_t), st));
    return MI355_OK;
}
static int splat_queue_flush(mi355_scene *s, const SplatQueue &q, hipStream_t st, bool dual, float *d_film_rgbw) {
    if (!q.cap) return MI355_OK;
    const uint32_t n_tiles = q.tiles_x * q.tiles_y;
    uint32_t *offsets = q.hist + n_tiles, *cursor = offsets + n_tiles + 1;
    const uint32_t chunks = std::min<uint32_t>(MI_SPLAT_SEGS * ((q.seg_cap + SPLAT_CHUNK - 1u) / SPLAT_CHUNK), 16384u), runs = std::min<uint32_t>((q.cap + SPLAT_RUN - 1u) / SPLAT_RUN, 65536u);
    hipLaunchKernelGGL(k_splat_hist, dim3(chunks), dim3(256), n_tiles * sizeof(uint32_t), st, s->ds, q, n_tiles);
    hipLaunchKernelGGL(k_splat_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *) q.hist, offsets, cursor, n_tiles);
    // LDS of the scatter pass: two words per tile + 20 (24 with a second record half) bytes per record: 88 KB on a 512^2 film (one 16-wave workgroup
    // per CU), 144 - 159 KB at SPLAT_MAX_TILES -- above the 64 KB a launch gets without asking
    const size_t scatter_lds = (((2 * (size_t) n_tiles + 3) & ~(size_t) 3) + (dual ? 6 : 5) * (size_t) SPLAT_SORT) * sizeof(uint32_t);
    const uint32_t sorts = std::min<uint32_t>(MI_SPLAT_SEGS * ((q.seg_cap + SPLAT_SORT - 1u) / SPLAT_SORT), 16384u);
    if (scatter_lds > 65536u) HIP_TRY(hipFuncSetAttribute((const void *) k_splat_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int) scatter_lds));
    hipLaunchKernelGGL(k_splat_scatter, dim3(sorts), dim3(SPLAT_SORT_T), scatter_lds, st, s->ds, q, (const uint32_t *) offsets, cursor, s->d_srt_a, s->d_srt_b,
                       dual ? s->d_srt_c : (float4 *) nullptr, dual ? s->d_srt_d : (float *) nullptr, n_tiles);
    if (dual) hipLaunchKernelGGL(k_splat_tiles<true>, dim3(runs), dim3(256), 0, st, s->ds, q, (const uint32_t *) offsets, (const float4 *) s->d_srt_a, (const float *) s->d_srt_b,
                                 (const float4 *) s->d_srt_c, (const float *) s->d_srt_d, d_film_rgbw, n_tiles);
    else hipLaunchKernelGGL(k_splat_tiles<false>, dim3(runs), dim3(256), 0, st, s->ds, q, (const uint32_t *) offsets, (const float4 *) s->d_srt_a, (const float *) s->d_srt_b,
                            (const float4 *) nullptr, (const float *) nullptr, d_film_rgbw, n_tiles);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
// Lists the launch's light samples cell by cell of their emitter's sample domain (see k_pt_keys).  *order = nullptr: lanes keep their
// natural order -- scenes without a BVH (every lane tests the same flat primitives anyway), MI355_FLAG_NO_LANE_SORT, tiny launches.
#define PT_SORT_MAX_BINS 65536u
static int pt_sort_lanes(mi355_scene *s, const mi355_render_params *p, const RenderArgs &a, hipStream_t st, const uint32_t **order) {
    *order = nullptr;
    if ((p->flags & MI355_FLAG_NO_LANE_SORT) || s->ds.n_bvh_shapes == 0u || a.n_local < 16384u || s->emitters.empty()) return MI355_OK;
    bool four_d = false; for (const DEmitter &e : s->emitters) four_d = four_d || e.type != MI355_EMITTER_DIRECTIONALAREA;
    const uint32_t n_em = (uint32_t) s->emitters.size(), dims = four_d ? 4u : 2u;
    uint32_t bits = 1;                                                      // cells of ~128 samples, at most 64 Ki bins over all emitters
    while (bits < (four_d ? 4u : 8u) && (uint64_t) n_em << (dims * (bits + 1u)) <= PT_SORT_MAX_BINS && ((uint64_t) n_em << (dims * (bits + 1u))) * 128u <= a.n_local) ++bits;
    const uint64_t bins64 = (uint64_t) n_em << (dims * bits);
    if (bins64 > PT_SORT_MAX_BINS) return MI355_OK;                         // (more emitters than the bins hold: keep the natural order)
    const uint32_t bins = (uint32_t) bins64;
    // The list is a pure function of the sampler streams (seed, lane layout), of how many numbers a stream draws before the emission
    // sample and of the emitter count -- not of the sensor, the geometry or any parameter value -- so the two most recent lists are kept:
    // the three sensors of a lens-design iteration render with one seed and differentiate with another (design_problem.py:560-640).
    mi355_scene::PtOrderTag tag = {};
    const uint32_t w[18] = { a.seed, a.spp_total, a.max_depth != 0u, a.flags & MI355_FLAG_HIDE_EMITTERS, a.pixel_begin, a.pixel_count, a.s_begin, a.s_count, a.n_local,
                             a.tile, a.tile_stride, a.passes, bits, four_d ? 1u : 0u, s->ds.shutter_open_time > 0.f ? 1u : 0u, s->ds.features != 0u ? 1u : 0u, n_em, bins };
    memcpy(tag.w, w, sizeof(w)); tag.stream = st; tag.valid = true;
    if (s->pt_order_capacity < a.n_local) {
        for (void *b : { (void *) s->d_pt_keys, (void *) s->d_pt_order }) if (b) HIP_TRY(hipFree(b));
        s->d_pt_keys = s->d_pt_order = nullptr; s->pt_order_capacity = 0; s->pt_tag[0].valid = s->pt_tag[1].valid = false;
        HIP_TRY(hipMalloc((void **) &s->d_pt_keys, (size_t) a.n_local * 4)); HIP_TRY(hipMalloc((void **) &s->d_pt_order, 2 * (size_t) a.n_local * 4));
        s->pt_order_capacity = a.n_local;
    }
    for (uint32_t k = 0; k < 2u; ++k)
        if (s->pt_tag[k].valid && s->pt_tag[k].stream == st && !memcmp(s->pt_tag[k].w, tag.w, sizeof(tag.w))) { *order = s->d_pt_order + (size_t) k * s->pt_order_capacity; s->pt_next = k ^ 1u; return MI355_OK; }
    const uint32_t slot = s->pt_next; s->pt_next ^= 1u; s->pt_tag[slot].valid = false;
    uint32_t *list = s->d_pt_order + (size_t) slot * s->pt_order_capacity;
    if (!s->d_pt_bins) HIP_TRY(hipMalloc((void **) &s->d_pt_bins, 2 * (size_t) PT_SORT_MAX_BINS * 4));
    uint32_t *hist = s->d_pt_bins, *cursor = s->d_pt_bins + PT_SORT_MAX_BINS;
    HIP_TRY(hipMemsetAsync(hist, 0, (size_t) bins * 4, st));
    const dim3 g((a.n_local + 255u) / 256u);
    if (s->ds.features != 0u) hipLaunchKernelGGL(k_pt_keys<true>, g, dim3(256), 0, st, s->ds, a, bits, four_d ? 1u : 0u, s->d_pt_keys, hist);
    else hipLaunchKernelGGL(k_pt_keys<false>, g, dim3(256), 0, st, s->ds, a, bits, four_d ? 1u : 0u, s->d_pt_keys, hist);
    hipLaunchKernelGGL(k_pt_scan, dim3(1), dim3(1024), 0, st, (const uint32_t *) hist, cursor, bins);
    hipLaunchKernelGGL(k_pt_scatter, g, dim3(256), 0, st, (const uint32_t *) s->d_pt_keys, cursor, list, a.n_local);
    HIP_TRY(hipGetLastError());
    s->pt_tag[slot] = tag;
    *order = list;
    return MI355_OK;
}
extern "C" int mi355_render_accumulate(mi355_scene *s, const mi355_render_params *p, float *d_film_rgbw, void *stream) {
    Phase ph(PHASE_RENDER);
    RenderArgs a; int rc = make_args(s, p, &a); if (rc) return rc;
    if (!d_film_rgbw) return fail(MI355_ERR_INVALID, "null film");
    if (p->mode != MI355_MODE_FUSED && p->mode != MI355_MODE_REPACK_ALWAYS && p->mode != MI355_MODE_STATIC) return fail(MI355_ERR_INVALID, "unknown mode");

    HIP_TRY(hipSetDevice(s->device));
    if (a.n_local == 0) return MI355_OK;
    hipStream_t st = (hipStream_t) stream; dim3 grid((a.n_local + BLOCK - 1) / BLOCK), block(BLOCK);
    if ((rc = ensure_geometry(s, st))) return rc;
    if (p->integrator == MI355_INTEGRATOR_PTRACER && s->emitters.empty()) return MI355_OK;   // "no emitters found, returning black image", integrator.cpp:590-600
    Timed tm = { s, st, (p->flags & MI355_FLAG_TIMING) != 0, 0 };
    if ((rc = tm.begin())) return rc;
    const bool full = (s->ds.features != 0u) != 0;
    const bool flat = full && s->ds.n_bvh_shapes == 0u;                      // every feature, but nothing to traverse: the <2> instantiations
    float *nof = nullptr;
    if (p->integrator == MI355_INTEGRATOR_PTRACER) {
        const float sample_scale = (float) (s->ds.crop_w * s->ds.crop_h) / (float) ((uint64_t) p->spp * s->ds.film_w * s->ds.film_h);
        // Reconstruction filters with more than one tap: deferred, tile-sorted splat (see film_put_deferred).  The queue holds every
        // connection of the launch when max_depth bounds their number; beyond its capacity a record falls back to direct atomics.
        SplatQueue q; if ((rc = splat_queue_setup(s, p, a, st, false, q))) return rc;
        const uint32_t *order; if ((rc = pt_sort_lanes(s, p, a, st, &order))) return rc;
        if (order) grid.x = xcd_grid(grid.x);
        if (flat) hipLaunchKernelGGL((k_ptracer<2, false>), grid, block, 0, st, s->ds, a, sample_scale, d_film_rgbw, (const float *) nullptr, (float *) nullptr, (const uint32_t *) nullptr, s->d_counters, q, 0u, 0u, 0u, order);
        else if (full) hipLaunchKernelGGL((k_ptracer<true, false>), grid, block, 0, st, s->ds, a, sample_scale, d_film_rgbw, (const float *) nullptr, (float *) nullptr, (const uint32_t *) nullptr, s->d_counters, q, 0u, 0u, 0u, order);
        else hipLaunchKernelGGL((k_ptracer<false, false>), grid, block, 0, st, s->ds, a, sample_scale, d_film_rgbw, (const float *) nullptr, (float *) nullptr, (const uint32_t *) nullptr, s->d_counters, q, 0u, 0u, 0u, order);
        HIP_TRY(hipGetLastError());
        if ((rc = splat_queue_flush(s, q, st, false, d_film_rgbw))) return rc;
        if ((rc = tm.end())) return rc;
        s->n_gen += a.n_local;
        return MI355_OK;
    }
    // path integrator: re-pack the workgroup's live paths from bounce 2 on, whenever that empties a wave (MODE_FUSED);
    // after every bounce unconditionally (MODE_REPACK_ALWAYS); never (MODE_STATIC)
    // (bit 31 stands for every bounce >= 31 and must stay set: the count exchange is also the loop's exit test when max_depth = -1)
    a.compact_mask = p->mode == MI355_MODE_REPACK_ALWAYS ? 0xffffffffu : 0xfffffffcu;
    a.compact_adaptive = p->mode == MI355_MODE_REPACK_ALWAYS ? 0u : 1u;
    if (p->integrator == MI355_INTEGRATOR_PATH && a.passes > 1u) {          // samples_per_pass: the static kernel walks the passes of a stream
        if (full) hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_FILM, true>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
        else hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_FILM, false>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
    } else
    if (p->integrator == MI355_INTEGRATOR_PRB && p->mode == MI355_MODE_STATIC) {
        if (full) hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PRB, OUT_FILM, true>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
        else hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PRB, OUT_FILM, false>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
    } else if (p->integrator == MI355_INTEGRATOR_PRB) {
        if (flat) hipLaunchKernelGGL((k_render_compact<MI355_INTEGRATOR_PRB, 2>), grid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
        else if (full) hipLaunchKernelGGL((k_render_compact<MI355_INTEGRATOR_PRB, true>), grid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
        else hipLaunchKernelGGL((k_render_compact<MI355_INTEGRATOR_PRB, false>), grid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
    } else if (a.s_count >= 128u && (p->flags & MI355_FLAG_REGEN) && p->mode != MI355_MODE_REPACK_ALWAYS) {       // enough samples per pixel to keep a wave's pool full
        uint32_t n_waves = (a.pixel_count + a.ppw - 1) / a.ppw; dim3 rgrid((n_waves + WAVES - 1) / WAVES);
        if (flat) hipLaunchKernelGGL(k_render_regen<2>, rgrid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
        else if (full) hipLaunchKernelGGL(k_render_regen<true>, rgrid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
        else hipLaunchKernelGGL(k_render_regen<false>, rgrid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
    }
    else if (p->mode != MI355_MODE_STATIC) {
        if (flat) hipLaunchKernelGGL((k_render_compact<MI355_INTEGRATOR_PATH, 2>), grid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
        else if (full) hipLaunchKernelGGL((k_render_compact<MI355_INTEGRATOR_PATH, true>), grid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
        else hipLaunchKernelGGL((k_render_compact<MI355_INTEGRATOR_PATH, false>), grid, block, 0, st, s->ds, a, d_film_rgbw, s->d_counters);
    }
    else if (flat) hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_FILM, 2>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
    else if (full) hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_FILM, true>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
    else hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_FILM, false>), grid, block, 0, st, s->ds, a, d_film_rgbw, nof, s->d_counters);
    HIP_TRY(hipGetLastError());
    if ((rc = tm.end())) return rc;
    s->n_gen += (uint64_t) a.n_local * a.passes; s->n_splat += (uint64_t) a.n_local * a.passes;
    return MI355_OK;
}
// The multi-GPU exchange of SURVEY 8(e), without a communication library in this one: the caller's hook sums a device buffer over all
// ranks in place (RCCL ncclAllReduce on `stream` in the shim of INTEGRATION.md section 4).  Collective: every rank reaches it, also one
// whose shard is empty.
static int reduce_hook(mi355_scene *s, float *d_buffer, uint64_t count, void *stream) {
    if (!s->reduce_fn) return MI355_OK;
    const int rc = s->reduce_fn(s->reduce_user, d_buffer, count, stream);
    return rc ? fail(MI355_ERR_DEVICE, "the reduce hook reported a failure") : MI355_OK;
}
extern "C" int mi355_set_reduce_hook(mi355_scene *s, mi355_reduce_fn fn, void *user) {
    if (!s) return fail(MI355_ERR_INVALID, "null scene");
    s->reduce_fn = fn; s->reduce_user = user;
    return MI355_OK;
}
extern "C" int mi355_film_develop(mi355_scene *s, uint32_t sensor, const float *d_film_rgbw, float *d_image_rgb, void *stream) {
    Phase ph(PHASE_DEVELOP);
    if (!s || !d_film_rgbw || !d_image_rgb) return fail(MI355_ERR_INVALID, "null argument");
    if (sensor >= s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range");
    HIP_TRY(hipSetDevice(s->device));
    uint32_t n = s->sensors[sensor].snap.crop_w * s->sensors[sensor].snap.crop_h;
    hipLaunchKernelGGL(k_develop, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t) stream, d_film_rgbw, d_image_rgb, n);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_render(mi355_scene *s, const mi355_render_params *p, float *d_film_rgbw, float *d_image_rgb, void *stream) {
    if (!s || !p || !d_film_rgbw) return fail(MI355_ERR_INVALID, "null argument");
    { int rc_s = select_sensor(s, p->sensor); if (rc_s) return rc_s; }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(d_film_rgbw, 0, (size_t) s->ds.crop_w * s->ds.crop_h * 16, (hipStream_t) stream));
    int rc = mi355_render_accumulate(s, p, d_film_rgbw, stream); if (rc) return rc;
    if ((rc = reduce_hook(s, d_film_rgbw, (uint64_t) s->ds.crop_w * s->ds.crop_h * 4u, stream))) return rc;   // the film of the whole job
    if (d_image_rgb) return mi355_film_develop(s, p->sensor, d_film_rgbw, d_image_rgb, stream);
    return MI355_OK;
}
extern "C" int mi355_prb_splat_weights(mi355_scene *s, const mi355_render_params *p, float *d_weight, void *stream) {
    RenderArgs a; int rc = make_args(s, p, &a); if (rc) return rc;
    if (!d_weight) return fail(MI355_ERR_INVALID, "null weight buffer");
    HIP_TRY(hipSetDevice(s->device));
    if (a.n_local != 0) {
        hipLaunchKernelGGL(k_splat_weights, dim3((a.n_local + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t) stream, s->ds, a, d_weight);
        HIP_TRY(hipGetLastError());
    }
    return reduce_hook(s, d_weight, (uint64_t) s->ds.crop_w * s->ds.crop_h, stream);
}
extern "C" int mi355_render_forward(mi355_scene *s, const mi355_render_params *p, const float *d_tangent_params, float *d_film_rgbw, void *stream) {
    Phase ph(PHASE_RENDER_FORWARD);
    RenderArgs a; int rc = make_args(s, p, &a); if (rc) return rc;
    if (!d_tangent_params || !d_film_rgbw) return fail(MI355_ERR_INVALID, "null argument");
    if (p->integrator != MI355_INTEGRATOR_PATH && p->integrator != MI355_INTEGRATOR_PRB) return fail(MI355_ERR_UNSUPPORTED, "render_forward is built for the path / prb integrators");
    HIP_TRY(hipSetDevice(s->device));
    if (a.n_local == 0) return MI355_OK;
    hipStream_t st = (hipStream_t) stream;
    if ((rc = ensure_geometry(s, st))) return rc;
    Timed tm = { s, st, (p->flags & MI355_FLAG_TIMING) != 0, 0 };
    if ((rc = tm.begin())) return rc;
    const dim3 grid((a.n_local + BLOCK - 1) / BLOCK), block(BLOCK);
    if (s->ds.features != 0u) hipLaunchKernelGGL(k_render_forward<true>, grid, block, 0, st, s->ds, a, d_tangent_params, (const uint32_t *) s->d_small_map, d_film_rgbw, s->d_counters);
    else hipLaunchKernelGGL(k_render_forward<false>, grid, block, 0, st, s->ds, a, d_tangent_params, (const uint32_t *) s->d_small_map, d_film_rgbw, s->d_counters);
    HIP_TRY(hipGetLastError());
    if ((rc = tm.end())) return rc;
    s->n_gen += a.n_local; s->n_splat += a.n_local;
    return MI355_OK;
}
extern "C" int mi355_ptracer_forward_design(mi355_scene *s, const mi355_render_params *p, const mi355_design_tangent *t, float *d_film_rgbw, void *stream) {
    Phase ph(PHASE_RENDER_FORWARD);
    if (!s || !p || !t || !d_film_rgbw) return fail(MI355_ERR_INVALID, "null argument");
    if (p->integrator != MI355_INTEGRATOR_PTRACER) return fail(MI355_ERR_INVALID, "mi355_ptracer_forward_design runs the particle tracer: set integrator = MI355_INTEGRATOR_PTRACER");
    if (t->shape != MI355_NONE) {
        if (!t->d_vertex_tangent) return fail(MI355_ERR_INVALID, "vertex tangent missing");
        if (t->shape >= s->shapes.size() || s->shapes[t->shape].type != MI355_SHAPE_MESH) return fail(MI355_ERR_INVALID, "the differentiated shape must be a mesh");
        for (const DEmitter &e : s->emitters) if (e.shape == t->shape) return fail(MI355_ERR_UNSUPPORTED, "the differentiated mesh must not carry an emitter");
    }
    float d_rel = 0.f;
    if (t->bsdf != MI355_NONE) {
        if (t->bsdf >= s->bsdfs.size() || s->bsdfs[t->bsdf].type != MI355_BSDF_DIELECTRIC) return fail(MI355_ERR_INVALID, "the differentiated bsdf must be a dielectric");
        const float ext = s->ext_iors[t->bsdf], rel = s->bsdfs[t->bsdf].rel_eta;   // rel_eta = int_ior / ext_ior
        d_rel = t->d_int_ior / ext - rel * t->d_ext_ior / ext;
    }
    RenderArgs a; int rc = make_args(s, p, &a); if (rc) return rc;
    if (s->ds.sensor_type != MI355_SENSOR_PERSPECTIVE) return fail(MI355_ERR_UNSUPPORTED, "the design forward mode is built for the perspective sensor");
    HIP_TRY(hipSetDevice(s->device));
    if (a.n_local == 0 || s->emitters.empty()) return MI355_OK;
    hipStream_t st = (hipStream_t) stream;
    if ((rc = ensure_geometry(s, st))) return rc;
    const float sample_scale = (float) (s->ds.crop_w * s->ds.crop_h) / (float) ((uint64_t) p->spp * s->ds.film_w * s->ds.film_h);
    const dim3 grid((a.n_local + BLOCK - 1) / BLOCK), block(BLOCK);
    const float *ntan = nullptr;
    if (t->shape != MI355_NONE && (s->shapes[t->shape].flags & DSHAPE_NRM_DEP)) {
        // the mesh's vertex normals are functions of its positions (they were recomputed by a parameter update): their tangent goes with it
        const DShape &sh = s->shapes[t->shape]; const uint32_t nv = s->vcount[t->shape];
        if (s->ntan_floats < 3 * (size_t) nv) {
            if (s->d_ntan) HIP_TRY(hipFree(s->d_ntan));
            s->d_ntan = nullptr; s->ntan_floats = 0;
            HIP_TRY(hipMalloc((void **) &s->d_ntan, 3 * (size_t) nv * sizeof(float))); s->ntan_floats = 3 * (size_t) nv;
        }
        hipLaunchKernelGGL(k_vertex_normals, dim3((nv + 255u) / 256u), dim3(256), 0, st, (const float *) s->d_positions, (const uint32_t *) s->d_faces, (const uint32_t *) s->d_vadj_off,
                           (const uint32_t *) s->d_vadj, sh.vtx_offset, sh.tri_offset, nv, (float *) nullptr, (float *) nullptr, t->d_vertex_tangent, s->d_ntan);
        HIP_TRY(hipGetLastError());
        ntan = s->d_ntan;
    }
    GeoTangent gt = { t->shape, t->shape != MI355_NONE ? t->d_vertex_tangent : nullptr, t->bsdf, d_rel, ntan };
    SplatQueue q; if ((rc = splat_queue_setup(s, p, a, st, true, q))) return rc;
    const uint32_t *order; if ((rc = pt_sort_lanes(s, p, a, st, &order))) return rc;
    const dim3 ggrid(order ? xcd_grid(grid.x) : grid.x);
    if (s->ds.features != 0u) hipLaunchKernelGGL(k_ptracer_geo<true>, ggrid, block, 0, st, s->ds, a, sample_scale, d_film_rgbw, gt, q, order);
    else hipLaunchKernelGGL(k_ptracer_geo<false>, ggrid, block, 0, st, s->ds, a, sample_scale, d_film_rgbw, gt, q, order);
    HIP_TRY(hipGetLastError());
    return splat_queue_flush(s, q, st, true, d_film_rgbw);
}
struct MultiAdjoint { DSensorSet set; GeoGradSet grads; };                  // mi355_render_backward_sensors: several cameras, one walk and one sweep
static int render_backward_impl(mi355_scene *s, const mi355_render_params *p, const float *d_grad_in, const float *d_weight, float *d_grad_params, void *stream, const MultiAdjoint *multi = nullptr);
extern "C" int mi355_render_backward(mi355_scene *s, const mi355_render_params *p, const float *d_grad_in, const float *d_weight,
                                     float *d_grad_params, void *stream) {
    Phase ph(PHASE_RENDER_BACKWARD);
    int rc = render_backward_impl(s, p, d_grad_in, d_weight, d_grad_params, stream); if (rc) return rc;
    return reduce_hook(s, d_grad_params, s->grad_total, stream);             // the gradient of the whole job, on every rank
}
// The adjoint of mi355_render_accumulate_sensors' films with respect to vertex positions and IORs: the cameras of `sensors` differentiate with one
// seed, hence over the same light paths -- ONE forward walk onto the tape, ONE backward sweep in which every vertex gathers the image adjoint of
// each camera that it connected to.  d_grad_in[k]: H x W x 3 for sensor k, or NULL (that image carries no adjoint).
extern "C" int mi355_render_backward_sensors(mi355_scene *s, const mi355_render_params *p, uint32_t n_sensors, const uint32_t *sensors, const float *const *d_grad_in,
                                             float *d_grad_params, void *stream) {
    Phase ph(PHASE_RENDER_BACKWARD);
    if (!s || !p || !sensors || !d_grad_in || !d_grad_params) return fail(MI355_ERR_INVALID, "null argument");
    if (p->integrator != MI355_INTEGRATOR_PTRACER || !(p->flags & MI355_FLAG_GEOMETRY_GRADS) || !(p->flags & MI355_FLAG_SKIP_COLOUR_GRADS))
        return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: the particle tracer's geometry / IOR adjoint only (MI355_FLAG_GEOMETRY_GRADS | MI355_FLAG_SKIP_COLOUR_GRADS)");
    if (!n_sensors || n_sensors > MI_MULTI_MAX) return fail(MI355_ERR_INVALID, "1 .. MI355_MAX_SENSORS_PER_PASS (8) sensors");
    if ((p->passes ? p->passes : 1u) != 1u) return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: samples_per_pass is not carried");   // (a shard of the light-sample grid is fine: it is the same for every camera)
    for (uint32_t k = 0; k < n_sensors; ++k) if (sensors[k] >= s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range");
    const DScene &f = s->sensors[sensors[0]].snap;
    MultiAdjoint m; memset(&m, 0, sizeof(m)); m.set.n = n_sensors; m.set.film_w = (float) f.film_w; m.set.film_h = (float) f.film_h;
    const float *first_grad = nullptr;
    for (uint32_t k = 0; k < n_sensors; ++k) {
        const DScene &g = s->sensors[sensors[k]].snap; DSensorView &v = m.set.v[k];
        if (g.sensor_type != MI355_SENSOR_PERSPECTIVE) return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: perspective sensors only");
        if (g.film_w != f.film_w || g.film_h != f.film_h || g.crop_w != g.film_w || g.crop_h != g.film_h || g.crop_x || g.crop_y || g.rfilter != f.rfilter || g.filter_radius != f.filter_radius ||
            memcmp(g.filter_coeff, f.filter_coeff, sizeof(f.filter_coeff)) || (g.shutter_open_time > 0.f) != (f.shutter_open_time > 0.f))
            return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: the films must have one size (no crop window), one reconstruction filter and one shutter setting");
        memcpy(v.cam_to_world, g.cam_to_world, sizeof(v.cam_to_world)); memcpy(v.camera_to_sample, g.camera_to_sample, sizeof(v.camera_to_sample)); memcpy(v.image_rect, g.image_rect, sizeof(v.image_rect));
        v.normalization = g.normalization; v.near_clip = g.near_clip; v.far_clip = g.far_clip; v.x_offset = 0.f;
        m.grads.g[k] = d_grad_in[k]; if (d_grad_in[k] && !first_grad) first_grad = d_grad_in[k];
    }
    if (!first_grad) return MI355_OK;                                       // no image carries an adjoint: the gradient is zero
    mi355_render_params p0 = *p; p0.sensor = sensors[0];
    int rc = render_backward_impl(s, &p0, first_grad, nullptr, d_grad_params, stream, &m); if (rc) return rc;
    return reduce_hook(s, d_grad_params, s->grad_total, stream);
}
static int render_backward_impl(mi355_scene *s, const mi355_render_params *p, const float *d_grad_in, const float *d_weight, float *d_grad_params, void *stream, const MultiAdjoint *multi) {
    RenderArgs a; int rc = make_args(s, p, &a); if (rc) return rc;
    const bool pt = p->integrator == MI355_INTEGRATOR_PTRACER;
    if (!d_grad_in || (!d_weight && !pt) || !d_grad_params) return fail(MI355_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    if (a.n_local == 0) return MI355_OK;
    hipStream_t st = (hipStream_t) stream;
    if ((rc = ensure_geometry(s, st))) return rc;
    if (pt && s->emitters.empty()) return MI355_OK;
    if (pt && (p->flags & MI355_FLAG_GEOMETRY_GRADS)) {
        // The geometry adjoint treats emitters as static (the emitter's sampled position and direction carry no derivative), so a gradient
        // slot on an emitter-carrying mesh would silently miss the term AD includes: refuse, like the forward design mode does.
        for (const DEmitter &e : s->emitters) if (e.shape < s->shapes.size() && s->shapes[e.shape].type == MI355_SHAPE_MESH && s->shapes[e.shape].grad_offset != 0xffffffffu)
            return fail(MI355_ERR_UNSUPPORTED, "geometry gradients: a mesh that carries an emitter cannot be differentiated (its emitted rays are treated as static)");
        if (s->shapes.size() >= (1u << (32 - MI_TF_SHIFT))) return fail(MI355_ERR_UNSUPPORTED, "geometry gradients: the path tape packs the shape index into 20 bits");
    }
    Timed tm = { s, st, (p->flags & MI355_FLAG_TIMING) != 0, 0 };
    if ((rc = tm.begin())) return rc;
    if (pt) {                                                               // Integrator::render_backward (integrator.cpp:65-84) over ptracer.cpp, radiances only
        const float sample_scale = (float) (s->ds.crop_w * s->ds.crop_h) / (float) ((uint64_t) p->spp * s->ds.film_w * s->ds.film_h);
        dim3 pgrid((a.n_local + BLOCK - 1) / BLOCK), pblock(BLOCK);
        const uint32_t n_small_pt = (uint32_t) s->small_map.size();
        uint32_t cs = 5;                                                    // 32 bank-spread copies while they fit in 8 KB
        while (cs > 0 && (n_small_pt << cs) > 2048u) --cs;
        const uint32_t all = (p->flags & MI355_FLAG_EMITTER_GRADS_ONLY) ? 0u : 1u;
        const size_t dyn = (size_t) (n_small_pt << cs) * sizeof(float);
        const uint32_t *order; if ((rc = pt_sort_lanes(s, p, a, st, &order))) return rc;
        if (order) pgrid.x = xcd_grid(pgrid.x);
        if (!((p->flags & MI355_FLAG_GEOMETRY_GRADS) && (p->flags & MI355_FLAG_SKIP_COLOUR_GRADS))) {      // (a lens-design step asks for geometry / IOR gradients only)
            if (s->ds.features != 0u) hipLaunchKernelGGL((k_ptracer<true, true>), pgrid, pblock, dyn, st, s->ds, a, sample_scale, (float *) nullptr, d_grad_in, d_grad_params, (const uint32_t *) s->d_small_map, s->d_counters, SplatQueue{}, n_small_pt, cs, all, order);
            else hipLaunchKernelGGL((k_ptracer<false, true>), pgrid, pblock, dyn, st, s->ds, a, sample_scale, (float *) nullptr, d_grad_in, d_grad_params, (const uint32_t *) s->d_small_map, s->d_counters, SplatQueue{}, n_small_pt, cs, all, order);
            HIP_TRY(hipGetLastError());
        }
        if (p->flags & MI355_FLAG_GEOMETRY_GRADS) {                         // vertex positions and IORs: forward walk onto the path tape, backward sweep
            if (s->ds.sensor_type != MI355_SENSOR_PERSPECTIVE) return fail(MI355_ERR_UNSUPPORTED, "geometry gradients are built for the perspective sensor");
            if (p->max_depth < 0 || p->max_depth > MI_TAPE_MAX_VERTS + 1) return fail(MI355_ERR_UNSUPPORTED, "geometry gradients need a finite max_depth <= 33 (the path tape holds max_depth - 1 vertices)");
            const uint32_t n_verts = p->max_depth > 1 ? (uint32_t) p->max_depth - 1u : 0u;
            if (n_verts) {
                const uint32_t chunk = std::min<uint32_t>(a.n_local, 1u << 22);    // lanes per launch: bounds the tape at 18 dwords x vertices x 4 Mi lanes
                const uint32_t stride = xcd_grid((chunk + BLOCK - 1) / BLOCK) * BLOCK;     // one tape slot per HARDWARE lane of a launch (the grid is rounded up to 8 workgroups)
                const size_t need = (size_t) MI_TAPE_FIELDS * n_verts * stride;
                if (s->tape_floats < need) {
                    if (s->d_tape) HIP_TRY(hipFree(s->d_tape));
                    s->d_tape = nullptr; s->tape_floats = 0;
                    HIP_TRY(hipMalloc((void **) &s->d_tape, need * sizeof(float))); s->tape_floats = need;
                }
                const size_t nv_total = s->positions.size() / 3;
                HIP_TRY(hipMemsetAsync(s->d_nrm_adj, 0, std::max<size_t>(nv_total, 1) * 3 * sizeof(float), st));
                const uint32_t n_ior = s->bsdfs.size() <= 2048 ? 2u * (uint32_t) s->bsdfs.size() : 0u;
                GeoTape tape = { s->d_tape, stride };
                // (walk and sweep as two kernels -- the walk at 3 waves per SIMD, the sweep without a traversal stack -- measured SLOWER, 10.7 against
                //  9.15 ms on the lens doublet: fused, a workgroup reads its tape rows back while they are still in L2)
                for (uint32_t first = 0; first < a.n_local; first += chunk) {
                    const uint32_t n = std::min<uint32_t>(chunk, a.n_local - first); const dim3 g(order ? xcd_grid((n + BLOCK - 1) / BLOCK) : (n + BLOCK - 1) / BLOCK);
                    if (multi) {
                        if (s->ds.features != 0u) hipLaunchKernelGGL((k_ptracer_geo_bwd<true, true>), g, pblock, n_ior * sizeof(float), st, s->ds, a, sample_scale, d_grad_in, d_grad_params, s->d_nrm_adj, tape, first, n, n_ior, order, multi->set, multi->grads);
                        else hipLaunchKernelGGL((k_ptracer_geo_bwd<false, true>), g, pblock, n_ior * sizeof(float), st, s->ds, a, sample_scale, d_grad_in, d_grad_params, s->d_nrm_adj, tape, first, n, n_ior, order, multi->set, multi->grads);
                    } else {
                        DSensorSet no_set; GeoGradSet no_grads; memset(&no_set, 0, sizeof(no_set)); memset(&no_grads, 0, sizeof(no_grads));
                        if (s->ds.features != 0u) hipLaunchKernelGGL((k_ptracer_geo_bwd<true, false>), g, pblock, n_ior * sizeof(float), st, s->ds, a, sample_scale, d_grad_in, d_grad_params, s->d_nrm_adj, tape, first, n, n_ior, order, no_set, no_grads);
                        else hipLaunchKernelGGL((k_ptracer_geo_bwd<false, false>), g, pblock, n_ior * sizeof(float), st, s->ds, a, sample_scale, d_grad_in, d_grad_params, s->d_nrm_adj, tape, first, n, n_ior, order, no_set, no_grads);
                    }
                }
                HIP_TRY(hipGetLastError());
                for (uint32_t i = 0; i < s->shapes.size(); ++i) {           // vertex-normal adjoints -> positions, for normals that follow the positions
                    const DShape &sh = s->shapes[i];
                    if (sh.type != MI355_SHAPE_MESH || !(sh.flags & DSHAPE_NRM_DEP) || sh.grad_offset == 0xffffffffu) continue;
                    hipLaunchKernelGGL(k_vertex_normals_bwd, dim3((uint32_t) (((uint64_t) s->vcount[i] * MI_VNB_LANES + 255u) / 256u)), dim3(256), 0, st, (const float *) s->d_positions, (const float *) s->d_normals, (const float *) s->d_nrm_len,
                                       (const uint32_t *) s->d_faces, (const uint32_t *) s->d_vadj_off, (const uint32_t *) s->d_vadj, sh.vtx_offset, sh.tri_offset, s->vcount[i],
                                       (const float *) s->d_nrm_adj, d_grad_params + sh.grad_offset);
                }
                HIP_TRY(hipGetLastError());
            }
        }
        return tm.end();
    }
    uint32_t npix = s->ds.crop_w * s->ds.crop_h;
    if (s->gw_pixels < npix) {                                              // grad_in / W scratch, sized by the largest film this handle has served
        if (s->d_gw) HIP_TRY(hipFree(s->d_gw));
        s->d_gw = nullptr; s->gw_pixels = 0;
        HIP_TRY(hipMalloc((void **) &s->d_gw, (size_t) npix * sizeof(float4))); s->gw_pixels = npix;
    }
    hipLaunchKernelGGL(k_prepare_grad, dim3((npix + 255) / 256), dim3(256), 0, st, d_grad_in, d_weight, s->d_gw, npix);
    const dim3 grid((a.n_local + BLOCK - 1) / BLOCK), block(BLOCK);
    const uint32_t n_small = (uint32_t) s->small_map.size();
    const bool full = s->ds.features != 0u;
    uint32_t *redo = nullptr;
    if (!(p->flags & MI355_FLAG_TWO_PASS)) {
        // one walk per path; workgroups whose LDS record queue overflowed are flagged in `redo` and redone below
        if (s->redo_capacity < grid.x) {
            if (s->d_redo) HIP_TRY(hipFree(s->d_redo));
            s->d_redo = nullptr; s->redo_capacity = 0;
            HIP_TRY(hipMalloc((void **) &s->d_redo, (size_t) grid.x * sizeof(uint32_t))); s->redo_capacity = grid.x;
        }
        redo = s->d_redo;
        HIP_TRY(hipMemsetAsync(redo, 0, (size_t) grid.x * sizeof(uint32_t), st));
        const bool bvh = s->ds.n_bvh_shapes != 0u;
        if (n_small > MAX_SMALL_GRAD) return fail(MI355_ERR_UNSUPPORTED, "too many constant parameters for the LDS accumulators");
        // dynamic LDS = small-parameter accumulators + record queue, sized so that 4 workgroups share a CU's 160 KB
        const uint32_t static_lds = (bvh ? WAVES * STACK_WORDS * 4u : 256u) + 3u * BLOCK * 4u + 64u + (full ? 0u : MI_LT_BYTES);   // BASIC: + the record tables staged in LDS
        uint32_t copy_shift = 5;                                            // 32 copies (one per LDS bank) while they fit in 8 KB
        while (copy_shift > 0 && (n_small << copy_shift) > 2048u) --copy_shift;
        const uint32_t small_bytes = (((n_small << copy_shift) + 3u) / 4u) * 16u + (((n_small << copy_shift) + 1u) / 2u) * 16u;   // fp32 accumulators + their 64-bit twins
        const uint32_t lds_budget = (full ? 163840u / 3u : 163840u / MI_MIN_WAVES_1P);   // FULL: three workgroups per CU (its launch bounds), 54 KB each
        if (static_lds + small_bytes + 256u + 64u * 16u > lds_budget) return fail(MI355_ERR_UNSUPPORTED, "no LDS left for the gradient record queue");
        uint32_t q_granules = (lds_budget - 256u - static_lds - small_bytes) / 16u;
        const uint32_t dyn_bytes = small_bytes + q_granules * 16u;
#define LAUNCH_1P(F, B) hipLaunchKernelGGL((k_prb_backward_1p<F, B>), grid, block, dyn_bytes, st, s->ds, a, (const float4 *) s->d_gw, d_grad_params, s->d_small_map, n_small, s->d_counters, redo, q_granules, copy_shift)
        if (full) { if (bvh) LAUNCH_1P(true, true); else LAUNCH_1P(true, false); }
        else { if (bvh) LAUNCH_1P(false, true); else LAUNCH_1P(false, false); }
#undef LAUNCH_1P
        HIP_TRY(hipGetLastError());
    }
    if (full) hipLaunchKernelGGL(k_prb_backward<true>, grid, block, 0, st, s->ds, a, (const float4 *) s->d_gw, d_grad_params, s->d_small_map, n_small, s->d_counters, (const uint32_t *) redo);
    else hipLaunchKernelGGL(k_prb_backward<false>, grid, block, 0, st, s->ds, a, (const float4 *) s->d_gw, d_grad_params, s->d_small_map, n_small, s->d_counters, (const uint32_t *) redo);
    HIP_TRY(hipGetLastError());
    if ((rc = tm.end())) return rc;
    s->n_gen += a.n_local;
    return MI355_OK;
}
extern "C" int mi355_get_counters(mi355_scene *s, mi355_counters *out, int reset) {
    if (!s || !out) return fail(MI355_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    int rc = drain_events(s); if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[8]; HIP_TRY(hipMemcpy(h, s->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    out->n_gen = s->n_gen; out->n_isect = h[1]; out->n_shade = h[2]; out->n_shadow = h[3]; out->n_splat = s->n_splat;
    out->kernel_ms = s->ms_accum; out->kernel_launches = s->launches_accum; out->n_redo_groups = h[4];
#ifdef MI_PROF
    { unsigned long long pr[8]; HIP_TRY(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_prof), sizeof(pr)));
      fprintf(stderr, "MI_PROF cycles: isect %llu si+hit %llu emitter %llu bsdf %llu shadow %llu tail %llu gen+repack %llu splat %llu\n",
              pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pr[6], pr[7]);
      if (reset) { memset(pr, 0, sizeof(pr)); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), pr, sizeof(pr))); } }
#endif
    if (reset) { HIP_TRY(hipMemset(s->d_counters, 0, sizeof(h))); s->n_gen = s->n_splat = 0; s->ms_accum = 0.f; s->launches_accum = 0; }
    return MI355_OK;
}
extern "C" int mi355_lens_exit_pupil_bounds(const mi355_scene *s, uint32_t sensor, float *out) {
    if (!s || !out) return fail(MI355_ERR_INVALID, "null argument");
    if (sensor >= s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range");
    const std::vector<float> &pupil = s->sensors[sensor].pupil;
    if (pupil.empty()) return fail(MI355_ERR_INVALID, "the sensor does not sample an exit pupil");
    memcpy(out, pupil.data(), pupil.size() * sizeof(float));
    return MI355_OK;
}
extern "C" int mi355_scene_info(const mi355_scene *s, uint32_t out[8]) {
    if (!s || !out) return fail(MI355_ERR_INVALID, "null argument");
    out[0] = (uint32_t) s->prims.size(); out[1] = (uint32_t) s->boxes.size(); out[2] = (uint32_t) s->bvh_shapes.size();
    out[3] = (uint32_t) s->bvh.size(); out[4] = (uint32_t) s->n_faces_total; out[5] = (uint32_t) s->grad_total;
    out[6] = (uint32_t) s->spheres.size(); out[7] = s->ds.features;
    return MI355_OK;
}
extern "C" int mi355_debug_lane_radiance(mi355_scene *s, const mi355_render_params *p, uint32_t lane_begin, uint32_t count, float *d_out, void *stream) {
    RenderArgs a; int rc = make_args(s, p, &a); if (rc) return rc;
    if (!d_out) return fail(MI355_ERR_INVALID, "null output");
    if (a.passes > 1u) return fail(MI355_ERR_UNSUPPORTED, "per-lane output is defined for single-pass jobs");
    uint64_t total = (uint64_t) s->ds.grid_w * s->ds.grid_h * p->spp;
    if ((uint64_t) lane_begin + count > total) return fail(MI355_ERR_INVALID, "lane range outside the job");
    HIP_TRY(hipSetDevice(s->device));
    if (!count) return MI355_OK;
    a.n_local = count; a.lane_begin = lane_begin; a.seg = 0;
    dim3 grid((count + BLOCK - 1) / BLOCK), block(BLOCK); hipStream_t st = (hipStream_t) stream;
    if ((rc = ensure_geometry(s, st))) return rc;
    float *nof = nullptr; unsigned long long *noc = nullptr; const bool full = (s->ds.features != 0u) != 0;
    if (p->integrator == MI355_INTEGRATOR_PRB) {
        if (full) hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PRB, OUT_LANES, true>), grid, block, 0, st, s->ds, a, nof, d_out, noc);
        else hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PRB, OUT_LANES, false>), grid, block, 0, st, s->ds, a, nof, d_out, noc);
    } else {
        if (full) hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_LANES, true>), grid, block, 0, st, s->ds, a, nof, d_out, noc);
        else hipLaunchKernelGGL((k_render<MI355_INTEGRATOR_PATH, OUT_LANES, false>), grid, block, 0, st, s->ds, a, nof, d_out, noc);
    }
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_debug_bvh_stats(mi355_scene *s, uint64_t out[4], int reset) {
    if (!s || !out) return fail(MI355_ERR_INVALID, "null argument");
    out[0] = out[1] = out[2] = out[3] = 0;
#ifdef MI_BVH_STATS
    HIP_TRY(hipSetDevice(s->device)); HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[4]; HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bvh_stats), sizeof(h)));
    for (int i = 0; i < 4; ++i) out[i] = h[i];
    if (reset) { memset(h, 0, sizeof(h)); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_bvh_stats), h, sizeof(h))); }
#else
    (void) reset;
#endif
    return MI355_OK;
}
extern "C" int mi355_debug_set_splat_capacity(mi355_scene *s, uint32_t records) {
    if (!s) return fail(MI355_ERR_INVALID, "null argument");
    s->splat_cap_override = records; return MI355_OK;
}
extern "C" int mi355_debug_surface_interaction(mi355_scene *s, const float *d_rays, uint32_t n, float *d_si, void *stream) {
    if (!s || !d_rays || !d_si) return fail(MI355_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    if (!n) return MI355_OK;
    { int rc = ensure_geometry(s, (hipStream_t) stream); if (rc) return rc; }
    hipLaunchKernelGGL(k_debug_si, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t) stream, s->ds, d_rays, n, d_si);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_debug_texture_eval(mi355_scene *s, uint32_t texture, const float *d_uv, uint32_t n, float *d_rgb, void *stream) {
    if (!s || !d_uv || !d_rgb) return fail(MI355_ERR_INVALID, "null argument");
    if (texture >= s->ds.n_textures) return fail(MI355_ERR_INVALID, "texture index out of range");
    HIP_TRY(hipSetDevice(s->device));
    if (!n) return MI355_OK;
    hipLaunchKernelGGL(k_debug_texture, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t) stream, s->ds, texture, d_uv, n, d_rgb);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_lens_sag(const float *d_eval_xy, const float *d_base_positions, uint32_t n, float radial_extent, const float *d_params, float *d_positions, void *stream) {
    if (!d_base_positions || !d_params || !d_positions) return fail(MI355_ERR_INVALID, "null argument");
    if (!(radial_extent > 0.f)) return fail(MI355_ERR_INVALID, "radial_extent must be positive");
    if (!n) return MI355_OK;
    hipLaunchKernelGGL(k_lens_sag, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t) stream, d_eval_xy, d_base_positions, n, radial_extent, d_params, d_positions);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_lens_sag_backward(const float *d_eval_xy, const float *d_base_positions, uint32_t n, float radial_extent, const float *d_params,
                                       const float *d_grad_positions, float *d_grad_params, void *stream) {
    if (!d_base_positions || !d_params || !d_grad_positions || !d_grad_params) return fail(MI355_ERR_INVALID, "null argument");
    if (!(radial_extent > 0.f)) return fail(MI355_ERR_INVALID, "radial_extent must be positive");
    if (!n) return MI355_OK;
    hipLaunchKernelGGL(k_lens_sag_bwd, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t) stream, d_eval_xy, d_base_positions, n, radial_extent, d_params, d_grad_positions, d_grad_params);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
// Several perspective sensors of equal film size in ONE light-traced pass (see DSensorSet).  d_film: H x virt_w x 4 floats, sensor k's film in the
// columns [k * pitch + MI_SPLAT_TILE, k * pitch + MI_SPLAT_TILE + W), pitch = roundup(W, 16) + 32, virt_w = n * pitch (mi355_multi_film_layout).
extern "C" int mi355_multi_film_layout(uint32_t film_w, uint32_t n_sensors, uint32_t *pitch, uint32_t *x0, uint32_t *virt_w) {
    if (!film_w || !n_sensors || n_sensors > MI_MULTI_MAX) return fail(MI355_ERR_INVALID, "1 .. MI355_MAX_SENSORS_PER_PASS (8) sensors");
    const uint32_t pt = (film_w + MI_SPLAT_TILE - 1u) / MI_SPLAT_TILE * MI_SPLAT_TILE + 2u * MI_SPLAT_TILE;
    if (pitch) *pitch = pt; if (x0) *x0 = MI_SPLAT_TILE; if (virt_w) *virt_w = n_sensors * pt;
    return MI355_OK;
}
extern "C" int mi355_render_accumulate_sensors(mi355_scene *s, const mi355_render_params *p, uint32_t n_sensors, const uint32_t *sensors, float *d_film, void *stream) {
    Phase ph(PHASE_RENDER);
    if (!s || !p || !sensors || !d_film) return fail(MI355_ERR_INVALID, "null argument");
    if (p->integrator != MI355_INTEGRATOR_PTRACER) return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: the particle tracer only");
    if (!n_sensors || n_sensors > MI_MULTI_MAX) return fail(MI355_ERR_INVALID, "1 .. MI355_MAX_SENSORS_PER_PASS (8) sensors");
    if ((p->passes ? p->passes : 1u) != 1u) return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: samples_per_pass is not carried");   // (a shard of the light-sample grid is fine: it is the same for every camera)
    for (uint32_t k = 0; k < n_sensors; ++k) if (sensors[k] >= s->sensors.size()) return fail(MI355_ERR_INVALID, "sensor index out of range");
    const DScene &f = s->sensors[sensors[0]].snap;
    for (uint32_t k = 0; k < n_sensors; ++k) {                              // one lane layout, one film geometry, one filter: what the cameras must share
        const DScene &g = s->sensors[sensors[k]].snap;
        if (g.sensor_type != MI355_SENSOR_PERSPECTIVE) return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: perspective sensors only");
        if (g.film_w != f.film_w || g.film_h != f.film_h || g.crop_w != g.film_w || g.crop_h != g.film_h || g.crop_x || g.crop_y || g.rfilter != f.rfilter || g.filter_radius != f.filter_radius ||
            memcmp(g.filter_coeff, f.filter_coeff, sizeof(f.filter_coeff)) || (g.shutter_open_time > 0.f) != (f.shutter_open_time > 0.f))
            return fail(MI355_ERR_UNSUPPORTED, "several sensors in one pass: the films must have one size (no crop window), one reconstruction filter and one shutter setting");
    }
    mi355_render_params p0 = *p; p0.sensor = sensors[0];
    RenderArgs a; int rc = make_args(s, &p0, &a); if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    if (a.n_local == 0) return MI355_OK;
    hipStream_t st = (hipStream_t) stream; dim3 grid((a.n_local + BLOCK - 1) / BLOCK), block(BLOCK);
    if ((rc = ensure_geometry(s, st))) return rc;
    if (s->emitters.empty()) return MI355_OK;
    uint32_t pitch, x0, virt_w; mi355_multi_film_layout(f.film_w, n_sensors, &pitch, &x0, &virt_w);
    DSensorSet set; memset(&set, 0, sizeof(set)); set.n = n_sensors; set.film_w = (float) f.film_w; set.film_h = (float) f.film_h;
    for (uint32_t k = 0; k < n_sensors; ++k) {
        const DScene &g = s->sensors[sensors[k]].snap; DSensorView &v = set.v[k];
        memcpy(v.cam_to_world, g.cam_to_world, sizeof(v.cam_to_world)); memcpy(v.camera_to_sample, g.camera_to_sample, sizeof(v.camera_to_sample)); memcpy(v.image_rect, g.image_rect, sizeof(v.image_rect));
        v.normalization = g.normalization; v.near_clip = g.near_clip; v.far_clip = g.far_clip; v.x_offset = (float) (k * pitch + x0);
    }
    Timed tm = { s, st, (p->flags & MI355_FLAG_TIMING) != 0, 0 };
    if ((rc = tm.begin())) return rc;
    const float sample_scale = (float) (f.crop_w * f.crop_h) / (float) ((uint64_t) p->spp * f.film_w * f.film_h);
    const DScene real = s->ds;                                              // the kernels and the splat passes see the VIRTUAL film: every camera's columns side by side
    s->ds.crop_w = virt_w;
    SplatQueue q; const uint32_t *order = nullptr;
    rc = splat_queue_setup(s, &p0, a, st, false, q);
    if (!rc) rc = pt_sort_lanes(s, &p0, a, st, &order);
    if (!rc) {
        if (order) grid.x = xcd_grid(grid.x);
        const bool full = s->ds.features != 0u, flat = full && s->ds.n_bvh_shapes == 0u;
        if (flat) hipLaunchKernelGGL((k_ptracer_multi<2>), grid, block, 0, st, s->ds, a, sample_scale, d_film, s->d_counters, q, order, set);
        else if (full) hipLaunchKernelGGL((k_ptracer_multi<true>), grid, block, 0, st, s->ds, a, sample_scale, d_film, s->d_counters, q, order, set);
        else hipLaunchKernelGGL((k_ptracer_multi<false>), grid, block, 0, st, s->ds, a, sample_scale, d_film, s->d_counters, q, order, set);
        if (hipGetLastError() != hipSuccess) rc = fail(MI355_ERR_DEVICE, "k_ptracer_multi launch failed");
    }
    if (!rc) rc = splat_queue_flush(s, q, st, false, d_film);
    s->ds = real;
    if (rc) return rc;
    if ((rc = tm.end())) return rc;
    s->n_gen += a.n_local;
    return MI355_OK;
}
extern "C" int mi355_spot_loss(const float *d_image, uint32_t height, uint32_t width, float *d_out, void *stream) {
    if (!d_image || !d_out) return fail(MI355_ERR_INVALID, "null argument");
    if (!height || !width || (uint64_t) height * width > 0x7fffffffull) return fail(MI355_ERR_INVALID, "bad image size");
    hipLaunchKernelGGL(k_spot_loss, dim3(1), dim3(1024), 0, (hipStream_t) stream, d_image, height, width, d_out);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_spot_loss_backward(const float *d_out, const float *d_grad_loss, uint32_t height, uint32_t width, float *d_grad_image, void *stream) {
    if (!d_out || !d_grad_loss || !d_grad_image) return fail(MI355_ERR_INVALID, "null argument");
    if (!height || !width || (uint64_t) height * width > 0x7fffffffull) return fail(MI355_ERR_INVALID, "bad image size");
    hipLaunchKernelGGL(k_spot_loss_bwd, dim3((height * width + 255u) / 256u), dim3(256), 0, (hipStream_t) stream, d_out, d_grad_loss, height, width, d_grad_image);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_adam_step(const mi355_adam_slot *slots, uint32_t n, double beta_1, double beta_2, double epsilon, void *stream) {
    if (!slots && n) return fail(MI355_ERR_INVALID, "null argument");
    if (!(beta_1 >= 0.0 && beta_1 < 1.0 && beta_2 >= 0.0 && beta_2 < 1.0 && epsilon > 0.0)) return fail(MI355_ERR_INVALID, "Adam: beta_1, beta_2 in [0, 1), epsilon > 0");
    for (uint32_t i = 0; i < n; ++i)
        if (!slots[i].param || !slots[i].grad || !slots[i].m || !slots[i].v || !slots[i].param_out) return fail(MI355_ERR_INVALID, "Adam: null pointer in slot");
    for (uint32_t b = 0; b < n; b += MI_ADAM_SLOTS) {
        AdamTable t; const uint32_t k = n - b < MI_ADAM_SLOTS ? n - b : MI_ADAM_SLOTS;
        memset(&t, 0, sizeof(t)); memcpy(t.slot, slots + b, (size_t) k * sizeof(mi355_adam_slot));
        hipLaunchKernelGGL(k_adam, dim3(k), dim3(256), 0, (hipStream_t) stream, t, (float) beta_1, (float) (1.0 - beta_1), (float) beta_2, (float) (1.0 - beta_2), (float) epsilon);   // (1 - beta formed in double, as the Python scalars of the reference are)
    }
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_trace_rays(mi355_scene *s, const float *d_rays, uint32_t n, uint32_t num_lenses, int32_t max_depth, float *d_out, void *stream) {
    if (!s || !d_rays || !d_out) return fail(MI355_ERR_INVALID, "null argument");
    if (max_depth < -1) return fail(MI355_ERR_INVALID, "\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    if (num_lenses == 0u || num_lenses > 4096u) return fail(MI355_ERR_INVALID, "num_lenses must lie in [1, 4096]");
    HIP_TRY(hipSetDevice(s->device));
    if (!n) return MI355_OK;
    { int rc = ensure_geometry(s, (hipStream_t) stream); if (rc) return rc; }
    hipLaunchKernelGGL(k_trace, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t) stream, s->ds, d_rays, n, num_lenses, (uint32_t) max_depth, d_out);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
extern "C" int mi355_debug_ray_intersect(mi355_scene *s, const float *d_rays, uint32_t n, float *d_hits, void *stream) {
    if (!s || !d_rays || !d_hits) return fail(MI355_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    if (!n) return MI355_OK;
    { int rc = ensure_geometry(s, (hipStream_t) stream); if (rc) return rc; }
    hipLaunchKernelGGL(k_debug_intersect, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, (hipStream_t) stream, s->ds, d_rays, n, d_hits);
    HIP_TRY(hipGetLastError());
    return MI355_OK;
}
