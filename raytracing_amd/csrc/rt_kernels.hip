// rt_kernels.hip — gfx950 kernels of the path tracer.
//
//   render_mega_bvh  persistent wave64 megakernel for BVH scenes.  Every lane is a small
//                    state machine (idle -> traversing -> hit ready); the wave interleaves
//                    traversal steps of its 64 rays with shading phases, so it never idles
//                    for its slowest ray.  Geometry, BVH nodes, per-slot shading records
//                    and the texture table are staged into LDS once per workgroup when they
//                    fit; traversal stacks live in LDS.
//   render_mega      the same persistent scheme with one whole ray per lane per iteration:
//                    used for small scenes that are scanned linearly (uniform control flow).
//   reduce_stage     sums the per-path radiance records of a launch, in sample order, into
//                    the caller's fp32 accumulation image (camera.h:43-47's pixel_color +=).
//   trace_batch      closest hit for a batch of rays (the parity entry point).
//
// Replaces camera::render's triple loop (camera.h:40-47) and everything below it.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_core.h"
#include "rt_launch.h"
#include "rt_sched.h"
#include "rt_stage.h"
#include "rt_wavepool.h"

using namespace rtc;

// This file is compiled three times: as is; with -DRT_LEAN (render kernels only, without the
// quad / noise / image / wrapper branches; launchers suffixed _lean); and with -DRT_INST (the
// render and batch kernels with the two-level traversal of scenes that have instances;
// launchers suffixed _inst).  Only the first build has the kernels that are no traversal
// (reduce_stage, accum_add, tonemap_u8).
#if defined(RT_LEAN)
#define RT_LAUNCH_NAME(x) x##_lean
#elif defined(RT_INST)
#define RT_LAUNCH_NAME(x) x##_inst
#else
#define RT_LAUNCH_NAME(x) x
#define RT_BASE_BUILD 1
#endif


namespace {

// ------------------------------------------------------------------------------------------
// One whole ray per lane per iteration.  Linear scenes: every lane walks the same
// primitive list, so the traversal itself is divergence-free.
// WPS: waves per SIMD the kernel is compiled for.  A scan of many quads does so little arithmetic
// per scalar load and branch that it is bound by each wave's own instruction latency: the build
// for 8 waves per SIMD (64 registers; some shading code spills) is 1.2-1.3x faster there, and
// slower on scenes of a few textured spheres, where the spills cost more than the occupancy gives.
// WAIT: a miss ends its path at once, and hits are shaded together once rp.shade_min of them wait (scenes
// whose shading costs far more than their scan: noise textures).
template <int TRAV, bool LDS_SCENE, bool COUNT, bool RING, int WPS = RT_MEGA_WAVES_PER_SIMD, bool WAIT = false>
__global__ void __launch_bounds__(RT_MEGA_MAX_THREADS, WPS)
render_mega(DevScene sc, DevCamera cam, RenderParams rp) {
    extern __shared__ f4 smem[];
    __shared__ BlockPool block_pool;
    block_pool_init(&block_pool);
    SceneView vw = global_view(sc);
    f4 *free_lds = smem;
    if constexpr (LDS_SCENE) free_lds = stage_scene(sc, smem, vw);
    else __syncthreads();
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;
    constexpr bool SPILL = !LDS_SCENE;
    typedef TravStack<StackT, RT_MEGA_MAX_THREADS> Stack;
    const Stack st = lane_stack<StackT, SPILL, RT_MEGA_MAX_THREADS>(free_lds, sc, rp.stack_lds_entries, rp.stack_spill);
    // RING: the camera rays of the wave's current unit (see render_mega_bvh)
    f4 *ring = RING ? smem + rp.ring_f4 + (size_t)(threadIdx.x >> 6) * RT_RING_F4_PER_WAVE : nullptr;

    const int lane = threadIdx.x & 63;
    const unsigned long long lane_mask_lt = (1ull << lane) - 1ull;
    WavePool wp;
    pool_init(wp);
    bool alive = false, pending = false; // pending: the ray's closest hit is known and waits to be shaded
    TraceHit h;
    h.slot = -1;
    h.t = 0.0f;
    const int shade_min = rp.shade_min;
    PathState ps;
    unsigned pixel = 0, sample = 0, n_rays = 0;
    Cnt<COUNT> cn;
    // instrumented variant: per-wave cycle split {whole loop, trace, shade, iterations, live lanes}
    unsigned long long w_cycles[5] = {0, 0, 0, 0, 0};
    long long t_iter = COUNT ? clock64() : 0;

    for (;;) {
        unsigned long long need = __ballot(!alive);
        if (need != 0ull) {
            const bool new_unit = pool_refill(wp, &block_pool, rp, lane);
            if (wp.k < wp.k_end) {
                if constexpr (RING) {
                    if (new_unit) { // all 64 camera rays of the unit at once, with every lane
                        unsigned g_pixel, g_sample;
                        int gx, gy;
                        const bool valid = pool_decode(wp, rp, (unsigned)lane, g_pixel, g_sample, gx, gy);
                        const PathState g = path_start(cam, rp.seed, g_pixel, gx, gy, (unsigned)rp.sample_begin + g_sample);
                        ring[lane] = f4{g.ray.o.x, g.ray.o.y, g.ray.o.z, g.ray.tm};
                        ring[64 + lane] = f4{g.ray.d.x, g.ray.d.y, g.ray.d.z, __builtin_bit_cast(float, g.rng.s)};
                        ring[128 + lane] = f4{__builtin_bit_cast(float, valid ? g_pixel : ~0u),
                                              __builtin_bit_cast(float, g_sample), 0.0f, 0.0f};
                    }
                }
                unsigned n_need = (unsigned)__popcll(need);
                if (!alive) {
                    unsigned k = wp.k + (unsigned)__popcll(need & lane_mask_lt);
                    if constexpr (RING) {
                        if (k < wp.k_end) {
                            const f4 a0 = ring[k], a1 = ring[64 + k], a2 = ring[128 + k];
                            pixel = __builtin_bit_cast(unsigned, a2.x);
                            sample = __builtin_bit_cast(unsigned, a2.y);
                            if (pixel != ~0u) {
                                ps.ray.o = mk(a0.x, a0.y, a0.z);
                                ps.ray.tm = a0.w;
                                ps.ray.d = mk(a1.x, a1.y, a1.z);
                                ps.rng.s = __builtin_bit_cast(unsigned, a1.w);
                                ps.beta = mk(1.0f, 1.0f, 1.0f);
                                ps.L = mk(0.0f, 0.0f, 0.0f);
                                ps.depth = cam.max_depth;
                                n_rays = 0;
                                alive = true;
                            }
                        }
                    } else {
                        int px, py;
                        unsigned pix, s_local;
                        if (k < wp.k_end && pool_decode(wp, rp, k, pix, s_local, px, py)) {
                            pixel = pix;
                            sample = s_local;
                            ps = path_start(cam, rp.seed, pixel, px, py, (unsigned)rp.sample_begin + s_local);
                            n_rays = 0;
                            alive = true;
                        }
                    }
                }
                wp.k = wp.k + n_need < wp.k_end ? wp.k + n_need : wp.k_end;
            }
        }
        if (!__any(alive)) {
            if (wp.exhausted && wp.k >= wp.k_end) break;
            continue;
        }
        if constexpr (!WAIT) {
            if (alive) {
                bool cont;
                if constexpr (COUNT) {
                    // instrumented variant: split the segment to time trace and shade separately
                    long long c0 = clock64();
                    cont = false;
                    TraceHit h;
                    h.slot = -1;
                    h.t = 0.0f;
                    bool go = ps.depth > 0;
                    if (go) {
                        n_rays++;
                        if constexpr (TRAV == TRAV_LINEAR)
                            h = trace_linear<COUNT>(sc, vw, ps.ray, 0.001f, RT_INF, cn);
                        else
                            h = trace_bvh<COUNT, Stack, SPILL, LDS_SCENE ? RT_NODE_LDS_F4 : RT_NODE_F4>(sc, vw, ps.ray, 0.001f, RT_INF, st, cn);
                    }
                    long long c1 = clock64();
                    if (go) cont = path_after_trace<COUNT>(sc, vw, cam, ps, h, cn);
                    long long c2 = clock64();
                    if (lane == __builtin_ctzll(__ballot(true))) {
                        w_cycles[1] += c1 - c0;
                        w_cycles[2] += c2 - c1;
                        w_cycles[3] += 1;
                        w_cycles[4] += __popcll(__ballot(true));
                    }
                } else {
                    cont = path_step<TRAV, COUNT, Stack, SPILL, LDS_SCENE ? RT_NODE_LDS_F4 : RT_NODE_F4>(sc, vw, cam, ps, st, n_rays, cn);
                }
                if (!cont) {
                    write_record(rp, sample, pixel, ps.L, n_rays);
                    alive = false;
                }
            }
        } else {
            // ---- closest hits of the lanes whose ray has none yet -----------------------------------
            long long c0 = COUNT ? clock64() : 0;
            if (alive && !pending) {
                if (ps.depth <= 0) { // camera.h:141
                    write_record(rp, sample, pixel, ps.L, n_rays);
                    alive = false;
                } else {
                    n_rays++;
                    if constexpr (TRAV == TRAV_LINEAR)
                        h = trace_linear<COUNT>(sc, vw, ps.ray, 0.001f, RT_INF, cn);
                    else
                        h = trace_bvh<COUNT, Stack, SPILL, LDS_SCENE ? RT_NODE_LDS_F4 : RT_NODE_F4>(sc, vw, ps.ray, 0.001f, RT_INF, st, cn);
                    pending = true;
                }
            }
            long long c1 = COUNT ? clock64() : 0;
            // ---- a miss ends its path at once (camera.h:148-161) ------------------------------------
            if (pending && h.slot < 0) {
                path_missed(cam, ps);
                write_record(rp, sample, pixel, ps.L, n_rays);
                alive = false;
                pending = false;
            }
            // ---- hits are shaded together: when shade_min of them wait, when every lane has one, or
            // when no new paths are left to start in the lanes that have none (a scene whose shading is
            // expensive next to its scan - noise textures - then shades with most lanes, and scans for
            // the few that missed in between; shade_min = 1 shades in every iteration)
            const unsigned long long hits = __ballot(pending);
            if (hits != 0ull && (__popcll(hits) >= shade_min || !__any(!alive) || (wp.exhausted && wp.k >= wp.k_end))) {
                if (pending) {
                    pending = false;
                    if (!path_after_hit<COUNT>(sc, vw, ps, h, cn)) {
                        write_record(rp, sample, pixel, ps.L, n_rays);
                        alive = false;
                    }
                }
            }
            if constexpr (COUNT) {
                long long c2 = clock64();
                if (lane == __builtin_ctzll(__ballot(true))) {
                    w_cycles[1] += c1 - c0;
                    w_cycles[2] += c2 - c1;
                    w_cycles[3] += 1;
                    w_cycles[4] += __popcll(__ballot(true));
                }
            }
        }
        if constexpr (COUNT) {
            long long now = clock64();
            if (lane == 0) {
                w_cycles[0] += now - t_iter;
                t_iter = now;
            }
        }
    }
    if constexpr (COUNT) {
        flush_counters<COUNT>(rp.counters, cn);
        for (int k = 0; k < 5; k++) {
            unsigned long long v = w_cycles[k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) atomicAdd(&rp.counters->sched[k], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
// BVH scenes: traversal and shading are decoupled (see the header comment).
// COOP: scenes read from HBM whose waves read their nodes together (rt_sched.h coop_fetch_nodes).
template <bool LDS_SCENE, bool COUNT, bool RING, bool COOP = false>
__global__ void __launch_bounds__(RT_MEGA_MAX_THREADS, RT_MEGA_WAVES_PER_SIMD)
render_mega_bvh(DevScene sc, DevCamera cam, RenderParams rp) {
    extern __shared__ f4 smem[];
    __shared__ BlockPool block_pool;
    block_pool_init(&block_pool);
    SceneView vw = global_view(sc);
    f4 *free_lds = smem;
    if constexpr (LDS_SCENE) free_lds = stage_scene(sc, smem, vw);
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;
    if constexpr (!LDS_SCENE) { // the scene stays in HBM: the top of the tree comes along
        stage_nodes_to_lds(free_lds, sc.nodes, sc.n_top);
        vw.top = free_lds;
        vw.n_top = sc.n_top;
        free_lds += (size_t)sc.n_top * RT_NODE_LDS_F4;
        __syncthreads();
    }
    constexpr bool SPILL = !LDS_SCENE;
    typedef TravStack<StackT, RT_MEGA_MAX_THREADS> Stack;
    const Stack st = lane_stack<StackT, SPILL, RT_MEGA_MAX_THREADS>(free_lds, sc, rp.stack_lds_entries, rp.stack_spill);
    // RING: the camera rays of the wave's current unit, structure of arrays over its 64 paths
    f4 *ring = RING ? smem + rp.ring_f4 + (size_t)(threadIdx.x >> 6) * RT_RING_F4_PER_WAVE : nullptr;
    f4 *coop_stage = COOP ? smem + rp.coop_f4 + (size_t)(threadIdx.x >> 6) * RT_COOP_F4_PER_WAVE : nullptr;

    const int lane = threadIdx.x & 63;
    const unsigned long long lane_mask_lt = (1ull << lane) - 1ull;
    const int min_ready = rp.min_ready;
    const int leaf_min = rp.leaf_min;
    const int blocked_min = rp.blocked_min;
    const int regen_min = rp.regen_min;
    Parked pend;

    WavePool wp;
    pool_init(wp);
    int state = ST_IDLE;
    PathState ps;
    Trav tv;
    unsigned pixel = 0, sample = 0, n_rays = 0;
    Cnt<COUNT> cn;
    const int waves_per_block = blockDim.x >> 6;
    // wave-level schedule statistics (instrumented variant only), kept by lane 0
    unsigned long long w_s_phases = 0, w_s_lanes = 0;
    SchedStats ss;
    long long c_shade = 0, c_regen = 0, c_mark = 0;
    const long long c_begin = COUNT ? clock64() : 0;
    unsigned long long w_start = 0, w_dry = 0;
    if constexpr (COUNT) w_start = wall_clock64();
    const bool timeline = rp.timeline != nullptr;
    unsigned long long tl_start = timeline ? wall_clock64() : 0ull, tl_dry = 0ull;
#define RT_TICK() if constexpr (COUNT) c_mark = clock64()
#define RT_TOCK(acc) if constexpr (COUNT) acc += clock64() - c_mark

    // The wave's loop exists twice: the first copy runs while the launch still has new paths to
    // hand out; the second copy takes over when the pool has run dry, and contains no code that
    // starts paths.
    auto run_phase = [&](auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    for (;;) {
        if constexpr (COUNT) {
            if (w_dry == 0 && wp.exhausted && wp.k >= wp.k_end) w_dry = wall_clock64();
            unsigned long long rm = __ballot(state == ST_READY);
            if (rm) {
                w_s_phases++;
                w_s_lanes += __popcll(rm);
            }
        }
        // ---- shade the lanes whose closest hit is known ---------------------------------------
        RT_MARK("shade");
        RT_TICK();
        bool begin = false; // the lane has a new ray: scattered here, or a camera ray below
        if (state == ST_READY) {
            TraceHit h;
            h.t = tv.best_t;
            h.slot = tv.best_slot;
            if (RT_FEAT_INSTANCE) h.inst = tv.best_inst;
            bool cont = path_after_trace<COUNT>(sc, vw, cam, ps, h, cn);
            if (cont) {
                n_rays++;
                begin = true;
                state = ST_TRAV;
            } else {
                write_record(rp, sample, pixel, ps.L, n_rays);
                state = ST_IDLE;
            }
        }
        RT_TOCK(c_shade);
        RT_TICK();
        RT_MARK("start");
        // ---- hand new paths to idle lanes -------------------------------------------------------
        unsigned long long need = TAIL ? 0ull : __ballot(state == ST_IDLE);
        // starting paths (camera ray, sampler seed) is batched like everything else: wait
        // until regen_min lanes are idle unless nothing else is left to do
        if (!TAIL && need != 0ull && (__popcll(need) >= regen_min || !__any(state == ST_TRAV))) {
            // the idle lanes take consecutive paths; a unit that runs out half-way is followed
            // by the next one at once, so that all of them start together below
            bool want = state == ST_IDLE, start = false;
            int px = 0, py = 0;
            for (int pass = 0; pass < 2 && need != 0ull; pass++) {
                const bool new_unit = pool_refill(wp, &block_pool, rp, lane);
                if (!(wp.k < wp.k_end)) break;
                if constexpr (RING) {
                    if (new_unit) {
                        // all 64 camera rays of the unit at once, with every lane (starting a handful
                        // of paths at a time costs the same instructions for a few lanes each time)
                        unsigned g_pixel, g_sample;
                        int gx, gy;
                        const bool valid = pool_decode(wp, rp, (unsigned)lane, g_pixel, g_sample, gx, gy);
                        const PathState g = path_start(cam, rp.seed, g_pixel, gx, gy, (unsigned)rp.sample_begin + g_sample);
                        ring[lane] = f4{g.ray.o.x, g.ray.o.y, g.ray.o.z, g.ray.tm};
                        ring[64 + lane] = f4{g.ray.d.x, g.ray.d.y, g.ray.d.z, __builtin_bit_cast(float, g.rng.s)};
                        ring[128 + lane] = f4{__builtin_bit_cast(float, valid ? g_pixel : ~0u),
                                              __builtin_bit_cast(float, g_sample), 0.0f, 0.0f};
                    }
                }
                const unsigned avail = wp.k_end - wp.k;
                const unsigned n_need = (unsigned)__popcll(need);
                const unsigned r = (unsigned)__popcll(need & lane_mask_lt);
                const bool mine = want && r < avail;
                if (mine) {
                    if constexpr (RING) {
                        const f4 a0 = ring[wp.k + r], a1 = ring[64 + wp.k + r], a2 = ring[128 + wp.k + r];
                        pixel = __builtin_bit_cast(unsigned, a2.x);
                        sample = __builtin_bit_cast(unsigned, a2.y);
                        start = pixel != ~0u;
                        ps.ray.o = mk(a0.x, a0.y, a0.z);
                        ps.ray.tm = a0.w;
                        ps.ray.d = mk(a1.x, a1.y, a1.z);
                        ps.rng.s = __builtin_bit_cast(unsigned, a1.w);
                    } else {
                        start = pool_decode(wp, rp, wp.k + r, pixel, sample, px, py);
                    }
                    want = false;
                }
                wp.k += n_need < avail ? n_need : avail;
                need &= ~__ballot(mine);
            }
            if (start) {
                if constexpr (RING) { // (the host does not pick this variant for max_depth <= 0)
                    ps.beta = mk(1.0f, 1.0f, 1.0f);
                    ps.L = mk(0.0f, 0.0f, 0.0f);
                    ps.depth = cam.max_depth;
                } else {
                    ps = path_start(cam, rp.seed, pixel, px, py, (unsigned)rp.sample_begin + sample);
                }
                if (ps.depth > 0) {
                    n_rays = 1;
                    begin = true;
                    state = ST_TRAV;
                } else { // max_depth 0: camera.h:141
                    write_record(rp, sample, pixel, mk(0.0f, 0.0f, 0.0f), 0u);
                }
            }
        }
        // scattered rays and camera rays set their traversal up together (more lanes per instruction)
        RT_MARK("begin");
        if (begin) {
            trav_begin(tv, ps.ray, RT_INF);
            trav_test_big<COUNT>(tv, sc, vw, ps.ray, 0.001f, cn);
        }
        RT_TOCK(c_regen);
        RT_MARK("loop");
        const bool pool_dry = TAIL || (wp.exhausted && wp.k >= wp.k_end);
        if constexpr (!TAIL) {
            if (pool_dry) return; // on to the tail copy of the loop
        }
        if (!__any(state != ST_IDLE)) {
            if (pool_dry) break;
            continue;
        }
        // ---- traversal steps until enough lanes wait for shading ------------------------------
        wave_traverse<LDS_SCENE, COUNT, Stack, SPILL, COOP>(state, tv, pend, ps.ray, sc, vw, st, 0.001f, min_ready, leaf_min,
                                                             blocked_min, !pool_dry, cn, ss, coop_stage);
    }
    };
    run_phase(std::false_type{});
    if (timeline) tl_dry = wall_clock64();
    run_phase(std::true_type{});
    if (timeline && lane == 0) {
        unsigned long long *t = rp.timeline + 4u * ((size_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6));
        t[0] = tl_start;
        t[1] = tl_dry;
        t[2] = wall_clock64();
        // HW_REG_HW_ID (4) and HW_REG_XCC_ID (20), all 32 bits of each
        t[3] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) |
               ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
    }
    if constexpr (COUNT) {
        flush_counters<COUNT>(rp.counters, cn);
        if (lane == 0) {
            atomicAdd(&rp.counters->sched[0], ss.t_iters);
            atomicAdd(&rp.counters->sched[1], ss.t_lanes);
            atomicAdd(&rp.counters->sched[2], ss.l_iters);
            atomicAdd(&rp.counters->sched[3], ss.l_lanes);
            atomicAdd(&rp.counters->sched[4], w_s_phases);
            atomicAdd(&rp.counters->sched[5], w_s_lanes);
            atomicAdd(&rp.counters->cycles[0], (unsigned long long)(clock64() - c_begin));
            atomicAdd(&rp.counters->cycles[1], (unsigned long long)c_shade);
            atomicAdd(&rp.counters->cycles[2], (unsigned long long)c_regen);
            atomicAdd(&rp.counters->coherence[0], ss.t_lanes);
            atomicAdd(&rp.counters->coherence[1], ss.same_node);
            atomicAdd(&rp.counters->cycles[3], (unsigned long long)ss.c_inner);
            atomicAdd(&rp.counters->cycles[4], (unsigned long long)ss.c_leaf);
            atomicMin(&rp.counters->tail[0], w_start);
            atomicMin(&rp.counters->tail[1], w_dry ? w_dry : ~0ull);
            atomicMax(&rp.counters->tail[2], w_dry);
            atomicMax(&rp.counters->tail[3], (unsigned long long)wall_clock64());
        }
    }
}

#undef RT_TICK
#undef RT_TOCK

#ifdef RT_BASE_BUILD
// accum[pixel] += sum over the launch's samples, in sample order; ray counts are
// summed per workgroup and added to *ray_count.
__global__ void __launch_bounds__(256) reduce_stage(const f4 *stage, int n_pixels, int sample_count,
                                                    float *accum, unsigned long long *ray_count,
                                                    unsigned *strip_cost) {
    __shared__ unsigned long long wave_rays[4];
    int pixel = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long rays = 0;
    unsigned long long squares = 0; // sum of (rays of a path)^2: long paths weigh more than many short ones
    if (pixel < n_pixels) {
        float r = 0.0f, g = 0.0f, b = 0.0f;
        for (int s = 0; s < sample_count; s++) {
            f4 v = stage[(size_t)s * n_pixels + pixel];
            r += v.x;
            g += v.y;
            b += v.z;
            const unsigned n = __builtin_bit_cast(unsigned, v.w);
            rays += n;
            squares += (unsigned long long)n * n;
        }
        accum[3 * (size_t)pixel + 0] += r;
        accum[3 * (size_t)pixel + 1] += g;
        accum[3 * (size_t)pixel + 2] += b;
    }
    for (int off = 32; off > 0; off >>= 1) {
        rays += __shfl_down(rays, off, 64);
        squares += __shfl_down(squares, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        wave_rays[threadIdx.x >> 6] = rays;
        // a wave covers one 64-pixel strip: its cost for the next launch's order
        if (strip_cost && pixel < n_pixels) strip_cost[pixel >> 6] = squares < 0xffffffffull ? (unsigned)squares : 0xffffffffu;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(ray_count, wave_rays[0] + wave_rays[1] + wave_rays[2] + wave_rays[3]);
}

// dst[i] += src[i]: the per-GPU accumulation images of a multi-GPU render, summed in device order.
__global__ void __launch_bounds__(256) accum_add(float *dst, const float *src, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// write_color (color.h:17-32) for one colour component per lane.
__global__ void __launch_bounds__(256) tonemap_u8(const float *accum, long long n, float scale, uint8_t *rgb8) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = scale * accum[i];
    float g = c > 0.0f ? sqrtf(c) : 0.0f;
    g = g < 0.000f ? 0.000f : (g > 0.999f ? 0.999f : g);
    rgb8[i] = (uint8_t) int(256 * g);
}

#endif // RT_BASE_BUILD
#ifndef RT_LEAN
// hit_record of the closest hit (hittable.h:10-26) plus the primitive id.
__device__ __forceinline__ rt_hit make_hit_record(const DevScene &sc, const SceneView &vw, const Ray &r, TraceHit h) {
    rt_hit out;
    out.prim = -1;
    out.front_face = 0;
    out.t = 0.0f;
    out.p[0] = out.p[1] = out.p[2] = 0.0f;
    out.normal[0] = out.normal[1] = out.normal[2] = 0.0f;
    out.u = out.v = 0.0f;
    if (h.slot >= 0) {
        Surf s = surface_at(sc, vw, r, h);
        out.prim = s.rec.w;
        out.front_face = s.front ? 1 : 0;
        out.t = h.t;
        out.p[0] = s.p.x;
        out.p[1] = s.p.y;
        out.p[2] = s.p.z;
        out.normal[0] = s.n.x;
        out.normal[1] = s.n.y;
        out.normal[2] = s.n.z;
        if (RT_KIND_PRIM(s.rec.x) == RT_PRIM_SPHERE) { // sphere.h:70
            sphere_uv(s.outward, out.u, out.v);
        } else {
            out.u = s.u;
            out.v = s.v;
        }
    }
    return out;
}

// Closest hit for a batch of rays, one whole ray per lane, scene read from global memory.
// IDS: only the closest primitive and its parameter are written, as two arrays (ids: -1 for a miss).
template <int TRAV, bool IDS = false>
__global__ void __launch_bounds__(RT_TRACE_THREADS)
trace_batch(DevScene sc, const rt_ray *rays, long long n_rays, float tmin, float tmax, rt_hit *hits, int *prim_out,
            float *t_out) {
    extern __shared__ f4 smem[];
    typedef TravStack<int, RT_TRACE_THREADS> Stack;
    const Stack st = lane_stack<int, false, RT_TRACE_THREADS>(smem, sc, sc.stack_cap, nullptr);
    const SceneView vw = global_view(sc);
    Cnt<false> cn;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_rays;
         k += (long long)gridDim.x * blockDim.x) {
        rt_ray in = rays[k];
        Ray r;
        r.o = mk(in.orig[0], in.orig[1], in.orig[2]);
        r.d = mk(in.dir[0], in.dir[1], in.dir[2]);
        r.tm = in.tm;
        TraceHit h;
        if constexpr (TRAV == TRAV_LINEAR)
            h = trace_linear<false>(sc, vw, r, tmin, tmax, cn);
        else
            h = trace_bvh<false, Stack, false>(sc, vw, r, tmin, tmax, st, cn);
        if constexpr (IDS) {
            prim_out[k] = h.slot < 0 ? -1 : vw.rec_i[h.slot].w;
            t_out[k] = h.t;
        } else {
            hits[k] = make_hit_record(sc, vw, r, h);
        }
    }
}

// The same query through the render kernel's own traversal: the launch shape, the LDS copy of
// the scene (or of the tree's top), the 16-bit / spilling stacks and the decoupled scheduling of
// render_mega_bvh (wave_traverse: blocks of node steps, parked leaves), with the caller's rays
// in place of paths.  A wave takes a run of consecutive rays at a time from a global counter (64 for
// small batches, up to 2048 for large ones: with 64 the one counter is what a 4 x 10^7-ray batch
// waits for, 12 ns per wave and run); lanes whose ray is done write its record and take the next one.
template <bool LDS_SCENE, bool COOP = false, bool IDS = false>
__global__ void __launch_bounds__(RT_MEGA_MAX_THREADS, RT_MEGA_WAVES_PER_SIMD)
trace_sched(DevScene sc, RenderParams rp, const rt_ray *rays, long long n_rays, float tmin, float tmax, rt_hit *hits,
            int *prim_out, float *t_out) {
    extern __shared__ f4 smem[];
    SceneView vw = global_view(sc);
    f4 *free_lds = smem;
    if constexpr (LDS_SCENE) free_lds = stage_scene(sc, smem, vw);
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;
    if constexpr (!LDS_SCENE) {
        stage_nodes_to_lds(free_lds, sc.nodes, sc.n_top);
        vw.top = free_lds;
        vw.n_top = sc.n_top;
        free_lds += (size_t)sc.n_top * RT_NODE_LDS_F4;
        __syncthreads();
    }
    constexpr bool SPILL = !LDS_SCENE;
    typedef TravStack<StackT, RT_MEGA_MAX_THREADS> Stack;
    const Stack st = lane_stack<StackT, SPILL, RT_MEGA_MAX_THREADS>(free_lds, sc, rp.stack_lds_entries, rp.stack_spill);
    f4 *coop_stage = COOP ? smem + rp.coop_f4 + (size_t)(threadIdx.x >> 6) * RT_COOP_F4_PER_WAVE : nullptr;
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_mask_lt = (1ull << lane) - 1ull;
    // rays per run: about eight runs per wave of the grid, whole multiples of 64
    const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
    long long run = (n_rays / (n_waves * 8) + 63) / 64 * 64;
    run = run < 64 ? 64 : (run > 2048 ? 2048 : run);
    const unsigned long long n_batches = (unsigned long long)((n_rays + run - 1) / run);
    unsigned long long *next_batch = reinterpret_cast<unsigned long long *>(rp.next_chunk);
    long long k = 0, k_end = 0; // the wave's current batch of rays
    bool exhausted = false;
    int state = ST_IDLE;
    Parked pend;
    long long mine = -1;
    Ray ray;
    Trav tv;
    Cnt<false> cn;
    SchedStats ss;
    for (;;) {
        if (state == ST_READY) {
            TraceHit h;
            h.t = tv.best_t;
            h.slot = tv.best_slot;
            if (RT_FEAT_INSTANCE) h.inst = tv.best_inst;
            if constexpr (IDS) {
                prim_out[mine] = h.slot < 0 ? -1 : vw.rec_i[h.slot].w;
                t_out[mine] = h.t;
            } else {
                hits[mine] = make_hit_record(sc, vw, ray, h);
            }
            state = ST_IDLE;
        }
        // idle lanes take the next rays of the batch; an empty batch is replaced first
        unsigned long long need = __ballot(state == ST_IDLE);
        bool want = state == ST_IDLE;
        for (int pass = 0; pass < 2 && need != 0ull && !exhausted; pass++) {
            if (k >= k_end) {
                unsigned long long b = 0;
                if (lane == 0) b = atomicAdd(next_batch, 1ull);
                b = ((unsigned long long)wave_bcast32((unsigned)(b >> 32)) << 32) | wave_bcast32((unsigned)b);
                if (b >= n_batches) {
                    exhausted = true;
                    break;
                }
                k = (long long)b * run;
                k_end = k + run < n_rays ? k + run : n_rays;
            }
            const long long avail = k_end - k;
            const int r = __popcll(need & lane_mask_lt);
            const bool take = want && r < avail;
            if (take) {
                mine = k + r;
                const rt_ray in = rays[mine];
                ray.o = mk(in.orig[0], in.orig[1], in.orig[2]);
                ray.d = mk(in.dir[0], in.dir[1], in.dir[2]);
                ray.tm = in.tm;
                trav_begin(tv, ray, tmax);
                trav_test_big<false>(tv, sc, vw, ray, tmin, cn);
                pend = Parked();
                state = ST_TRAV;
                want = false;
            }
            const int n_need = __popcll(need);
            k += n_need < avail ? n_need : avail;
            need &= ~__ballot(take);
        }
        if (!__any(state != ST_IDLE)) {
            if (exhausted) break;
            continue;
        }
        wave_traverse<LDS_SCENE, false, Stack, SPILL, COOP>(state, tv, pend, ray, sc, vw, st, tmin, rp.min_ready, rp.leaf_min,
                                                             rp.blocked_min, !exhausted, cn, ss, coop_stage);
    }
}

#endif // !RT_LEAN

} // namespace

// ------------------------------------------------------------------------------------------
// Host-side launchers (called from rt_host.cpp)
namespace rtl {

template <int TRAV, bool LDS_SCENE, bool RING>
static hipError_t launch_mega_tcr(const DevScene &sc, const DevCamera &cam, const RenderParams &rp,
                                  const LaunchGeom &lg, hipStream_t stream) {
    if (rp.counters)
        hipLaunchKernelGGL((render_mega<TRAV, LDS_SCENE, true, RING>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                           stream, sc, cam, rp);
    else if (TRAV == TRAV_LINEAR && lg.dense)
        hipLaunchKernelGGL((render_mega<TRAV, LDS_SCENE, false, RING, TRAV == TRAV_LINEAR ? 8 : RT_MEGA_WAVES_PER_SIMD>),
                           dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, cam, rp);
    else if (TRAV == TRAV_LINEAR && rp.shade_min > 1)
        hipLaunchKernelGGL((render_mega<TRAV, LDS_SCENE, false, RING, RT_MEGA_WAVES_PER_SIMD, TRAV == TRAV_LINEAR>),
                           dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, cam, rp);
    else
        hipLaunchKernelGGL((render_mega<TRAV, LDS_SCENE, false, RING>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                           stream, sc, cam, rp);
    return hipGetLastError();
}
template <int TRAV, bool LDS_SCENE>
static hipError_t launch_mega_tc(const DevScene &sc, const DevCamera &cam, const RenderParams &rp,
                                 const LaunchGeom &lg, hipStream_t stream) {
    return lg.ring ? launch_mega_tcr<TRAV, LDS_SCENE, true>(sc, cam, rp, lg, stream)
                   : launch_mega_tcr<TRAV, LDS_SCENE, false>(sc, cam, rp, lg, stream);
}

template <bool LDS_SCENE, bool RING>
static hipError_t launch_mega_bvh_tc(const DevScene &sc, const DevCamera &cam, const RenderParams &rp,
                                     const LaunchGeom &lg, hipStream_t stream) {
    if (rp.counters)
        hipLaunchKernelGGL((render_mega_bvh<LDS_SCENE, true, RING>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                           stream, sc, cam, rp);
    else
        hipLaunchKernelGGL((render_mega_bvh<LDS_SCENE, false, RING>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                           stream, sc, cam, rp);
    return hipGetLastError();
}

hipError_t RT_LAUNCH_NAME(launch_mega)(const DevScene &sc, const DevCamera &cam, const RenderParams &rp, const LaunchGeom &lg,
                       hipStream_t stream) {
    if (sc.traversal == TRAV_LINEAR)
        return lg.lds_scene ? launch_mega_tc<TRAV_LINEAR, true>(sc, cam, rp, lg, stream)
                            : launch_mega_tc<TRAV_LINEAR, false>(sc, cam, rp, lg, stream);
    if (rp.scheduler == 0)
        return lg.lds_scene ? launch_mega_tc<TRAV_BVH, true>(sc, cam, rp, lg, stream)
                            : launch_mega_tc<TRAV_BVH, false>(sc, cam, rp, lg, stream);
#ifndef RT_INST
    if (lg.coop && !lg.lds_scene) { // (the host only sets it together with the rings; never for two-level scenes)
        if (rp.counters)
            hipLaunchKernelGGL((render_mega_bvh<false, true, true, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                               stream, sc, cam, rp);
        else
            hipLaunchKernelGGL((render_mega_bvh<false, false, true, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                               stream, sc, cam, rp);
        return hipGetLastError();
    }
#endif
    // the ring variant starts every path with depth > 0 (camera.h:141 is the other one's business)
    if (lg.ring && cam.max_depth > 0)
        return lg.lds_scene ? launch_mega_bvh_tc<true, true>(sc, cam, rp, lg, stream)
                            : launch_mega_bvh_tc<false, true>(sc, cam, rp, lg, stream);
    return lg.lds_scene ? launch_mega_bvh_tc<true, false>(sc, cam, rp, lg, stream)
                        : launch_mega_bvh_tc<false, false>(sc, cam, rp, lg, stream);
}

#ifdef RT_BASE_BUILD
int mega_waves_per_simd() { return RT_MEGA_WAVES_PER_SIMD; }

// Workgroups of the decoupled render kernel the runtime will keep resident per CU.
int mega_blocks_per_cu(int block, int lds_bytes, bool lds_scene) {
    int n = 0;
    hipError_t e = lds_scene ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_mega_bvh<true, false, false>, block, lds_bytes)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_mega_bvh<false, false, false>, block, lds_bytes);
    return e == hipSuccess ? n : -1;
}
#endif // RT_BASE_BUILD

hipError_t RT_LAUNCH_NAME(set_mega_lds_limit)(int bytes) {
    hipError_t e = hipSuccess;
#define RT_SET(TR, L, C)                                                                                     \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega<TR, L, C, false>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                          \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega<TR, L, C, true>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#define RT_SET8(L, R)                                                                                        \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega<TRAV_LINEAR, L, false, R, 8>,                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    RT_SET8(true, true)
    RT_SET8(true, false)
    RT_SET8(false, true)
    RT_SET8(false, false)
#undef RT_SET8
#define RT_SETW(L, R)                                                                                        \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega<TRAV_LINEAR, L, false, R, RT_MEGA_WAVES_PER_SIMD, true>, \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    RT_SETW(true, true)
    RT_SETW(true, false)
    RT_SETW(false, true)
    RT_SETW(false, false)
#undef RT_SETW
    RT_SET(TRAV_LINEAR, true, true)
    RT_SET(TRAV_LINEAR, true, false)
    RT_SET(TRAV_LINEAR, false, true)
    RT_SET(TRAV_LINEAR, false, false)
    RT_SET(TRAV_BVH, true, true)
    RT_SET(TRAV_BVH, true, false)
    RT_SET(TRAV_BVH, false, true)
    RT_SET(TRAV_BVH, false, false)
#undef RT_SET
#define RT_SET2(L, C)                                                                                        \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega_bvh<L, C, false>,                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);                          \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega_bvh<L, C, true>,                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    RT_SET2(true, true)
    RT_SET2(true, false)
    RT_SET2(false, true)
    RT_SET2(false, false)
#undef RT_SET2
#ifndef RT_INST
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)render_mega_bvh<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)render_mega_bvh<false, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#endif
#ifndef RT_LEAN
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_sched<true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_sched<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_sched<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_batch<TRAV_BVH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_sched<false>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#ifndef RT_INST
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_sched<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#endif
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)trace_batch<TRAV_BVH>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
#endif
    return e;
}

#ifdef RT_BASE_BUILD
hipError_t launch_reduce(const f4 *stage, int n_pixels, int sample_count, float *accum,
                         unsigned long long *ray_count, unsigned *strip_cost, hipStream_t stream) {
    int grid = (n_pixels + 255) / 256;
    hipLaunchKernelGGL(reduce_stage, dim3(grid), dim3(256), 0, stream, stage, n_pixels, sample_count, accum,
                       ray_count, strip_cost);
    return hipGetLastError();
}

hipError_t launch_accum_add(float *dst, const float *src, long long n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(accum_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float *accum, long long n_components, float scale, uint8_t *rgb8,
                          hipStream_t stream) {
    if (n_components <= 0) return hipSuccess;
    long long grid = (n_components + 255) / 256;
    hipLaunchKernelGGL(tonemap_u8, dim3((unsigned)grid), dim3(256), 0, stream, accum, n_components, scale, rgb8);
    return hipGetLastError();
}

#endif // RT_BASE_BUILD
#ifndef RT_LEAN
hipError_t RT_LAUNCH_NAME(launch_trace)(const DevScene &sc, const rt_ray *rays, long long n, float tmin, float tmax,
                        rt_hit *hits, hipStream_t stream) {
    long long want = (n + RT_TRACE_THREADS - 1) / RT_TRACE_THREADS;
    int grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    const int lds = sc.traversal == TRAV_LINEAR ? 16 : sc.stack_cap * RT_TRACE_THREADS * (int)sizeof(int);
    int *const no_ids = nullptr;
    float *const no_t = nullptr;
    if (sc.traversal == TRAV_LINEAR)
        hipLaunchKernelGGL((trace_batch<TRAV_LINEAR>), dim3(grid), dim3(RT_TRACE_THREADS), lds, stream, sc, rays,
                           n, tmin, tmax, hits, no_ids, no_t);
    else
        hipLaunchKernelGGL((trace_batch<TRAV_BVH>), dim3(grid), dim3(RT_TRACE_THREADS), lds, stream, sc, rays, n,
                           tmin, tmax, hits, no_ids, no_t);
    return hipGetLastError();
}
// The closest primitive and parameter only.  Scenes with a tree go through the render kernel's traversal
// (rp, lg as for launch_trace_sched), scanned scenes through the whole-ray kernel.
hipError_t RT_LAUNCH_NAME(launch_closest)(const DevScene &sc, const RenderParams &rp, const LaunchGeom &lg, const rt_ray *rays,
                          long long n, float tmin, float tmax, int *prim, float *t, hipStream_t stream) {
    rt_hit *const no_hits = nullptr;
    if (sc.traversal == TRAV_LINEAR) {
        long long want = (n + RT_TRACE_THREADS - 1) / RT_TRACE_THREADS;
        int grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
        hipLaunchKernelGGL((trace_batch<TRAV_LINEAR, true>), dim3(grid), dim3(RT_TRACE_THREADS), 16, stream, sc, rays, n,
                           tmin, tmax, no_hits, prim, t);
    } else if (lg.lds_scene) {
        hipLaunchKernelGGL((trace_sched<true, false, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, rp,
                           rays, n, tmin, tmax, no_hits, prim, t);
    } else {
        hipLaunchKernelGGL((trace_sched<false, false, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, rp,
                           rays, n, tmin, tmax, no_hits, prim, t);
    }
    return hipGetLastError();
}

// rp: stack_lds_entries, stack_spill, min_ready / leaf_min / blocked_min and next_chunk (an
// 8-byte counter, zeroed) are used; lg is the scene's render geometry.
hipError_t RT_LAUNCH_NAME(launch_trace_sched)(const DevScene &sc, const RenderParams &rp, const LaunchGeom &lg, const rt_ray *rays,
                              long long n, float tmin, float tmax, rt_hit *hits, hipStream_t stream) {
    if (lg.lds_scene)
        hipLaunchKernelGGL((trace_sched<true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, rp, rays, n,
                           tmin, tmax, hits, (int *)nullptr, (float *)nullptr);
#ifndef RT_INST
    else if (lg.coop)
        hipLaunchKernelGGL((trace_sched<false, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, rp, rays, n,
                           tmin, tmax, hits, (int *)nullptr, (float *)nullptr);
#endif
    else
        hipLaunchKernelGGL((trace_sched<false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, rp, rays, n,
                           tmin, tmax, hits, (int *)nullptr, (float *)nullptr);
    return hipGetLastError();
}
#endif // !RT_LEAN

} // namespace rtl
