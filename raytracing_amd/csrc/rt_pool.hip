// rt_pool.hip — render_pool_bvh: the persistent megakernel with a per-wave pool of paths in LDS.
//
// A wave owns more paths than lanes (rp.pool_slots, normally 128).  A path's state lives in
// an 80-byte LDS slot; the 64 lanes are *engines* that hold only a ray and its traversal
// state in registers:
//
//   dispatch   idle engines take rays from the wave's list of pending slots
//   traverse   node steps / parked leaf tests as in render_mega_bvh; an engine that finishes
//              writes the hit into its slot and appends the slot to the ready list
//   shade      runs only when 64 slots are ready (or nothing else is left): lane i loads the
//              i-th ready slot — any lane can shade any slot — shades it, and either queues
//              the scattered ray or starts a new camera path in the same slot
//
// With 128 slots and a shading threshold of 64 there are always at least 64 rays to trace, so
// both the traversal and the shading run with (nearly) all lanes active; render_mega_bvh,
// where a lane owns one path, runs both at about 38 of 64.  All list updates are wave-level
// (ballot + prefix count into LDS arrays): no atomics, no barriers after the scene is staged.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_core.h"
#include "rt_launch.h"
#include "rt_stage.h"
#include "rt_wavepool.h"

using namespace rtc;

#ifndef RT_INNER_UNROLL
#define RT_INNER_UNROLL 4
#endif

namespace {

// Slot layout (RT_POOL_SLOT_F4 x 16 bytes):
//   [0] origin, time   [1] direction, hit t   [2] throughput, sampler state
//   [3] radiance, bounces left | rays traced << 16   [4] pixel, local sample, hit slot, -
struct WaveLists {
    unsigned short *pend;  // slots whose ray waits for an engine
    unsigned short *ready; // slots whose closest hit is known
    int n_pend, n_ready;   // wave-uniform
};

// Appends `slot` for the lanes with pred == true (wave-level, in lane order).
__device__ __forceinline__ void list_push(unsigned short *list, int &n, bool pred, int slot,
                                          unsigned long long lane_mask_lt) {
    unsigned long long m = __ballot(pred);
    if (pred) list[n + __popcll(m & lane_mask_lt)] = (unsigned short)slot;
    n += __popcll(m);
}

// Starts new camera paths in the slots of the lanes with want == true, as long as the launch
// has paths left; the started slots are appended to the pending list.  Returns the number
// of paths started (wave-uniform).
template <bool COUNT>
__device__ __forceinline__ int start_paths(bool want, int slot, f4 *pool, WaveLists &wl, WavePool &wp,
                                           BlockPool *bp, const DevCamera &cam, const RenderParams &rp, int lane,
                                           unsigned long long lane_mask_lt) {
    int started = 0;
    for (;;) {
        unsigned long long need = __ballot(want);
        if (need == 0ull) break;
        pool_refill(wp, bp, rp, lane);
        if (!(wp.k < wp.k_end)) {
            if (wp.exhausted) break; // the launch has no paths left
            continue;                // the workgroup's next run is on its way: a slot left empty stays empty
        }
        unsigned n_need = (unsigned)__popcll(need);
        bool begun = false;
        if (want) {
            unsigned k = wp.k + (unsigned)__popcll(need & lane_mask_lt);
            int px, py;
            unsigned pix, s_local;
            if (k < wp.k_end && pool_decode(wp, rp, k, pix, s_local, px, py)) {
                PathState ps = path_start(cam, rp.seed, pix, px, py, (unsigned)rp.sample_begin + s_local);
                if (ps.depth > 0) {
                    f4 *s = pool + slot * RT_POOL_SLOT_F4;
                    s[0] = f4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tm};
                    s[1] = f4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
                    s[2] = f4{1.0f, 1.0f, 1.0f, __builtin_bit_cast(float, ps.rng.s)};
                    s[3] = f4{0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, ((unsigned)ps.depth & 0xffffu) | (1u << 16))};
                    s[4] = f4{__builtin_bit_cast(float, pix), __builtin_bit_cast(float, s_local), 0.0f, 0.0f};
                    begun = true;
                    want = false;
                } else { // max_depth 0: camera.h:141
                    write_record(rp, s_local, pix, mk(0.0f, 0.0f, 0.0f), 0u);
                }
            }
        }
        wp.k = wp.k + n_need < wp.k_end ? wp.k + n_need : wp.k_end;
        unsigned long long bm = __ballot(begun);
        if (begun) wl.pend[wl.n_pend + __popcll(bm & lane_mask_lt)] = (unsigned short)slot;
        wl.n_pend += __popcll(bm);
        started += __popcll(bm);
    }
    return started;
}

template <bool LDS_SCENE, bool COUNT>
__global__ void __launch_bounds__(RT_POOL_MAX_THREADS, RT_POOL_WAVES_PER_SIMD)
render_pool_bvh(DevScene sc, DevCamera cam, RenderParams rp) {
    extern __shared__ f4 smem[];
    __shared__ BlockPool block_pool;
    block_pool_init(&block_pool);
    SceneView vw = global_view(sc);
    f4 *free_lds = smem;
    if constexpr (LDS_SCENE) free_lds = stage_scene(sc, smem, vw);
    else __syncthreads();
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    const int M = rp.pool_slots;
    const unsigned long long lane_mask_lt = (1ull << lane) - 1ull;
    // LDS after the scene: the waves' slots, their lists, the traversal stacks
    f4 *pool = free_lds + (size_t)wave * M * RT_POOL_SLOT_F4;
    unsigned short *lists = reinterpret_cast<unsigned short *>(free_lds + (size_t)n_waves * M * RT_POOL_SLOT_F4);
    WaveLists wl;
    wl.pend = lists + (size_t)wave * 2 * M;
    wl.ready = wl.pend + M;
    wl.n_pend = wl.n_ready = 0;
    StackT *stack = reinterpret_cast<StackT *>(lists + (size_t)n_waves * 2 * M) + threadIdx.x;
    const int stack_stride = blockDim.x;

    const int leaf_min = rp.leaf_min;
    const int blocked_min = rp.blocked_min;
    const int dispatch_min = rp.dispatch_min;

    WavePool wp;
    pool_init(wp);
    int n_live = 0; // slots that hold a path (pending + being traced + ready); wave-uniform

    // engine state
    int cur = -1;   // slot being traced, -1: idle
    int pend = 0;   // parked leaf reference (0 = none)
    Ray ray;
    Trav tv;
    ray.o = ray.d = mk(0.0f, 0.0f, 0.0f);
    ray.tm = 0.0f;
    trav_begin(tv, ray, 0.0f);
    tv.ref = RT_TRAV_DONE;

    Cnt<COUNT> cn;
    unsigned long long w_t_iters = 0, w_t_lanes = 0, w_l_iters = 0, w_l_lanes = 0, w_s_phases = 0, w_s_lanes = 0;
    long long c_shade = 0, c_regen = 0, c_inner = 0, c_leaf = 0, c_mark = 0;
    const long long c_begin = COUNT ? clock64() : 0;
#define RT_TICK() if constexpr (COUNT) c_mark = clock64()
#define RT_TOCK(acc) if constexpr (COUNT) acc += clock64() - c_mark

    // ---- fill the pool -----------------------------------------------------------------------
    RT_TICK();
    for (int base = 0; base < M; base += 64) {
        int slot = base + lane;
        n_live += start_paths<COUNT>(slot < M, slot, pool, wl, wp, &block_pool, cam, rp, lane, lane_mask_lt);
    }
    RT_TOCK(c_regen);

    while (n_live > 0) {
        // ---- shade: a full wave of ready slots, or whatever is ready when nothing else is left ----
        const bool any_busy = __any(cur >= 0);
        if (wl.n_ready >= 64 || (wl.n_ready > 0 && wl.n_pend == 0 && !any_busy)) {
            RT_TICK();
            const int take = wl.n_ready < 64 ? wl.n_ready : 64;
            const bool act = lane < take;
            const int slot = act ? (int)wl.ready[wl.n_ready - 1 - lane] : 0;
            wl.n_ready -= take;
            if constexpr (COUNT) {
                w_s_phases++;
                w_s_lanes += take;
            }
            bool cont = false, ended = false;
            if (act) {
                f4 *s = pool + slot * RT_POOL_SLOT_F4;
                const f4 a0 = s[0], a1 = s[1], a2 = s[2], a3 = s[3], a4 = s[4];
                PathState ps;
                ps.ray.o = mk(a0.x, a0.y, a0.z);
                ps.ray.tm = a0.w;
                ps.ray.d = mk(a1.x, a1.y, a1.z);
                ps.beta = mk(a2.x, a2.y, a2.z);
                ps.rng.s = __builtin_bit_cast(unsigned, a2.w);
                ps.L = mk(a3.x, a3.y, a3.z);
                const unsigned packed = __builtin_bit_cast(unsigned, a3.w);
                ps.depth = (int)(packed & 0xffffu);
                unsigned n_rays = packed >> 16;
                TraceHit h;
                h.t = a1.w;
                h.slot = __builtin_bit_cast(int, a4.z);
                cont = path_after_trace<COUNT>(sc, vw, cam, ps, h, cn);
                if (cont) {
                    n_rays++;
                    s[0] = f4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tm};
                    s[1] = f4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
                    s[2] = f4{ps.beta.x, ps.beta.y, ps.beta.z, __builtin_bit_cast(float, ps.rng.s)};
                    s[3] = f4{ps.L.x, ps.L.y, ps.L.z,
                              __builtin_bit_cast(float, ((unsigned)ps.depth & 0xffffu) | (n_rays << 16))};
                } else {
                    ended = true;
                    write_record(rp, __builtin_bit_cast(unsigned, a4.y), __builtin_bit_cast(unsigned, a4.x), ps.L,
                                 n_rays);
                }
            }
            list_push(wl.pend, wl.n_pend, cont, slot, lane_mask_lt);
            RT_TOCK(c_shade);
            RT_TICK();
            // a finished path hands its slot to a new camera path while the launch has any
            const int n_ended = __popcll(__ballot(ended));
            if (n_ended > 0)
                n_live += start_paths<COUNT>(ended, slot, pool, wl, wp, &block_pool, cam, rp, lane, lane_mask_lt) - n_ended;
            RT_TOCK(c_regen);
            continue;
        }
        // ---- dispatch: idle engines take pending rays ------------------------------------------------
        {
            unsigned long long idle = __ballot(cur < 0);
            int n_idle = __popcll(idle);
            if (wl.n_pend > 0 && n_idle > 0 && (n_idle >= dispatch_min || !any_busy)) {
                RT_TICK();
                int r = __popcll(idle & lane_mask_lt);
                if (cur < 0 && r < wl.n_pend) {
                    cur = (int)wl.pend[wl.n_pend - 1 - r];
                    const f4 *s = pool + cur * RT_POOL_SLOT_F4;
                    const f4 a0 = s[0], a1 = s[1];
                    ray.o = mk(a0.x, a0.y, a0.z);
                    ray.tm = a0.w;
                    ray.d = mk(a1.x, a1.y, a1.z);
                    trav_begin(tv, ray, RT_INF);
                    pend = 0;
                }
                wl.n_pend -= n_idle < wl.n_pend ? n_idle : wl.n_pend;
                RT_TOCK(c_regen);
            }
        }
        if (!__any(cur >= 0)) continue; // everything left is ready: the shading branch takes it
        // ---- traversal: RT_INNER_UNROLL node steps, parked leaves, retire finished rays -----------------
        const bool trav = cur >= 0;
        if constexpr (COUNT) {
            unsigned long long im0 = __ballot(trav && tv.ref >= 0);
            if (im0) {
                w_t_iters++;
                w_t_lanes += __popcll(im0);
            }
        }
        RT_TICK();
#pragma unroll
        for (int k = 0; k < RT_INNER_UNROLL; k++) {
            if (trav && tv.ref >= 0) trav_inner<COUNT, StackT>(tv, vw.nodes, 0.001f, stack, stack_stride, cn);
            // a leaf does not stop the engine: it is parked and tested with other lanes' leaves
            if (trav && tv.ref < 0 && tv.ref != RT_TRAV_DONE && pend == 0) {
                pend = tv.ref;
                trav_pop(tv, stack, stack_stride);
            }
        }
        RT_TOCK(c_inner);
        RT_TICK();
        const bool blocked = trav && tv.ref < 0 && tv.ref != RT_TRAV_DONE; // second leaf, slot taken
        const bool has_pend = trav && pend != 0;
        unsigned long long pmask = __ballot(has_pend);
        if (pmask != 0ull) {
            unsigned long long imask = __ballot(trav && tv.ref >= 0);
            unsigned long long bmask = __ballot(blocked);
            if (__popcll(pmask) >= leaf_min || __popcll(bmask) >= blocked_min || imask == 0ull) {
                if constexpr (COUNT) {
                    w_l_iters++;
                    w_l_lanes += __popcll(pmask);
                }
                if (has_pend) {
                    trav_test_leaf<COUNT>(tv, pend, sc, vw, ray, 0.001f, cn);
                    pend = 0;
                    if (blocked) {
                        pend = tv.ref;
                        trav_pop(tv, stack, stack_stride);
                    }
                }
            }
        }
        // retire: the hit goes into the slot, the slot onto the ready list
        const bool fin = trav && tv.ref == RT_TRAV_DONE && pend == 0;
        if (fin) {
            f4 *s = pool + cur * RT_POOL_SLOT_F4;
            s[1].w = tv.best_t;
            s[4].z = __builtin_bit_cast(float, tv.best_slot);
        }
        list_push(wl.ready, wl.n_ready, fin, cur, lane_mask_lt);
        if (fin) cur = -1;
        RT_TOCK(c_leaf);
    }
    if constexpr (COUNT) {
        flush_counters<COUNT>(rp.counters, cn);
        if (lane == 0) {
            atomicAdd(&rp.counters->sched[0], w_t_iters);
            atomicAdd(&rp.counters->sched[1], w_t_lanes);
            atomicAdd(&rp.counters->sched[2], w_l_iters);
            atomicAdd(&rp.counters->sched[3], w_l_lanes);
            atomicAdd(&rp.counters->sched[4], w_s_phases);
            atomicAdd(&rp.counters->sched[5], w_s_lanes);
            atomicAdd(&rp.counters->cycles[0], (unsigned long long)(clock64() - c_begin));
            atomicAdd(&rp.counters->cycles[1], (unsigned long long)c_shade);
            atomicAdd(&rp.counters->cycles[2], (unsigned long long)c_regen);
            atomicAdd(&rp.counters->cycles[3], (unsigned long long)c_inner);
            atomicAdd(&rp.counters->cycles[4], (unsigned long long)c_leaf);
        }
    }
#undef RT_TICK
#undef RT_TOCK
}

} // namespace

namespace rtl {

hipError_t launch_pool(const DevScene &sc, const DevCamera &cam, const RenderParams &rp, const LaunchGeom &lg,
                       hipStream_t stream) {
    if (lg.lds_scene) {
        if (rp.counters)
            hipLaunchKernelGGL((render_pool_bvh<true, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, cam, rp);
        else
            hipLaunchKernelGGL((render_pool_bvh<true, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, cam, rp);
    } else {
        if (rp.counters)
            hipLaunchKernelGGL((render_pool_bvh<false, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, cam, rp);
        else
            hipLaunchKernelGGL((render_pool_bvh<false, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, cam, rp);
    }
    return hipGetLastError();
}

hipError_t set_pool_lds_limit(int bytes) {
    hipError_t e = hipSuccess;
#define RT_SETP(L, C)                                                                                        \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_pool_bvh<L, C>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    RT_SETP(true, true)
    RT_SETP(true, false)
    RT_SETP(false, true)
    RT_SETP(false, false)
#undef RT_SETP
    return e;
}

} // namespace rtl
