// rt_wavefront.hip — the wavefront variant of the path tracer (RT_MODE_WAVEFRONT).
//
// Same integrator, sampler and per-ray code (rt_core.h) as the megakernel, organised as
// separate kernels around queues in HBM instead of one persistent kernel:
//
//   wf_generate  1 lane/slot: starts new camera paths in the path slots freed last round
//   wf_trace     persistent : closest hit for every queued ray (geometry staged in LDS)
//   wf_classify  1 lane/ray : appends each traced ray to the queue of the material it hit (or
//                             to the miss queue), one global atomic per workgroup per queue
//   wf_shade     1 lane/hit : walks the material queues back to back, so that a wave shades one
//                             material; continuing paths are queued for the next round,
//                             finished ones write their radiance record and free their slot
//   wf_prep      1 thread   : rotates the queue counters for the next round
//
// It exists to be compared with the megakernel (same images, see tests/test_gpu_wavefront.py):
// coherent shading and a trace kernel free of shading registers, paid for with ~200 bytes of
// HBM traffic per ray and five launches per bounce.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_core.h"
#include "rt_launch.h"
#include "rt_sched.h"
#include "rt_stage.h"

using namespace rtc;

namespace {

enum { Q_LAMBERTIAN = 0, Q_METAL = 1, Q_DIELECTRIC = 2, Q_LIGHT = 3, Q_MISS = 4, N_QUEUES = 5 };
// counters (unsigned words)
enum {
    C_TRACE_CUR = 0,  // rays queued for this round
    C_TRACE_NEXT = 1, // rays queued for the next round (written by the shading kernel) ...
    C_FREE = 2,       // ... and, next to it, the path slots that kernel freed
    C_Q0 = 3,         // .. C_Q0 + 4: material / miss queues
    C_FREE_SNAP = 8,  // free slots the generator looks at this round
    C_POOL = 9,       // next path index to start
    C_DONE = 10,      // finished paths (up to the previous round)
    C_GEN_BASE = 11,  // rays carried over from shading: the generator appends behind them
    C_GEN_COUNT = 12  // paths the generator starts this round
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Appends `item` to queue q[key] (key < 0: nothing) for a whole workgroup with ONE global atomic
// per non-empty queue: atomics on a single address are the scarce resource of queue-based
// schedulers (about 10 ns each on this GPU).  Every thread of the workgroup must call it.
// `lds` holds 2 * NQ words.
template <int NQ>
__device__ __forceinline__ void block_push(int *const (&q)[NQ], unsigned *counters, int key, int item,
                                           unsigned *lds) {
    if (threadIdx.x < NQ) lds[threadIdx.x] = 0;
    __syncthreads();
    unsigned my_off = 0;
    const unsigned long long lt = (1ull << lane_id()) - 1ull;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        unsigned long long m = __ballot(key == k);
        if (m != 0ull) {
            unsigned w = 0;
            unsigned leader = (unsigned)__builtin_ctzll(m);
            if (lane_id() == leader) w = atomicAdd(&lds[k], (unsigned)__popcll(m));
            w = __shfl(w, (int)leader, 64);
            if (key == k) my_off = w + (unsigned)__popcll(m & lt);
        }
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        unsigned c = lds[threadIdx.x];
        lds[NQ + threadIdx.x] = c ? atomicAdd(&counters[threadIdx.x], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NQ; k++)
        if (key == k) q[k][lds[NQ + k] + my_off] = item;
    __syncthreads();
}

// Book-keeping between two rounds, by one thread: the paths the generator started in the round
// that ends are taken off the pool, the slots its shading freed are credited to the finished
// paths, and the next round's queue is laid out (carried-over rays first, new paths behind).
// (A ticket taken by every workgroup of the shading kernel, so that the last one could do this, was
// measured slower than this one-thread launch: 2048 atomics on one address per round.)
__device__ __forceinline__ void prep_round(unsigned *cnt, unsigned total_paths) {
    cnt[C_POOL] += cnt[C_GEN_COUNT];
    cnt[C_DONE] += cnt[C_FREE];
    unsigned carried = cnt[C_TRACE_NEXT];
    unsigned n_free = cnt[C_FREE_SNAP] = cnt[C_FREE];
    unsigned left = total_paths - cnt[C_POOL];
    unsigned n_new = n_free < left ? n_free : left;
    cnt[C_GEN_BASE] = carried;
    cnt[C_GEN_COUNT] = n_new;
    cnt[C_TRACE_CUR] = carried + n_new;
    cnt[C_TRACE_NEXT] = 0;
    cnt[C_FREE] = 0;
    for (int k = 0; k < N_QUEUES; k++) cnt[C_Q0 + k] = 0;
}
__global__ void wf_prep(unsigned *cnt, unsigned total_paths) { prep_round(cnt, total_paths); }

// Thread i starts path number pool + i in the i-th free slot.  No atomics: prep_round has
// already reserved the queue space and moves the pool position afterwards.
// Free slots beyond the paths that are left are dropped (nothing will need them again).
__global__ void __launch_bounds__(256)
wf_generate(DevCamera cam, RenderParams rp, WfBuffers wf, int *q_trace_cur) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wf.cnt[C_GEN_COUNT]) return;
    unsigned p = wf.cnt[C_POOL] + i;
    int slot = wf.q_free[i];
    // paths are numbered sample-major: consecutive lanes take consecutive pixels
    unsigned s_local = p / (unsigned)rp.n_pixels;
    unsigned pixel = p - s_local * (unsigned)rp.n_pixels;
    int py = (int)(pixel / (unsigned)rp.width);
    int px = (int)(pixel - (unsigned)py * (unsigned)rp.width);
    PathState ps = path_start(cam, rp.seed, pixel, px, py, (unsigned)rp.sample_begin + s_local);
    f4 *st = wf.state + 4 * (size_t)slot;
    st[0] = f4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tm};
    st[1] = f4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
    st[2] = f4{1.0f, 1.0f, 1.0f, __builtin_bit_cast(float, ps.rng.s)};
    st[3] = f4{0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, (unsigned)ps.depth)};
    wf.aux[slot] = i4{0, -1, (int)pixel, (int)s_local};
    q_trace_cur[wf.cnt[C_GEN_BASE] + i] = slot;
}

// Persistent: one workgroup per CU stages the geometry into LDS once per launch; its waves
// take batches of 64 queued rays, wave w the batches w, w + (waves of the grid), ...  Linear
// scenes: one whole ray per lane.  BVH scenes: the decoupled scheme of the render kernel
// (wave_traverse: blocks of node steps, parked leaves); a lane whose ray is done writes its
// result and takes the next ray of the wave's batch, so the wave does not wait for its slowest
// ray.  Writes the hit and, per queue position, the key of the queue the ray goes to next.
template <int TRAV, bool LDS_SCENE>
__global__ void __launch_bounds__(RT_MEGA_MAX_THREADS, RT_MEGA_WAVES_PER_SIMD)
wf_trace(DevScene sc, WfBuffers wf, const int *q_trace_cur, RenderParams rp) {
    extern __shared__ f4 smem[];
    SceneView vw = global_view(sc);
    f4 *free_lds = smem;
    const unsigned n = wf.cnt[C_TRACE_CUR];
    if (blockIdx.x * blockDim.x >= n) return; // nothing for this workgroup: skip the staging
    if constexpr (LDS_SCENE) free_lds = stage_scene(sc, smem, vw);
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;
    if constexpr (!LDS_SCENE && TRAV == TRAV_BVH) { // the scene stays in HBM: the top of the tree comes along
        stage_nodes_to_lds(free_lds, sc.nodes, sc.n_top);
        vw.top = free_lds;
        vw.n_top = sc.n_top;
        free_lds += (size_t)sc.n_top * RT_NODE_LDS_F4;
        __syncthreads();
    }
    constexpr bool SPILL = !LDS_SCENE;
    typedef TravStack<StackT, RT_MEGA_MAX_THREADS> Stack;
    const Stack stk = lane_stack<StackT, SPILL, RT_MEGA_MAX_THREADS>(free_lds, sc, rp.stack_lds_entries, rp.stack_spill);
    Cnt<false> cn;
    if constexpr (TRAV == TRAV_LINEAR) {
        const unsigned stride = gridDim.x * blockDim.x;
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
            int slot = q_trace_cur[i];
            const f4 *st = wf.state + 4 * (size_t)slot;
            f4 o = st[0], d = st[1];
            Ray r;
            r.o = mk(o.x, o.y, o.z);
            r.d = mk(d.x, d.y, d.z);
            r.tm = o.w;
            TraceHit h = trace_linear<false>(sc, vw, r, 0.001f, RT_INF, cn);
            // the first 8 bytes of the auxiliary record: {t, slot}
            *reinterpret_cast<float2 *>(&wf.aux[slot]) = float2{h.t, __builtin_bit_cast(float, h.slot)};
            wf.key[i] = h.slot < 0 ? (int)Q_MISS : RT_KIND_MAT(vw.rec_i[h.slot].x);
        }
    } else {
        const int lane = threadIdx.x & 63;
        const unsigned long long lane_mask_lt = (1ull << lane) - 1ull;
        const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
        const unsigned n_batches = (n + 63u) / 64u;
        unsigned batch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); // the wave's next batch
        unsigned k = 0, k_end = 0;                                             // what is left of its current one
        int state = ST_IDLE;
        Parked pend;
        unsigned mine = 0;
        int slot = 0;
        Ray ray;
        Trav tv;
        SchedStats ss;
        for (;;) {
            if (state == ST_READY) {
                *reinterpret_cast<float2 *>(&wf.aux[slot]) = float2{tv.best_t, __builtin_bit_cast(float, tv.best_slot)};
                // (the instance the primitive was hit in travels in the spare word beside the direction)
                if (RT_FEAT_INSTANCE) wf.state[4 * (size_t)slot + 1].w = __builtin_bit_cast(float, tv.best_inst);
                wf.key[mine] = tv.best_slot < 0 ? (int)Q_MISS : RT_KIND_MAT(vw.rec_i[tv.best_slot].x);
                state = ST_IDLE;
            }
            unsigned long long need = __ballot(state == ST_IDLE);
            bool want = state == ST_IDLE;
            for (int pass = 0; pass < 2 && need != 0ull && (k < k_end || batch < n_batches); pass++) {
                if (k >= k_end) {
                    k = batch * 64u;
                    k_end = k + 64u < n ? k + 64u : n;
                    batch += n_waves;
                }
                const unsigned avail = k_end - k;
                const unsigned r = (unsigned)__popcll(need & lane_mask_lt);
                const bool take = want && r < avail;
                if (take) {
                    mine = k + r;
                    slot = q_trace_cur[mine];
                    const f4 *st = wf.state + 4 * (size_t)slot;
                    const f4 o = st[0], d = st[1];
                    ray.o = mk(o.x, o.y, o.z);
                    ray.d = mk(d.x, d.y, d.z);
                    ray.tm = o.w;
                    trav_begin(tv, ray, RT_INF);
                    trav_test_big<false>(tv, sc, vw, ray, 0.001f, cn);
                    pend = Parked();
                    state = ST_TRAV;
                    want = false;
                }
                const unsigned n_need = (unsigned)__popcll(need);
                k += n_need < avail ? n_need : avail;
                need &= ~__ballot(take);
            }
            const bool more = k < k_end || batch < n_batches;
            if (!__any(state != ST_IDLE)) {
                if (!more) break;
                continue;
            }
            wave_traverse<LDS_SCENE, false, Stack, SPILL>(state, tv, pend, ray, sc, vw, stk, 0.001f, rp.min_ready, rp.leaf_min,
                                                           rp.blocked_min, more, cn, ss);
        }
    }
}

// Distributes the traced rays over the material queues.
__global__ void __launch_bounds__(1024) wf_classify(WfBuffers wf, const int *q_trace_cur) {
    __shared__ unsigned lds[2 * N_QUEUES];
    const unsigned n = wf.cnt[C_TRACE_CUR];
    int *const q[N_QUEUES] = {wf.q_mat[0], wf.q_mat[1], wf.q_mat[2], wf.q_mat[3], wf.q_mat[4]};
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned base = blockIdx.x * blockDim.x; base < n; base += stride) {
        unsigned i = base + threadIdx.x;
        bool live = i < n;
        block_push<N_QUEUES>(q, &wf.cnt[C_Q0], live ? wf.key[i] : -1, live ? q_trace_cur[i] : 0, lds);
    }
}

// The material queues are walked back to back: lane i finds its queue by the running
// totals, so a wave shades one material except where two queues meet.
__global__ void __launch_bounds__(1024)
wf_shade(DevScene sc, DevCamera cam, RenderParams rp, WfBuffers wf, int *q_trace_next) {
    __shared__ unsigned lds[4];
    const SceneView vw = global_view(sc);
    unsigned c[N_QUEUES], total = 0;
#pragma unroll
    for (int k = 0; k < N_QUEUES; k++) {
        c[k] = wf.cnt[C_Q0 + k];
        total += c[k];
    }
    int *const out[2] = {q_trace_next, wf.q_free};
    Cnt<false> cn;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned base = blockIdx.x * blockDim.x; base < total; base += stride) {
        unsigned i = base + threadIdx.x;
        bool live = i < total;
        int slot = 0;
        int key = -1;
        if (live) {
            unsigned j = i;
            int k = 0;
            while (k < N_QUEUES - 1 && j >= c[k]) {
                j -= c[k];
                k++;
            }
            const int *q = k == 0 ? wf.q_mat[0] : k == 1 ? wf.q_mat[1] : k == 2 ? wf.q_mat[2] : k == 3 ? wf.q_mat[3] : wf.q_mat[4];
            slot = q[j];
            f4 *st = wf.state + 4 * (size_t)slot;
            f4 o = st[0], d = st[1], th = st[2], ra = st[3];
            const i4 ax = wf.aux[slot];
            PathState ps;
            ps.ray.o = mk(o.x, o.y, o.z);
            ps.ray.d = mk(d.x, d.y, d.z);
            ps.ray.tm = o.w;
            ps.beta = mk(th.x, th.y, th.z);
            ps.rng.s = __builtin_bit_cast(unsigned, th.w);
            ps.L = mk(ra.x, ra.y, ra.z);
            unsigned packed = __builtin_bit_cast(unsigned, ra.w);
            ps.depth = (int)(packed & 0xffffu);
            unsigned n_rays = (packed >> 16) + 1u;
            TraceHit h;
            h.t = __builtin_bit_cast(float, ax.x);
            h.slot = ax.y;
            if (RT_FEAT_INSTANCE) h.inst = __builtin_bit_cast(int, d.w);
            if (path_after_trace<false>(sc, vw, cam, ps, h, cn)) {
                key = 0;
                st[0] = f4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tm};
                st[1] = f4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
                st[2] = f4{ps.beta.x, ps.beta.y, ps.beta.z, __builtin_bit_cast(float, ps.rng.s)};
                st[3] = f4{ps.L.x, ps.L.y, ps.L.z,
                           __builtin_bit_cast(float, ((unsigned)ps.depth & 0xffffu) | (n_rays << 16))};
            } else {
                key = 1;
                unsigned pixel = (unsigned)ax.z, s_local = (unsigned)ax.w;
                rp.stage[(size_t)s_local * (size_t)rp.n_pixels + pixel] =
                    f4{ps.L.x, ps.L.y, ps.L.z, __builtin_bit_cast(float, n_rays)};
            }
        }
        block_push<2>(out, &wf.cnt[C_TRACE_NEXT], key, slot, lds);
    }
}

__global__ void wf_init(WfBuffers wf, unsigned n_slots, unsigned total_paths) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots) wf.q_free[i] = (int)i;
    if (i == 0) {
        for (int k = 0; k < 16; k++) wf.cnt[k] = 0;
        wf.cnt[C_FREE] = n_slots;
        // prep_round credits the freed slots of the previous round to the finished paths
        wf.cnt[C_DONE] = 0u - n_slots;
        prep_round(wf.cnt, total_paths);
    }
}

} // namespace

namespace rtl {

// This file is compiled twice: as is, and with -DRT_INST (the two-level traversal of scenes with
// instances; launchers suffixed _inst), like rt_kernels.hip.
#if defined(RT_INST)
#define RT_WF_NAME(x) x##_inst
#else
#define RT_WF_NAME(x) x
#endif

hipError_t RT_WF_NAME(set_wavefront_lds_limit)(int bytes) {
    hipError_t e = hipSuccess;
#define RT_SETW(TR, L)                                                                                       \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)wf_trace<TR, L>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    RT_SETW(TRAV_LINEAR, true)
    RT_SETW(TRAV_LINEAR, false)
    RT_SETW(TRAV_BVH, true)
    RT_SETW(TRAV_BVH, false)
#undef RT_SETW
    return e;
}

// Renders one launch chunk (rp.sample_count samples of every pixel) into rp.stage.
// `host_cnt` is pinned host memory (2 x 16 words) for looking at the finished-path counter.  Rounds
// are enqueued eight at a time, and the next eight go out before the host looks at the counters
// the previous eight left behind, so the GPU never waits for that look; the rounds that turn out
// to be superfluous find empty queues and end at once.
hipError_t RT_WF_NAME(render_wavefront)(const DevScene &sc, const DevCamera &cam, const RenderParams &rp, const WfBuffers &wf,
                            unsigned n_slots, const LaunchGeom &lg, unsigned *host_cnt, hipStream_t stream,
                            int *rounds_out) {
    const unsigned total = (unsigned)rp.n_pixels * (unsigned)rp.sample_count;
    const int slot_grid = (int)((n_slots + 255) / 256);
    const int big_grid = (int)((n_slots + 1023) / 1024);
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipError_t e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
    if (e != hipSuccess) {
        if (ev[0]) hipEventDestroy(ev[0]);
        return e;
    }
    hipLaunchKernelGGL(wf_init, dim3(slot_grid), dim3(256), 0, stream, wf, n_slots, total);
    int rounds = 0;
    // a batch of rounds, then a copy of the counters into half `b` of host_cnt and an event behind it;
    // a round advances every path by one bounce
    auto enqueue_batch = [&](int b) -> hipError_t {
        for (int k = 0; k < 8; k++, rounds++) {
            int *cur = wf.q_trace[rounds & 1], *next = wf.q_trace[(rounds + 1) & 1];
            hipLaunchKernelGGL(wf_generate, dim3(slot_grid), dim3(256), 0, stream, cam, rp, wf, cur);
            if (sc.traversal == TRAV_LINEAR) {
                if (lg.lds_scene)
                    hipLaunchKernelGGL((wf_trace<TRAV_LINEAR, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur, rp);
                else
                    hipLaunchKernelGGL((wf_trace<TRAV_LINEAR, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur, rp);
            } else {
                if (lg.lds_scene)
                    hipLaunchKernelGGL((wf_trace<TRAV_BVH, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur, rp);
                else
                    hipLaunchKernelGGL((wf_trace<TRAV_BVH, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur, rp);
            }
            hipLaunchKernelGGL(wf_classify, dim3(big_grid), dim3(1024), 0, stream, wf, cur);
            hipLaunchKernelGGL(wf_shade, dim3(big_grid), dim3(1024), 0, stream, sc, cam, rp, wf, next);
            hipLaunchKernelGGL(wf_prep, dim3(1), dim3(1), 0, stream, wf.cnt, total);
        }
        hipError_t r = hipGetLastError();
        if (r == hipSuccess) r = hipMemcpyAsync(host_cnt + 16 * b, wf.cnt, 16 * sizeof(unsigned), hipMemcpyDeviceToHost, stream);
        if (r == hipSuccess) r = hipEventRecord(ev[b], stream);
        return r;
    };
    int rounds_needed = 0;
    e = enqueue_batch(0);
    for (int b = 0; e == hipSuccess; b ^= 1) {
        const int rounds_before = rounds;
        e = enqueue_batch(b ^ 1);
        if (e == hipSuccess) e = hipEventSynchronize(ev[b]);
        if (e != hipSuccess) break;
        // (the round's closing book-keeping has already credited the last round's paths)
        if (host_cnt[16 * b + C_DONE] + host_cnt[16 * b + C_FREE] >= total) {
            rounds_needed = rounds_before;
            break;
        }
        if (rounds > 100000) e = hipErrorUnknown; // cannot happen: every round retires or advances paths
    }
    hipEventDestroy(ev[0]);
    hipEventDestroy(ev[1]);
    if (e != hipSuccess) return e;
    if (rounds_out) *rounds_out = rounds_needed;
    return hipSuccess;
}

} // namespace rtl
