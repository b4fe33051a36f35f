// rt_wavefront.hip — the wavefront variant of the path tracer (RT_MODE_WAVEFRONT).
//
// Same integrator, sampler and per-ray code (rt_core.h) as the megakernel, organised as
// separate kernels around queues in HBM instead of one persistent kernel:
//
//   wf_prep      1 thread   : rotates the queue counters for the round
//   wf_generate  1 lane/slot: starts new camera paths in the path slots freed last round
//   wf_trace     persistent : closest hit for every queued ray (geometry staged in LDS); each
//                             ray is then appended to the queue of the material it hit (or to
//                             the miss queue) with one atomic per wave per queue
//   wf_shade     1 lane/hit : walks the material queues back to back, so that a wave shades one
//                             material; continuing paths are queued for the next round,
//                             finished ones write their radiance record and free their slot
//
// It exists to be compared with the megakernel (same images, see tests/test_gpu_wavefront.py):
// coherent shading and a trace kernel free of shading registers, paid for with ~200 bytes of
// HBM traffic per ray and four launches per bounce.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_core.h"
#include "rt_launch.h"
#include "rt_stage.h"

using namespace rtc;

namespace {

enum { Q_LAMBERTIAN = 0, Q_METAL = 1, Q_DIELECTRIC = 2, Q_LIGHT = 3, Q_MISS = 4, N_QUEUES = 5 };
// counters (unsigned words)
enum {
    C_TRACE_CUR = 0, // rays queued for this round
    C_TRACE_NEXT = 1,
    C_Q0 = 2,        // .. C_Q0 + 4: material / miss queues
    C_FREE = 7,      // path slots freed during this round's shading
    C_FREE_SNAP = 8, // free slots the generator may fill this round
    C_POOL = 9,      // next path index to start
    C_DONE = 10      // finished paths
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Appends `item` to queue `q` for the lanes with pred == true: one atomic per wave.
__device__ __forceinline__ void wave_push(int *q, unsigned *counter, bool pred, int item) {
    unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    unsigned base = 0;
    unsigned leader = (unsigned)__builtin_ctzll(m);
    if (lane_id() == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
    base = __shfl(base, (int)leader, 64);
    if (pred) q[base + (unsigned)__popcll(m & ((1ull << lane_id()) - 1ull))] = item;
}

__global__ void wf_prep(unsigned *cnt) {
    cnt[C_TRACE_CUR] = cnt[C_TRACE_NEXT];
    cnt[C_TRACE_NEXT] = 0;
    for (int k = 0; k < N_QUEUES; k++) cnt[C_Q0 + k] = 0;
    cnt[C_FREE_SNAP] = cnt[C_FREE];
    cnt[C_FREE] = 0;
}

__global__ void __launch_bounds__(256)
wf_generate(DevCamera cam, RenderParams rp, WfBuffers wf, unsigned total_paths, int *q_trace_cur) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    bool have_slot = i < wf.cnt[C_FREE_SNAP];
    // one pool ticket per lane that owns a free slot (wave-aggregated)
    unsigned long long m = __ballot(have_slot);
    unsigned base = 0;
    if (m != 0ull) {
        unsigned leader = (unsigned)__builtin_ctzll(m);
        if (lane_id() == leader) base = atomicAdd(&wf.cnt[C_POOL], (unsigned)__popcll(m));
        base = __shfl(base, (int)leader, 64);
    }
    unsigned p = base + (unsigned)__popcll(m & ((1ull << lane_id()) - 1ull));
    bool start = have_slot && p < total_paths;
    int slot = have_slot ? wf.q_free[i] : 0;
    if (start) {
        // paths are numbered sample-major: consecutive lanes take consecutive pixels
        unsigned s_local = p / (unsigned)rp.n_pixels;
        unsigned pixel = p - s_local * (unsigned)rp.n_pixels;
        int py = (int)(pixel / (unsigned)rp.width);
        int px = (int)(pixel - (unsigned)py * (unsigned)rp.width);
        PathState ps = path_start(cam, rp.seed, pixel, px, py, (unsigned)rp.sample_begin + s_local);
        wf.ray_o[slot] = f4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tm};
        wf.ray_d[slot] = f4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
        wf.thr[slot] = f4{1.0f, 1.0f, 1.0f, __builtin_bit_cast(float, ps.rng.s)};
        wf.rad[slot] = f4{0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, (unsigned)ps.depth)};
        wf.pix[slot] = pixel;
        wf.smp[slot] = s_local;
    }
    // a slot that finds no path (pool exhausted) is dropped: nothing will need it again
    wave_push(q_trace_cur, &wf.cnt[C_TRACE_CUR], start, slot);
}

// Persistent: one workgroup per CU stages the geometry into LDS once per launch and walks the
// ray queue with a grid stride; one whole ray per lane per iteration.
template <int TRAV, bool LDS_SCENE>
__global__ void __launch_bounds__(RT_MEGA_MAX_THREADS, RT_MEGA_WAVES_PER_SIMD)
wf_trace(DevScene sc, WfBuffers wf, const int *q_trace_cur) {
    extern __shared__ f4 smem[];
    SceneView vw = global_view(sc);
    f4 *free_lds = smem;
    if constexpr (LDS_SCENE) free_lds = stage_scene(sc, smem, vw);
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;
    StackT *stack = reinterpret_cast<StackT *>(free_lds) + threadIdx.x;
    const int stack_stride = blockDim.x;
    const unsigned n = wf.cnt[C_TRACE_CUR];
    Cnt<false> cn;
    // whole waves iterate together so that the wave-aggregated pushes see every lane
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned base = blockIdx.x * blockDim.x; base < n; base += stride) {
        unsigned i = base + threadIdx.x;
        bool live = i < n;
        int slot = live ? q_trace_cur[i] : 0;
        int q = -1;
        if (live) {
            f4 o = wf.ray_o[slot], d = wf.ray_d[slot];
            Ray r;
            r.o = mk(o.x, o.y, o.z);
            r.d = mk(d.x, d.y, d.z);
            r.tm = o.w;
            TraceHit h;
            if constexpr (TRAV == TRAV_LINEAR)
                h = trace_linear<false>(sc, vw, r, 0.001f, RT_INF, cn);
            else
                h = trace_bvh<false, StackT>(sc, vw, r, 0.001f, RT_INF, stack, stack_stride, cn);
            wf.hit_t[slot] = h.t;
            wf.hit_slot[slot] = h.slot;
            q = h.slot < 0 ? (int)Q_MISS : RT_KIND_MAT(vw.rec_i[h.slot].x);
        }
#pragma unroll
        for (int k = 0; k < N_QUEUES; k++) wave_push(wf.q_mat[k], &wf.cnt[C_Q0 + k], q == k, slot);
    }
}

// The material queues are walked back to back: lane i finds its queue by the running
// totals, so a wave shades one material except where two queues meet.
__global__ void __launch_bounds__(256)
wf_shade(DevScene sc, DevCamera cam, RenderParams rp, WfBuffers wf, int *q_trace_next) {
    const SceneView vw = global_view(sc);
    unsigned c[N_QUEUES], total = 0;
#pragma unroll
    for (int k = 0; k < N_QUEUES; k++) {
        c[k] = wf.cnt[C_Q0 + k];
        total += c[k];
    }
    Cnt<false> cn;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned base = blockIdx.x * blockDim.x; base < total; base += stride) {
        unsigned i = base + threadIdx.x;
        bool live = i < total;
        int slot = 0;
        if (live) {
            unsigned j = i;
            int k = 0;
            while (k < N_QUEUES - 1 && j >= c[k]) {
                j -= c[k];
                k++;
            }
            const int *q = k == 0 ? wf.q_mat[0] : k == 1 ? wf.q_mat[1] : k == 2 ? wf.q_mat[2] : k == 3 ? wf.q_mat[3] : wf.q_mat[4];
            slot = q[j];
        }
        bool cont = false, done = false;
        if (live) {
            f4 o = wf.ray_o[slot], d = wf.ray_d[slot], th = wf.thr[slot], ra = wf.rad[slot];
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
            h.t = wf.hit_t[slot];
            h.slot = wf.hit_slot[slot];
            cont = path_after_trace<false>(sc, vw, cam, ps, h, cn);
            if (cont) {
                wf.ray_o[slot] = f4{ps.ray.o.x, ps.ray.o.y, ps.ray.o.z, ps.ray.tm};
                wf.ray_d[slot] = f4{ps.ray.d.x, ps.ray.d.y, ps.ray.d.z, 0.0f};
                wf.thr[slot] = f4{ps.beta.x, ps.beta.y, ps.beta.z, __builtin_bit_cast(float, ps.rng.s)};
                wf.rad[slot] = f4{ps.L.x, ps.L.y, ps.L.z,
                                  __builtin_bit_cast(float, ((unsigned)ps.depth & 0xffffu) | (n_rays << 16))};
            } else {
                done = true;
                unsigned pixel = wf.pix[slot], s_local = wf.smp[slot];
                rp.stage[(size_t)s_local * (size_t)rp.n_pixels + pixel] =
                    f4{ps.L.x, ps.L.y, ps.L.z, __builtin_bit_cast(float, n_rays)};
            }
        }
        wave_push(q_trace_next, &wf.cnt[C_TRACE_NEXT], cont, slot);
        wave_push(wf.q_free, &wf.cnt[C_FREE], done, slot);
        unsigned long long dm = __ballot(done);
        if (dm != 0ull && lane_id() == (unsigned)__builtin_ctzll(dm)) atomicAdd(&wf.cnt[C_DONE], (unsigned)__popcll(dm));
    }
}

__global__ void wf_init(WfBuffers wf, unsigned n_slots) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots) wf.q_free[i] = (int)i;
    if (i == 0) {
        for (int k = 0; k < 16; k++) wf.cnt[k] = 0;
        wf.cnt[C_FREE] = n_slots;
    }
}

} // namespace

namespace rtl {

hipError_t set_wavefront_lds_limit(int bytes) {
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
// `host_cnt` is pinned host memory for polling the finished-path counter.
hipError_t render_wavefront(const DevScene &sc, const DevCamera &cam, const RenderParams &rp, const WfBuffers &wf,
                            unsigned n_slots, const LaunchGeom &lg, unsigned *host_cnt, hipStream_t stream,
                            int *rounds_out) {
    const unsigned total = (unsigned)rp.n_pixels * (unsigned)rp.sample_count;
    const int slot_grid = (int)((n_slots + 255) / 256);
    hipLaunchKernelGGL(wf_init, dim3(slot_grid), dim3(256), 0, stream, wf, n_slots);
    int rounds = 0;
    for (;;) {
        // a batch of rounds between two looks at the counters; a round advances every path by one bounce
        for (int k = 0; k < 8; k++, rounds++) {
            int *cur = wf.q_trace[rounds & 1], *next = wf.q_trace[(rounds + 1) & 1];
            hipLaunchKernelGGL(wf_prep, dim3(1), dim3(1), 0, stream, wf.cnt);
            hipLaunchKernelGGL(wf_generate, dim3(slot_grid), dim3(256), 0, stream, cam, rp, wf, total, cur);
            if (sc.traversal == TRAV_LINEAR) {
                if (lg.lds_scene)
                    hipLaunchKernelGGL((wf_trace<TRAV_LINEAR, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur);
                else
                    hipLaunchKernelGGL((wf_trace<TRAV_LINEAR, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur);
            } else {
                if (lg.lds_scene)
                    hipLaunchKernelGGL((wf_trace<TRAV_BVH, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur);
                else
                    hipLaunchKernelGGL((wf_trace<TRAV_BVH, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes, stream, sc, wf, cur);
            }
            hipLaunchKernelGGL(wf_shade, dim3(slot_grid), dim3(256), 0, stream, sc, cam, rp, wf, next);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(host_cnt, wf.cnt, 16 * sizeof(unsigned), hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (host_cnt[C_DONE] >= total) break;
        if (rounds > 100000) return hipErrorUnknown; // cannot happen: every round retires or advances paths
    }
    if (rounds_out) *rounds_out = rounds;
    return hipSuccess;
}

} // namespace rtl
