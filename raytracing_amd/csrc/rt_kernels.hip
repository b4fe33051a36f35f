// rt_kernels.hip — gfx950 kernels of the path tracer.
//
//   render_mega   persistent wave64 megakernel: every lane owns one path at a time and
//                 pulls the next (pixel, sample) from a per-wave pool refilled from one
//                 global counter; geometry (primitive slots + BVH nodes) is staged into
//                 LDS once per workgroup when it fits, traversal stacks live in LDS.
//   reduce_stage  sums the per-path radiance records of a launch, in sample order, into
//                 the caller's fp32 accumulation image (camera.h:43-47's pixel_color +=).
//   trace_batch   closest hit for a batch of rays (the parity entry point).
//
// Replaces camera::render's triple loop (camera.h:40-47) and everything below it.
#include <hip/hip_runtime.h>

#include "rt_core.h"
#include "rt_launch.h"

using namespace rtc;

namespace {

__device__ __forceinline__ unsigned long long wave_bcast64(unsigned long long v) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(v & 0xffffffffull));
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

template <bool COUNT> __device__ __forceinline__ void flush_counters(Counters *out, const Cnt<COUNT> &cn) {
    if constexpr (COUNT) {
        atomicAdd(&out->box_tests, (unsigned long long)cn.box);
        atomicAdd(&out->sphere_tests, (unsigned long long)cn.sph);
        atomicAdd(&out->sphere_hits, (unsigned long long)cn.sphh);
        atomicAdd(&out->quad_tests, (unsigned long long)cn.quad);
        atomicAdd(&out->shades[0], (unsigned long long)cn.sh0);
        atomicAdd(&out->shades[1], (unsigned long long)cn.sh1);
        atomicAdd(&out->shades[2], (unsigned long long)cn.sh2);
        atomicAdd(&out->shades[3], (unsigned long long)cn.sh3);
        atomicAdd(&out->noise_evals, (unsigned long long)cn.noise);
        atomicAdd(&out->image_evals, (unsigned long long)cn.img);
        atomicAdd(&out->checker_evals, (unsigned long long)cn.chk);
    }
}

// Copies n 16-byte words from HBM to LDS with the whole workgroup.
__device__ __forceinline__ void stage_to_lds(f4 *dst, const f4 *src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

template <int TRAV, bool LDS_SCENE, bool COUNT>
__global__ void __launch_bounds__(RT_MEGA_MAX_THREADS)
render_mega(DevScene sc, DevCamera cam, RenderParams rp) {
    extern __shared__ f4 smem[];
    // LDS layout: [slots][nodes][stacks]
    const f4 *slots = sc.slots;
    const f4 *nodes = sc.nodes;
    f4 *stack_base_f4 = smem;
    if constexpr (LDS_SCENE) {
        int n_slot_words = sc.n_slots * sc.slot_f4;
        int n_node_words = sc.n_nodes * 4;
        stage_to_lds(smem, sc.slots, n_slot_words);
        stage_to_lds(smem + n_slot_words, sc.nodes, n_node_words);
        slots = smem;
        nodes = smem + n_slot_words;
        stack_base_f4 = smem + n_slot_words + n_node_words;
        __syncthreads();
    }
    using StackT = typename std::conditional<LDS_SCENE, short, int>::type;
    StackT *stack = reinterpret_cast<StackT *>(stack_base_f4) + threadIdx.x;
    const int stack_stride = blockDim.x;

    const int lane = threadIdx.x & 63;
    const unsigned long long lane_mask_lt = (1ull << lane) - 1ull;
    const unsigned paths_per_block64 = 64u * (unsigned)rp.sample_count;

    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false;
    bool alive = false;
    PathState ps;
    unsigned pixel = 0, sample = 0, n_rays = 0;
    Cnt<COUNT> cn;

    for (;;) {
        // ---- hand new paths to idle lanes -------------------------------------------------
        unsigned long long need = __ballot(!alive);
        if (need != 0ull) {
            if (pool_next >= pool_end && !exhausted) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(rp.next_path, (unsigned long long)RT_POOL_CHUNK);
                base = wave_bcast64(base);
                if (base >= rp.n_paths) {
                    exhausted = true;
                } else {
                    pool_next = base;
                    pool_end = base + RT_POOL_CHUNK < rp.n_paths ? base + RT_POOL_CHUNK : rp.n_paths;
                }
            }
            if (pool_next < pool_end) {
                unsigned n_need = (unsigned)__popcll(need);
                if (!alive) {
                    unsigned rank = (unsigned)__popcll(need & lane_mask_lt);
                    unsigned long long p = pool_next + rank;
                    if (p < pool_end) {
                        // path p -> (64-pixel strip, sample, pixel within the strip)
                        unsigned long long blk = p / paths_per_block64;
                        unsigned rem = (unsigned)(p - blk * paths_per_block64);
                        unsigned s_local = rem >> 6;
                        pixel = (unsigned)blk * 64u + (rem & 63u);
                        sample = s_local;
                        if (pixel < (unsigned)rp.n_pixels) {
                            ps = path_start(cam, rp.seed, pixel, (unsigned)rp.sample_begin + s_local);
                            n_rays = 0;
                            alive = true;
                        }
                    }
                }
                pool_next = pool_next + n_need < pool_end ? pool_next + n_need : pool_end;
            }
        }
        if (!__any(alive)) {
            if (exhausted) break;
            continue;
        }

        // ---- one segment of ray_color for every live lane -----------------------------------
        if (alive) {
            bool cont = path_step<TRAV, COUNT, StackT>(sc, slots, nodes, cam, ps, stack, stack_stride, n_rays, cn);
            if (!cont) {
                f4 rec;
                rec.x = ps.L.x;
                rec.y = ps.L.y;
                rec.z = ps.L.z;
                rec.w = __builtin_bit_cast(float, n_rays);
                rp.stage[(size_t)sample * (size_t)rp.n_pixels + pixel] = rec;
                alive = false;
            }
        }
    }
    if constexpr (COUNT) flush_counters<COUNT>(rp.counters, cn);
}

// accum[pixel] += sum over the launch's samples, in sample order; ray counts are
// summed per workgroup and added to *ray_count.
__global__ void __launch_bounds__(256) reduce_stage(const f4 *stage, int n_pixels, int sample_count,
                                                    float *accum, unsigned long long *ray_count) {
    __shared__ unsigned long long wave_rays[4];
    int pixel = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long rays = 0;
    if (pixel < n_pixels) {
        float r = 0.0f, g = 0.0f, b = 0.0f;
        for (int s = 0; s < sample_count; s++) {
            f4 v = stage[(size_t)s * n_pixels + pixel];
            r += v.x;
            g += v.y;
            b += v.z;
            rays += __builtin_bit_cast(unsigned, v.w);
        }
        accum[3 * (size_t)pixel + 0] += r;
        accum[3 * (size_t)pixel + 1] += g;
        accum[3 * (size_t)pixel + 2] += b;
    }
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_down(rays, off, 64);
    if ((threadIdx.x & 63) == 0) wave_rays[threadIdx.x >> 6] = rays;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(ray_count, wave_rays[0] + wave_rays[1] + wave_rays[2] + wave_rays[3]);
}

template <int TRAV>
__global__ void __launch_bounds__(RT_TRACE_THREADS)
trace_batch(DevScene sc, const rt_ray *rays, long long n_rays, float tmin, float tmax, rt_hit *hits) {
    __shared__ int stack_mem[RT_STACK_DEPTH * RT_TRACE_THREADS];
    int *stack = stack_mem + threadIdx.x;
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
            h = trace_linear<false>(sc, sc.slots, sc.meta, r, tmin, tmax, cn);
        else
            h = trace_bvh<false, int>(sc, sc.slots, sc.nodes, sc.meta, r, tmin, tmax, stack, RT_TRACE_THREADS,
                                      cn);
        rt_hit out;
        out.prim = -1;
        out.front_face = 0;
        out.t = 0.0f;
        out.p[0] = out.p[1] = out.p[2] = 0.0f;
        out.normal[0] = out.normal[1] = out.normal[2] = 0.0f;
        out.u = out.v = 0.0f;
        if (h.slot >= 0) {
            Surf s = surface_at(sc, sc.slots, sc.meta, r, h);
            out.prim = s.prim;
            out.front_face = s.front ? 1 : 0;
            out.t = h.t;
            out.p[0] = s.p.x;
            out.p[1] = s.p.y;
            out.p[2] = s.p.z;
            out.normal[0] = s.n.x;
            out.normal[1] = s.n.y;
            out.normal[2] = s.n.z;
            if (s.type == RT_PRIM_SPHERE) sphere_uv(s.outward, out.u, out.v); // sphere.h:70
            else {
                out.u = s.u;
                out.v = s.v;
            }
        }
        hits[k] = out;
    }
}

} // namespace

// ------------------------------------------------------------------------------------------
// Host-side launchers (called from rt_host.cpp)
namespace rtl {

template <int TRAV, bool LDS_SCENE>
static hipError_t launch_mega_tc(const DevScene &sc, const DevCamera &cam, const RenderParams &rp,
                                 const LaunchGeom &lg, hipStream_t stream) {
    if (rp.counters)
        hipLaunchKernelGGL((render_mega<TRAV, LDS_SCENE, true>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                           stream, sc, cam, rp);
    else
        hipLaunchKernelGGL((render_mega<TRAV, LDS_SCENE, false>), dim3(lg.grid), dim3(lg.block), lg.lds_bytes,
                           stream, sc, cam, rp);
    return hipGetLastError();
}

hipError_t launch_mega(const DevScene &sc, const DevCamera &cam, const RenderParams &rp, const LaunchGeom &lg,
                       hipStream_t stream) {
    if (sc.traversal == TRAV_LINEAR)
        return lg.lds_scene ? launch_mega_tc<TRAV_LINEAR, true>(sc, cam, rp, lg, stream)
                            : launch_mega_tc<TRAV_LINEAR, false>(sc, cam, rp, lg, stream);
    return lg.lds_scene ? launch_mega_tc<TRAV_BVH, true>(sc, cam, rp, lg, stream)
                        : launch_mega_tc<TRAV_BVH, false>(sc, cam, rp, lg, stream);
}

hipError_t set_mega_lds_limit(int bytes) {
    hipError_t e = hipSuccess;
#define RT_SET(TR, L, C)                                                                                     \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void *)render_mega<TR, L, C>,                                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    RT_SET(TRAV_LINEAR, true, true)
    RT_SET(TRAV_LINEAR, true, false)
    RT_SET(TRAV_LINEAR, false, true)
    RT_SET(TRAV_LINEAR, false, false)
    RT_SET(TRAV_BVH, true, true)
    RT_SET(TRAV_BVH, true, false)
    RT_SET(TRAV_BVH, false, true)
    RT_SET(TRAV_BVH, false, false)
#undef RT_SET
    return e;
}

hipError_t launch_reduce(const f4 *stage, int n_pixels, int sample_count, float *accum,
                         unsigned long long *ray_count, hipStream_t stream) {
    int grid = (n_pixels + 255) / 256;
    hipLaunchKernelGGL(reduce_stage, dim3(grid), dim3(256), 0, stream, stage, n_pixels, sample_count, accum,
                       ray_count);
    return hipGetLastError();
}

hipError_t launch_trace(const DevScene &sc, const rt_ray *rays, long long n, float tmin, float tmax,
                        rt_hit *hits, hipStream_t stream) {
    long long want = (n + RT_TRACE_THREADS - 1) / RT_TRACE_THREADS;
    int grid = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    if (sc.traversal == TRAV_LINEAR)
        hipLaunchKernelGGL((trace_batch<TRAV_LINEAR>), dim3(grid), dim3(RT_TRACE_THREADS), 0, stream, sc, rays,
                           n, tmin, tmax, hits);
    else
        hipLaunchKernelGGL((trace_batch<TRAV_BVH>), dim3(grid), dim3(RT_TRACE_THREADS), 0, stream, sc, rays, n,
                           tmin, tmax, hits);
    return hipGetLastError();
}

} // namespace rtl
