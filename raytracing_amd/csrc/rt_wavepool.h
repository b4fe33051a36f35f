// rt_wavepool.h — device-only helpers shared by the persistent render kernels: the two-level
// reservation of paths from the launch's global counter, the per-path radiance record, and
// the flush of the instrumented variants' counters.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_core.h"
#include "rt_launch.h"

namespace rtc {

__device__ __forceinline__ unsigned wave_bcast32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

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

// Pool of paths.  The launch is a sequence of units of 64 paths: unit u covers the pixels
// [64 * (u % n_strips), +64) of local sample u / n_strips (sample-major), or the pixels
// [64 * (u / sample_count), +64) of local sample u % sample_count (RenderParams::pixel_major:
// scenes read from HBM, where the waves of a CU should work on neighbouring rays).  Two levels:
//   * the workgroup reserves runs of consecutive units from the launch's global counter, one
//     atomic per run, and fewer units per run as the launch runs out ("guided" self-scheduling:
//     few atomics on the one hot counter while there is plenty of work — every access to it is
//     serialised with all other workgroups' — and single units at the end, so that the
//     workgroups run dry close together);
//   * its waves take one unit at a time from the run through LDS, so that all waves of a
//     workgroup run out within one unit of each other: a workgroup's LDS (the scene copy) is
//     only released to the next launch's workgroup when its last wave has left.
// Everything per unit is wave-uniform.
struct BlockPool {
    unsigned long long range; // next unit | end unit << 32 of the workgroup's current run
    unsigned lock;            // 1 while a wave is fetching the next run
    unsigned dry;             // 1 once the global counter has run out
    unsigned last_end;        // end of the previous run: how far the launch was then
    unsigned pad;
};

struct WavePool {
    unsigned k, k_end; // next / end path within the current unit (k_end = 0: none)
    unsigned first;    // first pixel of the unit
    unsigned s_base;   // its local sample
    unsigned x0, y0;   // pixel coordinates of `first`
    bool exhausted;
};

// The caller synchronises the workgroup before the first pool_refill().
__device__ __forceinline__ void block_pool_init(BlockPool *bp) {
    if (threadIdx.x == 0) {
        bp->range = 0ull;
        bp->lock = bp->dry = bp->last_end = bp->pad = 0u;
    }
}

__device__ __forceinline__ void pool_init(WavePool &wp) {
    wp.k = wp.k_end = 0;
    wp.first = wp.s_base = wp.x0 = wp.y0 = 0;
    wp.exhausted = false;
}

#define RT_LDS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define RT_LDS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// Makes sure the wave holds a unit with paths left, if the launch has any; it may come back
// empty-handed without the launch being over (another wave is fetching the workgroup's next
// run): the caller tries again on its next turn.  Wave-uniform.
// Returns true when the wave has just taken a new unit.
__device__ __forceinline__ bool pool_refill(WavePool &wp, BlockPool *bp, const RenderParams &rp, int lane) {
    if (wp.k < wp.k_end || wp.exhausted) return false;
    const unsigned NOTHING = ~0u, DRY = ~0u - 1u;
    unsigned unit = NOTHING;
    if (lane == 0) {
        for (;;) {
            unsigned long long r = RT_LDS_LOAD(&bp->range);
            if ((unsigned)r < (unsigned)(r >> 32)) {
                if (atomicCAS(&bp->range, r, r + 1ull) == r) {
                    unit = (unsigned)r;
                    break;
                }
                continue;
            }
            if (RT_LDS_LOAD(&bp->dry) != 0u) {
                unit = DRY;
                break;
            }
            if (atomicCAS(&bp->lock, 0u, 1u) != 0u) break; // a run is on its way
            r = RT_LDS_LOAD(&bp->range);                    // it may have arrived just before
            if ((unsigned)r < (unsigned)(r >> 32) || RT_LDS_LOAD(&bp->dry) != 0u) {
                RT_LDS_STORE(&bp->lock, 0u);
                continue;
            }
            // the counter is read only through the reservations themselves: the estimate of
            // what is left lags by one run
            const unsigned seen = RT_LDS_LOAD(&bp->last_end);
            const unsigned left = rp.n_units > seen ? rp.n_units - seen : 0u;
            unsigned g = left / (RT_POOL_GUIDE_DIV * gridDim.x);
            g = g < 1u ? 1u : (g > RT_POOL_MAX_UNITS ? RT_POOL_MAX_UNITS : g);
            const unsigned base = atomicAdd(rp.next_chunk, g);
            if (base >= rp.n_units) {
                RT_LDS_STORE(&bp->dry, 1u);
                RT_LDS_STORE(&bp->lock, 0u);
                unit = DRY;
                break;
            }
            if (g > rp.n_units - base) g = rp.n_units - base;
            RT_LDS_STORE(&bp->last_end, base + g);
            // this wave keeps the first unit; the others find the rest
            RT_LDS_STORE(&bp->range, ((unsigned long long)(base + g) << 32) | (unsigned long long)(base + 1u));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            RT_LDS_STORE(&bp->lock, 0u);
            unit = base;
            break;
        }
    }
    unit = wave_bcast32(unit);
    if (unit == NOTHING) return false;
    if (unit == DRY) {
        wp.exhausted = true;
        return false;
    }
    unsigned s0, strip0;
    if (rp.pixel_major) { // all samples of a strip in a row: the rays of a workgroup stay together
        strip0 = unit / (unsigned)rp.sample_count;
        s0 = unit - strip0 * (unsigned)rp.sample_count;
        if (rp.strip_order) strip0 = rp.strip_order[strip0]; // (a permutation: costly strips first)
    } else {
        s0 = unit / rp.n_strips;
        strip0 = unit - s0 * rp.n_strips;
    }
    wp.s_base = s0;
    wp.k = 0;
    wp.k_end = 64u;
    wp.first = strip0 * 64u;
    wp.y0 = wp.first / (unsigned)rp.width;
    wp.x0 = wp.first - wp.y0 * (unsigned)rp.width;
    return true;
}

#undef RT_LDS_LOAD
#undef RT_LDS_STORE

// Decodes path `k` of the current unit; false if it falls outside the image (the last
// strip of a sample is padded to 64 pixels).
__device__ __forceinline__ bool pool_decode(const WavePool &wp, const RenderParams &rp, unsigned k,
                                            unsigned &pixel, unsigned &s_local, int &px, int &py) {
    pixel = wp.first + k;
    s_local = wp.s_base;
    unsigned x = wp.x0 + k, y = wp.y0;
    while (x >= (unsigned)rp.width) {
        x -= (unsigned)rp.width;
        y++;
    }
    px = (int)x;
    py = (int)y;
    return pixel < (unsigned)rp.n_pixels;
}

__device__ __forceinline__ void write_record(const RenderParams &rp, unsigned s_local, unsigned pixel, f3 L,
                                             unsigned n_rays) {
    f4 rec;
    rec.x = L.x;
    rec.y = L.y;
    rec.z = L.z;
    rec.w = __builtin_bit_cast(float, n_rays);
    rp.stage[(size_t)s_local * (size_t)rp.n_pixels + pixel] = rec;
}

} // namespace rtc
