// rt_wavepool.h — device-only helpers shared by the persistent render kernels: the per-wave
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

// Per-wave pool of paths.  The global counter hands out chunks of 64 pixels x
// RT_POOL_SAMPLES samples; everything per chunk is wave-uniform.
struct WavePool {
    unsigned k, k_end;      // next / end path within the current chunk (k_end = 0: no chunk)
    unsigned strip, s_base; // first pixel = strip * 64, first local sample of the chunk
    unsigned x0, y0;        // pixel coordinates of the strip's first pixel
    bool exhausted;
};

__device__ __forceinline__ void pool_init(WavePool &wp) {
    wp.k = wp.k_end = 0;
    wp.strip = wp.s_base = wp.x0 = wp.y0 = 0;
    wp.exhausted = false;
}

// Makes sure the wave holds a chunk with paths left, if any remain.  Wave-uniform.
__device__ __forceinline__ void pool_refill(WavePool &wp, const RenderParams &rp, int lane) {
    if (wp.k < wp.k_end || wp.exhausted) return;
    unsigned chunk = 0;
    if (lane == 0) chunk = atomicAdd(rp.next_chunk, 1u);
    chunk = wave_bcast32(chunk);
    if (chunk >= rp.n_chunks) {
        wp.exhausted = true;
        return;
    }
    unsigned strip = chunk / rp.n_sample_groups;
    unsigned sg = chunk - strip * rp.n_sample_groups;
    wp.strip = strip;
    wp.s_base = sg * RT_POOL_SAMPLES;
    unsigned left = (unsigned)rp.sample_count - wp.s_base;
    wp.k = 0;
    wp.k_end = 64u * (left < RT_POOL_SAMPLES ? left : RT_POOL_SAMPLES);
    unsigned first = strip * 64u;
    wp.y0 = first / (unsigned)rp.width;
    wp.x0 = first - wp.y0 * (unsigned)rp.width;
}

// Decodes path index `k` of the current chunk; false if it falls outside the image.
__device__ __forceinline__ bool pool_decode(const WavePool &wp, const RenderParams &rp, unsigned k,
                                            unsigned &pixel, unsigned &s_local, int &px, int &py) {
    unsigned j = k & 63u;
    s_local = wp.s_base + (k >> 6);
    pixel = wp.strip * 64u + j;
    unsigned x = wp.x0 + j, y = wp.y0;
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
