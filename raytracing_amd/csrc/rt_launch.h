// rt_launch.h — launch parameters shared by the host library and the kernels.
#pragma once

#include <hip/hip_runtime_api.h>

#include "rt_core.h"

#define RT_MEGA_MAX_THREADS 1024
#define RT_TRACE_THREADS 256
#define RT_POOL_CHUNK 256ull // paths a wave reserves from the global counter at a time

struct RenderParams {
    int width, height, n_pixels;
    int sample_begin; // global index of the first sample of this launch
    int sample_count; // samples per pixel in this launch
    uint32_t seed;
    unsigned long long n_paths;    // ceil(n_pixels/64) * 64 * sample_count (64-pixel strips)
    rtc::f4 *stage;                // [sample_count][n_pixels] {r g b rays}
    unsigned long long *next_path; // pool counter, zeroed before the launch
    rtc::Counters *counters;       // non-null: count tests (instrumented kernel variant)
};

struct LaunchGeom {
    int grid, block;
    int lds_bytes;
    bool lds_scene;
};

namespace rtl {
hipError_t launch_mega(const rtc::DevScene &sc, const rtc::DevCamera &cam, const RenderParams &rp,
                       const LaunchGeom &lg, hipStream_t stream);
hipError_t set_mega_lds_limit(int bytes);
hipError_t launch_reduce(const rtc::f4 *stage, int n_pixels, int sample_count, float *accum,
                         unsigned long long *ray_count, hipStream_t stream);
hipError_t launch_trace(const rtc::DevScene &sc, const rt_ray *rays, long long n, float tmin, float tmax,
                        rt_hit *hits, hipStream_t stream);
} // namespace rtl
