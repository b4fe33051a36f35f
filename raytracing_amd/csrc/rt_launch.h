// rt_launch.h — launch parameters shared by the host library and the kernels.
#pragma once

#include <hip/hip_runtime_api.h>

#include "rt_core.h"

#ifndef RT_MEGA_MAX_THREADS
#define RT_MEGA_MAX_THREADS 1024 // lanes of a workgroup of the persistent render kernels (experiments: 512 ... 1024)
#endif
#ifndef RT_MEGA_WAVES_PER_SIMD
#define RT_MEGA_WAVES_PER_SIMD 4
#endif
#define RT_TRACE_THREADS 256
#ifndef RT_POOL_MAX_UNITS
#define RT_POOL_MAX_UNITS 32u // a workgroup reserves at most this many units of 64 paths at a time
#endif
#ifndef RT_POOL_GUIDE_DIV
#define RT_POOL_GUIDE_DIV 2u // ... and at most 1/RT_POOL_GUIDE_DIV of its share of what is left
#endif
#define RT_BLOCK_POOL_BYTES 64 // static LDS of the render kernels besides the dynamic part
// The statically allocated device block behind rt_render_device_static (rt_static.hip): a header
// of RT_STATIC_HEADER_BYTES (8-byte word 0: the launch's pool counter, word 1: its ray counter)
// followed by room for RT_STATIC_MAX_RECORDS per-path radiance records of 16 bytes.
#define RT_STATIC_HEADER_BYTES 4096u
#define RT_STATIC_MAX_RECORDS (4u << 20)
#define RT_STATIC_SCRATCH_BYTES (RT_STATIC_HEADER_BYTES + 16u * RT_STATIC_MAX_RECORDS)

struct RenderParams {
    int width, height, n_pixels;
    int sample_begin; // global index of the first sample of this launch
    int sample_count; // samples per pixel in this launch
    uint32_t seed;
    unsigned n_strips;             // ceil(n_pixels / 64)
    unsigned n_units;              // n_strips * sample_count: units of 64 paths (rt_wavepool.h)
    int pixel_major;               // order of the units: 0 sample-major, 1 pixel-major
    const unsigned *strip_order;   // pixel-major: the strips in the order they are handed out (null: as they lie)
    int ring_f4;                   // camera-ray rings (LaunchGeom::ring): their offset in dynamic LDS, 16-byte words
    int coop_f4;                   // staging areas of the cooperative node fetch (LaunchGeom::coop): their offset
    int stack_lds_entries;         // BVH scenes: stack entries per lane kept in LDS
    int *stack_spill;              // ... the rest: [block][entry][lane of the block] (null: none)
    rtc::f4 *stage;                // [sample_count][n_pixels] {r g b rays}
    unsigned *next_chunk;          // pool counter, zeroed before the launch
    rtc::Counters *counters;       // non-null: count tests (instrumented kernel variant)
    int scheduler; // BVH scenes: 0 = one whole ray per wave iteration, 1 = decoupled traverse / shade
    int min_ready; // decoupled: lanes that must wait for shading before traversal is interrupted
    int leaf_min;  // decoupled: lanes that must hold a parked leaf before the leaf block runs
    int blocked_min; // decoupled: ... or this many lanes stalled on a second leaf
    int regen_min;   // decoupled: idle lanes that must accumulate before new paths are started
    int shade_min;   // whole-ray kernels: hits that must wait before they are shaded (1: every iteration)
    // Debugging aid (non-null): every wave of the decoupled kernel leaves {start, out of new
    // paths, exit} in 100 MHz wall-clock ticks and its hardware id, 4 x 8 bytes per wave.
    unsigned long long *timeline;
};

struct LaunchGeom {
    int grid, block;
    int lds_bytes;
    bool lds_scene;  // slots + nodes are staged into LDS
    int lds_records; // ... and so are the per-slot records and the texture table
    int lds_perlins = 0; // ... and the Perlin tables
    int stack_cap = 0;         // entries a traversal stack needs for this scene's tree
    int stack_lds_entries = 0; // ... of which this many live in LDS (the rest spills to global memory)
    // decoupled kernel: every wave keeps the 64 camera rays of its current unit in LDS (generated
    // by all lanes at once when the unit is taken; idle lanes then only fetch theirs)
    bool ring = false;
    int ring_f4 = 0; // offset of the rings in dynamic LDS, in 16-byte words
    // linear scenes that are mostly scan: the kernel built for 8 waves per SIMD.  Pipelined launches
    // keep one workgroup per CU each (two launches share a CU); a launch on its own starts two.
    bool dense = false;
    int shade_min = 1; // whole-ray kernels: hits that wait before they are shaded together (RenderParams::shade_min)
    // scenes read from HBM: the waves read their nodes together through per-wave staging areas in LDS
    // (rt_sched.h coop_fetch_nodes); the tree's top is then not cached in LDS
    bool coop = false;
    int coop_f4 = 0; // offset of the staging areas in dynamic LDS, in 16-byte words
};
#define RT_RING_F4_PER_WAVE (3 * 64) // {origin, time} {direction, sampler state} {pixel, local sample, -, -}
#define RT_COOP_F4_PER_WAVE (32 * 8) // staging of the cooperative node fetch: 32 nodes of 8 x 16 bytes per wave

// Path state of the wavefront variant: structure-of-arrays over the in-flight path slots.
struct WfBuffers {
    rtc::f4 *state;  // 64 bytes per slot: {origin, time} {direction, -} {throughput, sampler state}
                     //                    {radiance so far, bounces left | rays traced << 16}
    rtc::i4 *aux;    // 16 bytes per slot: {hit t, hit slot, pixel, local sample index}
    int *key;        // per queue position: queue the traced ray goes to
    int *q_trace[2]; // rays to trace this round / next round
    int *q_mat[5];   // hits by material (lambertian, metal, dielectric, light) and misses
    int *q_free;     // slots without a path
    unsigned *cnt;   // 16 counters (queue lengths, pool position, finished paths)
};
#define RT_WF_BYTES_PER_SLOT (64 + 16 + 4 + 8 * 4)

namespace rtl {
hipError_t set_wavefront_lds_limit(int bytes);
hipError_t render_wavefront(const rtc::DevScene &sc, const rtc::DevCamera &cam, const RenderParams &rp,
                            const WfBuffers &wf, unsigned n_slots, const LaunchGeom &lg, unsigned *host_cnt,
                            hipStream_t stream, int *rounds_out);
hipError_t set_wavefront_lds_limit_inst(int bytes);
hipError_t render_wavefront_inst(const rtc::DevScene &sc, const rtc::DevCamera &cam, const RenderParams &rp,
                                 const WfBuffers &wf, unsigned n_slots, const LaunchGeom &lg, unsigned *host_cnt,
                                 hipStream_t stream, int *rounds_out);
hipError_t launch_mega(const rtc::DevScene &sc, const rtc::DevCamera &cam, const RenderParams &rp,
                       const LaunchGeom &lg, hipStream_t stream);
hipError_t set_mega_lds_limit(int bytes);
// the same kernels compiled without the quad / noise / image branches (rt_scene::lean)
hipError_t launch_mega_lean(const rtc::DevScene &sc, const rtc::DevCamera &cam, const RenderParams &rp,
                            const LaunchGeom &lg, hipStream_t stream);
hipError_t set_mega_lds_limit_lean(int bytes);
// ... and with the two-level traversal of scenes that have instances (DevScene::n_inst > 0)
hipError_t launch_mega_inst(const rtc::DevScene &sc, const rtc::DevCamera &cam, const RenderParams &rp,
                            const LaunchGeom &lg, hipStream_t stream);
hipError_t set_mega_lds_limit_inst(int bytes);
int mega_waves_per_simd(); // register budget the render kernels were compiled for
int mega_blocks_per_cu(int block, int lds_bytes, bool lds_scene);
hipError_t launch_reduce(const rtc::f4 *stage, int n_pixels, int sample_count, float *accum,
                         unsigned long long *ray_count, unsigned *strip_cost, hipStream_t stream);
size_t order_strips_tmp_bytes(unsigned n);
hipError_t fill_iota(unsigned *v, unsigned n, hipStream_t stream);
hipError_t order_strips(const unsigned *cost, const unsigned *iota, unsigned n, unsigned *cost_sorted, unsigned *order,
                        void *tmp, size_t tmp_bytes, hipStream_t stream);
hipError_t build_lbvh(const float *boxes, const int *types, int n, float *nodes_out, int *leaf_order,
                      int *inner_levels);
hipError_t static_scratch_setup(int device); // the device must be current
void *static_scratch(int device);
hipError_t launch_accum_add(float *dst, const float *src, long long n, hipStream_t stream);
hipError_t launch_tonemap(const float *accum, long long n_components, float scale, uint8_t *rgb8,
                          hipStream_t stream);
hipError_t launch_trace(const rtc::DevScene &sc, const rt_ray *rays, long long n, float tmin, float tmax,
                        rt_hit *hits, hipStream_t stream);
hipError_t launch_trace_sched(const rtc::DevScene &sc, const RenderParams &rp, const LaunchGeom &lg, const rt_ray *rays,
                              long long n, float tmin, float tmax, rt_hit *hits, hipStream_t stream);
hipError_t launch_closest(const rtc::DevScene &sc, const RenderParams &rp, const LaunchGeom &lg, const rt_ray *rays,
                          long long n, float tmin, float tmax, int *prim, float *t, hipStream_t stream);
hipError_t launch_closest_inst(const rtc::DevScene &sc, const RenderParams &rp, const LaunchGeom &lg, const rt_ray *rays,
                               long long n, float tmin, float tmax, int *prim, float *t, hipStream_t stream);
hipError_t launch_trace_inst(const rtc::DevScene &sc, const rt_ray *rays, long long n, float tmin, float tmax,
                             rt_hit *hits, hipStream_t stream);
hipError_t launch_trace_sched_inst(const rtc::DevScene &sc, const RenderParams &rp, const LaunchGeom &lg,
                                   const rt_ray *rays, long long n, float tmin, float tmax, rt_hit *hits,
                                   hipStream_t stream);
} // namespace rtl
