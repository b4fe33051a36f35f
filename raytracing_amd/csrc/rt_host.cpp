// rt_host.cpp — host half of librt_amd.so: the C ABI of include/rt_amd.h.
//
// Flattens an rt_scene_desc into the device layout of rt_core.h (primitive
// slots, BVH nodes, wrapper ops, material / texture tables), uploads it once,
// and enqueues the kernels of rt_kernels.hip.  There is no CPU rendering path
// in this library: without a HIP device every compute entry point fails.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_core.h"
#include "rt_flatten.h"
#include "rt_validate.h"
#include "rt_launch.h"
#include "rt_scene_io.h"

using rtc::f3;
using rtc::f4;
using rtc::i4;
using rtc::mk;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

// No C++ exception may cross the C ABI: bad_alloc / length_error from the host-side containers
// become error codes.
template <class F> int guarded(const char *what, F body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(RT_ERR_HIP, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return fail(RT_ERR_ARG, "%s: %s", what, e.what());
    } catch (...) {
        return fail(RT_ERR_ARG, "%s: unknown failure", what);
    }
}

// (a failed call also leaves its code behind for hipGetLastError(), which the kernel launchers
// use: taken out here, so that it is not reported again by an unrelated launch later)
#define RT_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError();                                                               \
            return fail(RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));                       \
        }                                                                                          \
    } while (0)

const int kTimelineLaunches = 32;
const int kMaxLdsBytes = 160 * 1024 - RT_BLOCK_POOL_BYTES; // LDS per CU on gfx950, less the kernels' static part

} // namespace

// ---------------------------------------------------------------------------------------
// Record sets that pipelined launches rotate through (rt_scene), and each set's counter words
#ifndef RT_SETS
#define RT_SETS 3
#endif
constexpr int kSets = RT_SETS;
static_assert(kSets >= 2 && kSets <= 4, "two to four record sets");
constexpr int kPoolWord[4] = {0, 2, 8, 10};

struct rt_scene {
    int device = 0;
    rtc::DevScene dev;
    rt_scene_info info;
    std::vector<void *> allocs;
    // render scratch (owned by the scene, grown on demand)
    // kSets sets of per-launch scratch: consecutive megakernel launches go to internal streams
    // of their own, so that launch k+1 fills the CUs that the tail of launch k (its last long
    // paths) leaves idle; the reduction into the caller's image stays on the caller's stream.
    // The third set keeps launch k+2 from waiting for the reduction of launch k, which itself
    // cannot start before launch k+1 (whose workgroups fill every CU) begins to leave.
    f4 *stage[kSets] = {};
    size_t stage_bytes[kSets] = {};
    hipStream_t mega_stream[kSets] = {};
    hipEvent_t ev_done[kSets] = {}; // megakernel finished (recorded on its internal stream)
    hipEvent_t ev_free[kSets] = {}; // the set's records have been reduced (caller's stream)
    bool ev_free_valid[kSets] = {};
    int flip = 0;
    // cost-ordered strips (per record set): reduce_stage leaves a cost per 64-pixel strip, sorted
    // into the order in which the set's next launch of the same image hands the strips out
    unsigned *strip_cost[kSets] = {}, *strip_order[kSets] = {};
    unsigned strip_valid[kSets] = {}; // strips the order covers (0: none yet)
    unsigned *strip_iota = nullptr, *strip_sorted = nullptr;
    void *strip_tmp = nullptr;
    size_t strip_tmp_bytes = 0;
    unsigned strip_cap = 0;
    // scenes read from HBM with a tree too deep for the LDS part of the stacks: the rest of every
    // lane's stack, one buffer per record set and one for rt_render_device_static
    int *stack_spill[kSets + 1] = {};
    // counters: word 1 counts rays, 6 is the batch query's pool counter; per record set a pool counter
    // (kPoolWord)
    unsigned long long *d_words = nullptr;
    rtc::Counters *d_counters = nullptr;
    int n_cu = 256;
    int tree_top = 0; // HostScene::n_top: how many leading nodes are the most visited ones
    bool lean = false; // no quads, no noise / image textures: the lean render kernels apply
    bool noise = false; // some material reaches a noise texture (shading then costs far more than a short scan)
    LaunchGeom geom; // persistent render kernel: grid, block, LDS bytes, what is staged
    // wavefront variant: path-state arrays and queues (allocated on first use)
    void *wf_mem = nullptr;
    unsigned wf_slots = 0;
    WfBuffers wf;
    unsigned *wf_host_cnt = nullptr; // pinned
    // RT_AMD_TIMELINE=<file>: per-wave wall-clock records of the last kTimelineLaunches launches,
    // written to the file when the scene is destroyed (tools/timeline.py reads it)
    unsigned long long *d_timeline = nullptr;
    int timeline_waves = 0, timeline_block_waves = 0;
    long long timeline_launches = 0;
    // optional per-launch timing of the render kernel
    bool profiling = false;
    std::vector<hipEvent_t> prof_events; // pairs
};

namespace {

template <class T> int upload(rt_scene *s, const std::vector<T> &v, const T **out) {
    // rounded up to whole 16-byte words: the kernels stage tables into LDS 16 bytes at a time
    size_t bytes = std::max<size_t>((v.size() * sizeof(T) + 15) / 16 * 16, 16);
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    if (!v.empty()) RT_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    s->info.device_bytes += bytes;
    *out = (const T *)p;
    return RT_OK;
}

int validate(const rt_scene_desc *d) {
    std::string err;
    const int rc = rtf::validate_desc(d, err);
    return rc == RT_OK ? RT_OK : fail(rc, "%s", err.c_str());
}

// Bytes of LDS the staged arrays take: slots, nodes, both per-slot records, textures
// (the latter padded to 16 bytes).  Must match the kernels' staging order.
size_t scene_lds_bytes(const rtc::DevScene &dv) {
    size_t words = (size_t)dv.n_slots * dv.slot_f4 + (size_t)dv.n_nodes * RT_NODE_LDS_F4;
#ifndef RT_REC_GLOBAL
    words += (size_t)dv.n_slots * 2;
#endif
    size_t tex = ((size_t)dv.n_textures * sizeof(rt_texture) + 15) / 16;
    return (words + tex) * sizeof(f4);
}

int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Dynamic LDS one workgroup may use when `bpc` of them share a CU: each also holds the kernels'
// static part, and the hardware hands LDS out in blocks (1 KB of slack covers the rounding).
size_t lds_budget(int bpc) {
    if (bpc <= 1) return (size_t)kMaxLdsBytes;
    return (size_t)(160 * 1024) / bpc - RT_BLOCK_POOL_BYTES - 1024;
}

// Workgroup shape of the persistent render kernel and what it stages into LDS.  The
// kernels are compiled for mega_waves_per_simd() waves per SIMD; a CU is filled with
// that many waves if the per-workgroup LDS (scene copy + traversal stacks) allows it.
LaunchGeom choose_geom(rt_scene *s, uint32_t flags) {
    LaunchGeom lg;
    rtc::DevScene &dv = s->dev;
    const bool bvh = dv.traversal == rtc::TRAV_BVH;
    const int cap = bvh ? RT_STACK_CAP(s->info.bvh_depth) : 0; // stack entries a ray can need

    const size_t all_bytes = scene_lds_bytes(dv);
    const int block = std::max(64, std::min(RT_MEGA_MAX_THREADS, env_int("RT_AMD_BLOCK", RT_MEGA_MAX_THREADS) / 64 * 64));
    const int threads_per_cu = rtl::mega_waves_per_simd() * 256;
    const int bpc_target = std::max(1, env_int("RT_AMD_BLOCKS_PER_CU", threads_per_cu / block));
    // 16-bit stack entries (instances and the mark that ends a visit to one are leaf "slots" too)
    const bool short_ok = dv.n_nodes < 32768 && dv.n_slots + dv.n_inst + 1 < 16384;
    lg.block = block;
    lg.lds_scene = false;
    lg.lds_records = 0;
    lg.stack_cap = cap;
    lg.stack_lds_entries = cap;
    int bpc = bpc_target;
    // the render kernels' camera-ray rings (one per wave) come after the stacks if there is room
    const size_t ring_bytes = env_int("RT_AMD_RING", 1) != 0
                                  ? (size_t)(block / 64) * RT_RING_F4_PER_WAVE * sizeof(f4) : 0;
    lg.ring = false;
    lg.ring_f4 = 0;
    size_t used = 0; // scene copy / tree top + stacks: where the rings start
    if (!(flags & RT_SCENE_FORCE_GLOBAL) && short_ok) {
        for (bpc = bpc_target; bpc >= 1 && !lg.lds_scene; bpc--) {
            size_t budget = lds_budget(bpc);
            size_t stack = (size_t)cap * RT_MEGA_MAX_THREADS * sizeof(short); // rows of the full workgroup size
            // A scene held in LDS is held there whole, records and texture table included: the kernels
            // then read them with LDS instructions (a pointer that may be either LDS or global memory
            // makes every record access a flat load with 64-bit address arithmetic).  The rings yield
            // first; a scene whose geometry alone would fit is read from HBM with the tree's top in LDS.
            if (all_bytes + stack <= budget) {
                lg.lds_scene = true;
                lg.lds_records = 1;
                used = all_bytes + stack;
            }
            if (lg.lds_scene) {
                lg.ring = ring_bytes && used + ring_bytes <= budget;
                // the Perlin tables come along if everything else fits (they follow the records)
                const size_t perlin_bytes = (size_t)dv.n_perlins * sizeof(rtc::DevPerlin);
                if (lg.lds_records && perlin_bytes > 0 && env_int("RT_AMD_LDS_PERLIN", 1) != 0 &&
                    used + perlin_bytes + (lg.ring ? ring_bytes : 0) <= budget) {
                    lg.lds_perlins = 1;
                    used += perlin_bytes;
                }
                break;
            }
        }
    }
    dv.n_top = 0;
    dv.ref_bits = 16;
    if (!lg.lds_scene) {
        // Geometry stays in HBM.  LDS holds the top entries of the 32-bit stacks (the rest of a
        // deep tree's stack, which few rays ever reach, spills to global memory), the rings and
        // the most visited nodes of the tree.
        int bits = 1;
        while ((1ll << bits) < std::max<long long>(dv.n_nodes, 2ll * (dv.n_slots + dv.n_inst + 1))) bits++;
        dv.ref_bits = std::max(16, std::min(32, bits + 1));
        // Cooperative node fetch (rt_sched.h coop_fetch_nodes), RT_AMD_COOP=1: 4 KB of staging per wave,
        // which a shorter LDS part of the stacks makes room for; the tree's top is then not kept in LDS.
        // Off by default: measured 34 % slower than the lane-by-lane loads on the 10^6-primitive scene
        // (the memory pipeline takes a wave's 64 x 16-byte requests four lanes a cycle however they
        // coalesce, so eight cooperative loads cost what seven private ones do, and the trip through
        // LDS comes on top).  Two-level scenes always keep the lane-by-lane loads.
        const size_t coop_bytes = (size_t)(block / 64) * RT_COOP_F4_PER_WAVE * sizeof(f4);
        const bool coop = bvh && ring_bytes && block == 1024 && env_int("RT_AMD_COOP", 0) != 0 && s->dev.n_inst == 0;
        lg.stack_lds_entries = bvh ? std::max(4, std::min(cap, env_int("RT_AMD_LDS_STACK", coop ? 10 : 16))) : 0;
        size_t stack = (size_t)lg.stack_lds_entries * RT_MEGA_MAX_THREADS * sizeof(int);
        bpc = bpc_target;
        while (bpc > 1 && lds_budget(bpc) < stack + ring_bytes) bpc--;
        size_t budget = lds_budget(bpc);
        lg.ring = ring_bytes && stack + ring_bytes <= budget;
        used = stack;
        if (coop && lg.ring && bpc == 1 && stack + ring_bytes + coop_bytes <= budget) {
            lg.coop = true;
        } else if (bvh && s->tree_top > 0 && env_int("RT_AMD_TREE_TOP", 1) != 0) {
            size_t taken = stack + (lg.ring ? ring_bytes : 0);
            size_t room = budget > taken ? budget - taken : 0;
            dv.n_top = (int)std::min<size_t>((size_t)s->tree_top, room / (RT_NODE_LDS_F4 * sizeof(f4)));
            dv.n_top = std::min(dv.n_top, std::max(0, env_int("RT_AMD_TREE_TOP_NODES", dv.n_top)));
            used += (size_t)dv.n_top * RT_NODE_LDS_F4 * sizeof(f4);
        }
    }
    // scans of many quads (or of nothing but quads) run the kernel built for 8 waves per SIMD:
    // measured +29 % on 18 quads, +9 % on 5, against -6 ... -14 % on scenes of one to three
    // (textured) spheres, which keep the 4-wave build
    lg.dense = !bvh && bpc == 1 && lg.block == 1024 && (s->info.n_spheres == 0 || dv.n_slots >= 8) &&
               env_int("RT_AMD_DENSE", 1) != 0 && dv.n_slots > 0;
    // scanned scenes with noise textures: hits wait until 48 lanes have one (measured on the reference's
    // two: +31 % and +9 %; scenes with cheap shading lose 5-15 % to the extra passes of the loop and
    // keep shading every iteration)
    lg.shade_min = !bvh && !lg.dense && s->noise ? 48 : 1;
    dv.stack_cap = cap;
    used = (used + 15) / 16 * 16;
    lg.ring_f4 = (int)(used / sizeof(f4));
    lg.lds_bytes = (int)(used + (lg.ring ? ring_bytes : 0));
    if (lg.coop) {
        lg.coop_f4 = lg.lds_bytes / (int)sizeof(f4);
        lg.lds_bytes += (block / 64) * RT_COOP_F4_PER_WAVE * (int)sizeof(f4);
    }
    lg.lds_bytes = std::max(lg.lds_bytes, 16);
    lg.grid = s->n_cu * std::max(bpc, 1);
    return lg;
}

int set_device(int device) {
    RT_HIP(hipSetDevice(device));
    return RT_OK;
}

} // namespace

// =========================================================================================
extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }
// A process has four hardware queues per GPU by default, shared by all of its streams.  A scene
// uses four streams itself (three internal ones and the caller's); with a communication library's
// streams in the same process two of the internal ones end up on one queue, where launches cannot
// overlap (measured: 7.4 instead of 8.3 Gpaths/s with RCCL initialised).  Eight queues are enough.
// This only helps if the library is loaded before the HIP runtime reads its settings (a C++
// program linked against it; Python harnesses set the variable themselves before importing torch),
// and never overrides a value the user has set.  RT_AMD_KEEP_HW_QUEUES=1 leaves the environment
// alone; RT_AMD_DEBUG=1 says on stderr when the variable was set here.
__attribute__((constructor)) static void rt_amd_more_hw_queues() {
    const char *keep = getenv("RT_AMD_KEEP_HW_QUEUES");
    if (keep && atoi(keep) != 0) return;
    if (getenv("GPU_MAX_HW_QUEUES")) return;
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const char *dbg = getenv("RT_AMD_DEBUG");
    if (dbg && atoi(dbg) != 0)
        fprintf(stderr, "librt_amd: set GPU_MAX_HW_QUEUES=8 for this process (RT_AMD_KEEP_HW_QUEUES=1 prevents it)\n");
}

int rt_abi_version(void) { return RT_AMD_ABI_VERSION; }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_camera_initialize(const rt_camera_desc *c, rt_camera_frame *f) { // camera.h:76-111
    if (!c || !f) return fail(RT_ERR_ARG, "null camera");
    if (c->image_width < 1 || c->samples_per_pixel < 1 || !(c->aspect_ratio > 0.0f))
        return fail(RT_ERR_ARG, "camera: image_width, samples_per_pixel and aspect_ratio must be positive");
    if (!((double)c->image_width / (double)c->aspect_ratio < 1073741824.0)) // (camera.h:77 converts it to int)
        return fail(RT_ERR_ARG, "camera: image_width / aspect_ratio = %g is no image height", (double)c->image_width / (double)c->aspect_ratio);
    rtf::camera_frame(c, f);
    return RT_OK;
}

int rt_scene_destroy(rt_scene *s) {
    if (!s) return RT_OK;
    hipSetDevice(s->device);
    for (void *p : s->allocs) hipFree(p);
    hipDeviceSynchronize();
    for (int b = 0; b < kSets; b++) {
        if (s->stage[b]) hipFree(s->stage[b]);
        if (s->strip_cost[b]) hipFree(s->strip_cost[b]);
        if (s->strip_order[b]) hipFree(s->strip_order[b]);
        if (s->mega_stream[b]) hipStreamDestroy(s->mega_stream[b]);
        if (s->ev_done[b]) hipEventDestroy(s->ev_done[b]);
        if (s->ev_free[b]) hipEventDestroy(s->ev_free[b]);
    }
    if (s->d_timeline) {
        const char *path = getenv("RT_AMD_TIMELINE");
        const size_t words = (size_t)kTimelineLaunches * s->timeline_waves * 4;
        std::vector<unsigned long long> host(words + 4);
        host[0] = (unsigned long long)std::min<long long>(s->timeline_launches, kTimelineLaunches);
        host[1] = (unsigned long long)s->timeline_waves;
        host[2] = (unsigned long long)s->timeline_block_waves;
        host[3] = (unsigned long long)s->timeline_launches; // the ring's next position is this modulo its size
        if (path && hipMemcpy(host.data() + 4, s->d_timeline, words * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(path, "wb")) {
                fwrite(host.data(), 8, host.size(), f);
                fclose(f);
            }
        }
        hipFree(s->d_timeline);
    }
    if (s->strip_iota) hipFree(s->strip_iota);
    if (s->strip_sorted) hipFree(s->strip_sorted);
    if (s->strip_tmp) hipFree(s->strip_tmp);
    if (s->wf_mem) hipFree(s->wf_mem);
    if (s->wf_host_cnt) hipHostFree(s->wf_host_cnt);
    for (hipEvent_t e : s->prof_events) hipEventDestroy(e);
    delete s;
    return RT_OK;
}

static int impl_rt_scene_create(const rt_scene_desc *d, int device, uint32_t flags, rt_scene **out) {
    if (!out) return fail(RT_ERR_ARG, "out is null");
    *out = nullptr;
    int rc = validate(d);
    if (rc != RT_OK) return rc;
    int n_dev = rt_device_count();
    if (n_dev <= 0) return fail(RT_ERR_HIP, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return fail(RT_ERR_ARG, "device %d out of range (%d visible)", device, n_dev);
    rc = set_device(device);
    if (rc != RT_OK) return rc;

    rtf::HostScene hs;
    const auto t_build0 = std::chrono::steady_clock::now();
    {
        std::string err;
        rtf::ExternalBuilder device_builder = nullptr;
        if ((flags & RT_SCENE_BUILD_DEVICE) || env_int("RT_AMD_BUILD_DEVICE", 0))
            device_builder = [&](const std::vector<rtf::Box3> &boxes, const std::vector<int> &types, rtf::BvhOut &bo) {
                static_assert(sizeof(rtf::Box3) == 6 * sizeof(float), "Box3 is six floats");
                const int np = (int)boxes.size();
                if (np < 2) return false;
                bo.leaf_order.resize((size_t)np);
                bo.nodes.resize((size_t)(np - 1) * 4);
                int levels = 0;
                hipError_t e = rtl::build_lbvh(reinterpret_cast<const float *>(boxes.data()), types.data(), np,
                                               reinterpret_cast<float *>(bo.nodes.data()), bo.leaf_order.data(), &levels);
                if (e != hipSuccess) return false; // host builder takes over
                bo.max_inner_depth = levels - 1;
                return true;
            };
        if (env_int("RT_AMD_FORCE_BVH", 0)) flags |= RT_SCENE_FORCE_BVH; // experiments: a tree for small scenes too
        if (env_int("RT_AMD_FORCE_GLOBAL", 0)) flags |= RT_SCENE_FORCE_GLOBAL; // ... no LDS copy of the scene
        rc = rtf::flatten_scene(d, flags, hs, err, device_builder);
        if (rc != RT_OK) return fail(rc, "%s", err.c_str());
    }
    const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
    const int n = hs.n_prims;
    const bool linear = hs.linear;
    const int slot_f4 = hs.slot_f4;
    const std::vector<f4> &slots = hs.slots, &nodes = hs.nodes, &xf_packed = hs.xf;
    const std::vector<i4> &rec_i = hs.rec_i;
    const std::vector<f4> &rec_f = hs.rec_f;
    const std::vector<int> &chain_first = hs.chain_first;

    rt_scene *s = new rt_scene();
    s->device = device;
    memset(&s->info, 0, sizeof(s->info));
    memset(&s->dev, 0, sizeof(s->dev));
    s->info.n_prims = n;
    s->info.n_spheres = hs.n_spheres;
    s->info.n_quads = hs.n_quads;
    s->lean = hs.n_quads == 0 && !hs.any_chain && !env_int("RT_AMD_NO_LEAN", 0);
    for (const rt_texture &t : hs.textures) // the table holds only what a material can reach
        if (t.type == RT_TEX_NOISE || t.type == RT_TEX_IMAGE) {
            s->lean = false;
            s->noise = s->noise || t.type == RT_TEX_NOISE;
        }
    s->info.n_nodes = hs.n_nodes;
    s->info.bvh_depth = hs.bvh_depth;
    s->info.build_ms = (float)build_ms;
    s->info.built_on_device = hs.external_nodes ? 1 : 0;
    s->info.n_upfront = hs.n_big;

    const std::vector<rt_texture> &texs = hs.textures;
    std::vector<rt_image> imgs(d->images, d->images + d->n_images);
    std::vector<uint8_t> texels(d->texels, d->texels + d->n_texel_bytes);

    rtc::DevScene &dv = s->dev;
    rc = upload(s, slots, &dv.slots);
    if (rc == RT_OK) rc = upload(s, rec_i, &dv.rec_i);
    if (rc == RT_OK) rc = upload(s, rec_f, &dv.rec_f);
    if (rc == RT_OK) rc = upload(s, nodes, &dv.nodes);
    if (rc == RT_OK) rc = upload(s, xf_packed, &dv.xf);
    if (rc == RT_OK) rc = upload(s, chain_first, &dv.chain_first);
    if (rc == RT_OK) rc = upload(s, hs.scan, &dv.scan);
    if (rc == RT_OK) rc = upload(s, hs.inst, &dv.inst);
    if (rc == RT_OK) rc = upload(s, texs, &dv.textures);
    if (rc == RT_OK) rc = upload(s, hs.perlins, &dv.perlins);
    if (rc == RT_OK) rc = upload(s, imgs, &dv.images);
    if (rc == RT_OK) rc = upload(s, texels, &dv.texels);
    std::vector<unsigned long long> words(16, 0ull);
    const unsigned long long *dw = nullptr;
    if (rc == RT_OK) rc = upload(s, words, &dw);
    std::vector<rtc::Counters> cnt(1);
    memset(cnt.data(), 0, sizeof(rtc::Counters));
    const rtc::Counters *dc = nullptr;
    if (rc == RT_OK) rc = upload(s, cnt, &dc);
    if (rc != RT_OK) {
        std::string keep = g_err;
        rt_scene_destroy(s);
        g_err = keep;
        return rc;
    }
    s->d_words = const_cast<unsigned long long *>(dw);
    s->d_counters = const_cast<rtc::Counters *>(dc);
    dv.n_slots = n;
    dv.n_inst = (int)hs.inst.size();
    dv.n_big = hs.n_big;
    dv.big_quads = hs.big_quads;
    dv.n_nodes = hs.n_nodes;
    dv.n_top = 0; // chosen with the launch geometry
    s->tree_top = hs.n_top;
    dv.n_textures = (int)texs.size();
    dv.slot_f4 = slot_f4;
    dv.traversal = linear ? rtc::TRAV_LINEAR : rtc::TRAV_BVH;
    dv.n_chains = d->n_chains;
    dv.n_xf_ops = (int)(xf_packed.size() / RT_XF_F4);
    dv.n_perlins = (int)hs.perlins.size();
    dv.lds_perlins = 0; // set per launch from the launch geometry
    dv.lds_records = 0; // set per launch from the launch geometry

    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0)
        s->n_cu = n_cu;
    hipError_t e = rtl::static_scratch_setup(device); // this GPU's copy of the block rt_render_device_static uses
    if (e == hipSuccess) e = rtl::set_mega_lds_limit(kMaxLdsBytes);
    if (e == hipSuccess) e = rtl::set_mega_lds_limit_lean(kMaxLdsBytes);
    if (e == hipSuccess) e = rtl::set_mega_lds_limit_inst(kMaxLdsBytes);
    if (e == hipSuccess) e = rtl::set_wavefront_lds_limit(kMaxLdsBytes);
    if (e == hipSuccess) e = rtl::set_wavefront_lds_limit_inst(kMaxLdsBytes);
    if (e != hipSuccess) {
        rt_scene_destroy(s);
        return fail(RT_ERR_HIP, "kernel set-up (static scratch block, dynamic LDS limit): %s", hipGetErrorString(e));
    }

    s->geom = choose_geom(s, flags);
    if (s->geom.stack_cap > s->geom.stack_lds_entries) {
        const size_t bytes = (size_t)(s->geom.stack_cap - s->geom.stack_lds_entries) * s->geom.grid * RT_MEGA_MAX_THREADS * sizeof(int);
        for (int k = 0; k < kSets + 1; k++) {
            void *p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) {
                rt_scene_destroy(s);
                return fail(RT_ERR_HIP, "cannot allocate the traversal stacks' overflow area (%zu bytes)", bytes);
            }
            s->allocs.push_back(p);
            s->stack_spill[k] = (int *)p;
            s->info.device_bytes += bytes;
        }
    }
    if (env_int("RT_AMD_DEBUG", 0)) {
        fprintf(stderr, "librt_amd: render geometry grid %d block %d lds %d (scene %d records %d), resident blocks/CU %d\n",
                s->geom.grid, s->geom.block, s->geom.lds_bytes, (int)s->geom.lds_scene, s->geom.lds_records,
                rtl::mega_blocks_per_cu(s->geom.block, s->geom.lds_bytes, s->geom.lds_scene));
        const double all_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build0).count();
        fprintf(stderr, "librt_amd: scene of %d primitives: flattening and tree %.1f ms, upload and kernel set-up %.1f ms\n", n,
                build_ms, all_ms - build_ms);
    }
    s->info.lds_resident = s->geom.lds_scene ? 1 : 0;
    *out = s;
    return RT_OK;
}

int rt_scene_read_nodes(rt_scene *s, void *out, uint64_t bytes) {
    if (!s || !out) return fail(RT_ERR_ARG, "null argument");
    if (bytes != (uint64_t)s->info.n_nodes * 8 * sizeof(f4)) return fail(RT_ERR_ARG, "expected n_nodes * 128 bytes");
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    if (bytes) RT_HIP(hipMemcpy(out, s->dev.nodes, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scene_get_info(const rt_scene *s, rt_scene_info *out) {
    if (!s || !out) return fail(RT_ERR_ARG, "null argument");
    *out = s->info;
    return RT_OK;
}

} // extern "C"

namespace {

// Most bytes of per-path records one launch may write (a call with more samples is split into
// several launches).  Large launches amortise their tail: 2 GB instead of 512 MB is worth 7 % on a
// 4K image of a scene read from HBM and 1.5-2.5 % on small images at thousands of samples per call;
// beyond 2 GB nothing more is gained.  A scene holds up to three such buffers.
size_t stage_budget_bytes() {
    const char *e = getenv("RT_AMD_STAGE_MB");
    size_t mb = e ? (size_t)atoll(e) : 2048;
    return std::max<size_t>(mb, 1) << 20;
}

// Path-state arrays and queues of the wavefront variant for `n_slots` paths in flight.
int ensure_wavefront(rt_scene *s, unsigned n_slots) {
    if (s->wf_mem && s->wf_slots >= n_slots) return RT_OK;
    if (s->wf_mem) RT_HIP(hipFree(s->wf_mem));
    s->wf_mem = nullptr;
    s->wf_slots = 0;
    const size_t n = n_slots;
    const size_t bytes = n * RT_WF_BYTES_PER_SLOT + 256;
    RT_HIP(hipMalloc(&s->wf_mem, bytes));
    char *p = (char *)s->wf_mem;
    auto take = [&](size_t b) {
        char *q = p;
        p += b;
        return (void *)q;
    };
    WfBuffers &w = s->wf;
    w.state = (f4 *)take(n * 64);
    w.aux = (rtc::i4 *)take(n * 16);
    w.key = (int *)take(n * 4);
    for (int k = 0; k < 2; k++) w.q_trace[k] = (int *)take(n * 4);
    for (int k = 0; k < 5; k++) w.q_mat[k] = (int *)take(n * 4);
    w.q_free = (int *)take(n * 4);
    w.cnt = (unsigned *)take(64);
    if (!s->wf_host_cnt) RT_HIP(hipHostMalloc((void **)&s->wf_host_cnt, 2 * 16 * sizeof(unsigned), hipHostMallocDefault));
    s->wf_slots = n_slots;
    return RT_OK;
}

// Knobs of the render kernels' scheduling, read from the environment once per call.
struct Tunables {
    int scheduler, pipeline, sets, unit_order, strip_order;
    int min_ready, leaf_min, blocked_min, regen_min, shade_min;
    bool timeline;
};
Tunables read_tunables(bool lds_scene) {
    Tunables t;
    t.scheduler = env_int("RT_AMD_SCHEDULER", 1) != 0 ? 1 : 0; // BVH scenes: 1 = decoupled (default), 0 = whole ray
    t.pipeline = env_int("RT_AMD_PIPELINE", 1);
    t.sets = std::min(kSets, std::max(2, env_int("RT_AMD_SETS", kSets))); // record sets that pipelined launches rotate through
    t.unit_order = env_int("RT_AMD_UNIT_ORDER", 1);
    t.strip_order = env_int("RT_AMD_STRIP_ORDER", lds_scene ? 1 : 0);
    t.min_ready = std::max(1, std::min(64, env_int("RT_AMD_MIN_READY", 48)));
    t.leaf_min = std::max(1, std::min(64, env_int("RT_AMD_LEAF_MIN", 8)));
    t.blocked_min = std::max(1, std::min(64, env_int("RT_AMD_BLOCKED_MIN", 2)));
    t.regen_min = std::max(1, std::min(64, env_int("RT_AMD_REGEN_MIN", 1)));
    t.shade_min = std::max(0, std::min(64, env_int("RT_AMD_SHADE_MIN", 0))); // 0: the scene's own (LaunchGeom)
    t.timeline = getenv("RT_AMD_TIMELINE") != nullptr;
    return t;
}

// The fields of RenderParams every launch has: image, sample range, units, scheduling knobs.
// Everything optional (strip order, counters, timeline) starts switched off.
RenderParams base_params(int W, int H, int n_pixels, const rt_render_opts *opts, int done, int sc_n,
                                const LaunchGeom &lg, const Tunables &tn) {
    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.width = W;
    rp.height = H;
    rp.n_pixels = n_pixels;
    rp.sample_begin = opts->sample_begin + done;
    rp.sample_count = sc_n;
    rp.seed = opts->seed;
    rp.n_strips = (unsigned)((n_pixels + 63) / 64);
    rp.n_units = rp.n_strips * (unsigned)sc_n;
    rp.pixel_major = tn.unit_order;
    rp.ring_f4 = lg.ring_f4;
    rp.coop_f4 = lg.coop_f4;
    rp.stack_lds_entries = lg.stack_lds_entries;
    rp.scheduler = tn.scheduler;
    rp.min_ready = tn.min_ready;
    rp.leaf_min = tn.leaf_min;
    rp.blocked_min = tn.blocked_min;
    rp.regen_min = tn.regen_min;
    rp.shade_min = tn.shade_min > 0 ? tn.shade_min : lg.shade_min;
    return rp;
}

// Checks shared by the render entry points; fills the device camera.
int check_render_args(rt_scene *s, const rt_camera_desc *cam, const rt_render_opts *opts, rtc::DevCamera &dc,
                             int &n_pixels) {
    if (opts->sample_count < 0 || opts->sample_begin < 0) return fail(RT_ERR_ARG, "negative sample range");
    if ((long long)opts->sample_begin + opts->sample_count > 0x7fffffffll) return fail(RT_ERR_ARG, "sample range overflows");
    if (cam->max_depth < 0) return fail(RT_ERR_ARG, "negative max_depth");
    // paths handed from wave to wave and the wavefront variant keep the bounces left in 16 bits
    if (cam->max_depth > 65535) return fail(RT_ERR_ARG, "max_depth %d: at most 65535", cam->max_depth);
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    rc = rt_camera_initialize(cam, &dc.f);
    if (rc != RT_OK) return rc;
    for (int k = 0; k < 3; k++) dc.background[k] = cam->background[k];
    dc.defocus_angle = cam->defocus_angle;
    dc.max_depth = cam->max_depth;
    const long long n_pixels_ll = (long long)dc.f.image_width * dc.f.image_height;
    if (n_pixels_ll > (1ll << 30)) return fail(RT_ERR_ARG, "image too large");
    n_pixels = (int)n_pixels_ll;
    return RT_OK;
}

struct EventPair { // destroyed on every way out of the function that made them
    hipEvent_t a = nullptr, b = nullptr;
    int create() {
        RT_HIP(hipEventCreate(&a));
        RT_HIP(hipEventCreate(&b));
        return RT_OK;
    }
    void release() { a = b = nullptr; }
    ~EventPair() {
        if (a) hipEventDestroy(a);
        if (b) hipEventDestroy(b);
    }
};

} // namespace

extern "C" {

static int impl_rt_render_device(rt_scene *s, const rt_camera_desc *cam, const rt_render_opts *opts, float *d_accum,
                     void *stream_v, rt_render_stats *stats) {
    if (!s || !cam || !opts || !d_accum) return fail(RT_ERR_ARG, "null argument");
    if (opts->mode != RT_MODE_MEGAKERNEL && opts->mode != RT_MODE_WAVEFRONT)
        return fail(RT_ERR_UNSUPPORTED, "render mode %d not available", opts->mode);
    const bool wavefront = opts->mode == RT_MODE_WAVEFRONT;
    if (wavefront && (opts->flags & RT_RENDER_COUNTERS))
        return fail(RT_ERR_UNSUPPORTED, "the wavefront variant has no instrumented build");
    rtc::DevCamera dc;
    int n_pixels = 0;
    int rc = check_render_args(s, cam, opts, dc, n_pixels);
    if (rc != RT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    const int W = dc.f.image_width, H = dc.f.image_height;

    if (stats) {
        memset(stats, 0, sizeof(*stats));
    }
    if (opts->sample_count == 0) return RT_OK;

    // samples per launch: the per-path record buffer is bounded
    size_t per_sample = (size_t)n_pixels * sizeof(f4);
    int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)opts->sample_count, stage_budget_bytes() / per_sample));
    {   // launches of equal size: a short last launch pays the same tail as a long one
        const int n_launches = (opts->sample_count + chunk - 1) / chunk;
        chunk = (opts->sample_count + n_launches - 1) / n_launches;
    }
    size_t need = per_sample * chunk;
    auto ensure_stage = [&](int b) -> int {
        if (need > s->stage_bytes[b]) {
            if (s->stage[b]) RT_HIP(hipFree(s->stage[b])); // (synchronises the device)
            s->stage[b] = nullptr;
            s->stage_bytes[b] = 0;
            RT_HIP(hipMalloc((void **)&s->stage[b], need));
            s->stage_bytes[b] = need;
        }
        return RT_OK;
    };

    const bool count = (opts->flags & RT_RENDER_COUNTERS) != 0 && stats != nullptr;
    LaunchGeom lg = s->geom;
    const Tunables tn = read_tunables(lg.lds_scene);
    const int scheduler = tn.scheduler;
    rtc::DevScene dev = s->dev;
    dev.lds_records = lg.lds_records;
    dev.lds_perlins = lg.lds_perlins;
    unsigned wf_slots = 0;
    int wf_rounds = 0;
    if (wavefront) {
        // paths in flight: enough to fill the GPU many times over, bounded so that the state
        // (about 110 bytes per slot) stays a small part of HBM
        size_t want = std::min<size_t>((size_t)n_pixels * chunk, (size_t)std::max(1, env_int("RT_AMD_WF_SLOTS", 1 << 21)));
        wf_slots = (unsigned)((want + 255) / 256 * 256);
        rc = ensure_wavefront(s, wf_slots);
        if (rc != RT_OK) return rc;
    }
    EventPair ev; // brackets each launch on the caller's stream when statistics are wanted
    if (stats) {
        rc = ev.create();
        if (rc != RT_OK) return rc;
        RT_HIP(hipMemsetAsync(s->d_words + 1, 0, sizeof(unsigned long long), stream));
        if (count) {
            rtc::Counters zero;
            memset(&zero, 0, sizeof(zero));
            zero.tail[0] = zero.tail[1] = ~0ull; // minima
            RT_HIP(hipMemcpyAsync(s->d_counters, &zero, sizeof(zero), hipMemcpyHostToDevice, stream));
            RT_HIP(hipStreamSynchronize(stream)); // `zero` lives on this stack frame
        }
    }
    // pipelined launches: megakernel only, and not when the caller wants per-call statistics
    const bool pipelined = !wavefront && !stats && !(opts->flags & RT_RENDER_NO_OVERLAP) && tn.pipeline != 0;
    // the 8-waves-per-SIMD scan kernel: two pipelined launches share a CU with a workgroup each;
    // launches that run one at a time start two workgroups per CU themselves
    if (lg.dense && !pipelined && !count && !wavefront) lg.grid *= 2;
    float mega_ms = 0.0f;
    for (int done = 0; done < opts->sample_count; done += chunk) {
        int sc_n = std::min(chunk, opts->sample_count - done);
        const int b = pipelined ? s->flip : 0;
        if (pipelined) s->flip = (s->flip + 1) % tn.sets;
        rc = ensure_stage(b);
        if (rc != RT_OK) return rc;
        // every launch leaves an event behind that says "this record set has been reduced"
        if (!s->ev_free[b]) RT_HIP(hipEventCreateWithFlags(&s->ev_free[b], hipEventDisableTiming));
        hipStream_t ms = stream; // the stream the render kernel goes to
        if (pipelined) {
            if (!s->mega_stream[b]) {
                RT_HIP(hipStreamCreateWithFlags(&s->mega_stream[b], hipStreamNonBlocking));
                RT_HIP(hipEventCreateWithFlags(&s->ev_done[b], hipEventDisableTiming));
            }
            ms = s->mega_stream[b];
            // this set's previous records must have been reduced (by a pipelined call or by one
            // that kept to its caller's stream); nothing else orders the render kernel (it reads
            // only the scene and writes only this set)
            if (s->ev_free_valid[b]) RT_HIP(hipStreamWaitEvent(ms, s->ev_free[b], 0));
        } else {
            // set 0 on the caller's stream: whatever used any set last must be through with it
            // (the others too: the strip costs and the ray counter are shared with them)
            for (int k = 0; k < kSets; k++)
                if (s->ev_free_valid[k]) RT_HIP(hipStreamWaitEvent(stream, s->ev_free[k], 0));
        }
        RenderParams rp = base_params(W, H, n_pixels, opts, done, sc_n, lg, tn);
        // Strips in the order of their cost in this set's previous launch (megakernel, scenes in
        // LDS: for scenes read from HBM neighbouring rays matter more than the tail of the launch)
        const bool strip_costs = !wavefront && rp.pixel_major && tn.strip_order != 0;
        if (strip_costs) {
            if (rp.n_strips > s->strip_cap) {
                RT_HIP(hipDeviceSynchronize());
                for (int k = 0; k < kSets; k++) {
                    if (s->strip_cost[k]) RT_HIP(hipFree(s->strip_cost[k]));
                    if (s->strip_order[k]) RT_HIP(hipFree(s->strip_order[k]));
                    s->strip_cost[k] = s->strip_order[k] = nullptr;
                    RT_HIP(hipMalloc((void **)&s->strip_cost[k], rp.n_strips * sizeof(unsigned)));
                    RT_HIP(hipMalloc((void **)&s->strip_order[k], rp.n_strips * sizeof(unsigned)));
                    s->strip_valid[k] = 0;
                }
                if (s->strip_iota) RT_HIP(hipFree(s->strip_iota));
                if (s->strip_sorted) RT_HIP(hipFree(s->strip_sorted));
                if (s->strip_tmp) RT_HIP(hipFree(s->strip_tmp));
                s->strip_iota = s->strip_sorted = nullptr;
                s->strip_tmp = nullptr;
                RT_HIP(hipMalloc((void **)&s->strip_iota, rp.n_strips * sizeof(unsigned)));
                RT_HIP(hipMalloc((void **)&s->strip_sorted, rp.n_strips * sizeof(unsigned)));
                s->strip_tmp_bytes = std::max<size_t>(rtl::order_strips_tmp_bytes(rp.n_strips), 16);
                RT_HIP(hipMalloc(&s->strip_tmp, s->strip_tmp_bytes));
                RT_HIP(rtl::fill_iota(s->strip_iota, rp.n_strips, nullptr));
                RT_HIP(hipDeviceSynchronize());
                s->strip_cap = rp.n_strips;
            }
            if (s->strip_valid[b] == rp.n_strips) rp.strip_order = s->strip_order[b];
        }
        rp.stage = s->stage[b];
        rp.stack_spill = s->stack_spill[b];
        rp.next_chunk = reinterpret_cast<unsigned *>(s->d_words + kPoolWord[b]);
        rp.counters = count ? s->d_counters : nullptr;
        if (!wavefront && scheduler == 1 && s->dev.traversal == rtc::TRAV_BVH && tn.timeline) {
            const int waves = lg.grid * (lg.block / 64);
            if (!s->d_timeline) {
                RT_HIP(hipMalloc((void **)&s->d_timeline, (size_t)kTimelineLaunches * waves * 32));
                RT_HIP(hipMemset(s->d_timeline, 0, (size_t)kTimelineLaunches * waves * 32));
                s->timeline_waves = waves;
                s->timeline_block_waves = lg.block / 64;
            }
            if (waves == s->timeline_waves)
                rp.timeline = s->d_timeline + (size_t)(s->timeline_launches++ % kTimelineLaunches) * waves * 4;
        }
        RT_HIP(hipMemsetAsync(s->d_words + kPoolWord[b], 0, sizeof(unsigned long long), ms));
        EventPair prof;
        if (s->profiling) {
            rc = prof.create();
            if (rc != RT_OK) return rc;
            RT_HIP(hipEventRecord(prof.a, ms));
        }
        if (stats) RT_HIP(hipEventRecord(ev.a, stream));
        if (!wavefront) {
            if (dev.n_inst > 0)
                RT_HIP(rtl::launch_mega_inst(dev, dc, rp, lg, ms));
            else if (s->lean)
                RT_HIP(rtl::launch_mega_lean(dev, dc, rp, lg, ms));
            else
                RT_HIP(rtl::launch_mega(dev, dc, rp, lg, ms));
        } else if (dc.max_depth <= 0) { // camera.h:141: every path is black
            RT_HIP(hipMemsetAsync(s->stage[b], 0, per_sample * sc_n, stream));
        } else {
            int rounds = 0;
            if (dev.n_inst > 0)
                RT_HIP(rtl::render_wavefront_inst(dev, dc, rp, s->wf, wf_slots, lg, s->wf_host_cnt, stream, &rounds));
            else
                RT_HIP(rtl::render_wavefront(dev, dc, rp, s->wf, wf_slots, lg, s->wf_host_cnt, stream, &rounds));
            wf_rounds += rounds;
        }
        if (stats) RT_HIP(hipEventRecord(ev.b, stream));
        if (s->profiling) {
            RT_HIP(hipEventRecord(prof.b, ms));
            s->prof_events.push_back(prof.a);
            s->prof_events.push_back(prof.b);
            prof.release(); // rt_profile_end destroys them
        }
        if (pipelined) { // the caller's stream takes over once the records are complete
            RT_HIP(hipEventRecord(s->ev_done[b], ms));
            RT_HIP(hipStreamWaitEvent(stream, s->ev_done[b], 0));
        }
        RT_HIP(rtl::launch_reduce(s->stage[b], n_pixels, sc_n, d_accum, s->d_words + 1,
                                  strip_costs ? s->strip_cost[b] : nullptr, stream));
        if (strip_costs) { // the set's next launch waits for ev_free
            RT_HIP(rtl::order_strips(s->strip_cost[b], s->strip_iota, rp.n_strips, s->strip_sorted, s->strip_order[b],
                                     s->strip_tmp, s->strip_tmp_bytes, stream));
            s->strip_valid[b] = rp.n_strips;
        }
        // whoever uses this set next (on whatever stream) waits for this
        RT_HIP(hipEventRecord(s->ev_free[b], stream));
        s->ev_free_valid[b] = true;
        if (stats) {
            RT_HIP(hipEventSynchronize(ev.b));
            float ms_f = 0.0f;
            RT_HIP(hipEventElapsedTime(&ms_f, ev.a, ev.b));
            mega_ms += ms_f;
        }
    }
    if (stats) {
        RT_HIP(hipStreamSynchronize(stream));
        unsigned long long rays = 0;
        RT_HIP(hipMemcpy(&rays, s->d_words + 1, sizeof(rays), hipMemcpyDeviceToHost));
        stats->paths = (uint64_t)n_pixels * (uint64_t)opts->sample_count;
        stats->rays = rays;
        stats->kernel_ms = mega_ms;
        if (wavefront) stats->sched[0] = (uint64_t)wf_rounds;
        if (count) {
            rtc::Counters c;
            RT_HIP(hipMemcpy(&c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
            stats->box_tests = c.box_tests;
            stats->sphere_tests = c.sphere_tests;
            stats->sphere_hits = c.sphere_hits;
            stats->quad_tests = c.quad_tests;
            for (int k = 0; k < 4; k++) stats->shades[k] = c.shades[k];
            stats->noise_evals = c.noise_evals;
            stats->image_evals = c.image_evals;
            stats->checker_evals = c.checker_evals;
            for (int k = 0; k < 6; k++) stats->sched[k] = c.sched[k];
            for (int k = 0; k < 6; k++) stats->cycles[k] = c.cycles[k];
            for (int k = 0; k < 2; k++) stats->coherence[k] = c.coherence[k];
            if (c.tail[3] > c.tail[0] && c.tail[0] != ~0ull)
                for (int k = 0; k < 3; k++) // wall_clock64 ticks at 100 MHz
                    stats->tail_us[k] = c.tail[k + 1] == ~0ull ? 0.0 : (double)(c.tail[k + 1] - c.tail[0]) * 0.01;
        }
    }
    return RT_OK;
}

// rt_render_device for callers that cannot allocate device memory (code under HIP graph
// capture, serving loops, harnesses that forbid it): the launch's counters and per-path records
// live in the library's statically allocated block (rt_static.hip) instead of buffers owned by
// the scene, everything is enqueued on the caller's stream, and nothing here calls hipMalloc /
// hipFree or creates streams or events.  Megakernel only, no statistics.
static int impl_rt_render_device_static(rt_scene *s, const rt_camera_desc *cam, const rt_render_opts *opts, float *d_accum,
                            void *stream_v) {
    if (!s || !cam || !opts || !d_accum) return fail(RT_ERR_ARG, "null argument");
    if (opts->mode != RT_MODE_MEGAKERNEL)
        return fail(RT_ERR_UNSUPPORTED, "rt_render_device_static: megakernel mode only");
    if (opts->flags & RT_RENDER_COUNTERS) return fail(RT_ERR_UNSUPPORTED, "rt_render_device_static keeps no statistics");
    rtc::DevCamera dc;
    int n_pixels = 0;
    int rc = check_render_args(s, cam, opts, dc, n_pixels);
    if (rc != RT_OK) return rc;
    unsigned char *ptr = (unsigned char *)rtl::static_scratch(s->device); // looked up by rt_scene_create
    if (!ptr) return fail(RT_ERR_HIP, "the static scratch block of device %d was not set up", s->device);
    if ((unsigned)n_pixels > RT_STATIC_MAX_RECORDS)
        return fail(RT_ERR_ARG, "rt_render_device_static: an image of %d pixels exceeds the static scratch block "
                                "(at most %u pixels); use rt_render_device", n_pixels, RT_STATIC_MAX_RECORDS);
    if (opts->sample_count == 0) return RT_OK;
    hipStream_t stream = (hipStream_t)stream_v;
    const int W = dc.f.image_width, H = dc.f.image_height;
    unsigned long long *words = reinterpret_cast<unsigned long long *>(ptr); // 0: pool counter, 1: rays
    f4 *stage = reinterpret_cast<f4 *>(ptr + RT_STATIC_HEADER_BYTES);

    int chunk = (int)std::min<size_t>((size_t)opts->sample_count, RT_STATIC_MAX_RECORDS / (size_t)n_pixels);
    {   // launches of equal size, as rt_render_device does
        const int n_launches = (opts->sample_count + chunk - 1) / chunk;
        chunk = (opts->sample_count + n_launches - 1) / n_launches;
    }
    LaunchGeom lg = s->geom;
    if (lg.dense) lg.grid *= 2; // (launches of this entry point run one at a time)
    const Tunables tn = read_tunables(lg.lds_scene);
    rtc::DevScene dev = s->dev;
    dev.lds_records = lg.lds_records;
    dev.lds_perlins = lg.lds_perlins;
    for (int done = 0; done < opts->sample_count; done += chunk) {
        const int sc_n = std::min(chunk, opts->sample_count - done);
        RenderParams rp = base_params(W, H, n_pixels, opts, done, sc_n, lg, tn);
        rp.stage = stage;
        rp.stack_spill = s->stack_spill[kSets]; // (allocated with the scene, not here)
        rp.next_chunk = reinterpret_cast<unsigned *>(words);
        // the block keeps the last call's contents: the counters must start at zero
        RT_HIP(hipMemsetAsync(words, 0, 2 * sizeof(unsigned long long), stream));
        if (dev.n_inst > 0)
            RT_HIP(rtl::launch_mega_inst(dev, dc, rp, lg, stream));
        else if (s->lean)
            RT_HIP(rtl::launch_mega_lean(dev, dc, rp, lg, stream));
        else
            RT_HIP(rtl::launch_mega(dev, dc, rp, lg, stream));
        RT_HIP(rtl::launch_reduce(stage, n_pixels, sc_n, d_accum, words + 1, nullptr, stream));
    }
    return RT_OK;
}

int rt_render(rt_scene *s, const rt_camera_desc *cam, const rt_render_opts *opts, float *accum_host,
              rt_render_stats *stats) {
    if (!s || !cam || !opts || !accum_host) return fail(RT_ERR_ARG, "null argument");
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    rt_camera_frame f;
    rc = rt_camera_initialize(cam, &f);
    if (rc != RT_OK) return rc;
    size_t bytes = (size_t)f.image_width * f.image_height * 3 * sizeof(float);
    float *d_accum = nullptr;
    RT_HIP(hipMalloc((void **)&d_accum, bytes));
    hipError_t e = hipMemcpy(d_accum, accum_host, bytes, hipMemcpyHostToDevice);
    rt_render_stats local;
    if (e == hipSuccess) {
        rc = rt_render_device(s, cam, opts, d_accum, nullptr, stats ? stats : &local);
        if (rc == RT_OK) e = hipMemcpy(accum_host, d_accum, bytes, hipMemcpyDeviceToHost);
    }
    hipFree(d_accum);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, "accumulation buffer copy: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_trace_batch_device(rt_scene *s, const rt_ray *d_rays, int64_t n, float tmin, float tmax, rt_hit *d_hits,
                          void *stream) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_hits))) return fail(RT_ERR_ARG, "bad argument");
    if (n == 0) return RT_OK;
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    rtc::DevScene dev = s->dev;
    dev.ref_bits = 32; // whole keys on 32-bit stacks would do; the parity kernel keeps plain references
    if (dev.n_inst > 0)
        RT_HIP(rtl::launch_trace_inst(dev, d_rays, n, tmin, tmax, d_hits, (hipStream_t)stream));
    else
        RT_HIP(rtl::launch_trace(dev, d_rays, n, tmin, tmax, d_hits, (hipStream_t)stream));
    return RT_OK;
}

// Scenes that are scanned have one closest-hit routine (trace_linear) in every kernel: the
// scheduled variant only exists for trees.
int rt_trace_batch_sched_device(rt_scene *s, const rt_ray *d_rays, int64_t n, float tmin, float tmax, rt_hit *d_hits,
                                void *stream_v) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_hits))) return fail(RT_ERR_ARG, "bad argument");
    if (n == 0) return RT_OK;
    if (s->dev.traversal != rtc::TRAV_BVH) return rt_trace_batch_device(s, d_rays, n, tmin, tmax, d_hits, stream_v);
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    // record set 0's stack overflow area and a spare counter word: order behind whatever used them
    for (int k = 0; k < kSets; k++)
        if (s->ev_free_valid[k]) RT_HIP(hipStreamWaitEvent(stream, s->ev_free[k], 0));
    const LaunchGeom lg = s->geom;
    const Tunables tn = read_tunables(lg.lds_scene);
    rt_render_opts o;
    memset(&o, 0, sizeof(o));
    RenderParams rp = base_params(1, 1, 1, &o, 0, 1, lg, tn);
    rp.stack_spill = s->stack_spill[0];
    rp.next_chunk = reinterpret_cast<unsigned *>(s->d_words + 6);
    rtc::DevScene dev = s->dev;
    dev.lds_records = lg.lds_records;
    dev.lds_perlins = lg.lds_perlins;
    RT_HIP(hipMemsetAsync(s->d_words + 6, 0, sizeof(unsigned long long), stream));
    if (dev.n_inst > 0)
        RT_HIP(rtl::launch_trace_sched_inst(dev, rp, lg, d_rays, n, tmin, tmax, d_hits, stream));
    else
        RT_HIP(rtl::launch_trace_sched(dev, rp, lg, d_rays, n, tmin, tmax, d_hits, stream));
    if (!s->ev_free[0]) RT_HIP(hipEventCreateWithFlags(&s->ev_free[0], hipEventDisableTiming));
    RT_HIP(hipEventRecord(s->ev_free[0], stream));
    s->ev_free_valid[0] = true;
    return RT_OK;
}

int rt_closest_hits_device(rt_scene *s, const rt_ray *d_rays, int64_t n, float tmin, float tmax, int32_t *d_prim,
                           float *d_t, void *stream_v) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_prim || !d_t))) return fail(RT_ERR_ARG, "bad argument");
    if (n == 0) return RT_OK;
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_v;
    const bool tree = s->dev.traversal == rtc::TRAV_BVH;
    // (as rt_trace_batch_sched_device: record set 0's stack overflow area and a spare counter word)
    if (tree)
        for (int k = 0; k < kSets; k++)
            if (s->ev_free_valid[k]) RT_HIP(hipStreamWaitEvent(stream, s->ev_free[k], 0));
    const LaunchGeom lg = s->geom;
    const Tunables tn = read_tunables(lg.lds_scene);
    rt_render_opts o;
    memset(&o, 0, sizeof(o));
    RenderParams rp = base_params(1, 1, 1, &o, 0, 1, lg, tn);
    rp.stack_spill = s->stack_spill[0];
    rp.next_chunk = reinterpret_cast<unsigned *>(s->d_words + 6);
    rtc::DevScene dev = s->dev;
    dev.lds_records = lg.lds_records;
    dev.lds_perlins = lg.lds_perlins;
    if (tree) RT_HIP(hipMemsetAsync(s->d_words + 6, 0, sizeof(unsigned long long), stream));
    if (dev.n_inst > 0)
        RT_HIP(rtl::launch_closest_inst(dev, rp, lg, d_rays, n, tmin, tmax, d_prim, d_t, stream));
    else
        RT_HIP(rtl::launch_closest(dev, rp, lg, d_rays, n, tmin, tmax, d_prim, d_t, stream));
    if (tree) {
        if (!s->ev_free[0]) RT_HIP(hipEventCreateWithFlags(&s->ev_free[0], hipEventDisableTiming));
        RT_HIP(hipEventRecord(s->ev_free[0], stream));
        s->ev_free_valid[0] = true;
    }
    return RT_OK;
}

int rt_closest_hits(rt_scene *s, const rt_ray *rays, int64_t n, float tmin, float tmax, int32_t *prim, float *t) {
    if (!s || n < 0 || (n > 0 && (!rays || !prim || !t))) return fail(RT_ERR_ARG, "bad argument");
    if (n == 0) return RT_OK;
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    rt_ray *d_rays = nullptr;
    int32_t *d_prim = nullptr;
    float *d_t = nullptr;
    hipError_t e = hipMalloc((void **)&d_rays, sizeof(rt_ray) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_prim, sizeof(int32_t) * n);
    if (e == hipSuccess) e = hipMalloc((void **)&d_t, sizeof(float) * n);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, sizeof(rt_ray) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = rt_closest_hits_device(s, d_rays, n, tmin, tmax, d_prim, d_t, nullptr);
        if (rc == RT_OK) e = hipDeviceSynchronize();
        if (rc == RT_OK && e == hipSuccess) e = hipMemcpy(prim, d_prim, sizeof(int32_t) * n, hipMemcpyDeviceToHost);
        if (rc == RT_OK && e == hipSuccess) e = hipMemcpy(t, d_t, sizeof(float) * n, hipMemcpyDeviceToHost);
    }
    if (d_rays) hipFree(d_rays);
    if (d_prim) hipFree(d_prim);
    if (d_t) hipFree(d_t);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, "closest hits: %s", hipGetErrorString(e));
    return RT_OK;
}

namespace {
int trace_batch_host(rt_scene *s, const rt_ray *rays, int64_t n, float tmin, float tmax, rt_hit *hits, bool sched) {
    if (!s || n < 0 || (n > 0 && (!rays || !hits))) return fail(RT_ERR_ARG, "bad argument");
    if (n == 0) return RT_OK;
    int rc = set_device(s->device);
    if (rc != RT_OK) return rc;
    rt_ray *d_rays = nullptr;
    rt_hit *d_hits = nullptr;
    RT_HIP(hipMalloc((void **)&d_rays, sizeof(rt_ray) * n));
    hipError_t e = hipMalloc((void **)&d_hits, sizeof(rt_hit) * n);
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, sizeof(rt_ray) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = sched ? rt_trace_batch_sched_device(s, d_rays, n, tmin, tmax, d_hits, nullptr)
                   : rt_trace_batch_device(s, d_rays, n, tmin, tmax, d_hits, nullptr);
        if (rc == RT_OK) e = hipDeviceSynchronize();
        if (rc == RT_OK && e == hipSuccess) e = hipMemcpy(hits, d_hits, sizeof(rt_hit) * n, hipMemcpyDeviceToHost);
    }
    hipFree(d_rays);
    if (d_hits) hipFree(d_hits);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, "trace batch: %s", hipGetErrorString(e));
    return RT_OK;
}
} // namespace

int rt_trace_batch(rt_scene *s, const rt_ray *rays, int64_t n, float tmin, float tmax, rt_hit *hits) {
    return trace_batch_host(s, rays, n, tmin, tmax, hits, false);
}
int rt_trace_batch_sched(rt_scene *s, const rt_ray *rays, int64_t n, float tmin, float tmax, rt_hit *hits) {
    return trace_batch_host(s, rays, n, tmin, tmax, hits, true);
}

// write_color (color.h:17-32): sqrt gamma, clamp to [0, 0.999], 256 * c truncated.
int rt_tonemap_u8(const float *accum, int64_t n_pixels, float scale, uint8_t *rgb8) {
    if (n_pixels < 0 || (n_pixels > 0 && (!accum || !rgb8))) return fail(RT_ERR_ARG, "bad argument");
    for (int64_t i = 0; i < n_pixels * 3; i++) {
        float c = scale * accum[i];
        float g = c > 0.0f ? sqrtf(c) : 0.0f;
        g = g < 0.000f ? 0.000f : (g > 0.999f ? 0.999f : g);
        rgb8[i] = (uint8_t) int(256 * g);
    }
    return RT_OK;
}

int rt_tonemap_u8_device(int device, const float *d_accum, int64_t n_pixels, float scale, uint8_t *d_rgb8,
                         void *stream) {
    if (n_pixels < 0 || n_pixels > (1ll << 36) || (n_pixels > 0 && (!d_accum || !d_rgb8)))
        return fail(RT_ERR_ARG, "bad argument");
    int rc = set_device(device);
    if (rc != RT_OK) return rc;
    RT_HIP(rtl::launch_tonemap(d_accum, n_pixels * 3, scale, d_rgb8, (hipStream_t)stream));
    return RT_OK;
}

// ---------------------------------------------------------------------------------------
// Film: accumulation image resident in HBM, rendered in passes, with asynchronous snapshots.
struct rt_film {
    int device = 0, width = 0, height = 0;
    int64_t samples = 0;      // samples per pixel accumulated (enqueued) so far
    int64_t snap_samples = 0; // ... at the last snapshot
    float *d_accum = nullptr;
    uint8_t *d_rgb = nullptr;
    uint8_t *h_rgb = nullptr; // pinned
    float *h_accum = nullptr; // pinned, allocated by the first snapshot that wants the sums
    hipStream_t stream = nullptr, copy_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_copied = nullptr;
    bool snap_pending = false, snap_has_accum = false;
    size_t n_values() const { return (size_t)width * height * 3; }
};

int rt_film_destroy(rt_film *f) {
    if (!f) return RT_OK;
    hipSetDevice(f->device);
    if (f->stream) hipStreamSynchronize(f->stream);
    if (f->copy_stream) hipStreamSynchronize(f->copy_stream);
    if (f->d_accum) hipFree(f->d_accum);
    if (f->d_rgb) hipFree(f->d_rgb);
    if (f->h_rgb) hipHostFree(f->h_rgb);
    if (f->h_accum) hipHostFree(f->h_accum);
    if (f->ev_ready) hipEventDestroy(f->ev_ready);
    if (f->ev_copied) hipEventDestroy(f->ev_copied);
    if (f->stream) hipStreamDestroy(f->stream);
    if (f->copy_stream) hipStreamDestroy(f->copy_stream);
    delete f;
    return RT_OK;
}

namespace {
int film_alloc(rt_film *f) {
    RT_HIP(hipMalloc((void **)&f->d_accum, f->n_values() * sizeof(float)));
    RT_HIP(hipMalloc((void **)&f->d_rgb, f->n_values()));
    RT_HIP(hipHostMalloc((void **)&f->h_rgb, f->n_values(), hipHostMallocDefault));
    RT_HIP(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    RT_HIP(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
    RT_HIP(hipEventCreateWithFlags(&f->ev_ready, hipEventDisableTiming));
    RT_HIP(hipEventCreateWithFlags(&f->ev_copied, hipEventDisableTiming));
    RT_HIP(hipMemsetAsync(f->d_accum, 0, f->n_values() * sizeof(float), f->stream));
    return RT_OK;
}
} // namespace

int rt_film_create(int device, int32_t width, int32_t height, rt_film **out) {
    if (!out) return fail(RT_ERR_ARG, "out is null");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 30))
        return fail(RT_ERR_ARG, "film size %d x %d", width, height);
    int n_dev = rt_device_count();
    if (n_dev <= 0) return fail(RT_ERR_HIP, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n_dev) return fail(RT_ERR_ARG, "device %d out of range (%d visible)", device, n_dev);
    int rc = set_device(device);
    if (rc != RT_OK) return rc;
    rt_film *f = new rt_film;
    f->device = device;
    f->width = width;
    f->height = height;
    rc = film_alloc(f);
    if (rc != RT_OK) {
        rt_film_destroy(f);
        return rc;
    }
    *out = f;
    return RT_OK;
}

int rt_film_render(rt_scene *s, rt_film *f, const rt_camera_desc *cam, const rt_render_opts *opts) {
    if (!s || !f || !cam || !opts) return fail(RT_ERR_ARG, "null argument");
    if (s->device != f->device) return fail(RT_ERR_ARG, "scene on device %d, film on device %d", s->device, f->device);
    rt_camera_frame fr;
    int rc = rt_camera_initialize(cam, &fr);
    if (rc != RT_OK) return rc;
    if (fr.image_width != f->width || fr.image_height != f->height)
        return fail(RT_ERR_ARG, "camera renders %d x %d, the film is %d x %d", fr.image_width, fr.image_height,
                    f->width, f->height);
    rc = rt_render_device(s, cam, opts, f->d_accum, f->stream, nullptr);
    if (rc != RT_OK) return rc;
    f->samples += opts->sample_count;
    return RT_OK;
}

int rt_film_snapshot(rt_film *f, int with_accum) {
    if (!f) return fail(RT_ERR_ARG, "null film");
    if (f->samples <= 0) return fail(RT_ERR_ARG, "the film holds no samples");
    int rc = set_device(f->device);
    if (rc != RT_OK) return rc;
    if (with_accum && !f->h_accum)
        RT_HIP(hipHostMalloc((void **)&f->h_accum, f->n_values() * sizeof(float), hipHostMallocDefault));
    // the previous snapshot's copies still read d_rgb
    if (f->snap_pending) RT_HIP(hipStreamWaitEvent(f->stream, f->ev_copied, 0));
    RT_HIP(rtl::launch_tonemap(f->d_accum, (long long)f->n_values(), 1.0f / (float)f->samples, f->d_rgb, f->stream));
    RT_HIP(hipEventRecord(f->ev_ready, f->stream));
    RT_HIP(hipStreamWaitEvent(f->copy_stream, f->ev_ready, 0));
    RT_HIP(hipMemcpyAsync(f->h_rgb, f->d_rgb, f->n_values(), hipMemcpyDeviceToHost, f->copy_stream));
    if (with_accum)
        RT_HIP(hipMemcpyAsync(f->h_accum, f->d_accum, f->n_values() * sizeof(float), hipMemcpyDeviceToHost, f->copy_stream));
    RT_HIP(hipEventRecord(f->ev_copied, f->copy_stream));
    // later passes add to d_accum on f->stream: they wait for the copy of the sums (their render
    // kernels run on the scene's internal streams and do not)
    if (with_accum) RT_HIP(hipStreamWaitEvent(f->stream, f->ev_copied, 0));
    f->snap_pending = true;
    f->snap_has_accum = with_accum != 0;
    f->snap_samples = f->samples;
    return RT_OK;
}

int rt_film_wait(rt_film *f, uint8_t *rgb8_out, float *accum_out, int64_t *samples_out) {
    if (!f) return fail(RT_ERR_ARG, "null film");
    if (!f->snap_pending) return fail(RT_ERR_ARG, "no snapshot was taken");
    if (accum_out && !f->snap_has_accum) return fail(RT_ERR_ARG, "the snapshot was taken without the sums");
    int rc = set_device(f->device);
    if (rc != RT_OK) return rc;
    RT_HIP(hipEventSynchronize(f->ev_copied));
    if (rgb8_out) memcpy(rgb8_out, f->h_rgb, f->n_values());
    if (accum_out) memcpy(accum_out, f->h_accum, f->n_values() * sizeof(float));
    if (samples_out) *samples_out = f->snap_samples;
    return RT_OK;
}

int rt_film_upload(rt_film *f, const float *accum_host, int64_t samples) {
    if (!f || samples < 0 || (!accum_host && samples != 0)) return fail(RT_ERR_ARG, "bad argument");
    int rc = set_device(f->device);
    if (rc != RT_OK) return rc;
    RT_HIP(hipStreamSynchronize(f->copy_stream));
    RT_HIP(hipStreamSynchronize(f->stream));
    if (accum_host)
        RT_HIP(hipMemcpy(f->d_accum, accum_host, f->n_values() * sizeof(float), hipMemcpyHostToDevice));
    else
        RT_HIP(hipMemset(f->d_accum, 0, f->n_values() * sizeof(float)));
    f->samples = samples;
    return RT_OK;
}

// Per-launch timing of the render kernel between begin and end (HIP events on the
// stream the launches go to).  end returns the summed time and the launch count.
int rt_profile_begin(rt_scene *s) {
    if (!s) return fail(RT_ERR_ARG, "null scene");
    for (hipEvent_t e : s->prof_events) hipEventDestroy(e);
    s->prof_events.clear();
    s->profiling = true;
    return RT_OK;
}
int rt_profile_end(rt_scene *s, double *total_ms, int *n_launches) {
    if (!s) return fail(RT_ERR_ARG, "null scene");
    s->profiling = false;
    double total = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < s->prof_events.size(); i += 2) {
        RT_HIP(hipEventSynchronize(s->prof_events[i + 1]));
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, s->prof_events[i], s->prof_events[i + 1]));
        total += ms;
        n++;
    }
    for (hipEvent_t e : s->prof_events) hipEventDestroy(e);
    s->prof_events.clear();
    if (total_ms) *total_ms = total;
    if (n_launches) *n_launches = n;
    return RT_OK;
}

int rt_scene_file_save(const rt_scene_desc *desc, const char *path) {
    if (!desc || !path) return fail(RT_ERR_ARG, "null argument");
    int rc = rtio::save(desc, path);
    if (rc != RT_OK) return fail(rc, "cannot write %s", path);
    return RT_OK;
}
int rt_scene_file_load(const char *path, rt_scene_desc **out) {
    if (!path || !out) return fail(RT_ERR_ARG, "null argument");
    *out = rtio::load(path);
    if (!*out) return fail(RT_ERR_IO, "cannot read %s (missing, truncated or inconsistent counts)", path);
    return RT_OK;
}
void rt_scene_file_free(rt_scene_desc *desc) { free(desc); }

} // extern "C"

// ---------------------------------------------------------------------------------------
// Several GPUs of one node behind one call: a replica of the scene per device, the sample range
// split into contiguous balanced shares (device k of n renders the k-th), one host thread per
// device for the enqueue, the per-device images gathered on the first device by peer copies
// (xGMI, one link per source device, in parallel) and added there in device order.
struct rt_multi {
    std::vector<int> devices;
    std::vector<rt_scene *> scenes;
    std::vector<float *> d_accum;   // per device, W*H*3 floats
    std::vector<float *> d_gather;  // on devices[0]: one landing buffer per other device
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> done;
    float *h_accum = nullptr; // pinned
    size_t cap_values = 0;
};

namespace {
// the split of spp_split.sample_range (raytracing_amd/spp_split.py)
void multi_share(int total, int k, int n, int *begin, int *count) {
    const int base = total / n, extra = total % n;
    *begin = k * base + std::min(k, extra);
    *count = base + (k < extra ? 1 : 0);
}

void multi_free_buffers(rt_multi *m) {
    for (size_t k = 0; k < m->devices.size(); k++) {
        hipSetDevice(m->devices[k]);
        if (k < m->d_accum.size() && m->d_accum[k]) hipFree(m->d_accum[k]);
    }
    if (!m->devices.empty()) hipSetDevice(m->devices[0]);
    for (float *p : m->d_gather)
        if (p) hipFree(p);
    if (m->h_accum) hipHostFree(m->h_accum);
    m->d_accum.assign(m->devices.size(), nullptr);
    m->d_gather.assign(m->devices.size(), nullptr);
    m->h_accum = nullptr;
    m->cap_values = 0;
}
} // namespace

extern "C" int rt_multi_destroy(rt_multi *m) {
    if (!m) return RT_OK;
    for (size_t k = 0; k < m->devices.size(); k++) {
        hipSetDevice(m->devices[k]);
        if (k < m->streams.size() && m->streams[k]) {
            hipStreamSynchronize(m->streams[k]);
            hipStreamDestroy(m->streams[k]);
        }
        if (k < m->done.size() && m->done[k]) hipEventDestroy(m->done[k]);
    }
    multi_free_buffers(m);
    for (rt_scene *s : m->scenes) rt_scene_destroy(s);
    delete m;
    return RT_OK;
}

static int impl_rt_multi_create(const rt_scene_desc *desc, const int32_t *devices, int32_t n_devices, uint32_t flags,
                                rt_multi **out) {
    if (!out) return fail(RT_ERR_ARG, "out is null");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return fail(RT_ERR_ARG, "between 1 and 64 devices");
    const int visible = rt_device_count();
    if (visible <= 0) return fail(RT_ERR_HIP, "no HIP device available (this library has no CPU path)");
    for (int k = 0; k < n_devices; k++)
        if (devices[k] < 0 || devices[k] >= visible)
            return fail(RT_ERR_ARG, "device %d out of range (%d visible)", devices[k], visible);
    rt_multi *m = new rt_multi();
    m->devices.assign(devices, devices + n_devices);
    m->d_accum.assign(n_devices, nullptr);
    m->d_gather.assign(n_devices, nullptr);
    m->streams.assign(n_devices, nullptr);
    m->done.assign(n_devices, nullptr);
    for (int k = 0; k < n_devices; k++) {
        rt_scene *s = nullptr;
        int rc = rt_scene_create(desc, devices[k], flags, &s); // (leaves that device current)
        if (rc == RT_OK && hipStreamCreateWithFlags(&m->streams[k], hipStreamNonBlocking) != hipSuccess)
            rc = fail(RT_ERR_HIP, "stream for device %d", devices[k]);
        if (rc == RT_OK && hipEventCreateWithFlags(&m->done[k], hipEventDisableTiming) != hipSuccess)
            rc = fail(RT_ERR_HIP, "event for device %d", devices[k]);
        if (s) m->scenes.push_back(s);
        if (rc != RT_OK) {
            std::string keep = g_err;
            rt_multi_destroy(m);
            g_err = keep;
            return rc;
        }
        if (k > 0 && devices[k] != devices[0]) { // direct copies over xGMI where the node allows them
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[0], devices[k]) == hipSuccess && can) {
                hipSetDevice(devices[0]);
                hipError_t e = hipDeviceEnablePeerAccess(devices[k], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    *out = m;
    return RT_OK;
}

static int impl_rt_multi_render(rt_multi *m, const rt_camera_desc *cam, const rt_render_opts *opts, float *accum_host,
                                rt_render_stats *stats) {
    if (!m || !cam || !opts || !accum_host) return fail(RT_ERR_ARG, "null argument");
    if (opts->sample_count < 0 || opts->sample_begin < 0) return fail(RT_ERR_ARG, "negative sample range");
    rt_camera_frame fr;
    int rc = rt_camera_initialize(cam, &fr);
    if (rc != RT_OK) return rc;
    const size_t n_values = (size_t)fr.image_width * fr.image_height * 3;
    const int n = (int)m->devices.size();
    if (stats) memset(stats, 0, sizeof(*stats));
    if (opts->sample_count == 0) return RT_OK;
    if (n_values > m->cap_values) {
        multi_free_buffers(m);
        for (int k = 0; k < n; k++) {
            rc = set_device(m->devices[k]);
            if (rc != RT_OK) return rc;
            RT_HIP(hipMalloc((void **)&m->d_accum[k], n_values * sizeof(float)));
        }
        rc = set_device(m->devices[0]);
        if (rc != RT_OK) return rc;
        for (int k = 1; k < n; k++) RT_HIP(hipMalloc((void **)&m->d_gather[k], n_values * sizeof(float)));
        RT_HIP(hipHostMalloc((void **)&m->h_accum, n_values * sizeof(float), hipHostMallocDefault));
        m->cap_values = n_values;
    }
    std::vector<int> rcs((size_t)n, RT_OK);
    std::vector<std::string> errs((size_t)n);
    std::vector<rt_render_stats> sts((size_t)n);
    auto work = [&](int k) {
        rcs[k] = guarded("rt_multi_render", [&]() -> int {
            int r = set_device(m->devices[k]);
            if (r != RT_OK) return r;
            rt_render_opts o = *opts;
            multi_share(opts->sample_count, k, n, &o.sample_begin, &o.sample_count);
            o.sample_begin += opts->sample_begin;
            RT_HIP(hipMemsetAsync(m->d_accum[k], 0, n_values * sizeof(float), m->streams[k]));
            memset(&sts[k], 0, sizeof(sts[k]));
            if (o.sample_count > 0) {
                r = rt_render_device(m->scenes[k], cam, &o, m->d_accum[k], m->streams[k], stats ? &sts[k] : nullptr);
                if (r != RT_OK) return r;
            }
            RT_HIP(hipEventRecord(m->done[k], m->streams[k]));
            return RT_OK;
        });
        if (rcs[k] != RT_OK) errs[k] = g_err; // (thread-local in the worker)
    };
    {
        std::vector<std::thread> pool;
        for (int k = 1; k < n; k++) pool.emplace_back(work, k);
        work(0);
        for (std::thread &t : pool) t.join();
    }
    for (int k = 0; k < n; k++)
        if (rcs[k] != RT_OK) return fail(rcs[k], "device %d: %s", m->devices[k], errs[k].c_str());
    // gather on the first device: every other image over its own link, added in device order
    rc = set_device(m->devices[0]);
    if (rc != RT_OK) return rc;
    hipStream_t s0 = m->streams[0];
    for (int k = 1; k < n; k++) {
        RT_HIP(hipStreamWaitEvent(s0, m->done[k], 0));
        RT_HIP(hipMemcpyPeerAsync(m->d_gather[k], m->devices[0], m->d_accum[k], m->devices[k], n_values * sizeof(float), s0));
    }
    for (int k = 1; k < n; k++) RT_HIP(rtl::launch_accum_add(m->d_accum[0], m->d_gather[k], (long long)n_values, s0));
    RT_HIP(hipMemcpyAsync(m->h_accum, m->d_accum[0], n_values * sizeof(float), hipMemcpyDeviceToHost, s0));
    RT_HIP(hipStreamSynchronize(s0));
    for (size_t i = 0; i < n_values; i++) accum_host[i] += m->h_accum[i];
    if (stats) {
        for (int k = 0; k < n; k++) {
            stats->paths += sts[k].paths;
            stats->rays += sts[k].rays;
            stats->kernel_ms = std::max(stats->kernel_ms, sts[k].kernel_ms);
        }
    }
    return RT_OK;
}

extern "C" int rt_multi_create(const rt_scene_desc *desc, const int32_t *devices, int32_t n_devices, uint32_t flags,
                               rt_multi **out) {
    return guarded("rt_multi_create", [&] { return impl_rt_multi_create(desc, devices, n_devices, flags, out); });
}
extern "C" int rt_multi_render(rt_multi *m, const rt_camera_desc *cam, const rt_render_opts *opts, float *accum_host,
                               rt_render_stats *stats) {
    return guarded("rt_multi_render", [&] { return impl_rt_multi_render(m, cam, opts, accum_host, stats); });
}

extern "C" int rt_scene_create(const rt_scene_desc *d, int device, uint32_t flags, rt_scene **out) {
    return guarded("rt_scene_create", [&] { return impl_rt_scene_create(d, device, flags, out); });
}

extern "C" int rt_render_device(rt_scene *s, const rt_camera_desc *cam, const rt_render_opts *opts, float *d_accum, void *stream_v, rt_render_stats *stats) {
    return guarded("rt_render_device", [&] { return impl_rt_render_device(s, cam, opts, d_accum, stream_v, stats); });
}

extern "C" int rt_render_device_static(rt_scene *s, const rt_camera_desc *cam, const rt_render_opts *opts, float *d_accum, void *stream_v) {
    return guarded("rt_render_device_static", [&] { return impl_rt_render_device_static(s, cam, opts, d_accum, stream_v); });
}
