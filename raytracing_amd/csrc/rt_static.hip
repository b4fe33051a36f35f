// rt_static.hip — the library's one statically allocated block of device memory.
//
// rt_render_device_static (rt_host.cpp) renders without calling hipMalloc: the per-launch
// scratch of the render kernels (the launch's pool / ray counters and the per-path radiance
// records that reduce_stage sums into the caller's image) lives in the file-scope array below
// instead of in buffers owned by the scene.  Every GPU the library runs on holds its own copy
// of the array, at its own address, until the process ends; the address is looked up once per
// device (static_scratch_setup, called by rt_scene_create with that device current) and from
// then on host code passes only that pointer around, never the array's name: in host code the
// name is a stand-in, not a device address.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rt_launch.h"

static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

namespace rtl {

namespace {
const int kMaxDevices = 64;
std::mutex g_lock;
void *g_scratch_ptr[kMaxDevices]; // per device: the address of that GPU's copy of `scratch`
} // namespace

// `device` must be the current device.  Idempotent.
hipError_t static_scratch_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_scratch_ptr[device]) return hipSuccess;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
    return hipSuccess;
}

// The device address looked up by static_scratch_setup (null before it ran for `device`).
void *static_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> guard(g_lock);
    return g_scratch_ptr[device];
}

} // namespace rtl
