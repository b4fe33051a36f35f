// rt_build.hip — BVH construction on the device (RT_SCENE_BUILD_DEVICE).
//
// Stands in for the reference's host builder (bvh.h:48-75: recursive median split with
// std::sort, 10 s for a million primitives) when the time to the first pixel matters more
// than the last per cent of traversal speed.  A linear BVH:
//
//   morton_codes    30-bit Morton code of every primitive's box centre
//   (rocPRIM)       radix sort of (code, primitive)
//   radix_tree      one lane per inner node finds its key range and split from common
//                   prefixes of neighbouring codes (ties broken by position), so the whole
//                   hierarchy is emitted in one pass
//   fit_boxes       leaves walk to the root; the second child to arrive at a node merges the
//                   two child boxes and measures the subtree height
//   emit_nodes      the 64-byte nodes of the traversal kernels: both child boxes, as centre
//                   and half extent rounded outwards, and the two child references
//
// The primitive boxes come from the host (the same padded boxes the host builder uses), the
// leaf order goes back to it (it lays out the primitive slots in that order).  Closest hits
// do not depend on the tree (ties are resolved by primitive id), so both builders give the
// same images; this one gives a somewhat slower tree.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <vector>

#include "rt_core.h"
#include "rt_launch.h"

using namespace rtc;

namespace {

struct DBox {
    float lo[3], hi[3];
};

__device__ __forceinline__ unsigned spread10(unsigned v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ void __launch_bounds__(256)
morton_codes(const DBox *boxes, int n, f3 origin, f3 inv_extent, unsigned *keys, int *vals) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DBox b = boxes[i];
    float q[3];
    const float o[3] = {origin.x, origin.y, origin.z}, s[3] = {inv_extent.x, inv_extent.y, inv_extent.z};
    for (int k = 0; k < 3; k++) {
        float c = 0.5f * (b.lo[k] + b.hi[k]);
        float t = (c - o[k]) * s[k] * 1024.0f;
        q[k] = t < 0.0f ? 0.0f : (t > 1023.0f ? 1023.0f : t);
    }
    keys[i] = (spread10((unsigned)q[0]) << 2) | (spread10((unsigned)q[1]) << 1) | spread10((unsigned)q[2]);
    vals[i] = i;
}

// Length of the common prefix of the keys at sorted positions i and j (-1 outside the array);
// equal keys are told apart by their positions.
__device__ __forceinline__ int prefix(const unsigned *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    unsigned a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((unsigned)i ^ (unsigned)j);
    return __clz(a ^ b);
}

// References: >= 0 inner node, < 0 leaf at sorted position ~ref.
__global__ void __launch_bounds__(256)
radix_tree(const unsigned *keys, int n, int *left, int *right, int *parent_inner, int *parent_leaf) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int floor_prefix = prefix(keys, n, i, i - d);
    int span = 2;
    while (prefix(keys, n, i, i + span * d) > floor_prefix) span *= 2;
    int len = 0;
    for (int t = span / 2; t >= 1; t /= 2)
        if (prefix(keys, n, i, i + (len + t) * d) > floor_prefix) len += t;
    int j = i + len * d;
    int node_prefix = prefix(keys, n, i, j);
    int s = 0, t = len;
    do {
        t = (t + 1) / 2;
        if (prefix(keys, n, i, i + (s + t) * d) > node_prefix) s += t;
    } while (t > 1);
    int split = i + s * d + (d < 0 ? d : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    int l = (lo == split) ? ~split : split;
    int r = (hi == split + 1) ? ~(split + 1) : split + 1;
    left[i] = l;
    right[i] = r;
    if (l >= 0) parent_inner[l] = i; else parent_leaf[~l] = i;
    if (r >= 0) parent_inner[r] = i; else parent_leaf[~r] = i;
    if (i == 0) parent_inner[0] = -1;
}

__device__ __forceinline__ DBox merge(const DBox &a, const DBox &b) {
    DBox m;
    for (int k = 0; k < 3; k++) {
        m.lo[k] = fminf(a.lo[k], b.lo[k]);
        m.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    return m;
}

// What one lane writes another reads, within one kernel and possibly from another XCD (whose
// L2 is a different one): device-scope atomic accesses go to memory both ways.
__device__ __forceinline__ DBox load_box(const DBox *p) {
    DBox b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = __hip_atomic_load(&p->lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.hi[k] = __hip_atomic_load(&p->hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}
__device__ __forceinline__ void store_box(DBox *p, const DBox &b) {
    for (int k = 0; k < 3; k++) {
        __hip_atomic_store(&p->lo[k], b.lo[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&p->hi[k], b.hi[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void __launch_bounds__(256)
fit_boxes(const DBox *prim_boxes, const int *vals, int n, const int *left, const int *right, const int *parent_inner,
          const int *parent_leaf, unsigned *arrivals, DBox *inner_boxes, int *height) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int p = parent_leaf[j];
    while (p >= 0) {
        __threadfence(); // what this lane wrote below is visible before it announces itself
        unsigned before = atomicAdd(&arrivals[p], 1u);
        if (before == 0u) return; // the sibling subtree is not done: its last lane carries on
        __threadfence();
        int l = left[p], r = right[p];
        DBox bl = l >= 0 ? load_box(&inner_boxes[l]) : prim_boxes[vals[~l]];
        DBox br = r >= 0 ? load_box(&inner_boxes[r]) : prim_boxes[vals[~r]];
        int hl = l >= 0 ? __hip_atomic_load(&height[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int hr = r >= 0 ? __hip_atomic_load(&height[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        store_box(&inner_boxes[p], merge(bl, br));
        __hip_atomic_store(&height[p], 1 + (hl > hr ? hl : hr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        p = parent_inner[p];
    }
}

// centre - half <= lo and centre + half >= hi must hold exactly (as rt_flatten.h center_half)
__device__ __forceinline__ void center_half(const DBox &b, float *c, float *h) {
    for (int k = 0; k < 3; k++) {
        double lo = b.lo[k], hi = b.hi[k];
        float cf = (float)(0.5 * (lo + hi));
        double need = fmax(hi - (double)cf, (double)cf - lo);
        float hf = (float)need;
        if ((double)hf < need) hf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, hf) + 1u); // next float up (need >= 0)
        c[k] = cf;
        h[k] = hf;
    }
}

__global__ void __launch_bounds__(256)
emit_nodes(const DBox *prim_boxes, const int *vals, const int *types, int n, const int *left, const int *right,
           const DBox *inner_boxes, f4 *nodes) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int ref[2] = {left[i], right[i]};
    float c[2][3], h[2][3];
    int out_ref[2];
    for (int k = 0; k < 2; k++) {
        DBox b;
        if (ref[k] >= 0) {
            b = inner_boxes[ref[k]];
            out_ref[k] = ref[k];
        } else {
            int slot = ~ref[k]; // primitive slots are laid out in sorted order
            int prim = vals[slot];
            b = prim_boxes[prim];
            out_ref[k] = ~(slot * 2 + (types[prim] == RT_PRIM_QUAD ? 1 : 0));
        }
        center_half(b, c[k], h[k]);
    }
    f4 *nd = nodes + (size_t)i * 4;
    nd[0] = f4{c[0][0], c[0][1], c[0][2], h[0][0]};
    nd[1] = f4{h[0][1], h[0][2], c[1][0], c[1][1]};
    nd[2] = f4{c[1][2], h[1][0], h[1][1], h[1][2]};
    nd[3] = f4{__builtin_bit_cast(float, out_ref[0]), __builtin_bit_cast(float, out_ref[1]), 0.0f, 0.0f};
}

struct Scratch { // frees what the build allocated, whichever way it ends
    std::vector<void *> ptrs;
    ~Scratch() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <class T> hipError_t alloc(T **out, size_t count) {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, count * sizeof(T) + 16);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
};

} // namespace

namespace rtl {

// boxes: n x {lo[3], hi[3]} (host); types: n x RT_PRIM_* (host).  On success nodes_out (host,
// (n - 1) x 64 bytes) holds the binary tree, leaf_order[slot] is the primitive of each leaf
// slot and *inner_levels the height of the tree.
hipError_t build_lbvh(const float *boxes, const int *types, int n, float *nodes_out, int *leaf_order,
                      int *inner_levels) {
#define RT_TRY(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return e_;                                                                    \
    } while (0)
    if (n < 2 || !nodes_out) return hipErrorInvalidValue;
    // bounds of the box centres (host: one pass over data that is already here)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {
            float c = 0.5f * (boxes[6 * (size_t)i + k] + boxes[6 * (size_t)i + 3 + k]);
            lo[k] = c < lo[k] ? c : lo[k];
            hi[k] = c > hi[k] ? c : hi[k];
        }
    f3 origin = mk(lo[0], lo[1], lo[2]);
    float ext[3];
    for (int k = 0; k < 3; k++) ext[k] = hi[k] - lo[k] > 0.0f ? 1.0f / (hi[k] - lo[k]) : 0.0f;
    f3 inv_extent = mk(ext[0], ext[1], ext[2]);

    Scratch sc;
    DBox *d_boxes, *d_inner;
    int *d_types, *d_vals, *d_vals_sorted, *d_left, *d_right, *d_parent_inner, *d_parent_leaf, *d_height;
    unsigned *d_keys, *d_keys_sorted, *d_arrivals;
    RT_TRY(sc.alloc(&d_boxes, (size_t)n));
    RT_TRY(sc.alloc(&d_inner, (size_t)n));
    RT_TRY(sc.alloc(&d_types, (size_t)n));
    RT_TRY(sc.alloc(&d_vals, (size_t)n));
    RT_TRY(sc.alloc(&d_vals_sorted, (size_t)n));
    RT_TRY(sc.alloc(&d_keys, (size_t)n));
    RT_TRY(sc.alloc(&d_keys_sorted, (size_t)n));
    RT_TRY(sc.alloc(&d_left, (size_t)n));
    RT_TRY(sc.alloc(&d_right, (size_t)n));
    RT_TRY(sc.alloc(&d_parent_inner, (size_t)n));
    RT_TRY(sc.alloc(&d_parent_leaf, (size_t)n));
    RT_TRY(sc.alloc(&d_height, (size_t)n));
    RT_TRY(sc.alloc(&d_arrivals, (size_t)n));
    f4 *d_nodes = nullptr;
    RT_TRY(sc.alloc(&d_nodes, (size_t)(n - 1) * 4));

    RT_TRY(hipMemcpy(d_boxes, boxes, (size_t)n * sizeof(DBox), hipMemcpyHostToDevice));
    RT_TRY(hipMemcpy(d_types, types, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    RT_TRY(hipMemset(d_arrivals, 0, (size_t)n * sizeof(unsigned)));
    const int grid = (n + 255) / 256;
    hipLaunchKernelGGL(morton_codes, dim3(grid), dim3(256), 0, 0, d_boxes, n, origin, inv_extent, d_keys, d_vals);
    size_t tmp_bytes = 0;
    RT_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, (size_t)n, 0, 30,
                                     (hipStream_t)0));
    char *d_tmp;
    RT_TRY(sc.alloc(&d_tmp, tmp_bytes));
    RT_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, (size_t)n, 0, 30,
                                     (hipStream_t)0));
    hipLaunchKernelGGL(radix_tree, dim3(grid), dim3(256), 0, 0, d_keys_sorted, n, d_left, d_right, d_parent_inner,
                       d_parent_leaf);
    hipLaunchKernelGGL(fit_boxes, dim3(grid), dim3(256), 0, 0, d_boxes, d_vals_sorted, n, d_left, d_right,
                       d_parent_inner, d_parent_leaf, d_arrivals, d_inner, d_height);
    hipLaunchKernelGGL(emit_nodes, dim3(grid), dim3(256), 0, 0, d_boxes, d_vals_sorted, d_types, n, d_left, d_right,
                       d_inner, d_nodes);
    RT_TRY(hipGetLastError());
    RT_TRY(hipMemcpy(leaf_order, d_vals_sorted, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    RT_TRY(hipMemcpy(inner_levels, d_height, sizeof(int), hipMemcpyDeviceToHost)); // node 0 is the root
    RT_TRY(hipMemcpy(nodes_out, d_nodes, (size_t)(n - 1) * 4 * sizeof(f4), hipMemcpyDeviceToHost));
    return hipSuccess;
#undef RT_TRY
}

// Order of a launch's 64-pixel strips for the next launch of the same image: most expensive
// first (keys from reduce_stage: the sum over the strip's paths of their squared ray counts), so that a
// launch ends on short paths and its waves run dry and drain together.
size_t order_strips_tmp_bytes(unsigned n) {
    size_t bytes = 0;
    unsigned *k = nullptr;
    if (rocprim::radix_sort_pairs_desc(nullptr, bytes, k, k, k, k, (size_t)n, 0, 32, (hipStream_t)0) != hipSuccess) return 0;
    return bytes;
}

namespace {
__global__ void iota_kernel(unsigned *v, unsigned n) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
} // namespace

hipError_t fill_iota(unsigned *v, unsigned n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, n);
    return hipGetLastError();
}

hipError_t order_strips(const unsigned *cost, const unsigned *iota, unsigned n, unsigned *cost_sorted, unsigned *order,
                        void *tmp, size_t tmp_bytes, hipStream_t stream) {
    return rocprim::radix_sort_pairs_desc(tmp, tmp_bytes, cost, cost_sorted, iota, order, (size_t)n, 0, 32, stream);
}

} // namespace rtl
