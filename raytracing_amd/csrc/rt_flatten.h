// rt_flatten.h — host-only: turns an rt_scene_desc into the flat arrays the
// kernels read (rt_core.h: primitive slots, per-slot meta, BVH nodes, wrapper ops).
// No HIP calls here; rt_host.cpp uploads the result.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_core.h"

namespace rtf {

using rtc::f3;
using rtc::f4;
using rtc::i4;
using rtc::mk;

const int kLinearMaxPrims = 32;       // at most this many primitives: scan, no tree
const int kMaxLdsBytes = 160 * 1024;  // LDS per CU on gfx950
const int kParallelBuildMinPrims = 65536; // scenes from this size on: subtrees built by several threads
const int kTreeTopMin = 1280;         // trees of at least this many 128-byte nodes never fit LDS whole
const int kTreeTopMax = 1024;         // nodes moved to the front for the LDS copy of the tree's top
const float kBigPrimArea = 0.5f;      // a primitive whose box has this share of the scene's box area is tested up front
const int kBigPrimMax = 4;            // ... at most this many of them
const int kMaxInnerDepth = RT_MAX_TREE_LEVELS - 1; // of the binary tree (the root is at depth 0)

// ---------------------------------------------------------------------------------------
// wrapper ops on the host (same formulas as rt_core.h uses on the device)
struct HostOp {
    int type;
    f3 p;
    f4 a, b; // {sa ca sb cb} {sg cg 0 0}
    rtc::Mat3 cw, ccw; // rotate_xyz: world -> object and object -> world
};

HostOp make_op(const rt_xform_op &op) {
    HostOp h;
    h.type = op.type;
    h.p = mk(op.p[0], op.p[1], op.p[2]);
    h.a = f4{0.0f, 1.0f, 0.0f, 1.0f};
    h.b = f4{0.0f, 1.0f, 0.0f, 0.0f};
    const float pi = RT_PI;
    if (op.type == RT_XF_ROTATE_Y) { // hittable.h:102-105
        float radians = op.p[0] * pi / 180.0f;
        h.a.z = sinf(radians);
        h.a.w = cosf(radians);
    } else if (op.type == RT_XF_ROTATE_XYZ) { // hittable.h:214-223
        float ar = op.p[0] * pi / 180.0f, br = op.p[1] * pi / 180.0f, gr = op.p[2] * pi / 180.0f;
        h.a.x = sinf(ar);
        h.a.y = cosf(ar);
        h.a.z = sinf(br);
        h.a.w = cosf(br);
        h.b.x = sinf(gr);
        h.b.y = cosf(gr);
    }
    h.cw = rtc::rot_matrix_cw(h.a.x, h.a.y, h.a.z, h.a.w, h.b.x, h.b.y);
    h.ccw = rtc::rot_matrix_ccw(h.a.x, h.a.y, h.a.z, h.a.w, h.b.x, h.b.y);
    return h;
}

f3 op_point(const HostOp &h, f3 p) {
    switch (h.type) {
    case RT_XF_TRANSLATE: return p + h.p;
    case RT_XF_ROTATE_Y: return rtc::rot_y_ccw(h.a, p);
    case RT_XF_ROTATE_XYZ: return rtc::mat_apply(h.ccw, p);
    default: return p * h.p;
    }
}
f3 op_vector(const HostOp &h, f3 v) {
    switch (h.type) {
    case RT_XF_TRANSLATE: return v;
    case RT_XF_ROTATE_Y: return rtc::rot_y_ccw(h.a, v);
    case RT_XF_ROTATE_XYZ: return rtc::mat_apply(h.ccw, v);
    default: return v * h.p;
    }
}

// quad constants as the constructor derives them (quad.h:15-20)
void quad_slot(f3 Q, f3 u, f3 v, f4 out[4]) {
    f3 n = rtc::cross(u, v);
    f3 normal = rtc::unit(n);
    float D = rtc::dot(normal, Q);
    f3 w = n / rtc::dot(n, n);
    out[0] = f4{normal.x, normal.y, normal.z, D};
    out[1] = f4{Q.x, Q.y, Q.z, w.x};
    out[2] = f4{w.y, w.z, u.x, u.y};
    out[3] = f4{u.z, v.x, v.y, v.z};
}

// Scan record of a slot (rt_core.h: RT_SCAN_*).  A quad whose normal, u and v each lie along a
// coordinate axis (the sides of box(), walls) gets the seven numbers quad_test_axis needs, in
// the canonical orientation (u along axis K + 1, v along K + 2; if the quad has them the other
// way round, u and v change places, which negates n, D and w).
inline void scan_record(int type, int chain, const f4 *g, f4 out[3]) {
    auto as_f = [](int v) { return __builtin_bit_cast(float, v); };
    out[0] = f4{as_f(type == RT_PRIM_SPHERE ? RT_SCAN_SPHERE : RT_SCAN_QUAD), as_f(chain), 0.0f, 0.0f};
    out[1] = out[2] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    if (type == RT_PRIM_SPHERE) { // centre, radius, centre motion: what sphere_test reads of the slot
        out[0].z = g[0].x, out[0].w = g[0].y;
        out[1] = f4{g[0].z, g[0].w, g[1].x, g[1].y};
        out[2].x = g[1].z;
        return;
    }
    const float n[3] = {g[0].x, g[0].y, g[0].z}, w[3] = {g[1].w, g[2].x, g[2].y}, Q[3] = {g[1].x, g[1].y, g[1].z};
    const float u[3] = {g[2].z, g[2].w, g[3].x}, v[3] = {g[3].y, g[3].z, g[3].w};
    const float D = g[0].w;
    for (int c = 0; c < 3; c++)
        if (!std::isfinite(n[c]) || !std::isfinite(w[c]) || !std::isfinite(u[c]) || !std::isfinite(v[c]) ||
            !std::isfinite(Q[c]) || !std::isfinite(D))
            return;
    for (int k = 0; k < 3; k++) {
        const int i = (k + 1) % 3, j = (k + 2) % 3;
        if (n[k] == 0.0f || n[i] != 0.0f || n[j] != 0.0f) continue;
        if (w[k] == 0.0f || w[i] != 0.0f || w[j] != 0.0f) continue;
        if (u[k] != 0.0f || v[k] != 0.0f) continue;
        const bool straight = u[i] != 0.0f && u[j] == 0.0f && v[i] == 0.0f && v[j] != 0.0f;
        const bool swapped = u[i] == 0.0f && u[j] != 0.0f && v[i] != 0.0f && v[j] == 0.0f;
        if (!straight && !swapped) continue;
        out[0].x = as_f(RT_SCAN_AXIS + k);
        if (straight) {
            out[0].z = n[k], out[0].w = D;
            out[1] = f4{Q[i], Q[j], w[k], u[i]};
            out[2].x = v[j];
        } else {
            out[0].z = -n[k], out[0].w = -D;
            out[1] = f4{Q[i], Q[j], -w[k], v[i]};
            out[2].x = u[j];
        }
        return;
    }
}

struct Box3 {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; k++) {
            lo[k] = INFINITY;
            hi[k] = -INFINITY;
        }
    }
    void add(f3 p) {
        const float c[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], c[k]);
            hi[k] = std::max(hi[k], c[k]);
        }
    }
    void add(const Box3 &b) {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

// The exact primitive tests run in fp32 and can report a hit for a ray that misses
// the ideal primitive by a hair (a sphere seen from distance L can be off by about
// 2e-7 L^2 / r).  Culling must never drop such a hit, so boxes are grown by a small
// fraction of their size plus a few ulps of their coordinates.
// rel: the fraction of the box's largest extent.  A sphere's box is grown by RT_SPHERE_PAD_REL of it
// (at least 2 % of the radius): the tree walk keeps a sphere's hit only if the ray's closest
// approach to the centre lies within 1.9 % of the radius of the sphere's own box
// (rt_core.h sphere_approach_in_box), and that smaller box must lie inside this one.
#define RT_SPHERE_PAD_REL 1e-2f
void pad_box(Box3 &b, float rel = 2e-3f) {
    float ext = std::max(b.hi[0] - b.lo[0], std::max(b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]));
    float mag = 0.0f;
    for (int k = 0; k < 3; k++) mag = std::max(mag, std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])));
    float pad = rel * ext + 1e-5f * mag + 1e-30f;
    for (int k = 0; k < 3; k++) {
        b.lo[k] -= pad;
        b.hi[k] += pad;
    }
    // A primitive with a coordinate that is not a number, infinite or beyond 3e37 (the reference
    // never hits one either) gets a box that spans everything along that axis: the builders then
    // only ever see finite, ordered numbers (their sorts and bins rely on that), every ray is sent
    // to the primitive's exact test, and that test decides as it would have anyway.
    const float kHuge = 3e37f;
    for (int k = 0; k < 3; k++)
        if (!(b.lo[k] >= -kHuge && b.hi[k] <= kHuge && b.lo[k] <= b.hi[k])) {
            b.lo[k] = -kHuge;
            b.hi[k] = kHuge;
        }
}

// Centre and half extent of a box; centre - half <= lo and centre + half >= hi hold exactly.
inline void center_half(const Box3 &b, float *c, float *h) {
    for (int k = 0; k < 3; k++) {
        double lo = b.lo[k], hi = b.hi[k];
        float cf = (float)(0.5 * (lo + hi));
        double need = std::max(hi - (double)cf, (double)cf - lo);
        float hf = (float)need;
        while ((double)hf < need) hf = std::nextafter(hf, INFINITY);
        c[k] = cf;
        h[k] = hf;
    }
}

// ---------------------------------------------------------------------------------------
// The 4-wide tree the kernels walk (rt_core.h trav_inner: 128-byte nodes holding the boxes of up
// to four children) is collapsed from the binary tree the builders make (64-byte nodes:
// {c0.xyz h0.x} {h0.yz c1.xy} {c1.z h1.xyz} {ref0 ref1 - -}).
inline float half_area(const float *h) { return h[0] * h[1] + h[1] * h[2] + h[2] * h[0]; }

struct Wide {
    std::vector<f4> nodes; // 8 per node
    int n_nodes = 0;
    int levels = 0;        // inner levels (a tree that is one node has 1)
};

// Which binary nodes become wide nodes is chosen to minimise the surface-area cost of the wide
// tree (sum over its nodes of the area of their box: the expected number of node visits of a
// random ray), by dynamic programming over the binary tree: cost[n][i] is the cheapest way to
// hang subtree n into i child slots of a wide node, either as one child that is a wide node
// itself (its own four slots shared out between n's two children) or, for i > 1, by sharing the
// i slots out between n's children directly.
inline Wide collapse_to_wide(const std::vector<f4> &bin) {
    Wide w;
    const int n2 = (int)(bin.size() / 4);
    if (n2 == 0) return w;
    auto words = [&](int node) { return reinterpret_cast<const float *>(&bin[(size_t)node * 4]); };
    auto child_ref = [&](int node, int k) { return __builtin_bit_cast(int, words(node)[12 + k]); };
    // post-order of the binary nodes, and the area of each one's box (kept in its parent)
    std::vector<int> order;
    std::vector<float> area((size_t)n2, 0.0f);
    order.reserve(n2);
    {
        std::vector<int> stack{0};
        while (!stack.empty()) {
            int n = stack.back();
            stack.pop_back();
            order.push_back(n);
            for (int k = 0; k < 2; k++) {
                int r = child_ref(n, k);
                if (r >= 0) {
                    area[r] = half_area(words(n) + 6 * k + 3);
                    stack.push_back(r);
                }
            }
        }
        const float *q = words(0); // the root's own box: the union of its children's
        float h[3];
        for (int a = 0; a < 3; a++) {
            float lo = std::min(q[a] - q[3 + a], q[6 + a] - q[9 + a]), hi = std::max(q[a] + q[3 + a], q[6 + a] + q[9 + a]);
            h[a] = 0.5f * (hi - lo);
        }
        area[0] = half_area(h);
    }
    struct Cost {
        float c[5];       // c[i], i = 1..4 (a leaf costs nothing wherever it hangs)
        unsigned char split[5]; // slots given to the left child when i slots are shared out; 0: c[i] = c[i-1]
        unsigned char node_split;  // the same for the node's own four slots
    };
    std::vector<Cost> cost((size_t)n2);
    auto sub = [&](int ref, int i) { return ref >= 0 ? cost[ref].c[i] : 0.0f; };
    for (size_t t = order.size(); t-- > 0;) {
        const int n = order[t];
        const int l = child_ref(n, 0), r = child_ref(n, 1);
        Cost &c = cost[n];
        float best = INFINITY;
        c.node_split = 2; // (kept when every cost is infinite: boxes beyond the range of float areas)
        for (int k = 1; k <= 3; k++) {
            float v = sub(l, k) + sub(r, 4 - k);
            if (v < best) {
                best = v;
                c.node_split = (unsigned char)k;
            }
        }
        c.c[1] = area[n] + best;
        c.split[1] = 0;
        for (int i = 2; i <= 4; i++) {
            c.c[i] = c.c[i - 1];
            c.split[i] = 0;
            for (int k = 1; k < i; k++) {
                float v = sub(l, k) + sub(r, i - k);
                if (v < c.c[i]) {
                    c.c[i] = v;
                    c.split[i] = (unsigned char)k;
                }
            }
        }
    }
    // top-down: every wide node gathers its children in the binary tree's left-to-right order
    w.nodes.reserve((size_t)(n2 / 2 + 1) * 8);
    struct Item {
        int bin_node, wide_node, level;
    };
    struct Child {
        float box[6];
        int ref;
    };
    std::vector<Item> todo;
    auto new_node = [&]() {
        w.nodes.resize(w.nodes.size() + 8, f4{0.0f, 0.0f, 0.0f, 0.0f});
        return w.n_nodes++;
    };
    int null_ref = child_ref(0, 0); // the leftmost leaf
    while (null_ref >= 0) null_ref = child_ref(null_ref, 0);
    todo.push_back({0, new_node(), 1});
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        w.levels = std::max(w.levels, it.level);
        Child c[4];
        int n = 0;
        // shares `slots` slots out below binary node `node`; `box` is the node's own box
        struct Frame {
            int node, k, slots;
        };
        std::vector<Frame> fr;
        const int ks = cost[it.bin_node].node_split;
        fr.push_back({it.bin_node, 1, 4 - ks});
        fr.push_back({it.bin_node, 0, ks});
        while (!fr.empty()) {
            const Frame f = fr.back();
            fr.pop_back();
            const float *q = words(f.node);
            const int ref = child_ref(f.node, f.k);
            int slots = f.slots;
            if (ref >= 0)
                while (slots > 1 && cost[ref].split[slots] == 0) slots--; // as good with fewer slots
            if (ref < 0 || slots == 1) { // a leaf, or a subtree that becomes a wide node of its own
                memcpy(c[n].box, q + 6 * f.k, 6 * sizeof(float));
                c[n].ref = ref;
                n++;
            } else {
                const int kl = cost[ref].split[slots];
                fr.push_back({ref, 1, slots - kl});
                fr.push_back({ref, 0, kl});
            }
        }
        // An unused child names a real leaf: no finite ray enters its box, but a ray of NaNs
        // passes every box test, and a reference to node 0 would send it round for ever (and its
        // stack past its end); this way it ends in one more exact test that it fails.
        int refs[4] = {null_ref, null_ref, null_ref, null_ref};
        float boxes[24];
        for (int k = 0; k < 4; k++) {
            if (k < n) {
                memcpy(boxes + 6 * k, c[k].box, 6 * sizeof(float));
                if (c[k].ref >= 0) {
                    refs[k] = new_node();
                    todo.push_back({c[k].ref, refs[k], it.level + 1});
                } else {
                    refs[k] = c[k].ref;
                }
            } else { // unused: a box no ray enters
                for (int a = 0; a < 3; a++) {
                    boxes[6 * k + a] = 0.0f;
                    boxes[6 * k + 3 + a] = -1e30f;
                }
            }
        }
        float *q = reinterpret_cast<float *>(&w.nodes[(size_t)it.wide_node * 8]);
        memcpy(q, boxes, sizeof(boxes));
        for (int k = 0; k < 4; k++) q[24 + k] = __builtin_bit_cast(float, refs[k]);
        q[28] = __builtin_bit_cast(float, n);
    }
    return w;
}

// Moves the most visited part of a wide tree to the front of the node array, so that a kernel
// can keep a prefix of it in LDS while the rest stays in HBM.  Nodes are picked greedily by
// the surface area of their box (the chance that a random ray visits them), starting at the
// root; every prefix of the new order is a connected top of the tree.  The remaining nodes
// keep their relative order.  Returns the number of nodes picked (at most max_top).
inline int reorder_tree_top(std::vector<f4> &nodes, int max_top) {
    const int n = (int)(nodes.size() / 8);
    if (n <= 1 || max_top <= 0) return n <= 1 ? n : 0;
    auto words = [&](int node) { return reinterpret_cast<float *>(&nodes[(size_t)node * 8]); };
    std::vector<int> picked;
    std::vector<std::pair<float, int>> heap; // (area, node)
    picked.reserve(max_top);
    heap.push_back({INFINITY, 0});
    while (!heap.empty() && (int)picked.size() < max_top) {
        std::pop_heap(heap.begin(), heap.end());
        int node = heap.back().second;
        heap.pop_back();
        picked.push_back(node);
        const float *q = words(node);
        const int nc = __builtin_bit_cast(int, q[28]);
        for (int k = 0; k < nc; k++) {
            int r = __builtin_bit_cast(int, q[24 + k]);
            if (r >= 0) {
                heap.push_back({half_area(q + 6 * k + 3), r});
                std::push_heap(heap.begin(), heap.end());
            }
        }
    }
    std::vector<int> new_index(n, -1);
    for (size_t i = 0; i < picked.size(); i++) new_index[picked[i]] = (int)i;
    int next = (int)picked.size();
    for (int i = 0; i < n; i++)
        if (new_index[i] < 0) new_index[i] = next++;
    std::vector<f4> out(nodes.size());
    for (int i = 0; i < n; i++) {
        float *q = reinterpret_cast<float *>(&out[(size_t)new_index[i] * 8]);
        memcpy(q, words(i), 8 * sizeof(f4));
        const int nc = __builtin_bit_cast(int, q[28]);
        for (int k = 0; k < nc; k++) {
            int r = __builtin_bit_cast(int, q[24 + k]);
            if (r >= 0) q[24 + k] = __builtin_bit_cast(float, new_index[r]);
        }
    }
    nodes.swap(out);
    return (int)picked.size();
}

// The node layout the kernels read (rt_core.h trav_inner): the builders and the passes above work
// on child boxes as centre + half extent, six floats per child; this last pass turns every node
// into planes grouped by axis, {lo_x of the four children} {hi_x} {lo_y} {hi_y} {lo_z} {hi_z},
// so that a ray reads its near and its far plane of an axis with one 16-byte load each, picked by
// the sign of its direction.  lo = c - h and hi = c + h are rounded outwards (the stored box
// contains the built one); an unused child keeps an inverted box.
inline void nodes_to_plane_layout(std::vector<f4> &nodes) {
    const size_t n = nodes.size() / 8;
    for (size_t i = 0; i < n; i++) {
        float *q = reinterpret_cast<float *>(&nodes[i * 8]);
        float lo[3][4], hi[3][4];
        for (int k = 0; k < 4; k++)
            for (int a = 0; a < 3; a++) {
                const float c = q[6 * k + a], h = q[6 * k + 3 + a];
                if (h < 0.0f) { // unused child
                    lo[a][k] = 1e30f;
                    hi[a][k] = -1e30f;
                } else {
                    lo[a][k] = std::nextafter(c - h, -INFINITY);
                    hi[a][k] = std::nextafter(c + h, INFINITY);
                }
            }
        for (int a = 0; a < 3; a++)
            for (int k = 0; k < 4; k++) {
                q[8 * a + k] = lo[a][k];
                q[8 * a + 4 + k] = hi[a][k];
            }
    }
}

// ---------------------------------------------------------------------------------------
// Binned-SAH builder of the binary tree.  Leaves hold one primitive; a parent stores the
// boxes of both children.
struct BvhOut {
    std::vector<f4> nodes;       // 4 per binary node
    std::vector<int> leaf_order; // slot -> input primitive index
    int max_inner_depth = 0;
};

class BvhBuilder {
  public:
    BvhBuilder(const std::vector<Box3> &boxes, const std::vector<int> &types)
        : boxes_(boxes), types_(types) {
        size_t n = boxes.size();
        idx_.resize(n);
        cen_.resize(n * 3);
        for (size_t i = 0; i < n; i++) {
            idx_[i] = (int)i;
            for (int k = 0; k < 3; k++) cen_[i * 3 + k] = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
        }
    }

    // Large scenes: the top of the tree is split on this thread until the subtrees are small, the
    // subtrees are built by a few threads (they own disjoint ranges of the index array), and the
    // pieces are joined; splits, leaf order and therefore the tree are the same as built by one thread.
    BvhOut run(int n_threads = 0) {
        const int n = (int)idx_.size();
        if (n_threads <= 0) {
            n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
            if (const char *e = getenv("RT_AMD_BUILD_THREADS")) n_threads = std::max(1, atoi(e));
        }
        Box3 b;
        if (n < kParallelBuildMinPrims || n_threads < 2) {
            out_.nodes.reserve((size_t)n * 4);
            out_.leaf_order.reserve(n);
            build(0, n, 0, b, out_, 0);
            return std::move(out_);
        }
        tasks_.clear();
        build(0, n, 0, b, out_, std::max(1024, n / (n_threads * 8)));
        std::atomic<size_t> next(0);
        auto work = [&]() {
            for (size_t t = next++; t < tasks_.size(); t = next++) {
                Task &k = tasks_[t];
                Box3 tb;
                k.out.nodes.reserve((size_t)k.count * 4);
                k.out.leaf_order.reserve(k.count);
                k.ref = build(k.first, k.count, k.depth, tb, k.out, 0);
            }
        };
        std::vector<std::thread> pool;
        for (int i = 1; i < n_threads; i++) pool.emplace_back(work);
        work();
        for (std::thread &th : pool) th.join();
        // join: task t's nodes follow the top nodes and the earlier tasks', its slots the earlier tasks'
        const int n_top = (int)(out_.nodes.size() / 4);
        std::vector<int> node_base(tasks_.size()), slot_base(tasks_.size());
        int nodes_total = n_top, slots_total = 0;
        for (size_t t = 0; t < tasks_.size(); t++) {
            node_base[t] = nodes_total;
            slot_base[t] = slots_total;
            nodes_total += (int)(tasks_[t].out.nodes.size() / 4);
            slots_total += (int)tasks_[t].out.leaf_order.size();
        }
        auto moved = [&](int ref, size_t t) {
            if (ref >= 0) return ref + node_base[t];
            int code = ~ref;
            return ~((((code >> 1) + slot_base[t]) << 1) | (code & 1));
        };
        out_.nodes.resize((size_t)nodes_total * 4);
        out_.leaf_order.resize(slots_total);
        for (int i = 0; i < n_top; i++) { // placeholders in the top nodes
            f4 &r = out_.nodes[(size_t)i * 4 + 3];
            int r0 = __builtin_bit_cast(int, r.x), r1 = __builtin_bit_cast(int, r.y);
            if (r0 >= kTaskRef) r.x = __builtin_bit_cast(float, moved(tasks_[r0 - kTaskRef].ref, (size_t)(r0 - kTaskRef)));
            if (r1 >= kTaskRef) r.y = __builtin_bit_cast(float, moved(tasks_[r1 - kTaskRef].ref, (size_t)(r1 - kTaskRef)));
        }
        for (size_t t = 0; t < tasks_.size(); t++) {
            const BvhOut &o = tasks_[t].out;
            for (size_t i = 0; i < o.nodes.size() / 4; i++) {
                f4 *q = &out_.nodes[((size_t)node_base[t] + i) * 4];
                memcpy(q, &o.nodes[i * 4], 4 * sizeof(f4));
                q[3].x = __builtin_bit_cast(float, moved(__builtin_bit_cast(int, q[3].x), t));
                q[3].y = __builtin_bit_cast(float, moved(__builtin_bit_cast(int, q[3].y), t));
            }
            std::copy(o.leaf_order.begin(), o.leaf_order.end(), out_.leaf_order.begin() + slot_base[t]);
            out_.max_inner_depth = std::max(out_.max_inner_depth, o.max_inner_depth);
        }
        tasks_.clear();
        return std::move(out_);
    }

  private:
    static const int kTaskRef = 1 << 30; // child references >= this name a task of the parallel build
    struct Task {
        int first, count, depth;
        BvhOut out;
        int ref = 0;
    };
    std::vector<Task> tasks_;

    static int ceil_log2(int n) {
        int l = 0;
        while ((1 << l) < n) l++;
        return l;
    }

    // Builds the subtree over idx_[first, first+count) into `out` and returns its reference.
    // task_size > 0: subtrees of at most that many primitives are only recorded as tasks.
    int build(int first, int count, int depth, Box3 &box, BvhOut &out, int task_size) {
        box.reset();
        for (int i = first; i < first + count; i++) box.add(boxes_[idx_[i]]);
        if (task_size > 0 && count <= task_size) {
            Task k;
            k.first = first;
            k.count = count;
            k.depth = depth;
            tasks_.push_back(std::move(k));
            return kTaskRef + (int)tasks_.size() - 1;
        }
        if (count == 1) {
            int prim = idx_[first];
            int slot = (int)out.leaf_order.size();
            out.leaf_order.push_back(prim);
            return ~(slot * 2 + (types_[prim] == RT_PRIM_QUAD ? 1 : 0));
        }
        out.max_inner_depth = std::max(out.max_inner_depth, depth);
        int node = (int)(out.nodes.size() / 4);
        out.nodes.resize(out.nodes.size() + 4);

        int mid = split_sah(first, count);
        int nl = mid - first, nr = count - nl;
        bool ok = nl > 0 && nr > 0 && depth + ceil_log2(nl) <= kMaxInnerDepth &&
                  depth + ceil_log2(nr) <= kMaxInnerDepth;
        if (!ok) mid = split_median(first, count);

        Box3 b0, b1;
        int r0 = build(first, mid - first, depth + 1, b0, out, task_size);
        int r1 = build(mid, first + count - mid, depth + 1, b1, out, task_size);
        f4 *n = &out.nodes[(size_t)node * 4];
        // boxes are stored as centre + half extent (rounded up so that the stored box contains
        // the built one): the slab test then needs no per-axis min / max
        float c0[3], h0[3], c1[3], h1[3];
        center_half(b0, c0, h0);
        center_half(b1, c1, h1);
        n[0] = f4{c0[0], c0[1], c0[2], h0[0]};
        n[1] = f4{h0[1], h0[2], c1[0], c1[1]};
        n[2] = f4{c1[2], h1[0], h1[1], h1[2]};
        n[3] = f4{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1), 0.0f, 0.0f};
        return node;
    }

    int split_median(int first, int count) {
        Box3 cb;
        cb.reset();
        for (int i = first; i < first + count; i++) {
            const float *c = &cen_[(size_t)idx_[i] * 3];
            cb.add(mk(c[0], c[1], c[2]));
        }
        int axis = 0;
        for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        int mid = first + count / 2;
        std::nth_element(idx_.begin() + first, idx_.begin() + mid, idx_.begin() + first + count,
                         [&](int a, int b) { return cen_[(size_t)a * 3 + axis] < cen_[(size_t)b * 3 + axis]; });
        return mid;
    }

    int split_sah(int first, int count) {
        const int NB = 16;
        Box3 cb;
        cb.reset();
        for (int i = first; i < first + count; i++) {
            const float *c = &cen_[(size_t)idx_[i] * 3];
            cb.add(mk(c[0], c[1], c[2]));
        }
        float best_cost = INFINITY;
        int best_axis = -1, best_bin = -1;
        for (int axis = 0; axis < 3; axis++) {
            float lo = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0.0f)) continue;
            Box3 bb[NB];
            int bc[NB];
            for (int b = 0; b < NB; b++) {
                bb[b].reset();
                bc[b] = 0;
            }
            float scale = NB / ext;
            for (int i = first; i < first + count; i++) {
                int p = idx_[i];
                int b = std::min(NB - 1, std::max(0, (int)((cen_[(size_t)p * 3 + axis] - lo) * scale)));
                bb[b].add(boxes_[p]);
                bc[b]++;
            }
            float right_area[NB];
            int right_cnt[NB];
            Box3 acc;
            acc.reset();
            int cnt = 0;
            for (int b = NB - 1; b > 0; b--) {
                acc.add(bb[b]);
                cnt += bc[b];
                right_area[b] = cnt ? acc.area() : 0.0f;
                right_cnt[b] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int b = 0; b < NB - 1; b++) {
                acc.add(bb[b]);
                cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.area() * cnt + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = axis;
                    best_bin = b;
                }
            }
        }
        if (best_axis < 0) return first; // all centroids coincide: caller falls back to the median
        float lo = cb.lo[best_axis], scale = NB / (cb.hi[best_axis] - cb.lo[best_axis]);
        auto it = std::partition(idx_.begin() + first, idx_.begin() + first + count, [&](int p) {
            int b = std::min(NB - 1, std::max(0, (int)((cen_[(size_t)p * 3 + best_axis] - lo) * scale)));
            return b <= best_bin;
        });
        return (int)(it - idx_.begin());
    }

    const std::vector<Box3> &boxes_;
    const std::vector<int> &types_;
    std::vector<int> idx_;
    std::vector<float> cen_;
    BvhOut out_;
};


// Flat scene on the host.
struct HostScene {
    std::vector<f4> slots;       // n_prims * slot_f4
    std::vector<i4> rec_i;       // per slot {kind, texture, chain, prim id}   (rt_core.h)
    std::vector<f4> rec_f;       // per slot {r g b param}
    std::vector<rt_texture> textures; // textures reachable from non-solid material textures
    std::vector<f4> nodes;       // 8 per node of the 4-wide tree (empty: linear scan)
    std::vector<f4> xf;          // RT_XF_F4 per wrapper op, packed chain by chain
    std::vector<f4> scan;        // linear scenes: RT_SCAN_F4 per slot (rt_core.h trace_linear)
    std::vector<rtc::DevPerlin> perlins;
    std::vector<int> chain_first; // n_chains + 1
    int slot_f4 = 2;
    int n_prims = 0, n_spheres = 0, n_quads = 0;
    int n_nodes = 0, bvh_depth = 0;
    int n_top = 0; // nodes [0, n_top) are the most visited ones (reorder_tree_top); 0: order as built
    bool linear = true;
    bool any_chain = false; // some primitive sits inside instance wrappers
    bool external_nodes = false; // the binary tree came from the external (device) builder
    int n_bin_nodes = 0;         // nodes of the binary tree the wide one was collapsed from
    // two-level scenes (rt_instance): per instance {root reference of its sub-scene's tree, wrapper
    // chain, what to add to a primitive's id to order it among all primitives of the world, -}
    std::vector<i4> inst;
    // primitives kept out of the tree and tested by every ray up front: slots [0, n_big)
    int n_big = 0, big_quads = 0;
};

// Adds node_base to the inner references of wide nodes [first, first + count) and maps their
// leaf slots through leaf_slot (joining trees that were built apart).
template <class F>
inline void relocate_wide(std::vector<f4> &nodes, size_t first, size_t count, int node_base, F leaf_slot) {
    for (size_t i = first; i < first + count; i++) {
        float *q = reinterpret_cast<float *>(&nodes[i * 8]);
        for (int k = 0; k < 4; k++) {
            int ref = __builtin_bit_cast(int, q[24 + k]);
            if (ref >= 0) {
                ref += node_base;
            } else {
                const int code = ~ref;
                ref = ~((leaf_slot(code >> 1) << 1) | (code & 1));
            }
            q[24 + k] = __builtin_bit_cast(float, ref);
        }
    }
}

// A wide node with one child (a tree over a single item still needs a root to start from).
inline void single_child_node(std::vector<f4> &nodes, const Box3 &b, int ref) {
    nodes.resize(nodes.size() + 8, f4{0.0f, 0.0f, 0.0f, 0.0f});
    float *q = reinterpret_cast<float *>(&nodes[nodes.size() - 8]);
    float c[3], h[3];
    center_half(b, c, h);
    for (int k = 0; k < 4; k++)
        for (int a = 0; a < 3; a++) {
            q[6 * k + a] = k == 0 ? c[a] : 0.0f;
            q[6 * k + 3 + a] = k == 0 ? h[a] : -1e30f;
        }
    for (int k = 0; k < 4; k++) q[24 + k] = __builtin_bit_cast(float, ref);
    q[28] = __builtin_bit_cast(float, 1);
}

// flags: RT_SCENE_FORCE_* (rt_amd.h).  The descriptor must already be validated.
// An external builder (the device-side one) gets the padded world-space boxes and the
// primitive types and fills the binary nodes, leaf_order and max_inner_depth; returning false
// falls back to the host builder.
using ExternalBuilder = std::function<bool(const std::vector<Box3> &, const std::vector<int> &, BvhOut &)>;

inline int flatten_scene(const rt_scene_desc *d, uint32_t flags, HostScene &out, std::string &err,
                         const ExternalBuilder &external = nullptr) {
    const int n = d->n_prims;
    bool linear = n <= kLinearMaxPrims;
    if (flags & RT_SCENE_FORCE_BVH) linear = n < 2;
    if (flags & RT_SCENE_FORCE_LINEAR) linear = true;
    const bool two_level = d->n_instances > 0;
    if (two_level) linear = false; // a sub-scene is entered through a leaf of a tree

    std::vector<HostOp> ops(d->n_xform_ops);
    for (int i = 0; i < d->n_xform_ops; i++) ops[i] = make_op(d->xform_ops[i]);

    bool any_quad = false;
    for (int i = 0; i < n; i++) any_quad = any_quad || d->prims[i].type == RT_PRIM_QUAD;
    const int slot_f4 = any_quad ? 4 : 2;

    // per-primitive constants in input (test) order
    std::vector<f4> geom((size_t)n * slot_f4, f4{0, 0, 0, 0});
    std::vector<Box3> boxes(linear ? 0 : n);
    std::vector<int> types(n);
    for (int i = 0; i < n; i++) {
        const rt_prim &p = d->prims[i];
        types[i] = p.type;
        f4 *g = &geom[(size_t)i * slot_f4];
        const bool wrapped = p.xform >= 0;
        int first = 0, last = 0;
        if (wrapped) {
            first = d->chains[p.xform].first_op;
            last = first + d->chains[p.xform].n_ops;
        }
        // A wrapped primitive keeps its object-space constants in either traversal: the ray is
        // taken into object space when the primitive is tested, wrapper by wrapper, as the
        // reference does (hittable.h:57-69, 251-269, 319-334).  A BVH only needs its world-space
        // box: the corners of the object-space box pushed through the wrappers.
        auto to_world = [&](f3 q) {
            for (int k = first; k < last; k++) q = op_point(ops[k], q);
            return q;
        };
        if (wrapped) out.any_chain = true;
        if (p.type == RT_PRIM_SPHERE) {
            f3 c1 = mk(p.a[0], p.a[1], p.a[2]);
            f3 cv = (p.flags & RT_PRIM_FLAG_MOVING) ? mk(p.b[0], p.b[1], p.b[2]) : mk(0.0f, 0.0f, 0.0f);
            float radius = p.c[0];
            g[0] = f4{c1.x, c1.y, c1.z, radius};
            // .w: the reciprocal sphere.h:68 multiplies by ((p - center) / radius = (1 / radius) * v, vec3.h:107)
            g[1] = f4{cv.x, cv.y, cv.z, 1.0f / radius};
            if (!linear) {
                Box3 ob; // object space
                ob.reset();
                f3 rv = mk(std::fabs(radius), std::fabs(radius), std::fabs(radius));
                f3 c2 = c1 + cv; // ray times lie in [0,1): c1 + t*cv stays between c1 and c1 + cv
                ob.add(c1 - rv);
                ob.add(c1 + rv);
                ob.add(c2 - rv);
                ob.add(c2 + rv);
                Box3 &b = boxes[i];
                if (!wrapped) {
                    b = ob;
                } else {
                    b.reset();
                    for (int corner = 0; corner < 8; corner++)
                        b.add(to_world(mk((corner & 1) ? ob.hi[0] : ob.lo[0], (corner & 2) ? ob.hi[1] : ob.lo[1],
                                          (corner & 4) ? ob.hi[2] : ob.lo[2])));
                }
                pad_box(b, RT_SPHERE_PAD_REL);
            }
        } else {
            f3 Q = mk(p.a[0], p.a[1], p.a[2]), u = mk(p.b[0], p.b[1], p.b[2]), v = mk(p.c[0], p.c[1], p.c[2]);
            quad_slot(Q, u, v, g);
            if (!linear) {
                Box3 &b = boxes[i];
                b.reset();
                b.add(to_world(Q));
                b.add(to_world(Q + u));
                b.add(to_world(Q + v));
                b.add(to_world(Q + u + v));
                pad_box(b);
            }
        }
    }

    // Binary tree over a set of boxes: the external (device-side) builder where one is given and it
    // delivers a tree of the expected shape, else the host's binned-SAH builder.
    bool any_external = false, all_external = true;
    auto build_binary = [&](const std::vector<Box3> &bx, const std::vector<int> &ty) {
        const int cnt = (int)bx.size();
        BvhOut bo;
        bool built = false;
        if (external) {
            built = external(bx, ty, bo) && (int)bo.leaf_order.size() == cnt &&
                    (int)bo.nodes.size() == 4 * (cnt - 1) && bo.max_inner_depth <= kMaxInnerDepth;
            if (!built) bo = BvhOut();
        }
        if (!built) {
            BvhBuilder builder(bx, ty);
            bo = builder.run();
        }
        any_external = any_external || built;
        all_external = all_external && built;
        return bo;
    };

    out.n_prims = n;
    out.slot_f4 = slot_f4;
    out.linear = linear;
    for (int i = 0; i < n; i++) (types[i] == RT_PRIM_SPHERE ? out.n_spheres : out.n_quads)++;

    std::vector<f4> &slots = out.slots;
    std::vector<i4> meta;
    std::vector<f4> &nodes = out.nodes;
    meta.resize(n);
    if (two_level) {
        // ---- two-level tree: one tree per distinct sub-scene (in its own space), one over the loose
        // primitives and the instances.  Slots: the loose primitives in the top tree's leaf order,
        // then every sub-scene's in its own tree's leaf order; an instance is the leaf "slot"
        // n + k of the top tree, and slot n + n_instances is the mark that ends a visit to one.
        out.any_chain = true;
        std::vector<std::pair<int, int>> t_range; // distinct sub-scenes
        std::vector<int> inst_template(d->n_instances);
        for (int k = 0; k < d->n_instances; k++) {
            const std::pair<int, int> r{d->instances[k].first_prim, d->instances[k].n_prims};
            size_t t = 0;
            while (t < t_range.size() && t_range[t] != r) t++;
            if (t == t_range.size()) t_range.push_back(r);
            inst_template[k] = (int)t;
        }
        std::vector<char> in_template(n, 0);
        for (const auto &r : t_range)
            for (int i = r.first; i < r.first + r.second; i++) in_template[i] = 1;
        struct Sub {
            Wide wide;
            std::vector<int> leaf_order; // local slot -> primitive
            Box3 box;
            int root_ref = 0, node_base = 0, slot_base = 0;
        };
        std::vector<Sub> subs(t_range.size());
        int max_sub_levels = 0;
        for (size_t t = 0; t < t_range.size(); t++) {
            const int first = t_range[t].first, cnt = t_range[t].second;
            std::vector<Box3> sb(boxes.begin() + first, boxes.begin() + first + cnt);
            std::vector<int> st(types.begin() + first, types.begin() + first + cnt);
            Sub &sub = subs[t];
            sub.box.reset();
            for (const Box3 &b : sb) sub.box.add(b);
            if (cnt == 1) {
                sub.leaf_order = {first};
            } else {
                BvhOut bo = build_binary(sb, st);
                sub.wide = collapse_to_wide(bo.nodes);
                for (int local : bo.leaf_order) sub.leaf_order.push_back(first + local);
                max_sub_levels = std::max(max_sub_levels, sub.wide.levels);
            }
        }
        // the instances' boxes in the world: the sub-scene's box pushed through the instance's wrappers
        std::vector<Box3> inst_boxes(d->n_instances);
        for (int k = 0; k < d->n_instances; k++) {
            const Sub &sub = subs[inst_template[k]];
            Box3 &b = inst_boxes[k];
            b.reset();
            const int c = d->instances[k].xform;
            const int first = c >= 0 ? d->chains[c].first_op : 0, last = c >= 0 ? first + d->chains[c].n_ops : 0;
            for (int corner = 0; corner < 8; corner++) {
                f3 q = mk((corner & 1) ? sub.box.hi[0] : sub.box.lo[0], (corner & 2) ? sub.box.hi[1] : sub.box.lo[1],
                          (corner & 4) ? sub.box.hi[2] : sub.box.lo[2]);
                for (int o = first; o < last; o++) q = op_point(ops[o], q);
                b.add(q);
            }
            pad_box(b);
        }
        // loose primitives whose box takes up most of the world's (loose primitives and instances
        // together) are tested up front (as in the one-level case below) and become the first slots
        std::vector<int> big;
        {
            Box3 all;
            all.reset();
            for (int i = 0; i < n; i++)
                if (!in_template[i]) all.add(boxes[i]);
            for (const Box3 &b : inst_boxes) all.add(b);
            const float limit = kBigPrimArea * all.area();
            for (int i = 0; i < n; i++)
                if (!in_template[i] && (int)big.size() < kBigPrimMax && boxes[i].area() >= limit) big.push_back(i);
            if (getenv("RT_AMD_NO_BIG")) big.clear();
            for (int i : big) in_template[i] = 2; // (neither in a sub-scene nor an item of the top tree)
        }
        // items of the top tree: the other loose primitives, then the instances
        std::vector<Box3> ib;
        std::vector<int> it, item_prim;
        for (int i = 0; i < n; i++)
            if (!in_template[i]) {
                ib.push_back(boxes[i]);
                it.push_back(types[i]);
                item_prim.push_back(i);
            }
        const int n_loose = (int)item_prim.size();
        for (int k = 0; k < d->n_instances; k++) {
            ib.push_back(inst_boxes[k]);
            it.push_back(RT_PRIM_SPHERE);
        }
        const int n_items = (int)ib.size();
        std::vector<int> top_order; // top slot -> item
        Wide top;
        if (n_items == 1) {
            top_order = {0};
            Box3 b = ib[0];
            single_child_node(top.nodes, b, ~(it[0] == RT_PRIM_QUAD ? 1 : 0));
            top.n_nodes = 1;
            top.levels = 1;
        } else {
            BvhOut bo = build_binary(ib, it);
            top = collapse_to_wide(bo.nodes);
            top_order = bo.leaf_order;
        }
        // final slots
        std::vector<int> item_slot(n_items, -1), slot_prim;
        out.n_big = (int)big.size();
        out.big_quads = 0;
        for (size_t j = 0; j < big.size(); j++) {
            slot_prim.push_back(big[j]);
            if (types[big[j]] == RT_PRIM_QUAD) out.big_quads |= 1 << j;
        }
        for (int item : top_order)
            if (item < n_loose) {
                item_slot[item] = (int)slot_prim.size();
                slot_prim.push_back(item_prim[item]);
            }
        for (int k = 0; k < d->n_instances; k++) item_slot[n_loose + k] = n + k;
        int node_total = top.n_nodes;
        for (Sub &sub : subs) {
            sub.slot_base = (int)slot_prim.size();
            slot_prim.insert(slot_prim.end(), sub.leaf_order.begin(), sub.leaf_order.end());
            sub.node_base = node_total;
            node_total += sub.wide.n_nodes;
        }
        nodes = std::move(top.nodes);
        relocate_wide(nodes, 0, (size_t)top.n_nodes, 0, [&](int s0) { return item_slot[top_order[s0]]; });
        for (size_t t = 0; t < subs.size(); t++) {
            Sub &sub = subs[t];
            const int cnt = t_range[t].second;
            if (cnt == 1) {
                sub.root_ref = ~(sub.slot_base * 2 + (types[t_range[t].first] == RT_PRIM_QUAD ? 1 : 0));
                continue;
            }
            const size_t at = nodes.size() / 8;
            nodes.insert(nodes.end(), sub.wide.nodes.begin(), sub.wide.nodes.end());
            relocate_wide(nodes, at, (size_t)sub.wide.n_nodes, sub.node_base, [&](int s0) { return sub.slot_base + s0; });
            sub.root_ref = sub.node_base;
        }
        nodes_to_plane_layout(nodes);
        slots.resize((size_t)n * slot_f4);
        for (int sl = 0; sl < n; sl++) {
            const int pi = slot_prim[sl];
            memcpy(&slots[(size_t)sl * slot_f4], &geom[(size_t)pi * slot_f4], sizeof(f4) * slot_f4);
            meta[sl] = i4{types[pi], d->prims[pi].material, d->prims[pi].xform, pi};
        }
        // ids that order all primitives of the world: a loose one keeps its index, instance k's
        // follow behind those of the instances before it
        int next_id = n;
        out.inst.resize(d->n_instances);
        for (int k = 0; k < d->n_instances; k++) {
            out.inst[k] = i4{subs[inst_template[k]].root_ref, d->instances[k].xform, next_id - d->instances[k].first_prim, 0};
            next_id += d->instances[k].n_prims;
        }
        out.n_nodes = node_total;
        out.bvh_depth = top.levels + max_sub_levels + 1; // (+1: the mark on the stack while inside an instance)
        out.n_top = 0;
        out.external_nodes = any_external && all_external; // every tree of the scene came from the external builder
    } else if (linear) {
        slots = geom;
        for (int i = 0; i < n; i++) meta[i] = i4{types[i], d->prims[i].material, d->prims[i].xform, i};
    } else {
        // Primitives whose box takes up most of the scene's (a ground sphere: every ray enters its box)
        // stay out of the tree: they become the first slots and every ray tests them before it starts
        // its walk (rt_core.h trav_test_big).
        std::vector<int> big, rest;
        {
            Box3 all;
            all.reset();
            for (const Box3 &b : boxes) all.add(b);
            const float limit = kBigPrimArea * all.area();
            for (int i = 0; i < n; i++)
                if ((int)big.size() < kBigPrimMax && boxes[i].area() >= limit) big.push_back(i);
            if (n - (int)big.size() < 2 || getenv("RT_AMD_NO_BIG")) big.clear();
            std::vector<char> is_big(n, 0);
            for (int i : big) is_big[i] = 1;
            for (int i = 0; i < n; i++)
                if (!is_big[i]) rest.push_back(i);
        }
        const int n_big = (int)big.size(), n_rest = (int)rest.size();
        std::vector<Box3> rest_boxes_store;
        std::vector<int> rest_types_store;
        const std::vector<Box3> *tb = &boxes;
        const std::vector<int> *tt = &types;
        if (n_big > 0) {
            rest_boxes_store.reserve(n_rest);
            rest_types_store.reserve(n_rest);
            for (int i : rest) {
                rest_boxes_store.push_back(boxes[i]);
                rest_types_store.push_back(types[i]);
            }
            tb = &rest_boxes_store;
            tt = &rest_types_store;
        }
        BvhOut bo = build_binary(*tb, *tt);
        out.external_nodes = any_external;
        out.n_bin_nodes = (int)(bo.nodes.size() / 4);
        Wide wide = collapse_to_wide(bo.nodes);
        nodes = std::move(wide.nodes);
        // trees too large for LDS: their top is cached there (kTreeTopMin nodes is more than a CU's LDS holds)
        out.n_top = wide.n_nodes >= kTreeTopMin ? reorder_tree_top(nodes, kTreeTopMax) : 0;
        if (n_big > 0) relocate_wide(nodes, 0, (size_t)wide.n_nodes, 0, [&](int s0) { return n_big + s0; });
        nodes_to_plane_layout(nodes);
        slots.resize((size_t)n * slot_f4);
        out.n_big = n_big;
        out.big_quads = 0;
        for (int sl = 0; sl < n; sl++) {
            int pi = sl < n_big ? big[sl] : rest[bo.leaf_order[sl - n_big]];
            if (sl < n_big && types[pi] == RT_PRIM_QUAD) out.big_quads |= 1 << sl;
            memcpy(&slots[(size_t)sl * slot_f4], &geom[(size_t)pi * slot_f4], sizeof(f4) * slot_f4);
            meta[sl] = i4{types[pi], d->prims[pi].material, d->prims[pi].xform, pi};
        }
        out.n_nodes = wide.n_nodes;
        out.bvh_depth = wide.levels;
    }

    // per-slot shading records: the material (and a solid colour) inlined, so that the
    // common case costs one 32-byte read instead of a chain of dependent table lookups
    {
        std::vector<int> tex_map(d->n_textures, -1);
        std::vector<rt_texture> &tex_out = out.textures;
        // depth-first copy of a texture and what it refers to; returns its new index
        struct Copier {
            const rt_scene_desc *d;
            std::vector<int> &map;
            std::vector<rt_texture> &outv;
            int visit(int t) {
                if (map[t] >= 0) return map[t];
                int id = (int)outv.size();
                map[t] = id;
                outv.push_back(d->textures[t]);
                if (d->textures[t].type == RT_TEX_CHECKER) {
                    int e = visit(d->textures[t].even), o = visit(d->textures[t].odd);
                    outv[id].even = e;
                    outv[id].odd = o;
                }
                return id;
            }
        } copier{d, tex_map, tex_out};
        out.rec_i.resize(n);
        out.rec_f.resize(n);
        for (int sl = 0; sl < n; sl++) {
            const i4 m = meta[sl]; // {type, material, chain, prim id}
            const rt_material &mat = d->materials[m.y];
            int tex_type = RT_TEX_SOLID, tex_id = -1;
            f4 rf{0.0f, 0.0f, 0.0f, 0.0f};
            if (mat.type == RT_MAT_LAMBERTIAN || mat.type == RT_MAT_DIFFUSE_LIGHT) {
                const rt_texture &t = d->textures[mat.texture];
                if (t.type == RT_TEX_SOLID) {
                    rf = f4{t.color[0], t.color[1], t.color[2], 0.0f};
                } else {
                    tex_type = t.type;
                    tex_id = copier.visit(mat.texture);
                }
            } else if (mat.type == RT_MAT_METAL) {
                rf = f4{mat.albedo[0], mat.albedo[1], mat.albedo[2], mat.param};
            } else {
                // dielectric (attenuation (1,1,1), material.h:74): what scatter() derives from the
                // refraction index alone, computed once here in the same fp32 operations
                // (material.h:76 1/ri; material.h:99-100 Schlick's r0 for either side)
                const float ior = mat.param, inv = 1.0f / ior;
                float r0_front = (1.0f - inv) / (1.0f + inv), r0_back = (1.0f - ior) / (1.0f + ior);
                r0_front = r0_front * r0_front;
                r0_back = r0_back * r0_back;
                rf = f4{inv, r0_front, r0_back, ior};
            }
            out.rec_i[sl] = i4{m.x | (mat.type << 2) | (tex_type << 4), tex_id, m.z, m.w};
            if (linear) {
                if (sl == 0) out.scan.resize((size_t)n * RT_SCAN_F4);
                scan_record(m.x, m.z, &slots[(size_t)sl * slot_f4], &out.scan[(size_t)sl * RT_SCAN_F4]);
            }
            out.rec_f[sl] = rf;
        }
    }

    // wrapper ops, re-packed chain by chain so that chain c covers ops [chain_first[c], chain_first[c+1])
    std::vector<f4> &xf_packed = out.xf;
    std::vector<int> &chain_first = out.chain_first;
    chain_first.assign(d->n_chains + 1, 0);
    for (int c = 0; c < d->n_chains; c++) {
        chain_first[c] = (int)(xf_packed.size() / RT_XF_F4);
        for (int k = 0; k < d->chains[c].n_ops; k++) {
            const HostOp &h = ops[d->chains[c].first_op + k];
            const float *cw = h.cw.m, *cc = h.ccw.m;
            xf_packed.push_back(f4{__builtin_bit_cast(float, h.type), h.p.x, h.p.y, h.p.z});
            xf_packed.push_back(h.a);
            xf_packed.push_back(f4{cw[0], cw[1], cw[2], cw[3]});
            xf_packed.push_back(f4{cw[4], cw[5], cw[6], cw[7]});
            xf_packed.push_back(f4{cw[8], cc[0], cc[1], cc[2]});
            xf_packed.push_back(f4{cc[3], cc[4], cc[5], cc[6]});
            xf_packed.push_back(f4{cc[7], cc[8], 0.0f, 0.0f});
        }
    }
    chain_first[d->n_chains] = (int)(xf_packed.size() / RT_XF_F4);

    // Perlin tables in the kernels' layout
    out.perlins.resize((size_t)d->n_perlins);
    for (int i = 0; i < d->n_perlins; i++) {
        for (int k = 0; k < 256; k++) {
            const float *rv = d->perlins[i].randvec[k];
            out.perlins[i].randvec[k] = f4{rv[0], rv[1], rv[2], 0.0f};
            out.perlins[i].perm_x[k] = d->perlins[i].perm_x[k];
        }
    }

    return RT_OK;
}

// camera::initialize (camera.h:76-111) in the reference's fp32 operation order.
inline void camera_frame(const rt_camera_desc *c, rt_camera_frame *f) {
    memset(f, 0, sizeof(*f));
    f->image_width = c->image_width;
    int image_height = int(c->image_width / c->aspect_ratio);
    image_height = (image_height < 1) ? 1 : image_height;
    f->image_height = image_height;
    f->pixel_samples_scale = 1.0f / c->samples_per_pixel;
    f3 lookfrom = mk(c->lookfrom[0], c->lookfrom[1], c->lookfrom[2]);
    f3 lookat = mk(c->lookat[0], c->lookat[1], c->lookat[2]);
    f3 vup = mk(c->vup[0], c->vup[1], c->vup[2]);
    f3 center = lookfrom;
    float theta = c->vfov * RT_PI / 180.0f;
    float h = tanf(theta / 2.0f);
    float viewport_height = 2.0f * h * c->focus_dist;
    float viewport_width = viewport_height * (float(c->image_width) / image_height);
    f3 w = rtc::unit(lookfrom - lookat);
    f3 u = rtc::unit(rtc::cross(vup, w));
    f3 v = rtc::cross(w, u);
    f3 viewport_u = viewport_width * u;
    f3 viewport_v = viewport_height * -v;
    f3 pdu = viewport_u / float(c->image_width);
    f3 pdv = viewport_v / float(image_height);
    f3 upper_left = center - (c->focus_dist * w) - viewport_u / 2.0f - viewport_v / 2.0f;
    f3 p00 = upper_left + 0.5f * (pdu + pdv);
    float defocus_radius = c->focus_dist * tanf((c->defocus_angle / 2.0f) * RT_PI / 180.0f);
    f3 ddu = u * defocus_radius;
    f3 ddv = v * defocus_radius;
    auto put = [](float *dst, f3 s) {
        dst[0] = s.x;
        dst[1] = s.y;
        dst[2] = s.z;
    };
    put(f->center, center);
    put(f->pixel00_loc, p00);
    put(f->pixel_delta_u, pdu);
    put(f->pixel_delta_v, pdv);
    put(f->u, u);
    put(f->v, v);
    put(f->w, w);
    put(f->defocus_disk_u, ddu);
    put(f->defocus_disk_v, ddv);
}

} // namespace rtf
