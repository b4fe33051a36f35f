// rt_sched.h — the traversal half of the decoupled scheduling scheme, shared by the render
// kernel (render_mega_bvh), the batch query that uses the render kernel's traversal
// (trace_sched) and the wavefront variant's trace kernel (wf_trace).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_core.h"

#ifndef RT_INNER_UNROLL
#define RT_INNER_UNROLL 2
#endif

namespace rtc {

// Leaves a lane has popped but not tested yet (leaf references are negative; 0 = none).
#ifndef RT_PARKED
#define RT_PARKED 1
#endif
struct Parked {
    int a = 0, b = 0; // tested in this order
    __device__ __forceinline__ bool full() const { return RT_PARKED > 1 ? b != 0 : a != 0; }
    __device__ __forceinline__ void put(int ref) {
        if (RT_PARKED > 1 && a != 0)
            b = ref;
        else
            a = ref;
    }
    __device__ __forceinline__ int take() {
        const int r = a;
        a = RT_PARKED > 1 ? b : 0;
        b = 0;
        return r;
    }
};

// Lane states of the decoupled kernels.
enum { ST_IDLE = 0, ST_TRAV = 1, ST_READY = 2 };

struct SchedStats { // wave-level schedule statistics (instrumented variants only)
    unsigned long long t_iters = 0, t_lanes = 0, l_iters = 0, l_lanes = 0, same_node = 0;
    long long c_inner = 0, c_leaf = 0;
};

// Cooperative node fetch for scenes read from HBM.  Every lane of a wave walking such a tree is at a
// node of its own, and a node is 128 contiguous bytes.  Read lane by lane (seven 16-byte loads from
// a line no other lane touches) one node step costs the vector L1 7 x 64 look-ups, and that cache,
// not the VALUs or HBM, is what such a kernel waits for (measured: 0.8 L1 accesses per cycle and CU
// over the whole launch at a 98 % hit rate).  Here the wave reads the nodes together: in pass p the
// eight lanes 8g .. 8g+7 read the eight 16-byte pieces of the node that lane 8p + g is at, so
// that one load instruction covers eight whole lines, and the pieces go through a staging area in LDS
// (32 nodes at a time, two rounds) from which every lane picks the planes its ray enters and leaves
// through.  All 64 lanes take part in the reads, whether they walk or not.
//   stage: this wave's RT_COOP_F4_PER_WAVE 16-byte words of LDS
//   want : the lane is at an inner node (ref >= 0)
// Staging layout: node slot s (0..31) holds its eight pieces at words 8 s + ((k + s) & 7): the eight
// lanes that write one node hit eight different bank groups, and so do lanes of consecutive slots
// reading the same piece.
template <bool COUNT>
__device__ __forceinline__ NodePlanes coop_fetch_nodes(bool want, int ref, const BoxRay &b, const f4 *nodes, f4 *stage,
                                                       int lane) {
    const int g = lane >> 3, k = lane & 7;
    const int mine = want ? ref : -1;
    f4 piece[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int oref = __builtin_amdgcn_ds_bpermute((8 * p + g) << 2, mine);
        piece[p] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if (oref >= 0) piece[p] = *((const RT_AS_GLOBAL f4 *)nodes + ((size_t)(unsigned)oref * RT_NODE_F4 + (unsigned)k));
    }
    NodePlanes r;
    r.n_children = 0;
    RT_AS_LDS f4 *lds = (RT_AS_LDS f4 *)stage;
    // pieces of this lane's ray: near / far plane per axis (BoxRay::sx is 0 or 16: lo first or hi first)
    const int ix = b.sx >> 4, iy = 2 + (b.sy >> 4), iz = 4 + (b.sz >> 4);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // (LDS executes a wave's instructions in order; the fences keep the compiler from moving the
        // reads of one round above the writes of the same round or below those of the next)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int p = 4 * h + q, s = 8 * q + g;
            lds[8 * s + ((k + s) & 7)] = piece[p]; // (zeros for a lane that is at no node: nobody reads that slot)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (want && (lane >> 5) == h) {
            const int s = lane & 31;
            const RT_AS_LDS f4 *n = lds + 8 * s;
            r.px = n[(ix + s) & 7], r.qx = n[((ix ^ 1) + s) & 7];
            r.py = n[(iy + s) & 7], r.qy = n[((iy ^ 1) + s) & 7];
            r.pz = n[(iz + s) & 7], r.qz = n[((iz ^ 1) + s) & 7];
            r.refs = n[(6 + s) & 7];
            if constexpr (COUNT) r.n_children = __builtin_bit_cast(int, n[(7 + s) & 7].x);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return r;
}

// The traversal half of the decoupled scheme, shared by the render kernel and by the batch
// kernel that puts caller-supplied rays through the same code (trace_sched): runs node steps
// for the lanes in ST_TRAV, RT_INNER_UNROLL per scheduling decision, until `min_ready` lanes
// wait for something else (a finished traversal, or — if idle lanes can be given new work,
// `idle_counts` — nothing at all) or no lane traverses any more.  A leaf does not stop its
// lane: it is parked in the lane's `pend` and the traversal goes on; parked leaves are tested
// together once leaf_min lanes hold one, blocked_min lanes have run into a second leaf, or no
// lane has an inner node left.  Lanes whose traversal ends leave in ST_READY with the closest
// hit in tv.best_t / tv.best_slot.
// COOP (scenes read from HBM only): nodes are read by coop_fetch_nodes through `coop_stage`, the wave's
// staging area in LDS; the tree's top is then not kept in LDS (vw.n_top = 0).
template <bool LDS_SCENE, bool COUNT, class ST, bool SPILL, bool COOP = false>
__device__ __forceinline__ void wave_traverse(int &state, Trav &tv, Parked &pend, const Ray &ray, const DevScene &sc,
                                              const SceneView &vw, const ST &st, float tmin, int min_ready,
                                              int leaf_min, int blocked_min, bool idle_counts, Cnt<COUNT> &cn,
                                              SchedStats &ss, f4 *coop_stage = nullptr) {
    static_assert(!(COOP && LDS_SCENE), "the cooperative fetch is for nodes in global memory");
    long long c_mark = 0;
    // two-level scenes: leaves that enter or leave an instance are dealt with among the leaf tests,
    // and a lane does not walk on past one (rt_core.h trav_barrier)
    const bool inst_scene = RT_FEAT_INSTANCE && sc.n_inst > 0;
    for (;;) {
        RT_MARK("sched");
        unsigned long long tmask = __ballot(state == ST_TRAV);
        if (tmask == 0ull) break;
        unsigned long long rmask = __ballot(state == ST_READY || (state == ST_IDLE && idle_counts));
        if (__popcll(rmask) >= min_ready) break;
        if constexpr (COUNT) {
            unsigned long long im0 = __ballot(state == ST_TRAV && tv.ref >= 0);
            if (im0) {
                ss.t_iters++;
                ss.t_lanes += __popcll(im0);
                // lanes that would be served by one wave-uniform fetch of the first lane's node
                const int leader = __builtin_amdgcn_readlane(tv.ref, __builtin_ctzll(im0));
                ss.same_node += __popcll(__ballot(state == ST_TRAV && tv.ref == leader));
            }
            c_mark = clock64();
        }
        const bool trav = state == ST_TRAV;
        RT_MARK("node");
#pragma unroll
        for (int k = 0; k < RT_INNER_UNROLL; k++) {
            if constexpr (COOP) {
                const bool at_node = trav && tv.ref >= 0;
                const NodePlanes np = coop_fetch_nodes<COUNT>(at_node, tv.ref, tv.br, vw.nodes, coop_stage, (int)(threadIdx.x & 63));
                if (at_node) trav_inner_planes<COUNT, ST, SPILL>(tv, np, tmin, st, sc.ref_bits, cn);
            } else if (trav && tv.ref >= 0)
                trav_inner<COUNT, ST, !LDS_SCENE, SPILL, LDS_SCENE ? RT_NODE_LDS_F4 : RT_NODE_F4>(tv, vw.nodes, tmin, st, sc.ref_bits, cn, vw.top,
                                                                                                     vw.n_top);
            if (trav && tv.ref < 0 && tv.ref != RT_TRAV_DONE && !pend.full() &&
                !(inst_scene && trav_is_barrier(sc, tv.ref))) {
                pend.put(tv.ref);
                trav_pop<ST, SPILL>(tv, st, sc.ref_bits);
            }
        }
        if constexpr (COUNT) {
            const long long now = clock64();
            ss.c_inner += now - c_mark;
            c_mark = now;
        }
        RT_MARK("leaf");
        const bool blocked = trav && tv.ref < 0 && tv.ref != RT_TRAV_DONE; // another leaf, no slot left
        const bool has_pend = trav && pend.a != 0;
        const bool at_barrier = inst_scene && trav && trav_is_barrier(sc, tv.ref);
        unsigned long long pmask = __ballot(has_pend || at_barrier);
        if (pmask != 0ull) {
            unsigned long long imask = __ballot(trav && tv.ref >= 0);
            unsigned long long bmask = __ballot(blocked);
            if (__popcll(pmask) >= leaf_min || __popcll(bmask) >= blocked_min || imask == 0ull) {
                if constexpr (COUNT) {
                    ss.l_iters++;
                    ss.l_lanes += __popcll(pmask);
                }
                if (has_pend) trav_test_leaf<COUNT>(tv, pend.take(), sc, vw, ray, tmin, cn);
                if (at_barrier) { // (after the parked leaf, which belongs to the space the lane is still in)
                    trav_barrier<ST, SPILL>(tv, sc, ray, st);
                } else if (has_pend && blocked) {
                    pend.put(tv.ref);
                    trav_pop<ST, SPILL>(tv, st, sc.ref_bits);
                }
            }
        }
        if constexpr (COUNT) ss.c_leaf += clock64() - c_mark;
        RT_MARK("sched");
        if (trav && tv.ref == RT_TRAV_DONE && pend.a == 0) state = ST_READY;
    }
}

} // namespace rtc
