// kernel_logic.cpp — TEST-ONLY host build of the kernel sources' logic.
//
// The container that authors this code has no GPU.  To check the scene
// flattening (rt_flatten.h), the BVH culling and the tie rules (rt_core.h)
// against the oracle BEFORE a GPU run, this file compiles the very same headers
// with g++ and runs their per-ray functions sequentially.  It is not a product
// path: nothing under raytracing_amd/ loads it, and it lives under tests/.
#include "rt_amd.h"
#include "rt_core.h"
#include "rt_flatten.h"

#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rtc;

struct SimScene {
    rtf::HostScene hs;
    std::vector<rt_perlin> perl;
    std::vector<rt_image> imgs;
    std::vector<uint8_t> texels;
    DevScene dev;
    std::string err;
};

extern "C" {

void *sim_scene_create(const rt_scene_desc *d, uint32_t flags) {
    SimScene *s = new SimScene();
    if (rtf::flatten_scene(d, flags, s->hs, s->err) != RT_OK) {
        delete s;
        return nullptr;
    }
    s->imgs.assign(d->images, d->images + d->n_images);
    s->texels.assign(d->texels, d->texels + d->n_texel_bytes);
    DevScene &dv = s->dev;
    memset(&dv, 0, sizeof(dv));
    dv.slots = s->hs.slots.data();
    dv.rec_i = s->hs.rec_i.data();
    dv.rec_f = s->hs.rec_f.data();
    dv.textures = s->hs.textures.data();
    dv.n_textures = (int)s->hs.textures.size();
    dv.nodes = s->hs.nodes.data();
    dv.xf = s->hs.xf.data();
    dv.chain_first = s->hs.chain_first.data();
    dv.scan = s->hs.scan.data();
    dv.perlins = s->hs.perlins.data();
    dv.n_perlins = (int)s->hs.perlins.size();
    dv.images = s->imgs.data();
    dv.texels = s->texels.data();
    dv.n_slots = s->hs.n_prims;
    dv.inst = s->hs.inst.data();
    dv.n_inst = (int)s->hs.inst.size();
    dv.n_big = s->hs.n_big;
    dv.big_quads = s->hs.big_quads;
    dv.n_nodes = s->hs.n_nodes;
    dv.slot_f4 = s->hs.slot_f4;
    dv.traversal = s->hs.linear ? TRAV_LINEAR : TRAV_BVH;
    dv.n_chains = d->n_chains;
    dv.n_xf_ops = (int)(s->hs.xf.size() / RT_XF_F4);
    dv.stack_cap = RT_STACK_CAP(s->hs.bvh_depth);
    {   // as the library chooses it for a scene that stays in global memory
        int bits = 1;
        while ((1ll << bits) < std::max<long long>(dv.n_nodes, 2ll * (dv.n_slots + dv.n_inst + 1))) bits++;
        dv.ref_bits = std::max(16, std::min(32, bits + 1));
    }
    return s;
}
void sim_scene_destroy(void *s) { delete (SimScene *)s; }

// Which of the kernels' stack variants the traversals below use: 0 = 32-bit entries, all in one
// column; 1 = 16-bit entries (scenes small enough for them: the variant of scenes held in LDS);
// 2 = 32-bit entries of which only the first four stay in the first column (the spilling variant
// of deep trees read from HBM).
static int g_stack_mode = 0;
void sim_set_stack_mode(int mode) { g_stack_mode = mode; }

extern "C++" {
namespace {
struct SimStacks {
    std::vector<int> a, b;
    std::vector<short> h;
    TravStack<int> st32;
    TravStack<short> st16;
    explicit SimStacks(const DevScene &sc) : a(sc.stack_cap + 8), b(sc.stack_cap + 8), h(sc.stack_cap + 8) {
        st32.lds = a.data();
        st32.spill = b.data();
        st32.rt_stride = 1;
        st32.lds_end = g_stack_mode == 2 ? 4 : sc.stack_cap + 8;
        st16.lds = h.data();
        st16.spill = nullptr;
        st16.rt_stride = 1;
        st16.lds_end = sc.stack_cap + 8;
    }
};
template <bool COUNT> TraceHit sim_trace_bvh(DevScene sc, const SceneView &vw, const Ray &r, float tmin, float tmax,
                                             SimStacks &k, Cnt<COUNT> &cn) {
    if (g_stack_mode == 1 && sc.n_nodes < 32768 && sc.n_slots + sc.n_inst + 1 < 16384) {
        sc.ref_bits = 16;
        return trace_bvh<COUNT, TravStack<short>, false>(sc, vw, r, tmin, tmax, k.st16, cn);
    }
    if (g_stack_mode == 2) return trace_bvh<COUNT, TravStack<int>, true>(sc, vw, r, tmin, tmax, k.st32, cn);
    return trace_bvh<COUNT, TravStack<int>, false>(sc, vw, r, tmin, tmax, k.st32, cn);
}
} // namespace
} // extern "C++"
int sim_scene_nodes(void *s) { return ((SimScene *)s)->hs.n_nodes; }
int sim_scene_depth(void *s) { return ((SimScene *)s)->hs.bvh_depth; }

void sim_trace_batch(void *scene, const rt_ray *rays, long long n, float tmin, float tmax, rt_hit *hits,
                     unsigned long long *box_tests, unsigned long long *prim_tests) {
    SimScene *s = (SimScene *)scene;
    const DevScene &sc = s->dev;
    const SceneView vw = global_view(sc);
    SimStacks stacks(sc);
    Cnt<true> cn;
    for (long long k = 0; k < n; k++) {
        Ray r;
        r.o = mk(rays[k].orig[0], rays[k].orig[1], rays[k].orig[2]);
        r.d = mk(rays[k].dir[0], rays[k].dir[1], rays[k].dir[2]);
        r.tm = rays[k].tm;
        TraceHit h = sc.traversal == TRAV_LINEAR ? trace_linear<true>(sc, vw, r, tmin, tmax, cn)
                                                 : sim_trace_bvh<true>(sc, vw, r, tmin, tmax, stacks, cn);
        rt_hit out;
        memset(&out, 0, sizeof(out));
        out.prim = -1;
        if (h.slot >= 0) {
            Surf sf = surface_at(sc, vw, r, h);
            out.prim = sf.rec.w;
            out.front_face = sf.front ? 1 : 0;
            out.t = h.t;
            out.p[0] = sf.p.x; out.p[1] = sf.p.y; out.p[2] = sf.p.z;
            out.normal[0] = sf.n.x; out.normal[1] = sf.n.y; out.normal[2] = sf.n.z;
            if (RT_KIND_PRIM(sf.rec.x) == RT_PRIM_SPHERE) sphere_uv(sf.outward, out.u, out.v);
            else { out.u = sf.u; out.v = sf.v; }
        }
        hits[k] = out;
    }
    if (box_tests) *box_tests = cn.box;
    if (prim_tests) *prim_tests = (unsigned long long)cn.sph + cn.quad;
}

// Certificate that the closest hit does not depend on the schedule.  The decoupled kernels test a
// leaf an arbitrary number of node steps after they have popped it, so the best_t a box is culled
// against varies from launch to launch between "nothing found yet" and "everything found".  For
// each ray: walk the tree without any culling by best_t, remembering for every leaf the largest
// entry parameter on its path from the root; test every leaf reached; the winner (and whatever ties
// with it) must then survive the box test against its OWN parameter on every box of its path: no
// best_t a schedule can produce is smaller than that, so no schedule can cull it.
//   out[0]: rays whose winner some schedule could cull (with the slack of Trav::cull_t)
//   out[1]: the same count if boxes were culled against best_t itself (no slack)
//   out[2]: rays whose winner differs from the whole-ray traversal's
//   out[3]: leaves reached per ray without culling, summed
// One-level scenes only (returns -1 otherwise).
int sim_schedule_certificate(void *scene, const rt_ray *rays, long long n, float tmin, float tmax,
                             unsigned long long *out) {
    SimScene *s = (SimScene *)scene;
    const DevScene &sc = s->dev;
    if (sc.traversal != TRAV_BVH || sc.n_inst > 0) return -1;
    const SceneView vw = global_view(sc);
    SimStacks stacks(sc);
    Cnt<false> cn;
    out[0] = out[1] = out[2] = out[3] = 0;
    struct Leaf {
        int ref;
        float pme;
    };
    std::vector<Leaf> leaves;
    struct Item {
        int ref;
        float pme;
    };
    std::vector<Item> todo;
    for (long long k = 0; k < n; k++) {
        Ray r;
        r.o = mk(rays[k].orig[0], rays[k].orig[1], rays[k].orig[2]);
        r.d = mk(rays[k].dir[0], rays[k].dir[1], rays[k].dir[2]);
        r.tm = rays[k].tm;
        Trav tv;
        trav_begin(tv, r, tmax);
        trav_test_big<false>(tv, sc, vw, r, tmin, cn);
        const float up_front_t = tv.best_t; // (tested before the walk in every schedule)
        leaves.clear();
        todo.clear();
        todo.push_back(Item{0, -RT_INF});
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            if (it.ref < 0) {
                leaves.push_back(Leaf{it.ref, it.pme});
                continue;
            }
            const NodePlanes np = node_planes_global<false>(sc.nodes + (size_t)it.ref * RT_NODE_F4, tv.br);
            const float px[4] = {np.px.x, np.px.y, np.px.z, np.px.w}, qx[4] = {np.qx.x, np.qx.y, np.qx.z, np.qx.w};
            const float py[4] = {np.py.x, np.py.y, np.py.z, np.py.w}, qy[4] = {np.qy.x, np.qy.y, np.qy.z, np.qy.w};
            const float pz[4] = {np.pz.x, np.pz.y, np.pz.z, np.pz.w}, qz[4] = {np.qz.x, np.qz.y, np.qz.z, np.qz.w};
            const float rf[4] = {np.refs.x, np.refs.y, np.refs.z, np.refs.w};
            for (int c = 0; c < 4; c++) {
                float e;
                // (culled against the up-front primitives' hit only: every schedule has that one)
                if (box_hit_planes(px[c], py[c], pz[c], qx[c], qy[c], qz[c], tv.br, tmin,
                                   cull_bound(up_front_t), e))
                    todo.push_back(Item{__builtin_bit_cast(int, rf[c]), e > it.pme ? e : it.pme});
            }
        }
        out[3] += leaves.size();
        // every leaf on its own, then all of them in sequence (the tie rules make the order irrelevant)
        std::vector<float> own_t(leaves.size(), RT_INF);
        for (size_t j = 0; j < leaves.size(); j++) {
            Trav one = tv;
            const int before = one.best_slot;
            const float before_t = one.best_t;
            trav_test_leaf<false>(one, leaves[j].ref, sc, vw, r, tmin, cn);
            if (one.best_slot != before || one.best_t != before_t) own_t[j] = one.best_t;
            trav_test_leaf<false>(tv, leaves[j].ref, sc, vw, r, tmin, cn);
        }
        bool risky = false, risky_no_slack = false;
        for (size_t j = 0; j < leaves.size(); j++)
            if (own_t[j] == tv.best_t && tv.best_slot >= sc.n_big) {
                const float wide = 1.0f + 1.9073486328125e-6f; // (box_hit_planes' own widening)
                if (!(leaves[j].pme <= __builtin_fmaf(cull_bound(tv.best_t), wide, tv.br.e2))) {
                    risky = true;
                    if (getenv("SIM_CERT_DEBUG"))
                        fprintf(stderr, "ray %lld: t %.9g path-max entry %.9g bound %.9g |d| %.6g slot %d o (%g %g %g) d (%g %g %g)\n",
                                k, tv.best_t, leaves[j].pme, cull_bound(tv.best_t), sqrtf(tv.dd),
                                tv.best_slot, r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z);
                    if (getenv("SIM_CERT_DEBUG")) {
                        const f4 *g = vw.slots + slot_words(sc, tv.best_slot);
                        fprintf(stderr, "   quad %d  g0 (%g %g %g %g) g1 (%g %g %g %g)\n", (int)tv.best_is_quad, g[0].x, g[0].y, g[0].z, g[0].w, g[1].x, g[1].y, g[1].z, g[1].w);
                    }
                }
                if (!(leaves[j].pme <= __builtin_fmaf(tv.best_t, wide, tv.br.e2))) risky_no_slack = true;
            }
        out[0] += risky;
        out[1] += risky_no_slack;
        const TraceHit h = sim_trace_bvh<false>(sc, vw, r, tmin, tmax, stacks, cn);
        if (h.slot != tv.best_slot || (h.slot >= 0 && h.t != tv.best_t)) out[2]++;
    }
    return 0;
}

// Same per-path code as the megakernel, one path after the other.
void sim_render(void *scene, const rt_camera_desc *cam, int sample_begin, int sample_count, uint32_t seed,
                float *accum, unsigned long long *rays_out) {
    SimScene *s = (SimScene *)scene;
    const DevScene &sc = s->dev;
    DevCamera dc;
    rtf::camera_frame(cam, &dc.f);
    for (int k = 0; k < 3; k++) dc.background[k] = cam->background[k];
    dc.defocus_angle = cam->defocus_angle;
    dc.max_depth = cam->max_depth;
    const SceneView vw = global_view(sc);
    SimStacks stacks(sc);
    Cnt<false> cn;
    unsigned long long rays = 0;
    int npix = dc.f.image_width * dc.f.image_height;
    for (int pix = 0; pix < npix; pix++) {
        float r = 0, g = 0, b = 0;
        for (int sidx = 0; sidx < sample_count; sidx++) {
            PathState ps = path_start(dc, seed, (unsigned)pix, pix % dc.f.image_width, pix / dc.f.image_width,
                                      (unsigned)(sample_begin + sidx));
            unsigned n_rays = 0;
            for (;;) { // path_step (rt_core.h) with the stack variant under test
                if (ps.depth <= 0) break;
                n_rays++;
                TraceHit h = sc.traversal == TRAV_LINEAR ? trace_linear<false>(sc, vw, ps.ray, 0.001f, RT_INF, cn)
                                                         : sim_trace_bvh<false>(sc, vw, ps.ray, 0.001f, RT_INF, stacks, cn);
                if (!path_after_trace<false>(sc, vw, dc, ps, h, cn)) break;
            }
            rays += n_rays;
            r += ps.L.x; g += ps.L.y; b += ps.L.z;
        }
        accum[3 * pix] += r; accum[3 * pix + 1] += g; accum[3 * pix + 2] += b;
    }
    if (rays_out) *rays_out = rays;
}

} // extern "C"
