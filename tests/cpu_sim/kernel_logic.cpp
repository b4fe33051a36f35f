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
