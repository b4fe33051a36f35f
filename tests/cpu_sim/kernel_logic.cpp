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
    s->perl.assign(d->perlins, d->perlins + d->n_perlins);
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
    dv.perlins = s->perl.data();
    dv.images = s->imgs.data();
    dv.texels = s->texels.data();
    dv.n_slots = s->hs.n_prims;
    dv.n_nodes = s->hs.n_nodes;
    dv.slot_f4 = s->hs.slot_f4;
    dv.traversal = s->hs.linear ? TRAV_LINEAR : TRAV_BVH;
    dv.n_chains = d->n_chains;
    dv.n_xf_ops = (int)(s->hs.xf.size() / 3);
    return s;
}
void sim_scene_destroy(void *s) { delete (SimScene *)s; }
int sim_scene_nodes(void *s) { return ((SimScene *)s)->hs.n_nodes; }
int sim_scene_depth(void *s) { return ((SimScene *)s)->hs.bvh_depth; }

void sim_trace_batch(void *scene, const rt_ray *rays, long long n, float tmin, float tmax, rt_hit *hits,
                     unsigned long long *box_tests, unsigned long long *prim_tests) {
    SimScene *s = (SimScene *)scene;
    const DevScene &sc = s->dev;
    const SceneView vw = global_view(sc);
    int stack[RT_STACK_DEPTH + 8];
    Cnt<true> cn;
    for (long long k = 0; k < n; k++) {
        Ray r;
        r.o = mk(rays[k].orig[0], rays[k].orig[1], rays[k].orig[2]);
        r.d = mk(rays[k].dir[0], rays[k].dir[1], rays[k].dir[2]);
        r.tm = rays[k].tm;
        TraceHit h = sc.traversal == TRAV_LINEAR ? trace_linear<true>(sc, vw, r, tmin, tmax, cn)
                                                 : trace_bvh<true, int>(sc, vw, r, tmin, tmax, stack, 1, cn);
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
    int stack[RT_STACK_DEPTH + 8];
    Cnt<false> cn;
    unsigned long long rays = 0;
    int npix = dc.f.image_width * dc.f.image_height;
    for (int pix = 0; pix < npix; pix++) {
        float r = 0, g = 0, b = 0;
        for (int sidx = 0; sidx < sample_count; sidx++) {
            PathState ps = path_start(dc, seed, (unsigned)pix, pix % dc.f.image_width, pix / dc.f.image_width,
                                      (unsigned)(sample_begin + sidx));
            unsigned n_rays = 0;
            for (;;) {
                bool cont = sc.traversal == TRAV_LINEAR
                                ? path_step<TRAV_LINEAR, false, int>(sc, vw, dc, ps, stack, 1, n_rays, cn)
                                : path_step<TRAV_BVH, false, int>(sc, vw, dc, ps, stack, 1, n_rays, cn);
                if (!cont) break;
            }
            rays += n_rays;
            r += ps.L.x; g += ps.L.y; b += ps.L.z;
        }
        accum[3 * pix] += r; accum[3 * pix + 1] += g; accum[3 * pix + 2] += b;
    }
    if (rays_out) *rays_out = rays;
}

} // extern "C"
