// Reference harness — TEST INFRASTRUCTURE, never shipped, never linked by the product.
//
// This translation unit #includes the UNMODIFIED reference sources from where
// they lie (-I/root/reference, see oracle/Makefile) and drives them to produce
// the golden fixtures under tests/golden/:
//   dump     <scene> <out.rtscene> <out.cam>      flattened scene + camera
//   rays     <scene> <n_paths> <out.rays>         ray batch with closest-hit records
//   render   <scene> <spp> <burn_in> <row0> <row1> <out.f32|-> [out.u8]
//                                                 fp32 radiance sums (+ write_color bytes)
//   bvhcheck <scene> <n_paths>                    bvh_node::hit vs linear scan of its leaves
//   refpng   <scene> <out.png>                    the reference's own camera::render, untouched
// Scene = bouncing_spheres | checkered_spheres | earth | perlin_spheres | quads |
//         simple_light | cornell_box  (the seven functions of main.cpp:14-288), or one of the
//         harness's own: rotated_in_bvh, synth60_list, synth60_bvh.
//
// Nothing from the reference is copied here: the file only calls into it.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

// Reach camera::initialize/get_ray/ray_color and the private members of the
// scene classes without editing the reference.
#define private public
#define protected public
#include "rt_constants.h"
#include "camera.h"
#include "hittable.h"
#include "hittable_list.h"
#include "material.h"
#include "sphere.h"
#include "quad.h"
#include "bvh.h"
#include "texture.h"
#undef private
#undef protected

#include "rt_amd.h"
#include "rt_scene_io.h"

static void harness_capture(camera &cam, const hittable &world);

// Every scene function of main.cpp ends in `cam.render(world, file_name);`
// (main.cpp:94,118,142,168,207,238,287).  Redirect that one call.
#define render(a, b) vfov += 0.0f; harness_capture(cam, a)
#define main reference_main_unused
#include "main.cpp"
#undef main
#undef render

// A scene of this harness, not of main.cpp: rotated objects INSIDE a bvh_node.  The constructors of
// rotate_y / rotate_xyz leave a box that is too small (hittable.h:123-126, 241-244: the running
// maximum is taken against the running minimum), and bvh_node::hit culls with it (bvh.h:78), so
// the reference clips such objects.  None of main.cpp's scenes does this; the records of this one
// pin the behaviour.  No random numbers.
static void rotated_in_bvh() {
    hittable_list world;
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f,
                                  make_shared<lambertian>(color(0.5f, 0.5f, 0.5f))));
    for (int a = -3; a <= 3; a++)
        for (int b = -3; b <= 3; b++) {
            if (a * a + b * b < 3) continue;
            point3 c(1.6f * a + 0.2f * (b & 1), 0.3f, 1.6f * b);
            if ((a + b) & 1)
                world.add(make_shared<sphere>(c, 0.3f, make_shared<metal>(color(0.8f, 0.7f, 0.6f), 0.1f)));
            else
                world.add(make_shared<sphere>(c, 0.3f, make_shared<lambertian>(color(0.2f + 0.1f * (a + 3), 0.3f, 0.8f - 0.1f * (b + 3)))));
        }
    auto white = make_shared<lambertian>(color(0.73f, 0.73f, 0.73f));
    shared_ptr<hittable> slab = box(point3(-1.5f, 0.0f, -0.4f), point3(1.5f, 1.2f, 0.4f), white);
    world.add(make_shared<rotate_y>(slab, 40.0f));
    shared_ptr<hittable> ball = make_shared<sphere>(point3(0.9f, 0.0f, 0.0f), 0.6f, make_shared<metal>(color(0.9f, 0.9f, 0.9f), 0.0f));
    world.add(make_shared<translate>(make_shared<rotate_xyz>(ball, 25.0f, 50.0f, 10.0f), vec3(-3.0f, 1.4f, 1.0f)));
    shared_ptr<hittable> post = box(point3(0.0f, 0.0f, 0.0f), point3(0.5f, 2.0f, 0.5f), make_shared<lambertian>(color(0.65f, 0.05f, 0.05f)));
    world.add(make_shared<translate>(make_shared<rotate_y>(post, -30.0f), vec3(2.5f, 0.0f, -1.5f)));
    world = hittable_list(make_shared<bvh_node>(world));

    camera cam;
    cam.aspect_ratio = 16.0f / 9.0f;
    cam.image_width = 240;
    cam.samples_per_pixel = 10;
    cam.max_depth = 50;
    cam.background = color(0.70f, 0.80f, 1.00f);
    cam.vfov = 35.0f;
    cam.lookfrom = point3(6.0f, 4.0f, 9.0f);
    cam.lookat = point3(0.0f, 0.8f, 0.0f);
    cam.vup = vec3(0.0f, 1.0f, 0.0f);
    cam.defocus_angle = 0.0f;
    harness_capture(cam, world);
}

// The synthetic grid scene of the 4K configuration (one sphere or quad per cell of a
// [-half, half)^2 grid + a ground sphere; the layout is keyed by splitmix64, twelve draws per cell
// in a fixed order, so it does not depend on the compiler's argument evaluation order), built from
// the reference's own classes: as a plain hittable_list (hittable_list.h:29-43 scans every
// primitive) and under one bvh_node (bvh.h:48-88).  The reference has no such scene; these two
// give its answers for it.  Camera: the 4K view at 480 x 270.
static void synth_grid(int half, bool under_bvh) {
    uint64_t state = 0x5EED1234ull;
    auto u01 = [&]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return float(z >> 40) * (1.0f / 16777216.0f);
    };
    hittable_list world;
    auto checker = make_shared<checker_texture>(0.32f, color(0.2f, 0.3f, 0.1f), color(0.9f, 0.9f, 0.9f));
    world.add(make_shared<sphere>(point3(0.0f, -10000.0f, 0.0f), 10000.0f, make_shared<lambertian>(checker)));
    for (int a = -half; a < half; a++)
        for (int b = -half; b < half; b++) {
            float kind = u01(), choose = u01(), jx = u01(), jz = u01(), rr = u01();
            float c0 = u01(), c1 = u01(), c2 = u01(), f = u01(), q0 = u01(), q1 = u01(), q2 = u01();
            float r = 0.1f + 0.15f * rr;
            point3 base(a + 0.9f * jx, 0.0f, b + 0.9f * jz);
            shared_ptr<material> mat;
            if (choose < 0.8f) mat = make_shared<lambertian>(color(c0 * c0, c1 * c1, c2 * c2));
            else if (choose < 0.95f) mat = make_shared<metal>(color(0.5f + 0.5f * c0, 0.5f + 0.5f * c1, 0.5f + 0.5f * c2), 0.5f * f);
            else if (choose < 0.99f) mat = make_shared<dielectric>(1.5f);
            else mat = make_shared<diffuse_light>(color(4.0f * c0, 4.0f * c1, 4.0f * c2));
            if (kind < 0.75f)
                world.add(make_shared<sphere>(base + vec3(0.0f, r, 0.0f), r, mat));
            else
                world.add(make_shared<quad>(base, vec3(0.4f * (q0 - 0.5f), 0.0f, 0.4f * (q1 - 0.5f) + 0.2f),
                                            vec3(0.1f * (q2 - 0.5f), 2.0f * r, 0.0f), mat));
        }
    if (under_bvh) world = hittable_list(make_shared<bvh_node>(world));

    camera cam;
    cam.aspect_ratio = 16.0f / 9.0f;
    cam.image_width = 480;
    cam.samples_per_pixel = 1;
    cam.max_depth = 50;
    cam.background = color(0.70f, 0.80f, 1.00f);
    cam.vfov = 40.0f;
    cam.lookfrom = point3(0.0f, 40.0f, 130.0f);
    cam.lookat = point3(0.0f, 0.0f, 0.0f);
    cam.vup = vec3(0.0f, 1.0f, 0.0f);
    cam.defocus_angle = 0.0f;
    harness_capture(cam, world);
}

// ---------------------------------------------------------------------------------------------
// argv state
static std::vector<std::string> g_args;
static std::string g_mode;

// rotate_* keep only sin/cos of their angles; the scenes use angles that are
// multiples of 1e-3 degree, so rounding the recovered value restores them exactly.
static float recover_degrees(float s, float c) {
    double deg = std::atan2((double)s, (double)c) * 180.0 / 3.14159265358979323846;
    return (float)(std::round(deg * 1000.0) / 1000.0);
}

// ---------------------------------------------------------------------------------------------
// Graph walk: collects leaf slots in the order the reference tests them.
struct LeafSlot {
    shared_ptr<hittable> *slot;        // where the graph holds the leaf
    std::vector<rt_xform_op> chain;    // wrappers around it, innermost first
    int group;                         // index into groups
};

struct Flattener {
    std::vector<LeafSlot> leaves;          // test order, deduplicated
    std::map<const hittable *, int> ids;   // leaf object -> id
    std::vector<rt_group> groups;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels;
    std::map<const void *, int> mat_ids, tex_ids, perlin_ids, image_ids;

    void open_group(int kind) {
        if (!groups.empty() && groups.back().kind == kind && kind == RT_GROUP_LIST) return;
        rt_group g = {kind, (int32_t)leaves.size(), 0, 0};
        groups.push_back(g);
    }

    void add_leaf(shared_ptr<hittable> *slot, const std::vector<rt_xform_op> &outer_first) {
        const hittable *obj = slot->get();
        if (ids.count(obj)) return; // span-1 bvh nodes hold the same leaf twice (bvh.h:62-63)
        ids[obj] = (int)leaves.size();
        LeafSlot ls;
        ls.slot = slot;
        ls.chain.assign(outer_first.rbegin(), outer_first.rend());
        ls.group = (int)groups.size() - 1;
        leaves.push_back(ls);
        groups.back().n_prims++;
    }

    void walk(shared_ptr<hittable> *slot, std::vector<rt_xform_op> &xf, bool in_bvh) {
        hittable *h = slot->get();
        if (auto *l = dynamic_cast<hittable_list *>(h)) {
            for (auto &o : l->objects) walk(&o, xf, in_bvh);
        } else if (auto *b = dynamic_cast<bvh_node *>(h)) {
            if (!in_bvh) open_group(RT_GROUP_BVH);
            walk(&b->left, xf, true);
            walk(&b->right, xf, true);
            if (!in_bvh) open_group(RT_GROUP_LIST);
        } else if (auto *t = dynamic_cast<translate *>(h)) {
            rt_xform_op op = {RT_XF_TRANSLATE, {t->offset.x(), t->offset.y(), t->offset.z()}};
            xf.push_back(op);
            walk(&t->object, xf, in_bvh);
            xf.pop_back();
        } else if (auto *s = dynamic_cast<scaling *>(h)) {
            rt_xform_op op = {RT_XF_SCALE, {s->scale.x(), s->scale.y(), s->scale.z()}};
            xf.push_back(op);
            walk(&s->object, xf, in_bvh);
            xf.pop_back();
        } else if (dynamic_cast<rotate_xyz *>(h) || dynamic_cast<rotate_y *>(h)) {
            // angles are not stored (only sin/cos); recover them in degrees
            rt_xform_op op;
            if (auto *r = dynamic_cast<rotate_xyz *>(h)) {
                op.type = RT_XF_ROTATE_XYZ;
                op.p[0] = recover_degrees(r->sin_alpha, r->cos_alpha);
                op.p[1] = recover_degrees(r->sin_beta, r->cos_beta);
                op.p[2] = recover_degrees(r->sin_gamma, r->cos_gamma);
                xf.push_back(op);
                walk(&r->object, xf, in_bvh);
            } else {
                auto *ry = dynamic_cast<rotate_y *>(h);
                op.type = RT_XF_ROTATE_Y;
                op.p[0] = recover_degrees(ry->sin_theta, ry->cos_theta);
                op.p[1] = op.p[2] = 0;
                xf.push_back(op);
                walk(&ry->object, xf, in_bvh);
            }
            xf.pop_back();
        } else {
            if (groups.empty()) open_group(RT_GROUP_LIST);
            add_leaf(slot, xf);
        }
    }

    int texture_id(const shared_ptr<texture> &tp) {
        const texture *t = tp.get();
        if (tex_ids.count(t)) return tex_ids[t];
        rt_texture d;
        memset(&d, 0, sizeof(d));
        d.even = d.odd = d.perlin = d.image = -1;
        if (auto *s = dynamic_cast<const solid_color *>(t)) {
            d.type = RT_TEX_SOLID;
            for (int k = 0; k < 3; k++) d.color[k] = s->albedo[k];
        } else if (auto *c = dynamic_cast<const checker_texture *>(t)) {
            d.type = RT_TEX_CHECKER;
            d.scale = c->inv_scale;
            d.even = texture_id(c->even);
            d.odd = texture_id(c->odd);
        } else if (auto *n = dynamic_cast<const noise_texture *>(t)) {
            d.type = RT_TEX_NOISE;
            d.scale = n->scale;
            rt_perlin p;
            for (int i = 0; i < 256; i++) {
                for (int k = 0; k < 3; k++) p.randvec[i][k] = n->noise.randvec[i][k];
                p.perm_x[i] = n->noise.perm_x[i];
            }
            d.perlin = (int)perlins.size();
            perlins.push_back(p);
        } else if (auto *im = dynamic_cast<const image_texture *>(t)) {
            d.type = RT_TEX_IMAGE;
            if (im->image.height() > 0) {
                rt_image ri = {im->image.image_width, im->image.image_height, (uint64_t)texels.size()};
                size_t nb = (size_t)ri.width * ri.height * 3;
                texels.insert(texels.end(), im->image.bdata, im->image.bdata + nb);
                d.image = (int)images.size();
                images.push_back(ri);
            }
        } else {
            fprintf(stderr, "harness: unknown texture class\n");
            exit(2);
        }
        int id = (int)textures.size();
        textures.push_back(d);
        tex_ids[t] = id;
        return id;
    }

    int material_id(const shared_ptr<material> &mp) {
        const material *m = mp.get();
        if (mat_ids.count(m)) return mat_ids[m];
        rt_material d;
        memset(&d, 0, sizeof(d));
        d.texture = -1;
        if (auto *l = dynamic_cast<const lambertian *>(m)) {
            d.type = RT_MAT_LAMBERTIAN;
            d.texture = texture_id(l->tex);
        } else if (auto *me = dynamic_cast<const metal *>(m)) {
            d.type = RT_MAT_METAL;
            for (int k = 0; k < 3; k++) d.albedo[k] = me->albedo[k];
            d.param = me->fuzz;
        } else if (auto *di = dynamic_cast<const dielectric *>(m)) {
            d.type = RT_MAT_DIELECTRIC;
            d.param = di->refraction_index;
        } else if (auto *li = dynamic_cast<const diffuse_light *>(m)) {
            d.type = RT_MAT_DIFFUSE_LIGHT;
            d.texture = texture_id(li->tex);
        } else {
            fprintf(stderr, "harness: unknown material class\n");
            exit(2);
        }
        int id = (int)materials.size();
        materials.push_back(d);
        mat_ids[m] = id;
        return id;
    }
};

// Leaf wrapper that remembers which primitive produced the last accepted hit.
// Inside hittable_list::hit and bvh_node::hit every accepted hit is at least as
// close as the previous one, so the last one accepted is the closest.
static int g_last_id = -1;
struct tagged_leaf : public hittable {
    shared_ptr<hittable> inner;
    int id;
    tagged_leaf(shared_ptr<hittable> o, int i) : inner(o), id(i) {}
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override {
        if (!inner->hit(r, ray_t, rec)) return false;
        g_last_id = id;
        return true;
    }
    aabb bounding_box() const override { return inner->bounding_box(); }
};

struct RayLog {
    std::vector<rt_ray> rays;
    std::vector<rt_hit> hits;
};

// Wraps the world: counts closest-hit queries and optionally logs them.
struct recording_world : public hittable {
    const hittable &inner;
    mutable uint64_t n_rays = 0;
    RayLog *log = nullptr;
    explicit recording_world(const hittable &w) : inner(w) {}
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override {
        n_rays++;
        g_last_id = -1;
        bool h = inner.hit(r, ray_t, rec);
        if (log) {
            rt_ray rr;
            rt_hit hh;
            memset(&hh, 0, sizeof(hh));
            for (int k = 0; k < 3; k++) {
                rr.orig[k] = r.origin()[k];
                rr.dir[k] = r.direction()[k];
            }
            rr.tm = r.time();
            hh.prim = -1;
            if (h) {
                hh.prim = g_last_id;
                hh.front_face = rec.front_face ? 1 : 0;
                hh.t = rec.t;
                for (int k = 0; k < 3; k++) {
                    hh.p[k] = rec.p[k];
                    hh.normal[k] = rec.normal[k];
                }
                hh.u = rec.u;
                hh.v = rec.v;
            }
            log->rays.push_back(rr);
            log->hits.push_back(hh);
        }
        return h;
    }
    aabb bounding_box() const override { return inner.bounding_box(); }
};

static void fill_camera_desc(const camera &cam, rt_camera_desc &cd, rt_camera_frame &cf) {
    memset(&cd, 0, sizeof(cd));
    memset(&cf, 0, sizeof(cf));
    cd.aspect_ratio = cam.aspect_ratio;
    cd.image_width = cam.image_width;
    cd.samples_per_pixel = cam.samples_per_pixel;
    cd.max_depth = cam.max_depth;
    cd.vfov = cam.vfov;
    cd.defocus_angle = cam.defocus_angle;
    cd.focus_dist = cam.focus_dist;
    cf.image_width = cam.image_width;
    cf.image_height = cam.image_height;
    cf.pixel_samples_scale = cam.pixel_samples_scale;
    for (int k = 0; k < 3; k++) {
        cd.background[k] = cam.background[k];
        cd.lookfrom[k] = cam.lookfrom[k];
        cd.lookat[k] = cam.lookat[k];
        cd.vup[k] = cam.vup[k];
        cf.center[k] = cam.center[k];
        cf.pixel00_loc[k] = cam.pixel00_loc[k];
        cf.pixel_delta_u[k] = cam.pixel_delta_u[k];
        cf.pixel_delta_v[k] = cam.pixel_delta_v[k];
        cf.u[k] = cam.u[k];
        cf.v[k] = cam.v[k];
        cf.w[k] = cam.w[k];
        cf.defocus_disk_u[k] = cam.defocus_disk_u[k];
        cf.defocus_disk_v[k] = cam.defocus_disk_v[k];
    }
}

static void write_file(const std::string &path, const void *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n) {
        fprintf(stderr, "harness: cannot write %s\n", path.c_str());
        exit(2);
    }
    fclose(f);
}

// A moving sphere keeps center_vec = center2 - center1 and the union box, not
// center2 itself.  Find a center2 (within a few ulps of center1 + center_vec)
// whose box reproduces the stored one exactly: that is all the BVH depends on.
static void recover_center2(const sphere &s, float out[3]) {
    vec3 rvec(s.radius, s.radius, s.radius);
    aabb box1(s.center1 - rvec, s.center1 + rvec);
    for (int k = 0; k < 3; k++) {
        float guess = s.center1[k] + s.center_vec[k];
        float cands[7] = {guess};
        float up = guess, dn = guess;
        for (int i = 0; i < 3; i++) {
            up = std::nextafter(up, infinity);
            dn = std::nextafter(dn, -infinity);
            cands[1 + 2 * i] = up;
            cands[2 + 2 * i] = dn;
        }
        bool found = false;
        for (float c : cands) {
            interval b2 = (c - s.radius <= c + s.radius) ? interval(c - s.radius, c + s.radius)
                                                         : interval(c + s.radius, c - s.radius);
            if (b2.size() < 0.0001f) b2 = b2.expand(0.0001f);
            interval un(box1.axis_interval(k), b2);
            const interval &have = s.bbox.axis_interval(k);
            if (un.min == have.min && un.max == have.max) {
                out[k] = c;
                found = true;
                break;
            }
        }
        if (!found) {
            fprintf(stderr, "harness: cannot recover center2\n");
            exit(2);
        }
    }
}

static void do_dump(camera &cam, Flattener &fl) {
    std::vector<rt_prim> prims;
    std::vector<rt_xform_op> ops;
    std::vector<rt_xform_chain> chains;
    std::map<std::string, int> chain_ids;
    for (auto &ls : fl.leaves) {
        rt_prim p;
        memset(&p, 0, sizeof(p));
        p.xform = -1;
        if (!ls.chain.empty()) {
            std::string key((const char *)ls.chain.data(), ls.chain.size() * sizeof(rt_xform_op));
            if (!chain_ids.count(key)) {
                rt_xform_chain c = {(int32_t)ops.size(), (int32_t)ls.chain.size()};
                ops.insert(ops.end(), ls.chain.begin(), ls.chain.end());
                chain_ids[key] = (int)chains.size();
                chains.push_back(c);
            }
            p.xform = chain_ids[key];
        }
        hittable *h = ls.slot->get();
        if (auto *s = dynamic_cast<sphere *>(h)) {
            p.type = RT_PRIM_SPHERE;
            p.material = fl.material_id(s->mat);
            p.flags = s->is_moving ? RT_PRIM_FLAG_MOVING : 0;
            for (int k = 0; k < 3; k++) {
                p.a[k] = s->center1[k];
                p.b[k] = s->is_moving ? s->center_vec[k] : 0.0f;
            }
            p.c[0] = s->radius;
            if (s->is_moving) recover_center2(*s, p.d);
        } else if (auto *q = dynamic_cast<quad *>(h)) {
            p.type = RT_PRIM_QUAD;
            p.material = fl.material_id(q->mat);
            for (int k = 0; k < 3; k++) {
                p.a[k] = q->Q[k];
                p.b[k] = q->u[k];
                p.c[k] = q->v[k];
            }
        } else {
            fprintf(stderr, "harness: unknown leaf class\n");
            exit(2);
        }
        prims.push_back(p);
    }
    std::vector<rt_group> groups;
    for (auto &g : fl.groups)
        if (g.n_prims > 0) groups.push_back(g);

    rt_scene_desc d;
    memset(&d, 0, sizeof(d));
    d.abi_version = RT_AMD_ABI_VERSION;
    d.n_prims = (int)prims.size();
    d.prims = prims.data();
    d.n_groups = (int)groups.size();
    d.groups = groups.data();
    d.n_materials = (int)fl.materials.size();
    d.materials = fl.materials.data();
    d.n_textures = (int)fl.textures.size();
    d.textures = fl.textures.data();
    d.n_perlins = (int)fl.perlins.size();
    d.perlins = fl.perlins.data();
    d.n_images = (int)fl.images.size();
    d.images = fl.images.data();
    d.n_texel_bytes = fl.texels.size();
    d.texels = fl.texels.data();
    d.n_xform_ops = (int)ops.size();
    d.xform_ops = ops.data();
    d.n_chains = (int)chains.size();
    d.chains = chains.data();
    if (rtio::save(&d, g_args[2].c_str()) != RT_OK) {
        fprintf(stderr, "harness: cannot write %s\n", g_args[2].c_str());
        exit(2);
    }
    cam.initialize();
    rt_camera_desc cd;
    rt_camera_frame cf;
    fill_camera_desc(cam, cd, cf);
    std::vector<char> blob(sizeof(cd) + sizeof(cf));
    memcpy(blob.data(), &cd, sizeof(cd));
    memcpy(blob.data() + sizeof(cd), &cf, sizeof(cf));
    write_file(g_args[3], blob.data(), blob.size());
    printf("{\"scene\":\"%s\",\"n_prims\":%d,\"n_groups\":%d,\"n_materials\":%d,\"n_textures\":%d,"
           "\"width\":%d,\"height\":%d}\n",
           g_args[1].c_str(), d.n_prims, d.n_groups, d.n_materials, d.n_textures, cf.image_width,
           cf.image_height);
}

// random_float() (rt_constants.h:28-33) keeps its generator in a function-local
// static, so the number of draws scene construction consumed cannot be read
// off.  Draw four numbers and find them in a fresh default-seeded mt19937:
// returns the stream position AFTER those four draws.
static long long locate_stream() {
    float probe[4];
    for (float &p : probe) p = random_float();
    std::mt19937 g;
    std::vector<float> win;
    for (long long pos = 0; pos < 200000000LL; pos++) {
        float r = float(g()) / 4294967296.0f;
        if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
        win.push_back(r);
        size_t n = win.size();
        if (n >= 4 && win[n - 4] == probe[0] && win[n - 3] == probe[1] && win[n - 2] == probe[2] &&
            win[n - 1] == probe[3])
            return pos + 1;
        if (n > 1024) win.erase(win.begin(), win.begin() + 512);
    }
    fprintf(stderr, "harness: cannot locate the random stream\n");
    exit(2);
}

// Shared sample loop: identical order to camera::render (camera.h:40-47).
static void sample_loop(camera &cam, const hittable &world, int spp, int row0, int row1,
                        std::vector<float> &sums) {
    sums.assign((size_t)cam.image_width * cam.image_height * 3, 0.0f);
    for (int j = row0; j < row1; j++) {
        for (int i = 0; i < cam.image_width; i++) {
            color pixel_color(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < spp; s++) {
                ray r = cam.get_ray(i, j);
                pixel_color += cam.ray_color(r, cam.max_depth, world);
            }
            size_t o = ((size_t)j * cam.image_width + i) * 3;
            sums[o] = pixel_color.x();
            sums[o + 1] = pixel_color.y();
            sums[o + 2] = pixel_color.z();
        }
    }
}

static void harness_capture(camera &cam, const hittable &world_const) {
    // the scene functions own `world`; we need to rewrite leaf slots for tagging
    hittable *root_raw = const_cast<hittable *>(&world_const);
    shared_ptr<hittable> root(root_raw, [](hittable *) {});

    if (g_mode == "refpng") {
        cam.render(world_const, g_args[2].c_str());
        return;
    }

    Flattener fl;
    std::vector<rt_xform_op> xf;
    fl.walk(&root, xf, false);

    if (g_mode == "dump") {
        do_dump(cam, fl);
        return;
    }

    if (g_mode == "render") {
        int spp = atoi(g_args[2].c_str());
        long long burn = atoll(g_args[3].c_str());
        cam.samples_per_pixel = spp;
        cam.initialize();
        int row0 = atoi(g_args[4].c_str());
        int row1 = atoi(g_args[5].c_str());
        if (row1 <= 0 || row1 > cam.image_height) row1 = cam.image_height;
        long long stream_pos = locate_stream();
        for (long long i = 0; i < burn; i++) (void)random_float();
        stream_pos += burn;
        recording_world rw(world_const);
        std::vector<float> sums;
        clock_t c0 = clock();
        sample_loop(cam, rw, spp, row0, row1, sums);
        double sec = double(clock() - c0) / CLOCKS_PER_SEC;
        if (g_args[6] != "-") write_file(g_args[6], sums.data(), sums.size() * sizeof(float));
        if (g_args.size() > 7) {
            std::vector<uint8_t> u8(sums.size());
            for (size_t px = 0; px < sums.size() / 3; px++) {
                int rgb[3];
                write_color(rgb, cam.pixel_samples_scale *
                                     color(sums[px * 3], sums[px * 3 + 1], sums[px * 3 + 2]));
                for (int k = 0; k < 3; k++) u8[px * 3 + k] = (uint8_t)rgb[k];
            }
            write_file(g_args[7], u8.data(), u8.size());
        }
        uint64_t paths = (uint64_t)cam.image_width * (row1 - row0) * spp;
        printf("{\"scene\":\"%s\",\"width\":%d,\"height\":%d,\"rows\":[%d,%d],\"spp\":%d,"
               "\"stream_pos\":%lld,\"paths\":%llu,\"rays\":%llu,\"seconds\":%.6f,\"mpaths_per_s\":%.6f,"
               "\"mrays_per_s\":%.6f}\n",
               g_args[1].c_str(), cam.image_width, cam.image_height, row0, row1, spp, stream_pos,
               (unsigned long long)paths, (unsigned long long)rw.n_rays, sec, paths / sec * 1e-6,
               rw.n_rays / sec * 1e-6);
        return;
    }

    // modes below need primitive ids
    {
        std::map<const hittable *, shared_ptr<hittable>> wrapped;
        std::vector<shared_ptr<hittable>> ordered(fl.leaves.size());
        for (auto &ls : fl.leaves) {
            const hittable *obj = ls.slot->get();
            int id = fl.ids[obj];
            wrapped[obj] = make_shared<tagged_leaf>(*ls.slot, id);
            ordered[id] = wrapped[obj];
        }
        struct Patcher {
            std::map<const hittable *, shared_ptr<hittable>> &w;
            void go(shared_ptr<hittable> *slot) {
                hittable *h = slot->get();
                if (auto *l = dynamic_cast<hittable_list *>(h)) {
                    for (auto &o : l->objects) go(&o);
                } else if (auto *b = dynamic_cast<bvh_node *>(h)) {
                    go(&b->left);
                    go(&b->right);
                } else if (auto *t = dynamic_cast<translate *>(h)) {
                    go(&t->object);
                } else if (auto *s = dynamic_cast<scaling *>(h)) {
                    go(&s->object);
                } else if (auto *r = dynamic_cast<rotate_xyz *>(h)) {
                    go(&r->object);
                } else if (auto *ry = dynamic_cast<rotate_y *>(h)) {
                    go(&ry->object);
                } else if (w.count(h)) {
                    *slot = w[h];
                }
            }
        } patch{wrapped};
        patch.go(&root);

        int n_paths = atoi(g_args[2].c_str());
        cam.initialize();
        recording_world rw(world_const);
        RayLog log;
        rw.log = &log;
        // spread the paths over the image on a regular lattice
        long long n_pix = (long long)cam.image_width * cam.image_height;
        for (int k = 0; k < n_paths; k++) {
            long long pix = (long long)((k + 0.5) * (double)n_pix / n_paths);
            int i = (int)(pix % cam.image_width), j = (int)(pix / cam.image_width);
            ray r = cam.get_ray(i, j);
            (void)cam.ray_color(r, cam.max_depth, rw);
        }
        if (g_mode == "rays") {
            uint64_t n = log.rays.size();
            float tmin = 0.001f, tmax = infinity; // camera.h:148
            FILE *f = fopen(g_args[3].c_str(), "wb");
            if (!f) exit(2);
            fwrite(&n, sizeof(n), 1, f);
            fwrite(&tmin, sizeof(float), 1, f);
            fwrite(&tmax, sizeof(float), 1, f);
            fwrite(log.rays.data(), sizeof(rt_ray), n, f);
            fwrite(log.hits.data(), sizeof(rt_hit), n, f);
            fclose(f);
            uint64_t nh = 0;
            for (auto &h : log.hits) nh += h.prim >= 0;
            printf("{\"scene\":\"%s\",\"paths\":%d,\"rays\":%llu,\"hits\":%llu}\n", g_args[1].c_str(),
                   n_paths, (unsigned long long)n, (unsigned long long)nh);
            return;
        }
        if (g_mode == "bvhcheck") {
            // linear scan over the same (tagged) leaves, in test order
            hittable_list linear;
            for (auto &o : ordered) linear.add(o);
            uint64_t n = log.rays.size(), flips = 0, idd = 0, td = 0;
            for (uint64_t k = 0; k < n; k++) {
                const rt_ray &rr = log.rays[k];
                ray r(point3(rr.orig[0], rr.orig[1], rr.orig[2]), vec3(rr.dir[0], rr.dir[1], rr.dir[2]),
                      rr.tm);
                hit_record rec;
                g_last_id = -1;
                bool h = linear.hit(r, interval(0.001f, infinity), rec);
                bool hb = log.hits[k].prim >= 0;
                if (h != hb)
                    flips++;
                else if (h) {
                    if (g_last_id != log.hits[k].prim) idd++;
                    if (rec.t != log.hits[k].t) td++;
                }
            }
            printf("{\"scene\":\"%s\",\"rays\":%llu,\"hit_flips\":%llu,\"id_diffs\":%llu,"
                   "\"t_diffs\":%llu}\n",
                   g_args[1].c_str(), (unsigned long long)n, (unsigned long long)flips,
                   (unsigned long long)idd, (unsigned long long)td);
            return;
        }
    }
    fprintf(stderr, "harness: unknown mode %s\n", g_mode.c_str());
    exit(2);
}

int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++) g_args.push_back(argv[i]);
    if (g_args.size() < 2) {
        fprintf(stderr, "usage: rt_ref <mode> <scene> ...\n");
        return 2;
    }
    g_mode = g_args[0];
    const std::string &s = g_args[1];
    if (g_mode == "image") { // rt_ref image <file> <out.rtimg>: the texels rtw_image (stb_image) produces
        if (g_args.size() < 3) return 2;
        rtw_image im;
        if (!im.load(s)) {
            fprintf(stderr, "harness: cannot load %s\n", s.c_str());
            return 1;
        }
        FILE *f = fopen(g_args[2].c_str(), "wb");
        if (!f) return 1;
        fprintf(f, "RTIMG1\n%d %d\n", im.image_width, im.image_height);
        fwrite(im.bdata, 1, (size_t)im.image_width * im.image_height * 3, f);
        fclose(f);
        printf("{\"width\": %d, \"height\": %d}\n", im.image_width, im.image_height);
        return 0;
    }
    if (s == "bouncing_spheres") bouncing_spheres();
    else if (s == "checkered_spheres") checkered_spheres();
    else if (s == "earth") earth();
    else if (s == "perlin_spheres") perlin_spheres();
    else if (s == "quads") quads();
    else if (s == "simple_light") simple_light();
    else if (s == "cornell_box") cornell_box();
    else if (s == "rotated_in_bvh") rotated_in_bvh();
    else if (s == "synth60_list") synth_grid(60, false);
    else if (s == "synth60_bvh") synth_grid(60, true);
    else {
        fprintf(stderr, "harness: unknown scene %s\n", s.c_str());
        return 2;
    }
    return 0;
}
