// scenes.cpp — the benchmark scenes, written against the drop-in scene API
// (raytracing_amd/dropin), rendered by librt_amd.so.
//
//   rt_scenes <scene> [--spp N] [--width W] [--depth D] [--out file.png] [--dump file.rtscene]
//                     [--hits file.rays out.hits] [--half N] [--bvh 1]   (the last two: synth1m)
//
// The seven scenes restate the reference's main.cpp:14-288 as data; `synth1m` is the
// 1 000 001-primitive stress scene (no counterpart in the reference).  Random numbers that
// the reference draws inside one argument list are drawn here one statement at a time, in
// the order the reference's toolchain (g++) evaluates them, so that the scenes are the same
// under any compiler; tests/test_dropin.py checks the result against the reference's dump.
#include "camera.h"
#include "bvh.h"
#include "hittable_list.h"
#include "material.h"
#include "quad.h"
#include "sphere.h"
#include "texture.h"

#include <cstring>

struct View {
    float aspect;
    int width, spp;
    color background;
    float vfov;
    point3 from, at;
    float defocus_angle, focus_dist;
};

static int g_spp = -1, g_width = -1, g_depth = 50;
static const char *g_out = "RayTracing.png";
static const char *g_rays_in = nullptr, *g_hits_out = nullptr;

// --hits <file.rays> <out>: instead of rendering, puts the rays of a .rays file (uint64 n, float
// tmin, tmax, n rt_ray records) through world.hit() on the host and writes n rt_hit records
// (prim = 0 for a hit: hit_record carries no primitive id).
static void hits_on_the_host(const hittable &world) {
    FILE *f = fopen(g_rays_in, "rb");
    uint64_t n = 0;
    float range[2];
    if (!f || fread(&n, 8, 1, f) != 1 || fread(range, 4, 2, f) != 2) {
        fprintf(stderr, "cannot read %s\n", g_rays_in);
        exit(1);
    }
    std::vector<rt_ray> rays(n);
    if (fread(rays.data(), sizeof(rt_ray), n, f) != n) exit(1);
    fclose(f);
    std::vector<rt_hit> hits(n);
    for (uint64_t i = 0; i < n; i++) {
        const rt_ray &q = rays[i];
        hit_record rec;
        rt_hit h;
        memset(&h, 0, sizeof(h));
        h.prim = -1;
        if (world.hit(ray(point3(q.orig[0], q.orig[1], q.orig[2]), vec3(q.dir[0], q.dir[1], q.dir[2]), q.tm),
                      interval(range[0], range[1]), rec)) {
            h.prim = 0;
            h.front_face = rec.front_face ? 1 : 0;
            h.t = rec.t;
            for (int k = 0; k < 3; k++) {
                h.p[k] = rec.p[k];
                h.normal[k] = rec.normal[k];
            }
            h.u = rec.u;
            h.v = rec.v;
        }
        hits[i] = h;
    }
    f = fopen(g_hits_out, "wb");
    if (!f || fwrite(hits.data(), sizeof(rt_hit), n, f) != n) exit(1);
    fclose(f);
}

static void shoot(const hittable &world, const View &v) {
    if (g_rays_in) {
        hits_on_the_host(world);
        return;
    }
    camera cam;
    cam.aspect_ratio = v.aspect;
    cam.image_width = g_width > 0 ? g_width : v.width;
    cam.samples_per_pixel = g_spp > 0 ? g_spp : v.spp;
    cam.max_depth = g_depth;
    cam.background = v.background;
    cam.vfov = v.vfov;
    cam.lookfrom = v.from;
    cam.lookat = v.at;
    cam.vup = vec3(0.0f, 1.0f, 0.0f);
    cam.defocus_angle = v.defocus_angle;
    cam.focus_dist = v.focus_dist;
    cam.render(world, g_out);
    const rt_render_stats &s = cam.last_stats;
    if (s.paths)
        fprintf(stderr, "%llu paths, %llu rays, render kernel %.3f ms (%.1f Mpaths/s, %.1f Mrays/s)\n",
                (unsigned long long)s.paths, (unsigned long long)s.rays, s.kernel_ms,
                s.paths / s.kernel_ms * 1e-3, s.rays / s.kernel_ms * 1e-3);
}

static const color kSky(0.7f, 0.8f, 1.0f), kBlack(0.0f, 0.0f, 0.0f);
static const float kWide = 16.0f / 9.0f;

static shared_ptr<texture> ground_checker() {
    return make_shared<checker_texture>(0.32f, color(0.2f, 0.3f, 0.1f), color(0.9f, 0.9f, 0.9f));
}

static void bouncing_spheres() {
    hittable_list world;
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f, make_shared<lambertian>(ground_checker())));
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            float choose_mat = random_float();
            float jz = random_float(); // the z jitter is drawn before the x jitter
            float jx = random_float();
            point3 center(a + 0.9f * jx, 0.2f, b + 0.9f * jz);
            if (!((center - point3(4.0f, 0.2f, 0.0f)).length() > 0.9f)) continue;
            if (choose_mat < 0.8f) {
                color second = color::random(); // right operand of the product first
                color first = color::random();
                auto mat = make_shared<lambertian>(first * second);
                point3 center2 = center + vec3(0.0f, random_float(0.0f, 0.5f), 0.0f);
                world.add(make_shared<sphere>(center, center2, 0.2f, mat));
            } else if (choose_mat < 0.95) {
                color albedo = color::random(0.5f, 1.0f);
                float fuzz = random_float(0.0f, 0.5f);
                world.add(make_shared<sphere>(center, 0.2f, make_shared<metal>(albedo, fuzz)));
            } else {
                world.add(make_shared<sphere>(center, 0.2f, make_shared<dielectric>(1.5f)));
            }
        }
    world.add(make_shared<sphere>(point3(0.0f, 1.0f, 0.0f), 1.0f, make_shared<dielectric>(1.5f)));
    world.add(make_shared<sphere>(point3(-4.0f, 1.0f, 0.0f), 1.0f, make_shared<lambertian>(color(0.4f, 0.2f, 0.1f))));
    world.add(make_shared<sphere>(point3(4.0f, 1.0f, 0.0f), 1.0f, make_shared<metal>(color(0.7f, 0.6f, 0.5f), 0.0f)));
    world = hittable_list(make_shared<bvh_node>(world));
    shoot(world, View{kWide, 400, 10, kSky, 20.0f, point3(13.0f, 2.0f, 3.0f), point3(0.0f, 0.0f, -1.0f), 0.6f, 10.0f});
}

static void checkered_spheres() {
    hittable_list world;
    auto checker = ground_checker();
    world.add(make_shared<sphere>(point3(0.0f, -10.0f, 0.0f), 10.0f, make_shared<lambertian>(checker)));
    world.add(make_shared<sphere>(point3(0.0f, 10.0f, 0.0f), 10.0f, make_shared<lambertian>(checker)));
    shoot(world, View{kWide, 400, 10, kSky, 20.0f, point3(13.0f, 2.0f, 3.0f), point3(0.0f, 0.0f, 0.0f), 0.0f, 10.0f});
}

static void earth() {
    auto surface = make_shared<lambertian>(make_shared<image_texture>("earthmap.jpg"));
    auto globe = make_shared<sphere>(point3(0.0f, 0.0f, 0.0f), 2.0f, surface);
    shoot(hittable_list(globe), View{kWide, 400, 100, kSky, 20.0f, point3(0.0f, 0.0f, 12.0f), point3(0.0f, 0.0f, 0.0f), 0.0f, 10.0f});
}

static void perlin_spheres() {
    hittable_list world;
    auto marble = make_shared<noise_texture>(4);
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f, make_shared<lambertian>(marble)));
    world.add(make_shared<sphere>(point3(0.0f, 2.0f, 0.0f), 2.0f, make_shared<lambertian>(marble)));
    shoot(world, View{kWide, 400, 100, kSky, 20.0f, point3(13.0f, 2.0f, 3.0f), point3(0.0f, 0.0f, 0.0f), 0.0f, 10.0f});
}

static void quads() {
    hittable_list world;
    const struct {
        point3 Q;
        vec3 u, v;
        color c;
    } q[5] = {{point3(-3, -2, 5), vec3(0, 0, -4), vec3(0, 4, 0), color(1.0f, 0.2f, 0.2f)},
              {point3(-2, -2, 0), vec3(4, 0, 0), vec3(0, 4, 0), color(0.2f, 1.0f, 0.2f)},
              {point3(3, -2, 1), vec3(0, 0, 4), vec3(0, 4, 0), color(0.2f, 0.2f, 1.0f)},
              {point3(-2, 3, 1), vec3(4, 0, 0), vec3(0, 0, 4), color(1.0f, 0.5f, 0.0f)},
              {point3(-2, -3, 5), vec3(4, 0, 0), vec3(0, 0, -4), color(0.2f, 0.8f, 0.8f)}};
    std::vector<shared_ptr<material>> mats;
    for (const auto &e : q) mats.push_back(make_shared<lambertian>(e.c)); // all materials first, as the reference
    for (int i = 0; i < 5; i++) world.add(make_shared<quad>(q[i].Q, q[i].u, q[i].v, mats[i]));
    shoot(world, View{1.0f, 400, 100, kSky, 80.0f, point3(0.0f, 0.0f, 9.0f), point3(0.0f, 0.0f, 0.0f), 0.0f, 10.0f});
}

static void simple_light() {
    hittable_list world;
    auto marble = make_shared<noise_texture>(4.0f);
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f, make_shared<lambertian>(marble)));
    world.add(make_shared<sphere>(point3(0.0f, 2.0f, 0.0f), 2.0f, make_shared<lambertian>(marble)));
    auto lamp = make_shared<diffuse_light>(color(4.0f, 4.0f, 4.0f));
    world.add(make_shared<quad>(point3(3.0f, 1.0f, -2.0f), vec3(2.0f, 0.0f, 0.0f), vec3(0.0f, 2.0f, 0.0f), lamp));
    world.add(make_shared<sphere>(point3(0.0f, 7.0f, 0.0f), 2.0f, lamp));
    shoot(world, View{kWide, 400, 100, kBlack, 20.0f, point3(26.0f, 3.0f, 6.0f), point3(0.0f, 2.0f, 0.0f), 0.0f, 10.0f});
}

static void cornell_box() {
    hittable_list world;
    auto red = make_shared<lambertian>(color(0.65f, 0.05f, 0.05f));
    auto white = make_shared<lambertian>(color(0.73, 0.73, 0.73));
    auto green = make_shared<lambertian>(color(0.12, 0.45, 0.15));
    auto lamp = make_shared<diffuse_light>(color(15.0f, 15.0f, 15.0f));
    const float s = 555.0f;
    world.add(make_shared<quad>(point3(s, 0, 0), vec3(0, s, 0), vec3(0, 0, s), green));
    world.add(make_shared<quad>(point3(0, 0, 0), vec3(0, s, 0), vec3(0, 0, s), red));
    world.add(make_shared<quad>(point3(343.0f, 554.0f, 332.0f), vec3(-130.0f, 0, 0), vec3(0, 0, -105.0f), lamp));
    world.add(make_shared<quad>(point3(0, 0, 0), vec3(s, 0, 0), vec3(0, 0, s), white));
    world.add(make_shared<quad>(point3(s, s, s), vec3(-s, 0, 0), vec3(0, 0, -s), white));
    world.add(make_shared<quad>(point3(0, 0, s), vec3(s, 0, 0), vec3(0, s, 0), white));
    shared_ptr<hittable> tall = box(point3(0, 0, 0), point3(165.0f, 330.0f, 165.0f), white);
    tall = make_shared<translate>(make_shared<rotate_xyz>(tall, 0.0f, 15.0f, 0.0f), vec3(265.0f, 0.0f, 295.0f));
    world.add(tall);
    shared_ptr<hittable> cube = box(point3(0, 0, 0), point3(165.0f, 165.0f, 165.0f), white);
    cube = make_shared<translate>(make_shared<rotate_xyz>(cube, -15.0f, -18.0f, 0.0f), vec3(130.0f, 0.0f, 65.0f));
    world.add(make_shared<scaling>(cube, vec3(0.5f, 0.5f, 0.5f)));
    shoot(world, View{1.0f, 600, 10, kBlack, 40.0f, point3(278.0f, 278.0f, -800.0f), point3(278.0f, 278.0f, 0.0f), 0.0f, 10.0f});
}

// Rotated objects inside a bvh_node (not one of main.cpp's scenes): the reference's rotate_y /
// rotate_xyz constructors leave a box that is too small (hittable.h:123-126, 241-244) and its
// bvh_node::hit culls with it (bvh.h:78).  The drop-in's host-side hit() reproduces that; the GPU
// renderer bounds the flattened primitives itself and shows the whole objects.
static void rotated_in_bvh() {
    hittable_list world;
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f, make_shared<lambertian>(color(0.5f, 0.5f, 0.5f))));
    for (int a = -3; a <= 3; a++)
        for (int b = -3; b <= 3; b++) {
            if (a * a + b * b < 3) continue;
            point3 c(1.6f * a + 0.2f * (b & 1), 0.3f, 1.6f * b);
            shared_ptr<material> mat;
            if ((a + b) & 1) mat = make_shared<metal>(color(0.8f, 0.7f, 0.6f), 0.1f);
            else mat = make_shared<lambertian>(color(0.2f + 0.1f * (a + 3), 0.3f, 0.8f - 0.1f * (b + 3)));
            world.add(make_shared<sphere>(c, 0.3f, mat));
        }
    shared_ptr<hittable> slab = box(point3(-1.5f, 0.0f, -0.4f), point3(1.5f, 1.2f, 0.4f), make_shared<lambertian>(color(0.73f, 0.73f, 0.73f)));
    world.add(make_shared<rotate_y>(slab, 40.0f));
    shared_ptr<hittable> ball = make_shared<sphere>(point3(0.9f, 0.0f, 0.0f), 0.6f, make_shared<metal>(color(0.9f, 0.9f, 0.9f), 0.0f));
    world.add(make_shared<translate>(make_shared<rotate_xyz>(ball, 25.0f, 50.0f, 10.0f), vec3(-3.0f, 1.4f, 1.0f)));
    shared_ptr<hittable> post = box(point3(0.0f, 0.0f, 0.0f), point3(0.5f, 2.0f, 0.5f), make_shared<lambertian>(color(0.65f, 0.05f, 0.05f)));
    world.add(make_shared<translate>(make_shared<rotate_y>(post, -30.0f), vec3(2.5f, 0.0f, -1.5f)));
    world = hittable_list(make_shared<bvh_node>(world));
    shoot(world, View{kWide, 240, 10, kSky, 35.0f, point3(6.0f, 4.0f, 9.0f), point3(0.0f, 0.8f, 0.0f), 0.0f, 10.0f});
}

// One shrub (60 spheres and 4 quads under a bvh_node) handed to 120 instance wrappers
// (hittable.h:51-344): the reference walks the one shrub through every wrapper; the drop-in records
// it once and places it with rt_instance records, so the library keeps 65 primitives, not 7 681
// (RTAMD_INSTANCES=0 writes every copy out, with the same image).  No counterpart in main.cpp.
static void grove() {
    uint64_t state = 0xC0FFEEull;
    auto u01 = [&]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return float(z >> 40) * (1.0f / 16777216.0f);
    };
    hittable_list shrub;
    auto leaf = make_shared<lambertian>(color(0.15f, 0.45f, 0.12f));
    auto berry = make_shared<metal>(color(0.8f, 0.2f, 0.2f), 0.2f);
    auto dew = make_shared<dielectric>(1.5f);
    for (int i = 0; i < 60; i++) {
        float x = 1.6f * (u01() - 0.5f), y = 0.2f + 1.4f * u01(), z = 1.6f * (u01() - 0.5f), r = 0.08f + 0.18f * u01();
        shared_ptr<material> m = i % 7 == 0 ? berry : (i % 11 == 0 ? dew : (shared_ptr<material>)leaf);
        shrub.add(make_shared<sphere>(point3(x, y, z), r, m));
    }
    auto bark = make_shared<lambertian>(color(0.35f, 0.25f, 0.15f));
    shrub.add(make_shared<quad>(point3(-0.05f, 0.0f, -0.05f), vec3(0.1f, 0.0f, 0.0f), vec3(0.0f, 0.9f, 0.0f), bark));
    shrub.add(make_shared<quad>(point3(-0.05f, 0.0f, 0.05f), vec3(0.1f, 0.0f, 0.0f), vec3(0.0f, 0.9f, 0.0f), bark));
    shrub.add(make_shared<quad>(point3(-0.05f, 0.0f, -0.05f), vec3(0.0f, 0.0f, 0.1f), vec3(0.0f, 0.9f, 0.0f), bark));
    shrub.add(make_shared<quad>(point3(0.05f, 0.0f, -0.05f), vec3(0.0f, 0.0f, 0.1f), vec3(0.0f, 0.9f, 0.0f), bark));
    shared_ptr<hittable> one = make_shared<bvh_node>(shrub);

    hittable_list world;
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f, make_shared<lambertian>(ground_checker())));
    for (int a = -6; a < 6; a++)
        for (int b = -5; b < 5; b++) {
            float jx = u01(), jz = u01(), angle = 360.0f * u01(), size = 0.7f + 0.8f * u01();
            shared_ptr<hittable> h = make_shared<rotate_y>(one, angle);
            if ((a + b) & 1) h = make_shared<scaling>(h, vec3(size, size * 1.2f, size));
            world.add(make_shared<translate>(h, vec3(3.0f * a + 1.5f * jx, 0.0f, 3.0f * b + 1.5f * jz)));
        }
    shoot(world, View{kWide, 640, 16, kSky, 35.0f, point3(4.0f, 9.0f, 26.0f), point3(0.0f, 1.0f, 0.0f), 0.0f, 10.0f});
}

// Sharing on two levels: a tree is a trunk, a dozen fruit and nine placements of one shrub (the
// shrub of `grove`); forty wrappers place the one tree.  The reference walks tree and shrub through
// both sets of wrappers; the drop-in records the shrub once and the tree's own primitives once, and
// places the shrub 360 times under the composed wrappers (RTAMD_INSTANCES=0: 24 000 primitives
// written out, same image).  No counterpart in main.cpp.
static void orchard() {
    uint64_t state = 0x0DDBA11ull;
    auto u01 = [&]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return float(z >> 40) * (1.0f / 16777216.0f);
    };
    hittable_list shrub;
    auto leaf = make_shared<lambertian>(color(0.15f, 0.45f, 0.12f));
    auto shine = make_shared<metal>(color(0.7f, 0.8f, 0.3f), 0.3f);
    for (int i = 0; i < 48; i++) {
        float x = 1.2f * (u01() - 0.5f), y = 1.2f * (u01() - 0.5f), z = 1.2f * (u01() - 0.5f), r = 0.08f + 0.16f * u01();
        shrub.add(make_shared<sphere>(point3(x, y, z), r, i % 9 == 0 ? (shared_ptr<material>)shine : leaf));
    }
    shared_ptr<hittable> one_shrub = make_shared<bvh_node>(shrub);

    hittable_list tree;
    auto bark = make_shared<lambertian>(color(0.35f, 0.25f, 0.15f));
    tree.add(box(point3(-0.15f, 0.0f, -0.15f), point3(0.15f, 2.2f, 0.15f), bark));
    auto fruit = make_shared<lambertian>(color(0.8f, 0.15f, 0.1f));
    auto glass = make_shared<dielectric>(1.5f);
    for (int i = 0; i < 12; i++) {
        float ang = 6.2831853f * u01(), rad = 0.5f + 0.9f * u01(), y = 1.4f + 1.6f * u01();
        tree.add(make_shared<sphere>(point3(rad * std::cos(ang), y, rad * std::sin(ang)), 0.09f, i % 4 == 0 ? (shared_ptr<material>)glass : fruit));
    }
    for (int i = 0; i < 9; i++) {
        float ang = 6.2831853f * (i + 0.5f * u01()) / 9.0f, rad = i == 0 ? 0.0f : 0.9f, y = i == 0 ? 3.0f : 1.9f + 0.9f * u01();
        float size = 0.8f + 0.5f * u01();
        shared_ptr<hittable> h = make_shared<rotate_xyz>(one_shrub, 40.0f * u01(), 360.0f * u01(), 40.0f * u01());
        if (i & 1) h = make_shared<scaling>(h, vec3(size, 0.8f * size, size));
        tree.add(make_shared<translate>(h, vec3(rad * std::cos(ang), y, rad * std::sin(ang))));
    }
    shared_ptr<hittable> one_tree = make_shared<bvh_node>(tree);

    hittable_list world;
    world.add(make_shared<sphere>(point3(0.0f, -1000.0f, 0.0f), 1000.0f, make_shared<lambertian>(ground_checker())));
    for (int a = -4; a < 4; a++)
        for (int b = -3; b < 2; b++) {
            float jx = u01(), jz = u01(), angle = 360.0f * u01(), size = 0.8f + 0.5f * u01();
            shared_ptr<hittable> h = make_shared<rotate_y>(one_tree, angle);
            if ((a + b) & 1) h = make_shared<scaling>(h, vec3(size, size * 1.1f, size));
            world.add(make_shared<translate>(h, vec3(5.0f * a + 2.0f * jx, 0.0f, 5.0f * b + 2.0f * jz)));
        }
    shoot(world, View{kWide, 640, 16, kSky, 35.0f, point3(4.0f, 11.0f, 30.0f), point3(0.0f, 1.5f, -4.0f), 0.0f, 10.0f});
}

// 10^6 cells on a grid, one sphere or quad per cell, plus the ground: splitmix64 keyed layout,
// twelve draws per cell whether they are used or not.
// under_bvh / width 480: the two reduced forms the reference harness records (synth60_list, synth60_bvh).
static void synth1m(int half, bool under_bvh = false, int width = 3840) {
    uint64_t state = 0x5EED1234ull;
    auto u01 = [&]() {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        return float(z >> 40) * (1.0f / 16777216.0f);
    };
    hittable_list world;
    world.add(make_shared<sphere>(point3(0.0f, -10000.0f, 0.0f), 10000.0f, make_shared<lambertian>(ground_checker())));
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
    // (the library builds its own tree over anything larger than a handful of primitives either way;
    // under a bvh_node the primitives are numbered in the order of that tree's leaves, which decides ties)
    if (under_bvh) world = hittable_list(make_shared<bvh_node>(world));
    shoot(world, View{kWide, width, 1, kSky, 40.0f, point3(0.0f, 40.0f, 130.0f), point3(0.0f, 0.0f, 0.0f), 0.0f, 10.0f});
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: rt_scenes <scene> [--spp N] [--width W] [--depth D] [--out file.png] [--dump file.rtscene]\n");
        return 2;
    }
    int synth_half = 500;
    bool synth_bvh = false; // --bvh 1: the world under one bvh_node, as a user of the reference would hold it
    for (int i = 2; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--hits") && i + 2 < argc) {
            g_rays_in = argv[i + 1];
            g_hits_out = argv[i + 2];
            i++;
        } else if (!strcmp(argv[i], "--spp")) g_spp = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--width")) g_width = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--depth")) g_depth = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--out")) g_out = argv[i + 1];
        else if (!strcmp(argv[i], "--dump")) setenv("RTAMD_DUMP_SCENE", argv[i + 1], 1);
        else if (!strcmp(argv[i], "--half")) synth_half = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--bvh")) synth_bvh = atoi(argv[i + 1]) != 0;
    }
    std::string s = argv[1];
    if (s == "bouncing_spheres") bouncing_spheres();
    else if (s == "checkered_spheres") checkered_spheres();
    else if (s == "earth") earth();
    else if (s == "perlin_spheres") perlin_spheres();
    else if (s == "quads") quads();
    else if (s == "simple_light") simple_light();
    else if (s == "cornell_box") cornell_box();
    else if (s == "rotated_in_bvh") rotated_in_bvh();
    else if (s == "grove") grove();
    else if (s == "orchard") orchard();
    else if (s == "synth1m") synth1m(synth_half, synth_bvh);
    else if (s == "synth60_list") synth1m(60, false, 480);
    else if (s == "synth60_bvh") synth1m(60, true, 480);
    else {
        fprintf(stderr, "unknown scene %s\n", argv[1]);
        return 2;
    }
    return 0;
}
