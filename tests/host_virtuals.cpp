// host_virtuals.cpp — test program: material::scatter / emitted and texture::value called on the
// host through the drop-in classes (material.h:13-20, texture.h:13 in the reference), and a
// user-defined hittable / material compiled against the drop-in's virtual interfaces.
// Built and run by tests/test_dropin.py; exits non-zero with a message on the first failure.
#include "camera.h"
#include "bvh.h"
#include "hittable_list.h"
#include "material.h"
#include "quad.h"
#include "sphere.h"
#include "texture.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>

static int g_failures = 0;
#define CHECK(cond)                                                                                \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #cond);               \
            g_failures++;                                                                          \
        }                                                                                          \
    } while (0)

static bool close_to(float a, float b, float tol = 1e-6f) { return std::fabs(a - b) <= tol * (1.0f + std::fabs(b)); }

// user code may derive from the interfaces (it just cannot be flattened for the GPU)
class plane_y0 : public hittable {
  public:
    explicit plane_y0(shared_ptr<material> m) : mat(m) {}
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override {
        if (r.direction().y() == 0.0f) return false;
        float t = -r.origin().y() / r.direction().y();
        if (!ray_t.surrounds(t)) return false;
        rec.t = t;
        rec.p = r.at(t);
        rec.set_face_normal(r, vec3(0.0f, 1.0f, 0.0f));
        rec.mat = mat;
        return true;
    }
    aabb bounding_box() const override { return aabb(point3(-1e9f, 0.0f, -1e9f), point3(1e9f, 0.0f, 1e9f)); }

  private:
    shared_ptr<material> mat;
};

class glow : public material {
  public:
    color emitted(float, float, const point3 &p) const override { return color(p.x(), 2.0f, 3.0f); }
};

int main() {
    // ---- textures ------------------------------------------------------------------------------
    solid_color red(0.9f, 0.1f, 0.2f);
    CHECK(red.value(0.3f, 0.7f, point3(5, 6, 7)).x() == 0.9f);
    checker_texture chk(0.5f, color(1, 1, 1), color(0, 0, 0)); // inv_scale 2: cells of 0.5
    CHECK(chk.value(0, 0, point3(0.1f, 0.1f, 0.1f)).x() == 1.0f);   // 0 + 0 + 0: even
    CHECK(chk.value(0, 0, point3(0.6f, 0.1f, 0.1f)).x() == 0.0f);   // 1 + 0 + 0: odd
    CHECK(chk.value(0, 0, point3(-0.1f, 0.1f, 0.1f)).x() == 0.0f);  // floor(-0.2) = -1: odd
    CHECK(chk.value(0, 0, point3(-0.1f, -0.1f, 0.1f)).x() == 1.0f); // -2: even (parity of a negative sum)
    image_texture missing("no_such_file_anywhere.jpg");
    color blue = missing.value(0.5f, 0.5f, point3(0, 0, 0)); // texture.h:62
    CHECK(blue.x() == 0.0f && blue.y() == 0.0f && blue.z() == 1.0f);
    noise_texture marble(4.0f);
    for (int i = 0; i < 200; i++) {
        point3 p(0.37f * i - 20.0f, 0.11f * i, 5.0f - 0.23f * i);
        color c = marble.value(0, 0, p);
        CHECK(c.x() == c.y() && c.y() == c.z());
        CHECK(c.x() >= 0.0f && c.x() <= 1.0f); // 0.5 (1 + sin)
    }
    {   // perlin: noise vanishes on the lattice (every corner term carries a zero offset vector there),
        // and turb is the absolute weighted sum of seven octaves
        perlin pn;
        CHECK(pn.noise(point3(3.0f, -2.0f, 7.0f)) == 0.0f);
        point3 q(1.3f, 2.7f, -0.4f);
        float acc = 0.0f, wgt = 1.0f;
        point3 t = q;
        for (int k = 0; k < 7; k++) {
            acc += wgt * pn.noise(t);
            wgt *= 0.5f;
            t = t * 2.0f;
        }
        CHECK(pn.turb(q, 7) == std::fabs(acc));
        CHECK(std::fabs(pn.noise(q)) <= 1.0f);
    }

    // ---- materials -------------------------------------------------------------------------------
    hit_record rec;
    rec.p = point3(1, 2, 3);
    rec.normal = vec3(0, 1, 0);
    rec.front_face = true;
    rec.u = 0.2f;
    rec.v = 0.4f;
    ray in(point3(0, 5, 0), vec3(0.3f, -1.0f, 0.1f), 0.625f);
    color att;
    ray out;
    lambertian matte(color(0.1f, 0.2f, 0.3f));
    for (int i = 0; i < 1000; i++) {
        CHECK(matte.scatter(in, rec, att, out));
        CHECK(att.x() == 0.1f && att.y() == 0.2f && att.z() == 0.3f);
        CHECK(out.origin().x() == 1.0f && out.time() == 0.625f);
        // normal + unit vector: never below the surface (up to the rounding of the unit vector)
        CHECK(dot(out.direction(), rec.normal) >= -2e-7f && out.direction().length() <= 2.0f + 1e-6f);
    }
    CHECK(matte.emitted(0, 0, rec.p).length() == 0.0f); // material.h:13-16
    metal mirror(color(0.8f, 0.8f, 0.8f), 0.0f);
    CHECK(mirror.scatter(in, rec, att, out));
    vec3 expect = unit_vector(reflect(in.direction(), rec.normal));
    CHECK(close_to(out.direction().x(), expect.x()) && close_to(out.direction().y(), expect.y()) &&
          close_to(out.direction().z(), expect.z()));
    CHECK(att.x() == 0.8f);
    metal rough(color(0.8f, 0.8f, 0.8f), 7.0f); // fuzz is clamped to 1 (material.h:51)
    int absorbed = 0;
    ray grazing(point3(0, 5, 0), vec3(1.0f, -0.05f, 0.0f), 0.0f);
    for (int i = 0; i < 2000; i++) {
        bool ok = rough.scatter(grazing, rec, att, out);
        absorbed += ok ? 0 : 1;
        CHECK(ok == (dot(out.direction(), rec.normal) > 0));
        CHECK((out.direction() - unit_vector(reflect(grazing.direction(), rec.normal))).length() <= 1.0f + 1e-5f);
    }
    CHECK(absorbed > 100 && absorbed < 1900); // a grazing mirror direction fuzzed by a unit sphere
    dielectric glass(1.5f);
    int reflected = 0;
    for (int i = 0; i < 4000; i++) {
        CHECK(glass.scatter(in, rec, att, out));
        CHECK(att.x() == 1.0f && att.y() == 1.0f && att.z() == 1.0f);
        reflected += out.direction().y() > 0 ? 1 : 0;
    }
    {   // Schlick at this angle: the reflected share of 4000 draws
        vec3 ud = unit_vector(in.direction());
        float c = std::fmin(dot(-ud, rec.normal), 1.0f), r0 = (1.0f - 1.0f / 1.5f) / (1.0f + 1.0f / 1.5f);
        r0 *= r0;
        float refl = r0 + (1.0f - r0) * std::pow(1.0f - c, 5.0f);
        CHECK(std::fabs(reflected / 4000.0f - refl) < 0.03f);
    }
    {   // total internal reflection: from inside (ri = 1.5) at a shallow angle nothing refracts
        hit_record inside = rec;
        inside.front_face = false;
        ray shallow(point3(0, 5, 0), vec3(1.0f, -0.2f, 0.0f), 0.0f);
        for (int i = 0; i < 100; i++) {
            CHECK(glass.scatter(shallow, inside, att, out));
            CHECK(out.direction().y() > 0);
        }
    }
    diffuse_light lamp(color(4, 5, 6));
    CHECK(!lamp.scatter(in, rec, att, out)); // material.h:17-20
    CHECK(lamp.emitted(0, 0, rec.p).y() == 5.0f);

    // ---- hittables through the virtual interface, user-defined ones among them ----------------------
    auto ground = make_shared<plane_y0>(make_shared<glow>());
    hittable_list world;
    world.add(ground);
    world.add(make_shared<sphere>(point3(0, 1, 0), 1.0f, make_shared<lambertian>(color(0.5f, 0.5f, 0.5f))));
    world.add(make_shared<quad>(point3(-1, 0.5f, 2), vec3(2, 0, 0), vec3(0, 2, 0), make_shared<metal>(color(1, 1, 1), 0.0f)));
    hit_record h;
    CHECK(world.hit(ray(point3(0, 5, 0), vec3(0, -1, 0)), interval(0.001f, infinity), h));
    CHECK(h.t == 3.0f && h.p.y() == 2.0f && h.front_face && h.normal.y() == 1.0f); // the sphere's top
    CHECK(world.hit(ray(point3(5, 5, 0), vec3(0, -1, 0)), interval(0.001f, infinity), h));
    CHECK(h.t == 5.0f && h.mat->emitted(0, 0, h.p).x() == 5.0f); // the user-defined plane and material
    CHECK(world.hit(ray(point3(0, 1.5f, 5), vec3(0, 0, -1)), interval(0.001f, infinity), h));
    CHECK(h.t == 3.0f && h.u == 0.5f && h.v == 0.5f); // the quad, in front of the sphere
    CHECK(!world.hit(ray(point3(0, 1.5f, 5), vec3(0, 0, -1)), interval(0.001f, 2.5f), h));
    CHECK(world.hit(ray(point3(0, 1.5f, 5), vec3(0, 0, -1)), interval(0.001f, 3.0f), h));   // closed: quad.h:46
    {   // a bvh_node over library objects: same closest hits as the list, children reachable
        hittable_list spheres;
        for (int i = 0; i < 40; i++)
            spheres.add(make_shared<sphere>(point3(float(i % 8) * 2.5f, 0.5f, float(i / 8) * 2.5f), 0.5f + 0.01f * i,
                                            make_shared<lambertian>(color(0.5f, 0.5f, 0.5f))));
        auto tree = make_shared<bvh_node>(spheres);
        CHECK(tree->get_left_child() && tree->get_right_child() && tree->get_left_child()->get_left_child());
        int both = 0;
        for (int i = 0; i < 2000; i++) {
            ray r(point3(random_float(-2, 20), 6.0f, random_float(-2, 12)),
                  vec3(random_float(-1, 1), -1.0f, random_float(-1, 1)), 0.0f);
            hit_record a, b;
            bool ha = spheres.hit(r, interval(0.001f, infinity), a), hb = tree->hit(r, interval(0.001f, infinity), b);
            CHECK(ha == hb);
            if (ha && hb) {
                both++;
                CHECK(a.t == b.t && a.p.x() == b.p.x() && a.normal.y() == b.normal.y());
            }
        }
        CHECK(both > 200);
    }
    {   // The order of a bvh_node's leaves is the one the recursion of bvh.h:48-75 produces when it is
        // written down plainly (sort the range of shared_ptr by box minimum along the longest axis of
        // the range's box, halve, recurse), equal keys included: positions on a coarse lattice give
        // thousands of them, and 70 000 objects reach the ranges that are sorted on several threads.
        struct plain {
            static void order(std::vector<shared_ptr<hittable>> &o, size_t a, size_t b) {
                if (b - a <= 2) return;
                aabb box = aabb::empty;
                for (size_t i = a; i < b; i++) box = aabb(box, o[i]->bounding_box());
                const int axis = box.longest_axis();
                std::sort(o.begin() + a, o.begin() + b, [axis](const shared_ptr<hittable> x, const shared_ptr<hittable> y) {
                    return x->bounding_box().axis_interval(axis).min < y->bounding_box().axis_interval(axis).min;
                });
                const size_t mid = a + (b - a) / 2;
                order(o, a, mid);
                order(o, mid, b);
            }
        };
        auto grey = make_shared<lambertian>(color(0.5f, 0.5f, 0.5f));
        for (int n : {7, 1000, 70000}) {
            std::vector<shared_ptr<hittable>> objs;
            for (int i = 0; i < n; i++)
                objs.push_back(make_shared<sphere>(point3(float(int(random_float(0, 40))), float(int(random_float(0, 6))),
                                                          float(int(random_float(0, 40)))), 0.25f, grey));
            std::vector<shared_ptr<hittable>> expect = objs, given = objs;
            plain::order(expect, 0, expect.size());
            bvh_node tree(given, 0, given.size());
            CHECK(given == expect); // the caller's vector is left in leaf order, as by the reference
            // ... and the tree's own leaves, walked left to right, are that order
            std::vector<shared_ptr<hittable>> walked;
            struct walk {
                static void go(const shared_ptr<hittable> &h, std::vector<shared_ptr<hittable>> &out) {
                    auto l = h->get_left_child(), r = h->get_right_child();
                    if (!l && !r) {
                        out.push_back(h);
                        return;
                    }
                    go(l, out);
                    if (r != l) go(r, out);
                }
            };
            walk::go(tree.get_left_child(), walked);
            if (tree.get_right_child() != tree.get_left_child()) walk::go(tree.get_right_child(), walked);
            CHECK(walked == expect);
            bvh_node copy = tree; // copies share the leaves and build their own children
            std::vector<shared_ptr<hittable>> again;
            walk::go(copy.get_left_child(), again);
            if (copy.get_right_child() != copy.get_left_child()) walk::go(copy.get_right_child(), again);
            CHECK(again == expect);
        }
    }
    if (g_failures) {
        fprintf(stderr, "%d check(s) failed\n", g_failures);
        return 1;
    }
    printf("host virtuals ok\n");
    return 0;
}
