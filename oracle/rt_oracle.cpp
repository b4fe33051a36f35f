// rt_oracle.cpp — CPU restatement of the reference's path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (raytracing_amd/, bench.py's
// measured leg) may link, import or call this file; it is the checker that the
// HIP path is compared against, and the CPU baseline leg of bench.py.
//
// It restates, in plain fp32 C++ with the reference's operation order (compile
// with -ffp-contract=off), the algorithm of YanjieShiFennec/RayTracing:
//   camera::initialize/get_ray/ray_color      camera.h:76-174
//   hittable_list::hit / bvh_node::hit        hittable_list.h:29-43, bvh.h:48-88
//   aabb::hit                                 aabb.h:40-63
//   sphere::hit / quad::hit                   sphere.h:28-113, quad.h:11-91
//   translate / rotate_y / rotate_xyz / scaling   hittable.h:51-344
//   lambertian / metal / dielectric / diffuse_light   material.h:24-119
//   solid / checker / image / noise textures  texture.h:16-95, perlin.h:17-95
//   random_float and the samplers             rt_constants.h:28-43, vec3.h:130-149
//   write_color                               color.h:11-32
//
// PINNING: the oracle is pinned against the unmodified reference compiled in the
// authoring container (oracle/_ref/rt_ref, built by oracle/Makefile): with the
// REF sampler it reproduces the reference's closest-hit records and its fp32
// radiance sums bit for bit (tests/test_oracle_pin.py against tests/golden/),
// and against the reference's own known answers (sphere.h:100-102 uv table).
//
// Two trace modes:
//   ORACLE_TRACE_REFERENCE  walks the scene the way the reference does: groups in
//       order, bvh_node recursion with aabb::hit culling, instance wrappers applied
//       to the ray at run time.
//   ORACLE_TRACE_BAKED      linear scan, in test order, over world-space primitives
//       whose instance wrappers were folded into the quad (what the GPU evaluates).
// Two samplers:
//   ORACLE_SAMPLER_REF      the reference's single mt19937 stream and rejection
//       samplers, drawn in the order g++ evaluates the reference's expressions.
//   ORACLE_SAMPLER_CB       the counter-based per-path stream and analytic samplers
//       of the GPU kernels (same integrator, different random numbers).

#include "rt_amd.h"
#include "rt_scene_io.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <thread>
#include <vector>

namespace {

const float kInf = std::numeric_limits<float>::infinity();
const float kPi = 3.1415926535897932385f; // rt_constants.h:21

// ------------------------------------------------------------------ vec3.h:6-174
struct V {
    float x, y, z;
    V() : x(0.0f), y(0.0f), z(0.0f) {}
    V(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit V(const float *p) : x(p[0]), y(p[1]), z(p[2]) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V operator-(V a) { return V(-a.x, -a.y, -a.z); }
inline V operator+(V a, V b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V operator-(V a, V b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V operator*(V a, V b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V operator/(V a, V b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); } // vec3.h:95
inline V operator*(float t, V v) { return V(t * v.x, t * v.y, t * v.z); }   // vec3.h:99
inline V operator*(V v, float t) { return t * v; }
inline V operator/(V v, float t) { return (1.0f / t) * v; }                  // vec3.h:107
inline float len2(V v) { return v.x * v.x + v.y * v.y + v.z * v.z; }        // vec3.h:56
inline float len(V v) { return std::sqrt(len2(v)); }
inline float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    // vec3.h:112
inline V cross(V a, V b) {                                                  // vec3.h:119
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline V unit(V v) { return v / len(v); }                                   // vec3.h:126
inline bool near_zero(V v) {                                                // vec3.h:60
    const float s = 1e-8f;
    return (std::fabs(v.x) < s) && (std::fabs(v.y) < s) && (std::fabs(v.z) < s);
}
inline V reflect(V v, V n) { return v - 2.0f * dot(v, n) * n; }             // vec3.h:163
inline V refract(V uv, V n, float etai_over_etat) {                         // vec3.h:169
    float cos_theta = std::fmin(dot(-uv, n), 1.0f);
    V r_out_perp = etai_over_etat * (uv + cos_theta * n);
    V r_out_parallel = -std::sqrt(std::fabs(1.0f - len2(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}

struct Ray {
    V o, d;
    float tm;
    V at(float t) const { return o + t * d; } // ray.h:20
};

struct Interval { // interval.h:5-42
    float min, max;
    bool contains(float x) const { return min <= x && x <= max; }
    bool surrounds(float x) const { return min < x && x < max; }
    float size() const { return max - min; }
};
inline Interval iv_union(Interval a, Interval b) { // interval.h:12-16
    Interval r;
    r.min = a.min <= b.min ? a.min : b.min;
    r.max = a.max >= b.max ? a.max : b.max;
    return r;
}
inline Interval iv_expand(Interval a, float delta) { // interval.h:36
    float padding = delta / 2.0f;
    return Interval{a.min - padding, a.max + padding};
}

struct Box { // aabb.h:6-82
    Interval x, y, z;
    const Interval &axis(int n) const { return n == 1 ? y : (n == 2 ? z : x); }
};
inline void pad_to_minimums(Box &b) { // aabb.h:74-81
    float delta = 0.0001f;
    if (b.x.size() < delta) b.x = iv_expand(b.x, delta);
    if (b.y.size() < delta) b.y = iv_expand(b.y, delta);
    if (b.z.size() < delta) b.z = iv_expand(b.z, delta);
}
inline Box box_points(V a, V b) { // aabb.h:17-26
    Box r;
    r.x = (a.x <= b.x) ? Interval{a.x, b.x} : Interval{b.x, a.x};
    r.y = (a.y <= b.y) ? Interval{a.y, b.y} : Interval{b.y, a.y};
    r.z = (a.z <= b.z) ? Interval{a.z, b.z} : Interval{b.z, a.z};
    pad_to_minimums(r);
    return r;
}
inline Box box_union(const Box &a, const Box &b) { // aabb.h:28-32
    return Box{iv_union(a.x, b.x), iv_union(a.y, b.y), iv_union(a.z, b.z)};
}
inline Box box_empty() { // aabb.h:84 (pad_to_minimums leaves +inf/-inf untouched: size = -inf)
    Interval e{+kInf, -kInf};
    Box b{e, e, e};
    // aabb(interval,interval,interval) pads: size() = -inf < delta -> expand; inf-0.00005 = inf
    pad_to_minimums(b);
    return b;
}
inline int longest_axis(const Box &b) { // aabb.h:65-70
    if (b.x.size() > b.y.size()) return b.x.size() > b.z.size() ? 0 : 2;
    return b.y.size() > b.z.size() ? 1 : 2;
}
inline bool box_hit(const Box &b, const Ray &r, Interval ray_t) { // aabb.h:40-63
    for (int axis = 0; axis < 3; axis++) {
        const Interval &ax = b.axis(axis);
        const float adinv = 1.0f / r.d[axis];
        float t0 = (ax.min - r.o[axis]) * adinv;
        float t1 = (ax.max - r.o[axis]) * adinv;
        if (t0 < t1) {
            if (t0 > ray_t.min) ray_t.min = t0;
            if (t1 < ray_t.max) ray_t.max = t1;
        } else {
            if (t1 > ray_t.min) ray_t.min = t1;
            if (t0 < ray_t.max) ray_t.max = t0;
        }
        if (ray_t.max <= ray_t.min) return false;
    }
    return true;
}

struct Hit { // hittable.h:10-26
    V p, normal;
    int prim = -1;
    float t = 0, u = 0, v = 0;
    bool front_face = false;
    void set_face_normal(const Ray &r, V outward) {
        front_face = dot(r.d, outward) < 0;
        normal = front_face ? outward : -outward;
    }
};

// ------------------------------------------------------------------ primitives
struct Sphere {
    V center1, center_vec, center2;
    float radius;
    bool moving;
};
struct Quad {
    V Q, u, v, w, normal;
    float D;
};
inline Quad make_quad(V Q, V u, V v) { // quad.h:11-23
    Quad q;
    q.Q = Q;
    q.u = u;
    q.v = v;
    V n = cross(u, v);
    q.normal = unit(n);
    q.D = dot(q.normal, Q);
    q.w = n / dot(n, n);
    return q;
}

inline void sphere_uv(V p, float &u, float &v) { // sphere.h:96-113
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + kPi;
    u = phi / (2.0f * kPi);
    v = theta / kPi;
}

inline bool sphere_hit(const Sphere &s, const Ray &r, Interval ray_t, Hit &rec) { // sphere.h:28-74
    V center = s.moving ? s.center1 + r.tm * s.center_vec : s.center1;
    V oc = center - r.o;
    float a = len2(r.d);
    float h = dot(r.d, oc);
    float c = len2(oc) - s.radius * s.radius;
    float discriminant = h * h - a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = std::sqrt(discriminant);
    float root = (h - sqrtd) / a;
    if (!ray_t.surrounds(root)) {
        root = (h + sqrtd) / a;
        if (!ray_t.surrounds(root)) return false;
    }
    rec.t = root;
    rec.p = r.at(rec.t);
    V outward = (rec.p - center) / s.radius;
    rec.set_face_normal(r, outward);
    sphere_uv(outward, rec.u, rec.v);
    return true;
}

inline bool quad_hit(const Quad &q, const Ray &r, Interval ray_t, Hit &rec) { // quad.h:32-91
    float denominator = dot(q.normal, r.d);
    if (fabsf(denominator) < 1e-8f) return false;
    float t = (q.D - dot(q.normal, r.o)) / denominator;
    if (!ray_t.contains(t)) return false;
    V intersection = r.at(t);
    V planar = intersection - q.Q;
    float alpha = dot(q.w, cross(planar, q.v));
    float beta = dot(q.w, cross(q.u, planar));
    Interval unit_iv{0.0f, 1.0f};
    if (!unit_iv.contains(alpha) || !unit_iv.contains(beta)) return false;
    rec.u = alpha;
    rec.v = beta;
    rec.t = t;
    rec.p = intersection;
    rec.set_face_normal(r, q.normal);
    return true;
}

// ------------------------------------------------------------------ instance wrappers
struct Xform {
    int type;
    V p;                     // translate offset / scale factors
    float sa, ca, sb, cb, sg, cg; // rotate_xyz (rotate_y uses sb, cb)
};
inline Xform make_xform(const rt_xform_op &op) {
    Xform x;
    x.type = op.type;
    x.p = V(op.p);
    x.sa = x.sb = x.sg = 0.0f;
    x.ca = x.cb = x.cg = 1.0f;
    if (op.type == RT_XF_ROTATE_Y) { // hittable.h:102-105
        float radians = op.p[0] * kPi / 180.0f;
        x.sb = sinf(radians);
        x.cb = cosf(radians);
    } else if (op.type == RT_XF_ROTATE_XYZ) { // hittable.h:214-223
        float ar = op.p[0] * kPi / 180.0f, br = op.p[1] * kPi / 180.0f, gr = op.p[2] * kPi / 180.0f;
        x.sa = sinf(ar);
        x.ca = cosf(ar);
        x.sb = sinf(br);
        x.cb = cosf(br);
        x.sg = sinf(gr);
        x.cg = cosf(gr);
    }
    return x;
}
inline V rot_y_ccw(const Xform &x, V v) { // hittable.h:162-166
    float vx = x.cb * v.x + x.sb * v.z;
    float vz = -x.sb * v.x + x.cb * v.z;
    return V(vx, v.y, vz);
}
inline V rot_y_cw(const Xform &x, V v) { // hittable.h:169-173
    float vx = x.cb * v.x - x.sb * v.z;
    float vz = x.sb * v.x + x.cb * v.z;
    return V(vx, v.y, vz);
}
inline V rot_xyz_ccw(const Xform &x, V v) { // hittable.h:284-295
    float vx = x.cb * x.cg * v.x + (x.sa * x.sb * x.cg - x.sg * x.ca) * v.y +
               (x.sb * x.ca * x.cg + x.sa * x.sg) * v.z;
    float vy = x.cb * x.sg * v.x + (x.ca * x.cg + x.sa * x.sb * x.sg) * v.y +
               (-x.sa * x.cg + x.sg * x.sb * x.ca) * v.z;
    float vz = -x.sb * v.x + x.sa * x.cb * v.y + x.ca * x.cb * v.z;
    return V(vx, vy, vz);
}
inline V rot_xyz_cw(const Xform &x, V v) { // hittable.h:298-309
    float vx = x.cb * x.cg * v.x + x.sg * x.cb * v.y - x.sb * v.z;
    float vy = (x.sa * x.sb * x.cg - x.sg * x.ca) * v.x + (x.sa * x.sb * x.sg + x.ca * x.cg) * v.y +
               x.sa * x.cb * v.z;
    float vz = (x.sb * x.ca * x.cg + x.sa * x.sg) * v.x + (x.sb * x.sg * x.ca - x.sa * x.cg) * v.y +
               x.ca * x.cb * v.z;
    return V(vx, vy, vz);
}

// ------------------------------------------------------------------ scene
struct Prim {
    int type, material;
    Sphere s;   // object space
    Quad q;     // object space
    Sphere sb;  // baked to world space
    Quad qb;
    std::vector<Xform> chain; // innermost first
    Box ref_box;              // the box the reference would give this leaf (no wrappers)
};

struct BvhNode {
    Box box;
    int left, right; // >= 0: node index; < 0: leaf ~prim
};

struct Group {
    int kind, first, count;
    int bvh_root; // node index or -1
};

struct Scene {
    std::vector<Prim> prims;
    std::vector<Group> groups;
    std::vector<BvhNode> nodes;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels;
    bool bake_ok = true;
};

struct Counters {
    uint64_t rays = 0, box_tests = 0, sphere_tests = 0, quad_tests = 0;
};

// bvh_node(objects, start, end) given the FINAL leaf order (bvh.h:48-75).  The
// std::sort calls of the reference already happened when the scene was
// flattened (prims are listed in leaf order), so the tree is the recursive
// median split of the run and every node box is the union of its leaves.
int build_ref_bvh(Scene &sc, int start, int end) {
    Box bb = box_empty();
    for (int i = start; i < end; i++) bb = box_union(bb, sc.prims[i].ref_box);
    int span = end - start;
    BvhNode n;
    n.box = bb;
    if (span == 1) {
        n.left = n.right = ~start;
    } else if (span == 2) {
        n.left = ~start;
        n.right = ~(start + 1);
    } else {
        int mid = start + span / 2;
        n.left = build_ref_bvh(sc, start, mid);
        n.right = build_ref_bvh(sc, mid, end);
    }
    sc.nodes.push_back(n);
    return (int)sc.nodes.size() - 1;
}

bool prim_hit_reference(const Prim &p, const Ray &r_world, Interval ray_t, Hit &rec) {
    // outermost wrapper first (hittable.h:57-69, 133-151, 251-269, 319-334)
    Ray r = r_world;
    for (int i = (int)p.chain.size() - 1; i >= 0; i--) {
        const Xform &x = p.chain[i];
        switch (x.type) {
        case RT_XF_TRANSLATE: r.o = r.o - x.p; break;
        case RT_XF_ROTATE_Y: r.o = rot_y_cw(x, r.o); r.d = rot_y_cw(x, r.d); break;
        case RT_XF_ROTATE_XYZ: r.o = rot_xyz_cw(x, r.o); r.d = rot_xyz_cw(x, r.d); break;
        case RT_XF_SCALE: r.o = r.o / x.p; r.d = r.d / x.p; break;
        }
    }
    bool h = (p.type == RT_PRIM_SPHERE) ? sphere_hit(p.s, r, ray_t, rec) : quad_hit(p.q, r, ray_t, rec);
    if (!h) return false;
    for (size_t i = 0; i < p.chain.size(); i++) {
        const Xform &x = p.chain[i];
        switch (x.type) {
        case RT_XF_TRANSLATE: rec.p = rec.p + x.p; break;
        case RT_XF_ROTATE_Y: rec.p = rot_y_ccw(x, rec.p); rec.normal = rot_y_ccw(x, rec.normal); break;
        case RT_XF_ROTATE_XYZ: rec.p = rot_xyz_ccw(x, rec.p); rec.normal = rot_xyz_ccw(x, rec.normal); break;
        case RT_XF_SCALE: rec.p = rec.p * x.p; rec.normal = unit(rec.normal / x.p); break;
        }
    }
    return true;
}

bool bvh_hit(const Scene &sc, int node, const Ray &r, Interval ray_t, Hit &rec, Counters &cn) {
    if (node < 0) { // leaf
        const Prim &p = sc.prims[~node];
        if (p.type == RT_PRIM_SPHERE) cn.sphere_tests++; else cn.quad_tests++;
        if (!prim_hit_reference(p, r, ray_t, rec)) return false;
        rec.prim = ~node;
        return true;
    }
    const BvhNode &n = sc.nodes[node];
    cn.box_tests++;
    if (!box_hit(n.box, r, ray_t)) return false; // bvh.h:78
    bool hl = bvh_hit(sc, n.left, r, ray_t, rec, cn);
    bool hr = bvh_hit(sc, n.right, r, Interval{ray_t.min, hl ? rec.t : ray_t.max}, rec, cn);
    return hl || hr;
}

// world.hit(r, interval(tmin,tmax), rec) as the reference evaluates it.
bool trace_reference(const Scene &sc, const Ray &r, float tmin, float tmax, Hit &rec, Counters &cn) {
    cn.rays++;
    bool any = false;
    float closest = tmax;
    Hit tmp;
    for (const Group &g : sc.groups) { // hittable_list.h:29-43
        if (g.kind == RT_GROUP_BVH && g.bvh_root >= 0) {
            if (bvh_hit(sc, g.bvh_root, r, Interval{tmin, closest}, tmp, cn)) {
                any = true;
                closest = tmp.t;
                rec = tmp;
            }
        } else {
            for (int i = g.first; i < g.first + g.count; i++) {
                const Prim &p = sc.prims[i];
                if (p.type == RT_PRIM_SPHERE) cn.sphere_tests++; else cn.quad_tests++;
                if (prim_hit_reference(p, r, Interval{tmin, closest}, tmp)) {
                    any = true;
                    closest = tmp.t;
                    tmp.prim = i;
                    rec = tmp;
                }
            }
        }
    }
    if (!any) rec.prim = -1;
    return any;
}

// Linear scan over baked world-space primitives, in test order.  The interval
// rules of the two primitive kinds (open for spheres, closed for quads) make
// this the same closest hit, ties included, as the sequential scan above.
bool trace_baked(const Scene &sc, const Ray &r, float tmin, float tmax, Hit &rec, Counters &cn) {
    cn.rays++;
    bool any = false;
    float closest = tmax;
    Hit tmp;
    for (int i = 0; i < (int)sc.prims.size(); i++) {
        const Prim &p = sc.prims[i];
        bool h;
        if (p.type == RT_PRIM_SPHERE) {
            cn.sphere_tests++;
            h = sphere_hit(p.sb, r, Interval{tmin, closest}, tmp);
        } else {
            cn.quad_tests++;
            h = quad_hit(p.qb, r, Interval{tmin, closest}, tmp);
        }
        if (h) {
            any = true;
            closest = tmp.t;
            tmp.prim = i;
            rec = tmp;
        }
    }
    if (!any) rec.prim = -1;
    return any;
}

Box sphere_ref_box(const Sphere &s) { // sphere.h:11-26
    V rvec(s.radius, s.radius, s.radius);
    if (!s.moving) return box_points(s.center1 - rvec, s.center1 + rvec);
    Box b1 = box_points(s.center1 - rvec, s.center1 + rvec);
    Box b2 = box_points(s.center2 - rvec, s.center2 + rvec);
    return box_union(b1, b2);
}
Box quad_ref_box(const Quad &q) { // quad.h:25-30
    Box d1 = box_points(q.Q, q.Q + q.u + q.v);
    Box d2 = box_points(q.Q + q.u, q.Q + q.v);
    return box_union(d1, d2);
}

// Folds a wrapper chain into a point / direction (innermost wrapper first).
V xf_point(const std::vector<Xform> &chain, V p) {
    for (const Xform &x : chain) {
        switch (x.type) {
        case RT_XF_TRANSLATE: p = p + x.p; break;
        case RT_XF_ROTATE_Y: p = rot_y_ccw(x, p); break;
        case RT_XF_ROTATE_XYZ: p = rot_xyz_ccw(x, p); break;
        case RT_XF_SCALE: p = p * x.p; break;
        }
    }
    return p;
}
V xf_vector(const std::vector<Xform> &chain, V v) {
    for (const Xform &x : chain) {
        switch (x.type) {
        case RT_XF_TRANSLATE: break;
        case RT_XF_ROTATE_Y: v = rot_y_ccw(x, v); break;
        case RT_XF_ROTATE_XYZ: v = rot_xyz_ccw(x, v); break;
        case RT_XF_SCALE: v = v * x.p; break;
        }
    }
    return v;
}

Scene *scene_from_desc(const rt_scene_desc *d) {
    Scene *sc = new Scene();
    sc->materials.assign(d->materials, d->materials + d->n_materials);
    sc->textures.assign(d->textures, d->textures + d->n_textures);
    sc->perlins.assign(d->perlins, d->perlins + d->n_perlins);
    sc->images.assign(d->images, d->images + d->n_images);
    sc->texels.assign(d->texels, d->texels + d->n_texel_bytes);
    sc->prims.resize(d->n_prims);
    for (int i = 0; i < d->n_prims; i++) {
        const rt_prim &rp = d->prims[i];
        Prim &p = sc->prims[i];
        p.type = rp.type;
        p.material = rp.material;
        if (rp.xform >= 0) {
            const rt_xform_chain &c = d->chains[rp.xform];
            for (int k = 0; k < c.n_ops; k++) p.chain.push_back(make_xform(d->xform_ops[c.first_op + k]));
        }
        if (rp.type == RT_PRIM_SPHERE) {
            p.s.center1 = V(rp.a);
            p.s.center_vec = V(rp.b);
            p.s.center2 = V(rp.d);
            p.s.radius = rp.c[0];
            p.s.moving = (rp.flags & RT_PRIM_FLAG_MOVING) != 0;
            p.ref_box = sphere_ref_box(p.s);
            // bake: translation and uniform scale keep a sphere a sphere
            p.sb = p.s;
            p.sb.moving = true; // center1 + tm*0 == center1 exactly
            for (const Xform &x : p.chain) {
                if (x.type == RT_XF_TRANSLATE) {
                    p.sb.center1 = p.sb.center1 + x.p;
                } else if (x.type == RT_XF_SCALE && x.p.x == x.p.y && x.p.y == x.p.z) {
                    p.sb.center1 = p.sb.center1 * x.p;
                    p.sb.center_vec = p.sb.center_vec * x.p;
                    p.sb.radius = p.sb.radius * x.p.x;
                } else {
                    sc->bake_ok = false;
                }
            }
        } else {
            p.q = make_quad(V(rp.a), V(rp.b), V(rp.c));
            p.ref_box = quad_ref_box(p.q);
            p.qb = p.chain.empty() ? p.q
                                   : make_quad(xf_point(p.chain, V(rp.a)), xf_vector(p.chain, V(rp.b)),
                                               xf_vector(p.chain, V(rp.c)));
        }
    }
    for (int g = 0; g < d->n_groups; g++) {
        Group gr{d->groups[g].kind, d->groups[g].first_prim, d->groups[g].n_prims, -1};
        if (gr.kind == RT_GROUP_BVH && gr.count > 0) {
            bool plain = true;
            for (int i = gr.first; i < gr.first + gr.count; i++) plain = plain && sc->prims[i].chain.empty();
            // wrapped leaves: the reference's node boxes would come from the wrappers'
            // own boxes (hittable.h:54,248,316); scanned as a list instead (ties aside, same hit)
            if (plain) gr.bvh_root = build_ref_bvh(*sc, gr.first, gr.first + gr.count);
        }
        sc->groups.push_back(gr);
    }
    return sc;
}

// ------------------------------------------------------------------ textures / materials
float perlin_interp(const V c[2][2][2], float u, float v, float w) { // perlin.h:75-95
    float uu = u * u * (3 - 2 * u);
    float vv = v * v * (3 - 2 * v);
    float ww = w * w * (3 - 2 * w);
    float accum = 0.0f;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                V weight_v(u - i, v - j, w - k);
                accum += (i * uu + (1 - i) * (1 - uu)) * (j * vv + (1 - j) * (1 - vv)) *
                         (k * ww + (1 - k) * (1 - ww)) * dot(c[i][j][k], weight_v);
            }
    return accum;
}
float perlin_noise(const rt_perlin &pn, V p) { // perlin.h:17-36
    float u = p.x - floorf(p.x);
    float v = p.y - floorf(p.y);
    float w = p.z - floorf(p.z);
    int i = int(floorf(p.x));
    int j = int(floorf(p.y));
    int k = int(floorf(p.z));
    V c[2][2][2];
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                // all three axes go through perm_x (perlin.h:33)
                int idx = pn.perm_x[(i + di) & 255] ^ pn.perm_x[(j + dj) & 255] ^ pn.perm_x[(k + dk) & 255];
                c[di][dj][dk] = V(pn.randvec[idx]);
            }
    return perlin_interp(c, u, v, w);
}
float perlin_turb(const rt_perlin &pn, V p, int depth) { // perlin.h:38-50
    float accum = 0.0f;
    V temp_p = p;
    float weight = 1.0f;
    for (int i = 0; i < depth; i++) {
        accum += weight * perlin_noise(pn, temp_p);
        weight *= 0.5f;
        temp_p = V(temp_p.x * 2, temp_p.y * 2, temp_p.z * 2);
    }
    return std::fabs(accum);
}

V texture_value(const Scene &sc, int tex, float u, float v, V p) {
    const rt_texture *t = &sc.textures[tex];
    while (t->type == RT_TEX_CHECKER) { // texture.h:40-47
        int xi = int(std::floor(t->scale * p.x));
        int yi = int(std::floor(t->scale * p.y));
        int zi = int(std::floor(t->scale * p.z));
        bool is_even = (xi + yi + zi) % 2 == 0;
        t = &sc.textures[is_even ? t->even : t->odd];
    }
    switch (t->type) {
    case RT_TEX_SOLID: return V(t->color); // texture.h:22-24
    case RT_TEX_NOISE: {                   // texture.h:85-90
        const rt_perlin &pn = sc.perlins[t->perlin];
        return V(0.5f, 0.5f, 0.5f) * (1.0f + sinf(t->scale * p.z + 10.0f * perlin_turb(pn, p, 7)));
    }
    case RT_TEX_IMAGE: { // texture.h:60-74, rtw_stb_image.h:69-94
        if (t->image < 0) return V(0.0f, 0.0f, 1.0f);
        const rt_image &im = sc.images[t->image];
        u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
        float vc = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        v = 1.0f - vc;
        int i = int(u * im.width);
        int j = int(v * im.height);
        i = i < 0 ? 0 : (i < im.width ? i : im.width - 1);
        j = j < 0 ? 0 : (j < im.height ? j : im.height - 1);
        const uint8_t *px = sc.texels.data() + im.offset + ((size_t)j * im.width + i) * 3;
        float cs = 1.0f / 255.0f;
        return V(cs * px[0], cs * px[1], cs * px[2]);
    }
    }
    return V();
}

// ------------------------------------------------------------------ samplers
// rt_constants.h:28-33 with libstdc++'s uniform_real_distribution<float>:
// one 32-bit draw, float(u32) / 2^32, and 1.0 replaced by the float below it.
struct RefStream {
    std::mt19937 gen; // default seed 5489
    float next() {
        float r = float(gen()) / 4294967296.0f;
        if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
        return r * (1.0f - 0.0f) + 0.0f;
    }
    float range(float lo, float hi) { return lo + (hi - lo) * next(); } // rt_constants.h:35-38
};

// Counter-based stream of the GPU kernels: the key is (seed, pixel, sample),
// successive numbers come from a 32-bit LCG with a PCG output permutation.
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct CbStream {
    uint32_t s;
    void seed(uint32_t seed, uint32_t pixel, uint32_t sample) {
        s = mix32(pixel * 0x9E3779B9u + mix32(sample * 0x85EBCA6Bu + mix32(seed + 0x68E31DA4u)));
    }
    float next() {
        s = s * 747796405u + 2891336453u;
        uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
        w = (w >> 22u) ^ w;
        return float(w >> 8) * (1.0f / 16777216.0f);
    }
};

struct Sampler {
    bool ref;
    RefStream *rs;
    CbStream cb;
    float next() { return ref ? rs->next() : cb.next(); }

    // camera.h:128-131.  g++ evaluates the constructor arguments right to left:
    // the first draw becomes y.
    void sample_square(float &dx, float &dy) {
        if (ref) {
            dy = rs->next() - 0.5f;
            dx = rs->next() - 0.5f;
        } else {
            dx = cb.next() - 0.5f;
            dy = cb.next() - 0.5f;
        }
    }
    // vec3.h:130-136 (rejection) | analytic: sqrt(u1) * (cos 2 pi u2, sin 2 pi u2)
    void unit_disk(float &px, float &py) {
        if (ref) {
            while (true) {
                float y = rs->range(-1.0f, 1.0f);
                float x = rs->range(-1.0f, 1.0f);
                if (x * x + y * y + 0.0f * 0.0f < 1) {
                    px = x;
                    py = y;
                    return;
                }
            }
        } else {
            float r = std::sqrt(cb.next());
            float phi = 2.0f * kPi * cb.next();
            px = r * cosf(phi);
            py = r * sinf(phi);
        }
    }
    // vec3.h:138-149 (rejection, then normalised) | analytic uniform direction
    V unit_vector() {
        if (ref) {
            while (true) {
                float z = rs->range(-1.0f, 1.0f);
                float y = rs->range(-1.0f, 1.0f);
                float x = rs->range(-1.0f, 1.0f);
                V p(x, y, z);
                if (len2(p) < 1.0f) return unit(p);
            }
        } else {
            float z = 1.0f - 2.0f * cb.next();
            float phi = 2.0f * kPi * cb.next();
            float rr = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
            return V(rr * cosf(phi), rr * sinf(phi), z);
        }
    }
};

// ------------------------------------------------------------------ camera
void camera_initialize(const rt_camera_desc &c, rt_camera_frame &f) { // camera.h:76-111
    memset(&f, 0, sizeof(f));
    f.image_width = c.image_width;
    int image_height = int(c.image_width / c.aspect_ratio);
    image_height = (image_height < 1) ? 1 : image_height;
    f.image_height = image_height;
    f.pixel_samples_scale = 1.0f / c.samples_per_pixel;
    V center(c.lookfrom);
    float theta = c.vfov * kPi / 180.0f;
    float h = tanf(theta / 2.0f);
    float viewport_height = 2.0f * h * c.focus_dist;
    float viewport_width = viewport_height * (float(c.image_width) / image_height);
    V w = unit(V(c.lookfrom) - V(c.lookat));
    V u = unit(cross(V(c.vup), w));
    V v = cross(w, u);
    V viewport_u = viewport_width * u;
    V viewport_v = viewport_height * -v;
    V pdu = viewport_u / float(c.image_width);
    V pdv = viewport_v / float(image_height);
    V upper_left = center - (c.focus_dist * w) - viewport_u / 2.0f - viewport_v / 2.0f;
    V p00 = upper_left + 0.5f * (pdu + pdv);
    float defocus_radius = c.focus_dist * tanf((c.defocus_angle / 2.0f) * kPi / 180.0f);
    V ddu = u * defocus_radius;
    V ddv = v * defocus_radius;
    auto put = [](float *dst, V s) { dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; };
    put(f.center, center);
    put(f.pixel00_loc, p00);
    put(f.pixel_delta_u, pdu);
    put(f.pixel_delta_v, pdv);
    put(f.u, u);
    put(f.v, v);
    put(f.w, w);
    put(f.defocus_disk_u, ddu);
    put(f.defocus_disk_v, ddv);
}

Ray get_ray(const rt_camera_desc &c, const rt_camera_frame &f, int i, int j, Sampler &sm) { // camera.h:113-126
    float dx, dy;
    sm.sample_square(dx, dy);
    V pixel_sample = V(f.pixel00_loc) + ((i + dx) * V(f.pixel_delta_u)) + ((j + dy) * V(f.pixel_delta_v));
    V origin = V(f.center);
    if (!(c.defocus_angle <= 0)) { // camera.h:133-137
        float px, py;
        sm.unit_disk(px, py);
        origin = V(f.center) + (px * V(f.defocus_disk_u)) + (py * V(f.defocus_disk_v));
    }
    Ray r;
    r.o = origin;
    r.d = pixel_sample - origin;
    r.tm = sm.next();
    return r;
}

float schlick(float cosine, float ri) { // material.h:97-103
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * powf((1.0f - cosine), 5.0f);
}

struct RenderCtx {
    const Scene *sc;
    int trace_mode;
    Counters cn;
    uint64_t shades[4] = {0, 0, 0, 0};
};

// ray_color (camera.h:139-174), unrolled into a loop that keeps the recursion's
// arithmetic: result = e0 + a0*(e1 + a1*(e2 + ...)) is evaluated innermost
// first, so the bounce chain is recorded and folded back to front.
V ray_color(RenderCtx &cx, Ray r, int depth, V background, Sampler &sm) {
    struct Term { V emit, atten; };
    Term stack[256];
    std::vector<Term> heap;
    int n = 0;
    V tail(0.0f, 0.0f, 0.0f);
    const Scene &sc = *cx.sc;
    while (true) {
        if (depth <= 0) { tail = V(0.0f, 0.0f, 0.0f); break; }
        Hit rec;
        bool h = cx.trace_mode == 0 ? trace_reference(sc, r, 0.001f, kInf, rec, cx.cn)
                                    : trace_baked(sc, r, 0.001f, kInf, rec, cx.cn);
        if (!h) { tail = background; break; }
        const rt_material &m = sc.materials[sc.prims[rec.prim].material];
        cx.shades[m.type]++;
        V emitted(0.0f, 0.0f, 0.0f);
        V attenuation;
        Ray scattered;
        bool did_scatter = false;
        switch (m.type) {
        case RT_MAT_DIFFUSE_LIGHT: // material.h:113-115
            emitted = texture_value(sc, m.texture, rec.u, rec.v, rec.p);
            break;
        case RT_MAT_LAMBERTIAN: { // material.h:30-42
            V dir = rec.normal + sm.unit_vector();
            if (near_zero(dir)) dir = rec.normal;
            scattered = Ray{rec.p, dir, r.tm};
            attenuation = texture_value(sc, m.texture, rec.u, rec.v, rec.p);
            did_scatter = true;
            break;
        }
        case RT_MAT_METAL: { // material.h:53-60
            V reflected = reflect(r.d, rec.normal);
            reflected = unit(reflected) + (m.param * sm.unit_vector());
            scattered = Ray{rec.p, reflected, r.tm};
            attenuation = V(m.albedo);
            did_scatter = dot(scattered.d, rec.normal) > 0;
            break;
        }
        case RT_MAT_DIELECTRIC: { // material.h:72-90
            attenuation = V(1.0f, 1.0f, 1.0f);
            float ri = rec.front_face ? (1.0f / m.param) : m.param;
            V unit_direction = unit(r.d);
            float cos_theta = fminf(dot(-unit_direction, rec.normal), 1.0f);
            float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
            bool cannot_refract = ri * sin_theta > 1.0f;
            V direction;
            bool do_reflect;
            if (sm.ref) {
                do_reflect = cannot_refract || schlick(cos_theta, ri) > sm.next();
            } else {
                float xi = sm.next(); // the GPU stream always draws
                do_reflect = cannot_refract || schlick(cos_theta, ri) > xi;
            }
            if (do_reflect)
                direction = reflect(unit_direction, rec.normal);
            else
                direction = refract(unit_direction, rec.normal, ri);
            scattered = Ray{rec.p, direction, r.tm};
            did_scatter = true;
            break;
        }
        }
        if (!did_scatter) { tail = emitted; break; }
        Term t{emitted, attenuation};
        if (n < 256) stack[n] = t; else heap.push_back(t);
        n++;
        r = scattered;
        depth--;
    }
    V result = tail;
    for (int k = n - 1; k >= 0; k--) {
        const Term &t = k < 256 ? stack[k] : heap[k - 256];
        result = t.emit + t.atten * result; // camera.h:171-173
    }
    return result;
}

} // namespace

// =====================================================================================
// C entry points (ctypes)
extern "C" {

enum { ORACLE_TRACE_REFERENCE = 0, ORACLE_TRACE_BAKED = 1 };
enum { ORACLE_SAMPLER_REF = 0, ORACLE_SAMPLER_CB = 1 };

typedef struct oracle_render_opts {
    int32_t sampler;       // ORACLE_SAMPLER_*
    int32_t trace_mode;    // ORACLE_TRACE_*
    int32_t sample_begin;  // CB: first global sample index
    int32_t sample_count;  // samples per pixel rendered
    uint32_t seed;         // CB stream seed
    int32_t row_begin, row_end; // rows rendered (row_end <= 0: all)
    int32_t n_threads;     // CB only; REF is one sequential stream
    int64_t burn_in;       // REF: draws discarded before the first sample
} oracle_render_opts;

typedef struct oracle_stats {
    uint64_t paths, rays, box_tests, sphere_tests, quad_tests;
    uint64_t shades[4];
    double seconds;
} oracle_stats;

void *oracle_scene_create(const rt_scene_desc *d) { return scene_from_desc(d); }
void oracle_scene_destroy(void *s) { delete (Scene *)s; }
int oracle_scene_bake_ok(void *s) { return ((Scene *)s)->bake_ok ? 1 : 0; }

void *oracle_scene_load(const char *path) {
    rt_scene_desc *d = rtio::load(path);
    if (!d) return nullptr;
    Scene *s = scene_from_desc(d);
    free(d);
    return s;
}

void oracle_camera_initialize(const rt_camera_desc *c, rt_camera_frame *f) { camera_initialize(*c, *f); }

void oracle_trace_batch(void *scene, const rt_ray *rays, int64_t n, float tmin, float tmax, int mode,
                        rt_hit *hits, oracle_stats *st) {
    const Scene &sc = *(Scene *)scene;
    Counters cn;
    for (int64_t k = 0; k < n; k++) {
        Ray r{V(rays[k].orig), V(rays[k].dir), rays[k].tm};
        Hit rec;
        bool h = mode == ORACLE_TRACE_REFERENCE ? trace_reference(sc, r, tmin, tmax, rec, cn)
                                                : trace_baked(sc, r, tmin, tmax, rec, cn);
        rt_hit &o = hits[k];
        memset(&o, 0, sizeof(o));
        o.prim = -1;
        if (h) {
            o.prim = rec.prim;
            o.front_face = rec.front_face ? 1 : 0;
            o.t = rec.t;
            o.p[0] = rec.p.x; o.p[1] = rec.p.y; o.p[2] = rec.p.z;
            o.normal[0] = rec.normal.x; o.normal[1] = rec.normal.y; o.normal[2] = rec.normal.z;
            o.u = rec.u;
            o.v = rec.v;
        }
    }
    if (st) {
        memset(st, 0, sizeof(*st));
        st->rays = cn.rays;
        st->box_tests = cn.box_tests;
        st->sphere_tests = cn.sphere_tests;
        st->quad_tests = cn.quad_tests;
    }
}

// The sample loop of camera::render (camera.h:40-47): ADDS the radiance sums of
// the requested samples to accum (H*W*3 floats).
void oracle_render(void *scene, const rt_camera_desc *cam, const oracle_render_opts *o, float *accum,
                   oracle_stats *st) {
    const Scene &sc = *(Scene *)scene;
    rt_camera_frame f;
    camera_initialize(*cam, f);
    int W = f.image_width, H = f.image_height;
    int row0 = o->row_begin < 0 ? 0 : o->row_begin;
    int row1 = (o->row_end <= 0 || o->row_end > H) ? H : o->row_end;
    V background(cam->background);
    auto t0 = std::chrono::steady_clock::now();
    oracle_stats total;
    memset(&total, 0, sizeof(total));

    if (o->sampler == ORACLE_SAMPLER_REF) {
        RefStream rs;
        for (int64_t i = 0; i < o->burn_in; i++) (void)rs.next();
        RenderCtx cx;
        cx.sc = &sc;
        cx.trace_mode = o->trace_mode;
        Sampler sm;
        sm.ref = true;
        sm.rs = &rs;
        for (int j = row0; j < row1; j++)
            for (int i = 0; i < W; i++) {
                V pixel(0.0f, 0.0f, 0.0f);
                for (int s = 0; s < o->sample_count; s++) {
                    Ray r = get_ray(*cam, f, i, j, sm);
                    pixel = pixel + ray_color(cx, r, cam->max_depth, background, sm);
                }
                float *a = accum + ((size_t)j * W + i) * 3;
                a[0] += pixel.x; a[1] += pixel.y; a[2] += pixel.z;
            }
        total.rays = cx.cn.rays;
        total.box_tests = cx.cn.box_tests;
        total.sphere_tests = cx.cn.sphere_tests;
        total.quad_tests = cx.cn.quad_tests;
        for (int k = 0; k < 4; k++) total.shades[k] = cx.shades[k];
    } else {
        int nt = o->n_threads < 1 ? 1 : o->n_threads;
        std::atomic<int> next_row(row0);
        std::vector<oracle_stats> per(nt);
        auto worker = [&](int tid) {
            RenderCtx cx;
            cx.sc = &sc;
            cx.trace_mode = o->trace_mode;
            Sampler sm;
            sm.ref = false;
            sm.rs = nullptr;
            for (;;) {
                int j = next_row.fetch_add(1);
                if (j >= row1) break;
                for (int i = 0; i < W; i++) {
                    V pixel(0.0f, 0.0f, 0.0f);
                    for (int s = 0; s < o->sample_count; s++) {
                        sm.cb.seed(o->seed, (uint32_t)(j * W + i), (uint32_t)(o->sample_begin + s));
                        Ray r = get_ray(*cam, f, i, j, sm);
                        pixel = pixel + ray_color(cx, r, cam->max_depth, background, sm);
                    }
                    float *a = accum + ((size_t)j * W + i) * 3;
                    a[0] += pixel.x; a[1] += pixel.y; a[2] += pixel.z;
                }
            }
            memset(&per[tid], 0, sizeof(oracle_stats));
            per[tid].rays = cx.cn.rays;
            per[tid].box_tests = cx.cn.box_tests;
            per[tid].sphere_tests = cx.cn.sphere_tests;
            per[tid].quad_tests = cx.cn.quad_tests;
            for (int k = 0; k < 4; k++) per[tid].shades[k] = cx.shades[k];
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++) th.emplace_back(worker, t);
        worker(0);
        for (auto &t : th) t.join();
        for (int t = 0; t < nt; t++) {
            total.rays += per[t].rays;
            total.box_tests += per[t].box_tests;
            total.sphere_tests += per[t].sphere_tests;
            total.quad_tests += per[t].quad_tests;
            for (int k = 0; k < 4; k++) total.shades[k] += per[t].shades[k];
        }
    }
    total.paths = (uint64_t)W * (row1 - row0) * o->sample_count;
    total.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st) *st = total;
}

// write_color (color.h:17-32) on scale * accum.
void oracle_tonemap_u8(const float *accum, int64_t n_pixels, float scale, uint8_t *rgb) {
    for (int64_t i = 0; i < n_pixels * 3; i++) {
        float c = scale * accum[i];
        float g = c > 0.0f ? std::sqrt(c) : 0.0f;
        g = g < 0.000f ? 0.000f : (g > 0.999f ? 0.999f : g);
        rgb[i] = (uint8_t) int(256 * g);
    }
}

void oracle_sphere_uv(const float *n, float *u, float *v) { sphere_uv(V(n), *u, *v); }

void oracle_texture_value(void *scene, int tex, float u, float v, const float *p, float *out) {
    V c = texture_value(*(Scene *)scene, tex, u, v, V(p));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// First n numbers of the reference's random_float() stream.
void oracle_ref_random_floats(int64_t n, float *out) {
    RefStream rs;
    for (int64_t i = 0; i < n; i++) out[i] = rs.next();
}

// First n numbers of the counter-based stream for (seed, pixel, sample).
void oracle_cb_random_floats(uint32_t seed, uint32_t pixel, uint32_t sample, int64_t n, float *out) {
    CbStream cb;
    cb.seed(seed, pixel, sample);
    for (int64_t i = 0; i < n; i++) out[i] = cb.next();
}

// World-space quad/sphere constants after baking (for comparing with the library).
void oracle_baked_prim(void *scene, int i, float *out20) {
    const Prim &p = ((Scene *)scene)->prims[i];
    memset(out20, 0, 20 * sizeof(float));
    if (p.type == RT_PRIM_SPHERE) {
        const Sphere &s = p.sb;
        float v[8] = {s.center1.x, s.center1.y, s.center1.z, s.radius, s.center_vec.x, s.center_vec.y,
                      s.center_vec.z, 0};
        memcpy(out20, v, sizeof(v));
    } else {
        const Quad &q = p.qb;
        float v[16] = {q.Q.x, q.Q.y, q.Q.z, q.D, q.u.x, q.u.y, q.u.z, 0, q.v.x, q.v.y, q.v.z, 0,
                       q.w.x, q.w.y, q.w.z, 0};
        memcpy(out20, v, sizeof(v));
        out20[16] = q.normal.x; out20[17] = q.normal.y; out20[18] = q.normal.z;
    }
}

} // extern "C"
