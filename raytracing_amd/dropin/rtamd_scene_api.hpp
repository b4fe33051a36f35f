// rtamd_scene_api.hpp — the reference's C++ scene-construction API as a recording front end.
//
// A program written against YanjieShiFennec/RayTracing (hittable_list, make_shared<sphere|quad>,
// box, translate / rotate_y / rotate_xyz / scaling, bvh_node, the material and texture classes,
// the public camera fields and camera::render) compiles against these headers unchanged.
// Nothing here traces a ray: the classes only remember how they were built.  camera::render
// (camera.h:31 in the reference) flattens the graph into an rt_scene_desc and hands it to
// librt_amd.so through the C ABI of rt_amd.h; the image is rendered by the HIP kernels.
//
// Header-only and multi-TU safe (everything inline).
#pragma once

#include "rtamd_jpeg.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "rt_amd.h"
#include "rt_core.h" // the kernels' per-ray arithmetic, compiled for the host: hit(), scatter(), value()
#include "rt_scene_io.h"

// rt_constants.h:11-17 puts these names at global scope; scene code relies on them
using std::make_shared;
using std::shared_ptr;
using std::sqrt;
using std::string;
using std::fabs;
using std::cout;

const float infinity = std::numeric_limits<float>::infinity(); // rt_constants.h:20
const float pi = 3.1415926535897932385f;                       // rt_constants.h:21

inline float degrees_to_radians(float degrees) { return degrees * pi / 180.0f; }

namespace rtamd {
// The reference draws every random number, scene construction included, from one
// default-seeded std::mt19937 through libstdc++'s uniform_real_distribution<float>
// (rt_constants.h:28-33): one 32-bit output, converted to float, divided by 2^32, and 1.0
// replaced by the float just below it.  Restated explicitly so that the stream does not
// depend on the standard library this header is compiled with.
inline std::mt19937 &generator() {
    static std::mt19937 g; // seed 5489
    return g;
}
} // namespace rtamd

inline float random_float() {
    float r = float(rtamd::generator()()) / 4294967296.0f;
    if (r >= 1.0f) r = std::nextafter(1.0f, 0.0f);
    return r;
}
inline float random_float(float min, float max) { return min + (max - min) * random_float(); }
inline int random_int(int min, int max) { return int(random_float(float(min), float(max + 1))); }

// ---------------------------------------------------------------------------------------------
// vec3.h
class vec3 {
  public:
    float e[3];
    vec3() : e{0.0f, 0.0f, 0.0f} {}
    vec3(float e0, float e1, float e2) : e{e0, e1, e2} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    vec3 operator-() const { return vec3(-e[0], -e[1], -e[2]); }
    float operator[](int i) const { return e[i]; }
    float &operator[](int i) { return e[i]; }
    vec3 &operator+=(const vec3 &v) {
        for (int k = 0; k < 3; k++) e[k] += v.e[k];
        return *this;
    }
    vec3 &operator*=(float t) {
        for (int k = 0; k < 3; k++) e[k] *= t;
        return *this;
    }
    vec3 &operator*=(const vec3 &v) {
        for (int k = 0; k < 3; k++) e[k] *= v.e[k];
        return *this;
    }
    vec3 &operator/=(float t) { return *this *= 1.0f / t; }
    float length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    float length() const { return std::sqrt(length_squared()); }
    bool near_zero() const {
        const float s = 1e-8f;
        return std::fabs(e[0]) < s && std::fabs(e[1]) < s && std::fabs(e[2]) < s;
    }
    // vec3.h:66-72 draws the three numbers inside one argument list; the reference's
    // toolchain (g++) evaluates such a list right to left, so the FIRST draw becomes z.
    // Sequenced explicitly here: the scene no longer depends on the compiler.
    static vec3 random() {
        float c = random_float(), b = random_float(), a = random_float();
        return vec3(a, b, c);
    }
    static vec3 random(float min, float max) {
        float c = random_float(min, max), b = random_float(min, max), a = random_float(min, max);
        return vec3(a, b, c);
    }
};
using point3 = vec3;
using color = vec3;

inline std::ostream &operator<<(std::ostream &out, const vec3 &v) { return out << v.e[0] << ' ' << v.e[1] << ' ' << v.e[2]; }
inline vec3 operator+(const vec3 &u, const vec3 &v) { return vec3(u.e[0] + v.e[0], u.e[1] + v.e[1], u.e[2] + v.e[2]); }
inline vec3 operator-(const vec3 &u, const vec3 &v) { return vec3(u.e[0] - v.e[0], u.e[1] - v.e[1], u.e[2] - v.e[2]); }
inline vec3 operator*(const vec3 &u, const vec3 &v) { return vec3(u.e[0] * v.e[0], u.e[1] * v.e[1], u.e[2] * v.e[2]); }
inline vec3 operator/(const vec3 &u, const vec3 &v) { return vec3(u.e[0] / v.e[0], u.e[1] / v.e[1], u.e[2] / v.e[2]); }
inline vec3 operator*(float t, const vec3 &v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(const vec3 &v, float t) { return t * v; }
inline vec3 operator/(const vec3 &v, float t) { return (1.0f / t) * v; }
inline float dot(const vec3 &u, const vec3 &v) { return u.e[0] * v.e[0] + u.e[1] * v.e[1] + u.e[2] * v.e[2]; }
inline vec3 cross(const vec3 &u, const vec3 &v) {
    return vec3(u.e[1] * v.e[2] - u.e[2] * v.e[1], u.e[2] * v.e[0] - u.e[0] * v.e[2], u.e[0] * v.e[1] - u.e[1] * v.e[0]);
}
inline vec3 unit_vector(const vec3 &v) { return v / v.length(); }
inline vec3 random_in_unit_disk() {
    for (;;) {
        float b = random_float(-1.0f, 1.0f), a = random_float(-1.0f, 1.0f); // first draw -> y (see vec3::random)
        vec3 p(a, b, 0.0f);
        if (p.length_squared() < 1) return p;
    }
}
inline vec3 random_in_unit_sphere() {
    for (;;) {
        vec3 p = vec3::random(-1.0f, 1.0f);
        if (p.length_squared() < 1.0f) return p;
    }
}
inline vec3 random_unit_vector() { return unit_vector(random_in_unit_sphere()); }
inline vec3 random_on_hemisphere(const vec3 &normal) {
    vec3 v = random_unit_vector();
    return dot(v, normal) > 0.0f ? v : -v;
}
inline vec3 reflect(const vec3 &v, const vec3 &n) { return v - 2.0f * dot(v, n) * n; }
inline vec3 refract(const vec3 &uv, const vec3 &n, float etai_over_etat) {
    float cos_theta = std::fmin(dot(-uv, n), 1.0f);
    vec3 perp = etai_over_etat * (uv + cos_theta * n);
    vec3 parallel = -std::sqrt(std::fabs(1.0f - perp.length_squared())) * n;
    return perp + parallel;
}

// ---------------------------------------------------------------------------------------------
// ray.h, interval.h, aabb.h, color.h (value types scene code may touch)
class ray {
  public:
    ray() {}
    ray(const point3 &origin, const vec3 &direction) : orig(origin), dir(direction), tm(0.0f) {}
    ray(const point3 &origin, const vec3 &direction, float time) : orig(origin), dir(direction), tm(time) {}
    const point3 &origin() const { return orig; }
    const vec3 &direction() const { return dir; }
    float time() const { return tm; }
    point3 at(float t) const { return orig + t * dir; }

  private:
    point3 orig;
    vec3 dir;
    float tm = 0.0f;
};

class interval {
  public:
    float min, max;
    interval() : min(+infinity), max(-infinity) {}
    interval(float lo, float hi) : min(lo), max(hi) {}
    interval(const interval &a, const interval &b) {
        min = a.min <= b.min ? a.min : b.min;
        max = a.max >= b.max ? a.max : b.max;
    }
    float size() const { return max - min; }
    bool contains(float x) const { return min <= x && x <= max; }
    bool surrounds(float x) const { return min < x && x < max; }
    float clamp(float x) const { return x < min ? min : (x > max ? max : x); }
    interval expand(float delta) const {
        float padding = delta / 2.0f;
        return interval(min - padding, max + padding);
    }
    static const interval empty, universe;
};
inline const interval interval::empty = interval(+infinity, -infinity);
inline const interval interval::universe = interval(-infinity, +infinity);
inline interval operator+(const interval &i, float d) { return interval(i.min + d, i.max + d); }
inline interval operator+(float d, const interval &i) { return i + d; }
inline interval operator*(const interval &i, float d) { return interval(i.min * d, i.max * d); }
inline interval operator*(float d, const interval &i) { return i * d; }

// aabb.h.  The renderer builds its own (correct, padded) bounds on the flattened primitives;
// these boxes exist because bvh_node orders its leaves by them (bvh.h:54-73), and the order
// decides ties, so they follow the reference's arithmetic exactly.
class aabb {
  public:
    interval x, y, z;
    aabb() {}
    aabb(const interval &ix, const interval &iy, const interval &iz) : x(ix), y(iy), z(iz) { pad_to_minimums(); }
    aabb(const point3 &a, const point3 &b) {
        x = (a[0] <= b[0]) ? interval(a[0], b[0]) : interval(b[0], a[0]);
        y = (a[1] <= b[1]) ? interval(a[1], b[1]) : interval(b[1], a[1]);
        z = (a[2] <= b[2]) ? interval(a[2], b[2]) : interval(b[2], a[2]);
        pad_to_minimums();
    }
    aabb(const aabb &b0, const aabb &b1) : x(b0.x, b1.x), y(b0.y, b1.y), z(b0.z, b1.z) {}
    const interval &axis_interval(int n) const { return n == 1 ? y : (n == 2 ? z : x); }
    int longest_axis() const {
        if (x.size() > y.size()) return x.size() > z.size() ? 0 : 2;
        return y.size() > z.size() ? 1 : 2;
    }
    // aabb.h:40-63: slab test on a copy of the interval, one division per axis
    bool hit(const ray &r, interval ray_t) const;
    static const aabb empty, universe;

  private:
    void pad_to_minimums() {
        const float delta = 0.0001f;
        if (x.size() < delta) x = x.expand(delta);
        if (y.size() < delta) y = y.expand(delta);
        if (z.size() < delta) z = z.expand(delta);
    }
};
inline bool aabb::hit(const ray &r, interval ray_t) const {
    for (int axis = 0; axis < 3; axis++) {
        const interval &ax = axis_interval(axis);
        const float adinv = 1.0f / r.direction()[axis];
        const float ta = (ax.min - r.origin()[axis]) * adinv, tb = (ax.max - r.origin()[axis]) * adinv;
        const bool ordered = ta < tb; // (false for NaN: the untouched branch, as in the reference)
        const float enter = ordered ? ta : tb, leave = ordered ? tb : ta;
        if (enter > ray_t.min) ray_t.min = enter;
        if (leave < ray_t.max) ray_t.max = leave;
        if (ray_t.max <= ray_t.min) return false;
    }
    return true;
}
inline const aabb aabb::empty = aabb(interval::empty, interval::empty, interval::empty);
inline const aabb aabb::universe = aabb(interval::universe, interval::universe, interval::universe);
inline aabb operator+(const aabb &b, const vec3 &o) { return aabb(b.x + o.x(), b.y + o.y(), b.z + o.z()); }
inline aabb operator+(const vec3 &o, const aabb &b) { return b + o; }
inline aabb operator*(const aabb &b, const vec3 &s) { return aabb(b.x * s.x(), b.y * s.y(), b.z * s.z()); }
inline aabb operator*(const vec3 &s, const aabb &b) { return b * s; }

inline float linear_to_gamma(float c) { return c > 0.0f ? std::sqrt(c) : 0.0f; }
inline void write_color(int rgb[3], const color &pixel_color) { // color.h:17-32
    static const interval intensity(0.000f, 0.999f);
    for (int k = 0; k < 3; k++) rgb[k] = int(256 * intensity.clamp(linear_to_gamma(pixel_color[k])));
}

// ---------------------------------------------------------------------------------------------
// Recorder: the flat scene under construction
namespace rtamd {

class Recorder {
  public:
    std::vector<rt_prim> prims;
    std::vector<rt_group> groups;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_perlin> perlins;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels;
    std::vector<rt_xform_op> ops;
    std::vector<rt_xform_chain> chains;
    std::vector<rt_xform_op> open_wrappers; // outermost first
    // Every recorder has a number of its own: materials and textures remember the index they got
    // from the recorder of that number (a large scene has 10^6 of each; a table from object to
    // index would be searched three times per primitive).
    const uint64_t serial = next_serial();
    static uint64_t next_serial() {
        static std::atomic<uint64_t> n{0};
        return ++n;
    }
    std::map<std::string, int> chain_ids;
    int bvh_depth = 0; // > 0 while recording the leaves of a bvh_node
    // Objects that several instance wrappers hold (hittable.h:51-344 keep a shared_ptr to what they
    // move) are recorded once, as a sub-scene, and placed by rt_instance records: the library then
    // keeps one copy and one tree of them.  A first, counting pass over the scene (`counting`) finds
    // them: per object, how often a wrapper reached it and how many primitives it holds; the
    // recording pass gets that table in `shared`.
    // A shared object may hold shared objects in turn (a tree of shrubs placed many times).  The
    // library's scenes have two levels, so the hierarchy is flattened where it costs nothing: the
    // inner object becomes an instance of the world once per placement of the outer one, under the
    // outer placement's wrappers followed by its own (one chain: the ray passes the same
    // operations in the same order as in the reference's nested hit() calls).  Primitives are
    // never written out for that; only the 16-byte instance records multiply.
    bool counting = false;
    size_t n_counted = 0; // primitives the counting pass has passed, written out
    std::map<const void *, std::pair<int, int>> seen;
    const std::map<const void *, std::pair<int, int>> *shared = nullptr;
    struct SubScene {
        std::vector<rt_prim> prims; // its own primitives, in its own space
        std::vector<rt_group> groups;
        // shared objects inside it: {object, its wrappers relative to this sub-scene, outermost first}
        std::vector<std::pair<const void *, std::vector<rt_xform_op>>> children;
        bool recorded = false;
        int first_prim = -1; // among all primitives, once desc() has appended it
    };
    std::map<const void *, SubScene> sub_scenes;
    std::vector<const void *> sub_order;   // in the order of their first visit
    std::vector<SubScene *> recording;      // the sub-scenes being recorded, innermost last
    struct Placement {
        const void *object;
        int chain;
    };
    std::vector<Placement> placements;
    std::vector<rt_instance> instances; // filled by desc()
    static const int kMinInstancePrims = 8; // smaller objects are cheaper written out

    // Called by a wrapper for the object it holds, with its own op already on open_wrappers.
    template <class F> void child(const void *obj, F record_it) {
        if (counting) {
            // (an object is walked on its first visit only, so the pass costs what the hierarchy
            // holds, not what it expands to)
            auto it = seen.find(obj);
            if (it != seen.end()) {
                it->second.first++;
                n_counted += (size_t)it->second.second;
                return;
            }
            const size_t before = n_counted;
            record_it();
            seen[obj] = std::make_pair(1, (int)(n_counted - before));
            return;
        }
        const std::pair<int, int> *c = nullptr;
        if (shared) {
            auto it = shared->find(obj);
            if (it != shared->end() && it->second.first >= 2 && it->second.second >= kMinInstancePrims) c = &it->second;
        }
        if (!c) {
            record_it();
            return;
        }
        SubScene &sub = sub_scenes[obj];
        if (!sub.recorded) { // first visit: its primitives, in its own space, into its own arrays
            sub.recorded = true;
            sub_order.push_back(obj);
            std::vector<rt_xform_op> outer;
            outer.swap(open_wrappers);
            const int outer_depth = bvh_depth;
            bvh_depth = 0;
            bvh_closed = true;
            recording.push_back(&sub);
            record_it();
            recording.pop_back();
            bvh_depth = outer_depth;
            bvh_closed = true;
            open_wrappers.swap(outer);
        }
        if (recording.empty())
            place(obj, open_wrappers);
        else // inside another shared object: placed with every placement of that one
            recording.back()->children.emplace_back(obj, open_wrappers);
    }
    // An instance of the world for `obj` under `wrappers` (outermost first), and one for every shared
    // object inside it under the wrappers of both.
    void place(const void *obj, const std::vector<rt_xform_op> &wrappers) {
        const SubScene &sub = sub_scenes[obj];
        if (!sub.prims.empty()) placements.push_back(Placement{obj, chain_of(wrappers)});
        for (const auto &ch : sub.children) {
            std::vector<rt_xform_op> w(wrappers);
            w.insert(w.end(), ch.second.begin(), ch.second.end());
            place(ch.first, w);
        }
    }
    bool any_shared() const {
        for (const auto &kv : seen)
            if (kv.second.first >= 2 && kv.second.second >= kMinInstancePrims) return true;
        return false;
    }

    int current_chain() { return chain_of(open_wrappers); }
    int chain_of(const std::vector<rt_xform_op> &outer_first) {
        if (outer_first.empty()) return -1;
        std::vector<rt_xform_op> inner_first(outer_first.rbegin(), outer_first.rend());
        std::string key((const char *)inner_first.data(), inner_first.size() * sizeof(rt_xform_op));
        auto it = chain_ids.find(key);
        if (it != chain_ids.end()) return it->second;
        rt_xform_chain c = {(int32_t)ops.size(), (int32_t)inner_first.size()};
        ops.insert(ops.end(), inner_first.begin(), inner_first.end());
        chains.push_back(c);
        return chain_ids[key] = (int)chains.size() - 1;
    }
    void add_prim(rt_prim p) {
        if (counting) {
            n_counted++;
            return;
        }
        p.xform = current_chain();
        // (a shared object's primitives go to its own arrays while it is being recorded)
        std::vector<rt_prim> &pr = recording.empty() ? prims : recording.back()->prims;
        std::vector<rt_group> &gr = recording.empty() ? groups : recording.back()->groups;
        int kind = bvh_depth > 0 ? RT_GROUP_BVH : RT_GROUP_LIST;
        if (gr.empty() || gr.back().kind != kind || (kind == RT_GROUP_BVH && bvh_closed)) {
            rt_group g = {kind, (int32_t)pr.size(), 0, 0};
            gr.push_back(g);
            bvh_closed = false;
        }
        gr.back().n_prims++;
        pr.push_back(p);
    }
    void begin_bvh() {
        if (bvh_depth++ == 0) bvh_closed = true; // the next leaf opens a fresh BVH group
    }
    void end_bvh() {
        if (--bvh_depth == 0) bvh_closed = true;
    }
    // (appends the sub-scenes to the world's primitives, once, and resolves the placements)
    rt_scene_desc desc() {
        if (!finalized) {
            finalized = true;
            for (const void *obj : sub_order) {
                SubScene &sub = sub_scenes[obj];
                sub.first_prim = (int)prims.size();
                for (rt_group g : sub.groups) {
                    g.first_prim += sub.first_prim;
                    groups.push_back(g);
                }
                prims.insert(prims.end(), sub.prims.begin(), sub.prims.end());
            }
            for (const Placement &pl : placements) {
                const SubScene &sub = sub_scenes[pl.object];
                rt_instance in = {sub.first_prim, (int32_t)sub.prims.size(), pl.chain, 0};
                instances.push_back(in);
            }
        }
        rt_scene_desc d;
        memset(&d, 0, sizeof(d));
        d.abi_version = RT_AMD_ABI_VERSION;
        d.n_prims = (int)prims.size();
        d.prims = prims.data();
        d.n_groups = (int)groups.size();
        d.groups = groups.data();
        d.n_materials = (int)materials.size();
        d.materials = materials.data();
        d.n_textures = (int)textures.size();
        d.textures = textures.data();
        d.n_perlins = (int)perlins.size();
        d.perlins = perlins.data();
        d.n_images = (int)images.size();
        d.images = images.data();
        d.n_texel_bytes = texels.size();
        d.texels = texels.data();
        d.n_xform_ops = (int)ops.size();
        d.xform_ops = ops.data();
        d.n_chains = (int)chains.size();
        d.chains = chains.data();
        d.n_instances = (int)instances.size();
        d.instances = instances.data();
        return d;
    }

  private:
    bool bvh_closed = true;
    bool finalized = false;
};

// Minimal PNG writer (8-bit RGB, stored deflate blocks): what camera::render needs to put its
// pixels on disk (camera.h:56).  The file decodes to exactly the bytes passed in.  At GPU
// rendering speeds this is host time that shows: the checksums run at memory speed (CRC-32
// eight bytes at a time, Adler-32 with the modulo taken once per 5552 bytes) and the file is
// assembled once.
struct PngCrc {
    uint32_t t[8][256];
    PngCrc() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t v = i;
            for (int k = 0; k < 8; k++) v = (v & 1) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
            t[0][i] = v;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) t[k][i] = t[0][t[k - 1][i] & 255] ^ (t[k - 1][i] >> 8);
    }
    uint32_t run(const uint8_t *p, size_t n, uint32_t c) const { // c: running value, start with 0
        c = ~c;
        while (n >= 8) {
            uint32_t lo, hi;
            memcpy(&lo, p, 4);
            memcpy(&hi, p + 4, 4);
            lo ^= c; // (little-endian hosts; the reference's platforms are)
            c = t[7][lo & 255] ^ t[6][(lo >> 8) & 255] ^ t[5][(lo >> 16) & 255] ^ t[4][lo >> 24] ^
                t[3][hi & 255] ^ t[2][(hi >> 8) & 255] ^ t[1][(hi >> 16) & 255] ^ t[0][hi >> 24];
            p += 8;
            n -= 8;
        }
        while (n--) c = t[0][(c ^ *p++) & 255] ^ (c >> 8);
        return ~c;
    }
};

inline bool write_png_rgb8(const char *path, int w, int h, const uint8_t *rgb) {
    static const PngCrc crc;
    const size_t row = (size_t)3 * w + 1, raw_bytes = row * h;
    const size_t n_blocks = raw_bytes ? (raw_bytes + 65534) / 65535 : 1;
    const size_t z_bytes = 2 + 5 * n_blocks + raw_bytes + 4;
    std::vector<uint8_t> out;
    out.reserve(8 + 25 + 12 + z_bytes + 12);
    auto be32 = [&](uint32_t x) {
        for (int s = 24; s >= 0; s -= 8) out.push_back((uint8_t)(x >> s));
    };
    auto begin_chunk = [&](const char *tag, size_t bytes) {
        be32((uint32_t)bytes);
        size_t start = out.size();
        out.insert(out.end(), tag, tag + 4);
        return start;
    };
    auto end_chunk = [&](size_t start) { be32(crc.run(out.data() + start, out.size() - start, 0)); };
    static const uint8_t magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.insert(out.end(), magic, magic + 8);
    size_t c0 = begin_chunk("IHDR", 13);
    be32((uint32_t)w);
    be32((uint32_t)h);
    static const uint8_t fmt[5] = {8, 2, 0, 0, 0};
    out.insert(out.end(), fmt, fmt + 5);
    end_chunk(c0);

    c0 = begin_chunk("IDAT", z_bytes);
    out.push_back(0x78);
    out.push_back(0x01);
    // the filtered scanlines (filter byte 0 + pixels) flow straight into stored blocks
    size_t in_block = 0, left = raw_bytes;
    auto open_block = [&]() {
        size_t n = std::min<size_t>(65535, left);
        out.push_back(left <= 65535 ? 1 : 0);
        out.push_back((uint8_t)(n & 255));
        out.push_back((uint8_t)(n >> 8));
        out.push_back((uint8_t)(~n & 255));
        out.push_back((uint8_t)((~n >> 8) & 255));
        in_block = n;
    };
    auto put = [&](const uint8_t *p, size_t n) {
        while (n > 0) {
            if (in_block == 0) open_block();
            size_t m = std::min(n, in_block);
            out.insert(out.end(), p, p + m);
            p += m;
            n -= m;
            in_block -= m;
            left -= m;
        }
    };
    if (raw_bytes == 0) open_block();
    uint64_t a = 1, b = 0; // Adler-32 of the scanlines, modulo taken once per scanline piece
    const uint8_t zero = 0;
    for (int y = 0; y < h; y++) {
        const uint8_t *px = rgb + (size_t)y * w * 3;
        put(&zero, 1); // filter: none
        put(px, (size_t)w * 3);
        b += a; // the filter byte adds 0 to a
        for (size_t i = 0, n = (size_t)w * 3; i < n;) {
            size_t run = std::min<size_t>(n - i, 1u << 16); // a, b stay far below 2^64
            for (size_t k = 0; k < run; k++) {
                a += px[i + k];
                b += a;
            }
            a %= 65521u;
            b %= 65521u;
            i += run;
        }
    }
    be32((uint32_t)((b << 16) | a));
    end_chunk(c0);
    c0 = begin_chunk("IEND", 0);
    end_chunk(c0);
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

} // namespace rtamd

namespace rtamd {
inline rtc::f3 to_f3(const vec3 &v) { return rtc::mk(v[0], v[1], v[2]); }
inline vec3 to_vec3(rtc::f3 v) { return vec3(v.x, v.y, v.z); }
} // namespace rtamd

// ---------------------------------------------------------------------------------------------
// perlin.h, rtw_stb_image.h, texture.h
class perlin {
  public:
    perlin() { // perlin.h:8-15: 256 unit vectors, then three permutations (two of them unused)
        for (int i = 0; i < 256; i++) {
            vec3 v = unit_vector(vec3::random(-1.0f, 1.0f));
            for (int k = 0; k < 3; k++) table.randvec[i][k] = v[k];
        }
        int scratch[256];
        generate_perm(table.perm_x);
        generate_perm(scratch); // perm_y: drawn so that the random stream stays aligned
        generate_perm(scratch); // perm_z
        for (int i = 0; i < 256; i++) { // the layout the kernels' noise code reads
            dev.randvec[i] = rtc::f4{table.randvec[i][0], table.randvec[i][1], table.randvec[i][2], 0.0f};
            dev.perm_x[i] = table.perm_x[i];
        }
    }
    rt_perlin table;
    rtc::DevPerlin dev;
    float noise(const point3 &p) const { return rtc::perlin_noise(dev, rtamd::to_f3(p)); }           // perlin.h:17-36
    float turb(const point3 &p, int depth) const { return rtc::perlin_turb(dev, rtamd::to_f3(p), depth); } // perlin.h:38-50

  private:
    static void generate_perm(int32_t *p) {
        for (int i = 0; i < 256; i++) p[i] = i;
        for (int i = 255; i > 0; i--) {
            int target = random_int(0, i);
            std::swap(p[i], p[target]);
        }
    }
};

// Image files.  The kernels read texels in the reference's "linear byte" form: the decoded
// 8-bit value b becomes (uchar)(256 * powf(b / 255, 2.2)) (stb's LDR->HDR conversion followed by
// rtw_stb_image.h:96-102).  Formats understood here: binary PPM (P6), converted as above, and
// .rtimg (a raw dump of already converted texels: "RTIMG1\n<w> <h>\n" + w*h*3 bytes).
class rtw_image {
  public:
    rtw_image() {}
    rtw_image(const char *image_filename) {
        std::string filename(image_filename);
        const char *imagedir = getenv("RTW_IMAGES");
        if (imagedir && load(std::string(imagedir) + "/" + filename)) return;
        std::string prefix;
        if (load(filename)) return;
        for (int up = 0; up < 7; up++) { // images/, ../images/, ... six levels up
            if (load(prefix + "images/" + filename)) return;
            prefix += "../";
        }
        std::cerr << "ERROR: Could not load image file '" << image_filename << "'.\n";
    }
    bool load(const std::string &name) { return load_one(name) || load_one(replace_ext(name, ".rtimg")) || load_one(replace_ext(name, ".ppm")); }
    int width() const { return data.empty() ? 0 : w; }
    int height() const { return data.empty() ? 0 : h; }
    const std::vector<uint8_t> &bytes() const { return data; }

  private:
    int w = 0, h = 0;
    std::vector<uint8_t> data;
    static std::string replace_ext(const std::string &n, const char *ext) {
        size_t dot = n.find_last_of('.');
        return (dot == std::string::npos ? n : n.substr(0, dot)) + ext;
    }
    // stb_image's float decode followed by rtw_stb_image.h:96-118: byte -> pow(b/255, 2.2)
    // (double pow of float arguments, rounded to float) -> byte
    static uint8_t linear_byte(uint8_t b) {
        float v = (float)pow((double)(b / 255.0f), (double)2.2f);
        return v <= 0.0 ? 0 : (1.0 <= v ? 255 : (uint8_t)(256.0 * v));
    }
    bool load_one(const std::string &name) {
        FILE *f = fopen(name.c_str(), "rb");
        if (!f) return false;
        std::vector<uint8_t> file;
        uint8_t chunk[65536];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), f)) > 0) file.insert(file.end(), chunk, chunk + got);
        fclose(f);
        bool ok = false;
        int ww = 0, hh = 0;
        data.clear();
        if (file.size() > 3 && file[0] == 0xff && file[1] == 0xd8) { // JPEG, as the reference's earthmap.jpg
            rtamd::jpeg::Image im;
            std::string why;
            if (rtamd::jpeg::decode(file.data(), file.size(), im, &why)) {
                ww = im.width;
                hh = im.height;
                data.resize(im.rgb.size());
                uint8_t lut[256];
                for (int i = 0; i < 256; i++) lut[i] = linear_byte((uint8_t)i);
                for (size_t i = 0; i < data.size(); i++) data[i] = lut[im.rgb[i]];
                ok = true;
            } else {
                std::cerr << "ERROR: " << name << ": " << why << "\n";
            }
        } else {
            // text header, then raw bytes: "RTIMG1\nW H\n" (texels already linear) or binary PPM
            size_t pos = 0;
            auto token = [&]() {
                while (pos < file.size() && isspace(file[pos])) pos++;
                std::string t;
                while (pos < file.size() && !isspace(file[pos])) t.push_back((char)file[pos++]);
                return t;
            };
            std::string magic = token();
            if (magic == "RTIMG1" || magic == "P6") {
                ww = atoi(token().c_str());
                hh = atoi(token().c_str());
                int maxv = magic == "P6" ? atoi(token().c_str()) : 255;
                pos++; // the single whitespace byte that ends the header
                size_t need = (size_t)(ww > 0 ? ww : 0) * (size_t)(hh > 0 ? hh : 0) * 3;
                if (ww > 0 && hh > 0 && maxv == 255 && pos + need <= file.size()) {
                    data.assign(file.begin() + (long)pos, file.begin() + (long)(pos + need));
                    if (magic == "P6")
                        for (uint8_t &b : data) b = linear_byte(b);
                    ok = true;
                }
            }
        }
        if (!ok) data.clear();
        w = ww;
        h = hh;
        return ok;
    }
};

class texture {
  public:
    virtual ~texture() = default;
    virtual color value(float u, float v, const point3 &p) const = 0; // texture.h:13 (host side)
    // returns the texture's index in the flat scene; a user-defined texture cannot be flattened
    virtual int record(rtamd::Recorder &r) const {
        (void)r;
        std::cerr << "librt_amd drop-in: a user-defined texture cannot be rendered on the GPU\n";
        std::exit(1);
    }

  protected:
    int memo(rtamd::Recorder &r, const rt_texture &t) const {
        int id;
        if (known(r, id)) return id;
        r.textures.push_back(t);
        recorded_by = r.serial;
        return recorded_as = (int)r.textures.size() - 1;
    }
    bool known(const rtamd::Recorder &r, int &id) const {
        if (recorded_by != r.serial) return false;
        id = recorded_as;
        return true;
    }
    mutable uint64_t recorded_by = 0; // the recorder (Recorder::serial) that holds this texture ...
    mutable int recorded_as = -1;     // ... at this index
    static rt_texture blank(int type) {
        rt_texture t;
        memset(&t, 0, sizeof(t));
        t.type = type;
        t.even = t.odd = t.perlin = t.image = -1;
        return t;
    }
};

class solid_color : public texture {
  public:
    solid_color(const color &albedo) : albedo(albedo) {}
    solid_color(float red, float green, float blue) : solid_color(color(red, green, blue)) {}
    color value(float, float, const point3 &) const override { return albedo; } // texture.h:22-24
    int record(rtamd::Recorder &r) const override {
        rt_texture t = blank(RT_TEX_SOLID);
        for (int k = 0; k < 3; k++) t.color[k] = albedo[k];
        return memo(r, t);
    }

  private:
    color albedo;
};

class checker_texture : public texture {
  public:
    checker_texture(float scale, shared_ptr<texture> even, shared_ptr<texture> odd)
        : inv_scale(1.0f / scale), even(even), odd(odd) {}
    checker_texture(float scale, const color &c1, const color &c2)
        : checker_texture(scale, make_shared<solid_color>(c1), make_shared<solid_color>(c2)) {}
    color value(float u, float v, const point3 &p) const override { // texture.h:40-47
        const int cell = int(std::floor(inv_scale * p.x())) + int(std::floor(inv_scale * p.y())) +
                         int(std::floor(inv_scale * p.z()));
        return (cell % 2 == 0 ? even : odd)->value(u, v, p);
    }
    int record(rtamd::Recorder &r) const override {
        int known_id;
        if (known(r, known_id)) return known_id;
        rt_texture t = blank(RT_TEX_CHECKER);
        t.scale = inv_scale;
        t.even = even->record(r);
        t.odd = odd->record(r);
        return memo(r, t);
    }

  private:
    float inv_scale;
    shared_ptr<texture> even, odd;
};

class image_texture : public texture {
  public:
    image_texture(const char *filename) : image(filename) {}
    color value(float u, float v, const point3 &) const override { // texture.h:60-74, rtw_stb_image.h:69-79
        if (image.height() <= 0) return color(0.0f, 0.0f, 1.0f);
        u = interval(0.0f, 1.0f).clamp(u);
        v = 1.0f - interval(0.0f, 1.0f).clamp(v);
        int i = int(u * image.width()), j = int(v * image.height());
        i = i < 0 ? 0 : (i < image.width() ? i : image.width() - 1);
        j = j < 0 ? 0 : (j < image.height() ? j : image.height() - 1);
        const uint8_t *px = &image.bytes()[((size_t)j * image.width() + i) * 3];
        const float cs = 1.0f / 255.0f;
        return color(cs * px[0], cs * px[1], cs * px[2]);
    }
    int record(rtamd::Recorder &r) const override {
        int known_id;
        if (known(r, known_id)) return known_id;
        rt_texture t = blank(RT_TEX_IMAGE);
        if (image.height() > 0) { // a missing file renders as (0,0,1): texture.h:62
            rt_image im = {image.width(), image.height(), (uint64_t)r.texels.size()};
            r.texels.insert(r.texels.end(), image.bytes().begin(), image.bytes().end());
            t.image = (int)r.images.size();
            r.images.push_back(im);
        }
        return memo(r, t);
    }

  private:
    rtw_image image;
};

class noise_texture : public texture {
  public:
    noise_texture(float scale) : scale(scale) {}
    color value(float, float, const point3 &p) const override { // texture.h:85-90
        return color(0.5f, 0.5f, 0.5f) * (1.0f + sinf(scale * p.z() + 10.0f * noise.turb(p, 7)));
    }
    int record(rtamd::Recorder &r) const override {
        int known_id;
        if (known(r, known_id)) return known_id;
        rt_texture t = blank(RT_TEX_NOISE);
        t.scale = scale;
        t.perlin = (int)r.perlins.size();
        r.perlins.push_back(noise.table);
        return memo(r, t);
    }

  private:
    perlin noise;
    float scale;
};

// ---------------------------------------------------------------------------------------------
// material.h
class material;
class hit_record { // hittable.h:10-26
  public:
    point3 p;
    vec3 normal;
    shared_ptr<material> mat;
    float t = 0, u = 0, v = 0;
    bool front_face = false;
    void set_face_normal(const ray &r, const vec3 &outward_normal) {
        front_face = dot(r.direction(), outward_normal) < 0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

class material {
  public:
    virtual ~material() = default;
    // material.h:13-20 (host side): nothing emitted, nothing scattered unless a subclass says so
    virtual color emitted(float u, float v, const point3 &p) const {
        (void)u, (void)v, (void)p;
        return color(0.0f, 0.0f, 0.0f);
    }
    virtual bool scatter(const ray &r_in, const hit_record &rec, color &attenuation, ray &scattered) const {
        (void)r_in, (void)rec, (void)attenuation, (void)scattered;
        return false;
    }
    // index in the flat scene; a user-defined material cannot be flattened
    virtual int record(rtamd::Recorder &r) const {
        (void)r;
        std::cerr << "librt_amd drop-in: a user-defined material cannot be rendered on the GPU\n";
        std::exit(1);
    }

  protected:
    int memo(rtamd::Recorder &r, const rt_material &m) const {
        r.materials.push_back(m);
        recorded_by = r.serial;
        return recorded_as = (int)r.materials.size() - 1;
    }
    bool known(rtamd::Recorder &r, int &id) const {
        if (recorded_by != r.serial) return false;
        id = recorded_as;
        return true;
    }
    mutable uint64_t recorded_by = 0; // the recorder (Recorder::serial) that holds this material ...
    mutable int recorded_as = -1;     // ... at this index
};

class lambertian : public material {
  public:
    lambertian(const color &albedo) : tex(make_shared<solid_color>(albedo)) {}
    lambertian(shared_ptr<texture> tex) : tex(tex) {}
    bool scatter(const ray &r_in, const hit_record &rec, color &attenuation, ray &scattered) const override {
        vec3 direction = rec.normal + random_unit_vector(); // material.h:30-42
        if (direction.near_zero()) direction = rec.normal;
        scattered = ray(rec.p, direction, r_in.time());
        attenuation = tex->value(rec.u, rec.v, rec.p);
        return true;
    }
    int record(rtamd::Recorder &r) const override {
        int id;
        if (known(r, id)) return id;
        rt_material m = {RT_MAT_LAMBERTIAN, tex->record(r), {0, 0, 0}, 0};
        return memo(r, m);
    }

  private:
    shared_ptr<texture> tex;
};

class metal : public material {
  public:
    metal(const color &albedo, float fuzz) : albedo(albedo), fuzz(fuzz < 1 ? fuzz : 1) {}
    bool scatter(const ray &r_in, const hit_record &rec, color &attenuation, ray &scattered) const override {
        vec3 mirrored = reflect(r_in.direction(), rec.normal); // material.h:53-60
        mirrored = unit_vector(mirrored) + (fuzz * random_unit_vector());
        scattered = ray(rec.p, mirrored, r_in.time());
        attenuation = albedo;
        return dot(scattered.direction(), rec.normal) > 0;
    }
    int record(rtamd::Recorder &r) const override {
        int id;
        if (known(r, id)) return id;
        rt_material m = {RT_MAT_METAL, -1, {albedo[0], albedo[1], albedo[2]}, fuzz};
        return memo(r, m);
    }

  private:
    color albedo;
    float fuzz;
};

class dielectric : public material {
  public:
    dielectric(float refraction_index) : refraction_index(refraction_index) {}
    bool scatter(const ray &r_in, const hit_record &rec, color &attenuation, ray &scattered) const override {
        attenuation = color(1.0f, 1.0f, 1.0f); // material.h:72-90
        const float ri = rec.front_face ? (1.0f / refraction_index) : refraction_index;
        const vec3 unit_direction = unit_vector(r_in.direction());
        const float cos_theta = std::fmin(dot(-unit_direction, rec.normal), 1.0f);
        const float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        // the random number is drawn only when refraction is possible (short-circuit, material.h:83)
        const bool mirror = ri * sin_theta > 1.0f || rtc::schlick(cos_theta, ri) > random_float();
        scattered = ray(rec.p, mirror ? reflect(unit_direction, rec.normal) : refract(unit_direction, rec.normal, ri),
                        r_in.time());
        return true;
    }
    int record(rtamd::Recorder &r) const override {
        int id;
        if (known(r, id)) return id;
        rt_material m = {RT_MAT_DIELECTRIC, -1, {0, 0, 0}, refraction_index};
        return memo(r, m);
    }

  private:
    float refraction_index;
};

class diffuse_light : public material {
  public:
    diffuse_light(shared_ptr<texture> tex) : tex(tex) {}
    diffuse_light(const color &emit) : tex(make_shared<solid_color>(emit)) {}
    color emitted(float u, float v, const point3 &p) const override { return tex->value(u, v, p); } // material.h:113-115
    int record(rtamd::Recorder &r) const override {
        int id;
        if (known(r, id)) return id;
        rt_material m = {RT_MAT_DIFFUSE_LIGHT, tex->record(r), {0, 0, 0}, 0};
        return memo(r, m);
    }

  private:
    shared_ptr<texture> tex;
};

// ---------------------------------------------------------------------------------------------
// hittable.h, hittable_list.h, sphere.h, quad.h, bvh.h
class hittable {
  public:
    virtual ~hittable() = default;
    // hittable.h:34-36.  Host side: the arithmetic of the kernels (rt_core.h compiled for the host),
    // so that code that calls hit() itself sees what the renderer sees.
    virtual bool hit(const ray &r, interval ray_t, hit_record &rec) const = 0;
    virtual aabb bounding_box() const = 0;
    // appends this object's primitives, in the order the reference would test them; a
    // user-defined hittable cannot be flattened
    virtual void record(rtamd::Recorder &r) const {
        (void)r;
        std::cerr << "librt_amd drop-in: a user-defined hittable cannot be rendered on the GPU\n";
        std::exit(1);
    }
    virtual void print() const {}
    virtual shared_ptr<hittable> get_left_child() const { return nullptr; }  // hittable.h:41-47
    virtual shared_ptr<hittable> get_right_child() const { return nullptr; }
};

class translate : public hittable {
  public:
    translate(shared_ptr<hittable> object, const vec3 &offset) : object(object), offset(offset) {
        bbox = object->bounding_box() + offset;
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // hittable.h:57-69
        if (!object->hit(ray(r.origin() - offset, r.direction(), r.time()), ray_t, rec)) return false;
        rec.p += offset;
        return true;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        rt_xform_op op = {RT_XF_TRANSLATE, {offset[0], offset[1], offset[2]}};
        r.open_wrappers.push_back(op);
        r.child(object.get(), [&] { object->record(r); });
        r.open_wrappers.pop_back();
    }

  private:
    shared_ptr<hittable> object;
    vec3 offset;
    aabb bbox;
};

namespace rtamd {
// Box of a rotated object as the reference computes it (hittable.h:109-130, 227-248): it
// takes max[c] = fmax(min[c], corner[c]), so "max" ends up as the larger of the running
// minimum and the LAST corner: the box is too small.  Kept as is, because the leaf order of a
// bvh_node depends on it, and that order decides ties.  In the reference the box also culls: a
// rotate_y / rotate_xyz object INSIDE a bvh_node is clipped to it there (bvh.h:78), and so it is
// by bvh_node::hit() on the host below.  The GPU renderer does not inherit this: it bounds the
// flattened primitives itself (correctly) and renders the whole object.  None of the
// reference's scenes rotates anything inside a bvh_node (main.cpp:260-269 rotate inside a
// hittable_list, which ignores boxes), so no reference image shows the difference.
template <class Rot> inline aabb rotated_box(const aabb &b, Rot rot) {
    point3 mn(infinity, infinity, infinity), mx(-infinity, -infinity, -infinity);
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++)
            for (int k = 0; k < 2; k++) {
                float x = i * b.x.max + (1 - i) * b.x.min;
                float y = j * b.y.max + (1 - j) * b.y.min;
                float z = k * b.z.max + (1 - k) * b.z.min;
                vec3 t = rot(vec3(x, y, z));
                for (int c = 0; c < 3; c++) {
                    mn[c] = fminf(mn[c], t[c]);
                    mx[c] = fmaxf(mn[c], t[c]);
                }
            }
    return aabb(mn, mx);
}
} // namespace rtamd

class rotate_y : public hittable {
  public:
    rotate_y(shared_ptr<hittable> object, float angle) : object(object), angle(angle) {
        float radians = degrees_to_radians(angle);
        float s = sinf(radians), c = cosf(radians);
        trig = rtc::f4{0.0f, 1.0f, s, c};
        bbox = rtamd::rotated_box(object->bounding_box(),
                                  [=](vec3 v) { return vec3(c * v.x() + s * v.z(), v.y(), -s * v.x() + c * v.z()); });
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // hittable.h:132-151
        const ray turned(rtamd::to_vec3(rtc::rot_y_cw(trig, rtamd::to_f3(r.origin()))),
                         rtamd::to_vec3(rtc::rot_y_cw(trig, rtamd::to_f3(r.direction()))), r.time());
        if (!object->hit(turned, ray_t, rec)) return false;
        rec.p = rtamd::to_vec3(rtc::rot_y_ccw(trig, rtamd::to_f3(rec.p)));
        rec.normal = rtamd::to_vec3(rtc::rot_y_ccw(trig, rtamd::to_f3(rec.normal)));
        return true;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        rt_xform_op op = {RT_XF_ROTATE_Y, {angle, 0.0f, 0.0f}};
        r.open_wrappers.push_back(op);
        r.child(object.get(), [&] { object->record(r); });
        r.open_wrappers.pop_back();
    }

  private:
    shared_ptr<hittable> object;
    float angle;
    rtc::f4 trig; // {-, -, sin, cos} as the kernels' wrapper ops hold them
    aabb bbox;
};

class rotate_xyz : public hittable {
  public:
    rotate_xyz(shared_ptr<hittable> object, float alpha, float beta, float gamma)
        : object(object), alpha(alpha), beta(beta), gamma(gamma) {
        float sa = sinf(degrees_to_radians(alpha)), ca = cosf(degrees_to_radians(alpha));
        float sb = sinf(degrees_to_radians(beta)), cb = cosf(degrees_to_radians(beta));
        float sg = sinf(degrees_to_radians(gamma)), cg = cosf(degrees_to_radians(gamma));
        to_object = rtc::rot_matrix_cw(sa, ca, sb, cb, sg, cg);
        to_world = rtc::rot_matrix_ccw(sa, ca, sb, cb, sg, cg);
        bbox = rtamd::rotated_box(object->bounding_box(), [=](vec3 v) { // hittable.h:284-295
            float vx = cb * cg * v.x() + (sa * sb * cg - sg * ca) * v.y() + (sb * ca * cg + sa * sg) * v.z();
            float vy = cb * sg * v.x() + (ca * cg + sa * sb * sg) * v.y() + (-sa * cg + sg * sb * ca) * v.z();
            float vz = -sb * v.x() + sa * cb * v.y() + ca * cb * v.z();
            return vec3(vx, vy, vz);
        });
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // hittable.h:251-269
        const ray turned(rtamd::to_vec3(rtc::mat_apply(to_object, rtamd::to_f3(r.origin()))),
                         rtamd::to_vec3(rtc::mat_apply(to_object, rtamd::to_f3(r.direction()))), r.time());
        if (!object->hit(turned, ray_t, rec)) return false;
        rec.p = rtamd::to_vec3(rtc::mat_apply(to_world, rtamd::to_f3(rec.p)));
        rec.normal = rtamd::to_vec3(rtc::mat_apply(to_world, rtamd::to_f3(rec.normal)));
        return true;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        rt_xform_op op = {RT_XF_ROTATE_XYZ, {alpha, beta, gamma}};
        r.open_wrappers.push_back(op);
        r.child(object.get(), [&] { object->record(r); });
        r.open_wrappers.pop_back();
    }

  private:
    shared_ptr<hittable> object;
    float alpha, beta, gamma;
    rtc::Mat3 to_object, to_world; // hittable.h:299-307 and 285-293, row by row
    aabb bbox;
};

class scaling : public hittable {
  public:
    scaling(shared_ptr<hittable> object, const vec3 &scale) : object(object), scale(scale) {
        bbox = object->bounding_box() * scale;
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // hittable.h:319-334
        if (!object->hit(ray(r.origin() / scale, r.direction() / scale, r.time()), ray_t, rec)) return false;
        rec.p *= scale;
        rec.normal = unit_vector(rec.normal / scale);
        return true;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        rt_xform_op op = {RT_XF_SCALE, {scale[0], scale[1], scale[2]}};
        r.open_wrappers.push_back(op);
        r.child(object.get(), [&] { object->record(r); });
        r.open_wrappers.pop_back();
    }

  private:
    shared_ptr<hittable> object;
    vec3 scale;
    aabb bbox;
};

class hittable_list : public hittable {
  public:
    std::vector<shared_ptr<hittable>> objects;
    hittable_list() {}
    hittable_list(shared_ptr<hittable> object) { add(object); }
    void clear() { objects.clear(); }
    void add(shared_ptr<hittable> object) {
        objects.push_back(object);
        bbox = aabb(bbox, object->bounding_box());
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // hittable_list.h:29-43
        hit_record candidate;
        bool any = false;
        for (const auto &o : objects)
            if (o->hit(r, interval(ray_t.min, any ? rec.t : ray_t.max), candidate)) {
                any = true;
                rec = candidate;
            }
        return any;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        for (const auto &o : objects) o->record(r);
    }
    void print() const override {
        for (const auto &o : objects) o->print();
    }

  private:
    aabb bbox;
};

class sphere : public hittable {
  public:
    sphere(const point3 &center, float radius, shared_ptr<material> mat)
        : center1(center), center2(center), radius(std::fmax(0.0f, radius)), mat(mat), is_moving(false) {
        vec3 rvec(radius, radius, radius);
        bbox = aabb(center1 - rvec, center1 + rvec);
    }
    sphere(const point3 &c1, const point3 &c2, float radius, shared_ptr<material> mat)
        : center1(c1), center2(c2), radius(std::fmax(0.0f, radius)), mat(mat), is_moving(true) {
        vec3 rvec(radius, radius, radius);
        bbox = aabb(aabb(c1 - rvec, c1 + rvec), aabb(c2 - rvec, c2 + rvec));
        center_vec = c2 - c1;
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // sphere.h:28-74
        const rtc::f4 g0{center1[0], center1[1], center1[2], radius};
        const rtc::f4 g1{is_moving ? center_vec[0] : 0.0f, is_moving ? center_vec[1] : 0.0f,
                         is_moving ? center_vec[2] : 0.0f, 0.0f};
        const rtc::f3 o = rtamd::to_f3(r.origin()), d = rtamd::to_f3(r.direction());
        float t;
        if (!rtc::sphere_test(g0, g1, o, d, rtc::len2(d), r.time(), ray_t.min, ray_t.max, t)) return false;
        rec.t = t;
        rec.p = r.at(t);
        const vec3 centre = rtamd::to_vec3(rtc::xyz(g0) + r.time() * rtc::xyz(g1));
        const vec3 outward = (rec.p - centre) / radius;
        rec.set_face_normal(r, outward);
        rtc::sphere_uv(rtamd::to_f3(outward), rec.u, rec.v); // sphere.h:96-113
        rec.mat = mat;
        return true;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        rt_prim p;
        memset(&p, 0, sizeof(p));
        if (r.counting) return r.add_prim(p); // (the counting pass only counts)
        p.type = RT_PRIM_SPHERE;
        p.material = mat->record(r);
        p.flags = is_moving ? RT_PRIM_FLAG_MOVING : 0;
        for (int k = 0; k < 3; k++) {
            p.a[k] = center1[k];
            p.b[k] = is_moving ? center_vec[k] : 0.0f;
            p.d[k] = is_moving ? center2[k] : 0.0f;
        }
        p.c[0] = radius;
        r.add_prim(p);
    }
    void print() const override { std::cout << bbox.x.min << " " << bbox.y.min << " " << bbox.z.min << std::endl; }

  private:
    point3 center1, center2;
    float radius;
    shared_ptr<material> mat;
    bool is_moving;
    vec3 center_vec;
    aabb bbox;
};

class quad : public hittable {
  public:
    quad(const point3 &Q, const point3 &u, const point3 &v, shared_ptr<material> mat) : Q(Q), u(u), v(v), mat(mat) {
        bbox = aabb(aabb(Q, Q + u + v), aabb(Q + u, Q + v)); // quad.h:25-30
        const vec3 n = cross(u, v);                          // quad.h:15-20
        normal = unit_vector(n);
        D = dot(normal, Q);
        w = n / dot(n, n);
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // quad.h:32-91
        const rtc::f4 g0{normal[0], normal[1], normal[2], D}, g1{Q[0], Q[1], Q[2], w[0]};
        const rtc::f4 g2{w[1], w[2], u[0], u[1]}, g3{u[2], v[0], v[1], v[2]};
        float t;
        if (!rtc::quad_test(g0, g1, g2, g3, rtamd::to_f3(r.origin()), rtamd::to_f3(r.direction()), ray_t.min,
                            ray_t.max, t))
            return false;
        rec.t = t;
        rec.p = r.at(t);
        const vec3 planar = rec.p - Q;
        rec.u = dot(w, cross(planar, v));
        rec.v = dot(w, cross(u, planar));
        rec.mat = mat;
        rec.set_face_normal(r, normal);
        return true;
    }
    aabb bounding_box() const override { return bbox; }
    void record(rtamd::Recorder &r) const override {
        rt_prim p;
        memset(&p, 0, sizeof(p));
        if (r.counting) return r.add_prim(p);
        p.type = RT_PRIM_QUAD;
        p.material = mat->record(r);
        for (int k = 0; k < 3; k++) {
            p.a[k] = Q[k];
            p.b[k] = u[k];
            p.c[k] = v[k];
        }
        r.add_prim(p);
    }
    void print() const override { std::cout << bbox.x.min << " " << bbox.y.min << " " << bbox.z.min << std::endl; }

  private:
    point3 Q;
    vec3 u, v;
    vec3 normal, w;
    float D;
    shared_ptr<material> mat;
    aabb bbox;
};

inline shared_ptr<hittable_list> box(const point3 &a, const point3 &b, shared_ptr<material> mat) { // quad.h:111-131
    auto sides = make_shared<hittable_list>();
    point3 lo(std::fmin(a.x(), b.x()), std::fmin(a.y(), b.y()), std::fmin(a.z(), b.z()));
    point3 hi(std::fmax(a.x(), b.x()), std::fmax(a.y(), b.y()), std::fmax(a.z(), b.z()));
    vec3 dx(hi.x() - lo.x(), 0.0f, 0.0f), dy(0.0f, hi.y() - lo.y(), 0.0f), dz(0.0f, 0.0f, hi.z() - lo.z());
    const struct {
        point3 corner;
        vec3 e0, e1;
    } faces[6] = {{point3(lo.x(), lo.y(), hi.z()), dx, dy},  {point3(hi.x(), lo.y(), hi.z()), -dz, dy},
                  {point3(hi.x(), lo.y(), lo.z()), -dx, dy}, {point3(lo.x(), lo.y(), lo.z()), dz, dy},
                  {point3(lo.x(), hi.y(), hi.z()), dx, -dz}, {point3(lo.x(), lo.y(), lo.z()), dx, dz}};
    for (const auto &f : faces) sides->add(make_shared<quad>(f.corner, f.e0, f.e1, mat));
    return sides;
}

// bvh.h.  The renderer builds its own tree over the flattened primitives; what it takes from
// this class is the ORDER in which the reference's tree would test its leaves (the recursive
// std::sort by box minimum along the longest axis, bvh.h:48-75), because that order decides
// which of two coincident primitives wins.  The reference's constructor is the slow part of a
// large scene (10 s for 10^6 primitives: every comparison copies two shared_ptr and makes two
// virtual calls, every level sorts again), so the order is computed here on plain arrays: the
// boxes are fetched once, every range is sorted as {key, index} pairs by the same std::sort with
// the same comparison outcomes (so the permutation is the one the reference's sort produces,
// whatever it does with equal keys), and the two halves of a large range are sorted on two
// threads.  The tree itself, for host code that calls hit() or walks get_left_child() /
// get_right_child(), is made of the same ranges and is built node by node when first asked for.
class bvh_node : public hittable {
  public:
    bvh_node(hittable_list list) : bvh_node(list.objects, 0, list.objects.size()) {}
    // (like the reference's, this constructor leaves objects[start, end) in the tree's leaf order)
    bvh_node(std::vector<shared_ptr<hittable>> &objects, size_t start, size_t end) {
        const size_t n = end > start ? end - start : 0;
        std::vector<aabb> boxes(n);
        for (size_t i = 0; i < n; i++) boxes[i] = objects[start + i]->bounding_box();
        std::vector<Key> keys(n);
        for (size_t i = 0; i < n; i++) keys[i].index = (uint32_t)i;
        order_range(boxes, keys.data(), 0, n, 0);
        auto sh = make_shared<Ranges>();
        sh->leaves.resize(n);
        sh->boxes.resize(n);
        for (size_t i = 0; i < n; i++) {
            sh->leaves[i] = objects[start + keys[i].index];
            sh->boxes[i] = boxes[keys[i].index];
        }
        std::copy(sh->leaves.begin(), sh->leaves.end(), objects.begin() + start);
        all = sh;
        first = 0;
        last = n;
        bbox = box_of(sh->boxes, 0, n);
    }
    bool hit(const ray &r, interval ray_t, hit_record &rec) const override { // bvh.h:77-88
        children();
        if (!left || !bbox.hit(r, ray_t)) return false;
        const bool hit_left = left->hit(r, ray_t, rec);
        const bool hit_right = right->hit(r, interval(ray_t.min, hit_left ? rec.t : ray_t.max), rec);
        return hit_left || hit_right;
    }
    aabb bounding_box() const override { return bbox; }
    shared_ptr<hittable> get_left_child() const override { // bvh.h:143-149
        children();
        return left;
    }
    shared_ptr<hittable> get_right_child() const override {
        children();
        return right;
    }
    void record(rtamd::Recorder &r) const override {
        r.begin_bvh();
        for (size_t i = first; i < last; i++) all->leaves[i]->record(r); // left to right, each leaf once
        r.end_bvh();
    }
    void print() const override {
        for (size_t i = first; i < last; i++) all->leaves[i]->print();
    }

  private:
    struct Key {
        float key;
        uint32_t index;
    };
    struct Ranges { // the whole tree's leaves in their final order, and their boxes
        std::vector<shared_ptr<hittable>> leaves;
        std::vector<aabb> boxes;
    };
    // a node below the root: the range [first, last) of the root's leaves (bvh.h:54-71 splits at the middle)
    bvh_node(shared_ptr<const Ranges> ranges, size_t first_leaf, size_t last_leaf)
        : all(std::move(ranges)), first(first_leaf), last(last_leaf) {
        bbox = box_of(all->boxes, first, last);
    }
    static aabb box_of(const std::vector<aabb> &boxes, size_t a, size_t b) { // bvh.h:50-52
        aabb box = aabb::empty;
        for (size_t i = a; i < b; i++) box = aabb(box, boxes[i]);
        return box;
    }
    // bvh.h:48-75 on indices: keys[a, b) holds the indices of the range in their current order
    static void order_range(const std::vector<aabb> &boxes, Key *keys, size_t a, size_t b, int forks) {
        const size_t span = b - a;
        if (span <= 2) return;
        aabb box = aabb::empty;
        for (size_t i = a; i < b; i++) box = aabb(box, boxes[keys[i].index]);
        const int axis = box.longest_axis();
        for (size_t i = a; i < b; i++) keys[i].key = boxes[keys[i].index].axis_interval(axis).min;
        std::sort(keys + a, keys + b, [](const Key &x, const Key &y) { return x.key < y.key; });
        const size_t mid = a + span / 2;
        if (span >= 32768 && forks < 4) { // the halves are independent: up to 16 threads
            std::thread other(order_range, std::cref(boxes), keys, a, mid, forks + 1);
            order_range(boxes, keys, mid, b, forks + 1);
            other.join();
        } else {
            order_range(boxes, keys, a, mid, forks);
            order_range(boxes, keys, mid, b, forks);
        }
    }
    void children() const {
        if (built.load(std::memory_order_acquire)) return;
        static std::mutex building; // (one for all nodes: a node is built once, in microseconds)
        std::lock_guard<std::mutex> lock(building);
        if (built.load(std::memory_order_relaxed)) return;
        const size_t span = last - first;
        if (span == 1) {
            left = right = all->leaves[first];
        } else if (span == 2) {
            left = all->leaves[first];
            right = all->leaves[first + 1];
        } else if (span > 2) {
            const size_t mid = first + span / 2;
            left = shared_ptr<bvh_node>(new bvh_node(all, first, mid));
            right = shared_ptr<bvh_node>(new bvh_node(all, mid, last));
        }
        built.store(true, std::memory_order_release);
    }

    shared_ptr<const Ranges> all;
    size_t first = 0, last = 0;
    aabb bbox;
    mutable std::atomic<bool> built{false};
    mutable shared_ptr<hittable> left, right;

  public:
    // copyable like the reference's class (a copy builds its own children when asked)
    bvh_node(const bvh_node &o) : hittable(o), all(o.all), first(o.first), last(o.last), bbox(o.bbox) {}
    bvh_node &operator=(const bvh_node &o) {
        if (this != &o) {
            all = o.all, first = o.first, last = o.last, bbox = o.bbox;
            left.reset(), right.reset();
            built.store(false, std::memory_order_release);
        }
        return *this;
    }
};

// ---------------------------------------------------------------------------------------------
// camera.h
class camera {
  public:
    float aspect_ratio = 1.0f;
    int image_width = 100;
    int samples_per_pixel = 10;
    int max_depth = 10;
    color background;
    float vfov = 90.0f;
    point3 lookfrom = point3(0.0f, 0.0f, 0.0f);
    point3 lookat = point3(0.0f, 0.0f, -1.0f);
    vec3 vup = vec3(0.0f, 1.0f, 0.0f);
    float defocus_angle = 0.0f;
    float focus_dist = 10.0f;

    // Extensions (not in the reference): which GPU renders, and the sampler seed.
    int device = 0;
    uint32_t seed = 0;
    // Several GPUs of one node: with more than one entry here (or in the environment variable
    // RTAMD_DEVICES, e.g. "0,1,2,3,4,5,6,7") the samples per pixel are split across these devices
    // and the per-device images are summed on the first one (rt_multi_render).
    std::vector<int> devices;
    // Progressive rendering (not in the reference, which writes the image once at the end,
    // camera.h:56): with pass_samples > 0 the image is accumulated on the GPU in passes of that
    // many samples per pixel; after every write_every passes the PNG is rewritten from the
    // samples so far (encoded on the host while the GPU renders the next pass), and with a
    // checkpoint file the sums are saved beside it, so that a later render() of the same
    // image continues from there.  Environment overrides: RTAMD_PASS_SPP, RTAMD_WRITE_EVERY,
    // RTAMD_CHECKPOINT.
    int pass_samples = 0;
    int write_every = 1;
    std::string checkpoint;

    rt_camera_desc descriptor() const {
        rt_camera_desc c;
        memset(&c, 0, sizeof(c));
        c.aspect_ratio = aspect_ratio;
        c.image_width = image_width;
        c.samples_per_pixel = samples_per_pixel;
        c.max_depth = max_depth;
        c.vfov = vfov;
        c.defocus_angle = defocus_angle;
        c.focus_dist = focus_dist;
        for (int k = 0; k < 3; k++) {
            c.background[k] = background[k];
            c.lookfrom[k] = lookfrom[k];
            c.lookat[k] = lookat[k];
            c.vup[k] = vup[k];
        }
        return c;
    }

    // Replaces camera::render (camera.h:31-63): same arguments, same outputs (a PNG at
    // file_name, progress on clog, "Render took ... seconds." on cerr).  The sample loop runs
    // on the GPU.  Like the reference it reports no errors to the caller; a failure of the
    // library is printed and the process exits, since no image can be produced.
    void render(const hittable &world, const char *file_name) {
        auto t0 = std::chrono::steady_clock::now();
        // RTAMD_TIMING=1: where the time of this call goes, on stderr
        const bool timing = getenv("RTAMD_TIMING") && atoi(getenv("RTAMD_TIMING")) != 0;
        auto t_lap = t0;
        auto lap = [&](const char *what) {
            if (!timing) return;
            auto now = std::chrono::steady_clock::now();
            std::cerr << "[rtamd] " << what << ": " << std::chrono::duration<double, std::milli>(now - t_lap).count() << " ms\n";
            t_lap = now;
        };
        // objects that several wrappers share become sub-scenes with instances (RTAMD_INSTANCES=0:
        // everything written out, as before); a scene that shares nothing records exactly as it did
        rtamd::Recorder counted, rec;
        const char *inst_env = getenv("RTAMD_INSTANCES");
        if (!inst_env || atoi(inst_env) != 0) {
            counted.counting = true;
            world.record(counted);
            if (counted.any_shared()) rec.shared = &counted.seen;
            rec.prims.reserve(counted.n_counted); // (at most: shared objects are recorded once)
        }
        world.record(rec);
        rt_scene_desc sd = rec.desc();
        rt_camera_desc cd = descriptor();
        lap("scene graph recorded");
        const char *dump = getenv("RTAMD_DUMP_SCENE"); // fixture generation: write the flat scene and stop
        if (dump) {
            if (rt_scene_file_save(&sd, dump) != RT_OK) die("rt_scene_file_save");
            std::string camfile = std::string(dump) + ".cam";
            rt_camera_frame fr;
            if (rt_camera_initialize(&cd, &fr) != RT_OK) die("rt_camera_initialize");
            FILE *f = fopen(camfile.c_str(), "wb");
            if (!f || fwrite(&cd, sizeof(cd), 1, f) != 1 || fwrite(&fr, sizeof(fr), 1, f) != 1) die("camera dump");
            fclose(f);
            return;
        }
        rt_camera_frame frame;
        if (rt_camera_initialize(&cd, &frame) != RT_OK) die("rt_camera_initialize");
        const int W = frame.image_width, H = frame.image_height;
        std::vector<int> devs = devices;
        if (const char *e = getenv("RTAMD_DEVICES")) {
            devs.clear();
            for (const char *p = e; *p;) {
                char *end = nullptr;
                long v = strtol(p, &end, 10);
                if (end == p) break;
                devs.push_back((int)v);
                p = *end == ',' ? end + 1 : end;
            }
        }
        if (devs.size() > 1) { // the sample loop split over several GPUs
            std::clog << "\rScan lines remaining: " << H << ' ' << std::flush;
            rt_multi *multi = nullptr;
            std::vector<int32_t> d32(devs.begin(), devs.end());
            if (rt_multi_create(&sd, d32.data(), (int32_t)d32.size(), 0, &multi) != RT_OK) die("rt_multi_create");
            std::vector<float> accum((size_t)W * H * 3, 0.0f);
            rt_render_opts opts;
            memset(&opts, 0, sizeof(opts));
            opts.sample_count = samples_per_pixel;
            opts.seed = seed;
            opts.mode = RT_MODE_MEGAKERNEL;
            rt_render_stats mstats;
            if (rt_multi_render(multi, &cd, &opts, accum.data(), &mstats) != RT_OK) die("rt_multi_render");
            std::vector<uint8_t> rgb((size_t)W * H * 3);
            if (rt_tonemap_u8(accum.data(), (int64_t)W * H, frame.pixel_samples_scale, rgb.data()) != RT_OK)
                die("rt_tonemap_u8");
            if (!rtamd::write_png_rgb8(file_name, W, H, rgb.data()))
                std::cerr << "ERROR: could not write '" << file_name << "'.\n";
            rt_multi_destroy(multi);
            std::clog << "\rDone.\n";
            double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cerr << "Render took " << seconds << " seconds.\n";
            last_stats = mstats;
            return;
        }
        rt_scene *scene = nullptr;
        if (rt_scene_create(&sd, devs.size() == 1 ? devs[0] : device, 0, &scene) != RT_OK) die("rt_scene_create");
        lap("rt_scene_create (tree, upload; the first call also starts the HIP runtime)");
        int pass = pass_samples, every = write_every;
        std::string ckpt = checkpoint;
        if (const char *e = getenv("RTAMD_PASS_SPP")) pass = atoi(e);
        if (const char *e = getenv("RTAMD_WRITE_EVERY")) every = atoi(e);
        if (const char *e = getenv("RTAMD_CHECKPOINT")) ckpt = e;
        std::clog << "\rScan lines remaining: " << H << ' ' << std::flush;
        rt_render_stats stats;
        memset(&stats, 0, sizeof(stats));
        if (pass > 0) {
            render_in_passes(scene, cd, W, H, pass, every < 1 ? 1 : every, ckpt, file_name);
        } else {
            std::vector<float> accum((size_t)W * H * 3, 0.0f);
            rt_render_opts opts;
            memset(&opts, 0, sizeof(opts));
            opts.sample_begin = 0;
            opts.sample_count = samples_per_pixel;
            opts.seed = seed;
            opts.mode = RT_MODE_MEGAKERNEL;
            if (rt_render(scene, &cd, &opts, accum.data(), &stats) != RT_OK) die("rt_render");
            lap("rt_render");
            std::vector<uint8_t> rgb((size_t)W * H * 3);
            if (rt_tonemap_u8(accum.data(), (int64_t)W * H, frame.pixel_samples_scale, rgb.data()) != RT_OK)
                die("rt_tonemap_u8");
            if (!rtamd::write_png_rgb8(file_name, W, H, rgb.data()))
                std::cerr << "ERROR: could not write '" << file_name << "'.\n";
            lap("tone map and PNG");
        }
        rt_scene_destroy(scene);
        std::clog << "\rDone.\n";
        double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cerr << "Render took " << seconds << " seconds.\n";
        last_stats = stats;
    }
    rt_render_stats last_stats = {};

    // Checkpoint file of a progressive render: "RTACCUM1", width, height (int32), samples
    // (int64), seed (uint32), 0 (uint32), then width*height*3 floats (radiance sums).
    static bool save_checkpoint(const std::string &path, int W, int H, int64_t samples, uint32_t seed,
                                const float *accum) {
        std::string tmp = path + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) return false;
        int32_t wh[2] = {W, H};
        uint32_t tail[2] = {seed, 0u};
        bool ok = fwrite("RTACCUM1", 1, 8, f) == 8 && fwrite(wh, 4, 2, f) == 2 && fwrite(&samples, 8, 1, f) == 1 &&
                  fwrite(tail, 4, 2, f) == 2 && fwrite(accum, sizeof(float), (size_t)W * H * 3, f) == (size_t)W * H * 3;
        ok = fclose(f) == 0 && ok;
        return ok && rename(tmp.c_str(), path.c_str()) == 0; // a reader never sees half a file
    }
    static bool load_checkpoint(const std::string &path, int W, int H, uint32_t seed, int64_t &samples,
                                std::vector<float> &accum) {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        char magic[8];
        int32_t wh[2];
        uint32_t tail[2];
        accum.resize((size_t)W * H * 3);
        bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "RTACCUM1", 8) == 0 && fread(wh, 4, 2, f) == 2 &&
                  wh[0] == W && wh[1] == H && fread(&samples, 8, 1, f) == 1 && fread(tail, 4, 2, f) == 2 &&
                  tail[0] == seed && samples >= 0 &&
                  fread(accum.data(), sizeof(float), accum.size(), f) == accum.size();
        fclose(f);
        return ok;
    }

  private:
    void render_in_passes(rt_scene *scene, const rt_camera_desc &cd, int W, int H, int pass, int every,
                          const std::string &ckpt, const char *file_name) {
        rt_film *film = nullptr;
        if (rt_film_create(device, W, H, &film) != RT_OK) die("rt_film_create");
        int64_t done = 0;
        std::vector<float> sums;
        if (!ckpt.empty() && load_checkpoint(ckpt, W, H, seed, done, sums) && done <= samples_per_pixel) {
            if (rt_film_upload(film, sums.data(), done) != RT_OK) die("rt_film_upload");
            std::clog << "\rResuming at sample " << done << " of " << samples_per_pixel << ".\n";
        } else {
            done = 0;
        }
        std::vector<uint8_t> rgb((size_t)W * H * 3);
        if (!ckpt.empty()) sums.resize((size_t)W * H * 3);
        bool pending = false;
        // what the host does with a finished snapshot, while the GPU renders on
        auto finish_snapshot = [&]() {
            int64_t n = 0;
            if (rt_film_wait(film, rgb.data(), ckpt.empty() ? nullptr : sums.data(), &n) != RT_OK) die("rt_film_wait");
            if (!rtamd::write_png_rgb8(file_name, W, H, rgb.data()))
                std::cerr << "ERROR: could not write '" << file_name << "'.\n";
            if (!ckpt.empty() && !save_checkpoint(ckpt, W, H, n, seed, sums.data()))
                std::cerr << "ERROR: could not write '" << ckpt << "'.\n";
            std::clog << "\rSamples per pixel done: " << n << " of " << samples_per_pixel << ' ' << std::flush;
            pending = false;
        };
        if (done >= samples_per_pixel && done > 0) { // nothing left to render: rewrite the image
            if (rt_film_snapshot(film, ckpt.empty() ? 0 : 1) != RT_OK) die("rt_film_snapshot");
            pending = true;
        }
        for (int k = 1; done < samples_per_pixel; k++) {
            rt_render_opts opts;
            memset(&opts, 0, sizeof(opts));
            opts.sample_begin = (int32_t)done;
            opts.sample_count = (int32_t)std::min<int64_t>(pass, samples_per_pixel - done);
            opts.seed = seed;
            opts.mode = RT_MODE_MEGAKERNEL;
            if (rt_film_render(scene, film, &cd, &opts) != RT_OK) die("rt_film_render");
            done += opts.sample_count;
            if (pending) finish_snapshot(); // the previous snapshot, beside the pass just enqueued
            if (k % every == 0 || done >= samples_per_pixel) {
                if (rt_film_snapshot(film, ckpt.empty() ? 0 : 1) != RT_OK) die("rt_film_snapshot");
                pending = true;
            }
        }
        if (pending) finish_snapshot();
        rt_film_destroy(film);
    }

    [[noreturn]] static void die(const char *what) {
        std::cerr << "rtamd: " << what << " failed: " << rt_last_error() << "\n";
        exit(1);
    }
};
