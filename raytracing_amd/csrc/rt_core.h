// rt_core.h — device-side core of the path tracer: math, sampler, intersection,
// traversal, surface reconstruction, shading and the per-ray path step.
//
// Everything here is a plain inlined function over POD types so that the HIP
// kernels (rt_kernels.hip) instantiate it for gfx950.  The arithmetic that
// decides hit / miss / closest primitive is written operation for operation
// like the reference (compile with -ffp-contract=off: a contracted h*h - a*c
// changes t far beyond 1e-5 on the r = 1000 ground sphere); FMAs appear only
// where they are written explicitly (bounding-box culling, which is
// conservative by construction).
#pragma once

#include "rt_amd.h"

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rtc {

// ------------------------------------------------------------------------------------------
// POD vector types (own types: identical layout on host and device)
struct alignas(16) f4 {
    float x, y, z, w;
};
struct alignas(16) i4 {
    int x, y, z, w;
};
struct f3 {
    float x, y, z;
};

RT_HD f3 mk(float x, float y, float z) {
    f3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
RT_HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator/(f3 a, f3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
RT_HD f3 operator*(float t, f3 v) { return mk(t * v.x, t * v.y, t * v.z); }
RT_HD f3 operator*(f3 v, float t) { return t * v; }
RT_HD f3 operator/(f3 v, float t) { return (1.0f / t) * v; } // vec3.h:107: multiply by the reciprocal
RT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_HD float len(f3 a) { return sqrtf(len2(a)); }
RT_HD f3 cross(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_HD f3 unit(f3 v) { return v / len(v); }
RT_HD f3 xyz(f4 v) { return mk(v.x, v.y, v.z); }

// Shading arithmetic.  What happens to a path after its closest hit is known (the hit point and
// normal of the winner, scatter directions, textures, camera rays) decides no hit, miss or
// closest primitive, and the sampler's stream differs from the reference's anyway, so only its
// distribution matters.  On the device that code fuses multiply-adds and uses the hardware's
// 1-ulp reciprocal, square root and reciprocal square root (a correctly rounded sqrtf is 17
// instructions, a correctly rounded division 10, many of them quarter rate).  Host builds keep the
// plain operations, in the same order.  Off by default (-DRT_FAST_SHADE=1 turns it on): the kernel
// gains 3 %, and paths through glass and mirrors, which otherwise follow the CPU restatement bit for
// bit, then differ from it in the last bits of their directions, which a high-frequency texture far
// away turns into visibly different texels (same distribution, different pixels).
#ifndef RT_FAST_SHADE
#define RT_FAST_SHADE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && RT_FAST_SHADE
#define RT_SHADE_FUSED 1
#else
#define RT_SHADE_FUSED 0
#endif
RT_HD float sh_fma(float a, float b, float c) { // a * b + c
#if RT_SHADE_FUSED
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
RT_HD float sh_sqrt(float x) {
#if RT_SHADE_FUSED
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
RT_HD float sh_dot(f3 a, f3 b) { return sh_fma(a.z, b.z, sh_fma(a.y, b.y, a.x * b.x)); }
RT_HD f3 sh_madd(float t, f3 v, f3 w) { return mk(sh_fma(t, v.x, w.x), sh_fma(t, v.y, w.y), sh_fma(t, v.z, w.z)); } // t * v + w
RT_HD f3 sh_madd3(f3 a, f3 v, f3 w) { return mk(sh_fma(a.x, v.x, w.x), sh_fma(a.y, v.y, w.y), sh_fma(a.z, v.z, w.z)); }
RT_HD f3 sh_unit(f3 v) { // vec3.h:124-126
#if RT_SHADE_FUSED
    return __builtin_amdgcn_rsqf(sh_dot(v, v)) * v;
#else
    return v / len(v);
#endif
}

// Device code can be compiled without the quad, Perlin-noise, image-texture and instance-wrapper
// branches (RT_LEAN: the "lean" render kernels for scenes that have none of them; fewer registers and
// a shorter shading phase).  Host code always has every feature.
#if defined(RT_LEAN) && defined(__HIP_DEVICE_COMPILE__)
#define RT_FEAT_QUAD 0
#define RT_FEAT_NOISE 0
#define RT_FEAT_IMAGE 0
#define RT_FEAT_CHAIN 0
#define RT_FEAT_INSTANCE 0
#else
#define RT_FEAT_QUAD 1
#define RT_FEAT_NOISE 1
#define RT_FEAT_IMAGE 1
#define RT_FEAT_CHAIN 1
// the two-level traversal of scenes with instances: device code has it only in the build made for
// such scenes (RT_INST): its checks cost the 10^6-primitive scene 3.5 % when they sat in every kernel
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RT_INST)
#define RT_FEAT_INSTANCE 0
#else
#define RT_FEAT_INSTANCE 1
#endif
#endif
// True if the condition holds in any lane of the wave (host code: the one "lane" there is).
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_WAVE_ANY(x) (__builtin_amdgcn_ballot_w64(x) != 0ull)
#else
#define RT_WAVE_ANY(x) (x)
#endif
// -DRT_ISA_MARKS: comment lines in the generated assembly at the phase boundaries of the render
// kernels, for tools/opcode_hist.py (they emit no instruction, but they pin the order of the code around
// them: not for production builds).
#if defined(RT_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define RT_MARK(name) asm volatile("; RT_MARK " name ::: "memory")
#else
#define RT_MARK(name) ((void)0)
#endif
#define RT_PI 3.1415926535897932385f
#define RT_INF __builtin_huge_valf()

struct Ray {
    f3 o, d;
    float tm;
};

// Pointers for data every lane of a wave reads at the same address (the primitive list of a
// scene that is scanned, the wrapper ops of its chains): in device code they point into the
// constant address space, so that the compiler fetches through the scalar cache into SGPRs
// (s_load) and branches on the values with scalar branches.  Read-only data only.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_UNIFORM_AS __attribute__((address_space(4)))
#else
#define RT_UNIFORM_AS
#endif
typedef const RT_UNIFORM_AS f4 *uf4_ptr;
typedef const RT_UNIFORM_AS i4 *ui4_ptr;
typedef const RT_UNIFORM_AS int *uint_ptr;

// perlin.h:55-56 as the kernels read it: the 256 gradient vectors padded to 16 bytes (one
// ds_read_b128 / global_load_dwordx4 each) and perm_x.
struct DevPerlin {
    f4 randvec[256];
    int perm_x[256];
};

// ------------------------------------------------------------------------------------------
// Scene as the kernels see it.  All pointers are device pointers (the kernels
// may substitute LDS copies of the first five through a SceneView).
//
// slots: one per primitive, slot_f4 x 16 bytes each.
//   sphere: {c1.x c1.y c1.z radius} {cv.x cv.y cv.z 0}
//   quad  : {n.x n.y n.z D} {Q.x Q.y Q.z w.x} {w.y w.z u.x u.y} {u.z v.x v.y v.z}
// rec_i: per slot {kind, texture, chain (-1 none), prim id}
//   kind = prim type | material type << 2 | texture type << 4 (texture type of the
//   material's texture; a solid colour is inlined into rec_f and texture = -1)
// rec_f: per slot {r g b param}: lambertian / light with a solid texture: the colour;
//   metal: albedo + fuzz; dielectric: {1/ri, Schlick's r0 from outside, r0 from inside, ri}
//   (ri = refraction index; the attenuation (1,1,1) is implied)
// nodes: 4-wide BVH, the child boxes stored in the parent, 128 bytes (layout: trav_inner);
//   ref >= 0: inner node index; ref < 0: leaf, ~ref = slot*2 + type
// textures: only the textures reachable from a non-solid material texture (re-indexed)
// xf   : RT_XF_F4 x 16 bytes per wrapper op: {type p0 p1 p2} {sa ca sb cb}, then the two
//   rotation matrices of rotate_xyz row by row, world -> object (hittable.h:299-307) and object ->
//   world (hittable.h:285-293): {cw0 cw1 cw2 cw3} {cw4 cw5 cw6 cw7} {cw8 ccw0 ccw1 ccw2}
//   {ccw3 ccw4 ccw5 ccw6} {ccw7 ccw8 - -}.  The entries are the parenthesised products of sines
//   and cosines the reference evaluates per call, computed once in the same fp32 order.
enum { TRAV_LINEAR = 0, TRAV_BVH = 1 };

#define RT_XF_F4 7
// A node is 8 x 16 bytes in global memory.  Copies in LDS are spaced RT_NODE_LDS_F4 x 16 bytes
// apart: at 128 bytes every node would start in bank 0 and the 64 lanes of a wave, each reading
// the same word of a different node, would all hit the same four banks.
#define RT_NODE_F4 8
#ifndef RT_NODE_LDS_F4
#define RT_NODE_LDS_F4 9
#endif
#define RT_KIND_PRIM(k) ((k)&1)
#define RT_KIND_MAT(k) (((k) >> 2) & 3)
#define RT_KIND_TEX(k) (((k) >> 4) & 3)

struct DevScene {
    const f4 *slots;
    const f4 *nodes;
    const i4 *rec_i;
    const f4 *rec_f;
    const rt_texture *textures;
    const f4 *xf;
    const int *chain_first; // chain c covers ops [chain_first[c], chain_first[c+1])
    const f4 *scan;         // linear scenes: RT_SCAN_F4 words per slot, what the scan needs of it (trace_linear)
    const DevPerlin *perlins;
    const rt_image *images;
    const uint8_t *texels;
    // two-level scenes: per instance {root reference of the sub-scene's tree, wrapper chain (-1: none),
    // offset of its primitives' ids among all primitives of the world, -}.  Instance k is the leaf
    // "slot" n_slots + k of the top tree; slot n_slots + n_inst marks the end of a visit on the stack.
    const i4 *inst;
    int n_inst;
    // Primitives whose box covers most of the scene (a ground sphere) are not in the tree: they are
    // slots [0, n_big) and every ray tests them before it starts its walk (bit j of big_quads: slot
    // j is a quad).  Every ray would reach their leaf anyway; tested first, their hit culls from the
    // first node on, and the test runs when all lanes of a wave start rays together.
    int n_big;
    int big_quads;
    int n_slots;
    int n_nodes;
    int n_textures;
    int slot_f4;   // 2 (spheres only) or 4
    int traversal; // TRAV_*
    int n_chains;
    int n_xf_ops;
    int lds_records; // LDS staging also covers rec_i / rec_f / textures (else only slots + nodes)
    int n_perlins;
    int lds_perlins; // ... and the Perlin tables
    int n_top;       // scenes that stay in HBM: nodes [0, n_top) are also kept in LDS (the top of the tree)
    int ref_bits;    // low bits of a stack entry that hold a child reference (16 with 16-bit stacks)
    int stack_cap;   // entries a traversal stack needs (RT_STACK_CAP of the tree's inner levels)
};

// The five hot arrays, wherever they currently live (HBM or this workgroup's LDS).
struct SceneView {
    const f4 *slots;
    const f4 *nodes;
    const i4 *rec_i;
    const f4 *rec_f;
    const rt_texture *textures;
    const DevPerlin *perlins;
    const f4 *top; // LDS copy of nodes [0, n_top) when the scene itself stays in HBM
    int n_top;
};
RT_HD SceneView global_view(const DevScene &sc) {
    SceneView v;
    v.slots = sc.slots;
    v.nodes = sc.nodes;
    v.rec_i = sc.rec_i;
    v.rec_f = sc.rec_f;
    v.textures = sc.textures;
    v.perlins = sc.perlins;
    v.top = nullptr;
    v.n_top = 0;
    return v;
}

struct DevCamera {
    rt_camera_frame f;
    float background[3];
    float defocus_angle;
    int max_depth;
};

struct Counters {
    unsigned long long box_tests, sphere_tests, sphere_hits, quad_tests;
    unsigned long long shades[4];
    unsigned long long noise_evals, image_evals, checker_evals;
    // wave-level schedule of the decoupled kernel: inner-block executions and the lanes
    // active in them, leaf-block executions / lanes, shading phases / lanes
    unsigned long long sched[6];
    unsigned long long cycles[6]; // wave cycles: whole kernel, shading, starting paths, node steps, leaf tests
    // wall clock (100 MHz ticks): earliest wave start, first / last moment a wave found the launch
    // out of new paths, last wave exit
    unsigned long long tail[4];
    unsigned long long coherence[2]; // lanes in node steps; lanes at the same node as the wave's first active lane
};

template <bool COUNT> struct Cnt {
    unsigned box = 0, sph = 0, sphh = 0, quad = 0, sh0 = 0, sh1 = 0, sh2 = 0, sh3 = 0, noise = 0, img = 0,
             chk = 0;
};
template <> struct Cnt<false> {};

#define RT_COUNT(field)                                                                            \
    if constexpr (COUNT) cn.field++

// ------------------------------------------------------------------------------------------
// Counter-based sampler.  Key = (seed, pixel, global sample index); numbers within
// a path come from a 32-bit LCG with a PCG (RXS-M-XS) output permutation.  The
// image therefore does not depend on how paths are scheduled or on the GPU count.
RT_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct Rng {
    uint32_t s;
};
RT_HD Rng rng_seed(uint32_t seed, uint32_t pixel, uint32_t sample) {
    Rng r;
    r.s = mix32(pixel * 0x9E3779B9u + mix32(sample * 0x85EBCA6Bu + mix32(seed + 0x68E31DA4u)));
    return r;
}
RT_HD float rng_next(Rng &r) { // uniform in [0,1), 24 bits (rt_constants.h:28-33 draws 32 and rounds)
    r.s = r.s * 747796405u + 2891336453u;
    uint32_t w = ((r.s >> ((r.s >> 28u) + 4u)) ^ r.s) * 277803737u;
    w = (w >> 22u) ^ w;
    return float(w >> 8) * (1.0f / 16777216.0f);
}

// sin and cos of 2*pi*u, u in [0,1).  On the device these are the hardware's
// v_sin_f32 / v_cos_f32, which take their argument in revolutions (absolute error
// about 1e-6: it only perturbs a random direction).
RT_HD void sincos_2pi(float u, float &s, float &c) {
#if defined(__HIP_DEVICE_COMPILE__)
    s = __builtin_amdgcn_sinf(u);
    c = __builtin_amdgcn_cosf(u);
#else
    s = sinf(2.0f * RT_PI * u);
    c = cosf(2.0f * RT_PI * u);
#endif
}

// Uniform direction on the unit sphere.  The reference normalises a point drawn
// by rejection from the unit ball (vec3.h:138-149); the distribution is the same.
RT_HD f3 sample_unit_vector(Rng &rng) {
    float z = 1.0f - 2.0f * rng_next(rng);
    float s, c;
    sincos_2pi(rng_next(rng), s, c);
    float rr = sh_sqrt(fmaxf(0.0f, 1.0f - z * z));
    return mk(rr * c, rr * s, z);
}
// Uniform point in the unit disk (vec3.h:130-136 uses rejection).
RT_HD void sample_unit_disk(Rng &rng, float &px, float &py) {
    float r = sh_sqrt(rng_next(rng));
    float s, c;
    sincos_2pi(rng_next(rng), s, c);
    px = r * c;
    py = r * s;
}

// ------------------------------------------------------------------------------------------
// Primitive tests.  Return the candidate parameter only; the surface is rebuilt
// once, for the winner (surface_at).

// sphere::hit up to the accepted root (sphere.h:40-63).  The interval is open.
// a = |d|^2 (sphere.h:44), which callers that test many spheres against one ray keep.
RT_HD bool sphere_test(f4 g0, f4 g1, f3 o, f3 d, float a, float tm, float tmin, float tmax, float &t_out) {
    f3 center = xyz(g0) + tm * xyz(g1); // sphere.h:93 (static spheres carry a zero center_vec)
    f3 oc = center - o;
    float h = dot(d, oc);
    float c = len2(oc) - g0.w * g0.w;
    float discriminant = h * h - a * c;
    if (discriminant < 0.0f) return false;
    float sqrtd = sqrtf(discriminant);
    float root = (h - sqrtd) / a;
    if (!(tmin < root && root < tmax)) {
        root = (h + sqrtd) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t_out = root;
    return true;
}

// True if the point of the ray closest to the sphere's centre lies inside the sphere's own box
// (centre +- 1.019 radius: inside the tree's box for the sphere, which the host grows by at least
// 2 % of the radius).  Far
// from a small sphere the discriminant of sphere.h:47-52 is dominated by rounding, and a ray that
// passes the sphere at more than its radius can still be reported as hitting it, at about the
// parameter of its closest approach.  When that ray only clips a corner of the sphere's box, the
// reported parameter can lie in front of the box by up to the box's size, and whether a tree walk
// finds such a hit then depends on what else it has found by the time it reaches the box.  Hits
// of a real intersection always pass (their closest point lies inside the sphere); the tree
// traversal drops the others, so that every hit it reports lies inside its leaf's box up to the
// rounding that Trav::cull_t allows for.
RT_HD bool sphere_far(f4 g0, f4 g1, f3 o, float tm) {
    // Nearer than 160 radii no such hit exists: the computed discriminant is off by less than
    // 24 * 2^-24 |d|^2 |oc|^2 (19 roundings of terms of that size at most), and a closest approach
    // outside the box below means a true discriminant below -0.038 |d|^2 radius^2.
    f3 center = xyz(g0) + tm * xyz(g1);
    f3 oc = center - o;
    float r2 = g0.w * g0.w;
    return (len2(oc) - r2) > 26000.0f * r2;
}
RT_HD bool sphere_approach_in_box(f4 g0, f4 g1, f3 o, f3 d, float a, float tm) {
    f3 center = xyz(g0) + tm * xyz(g1);
    f3 oc = center - o;
    float h = dot(d, oc);
    // the closest point is o + (h / a) d; its offset from the centre, times a = |d|^2 > 0 (no division)
    f3 off = h * d - a * oc;
    float half = (a * 1.019f) * fabsf(g0.w);
    return fabsf(off.x) <= half && fabsf(off.y) <= half && fabsf(off.z) <= half;
}

// quad::hit + is_interior (quad.h:38-69, 80-91).  The interval is closed.
RT_HD bool quad_test(f4 g0, f4 g1, f4 g2, f4 g3, f3 o, f3 d, float tmin, float tmax, float &t_out) {
    f3 n = xyz(g0);
    float denominator = dot(n, d);
    if (fabsf(denominator) < 1e-8f) return false;
    float t = (g0.w - dot(n, o)) / denominator;
    if (!(tmin <= t && t <= tmax)) return false;
    f3 Q = xyz(g1);
    f3 w = mk(g1.w, g2.x, g2.y);
    f3 u = mk(g2.z, g2.w, g3.x);
    f3 v = mk(g3.y, g3.z, g3.w);
    f3 planar = (o + t * d) - Q;
    float alpha = dot(w, cross(planar, v));
    float beta = dot(w, cross(u, planar));
    if (!(0.0f <= alpha && alpha <= 1.0f) || !(0.0f <= beta && beta <= 1.0f)) return false;
    t_out = t;
    return true;
}

// The same test as straight-line code: every lane computes everything and the conditions are
// combined at the end.  Same arithmetic, same result (a denominator below the threshold or a
// parameter outside the interval still means "no hit", whatever the later terms evaluate to);
// for scans in which all lanes of a wave test the same quad, where the early exits above only
// idle the lanes that take them while the others finish.
RT_HD bool quad_test_straight(f4 g0, f4 g1, f4 g2, f4 g3, f3 o, f3 d, float tmin, float tmax, float &t_out) {
    f3 n = xyz(g0);
    float denominator = dot(n, d);
    float t = (g0.w - dot(n, o)) / denominator;
    f3 Q = xyz(g1);
    f3 w = mk(g1.w, g2.x, g2.y);
    f3 u = mk(g2.z, g2.w, g3.x);
    f3 v = mk(g3.y, g3.z, g3.w);
    f3 planar = (o + t * d) - Q;
    float alpha = dot(w, cross(planar, v));
    float beta = dot(w, cross(u, planar));
    bool hit = !(fabsf(denominator) < 1e-8f) & (tmin <= t) & (t <= tmax) & (0.0f <= alpha) & (alpha <= 1.0f) &
               (0.0f <= beta) & (beta <= 1.0f);
    t_out = t;
    return hit;
}

// A quad whose normal lies along axis K and whose u and v lie along the two other axes I = K + 1,
// J = K + 2 (mod 3), u along I and v along J (the host swaps u and v where it is the other way
// round, which negates n, D and w and exchanges alpha and beta: the same t and the same verdict).
// The terms of quad.h:32-62 that multiply a zero component are zeros, and adding or subtracting
// a zero changes nothing, so what is left gives the same bits:
//   dot(n, x) = n_K x_K;   alpha = w_K (planar_I v_J);   beta = w_K (u_I planar_J).
// 27 instructions instead of 75.
template <int K> RT_HD float axis_of(f3 v) { return K == 0 ? v.x : K == 1 ? v.y : v.z; }
template <int K>
RT_HD bool quad_test_axis(float nk, float D, float qi, float qj, float wk, float ui, float vj, f3 o, f3 d, float tmin,
                          float tmax, float &t_out) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
    const float denominator = nk * axis_of<K>(d);
    const float t = (D - nk * axis_of<K>(o)) / denominator;
    const float planar_i = (axis_of<I>(o) + t * axis_of<I>(d)) - qi;
    const float planar_j = (axis_of<J>(o) + t * axis_of<J>(d)) - qj;
    const float alpha = wk * (planar_i * vj);
    const float beta = wk * (ui * planar_j);
    t_out = t;
    return !(fabsf(denominator) < 1e-8f) & (tmin <= t) & (t <= tmax) & (0.0f <= alpha) & (alpha <= 1.0f) &
           (0.0f <= beta) & (beta <= 1.0f);
}
// Scan record of a slot, RT_SCAN_F4 x 16 bytes, by kind:
//   RT_SCAN_AXIS + K: {kind, chain, n_K, D} {Q_I, Q_J, w_K, u_I} {v_J, -, -, -}
//   RT_SCAN_SPHERE:   {kind, chain, c.x, c.y} {c.z, radius, cv.x, cv.y} {cv.z, -, -, -}
//   RT_SCAN_QUAD:     {kind, chain, -, -} ... (tested from the slot itself)
#define RT_SCAN_F4 3
#define RT_SCAN_SPHERE 0
#define RT_SCAN_QUAD 1
#define RT_SCAN_AXIS 2

// ------------------------------------------------------------------------------------------
// Instance wrappers (hittable.h:51-344), applied to the ray at run time exactly
// as the reference does; used by the linear traversal, where it costs one ray
// transform per wrapped object.
RT_HD f3 rot_y_ccw(f4 a, f3 v) { // hittable.h:162-166   a = {sa ca sb cb}
    float vx = a.w * v.x + a.z * v.z;
    float vz = -a.z * v.x + a.w * v.z;
    return mk(vx, v.y, vz);
}
RT_HD f3 rot_y_cw(f4 a, f3 v) { // hittable.h:169-173
    float vx = a.w * v.x - a.z * v.z;
    float vz = a.z * v.x + a.w * v.z;
    return mk(vx, v.y, vz);
}
// rotate_xyz with the matrix rows precomputed (rot_matrix_* below evaluate the entries exactly as
// hittable.h:285-293 / 299-307 write them; a row is applied as m0*x + m1*y + m2*z, left to right).
struct Mat3 {
    float m[9];
};
RT_HD f3 mat_apply(const Mat3 &r, f3 v) {
    float vx = r.m[0] * v.x + r.m[1] * v.y + r.m[2] * v.z;
    float vy = r.m[3] * v.x + r.m[4] * v.y + r.m[5] * v.z;
    float vz = r.m[6] * v.x + r.m[7] * v.y + r.m[8] * v.z;
    return mk(vx, vy, vz);
}
RT_HD Mat3 rot_matrix_ccw(float sa, float ca, float sb, float cb, float sg, float cg) { // hittable.h:284-295
    Mat3 r;
    r.m[0] = cb * cg, r.m[1] = sa * sb * cg - sg * ca, r.m[2] = sb * ca * cg + sa * sg;
    r.m[3] = cb * sg, r.m[4] = ca * cg + sa * sb * sg, r.m[5] = -sa * cg + sg * sb * ca;
    r.m[6] = -sb, r.m[7] = sa * cb, r.m[8] = ca * cb;
    return r;
}
RT_HD Mat3 rot_matrix_cw(float sa, float ca, float sb, float cb, float sg, float cg) { // hittable.h:298-309
    Mat3 r;
    r.m[0] = cb * cg, r.m[1] = sg * cb, r.m[2] = -sb;
    r.m[3] = sa * sb * cg - sg * ca, r.m[4] = sa * sb * sg + ca * cg, r.m[5] = sa * cb;
    r.m[6] = sb * ca * cg + sa * sg, r.m[7] = sb * sg * ca - sa * cg, r.m[8] = ca * cb;
    return r;
}
// the two matrices of one packed wrapper op (q = its RT_XF_F4 words)
template <class P4> RT_HD Mat3 xf_matrix_cw(P4 q) {
    f4 a = q[2], b = q[3], c = q[4];
    Mat3 r;
    r.m[0] = a.x, r.m[1] = a.y, r.m[2] = a.z, r.m[3] = a.w, r.m[4] = b.x, r.m[5] = b.y, r.m[6] = b.z, r.m[7] = b.w,
    r.m[8] = c.x;
    return r;
}
template <class P4> RT_HD Mat3 xf_matrix_ccw(P4 q) {
    f4 c = q[4], d = q[5], e = q[6];
    Mat3 r;
    r.m[0] = c.y, r.m[1] = c.z, r.m[2] = c.w, r.m[3] = d.x, r.m[4] = d.y, r.m[5] = d.z, r.m[6] = d.w, r.m[7] = e.x,
    r.m[8] = e.y;
    return r;
}

// world ray -> object ray: outermost wrapper first.  P4 / PI: plain pointers (per-lane chain)
// or uniform pointers (the whole wave applies the same chain).
template <class P4, class PI> RT_HD void chain_ray_to_object_p(P4 xf, PI chain_first, int chain, f3 &o, f3 &d) {
    int first = chain_first[chain], last = chain_first[chain + 1];
    for (int k = last - 1; k >= first; k--) {
        P4 q = xf + (size_t)k * RT_XF_F4;
        f4 op = q[0];
        int type = __builtin_bit_cast(int, op.x);
        f3 p = mk(op.y, op.z, op.w);
        if (type == RT_XF_TRANSLATE) {
            o = o - p;
        } else if (type == RT_XF_ROTATE_Y) {
            f4 a = q[1];
            o = rot_y_cw(a, o);
            d = rot_y_cw(a, d);
        } else if (type == RT_XF_ROTATE_XYZ) {
            Mat3 r = xf_matrix_cw(q);
            o = mat_apply(r, o);
            d = mat_apply(r, d);
        } else {
            o = o / p;
            d = d / p;
        }
    }
}
// object-space hit point and normal -> world: innermost wrapper first.
template <class P4, class PI> RT_HD void chain_hit_to_world_p(P4 xf, PI chain_first, int chain, f3 &p, f3 &n) {
    int first = chain_first[chain], last = chain_first[chain + 1];
    for (int k = first; k < last; k++) {
        P4 q = xf + (size_t)k * RT_XF_F4;
        f4 op = q[0];
        int type = __builtin_bit_cast(int, op.x);
        f3 s = mk(op.y, op.z, op.w);
        if (type == RT_XF_TRANSLATE) {
            p = p + s;
        } else if (type == RT_XF_ROTATE_Y) {
            f4 a = q[1];
            p = rot_y_ccw(a, p);
            n = rot_y_ccw(a, n);
        } else if (type == RT_XF_ROTATE_XYZ) {
            Mat3 r = xf_matrix_ccw(q);
            p = mat_apply(r, p);
            n = mat_apply(r, n);
        } else {
            p = p * s;
            n = unit(n / s);
        }
    }
}
RT_HD void chain_ray_to_object(const DevScene &sc, int chain, f3 &o, f3 &d) {
    chain_ray_to_object_p(sc.xf, sc.chain_first, chain, o, d);
}
RT_HD void chain_hit_to_world(const DevScene &sc, int chain, f3 &p, f3 &n) {
    chain_hit_to_world_p(sc.xf, sc.chain_first, chain, p, n);
}

// ------------------------------------------------------------------------------------------
// Traversal
// Offset of a slot in 16-byte words: slot_f4 is 2 or 4, so a shift does (a 32-bit multiply
// issues at quarter rate).  Builds without quads have 2-word slots only.
RT_HD size_t slot_words(const DevScene &sc, int slot) {
    return (size_t)(unsigned)slot << (RT_FEAT_QUAD ? (sc.slot_f4 >> 1) : 1);
}

struct TraceHit {
    float t;
    int slot;      // -1: miss
    int inst = -1; // the instance the primitive was hit in (-1: a primitive of the world itself)
};

// hittable_list::hit (hittable_list.h:29-43): scan in test order with a shrinking
// interval.  Every lane of the wave walks the same list: the primitives, their records and
// the wrapper ops are read through uniform pointers (scalar loads on the device), whatever
// copy of the scene `v` names.
template <bool COUNT>
RT_HD TraceHit trace_linear(const DevScene &sc, const SceneView &v, const Ray &r, float tmin, float tmax,
                            Cnt<COUNT> &cn) {
    TraceHit best;
    best.t = tmax;
    best.slot = -1;
    int cur_chain = -1;
    f3 o = r.o, d = r.d;
    float dd = len2(d);
    const int stride = sc.slot_f4;
    const uf4_ptr u_slots = (uf4_ptr)sc.slots;
    const uf4_ptr u_scan = (uf4_ptr)sc.scan;
    if (sc.n_slots <= 0) return best;
    // The scan record of slot s + 1 is requested before slot s is tested: a scalar load takes a
    // few hundred cycles and an axis-aligned quad is tested in under thirty instructions.
    f4 r0 = u_scan[0], r1 = u_scan[1], r2 = u_scan[2];
    for (int s = 0; s < sc.n_slots; s++) {
        const uf4_ptr nx = u_scan + (size_t)(s + 1 < sc.n_slots ? s + 1 : s) * RT_SCAN_F4;
        const f4 n0 = nx[0], n1 = nx[1], n2 = nx[2];
        const int kind = __builtin_bit_cast(int, r0.x), chain = __builtin_bit_cast(int, r0.y);
        if (chain != cur_chain) {
            cur_chain = chain;
            o = r.o;
            d = r.d;
            if (RT_FEAT_CHAIN && cur_chain >= 0)
                chain_ray_to_object_p((uf4_ptr)sc.xf, (uint_ptr)sc.chain_first, cur_chain, o, d);
            dd = len2(d);
        }
        float t;
        bool h;
        if (!RT_FEAT_QUAD || kind == RT_SCAN_SPHERE) {
            RT_COUNT(sph);
            h = sphere_test(f4{r0.z, r0.w, r1.x, r1.y}, f4{r1.z, r1.w, r2.x, 0.0f}, o, d, dd, r.tm, tmin, best.t, t);
            if constexpr (COUNT) cn.sphh += h ? 1 : 0;
        } else {
            RT_COUNT(quad);
            if (kind == RT_SCAN_AXIS + 0) {
                h = quad_test_axis<0>(r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, o, d, tmin, best.t, t);
            } else if (kind == RT_SCAN_AXIS + 1) {
                h = quad_test_axis<1>(r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, o, d, tmin, best.t, t);
            } else if (kind == RT_SCAN_AXIS + 2) {
                h = quad_test_axis<2>(r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, o, d, tmin, best.t, t);
            } else {
                const uf4_ptr g = u_slots + (size_t)s * stride;
                h = quad_test_straight(g[0], g[1], g[2], g[3], o, d, tmin, best.t, t);
            }
        }
        if (h) {
            best.t = t;
            best.slot = s;
        }
        r0 = n0, r1 = n1, r2 = n2;
    }
    return best;
}

// Decides between two candidates with the same t as the reference's sequential
// scan would: a later quad replaces (closed interval, quad.h:46), a later
// sphere does not (open interval, sphere.h:58).
RT_HD bool tie_prefers_candidate(int cand_id, bool cand_is_quad, int best_id, bool best_is_quad) {
    if (best_id < 0) return cand_is_quad; // t == tmax: only the closed interval admits it
    bool cand_later = cand_id > best_id;
    bool later_is_quad = cand_later ? cand_is_quad : best_is_quad;
    return later_is_quad ? cand_later : !cand_later;
}

struct BoxRay { // per-ray constants of the slab test
    f3 inv;  // ~1/d (a zero component is replaced by a huge finite reciprocal)
    f3 oi;   // -(o * inv)
    float e2; // absolute slack: covers the rounding of o*inv inside fma(plane, inv, -o*inv)
    // byte offset of the ray's NEAR plane inside an axis' {lo[4]} {hi[4]} pair of a node: 0 where the
    // direction component is positive (lo is entered first), 16 where it is negative
    int sx, sy, sz;
};
RT_HD float safe_rcp(float d) {
    // |d| below 1e-30 would overflow; a slab that the ray runs (almost) parallel to
    // then yields +-huge parameters of the right sign, which is what the test needs.
    // The reciprocal only feeds the conservative box test: 1 ulp accuracy is plenty.
    float ad = fabsf(d);
    float m = ad < 1e-30f ? 1e-30f : ad;
#if defined(__HIP_DEVICE_COMPILE__)
    return copysignf(__builtin_amdgcn_rcpf(m), d);
#else
    return copysignf(1.0f / m, d);
#endif
}
RT_HD BoxRay make_box_ray(const Ray &r) {
    BoxRay b;
    b.inv = mk(safe_rcp(r.d.x), safe_rcp(r.d.y), safe_rcp(r.d.z));
    f3 oi = r.o * b.inv;
    b.oi = -oi;
    // fma(plane, inv, -o*inv) carries an absolute error of about 2^-24 |o*inv| from the
    // rounded product; widen the interval by 2^-21 of the largest such magnitude.  A ray
    // that runs parallel to an axis (huge reciprocal) is excluded: its slab parameters are
    // +-huge and decide by sign alone.
    float mx = fabsf(r.d.x) < 1e-30f ? 0.0f : fabsf(oi.x);
    float my = fabsf(r.d.y) < 1e-30f ? 0.0f : fabsf(oi.y);
    float mz = fabsf(r.d.z) < 1e-30f ? 0.0f : fabsf(oi.z);
    b.e2 = 9.5367431640625e-7f * fmaxf(mx, fmaxf(my, mz));
    // (the sign of the reciprocal is the sign of the direction component, zeros included)
    b.sx = (int)((__builtin_bit_cast(unsigned, b.inv.x) >> 31) << 4);
    b.sy = (int)((__builtin_bit_cast(unsigned, b.inv.y) >> 31) << 4);
    b.sz = (int)((__builtin_bit_cast(unsigned, b.inv.z) >> 31) << 4);
    return b;
}
// The same test from the ray's near and far planes of each axis (a node stores them apart, so that
// a ray loads the ones it enters and leaves through directly): two FMAs per axis.
RT_HD bool box_hit_planes(float px, float py, float pz, float qx, float qy, float qz, const BoxRay &b, float tmin,
                          float tmax, float &tn) {
    float nx = __builtin_fmaf(px, b.inv.x, b.oi.x), fx = __builtin_fmaf(qx, b.inv.x, b.oi.x);
    float ny = __builtin_fmaf(py, b.inv.y, b.oi.y), fy = __builtin_fmaf(qy, b.inv.y, b.oi.y);
    float nz = __builtin_fmaf(pz, b.inv.z, b.oi.z), fz = __builtin_fmaf(qz, b.inv.z, b.oi.z);
    tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), tmin);
    float tf = fminf(fminf(fminf(fx, fy), fz), tmax);
#if defined(RT_WHATIF_MINMAX) && defined(__HIP_DEVICE_COMPILE__)
    // What-if build: RT_WHATIF_MINMAX more half-rate instructions per box that change nothing.
#pragma unroll
    for (int k = 0; k < RT_WHATIF_MINMAX; k++) asm volatile("v_min_f32 %0, %0, %1" : "+v"(tf) : "v"(tmax));
#endif
#if defined(RT_WHATIF_FMA) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int k = 0; k < RT_WHATIF_FMA; k++) asm volatile("v_fma_f32 %0, %0, 1.0, 0" : "+v"(tf));
#endif
    float tf_wide = __builtin_fmaf(tf, 1.0f + 1.9073486328125e-6f, b.e2); // (see box_hit)
    return tn <= tf_wide;
}
// Conservative ray/box overlap on [tmin, tmax] for a box given as centre c and half extent
// h: true if the box may be hit, tn = the (slightly early) entry parameter.  With
// m = (c - o) / d per axis the slab is [m - h/|d|, m + h/|d|]: three FMAs per axis and no
// per-axis min / max (v_min_f32 / v_max_f32 issue at half the rate of v_fma_f32 on gfx950).
// Boxes are padded at build time.
RT_HD bool box_hit(f3 c, f3 h, const BoxRay &b, float tmin, float tmax, float &tn) {
    float mx = __builtin_fmaf(c.x, b.inv.x, b.oi.x);
    float my = __builtin_fmaf(c.y, b.inv.y, b.oi.y);
    float mz = __builtin_fmaf(c.z, b.inv.z, b.oi.z);
    float ix = fabsf(b.inv.x), iy = fabsf(b.inv.y), iz = fabsf(b.inv.z);
    float nx = __builtin_fmaf(-h.x, ix, mx), fx = __builtin_fmaf(h.x, ix, mx);
    float ny = __builtin_fmaf(-h.y, iy, my), fy = __builtin_fmaf(h.y, iy, my);
    float nz = __builtin_fmaf(-h.z, iz, mz), fz = __builtin_fmaf(h.z, iz, mz);
    tn = fmaxf(fmaxf(fmaxf(nx, ny), nz), tmin); // (v_max3_f32 + v_max_f32)
    float tf = fminf(fminf(fminf(fx, fy), fz), tmax);
    // exit parameter widened by the absolute slack (2e) and a relative one (2^-19 tf) for the
    // rounding of the fma results themselves; when the test is close, |tn| ~ |tf|.  One FMA:
    // for tf < 0 the relative part narrows instead, by less than 2^-19 * tmin where it could
    // matter (tn >= tmin > 0 rejects anything whose widened exit stays below tmin)
    float tf_wide = __builtin_fmaf(tf, 1.0f + 1.9073486328125e-6f, b.e2);
    return tn <= tf_wide;
}

// Deepest tree the builders produce (inner levels of the binary tree the wide tree is
// collapsed from; the wide tree is never deeper).
#define RT_MAX_TREE_LEVELS 30
// Entries a traversal stack needs for a wide tree of `levels` inner levels: a node leaves at
// most three children behind, and the step writes four entries before it takes one back.
#define RT_STACK_CAP(levels) (3 * (levels) + 2)

// Traversal stack of one lane: a column of a [entry][lane] array, so that the lanes of a
// wave touch consecutive banks / addresses.  Entries [0, lds_cap) live in LDS; scenes whose
// tree is too deep for that keep the entries beyond in a second column in global memory (only
// rays that have that many subtrees pending ever get there).
// STRIDE: elements between consecutive entries of a column (both columns), a compile-time
// constant where the kernel's workgroup size is one (the entry offsets then fold into the
// instructions' immediates); 0: taken from rt_stride.
template <typename T, int STRIDE = 0> struct TravStack {
    typedef T elem;
    T *lds;
    int *spill;    // null: the LDS column holds the whole stack
    int rt_stride; // used if STRIDE == 0
    int lds_end;   // lds_cap * stride
    RT_HD int stride() const { return STRIDE ? STRIDE : rt_stride; }
};

// BVH traversal as a resumable state machine, so that a wave can interleave the
// traversal steps of its 64 rays with shading (rt_kernels.hip) instead of waiting
// for its slowest ray.  ref >= 0: inner node; ref < 0: leaf (~ref = slot*2 + type);
// RT_TRAV_DONE: finished.
#define RT_TRAV_DONE ((int)0x80000000)
// Relative part of the culling slack (2^-9).  The root sphere.h:47-60 computes is off by up to
// sqrt(k eps) |oc| / |d| near a silhouette (k: a handful of roundings of terms of size |oc|^2 |d|^2 in
// the discriminant; eps = 2^-24: about 1e-3 of the distance), so a primitive's reported hit can lie
// that far in front of its own box.
#ifndef RT_CULL_REL
#define RT_CULL_REL 1.953125e-3f
#endif
#ifndef RT_OWN_BOX_RULE
#define RT_OWN_BOX_RULE 1 // (0: what-if builds that measure the rule's cost)
#endif
struct Trav {
    BoxRay br;
    float best_t;
    // Boxes are culled against cull_t = best_t (1 + RT_CULL_REL), not against best_t: a box that is
    // dropped then lies so far beyond the closest hit so far that no primitive inside it can report
    // a parameter before it, rounding included.  The closest hit is then the closest over ALL
    // primitives whose boxes the ray meets, whatever the order and the moment in which leaves are
    // tested (lanes of the decoupled kernels walk on past leaves they have not tested yet).  Why
    // a relative slack is enough: within 16 radii of a sphere the rounding of its root stays below
    // the 2 % of the radius by which the host grows the sphere's box (a real hit lies that far
    // behind the box's entry at least), beyond that it stays below RT_CULL_REL of the distance; a
    // quad's hit point lies inside its (grown) box, so its parameter is never before the entry.
    float cull_t;
    int best_slot;
    int ref;
    int sp; // entries on the stack, times the stride
    bool best_is_quad;
    float dd; // |d|^2 of the world-space ray
    int cur_inst;  // the instance whose sub-scene is being walked (-1: the world's own tree)
    int best_inst; // the instance of the closest hit so far
    f3 io, id;     // inside an instance: the ray in the sub-scene's space (taken there once, on entry)
    float idd;     // ... and |d|^2 there
};
RT_HD float cull_bound(float best_t) { return best_t * (1.0f + RT_CULL_REL); }
RT_HD void trav_begin(Trav &tv, const Ray &r, float tmax) {
    tv.br = make_box_ray(r);
    tv.dd = len2(r.d);
    tv.best_t = tmax;
    tv.cull_t = cull_bound(tmax);
    tv.best_slot = -1;
    tv.best_is_quad = false;
    tv.ref = 0; // root
    tv.sp = 0;
    tv.cur_inst = -1;
    tv.best_inst = -1;
}

// A stack entry is a sort key: the child's entry parameter in the high bits (a positive float
// orders like its bit pattern), its reference in the low ref_bits bits.  16-bit stacks keep
// only the low half (ref_bits = 16 there); 32-bit stacks keep the key and cut the reference
// out when it is popped.
template <typename T> RT_HD int stack_entry_ref(T e, int ref_bits) {
    if (sizeof(T) == 2) return (int)e;
    const int sh = 32 - ref_bits;
    return (int)((unsigned)e << sh) >> sh; // (sign extension of the low ref_bits bits)
}
template <class ST, bool SPILL> RT_HD void trav_pop(Trav &tv, const ST &st, int ref_bits) {
    typedef typename ST::elem T;
    if (tv.sp == 0) {
        tv.ref = RT_TRAV_DONE;
        return;
    }
    tv.sp -= st.stride();
    if (!SPILL || tv.sp < st.lds_end)
        tv.ref = stack_entry_ref<T>(st.lds[tv.sp], ref_bits);
    else
        tv.ref = stack_entry_ref<int>(st.spill[tv.sp - st.lds_end], ref_bits);
}

// Offset of LDS node `ref` in 16-byte words.  The spacing of 9 words is not a power of two, and
// the compiler turns every spelling of ref * 9 back into a full 32-bit multiply, which issues at
// quarter rate; one shift-and-add does it at full rate.
RT_HD unsigned lds_node_words(int ref) {
    static_assert(RT_NODE_LDS_F4 == 9, "the shift-and-add below is ref * 9");
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned w;
    asm("v_lshl_add_u32 %0, %1, 3, %1" : "=v"(w) : "v"(ref));
    return w;
#else
    return (unsigned)ref * 9u;
#endif
}
RT_HD unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
RT_HD unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }
RT_HD unsigned umed3_(unsigned a, unsigned b, unsigned c) { // the middle one of three
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return umax_(umin_(a, b), umin_(umax_(a, b), c));
#endif
}
// Stack key of a child: the high bits of its entry distance, the reference in the low bits.
RT_HD unsigned key_of(unsigned hi_mask, float entry, float ref_bits_as_float) {
    const unsigned e = __builtin_bit_cast(unsigned, entry), r = __builtin_bit_cast(unsigned, ref_bits_as_float);
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned k; // (the compiler keeps both masks in registers and spends an and + an and-or)
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(k) : "s"(hi_mask), "v"(e), "v"(r));
    return k;
#else
    return (e & hi_mask) | (r & ~hi_mask);
#endif
}
// Four keys in descending order with seven three-operand instructions: the first three are
// sorted by min3 / med3 / max3, the fourth is placed among them (a compare-exchange network
// takes ten two-operand ones).
RT_HD void sort4_desc(unsigned k0, unsigned k1, unsigned k2, unsigned k3, unsigned &a0, unsigned &a1, unsigned &a2,
                      unsigned &a3) {
    const unsigned l = umin_(umin_(k0, k1), k2), h = umax_(umax_(k0, k1), k2), m = umed3_(k0, k1, k2);
    a0 = umax_(h, k3);
    a1 = umed3_(m, h, k3);
    a2 = umed3_(l, m, k3);
    a3 = umin_(l, k3);
}

// One inner node (requires tv.ref >= 0) of the 4-wide tree: tests the four child boxes, puts
// the children that are hit on the stack, farthest first, and takes the nearest back.
// Node, 8 x 16 bytes: the planes of the four child boxes grouped by axis
//   {lo_x[0..3]} {hi_x[0..3]} {lo_y[0..3]} {hi_y[0..3]} {lo_z[0..3]} {hi_z[0..3]}
//   {ref0 ref1 ref2 ref3} {number of children, - - -}
// A ray reads, per axis, the plane it enters through and the one it leaves through (lo then hi for
// a positive direction component, hi then lo for a negative one: BoxRay::sx/sy/sz), so that a box
// costs two FMAs per axis and no per-axis min / max.  An unused child has an inverted box (never
// hit).  The four keys (0 for a child that is missed) are sorted in descending order with integer
// min / max and all four are written above the stack's top: the farthest child lands lowest, the
// missed ones above the new top where they do not count.  No branch depends on how many children
// were hit.
// TOP: nodes [0, n_top) are read from `top` (the workgroup's LDS) instead of `nodes` (HBM).
// NODE_F4: spacing of `nodes` (RT_NODE_F4 in global memory, RT_NODE_LDS_F4 for a copy in LDS).
// The loads name their address space (LDS or global memory) themselves: the scene kept in HBM
// with the top of its tree in LDS has both kinds of node in one kernel, and with plain pointers the
// compiler merges the two paths into flat loads through a selected pointer.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_AS_LDS __attribute__((address_space(3)))
#define RT_AS_GLOBAL __attribute__((address_space(1)))
#else
#define RT_AS_LDS
#define RT_AS_GLOBAL
#endif
struct NodePlanes {
    f4 px, qx, py, qy, pz, qz, refs; // near (p) and far (q) planes per axis, child references
    int n_children;
};
template <bool COUNT> RT_HD NodePlanes node_planes_lds(const f4 *node, const BoxRay &b) {
    const RT_AS_LDS char *n = (const RT_AS_LDS char *)node;
    NodePlanes r;
    r.px = *(const RT_AS_LDS f4 *)(n + b.sx), r.qx = *(const RT_AS_LDS f4 *)(n + (16 - b.sx));
    r.py = *(const RT_AS_LDS f4 *)(n + 32 + b.sy), r.qy = *(const RT_AS_LDS f4 *)(n + 32 + (16 - b.sy));
    r.pz = *(const RT_AS_LDS f4 *)(n + 64 + b.sz), r.qz = *(const RT_AS_LDS f4 *)(n + 64 + (16 - b.sz));
    r.refs = *(const RT_AS_LDS f4 *)(n + 96);
    r.n_children = COUNT ? *(const RT_AS_LDS int *)(n + 112) : 0;
#if defined(RT_WHATIF_LDS_READS) && defined(__HIP_DEVICE_COMPILE__)
    // What-if build (tools/whatif.sh): RT_WHATIF_LDS_READS more 16-byte LDS reads per node step, of
    // the neighbouring node, folded into a value that never changes anything.
    {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < RT_WHATIF_LDS_READS; k++) {
            const f4 x = *(const RT_AS_LDS f4 *)(n + 144 + 16 * k + b.sx);
            acc += x.x;
        }
        if (acc == 1.2345e-33f) r.refs.x = 0.0f;
    }
#endif
    return r;
}
template <bool COUNT> RT_HD NodePlanes node_planes_global(const f4 *node, const BoxRay &b) {
    const RT_AS_GLOBAL char *n = (const RT_AS_GLOBAL char *)node;
    NodePlanes r;
    r.px = *(const RT_AS_GLOBAL f4 *)(n + b.sx), r.qx = *(const RT_AS_GLOBAL f4 *)(n + (16 - b.sx));
    r.py = *(const RT_AS_GLOBAL f4 *)(n + 32 + b.sy), r.qy = *(const RT_AS_GLOBAL f4 *)(n + 32 + (16 - b.sy));
    r.pz = *(const RT_AS_GLOBAL f4 *)(n + 64 + b.sz), r.qz = *(const RT_AS_GLOBAL f4 *)(n + 64 + (16 - b.sz));
    r.refs = *(const RT_AS_GLOBAL f4 *)(n + 96);
    r.n_children = COUNT ? *(const RT_AS_GLOBAL int *)(n + 112) : 0;
    return r;
}
template <bool COUNT, class ST, bool SPILL>
RT_HD void trav_inner_planes(Trav &tv, const NodePlanes &np, float tmin, const ST &st, int ref_bits, Cnt<COUNT> &cn);
// NODES_IN_LDS: `nodes` itself is a copy in LDS (the whole scene is); else it is global memory.
template <bool COUNT, class ST, bool TOP, bool SPILL, int NODE_F4 = RT_NODE_F4>
RT_HD void trav_inner(Trav &tv, const f4 *nodes, float tmin, const ST &st, int ref_bits, Cnt<COUNT> &cn,
                      const f4 *top = nullptr, int n_top = 0) {
    NodePlanes np;
    if (TOP && tv.ref < n_top) { // two explicit paths: LDS reads and global loads, no flat addressing
        np = node_planes_lds<COUNT>(top + lds_node_words(tv.ref), tv.br);
    } else if (NODE_F4 == RT_NODE_LDS_F4) {
        np = node_planes_lds<COUNT>(nodes + lds_node_words(tv.ref), tv.br);
    } else {
        np = node_planes_global<COUNT>(nodes + (size_t)tv.ref * NODE_F4, tv.br);
    }
    trav_inner_planes<COUNT, ST, SPILL>(tv, np, tmin, st, ref_bits, cn);
}
// The step itself, from the planes and references of the node wherever they were read from.
template <bool COUNT, class ST, bool SPILL>
RT_HD void trav_inner_planes(Trav &tv, const NodePlanes &np, float tmin, const ST &st, int ref_bits, Cnt<COUNT> &cn) {
    typedef typename ST::elem T;
    if constexpr (COUNT) cn.box += np.n_children;
    const f4 px = np.px, qx = np.qx, py = np.py, qy = np.qy, pz = np.pz, qz = np.qz, q6 = np.refs;
    float e0, e1, e2, e3;
    const bool h0 = box_hit_planes(px.x, py.x, pz.x, qx.x, qy.x, qz.x, tv.br, tmin, tv.cull_t, e0);
    const bool h1 = box_hit_planes(px.y, py.y, pz.y, qx.y, qy.y, qz.y, tv.br, tmin, tv.cull_t, e1);
    const bool h2 = box_hit_planes(px.z, py.z, pz.z, qx.z, qy.z, qz.z, tv.br, tmin, tv.cull_t, e2);
    const bool h3 = box_hit_planes(px.w, py.w, pz.w, qx.w, qy.w, qz.w, tv.br, tmin, tv.cull_t, e3);
    const unsigned hi = ref_bits >= 32 ? 0u : ~((1u << ref_bits) - 1u); // the bits of a key that hold the distance
    // (the keys first, then the selects: a key built inside the conditional would become a branch)
    const unsigned u0 = key_of(hi, e0, q6.x), u1 = key_of(hi, e1, q6.y), u2 = key_of(hi, e2, q6.z), u3 = key_of(hi, e3, q6.w);
    const unsigned k0 = h0 ? u0 : 0u, k1 = h1 ? u1 : 0u, k2 = h2 ? u2 : 0u, k3 = h3 ? u3 : 0u;
    unsigned a0, a1, a2, a3;
    sort4_desc(k0, k1, k2, k3, a0, a1, a2, a3);
    const int n_hit = (int)h0 + (int)h1 + (int)h2 + (int)h3;
    const int sp = tv.sp, sd = st.stride();
    if (!SPILL || sp + 3 * sd < st.lds_end) {
        st.lds[sp] = (T)a0;
        st.lds[sp + sd] = (T)a1;
        st.lds[sp + 2 * sd] = (T)a2;
        st.lds[sp + 3 * sd] = (T)a3;
    } else {
        const unsigned a[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = sp + i * sd;
            if (p < st.lds_end)
                st.lds[p] = (T)a[i];
            else
                st.spill[p - st.lds_end] = (int)a[i];
        }
    }
    tv.sp = sp + n_hit * sd;
    trav_pop<ST, SPILL>(tv, st, ref_bits);
}
// Exact test of one leaf against the current best (ties included).
// IN_TREE: the leaf was reached through the tree (false: one of the primitives tested up front).
template <bool COUNT, bool IN_TREE = true>
RT_HD void trav_test_leaf(Trav &tv, int leaf_ref, const DevScene &sc, const SceneView &v, const Ray &r,
                          float tmin, Cnt<COUNT> &cn) {
    int code = ~leaf_ref;
    int s = code >> 1;
    bool is_quad = RT_FEAT_QUAD && (code & 1) != 0;
    const f4 *g = v.slots + slot_words(sc, s);
    float t;
    bool h;
    // a primitive inside instance wrappers is tested in its object space (t is the same
    // parameter there: the direction is transformed, not normalised)
    f3 o = r.o, d = r.d;
    float dd = tv.dd;
    int id_off = 0, best_id_off = 0;
    if (RT_FEAT_INSTANCE && sc.n_inst > 0) { // inside an instance: its wrappers first (the outer ones)
        if (tv.cur_inst >= 0) {
            id_off = sc.inst[tv.cur_inst].z;
            o = tv.io;
            d = tv.id;
            dd = tv.idd;
        }
        if (tv.best_inst >= 0) best_id_off = sc.inst[tv.best_inst].z;
    }
    if (RT_FEAT_CHAIN && sc.n_chains > 0) {
        int chain = v.rec_i[s].z;
        if (chain >= 0) {
            chain_ray_to_object(sc, chain, o, d);
            dd = len2(d);
        }
    }
    // candidates at t == best_t are admitted here and sorted out by the tie rule
    if (!is_quad) {
        RT_COUNT(sph);
        h = sphere_test(g[0], g[1], o, d, dd, r.tm, tmin, RT_INF, t);
        if constexpr (COUNT) cn.sphh += h ? 1 : 0;
        h = h && t <= tv.best_t;
        // (primitives tested up front are never culled: their hits stand as computed)
        if (RT_OWN_BOX_RULE && IN_TREE && h && sphere_far(g[0], g[1], o, r.tm)) // (rare: most waves skip it)
            h = sphere_approach_in_box(g[0], g[1], o, d, dd, r.tm);
    } else {
        RT_COUNT(quad);
        h = quad_test(g[0], g[1], g[2], g[3], o, d, tmin, tv.best_t, t);
    }
    if (h) {
        bool take = t < tv.best_t;
        if (!take) { // t == best_t: rare
            int cid = v.rec_i[s].w + id_off;
            int bid = tv.best_slot >= 0 ? v.rec_i[tv.best_slot].w + best_id_off : -1;
            take = tie_prefers_candidate(cid, is_quad, bid, tv.best_is_quad);
        }
        if (take) {
            tv.best_t = t;
            tv.cull_t = cull_bound(t);
            tv.best_slot = s;
            tv.best_is_quad = is_quad;
            if (RT_FEAT_INSTANCE) tv.best_inst = tv.cur_inst;
        }
    }
}

// The primitives that are not in the tree (DevScene::n_big), right after trav_begin.
template <bool COUNT>
RT_HD void trav_test_big(Trav &tv, const DevScene &sc, const SceneView &v, const Ray &r, float tmin, Cnt<COUNT> &cn) {
    for (int j = 0; j < sc.n_big; j++)
        trav_test_leaf<COUNT, false>(tv, ~(j * 2 + ((sc.big_quads >> j) & 1)), sc, v, r, tmin, cn);
}

// Two-level scenes.  A leaf whose slot lies beyond the primitives is not a primitive: slot
// n_slots + k enters instance k (the ray goes into the sub-scene's space through the instance's
// wrappers, a mark is left on the stack, and the walk goes on at the root of the sub-scene's
// tree; t is the same parameter in both spaces, so the closest hit so far keeps culling), and
// slot n_slots + n_inst is that mark coming off the stack again: back to the world's space.
// Such a leaf is a barrier: a lane must not walk on past it (the boxes ahead are in the other
// space) before it has been dealt with.
RT_HD bool trav_is_barrier(const DevScene &sc, int ref) {
    return RT_FEAT_INSTANCE && ref < 0 && ref != RT_TRAV_DONE && ((~ref) >> 1) >= sc.n_slots;
}
template <class ST, bool SPILL>
RT_HD void trav_barrier(Trav &tv, const DevScene &sc, const Ray &world, const ST &st) {
    typedef typename ST::elem T;
    const int k = ((~tv.ref) >> 1) - sc.n_slots;
    if (k < sc.n_inst) { // enter
        const i4 in = sc.inst[k];
        const int mark = ~((sc.n_slots + sc.n_inst) << 1);
        // (a stack entry is a key whose low ref_bits bits are the reference; the distance bits of the
        // mark do not matter: nothing is sorted against it)
        const unsigned lo = sc.ref_bits >= 32 ? 0xffffffffu : (1u << sc.ref_bits) - 1u;
        if (!SPILL || tv.sp < st.lds_end)
            st.lds[tv.sp] = (T)((unsigned)mark & lo);
        else
            st.spill[tv.sp - st.lds_end] = (int)((unsigned)mark & lo);
        tv.sp += st.stride();
        tv.cur_inst = k;
        Ray r = world;
        if (in.y >= 0) chain_ray_to_object(sc, in.y, r.o, r.d);
        tv.io = r.o;
        tv.id = r.d;
        tv.idd = in.y >= 0 ? len2(r.d) : tv.dd;
        tv.br = make_box_ray(r);
        tv.ref = in.x;
    } else { // leave
        tv.cur_inst = -1;
        tv.br = make_box_ray(world);
        trav_pop<ST, SPILL>(tv, st, sc.ref_bits);
    }
}

// Whole traversal of one ray.
template <bool COUNT, class ST, bool SPILL = false, int NODE_F4 = RT_NODE_F4>
RT_HD TraceHit trace_bvh(const DevScene &sc, const SceneView &v, const Ray &r, float tmin, float tmax,
                         const ST &st, Cnt<COUNT> &cn) {
    Trav tv;
    trav_begin(tv, r, tmax);
    trav_test_big<COUNT>(tv, sc, v, r, tmin, cn);
    while (tv.ref != RT_TRAV_DONE) {
        if (tv.ref >= 0) {
            trav_inner<COUNT, ST, false, SPILL, NODE_F4>(tv, v.nodes, tmin, st, sc.ref_bits, cn);
        } else if (RT_FEAT_INSTANCE && sc.n_inst > 0 && trav_is_barrier(sc, tv.ref)) {
            trav_barrier<ST, SPILL>(tv, sc, r, st);
        } else {
            trav_test_leaf<COUNT>(tv, tv.ref, sc, v, r, tmin, cn);
            trav_pop<ST, SPILL>(tv, st, sc.ref_bits);
        }
    }
    TraceHit best;
    best.t = tv.best_t;
    best.slot = tv.best_slot;
    if (RT_FEAT_INSTANCE) best.inst = tv.best_inst;
    return best;
}

// ------------------------------------------------------------------------------------------
// Surface reconstruction for the winning primitive (the tail of sphere::hit /
// quad::hit, then the wrappers' way back: hittable.h:66,146-148,264-266,328-331).
struct Surf {
    f3 p, n;       // world-space hit point and shading normal (against the ray)
    f3 outward;    // spheres: object-space outward unit normal (for get_sphere_uv)
    float u, v;    // quads: alpha, beta
    i4 rec;        // the slot's {kind, texture, chain, prim id}
    bool front;
};

RT_HD void sphere_uv(f3 p, float &u, float &v) { // sphere.h:96-113
    float theta = acosf(-p.y);
    float phi = atan2f(-p.z, p.x) + RT_PI;
    u = phi / (2.0f * RT_PI);
    v = theta / RT_PI;
}

// EXACT: the reference's operations one by one (the hit records of the batch queries, which are
// compared with the reference's bit for bit); otherwise shading arithmetic (see sh_fma).
template <bool EXACT = true> RT_HD Surf surface_at(const DevScene &sc, const SceneView &vw, const Ray &r, TraceHit h) {
    Surf s;
    i4 m = vw.rec_i[h.slot];
    s.rec = m;
    f3 o = r.o, d = r.d;
    int inst_chain = -1; // the wrappers of the instance the primitive was hit in: the outermost ones
    if (RT_FEAT_INSTANCE && sc.n_inst > 0 && h.inst >= 0) inst_chain = sc.inst[h.inst].y;
    if (RT_FEAT_INSTANCE && inst_chain >= 0) chain_ray_to_object(sc, inst_chain, o, d);
    if (RT_FEAT_CHAIN && m.z >= 0) chain_ray_to_object(sc, m.z, o, d);
    const f4 *g = vw.slots + slot_words(sc, h.slot);
    f3 p = EXACT ? o + h.t * d : sh_madd(h.t, d, o); // ray.h:20
    f3 outward;
    s.u = 0.0f;
    s.v = 0.0f;
    if (!RT_FEAT_QUAD || RT_KIND_PRIM(m.x) == RT_PRIM_SPHERE) {
        f4 g0 = g[0], g1 = g[1];
        f3 center = EXACT ? xyz(g0) + r.tm * xyz(g1) : sh_madd(r.tm, xyz(g1), xyz(g0));
        outward = g1.w * (p - center); // sphere.h:68 (g1.w = 1 / radius)
    } else {
        f4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
        outward = xyz(g0);
        f3 planar = p - xyz(g1);
        f3 w = mk(g1.w, g2.x, g2.y);
        f3 u = mk(g2.z, g2.w, g3.x);
        f3 v = mk(g3.y, g3.z, g3.w);
        s.u = dot(w, cross(planar, v));
        s.v = dot(w, cross(u, planar));
    }
    s.outward = outward;
    s.front = (EXACT ? dot(d, outward) : sh_dot(d, outward)) < 0; // hittable.h:20-25
    f3 n = s.front ? outward : -outward;
    if (RT_FEAT_CHAIN && m.z >= 0) chain_hit_to_world(sc, m.z, p, n);
    if (RT_FEAT_INSTANCE && inst_chain >= 0) chain_hit_to_world(sc, inst_chain, p, n);
    s.p = p;
    s.n = n;
    return s;
}

// ------------------------------------------------------------------------------------------
// Textures (texture.h:16-95, perlin.h:17-95)
RT_HD float perlin_noise(const DevPerlin &pn, f3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = int(fx), j = int(fy), k = int(fz);
    float uu = u * u * sh_fma(-2.0f, u, 3.0f);
    float vv = v * v * sh_fma(-2.0f, v, 3.0f);
    float ww = w * w * sh_fma(-2.0f, w, 3.0f);
    // the reference hashes all three axes through perm_x (perlin.h:33)
    int pi0 = pn.perm_x[i & 255], pi1 = pn.perm_x[(i + 1) & 255];
    int pj0 = pn.perm_x[j & 255], pj1 = pn.perm_x[(j + 1) & 255];
    int pk0 = pn.perm_x[k & 255], pk1 = pn.perm_x[(k + 1) & 255];
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; di++)
#pragma unroll
        for (int dj = 0; dj < 2; dj++)
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                int idx = (di ? pi1 : pi0) ^ (dj ? pj1 : pj0) ^ (dk ? pk1 : pk0);
                f3 c = xyz(pn.randvec[idx]);
                f3 wv = mk(u - di, v - dj, w - dk);
                // perlin.h:75-77 weighs a corner with (i*uu + (1-i)*(1-uu)) ... for i, j, k in {0, 1}: one
                // of the two products is 0 * x = 0 and the other 1 * y = y (uu, vv, ww lie in [0, 1]),
                // and y + 0 = y, so each factor is uu or 1 - uu bit for bit; the compiler may not fold
                // that itself (0 * x is not 0 for every x), so it is written out
                accum = sh_fma((di ? uu : 1 - uu) * (dj ? vv : 1 - vv) * (dk ? ww : 1 - ww), sh_dot(c, wv), accum);
            }
    return accum;
}
RT_HD float perlin_turb(const DevPerlin &pn, f3 p, int depth) {
    float accum = 0.0f;
    float weight = 1.0f;
    for (int i = 0; i < depth; i++) {
        accum = sh_fma(weight, perlin_noise(pn, p), accum);
        weight *= 0.5f;
        p = mk(p.x * 2, p.y * 2, p.z * 2);
    }
    return fabsf(accum);
}

// Value of a non-solid texture (solid colours never get here: they are inlined
// into the slot record).
template <bool COUNT>
RT_HD f3 texture_value(const DevScene &sc, const SceneView &vw, int tex, const Surf &s, Cnt<COUNT> &cn) {
    rt_texture t = vw.textures[tex];
    for (int guard = 0; guard < 16 && t.type == RT_TEX_CHECKER; guard++) { // texture.h:40-47
        RT_COUNT(chk);
        int xi = int(floorf(t.scale * s.p.x));
        int yi = int(floorf(t.scale * s.p.y));
        int zi = int(floorf(t.scale * s.p.z));
        // (x + y + z) % 2 == 0 of texture.h:45 by the sum's lowest bit, which a sum that leaves the
        // range of int keeps
        bool is_even = (((unsigned)xi + (unsigned)yi + (unsigned)zi) & 1u) == 0u;
        t = vw.textures[is_even ? t.even : t.odd];
    }
    if (t.type == RT_TEX_SOLID) return mk(t.color[0], t.color[1], t.color[2]);
    if (RT_FEAT_NOISE && t.type == RT_TEX_NOISE) { // texture.h:85-90
        RT_COUNT(noise);
        float ph = sh_fma(10.0f, perlin_turb(vw.perlins[t.perlin], s.p, 7), t.scale * s.p.z);
        return mk(0.5f, 0.5f, 0.5f) * (1.0f + sinf(ph));
    }
    if (RT_FEAT_IMAGE && t.type == RT_TEX_IMAGE) { // texture.h:60-74, rtw_stb_image.h:69-94
        RT_COUNT(img);
        if (t.image < 0) return mk(0.0f, 0.0f, 1.0f);
        float u = s.u, v = s.v;
        if (RT_KIND_PRIM(s.rec.x) == RT_PRIM_SPHERE) sphere_uv(s.outward, u, v);
        rt_image im = sc.images[t.image];
        u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
        float vc = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        v = 1.0f - vc;
        int i = int(u * im.width);
        int j = int(v * im.height);
        i = i < 0 ? 0 : (i < im.width ? i : im.width - 1);
        j = j < 0 ? 0 : (j < im.height ? j : im.height - 1);
        const uint8_t *px = sc.texels + im.offset + ((size_t)j * im.width + i) * 3;
        float cs = 1.0f / 255.0f;
        return mk(cs * px[0], cs * px[1], cs * px[2]);
    }
    return mk(0.0f, 0.0f, 0.0f);
}

// ------------------------------------------------------------------------------------------
// Camera ray (camera.h:113-137)
RT_HD Ray camera_ray(const DevCamera &cam, int i, int j, Rng &rng) {
    const rt_camera_frame &f = cam.f;
    float dx = rng_next(rng) - 0.5f;
    float dy = rng_next(rng) - 0.5f;
    f3 p00 = mk(f.pixel00_loc[0], f.pixel00_loc[1], f.pixel00_loc[2]);
    f3 du = mk(f.pixel_delta_u[0], f.pixel_delta_u[1], f.pixel_delta_u[2]);
    f3 dv = mk(f.pixel_delta_v[0], f.pixel_delta_v[1], f.pixel_delta_v[2]);
    f3 pixel_sample = sh_madd(j + dy, dv, sh_madd(i + dx, du, p00));
    f3 center = mk(f.center[0], f.center[1], f.center[2]);
    f3 origin = center;
    if (!(cam.defocus_angle <= 0)) {
        float px, py;
        sample_unit_disk(rng, px, py);
        f3 ddu = mk(f.defocus_disk_u[0], f.defocus_disk_u[1], f.defocus_disk_u[2]);
        f3 ddv = mk(f.defocus_disk_v[0], f.defocus_disk_v[1], f.defocus_disk_v[2]);
        origin = sh_madd(py, ddv, sh_madd(px, ddu, center));
    }
    Ray r;
    r.o = origin;
    r.d = pixel_sample - origin;
    r.tm = rng_next(rng);
    return r;
}

// ------------------------------------------------------------------------------------------
// One bounce of ray_color (camera.h:139-174) in iterative form: radiance
// L += beta * emitted, beta *= attenuation.  The recursion's
// e0 + a0*(e1 + a1*(...)) and this running product differ by rounding only.
struct PathState {
    Ray ray;
    f3 beta, L;
    Rng rng;
    int depth; // bounces left
};

RT_HD f3 reflect(f3 v, f3 n) { return sh_madd(-(2.0f * sh_dot(v, n)), n, v); } // vec3.h:163
RT_HD f3 refract(f3 uv, f3 n, float etai_over_etat) {                          // vec3.h:169
    float cos_theta = fminf(sh_dot(-uv, n), 1.0f);
    f3 r_out_perp = etai_over_etat * sh_madd(cos_theta, n, uv);
    return sh_madd(-sh_sqrt(fabsf(1.0f - sh_dot(r_out_perp, r_out_perp))), n, r_out_perp);
}
RT_HD float schlick_r0(float cosine, float r0) { // material.h:101-102, r0 given
    float x = 1.0f - cosine;
    float x2 = x * x;
    return sh_fma(1.0f - r0, x2 * x2 * x, r0);
}
RT_HD float schlick(float cosine, float ri) { // material.h:97-103
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    return schlick_r0(cosine, r0);
}

// Returns true while the path continues.
template <bool COUNT>
RT_HD bool shade_hit(const DevScene &sc, const SceneView &vw, int slot, PathState &ps, const Surf &s,
                     Cnt<COUNT> &cn) {
    const int kind = s.rec.x;
    const int mat = RT_KIND_MAT(kind);
    const f4 rf = vw.rec_f[slot];
    // colour of the material's texture: inlined when solid (dielectric: see below)
    f3 col = mk(rf.x, rf.y, rf.z);
    if ((mat == RT_MAT_LAMBERTIAN || mat == RT_MAT_DIFFUSE_LIGHT) && RT_KIND_TEX(kind) != RT_TEX_SOLID)
        col = texture_value<COUNT>(sc, vw, s.rec.y, s, cn);
    if (mat == RT_MAT_DIFFUSE_LIGHT) { // material.h:113-115: emits on both faces, never scatters
        RT_COUNT(sh3);
        ps.L = sh_madd3(ps.beta, col, ps.L);
        return false;
    }
    // The three scattering materials share what is expensive in them: every one draws a first
    // random number (z of the random unit vector of lambertian and metal, material.h:30-60; the
    // number the reflectance is compared with for a dielectric, material.h:86), metal and
    // dielectric normalise one vector each (the reflected direction, material.h:56; the incoming
    // one, material.h:77) and each takes one square root (vec3.h:147-148's normalisation as
    // sample_unit_vector restates it; sin_theta, material.h:80).  A wave that shades all three
    // kinds at once (nearly every wave of a scene that has them) runs those once for all its
    // lanes instead of once per kind; each lane gets the value its own material's code computes.
    const bool diel = mat == RT_MAT_DIELECTRIC, lamb = mat == RT_MAT_LAMBERTIAN;
    const float u1 = rng_next(ps.rng);
    f3 vu = mk(0.0f, 0.0f, 0.0f);
    if (RT_WAVE_ANY(!lamb)) { // (scenes of lambertian surfaces only skip this as a wave)
        const f3 reflected = reflect(ps.ray.d, s.n);
        vu = sh_unit(diel ? ps.ray.d : reflected);
    }
    const float cos_theta = fminf(sh_dot(-vu, s.n), 1.0f);
    const float z = sh_fma(-2.0f, u1, 1.0f);
    const float root = sh_sqrt(diel ? sh_fma(-cos_theta, cos_theta, 1.0f) : fmaxf(0.0f, sh_fma(-z, z, 1.0f)));
    f3 dir;
    if (diel) { // material.h:72-90
        RT_COUNT(sh2);
        // the record holds {1/ri, r0 seen from outside, r0 seen from inside, ri}: the host has done
        // the divisions of material.h:76 and 99, in the same fp32 arithmetic
        float ri = s.front ? rf.x : rf.w;
        float r0 = s.front ? rf.y : rf.z;
        col = mk(1.0f, 1.0f, 1.0f);
        const f3 unit_direction = vu;
        const float sin_theta = root;
        bool cannot_refract = ri * sin_theta > 1.0f;
        if (cannot_refract || schlick_r0(cos_theta, r0) > u1)
            dir = reflect(unit_direction, s.n);
        else
            dir = refract(unit_direction, s.n, ri);
    } else {
        // lambertian (material.h:30-42) and metal (material.h:53-60) both draw one
        // random unit vector
        float sn, cs;
        sincos_2pi(rng_next(ps.rng), sn, cs);
        const f3 rv = mk(root * cs, root * sn, z);
        if (lamb) {
            RT_COUNT(sh0);
            dir = s.n + rv;
            if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = s.n;
        } else {
            RT_COUNT(sh1);
            dir = sh_madd(rf.w, rv, vu);
            if (!(sh_dot(dir, s.n) > 0)) return false; // absorbed: contributes nothing
        }
    }
    ps.beta = ps.beta * col;
    ps.ray.o = s.p;
    ps.ray.d = dir;
    return true;
}

// Starts the path of (pixel, global sample index): camera.h:44-46.
RT_HD PathState path_start(const DevCamera &cam, uint32_t seed, unsigned pixel, int px, int py, unsigned sample) {
    PathState ps;
    ps.rng = rng_seed(seed, pixel, sample);
    ps.ray = camera_ray(cam, px, py, ps.rng);
    ps.beta = mk(1.0f, 1.0f, 1.0f);
    ps.L = mk(0.0f, 0.0f, 0.0f);
    ps.depth = cam.max_depth;
    return ps;
}

// The part of a ray_color segment that follows the closest-hit query: miss ->
// background (camera.h:148-161), otherwise emission / scatter (camera.h:164-173).
// Returns true while the path continues (ps.ray then holds the scattered ray).
// (the two halves are also called on their own by the kernel that settles misses at once and lets
// hits wait for company)
RT_HD void path_missed(const DevCamera &cam, PathState &ps) {
    ps.L = sh_madd3(ps.beta, mk(cam.background[0], cam.background[1], cam.background[2]), ps.L);
}
template <bool COUNT>
RT_HD bool path_after_hit(const DevScene &sc, const SceneView &vw, PathState &ps, TraceHit h, Cnt<COUNT> &cn) {
    Surf s = surface_at<false>(sc, vw, ps.ray, h);
    ps.depth--;
    return shade_hit<COUNT>(sc, vw, h.slot, ps, s, cn) && ps.depth > 0; // depth 0: camera.h:141 returns black
}
template <bool COUNT>
RT_HD bool path_after_trace(const DevScene &sc, const SceneView &vw, const DevCamera &cam, PathState &ps,
                            TraceHit h, Cnt<COUNT> &cn) {
    if (h.slot < 0) {
        path_missed(cam, ps);
        return false;
    }
    return path_after_hit<COUNT>(sc, vw, ps, h, cn);
}

// One whole segment of ray_color: trace, then miss / emit / scatter.  Returns true
// while the path continues; when it returns false ps.L holds the path's radiance.
template <int TRAV, bool COUNT, class ST, bool SPILL, int NODE_F4 = RT_NODE_F4>
RT_HD bool path_step(const DevScene &sc, const SceneView &vw, const DevCamera &cam, PathState &ps,
                     const ST &st, unsigned &n_rays, Cnt<COUNT> &cn) {
    if (ps.depth <= 0) return false; // camera.h:141
    n_rays++;
    TraceHit h;
    if constexpr (TRAV == TRAV_LINEAR)
        h = trace_linear<COUNT>(sc, vw, ps.ray, 0.001f, RT_INF, cn);
    else
        h = trace_bvh<COUNT, ST, SPILL, NODE_F4>(sc, vw, ps.ray, 0.001f, RT_INF, st, cn);
    return path_after_trace<COUNT>(sc, vw, cam, ps, h, cn);
}

} // namespace rtc
