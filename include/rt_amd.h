/*
 * rt_amd.h — C ABI of the MI355X path-tracing library (librt_amd.so).
 *
 * The reference (YanjieShiFennec/RayTracing) has no FFI layer: its hot path is
 * entered through the C++ call camera::render(world, file) (camera.h:31).  This
 * ABI is the boundary that call is replaced by.  A host program (the drop-in
 * C++ scene API under raytracing_amd/dropin, or any FFI) flattens its scene
 * graph into an rt_scene_desc, uploads it once with rt_scene_create, and asks
 * for samples with rt_render*.  Plain pointers and sizes only.
 *
 * Every function returns 0 on success and a negative RT_ERR_* code on failure;
 * rt_last_error() returns a thread-local description of the last failure.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_AMD_ABI_VERSION 4

/* ---- error codes ------------------------------------------------------- */
enum {
    RT_OK = 0,
    RT_ERR_ARG = -1,         /* bad argument / malformed descriptor          */
    RT_ERR_HIP = -2,         /* a HIP runtime call failed (no GPU, OOM, ...) */
    RT_ERR_UNSUPPORTED = -3, /* scene uses a feature outside the hot path    */
    RT_ERR_IO = -4           /* scene file could not be read / written       */
};

/* ---- scene vocabulary (mirrors the reference's classes) ---------------- */
enum { RT_PRIM_SPHERE = 0, RT_PRIM_QUAD = 1 };                       /* sphere.h:8, quad.h:9 */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2,
       RT_MAT_DIFFUSE_LIGHT = 3 };                                   /* material.h:24,49,68,107 */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_IMAGE = 2,
       RT_TEX_NOISE = 3 };                                           /* texture.h:16,31,56,81 */
enum { RT_XF_TRANSLATE = 0, RT_XF_ROTATE_Y = 1, RT_XF_ROTATE_XYZ = 2,
       RT_XF_SCALE = 3 };                                            /* hittable.h:51,100,212,313 */
enum { RT_GROUP_LIST = 0, RT_GROUP_BVH = 1 };                        /* hittable_list.h:14, bvh.h:39 */

/*
 * One leaf primitive, in the order the reference would test it ("test order":
 * insertion order inside a hittable_list, left-to-right leaf order inside a
 * bvh_node).  The position of a primitive in rt_scene_desc::prims is its id;
 * closest-hit ties are resolved as the reference's sequential scan would
 * (sphere.h:58 open interval, quad.h:46 closed interval).
 *
 *   sphere: a = center1, b = center_vec (center2 - center1), c[0] = radius,
 *           d = center2 as given to the moving-sphere constructor (only its
 *           bounding box is built from it), flags bit0 = is_moving (sphere.h:11-26)
 *   quad  : a = Q, b = u, c = v (object space, as passed to the constructor;
 *           normal, D and w are derived by the library as quad.h:15-20 does)
 */
typedef struct rt_prim {
    int32_t type;
    int32_t material; /* index into rt_scene_desc::materials */
    int32_t xform;    /* index into rt_scene_desc::chains, or -1 */
    int32_t flags;
    float a[3];
    float b[3];
    float c[3];
    float d[3];
} rt_prim; /* 64 bytes */

#define RT_PRIM_FLAG_MOVING 1

/* lambertian(tex) | metal(albedo, fuzz) | dielectric(ior) | diffuse_light(tex) */
typedef struct rt_material {
    int32_t type;
    int32_t texture; /* lambertian / diffuse_light: index into textures; else -1 */
    float albedo[3]; /* metal */
    float param;     /* metal: fuzz (already clamped to <= 1); dielectric: refraction index */
} rt_material;

/* solid(color) | checker(inv_scale, even, odd) | image(image) | noise(scale, perlin) */
typedef struct rt_texture {
    int32_t type;
    int32_t even, odd; /* checker: child texture indices */
    int32_t perlin;    /* noise: index into perlins */
    int32_t image;     /* image: index into images, -1 = file missing (texture.h:62) */
    float color[3];    /* solid */
    float scale;       /* checker: inv_scale (texture.h:33); noise: scale (texture.h:83) */
} rt_texture;

/* perlin.h:55-56 — only randvec and perm_x are read by perlin::noise (perlin.h:33). */
typedef struct rt_perlin {
    float randvec[256][3];
    int32_t perm_x[256];
} rt_perlin;

/* rtw_stb_image.h:69-79 — 3 bytes per texel, row-major, already in the
 * reference's "linear byte" form (gamma-2.2 decoded then 256*v quantised). */
typedef struct rt_image {
    int32_t width, height;
    uint64_t offset; /* byte offset of texel (0,0) in rt_scene_desc::texels */
} rt_image;

typedef struct rt_xform_op {
    int32_t type;
    float p[3]; /* translate: offset; rotate_y: (deg,0,0); rotate_xyz: (a,b,g) deg; scale: factors */
} rt_xform_op;

/* A chain lists the wrappers around a primitive, innermost first:
 * scaling(translate(rotate_xyz(box))) -> {rotate_xyz, translate, scale}. */
typedef struct rt_xform_chain {
    int32_t first_op, n_ops;
} rt_xform_chain;

/* Consecutive runs of prims: a plain list run, or the leaves of one bvh_node
 * (bvh.h:48-75: the tree is the recursive median split of that run). */
typedef struct rt_group {
    int32_t kind;
    int32_t first_prim, n_prims;
    int32_t reserved;
} rt_group;

/* A sub-scene placed in the world several times: what the reference builds when one hittable
 * (a list, a bvh_node, a primitive) is handed to several instance wrappers (hittable.h:51-344:
 * translate / rotate_y / rotate_xyz / scaling keep a shared_ptr to the object they move).
 * prims [first_prim, first_prim + n_prims) are the sub-scene, in its own space (each may still
 * carry wrappers of its own in `xform`); they are not part of the world by themselves.  Instance i
 * puts them into the world under the wrapper chain `xform` (-1: as they are).  The ranges of two
 * instances are either the same or disjoint.  The library keeps one copy of the primitives and
 * one tree per distinct range and a tree over the instances (a two-level BVH): memory and build
 * time do not grow with the number of instances.  Results equal those of the same scene with
 * every instance written out as primitives (loose primitives first, in their order, then the
 * instances in theirs: that order decides ties between equal hits); rt_hit.prim names the
 * primitive of the sub-scene.  A range holds primitives only: a shared object inside a shared
 * object is given as one more instance of the world per placement of the outer object, whose
 * chain is the inner object's wrappers followed by the outer placement's (innermost first, as in
 * every chain), so that a ray passes the operations in the order of the reference's nested hit()
 * calls. */
typedef struct rt_instance {
    int32_t first_prim, n_prims;
    int32_t xform; /* index into chains[], -1 = none */
    int32_t reserved;
} rt_instance;

typedef struct rt_scene_desc {
    uint32_t abi_version; /* RT_AMD_ABI_VERSION */
    int32_t n_prims;
    const rt_prim *prims;
    int32_t n_groups;
    const rt_group *groups;
    int32_t n_materials;
    const rt_material *materials;
    int32_t n_textures;
    const rt_texture *textures;
    int32_t n_perlins;
    const rt_perlin *perlins;
    int32_t n_images;
    const rt_image *images;
    uint64_t n_texel_bytes;
    const uint8_t *texels;
    int32_t n_xform_ops;
    const rt_xform_op *xform_ops;
    int32_t n_chains;
    const rt_xform_chain *chains;
    int32_t n_instances; /* 0: every primitive is part of the world */
    const rt_instance *instances;
} rt_scene_desc;

/* ---- camera (camera.h:15-29 public fields, same defaults expected) ------ */
typedef struct rt_camera_desc {
    float aspect_ratio;
    int32_t image_width;
    int32_t samples_per_pixel;
    int32_t max_depth;
    float background[3];
    float vfov;
    float lookfrom[3], lookat[3], vup[3];
    float defocus_angle, focus_dist;
} rt_camera_desc;

/* Result of camera::initialize (camera.h:76-111). */
typedef struct rt_camera_frame {
    int32_t image_width, image_height;
    float pixel_samples_scale;
    float center[3], pixel00_loc[3], pixel_delta_u[3], pixel_delta_v[3];
    float u[3], v[3], w[3];
    float defocus_disk_u[3], defocus_disk_v[3];
} rt_camera_frame;

/* ---- rays / hits for the parity entry point ----------------------------- */
typedef struct rt_ray {
    float orig[3];
    float dir[3];
    float tm;
} rt_ray;

/* hit_record (hittable.h:10-26) plus the id of the closest primitive. */
typedef struct rt_hit {
    int32_t prim; /* index into rt_scene_desc::prims, -1 on miss */
    int32_t front_face;
    float t;
    float p[3];
    float normal[3];
    float u, v;
} rt_hit;

/* ---- render options / statistics --------------------------------------- */
enum { RT_MODE_MEGAKERNEL = 0, RT_MODE_WAVEFRONT = 1 };

#define RT_RENDER_COUNTERS 1u      /* fill the per-test counters of rt_render_stats */
#define RT_RENDER_NO_OVERLAP 2u    /* keep this call's launches on the caller's stream: no overlap
                                    * with the launches before and after (isolated kernel timing) */
#define RT_SCENE_FORCE_BVH 1u      /* rt_scene_create: build a BVH even for tiny scenes */
#define RT_SCENE_FORCE_LINEAR 2u   /* never build a BVH (linear scan in test order)   */
#define RT_SCENE_FORCE_GLOBAL 4u   /* keep scene arrays in HBM even if they fit LDS   */
#define RT_SCENE_BUILD_DEVICE 8u   /* build the BVH on the GPU (linear BVH: fast to build,
                                      a little slower to traverse than the host's SAH tree) */

typedef struct rt_render_opts {
    int32_t sample_begin; /* first global sample index rendered by this call */
    int32_t sample_count; /* samples per pixel rendered by this call         */
    uint32_t seed;        /* stream seed of the counter-based sampler        */
    int32_t mode;         /* RT_MODE_*                                        */
    uint32_t flags;       /* RT_RENDER_*                                      */
    int32_t reserved[3];
} rt_render_opts;

typedef struct rt_render_stats {
    uint64_t paths;        /* camera samples traced                          */
    uint64_t rays;         /* closest-hit queries (camera + bounce segments) */
    uint64_t box_tests;    /* the rest only with RT_RENDER_COUNTERS          */
    uint64_t sphere_tests;
    uint64_t sphere_hits;
    uint64_t quad_tests;
    uint64_t shades[4];    /* per RT_MAT_*                                   */
    uint64_t noise_evals, image_evals, checker_evals;
    /* wave-level schedule of the render kernel (RT_RENDER_COUNTERS, BVH scenes):
     * node-step executions and the lanes active in them, leaf-step executions / lanes,
     * shading phases / lanes */
    uint64_t sched[6];
    double kernel_ms;      /* HIP-event time of the render kernels of this call */
    /* wave cycles of the decoupled render kernel (RT_RENDER_COUNTERS, BVH scenes), summed
     * over waves: whole kernel, shading, starting paths, node steps, leaf tests */
    uint64_t cycles[6];
    /* wall-clock profile of the decoupled kernel's launch, microseconds after the first wave
     * started: first / last wave that found the launch out of new paths, last wave exit */
    double tail_us[3];
    /* how alike the lanes of a wave are in a node step of the decoupled kernel (RT_RENDER_COUNTERS,
     * BVH scenes): lanes that took part in node steps, and those of them that were at the same
     * node as the wave's first active lane (the share a wave-uniform fetch of that node would serve) */
    uint64_t coherence[2];
} rt_render_stats;

typedef struct rt_scene_info {
    int32_t n_prims, n_spheres, n_quads;
    int32_t n_nodes;      /* nodes of the 4-wide tree; 0 = linear scan */
    int32_t bvh_depth;    /* its inner levels */
    int32_t lds_resident; /* 1 if geometry is staged into LDS by the kernels */
    uint64_t device_bytes;
    float build_ms;       /* flattening + BVH build (host or device), without the upload */
    int32_t built_on_device; /* 1 if the BVH came from the device builder */
    int32_t n_upfront;    /* primitives kept out of the tree because their box covers most of the scene
                             (a ground sphere): every ray tests them before it walks the tree */
} rt_scene_info;

typedef struct rt_scene rt_scene; /* opaque: owns the device copy of one scene */

/* ---- entry points -------------------------------------------------------- */

const char *rt_last_error(void);
int rt_abi_version(void);

/* Number of visible HIP devices (0 if none / no driver). */
int rt_device_count(void);

/* camera::initialize (camera.h:76-111), same fp32 arithmetic. Host only. */
int rt_camera_initialize(const rt_camera_desc *cam, rt_camera_frame *out);

/* Replaces the scene half of camera::render's prologue: packs the instance
 * wrappers (hittable.h:51-344) for use at intersection time, derives the quad
 * constants (quad.h:15-20), builds the traversal structure that stands in for
 * bvh_node (bvh.h:48-75) / hittable_list (hittable_list.h:29-43) and copies
 * everything to `device`.  The descriptor may be freed afterwards. */
int rt_scene_create(const rt_scene_desc *desc, int device, uint32_t flags, rt_scene **out);
int rt_scene_destroy(rt_scene *scene);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);
/* Diagnostic (no counterpart in the reference): copies the nodes of the scene's 4-wide BVH back
 * to the host, n_nodes x 128 bytes = 32 words: words 8a + k and 8a + 4 + k = the lower and the
 * upper plane of child k's box along axis a (a = 0, 1, 2: the planes of the four children are
 * grouped by axis), words 24..27 = the four child references (int; >= 0 inner node, < 0 leaf
 * ~(slot*2 + type)), word 28 = number of children (int, 2..4; an unused child has an inverted
 * box, which no finite ray enters, and the reference of some real leaf), words 29..31 unused. */
int rt_scene_read_nodes(rt_scene *scene, void *out, uint64_t bytes);

/* Replaces world.hit(r, interval(tmin, tmax), rec) (camera.h:148) for a batch
 * of rays: closest hit per ray.  Host buffers; the library stages them. */
int rt_trace_batch(rt_scene *scene, const rt_ray *rays, int64_t n_rays, float tmin, float tmax,
                   rt_hit *hits);
/* Same with device-resident buffers, enqueued on `stream` (a hipStream_t). */
int rt_trace_batch_device(rt_scene *scene, const rt_ray *d_rays, int64_t n_rays, float tmin,
                          float tmax, rt_hit *d_hits, void *stream);

/* The same query (world.hit, camera.h:148; bvh_node::hit, bvh.h:77-88) answered by the render
 * kernel's own traversal code: its launch shape, its LDS copy of the scene (or of the top of a
 * tree that stays in HBM), its stacks and its interleaved scheduling of node steps and parked
 * leaf tests, with the caller's rays in place of paths.  Same results as rt_trace_batch; it
 * exists so that the reference's hit records can be checked against the code that renders.
 * Scenes that are scanned (no tree) have only one closest-hit routine: for them this is
 * rt_trace_batch. */
int rt_trace_batch_sched(rt_scene *scene, const rt_ray *rays, int64_t n_rays, float tmin, float tmax,
                         rt_hit *hits);
int rt_trace_batch_sched_device(rt_scene *scene, const rt_ray *d_rays, int64_t n_rays, float tmin,
                                float tmax, rt_hit *d_hits, void *stream);

/* The closest-hit query reduced to what decides parity with the reference (world.hit, camera.h:148:
 * whether anything is hit, which primitive, at which parameter): prim[i] = the primitive of ray i's
 * closest hit on (tmin, tmax) or -1, t[i] = its parameter (tmax for a miss), both bit for bit what
 * rt_trace_batch reports in rt_hit.prim and rt_hit.t.  Trees are walked by the render kernel's own
 * traversal (as rt_trace_batch_sched), scanned scenes by the scan.  No point, normal or texture
 * coordinates are derived, so this is the rate of intersection and traversal alone. */
int rt_closest_hits(rt_scene *scene, const rt_ray *rays, int64_t n_rays, float tmin, float tmax, int32_t *prim,
                    float *t);
int rt_closest_hits_device(rt_scene *scene, const rt_ray *d_rays, int64_t n_rays, float tmin, float tmax,
                           int32_t *d_prim, float *d_t, void *stream);

/* Replaces the sample loop of camera::render (camera.h:40-47) for samples
 * [sample_begin, sample_begin+sample_count) of every pixel: ADDS the radiance
 * sum of those samples to accum (image_height*image_width*3 floats, row-major
 * RGB).  The caller applies 1/spp and write_color (rt_tonemap_u8). */
int rt_render(rt_scene *scene, const rt_camera_desc *cam, const rt_render_opts *opts,
              float *accum_host, rt_render_stats *stats);
/* accum lives in device memory; kernels are enqueued on `stream`.  With
 * stats == NULL the megakernel call does not synchronise (RT_MODE_WAVEFRONT polls its
 * queue counters and does).  A scene owns the per-launch scratch (path records, queues), so
 * calls on one scene must be ordered: one stream at a time.  That scratch is up to three sets of
 * per-path records (16 bytes x samples x pixels of one launch each, a set bounded by the
 * environment's RT_AMD_STAGE_MB, default 2048 MB: larger sample counts are split into launches);
 * sets are allocated when a call first needs them and released by rt_scene_destroy. */
int rt_render_device(rt_scene *scene, const rt_camera_desc *cam, const rt_render_opts *opts,
                     float *d_accum, void *stream, rt_render_stats *stats);

/* No counterpart in the reference.  rt_render_device for callers that cannot allocate device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * The launch's scratch (pool / ray counters and the per-path radiance records) lives in one
 * statically allocated block of device memory inside the library instead of buffers owned by
 * the scene: this call makes no hipMalloc / hipFree and creates no streams or events;
 * everything is enqueued on `stream`.  Megakernel mode only, no statistics
 * (opts->flags must not have RT_RENDER_COUNTERS).
 * Limit: image_width * image_height <= RT_STATIC_MAX_PIXELS, else RT_ERR_ARG (larger sample
 * counts are split into several launches).  Every GPU holds one such block (64 MiB + 4 KiB)
 * from the first rt_scene_create on it until the process ends.  All calls on one GPU share
 * that GPU's block, whatever the scene or stream: calls that overlap on one GPU race — order
 * them on one stream, or synchronise between them.  Same image as rt_render_device. */
#define RT_STATIC_MAX_PIXELS (4 << 20)
int rt_render_device_static(rt_scene *scene, const rt_camera_desc *cam, const rt_render_opts *opts,
                            float *d_accum, void *stream);

/* write_color / linear_to_gamma (color.h:11-32): rgb8 = int(256*clamp(sqrt(scale*c),0,0.999)). */
int rt_tonemap_u8(const float *accum, int64_t n_pixels, float scale, uint8_t *rgb8);
/* The same on the device: d_accum and d_rgb8 are device pointers on `device`, the kernel is
 * enqueued on `stream` (the image can stay in HBM until the 3 bytes per pixel are wanted). */
int rt_tonemap_u8_device(int device, const float *d_accum, int64_t n_pixels, float scale, uint8_t *d_rgb8,
                         void *stream);

/* ---- film: a device-resident accumulation image rendered in passes ------
 * No counterpart in the reference (camera.h:39-57 keeps one host image and writes it once at
 * the end): a film holds the fp32 radiance sums of camera::render's pixel loop in HBM between
 * calls, so that a render can proceed in passes of a few samples per pixel, hand out 8-bit
 * snapshots (write_color, color.h:17-32) while the next pass is already running, and be
 * checkpointed / resumed through its sums and sample count.  One host thread per film. */
typedef struct rt_film rt_film;
int rt_film_create(int device, int32_t width, int32_t height, rt_film **out); /* zeroed */
int rt_film_destroy(rt_film *film);
/* rt_render_device into the film on the film's own stream (asynchronous): adds samples
 * [opts->sample_begin, +opts->sample_count) and counts them.  The camera's image size must be
 * the film's. */
int rt_film_render(rt_scene *scene, rt_film *film, const rt_camera_desc *cam, const rt_render_opts *opts);
/* Asynchronous: after everything rendered so far, tone-maps sums / samples to 3 bytes per pixel
 * on the device and copies them (and, if with_accum != 0, the sums themselves) to pinned host
 * memory on a second stream.  Passes rendered after this call overlap the copies; their
 * accumulation waits for the copy of the sums.  One snapshot at a time: a second call waits
 * for the first one's copies. */
int rt_film_snapshot(rt_film *film, int with_accum);
/* Waits for the last snapshot and copies it out: rgb8_out (width*height*3 bytes) and/or
 * accum_out (width*height*3 floats; only after a snapshot with_accum) may be NULL;
 * samples_out receives the samples per pixel the snapshot holds. */
int rt_film_wait(rt_film *film, uint8_t *rgb8_out, float *accum_out, int64_t *samples_out);
/* Replaces the film's contents (resume from a checkpoint); accum_host == NULL clears it.
 * Synchronises the film's stream. */
int rt_film_upload(rt_film *film, const float *accum_host, int64_t samples);

/* ---- several GPUs of one node behind one call -----------------------------
 * No counterpart in the reference, whose camera::render (camera.h:31-63) is one thread on one
 * CPU: the sample loop (camera.h:40-47) partitioned over GPUs.  rt_multi_create uploads a
 * replica of the scene to every listed device (a device may be listed more than once).
 * rt_multi_render splits samples [sample_begin, +sample_count) into contiguous balanced shares
 * (share k of n: begin = k*(c/n) + min(k, c%n), one more sample for k < c%n), renders share k
 * on device k (one host thread per device), copies the per-device fp32 images to the first
 * device (peer copies, one xGMI link per source), adds them there in device order and ADDS the
 * total to accum_host.  Every (pixel, sample) path is keyed by its global sample index, so the
 * image equals the one-device image up to the order of the fp32 additions.  stats (may be
 * NULL): paths and rays summed over devices, kernel_ms the slowest device's. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc *desc, const int32_t *devices, int32_t n_devices, uint32_t flags,
                    rt_multi **out);
int rt_multi_destroy(rt_multi *multi);
int rt_multi_render(rt_multi *multi, const rt_camera_desc *cam, const rt_render_opts *opts,
                    float *accum_host, rt_render_stats *stats);

/* Per-launch timing of the render kernel: between begin and end the library
 * brackets every render-kernel launch of this scene with HIP events on the
 * stream it is launched on; end waits for them and returns the summed
 * duration and the number of launches. */
int rt_profile_begin(rt_scene *scene);
int rt_profile_end(rt_scene *scene, double *total_ms, int *n_launches);

/* Scene files (.rtscene): a flat dump of rt_scene_desc used for fixtures. */
int rt_scene_file_save(const rt_scene_desc *desc, const char *path);
/* Loads into one malloc'd block; *out points into it; free with rt_scene_file_free. */
int rt_scene_file_load(const char *path, rt_scene_desc **out);
void rt_scene_file_free(rt_scene_desc *desc);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
