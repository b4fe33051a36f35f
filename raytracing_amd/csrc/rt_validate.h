// Checks of a scene descriptor that come before anything reads through its pointers' contents:
// counts, index ranges, known enumerators.  Shared by the library and by host-only test harnesses.
#pragma once
#include "rt_amd.h"
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

namespace rtf {

inline int validate_desc(const rt_scene_desc *d, std::string &err) {
    auto fail = [&err](int code, const char *fmt, ...) __attribute__((format(printf, 3, 4))) {
        char buf[256];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    };
    if (!d) return fail(RT_ERR_ARG, "scene descriptor is null");
    if (d->abi_version != RT_AMD_ABI_VERSION)
        return fail(RT_ERR_ARG, "descriptor abi_version %u, library %d", d->abi_version, RT_AMD_ABI_VERSION);
    // every count non-negative, every non-empty array present
    const struct {
        const char *name;
        long long n;
        const void *p;
    } arrays[] = {{"prims", d->n_prims, d->prims},           {"groups", d->n_groups, d->groups},
                  {"materials", d->n_materials, d->materials}, {"textures", d->n_textures, d->textures},
                  {"perlins", d->n_perlins, d->perlins},       {"images", d->n_images, d->images},
                  {"xform_ops", d->n_xform_ops, d->xform_ops}, {"chains", d->n_chains, d->chains},
                  {"instances", d->n_instances, d->instances}};
    for (const auto &a : arrays) {
        if (a.n < 0) return fail(RT_ERR_ARG, "negative count of %s", a.name);
        if (a.n > 0 && !a.p) return fail(RT_ERR_ARG, "%s: %lld entries but a null array", a.name, a.n);
    }
    if (d->n_texel_bytes > 0 && !d->texels) return fail(RT_ERR_ARG, "texels: bytes but a null array");
    if (d->n_texel_bytes > (1ull << 40)) return fail(RT_ERR_ARG, "texels: too many bytes");
    if (d->n_prims >= (1 << 29)) return fail(RT_ERR_ARG, "too many primitives");
    for (int i = 0; i < d->n_prims; i++) {
        const rt_prim &p = d->prims[i];
        if (p.type != RT_PRIM_SPHERE && p.type != RT_PRIM_QUAD)
            return fail(RT_ERR_ARG, "prim %d: unknown type %d", i, p.type);
        if (p.material < 0 || p.material >= d->n_materials)
            return fail(RT_ERR_ARG, "prim %d: material %d out of range", i, p.material);
        if (p.xform < -1 || p.xform >= d->n_chains) return fail(RT_ERR_ARG, "prim %d: chain out of range", i);
    }
    for (int i = 0; i < d->n_groups; i++) {
        const rt_group &g = d->groups[i];
        if ((g.kind != RT_GROUP_LIST && g.kind != RT_GROUP_BVH) || g.first_prim < 0 || g.n_prims < 0 ||
            (long long)g.first_prim + g.n_prims > d->n_prims)
            return fail(RT_ERR_ARG, "group %d: bad kind or primitive range", i);
    }
    for (int i = 0; i < d->n_materials; i++) {
        const rt_material &m = d->materials[i];
        if (m.type < RT_MAT_LAMBERTIAN || m.type > RT_MAT_DIFFUSE_LIGHT)
            return fail(RT_ERR_ARG, "material %d: unknown type %d", i, m.type);
        if ((m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_DIFFUSE_LIGHT) &&
            (m.texture < 0 || m.texture >= d->n_textures))
            return fail(RT_ERR_ARG, "material %d: texture out of range", i);
    }
    for (int i = 0; i < d->n_textures; i++) {
        const rt_texture &t = d->textures[i];
        if (t.type < RT_TEX_SOLID || t.type > RT_TEX_NOISE)
            return fail(RT_ERR_ARG, "texture %d: unknown type %d", i, t.type);
        if (t.type == RT_TEX_CHECKER &&
            (t.even < 0 || t.even >= d->n_textures || t.odd < 0 || t.odd >= d->n_textures))
            return fail(RT_ERR_ARG, "texture %d: checker child out of range", i);
        if (t.type == RT_TEX_NOISE && (t.perlin < 0 || t.perlin >= d->n_perlins))
            return fail(RT_ERR_ARG, "texture %d: perlin table out of range", i);
        if (t.type == RT_TEX_IMAGE && (t.image < -1 || t.image >= d->n_images)) // -1: file missing (texture.h:62)
            return fail(RT_ERR_ARG, "texture %d: image out of range", i);
    }
    for (int i = 0; i < d->n_perlins; i++) // the table is an index into randvec (perlin.h:33)
        for (int k = 0; k < 256; k++)
            if (d->perlins[i].perm_x[k] < 0 || d->perlins[i].perm_x[k] > 255)
                return fail(RT_ERR_ARG, "perlin table %d: perm_x[%d] = %d is not in 0..255", i, k, d->perlins[i].perm_x[k]);
    for (int i = 0; i < d->n_images; i++) {
        const rt_image &im = d->images[i];
        const uint64_t bytes = im.width > 0 && im.height > 0 ? (uint64_t)im.width * (uint64_t)im.height * 3u : 0;
        if (bytes == 0 || im.offset > d->n_texel_bytes || bytes > d->n_texel_bytes - im.offset)
            return fail(RT_ERR_ARG, "image %d: texels out of range", i);
    }
    for (int i = 0; i < d->n_xform_ops; i++)
        if (d->xform_ops[i].type < RT_XF_TRANSLATE || d->xform_ops[i].type > RT_XF_SCALE)
            return fail(RT_ERR_ARG, "wrapper op %d: unknown type %d", i, d->xform_ops[i].type);
    for (int i = 0; i < d->n_chains; i++) {
        const rt_xform_chain &c = d->chains[i];
        if (c.first_op < 0 || c.n_ops < 0 || (long long)c.first_op + c.n_ops > d->n_xform_ops)
            return fail(RT_ERR_ARG, "chain %d: ops out of range", i);
    }
    // instances: ranges inside the primitives, equal or disjoint; chains in range
    if (d->n_instances > 0) {
        std::vector<std::pair<int, int>> ranges;
        for (int i = 0; i < d->n_instances; i++) {
            const rt_instance &in = d->instances[i];
            if (in.first_prim < 0 || in.n_prims < 1 || (long long)in.first_prim + in.n_prims > d->n_prims)
                return fail(RT_ERR_ARG, "instance %d: primitive range out of bounds", i);
            if (in.xform < -1 || in.xform >= d->n_chains) return fail(RT_ERR_ARG, "instance %d: chain out of range", i);
            ranges.push_back({in.first_prim, in.n_prims});
        }
        std::sort(ranges.begin(), ranges.end());
        for (size_t i = 1; i < ranges.size(); i++)
            if (ranges[i] != ranges[i - 1] && ranges[i].first < ranges[i - 1].first + ranges[i - 1].second)
                return fail(RT_ERR_ARG, "instances: two different primitive ranges overlap");
    }
    return RT_OK;
}

} // namespace rtf
