/*
 * rt_scene_io.h — header-only reader/writer of the .rtscene fixture format.
 *
 * A .rtscene file is a flat dump of an rt_scene_desc (rt_amd.h):
 *   char     magic[8] = "RTSCENE1"
 *   uint64_t counts[10] = { n_prims, n_groups, n_materials, n_textures, n_perlins,
 *                           n_images, n_texel_bytes, n_xform_ops, n_chains, n_instances }
 *   then the arrays in that order, each padded to a multiple of 8 bytes.
 * Shared by the library, the CPU oracle and the reference harness so that all
 * three exchange scenes through one format.
 */
#ifndef RT_SCENE_IO_H
#define RT_SCENE_IO_H

#include "rt_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rtio {

static const char kMagic[8] = {'R', 'T', 'S', 'C', 'E', 'N', 'E', '1'};

inline size_t pad8(size_t n) { return (n + 7u) & ~size_t(7); }

inline bool write_block(FILE *f, const void *p, size_t bytes) {
    static const char zeros[8] = {0};
    if (bytes && fwrite(p, 1, bytes, f) != bytes) return false;
    size_t pad = pad8(bytes) - bytes;
    return pad == 0 || fwrite(zeros, 1, pad, f) == pad;
}

inline int save(const rt_scene_desc *d, const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) return RT_ERR_IO;
    uint64_t counts[10] = {(uint64_t)d->n_prims,     (uint64_t)d->n_groups,
                           (uint64_t)d->n_materials, (uint64_t)d->n_textures,
                           (uint64_t)d->n_perlins,   (uint64_t)d->n_images,
                           d->n_texel_bytes,         (uint64_t)d->n_xform_ops,
                           (uint64_t)d->n_chains,    (uint64_t)d->n_instances};
    bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(counts, sizeof(uint64_t), 10, f) == 10;
    ok = ok && write_block(f, d->prims, sizeof(rt_prim) * d->n_prims);
    ok = ok && write_block(f, d->groups, sizeof(rt_group) * d->n_groups);
    ok = ok && write_block(f, d->materials, sizeof(rt_material) * d->n_materials);
    ok = ok && write_block(f, d->textures, sizeof(rt_texture) * d->n_textures);
    ok = ok && write_block(f, d->perlins, sizeof(rt_perlin) * d->n_perlins);
    ok = ok && write_block(f, d->images, sizeof(rt_image) * d->n_images);
    ok = ok && write_block(f, d->texels, d->n_texel_bytes);
    ok = ok && write_block(f, d->xform_ops, sizeof(rt_xform_op) * d->n_xform_ops);
    ok = ok && write_block(f, d->chains, sizeof(rt_xform_chain) * d->n_chains);
    ok = ok && write_block(f, d->instances, sizeof(rt_instance) * d->n_instances);
    ok = (fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_ERR_IO;
}

/* Loads the file into ONE malloc'd block that starts with the rt_scene_desc
 * itself; free() of the returned pointer releases everything. */
inline rt_scene_desc *load(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return nullptr;
    char magic[8];
    uint64_t c[10];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 ||
        fread(c, sizeof(uint64_t), 10, f) != 10) {
        fclose(f);
        return nullptr;
    }
    /* the counts come from the file: every element count must fit the descriptor's int32
     * fields, no size may wrap, and the arrays together must be what the file actually holds */
    static const size_t elem[10] = {sizeof(rt_prim),     sizeof(rt_group),    sizeof(rt_material),
                                    sizeof(rt_texture),  sizeof(rt_perlin),   sizeof(rt_image),
                                    1,                   sizeof(rt_xform_op), sizeof(rt_xform_chain),
                                    sizeof(rt_instance)};
    long here = ftell(f);
    if (here < 0 || fseek(f, 0, SEEK_END) != 0) {
        fclose(f);
        return nullptr;
    }
    long end = ftell(f);
    if (end < here || fseek(f, here, SEEK_SET) != 0) {
        fclose(f);
        return nullptr;
    }
    const uint64_t file_payload = (uint64_t)(end - here);
    size_t sz[10];
    uint64_t sum = 0;
    for (int i = 0; i < 10; i++) {
        const uint64_t limit = i == 6 ? file_payload : (uint64_t)0x7fffffff;
        if (c[i] > limit || c[i] > file_payload / elem[i]) { /* (also rules out a wrapping product) */
            fclose(f);
            return nullptr;
        }
        sz[i] = (size_t)(c[i] * elem[i]);
        sum += pad8(sz[i]);
    }
    if (sum != file_payload) {
        fclose(f);
        return nullptr;
    }
    size_t total = pad8(sizeof(rt_scene_desc)) + (size_t)sum;
    char *blk = (char *)malloc(total);
    if (!blk) {
        fclose(f);
        return nullptr;
    }
    size_t payload = total - pad8(sizeof(rt_scene_desc));
    if (fread(blk + pad8(sizeof(rt_scene_desc)), 1, payload, f) != payload) {
        free(blk);
        fclose(f);
        return nullptr;
    }
    fclose(f);
    rt_scene_desc *d = (rt_scene_desc *)blk;
    memset(d, 0, sizeof(*d));
    char *p = blk + pad8(sizeof(rt_scene_desc));
    const void *ptr[10];
    for (int i = 0; i < 10; i++) {
        ptr[i] = sz[i] ? p : nullptr;
        p += pad8(sz[i]);
    }
    d->abi_version = RT_AMD_ABI_VERSION;
    d->n_prims = (int32_t)c[0];
    d->prims = (const rt_prim *)ptr[0];
    d->n_groups = (int32_t)c[1];
    d->groups = (const rt_group *)ptr[1];
    d->n_materials = (int32_t)c[2];
    d->materials = (const rt_material *)ptr[2];
    d->n_textures = (int32_t)c[3];
    d->textures = (const rt_texture *)ptr[3];
    d->n_perlins = (int32_t)c[4];
    d->perlins = (const rt_perlin *)ptr[4];
    d->n_images = (int32_t)c[5];
    d->images = (const rt_image *)ptr[5];
    d->n_texel_bytes = c[6];
    d->texels = (const uint8_t *)ptr[6];
    d->n_xform_ops = (int32_t)c[7];
    d->xform_ops = (const rt_xform_op *)ptr[7];
    d->n_chains = (int32_t)c[8];
    d->chains = (const rt_xform_chain *)ptr[8];
    d->n_instances = (int32_t)c[9];
    d->instances = (const rt_instance *)ptr[9];
    return d;
}

} // namespace rtio
#endif /* RT_SCENE_IO_H */
