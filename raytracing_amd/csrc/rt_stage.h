// rt_stage.h — device-only: staging of the scene's hot arrays into a workgroup's LDS.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_core.h"

namespace rtc {

// Copies n 16-byte words from HBM to LDS with the whole workgroup.
__device__ __forceinline__ void stage_to_lds(f4 *dst, const f4 *src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// Stages slots, nodes, the per-slot records and the texture table into LDS (the layout
// matches scene_lds_bytes() in rt_host.cpp); returns the first free 16-byte word.
__device__ __forceinline__ f4 *stage_scene(const DevScene &sc, f4 *smem, SceneView &v) {
    int n_slot_words = sc.n_slots * sc.slot_f4;
    int n_node_words = sc.n_nodes * 4;
    int n_tex_words = (sc.n_textures * (int)sizeof(rt_texture) + 15) / 16;
    f4 *p = smem;
    stage_to_lds(p, sc.slots, n_slot_words);
    v.slots = p;
    p += n_slot_words;
    stage_to_lds(p, sc.nodes, n_node_words);
    v.nodes = p;
    p += n_node_words;
    if (sc.lds_records) {
        stage_to_lds(p, reinterpret_cast<const f4 *>(sc.rec_i), sc.n_slots);
        v.rec_i = reinterpret_cast<const i4 *>(p);
        p += sc.n_slots;
        stage_to_lds(p, sc.rec_f, sc.n_slots);
        v.rec_f = p;
        p += sc.n_slots;
        // the device copy of the texture table is padded to whole 16-byte words
        stage_to_lds(p, reinterpret_cast<const f4 *>(sc.textures), n_tex_words);
        v.textures = reinterpret_cast<const rt_texture *>(p);
        p += n_tex_words;
    }
    __syncthreads();
    return p;
}

} // namespace rtc
