// rt_stage.h — device-only: staging of the scene's hot arrays into a workgroup's LDS.
#pragma once

#include <hip/hip_runtime.h>

#include "rt_core.h"

namespace rtc {

// Copies n 16-byte words from HBM to LDS with the whole workgroup.
__device__ __forceinline__ void stage_to_lds(f4 *dst, const f4 *src, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// Copies n_nodes nodes (RT_NODE_F4 words apart) to LDS, RT_NODE_LDS_F4 words apart.
__device__ __forceinline__ void stage_nodes_to_lds(f4 *dst, const f4 *src, int n_nodes) {
    for (int i = threadIdx.x; i < n_nodes * RT_NODE_F4; i += blockDim.x)
        dst[(i >> 3) * RT_NODE_LDS_F4 + (i & 7)] = src[i];
}

// Stages slots, nodes, the per-slot records, the texture table and the Perlin tables into LDS
// (the layout matches choose_geom() in rt_host.cpp); returns the first free 16-byte word.
__device__ __forceinline__ f4 *stage_scene(const DevScene &sc, f4 *smem, SceneView &v) {
    int n_slot_words = sc.n_slots * sc.slot_f4;
    int n_node_words = sc.n_nodes * RT_NODE_LDS_F4;
    int n_tex_words = (sc.n_textures * (int)sizeof(rt_texture) + 15) / 16;
    f4 *p = smem;
    stage_to_lds(p, sc.slots, n_slot_words);
    v.slots = p;
    p += n_slot_words;
    stage_nodes_to_lds(p, sc.nodes, sc.n_nodes);
    v.nodes = p;
    p += n_node_words;
    // (unconditionally: the host only picks the LDS variants when all of this fits, and pointers that
    // are LDS on every path let the compiler use LDS instructions for the records)
#ifndef RT_REC_GLOBAL // (experiment builds: the records stay in global memory, which halves the LDS copy's size)
    stage_to_lds(p, reinterpret_cast<const f4 *>(sc.rec_i), sc.n_slots);
    v.rec_i = reinterpret_cast<const i4 *>(p);
    p += sc.n_slots;
    stage_to_lds(p, sc.rec_f, sc.n_slots);
    v.rec_f = p;
    p += sc.n_slots;
#endif
    // the device copy of the texture table is padded to whole 16-byte words
    stage_to_lds(p, reinterpret_cast<const f4 *>(sc.textures), n_tex_words);
    v.textures = reinterpret_cast<const rt_texture *>(p);
    p += n_tex_words;
    if (sc.lds_perlins) { // noise textures gather from these tables 50 times per evaluation
        const int n_perlin_words = sc.n_perlins * (int)(sizeof(DevPerlin) / sizeof(f4));
        stage_to_lds(p, reinterpret_cast<const f4 *>(sc.perlins), n_perlin_words);
        v.perlins = reinterpret_cast<const DevPerlin *>(p);
        p += n_perlin_words;
    }
    __syncthreads();
    return p;
}

// This lane's traversal stack: a column of the workgroup's [entry][lane] array that starts at
// `lds_words`, and (SPILL) a column of the same shape in global memory for the entries beyond.
// STRIDE: lanes per entry row, whatever the workgroup's size (0: the workgroup's size).
template <typename T, bool SPILL, int STRIDE>
__device__ __forceinline__ TravStack<T, STRIDE> lane_stack(f4 *lds_words, const DevScene &sc, int lds_entries, int *spill) {
    TravStack<T, STRIDE> st;
    st.lds = reinterpret_cast<T *>(lds_words) + threadIdx.x;
    st.rt_stride = (int)blockDim.x;
    st.lds_end = lds_entries * st.stride();
    st.spill = nullptr;
    if constexpr (SPILL)
        st.spill = spill + (size_t)blockIdx.x * (size_t)(sc.stack_cap - lds_entries) * st.stride() + threadIdx.x;
    return st;
}

} // namespace rtc
