// host-only sanitizer run: flatten + wide tree + traversal logic over the golden scenes
#include "../cpu_sim/kernel_logic.cpp"
#include "rt_scene_io.h"
#include <cstdio>
#include <random>
int main(int argc, char **argv) {
    std::mt19937 g(7);
    std::uniform_real_distribution<float> U(-15, 15);
    for (int a = 1; a < argc; a++) {
        rt_scene_desc *d = rtio::load(argv[a]);
        if (!d) { printf("cannot load %s\n", argv[a]); return 1; }
        for (uint32_t flags : {0u, (uint32_t)RT_SCENE_FORCE_BVH, (uint32_t)RT_SCENE_FORCE_LINEAR}) {
            if (flags == RT_SCENE_FORCE_LINEAR && d->n_prims > 64) continue;
            void *s = sim_scene_create(d, flags);
            if (!s) { printf("flatten failed %s\n", argv[a]); return 1; }
            const int n = 20000;
            std::vector<rt_ray> rays(n);
            std::vector<rt_hit> hits(n);
            for (auto &r : rays) {
                for (int k = 0; k < 3; k++) r.orig[k] = U(g), r.dir[k] = U(g) - r.orig[k];
                r.tm = 0.5f;
            }
            for (int mode = 0; mode < 3; mode++) {
                sim_set_stack_mode(mode);
                unsigned long long bt = 0, pt = 0;
                sim_trace_batch(s, rays.data(), n, 0.001f, INFINITY, hits.data(), &bt, &pt);
            }
            sim_set_stack_mode(0);
            long nh = 0;
            for (auto &h : hits) nh += h.prim >= 0;
            printf("%s flags %u: nodes %d depth %d hits %ld\n", argv[a], flags, sim_scene_nodes(s), sim_scene_depth(s), nh);
            sim_scene_destroy(s);
        }
        free(d);
    }
    return 0;
}
