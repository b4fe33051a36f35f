// host-only sanitizer run over scene files that may be damaged: what the library does with a
// descriptor before anything reaches the GPU (the file reader, the descriptor checks, flattening
// and the tree builders), then the kernels' traversal and shading logic on the result.
// A file ends as "refused" or "ok"; a sanitizer report, a crash or a hang is a failure.
#include "../cpu_sim/kernel_logic.cpp"
#include "rt_scene_io.h"
#include "rt_validate.h"
#include <cstdio>
#include <random>
int main(int argc, char **argv) {
    std::mt19937 g(11);
    std::uniform_real_distribution<float> U(-15, 15);
    int n_ok = 0, n_refused = 0;
    for (int a = 1; a < argc; a++) {
        rt_scene_desc *d = rtio::load(argv[a]);
        if (!d) {
            n_refused++;
            continue;
        }
        std::string err;
        if (rtf::validate_desc(d, err) != RT_OK) {
            n_refused++;
            free(d);
            continue;
        }
        bool built = false;
        for (uint32_t flags : {0u, (uint32_t)RT_SCENE_FORCE_BVH}) {
            void *s = sim_scene_create(d, flags);
            if (!s) continue; // flattening refused it (a message, no crash)
            built = true;
            const int n = 400;
            std::vector<rt_ray> rays(n);
            std::vector<rt_hit> hits(n);
            for (auto &r : rays) {
                for (int k = 0; k < 3; k++) r.orig[k] = U(g), r.dir[k] = U(g) - r.orig[k];
                r.tm = 0.5f;
            }
            for (int mode = 0; mode < 3; mode++) {
                sim_set_stack_mode(mode);
                sim_trace_batch(s, rays.data(), n, 0.001f, INFINITY, hits.data(), nullptr, nullptr);
            }
            sim_set_stack_mode(0);
            rt_camera_desc cam;
            memset(&cam, 0, sizeof cam);
            cam.aspect_ratio = 1.0f;
            cam.image_width = 12;
            cam.samples_per_pixel = 2;
            cam.max_depth = 6;
            cam.background[0] = cam.background[1] = cam.background[2] = 0.5f;
            cam.vfov = 60.0f;
            cam.lookfrom[0] = 13, cam.lookfrom[1] = 3, cam.lookfrom[2] = 9;
            cam.vup[1] = 1.0f;
            cam.focus_dist = 10.0f;
            std::vector<float> accum(12 * 12 * 3, 0.0f);
            unsigned long long rays_out = 0;
            sim_render(s, &cam, 0, 2, 1u, accum.data(), &rays_out);
            sim_scene_destroy(s);
        }
        built ? n_ok++ : n_refused++;
        free(d);
    }
    printf("ok %d refused %d\n", n_ok, n_refused);
    return 0;
}
