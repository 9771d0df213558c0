// Sanitizer driver for the host side above the C ABI (raytracing_amd/csrc/host_scene.cpp): builds every scene of the
// factory with several seeds and aspect ratios, walks every array an rt_scene points into, exercises the camera
// builder's error paths.  Built with -fsanitize=address,undefined by `make -C raytracing_amd/csrc asan` and run by
// tests/test_sanitizers.py; any out-of-bounds access, leak of a destroyed world or undefined arithmetic aborts it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <initializer_list>

#include "../include/rtx_hip.h"

static double walk(const rt_scene* s) {
    double acc = 0;
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        acc += s->c0x[i] + s->c0y[i] + s->c0z[i] + s->dcx[i] + s->dcy[i] + s->dcz[i] + s->radius[i];
        if (s->material_id[i] >= s->n_materials) { std::printf("material id out of range\n"); return NAN; }
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rt_material& m = s->materials[i];
        acc += m.albedo[0] + m.fuzz + m.ior + m.type;
        if (m.type == RT_MAT_LAMBERTIAN && (m.texture_id < 0 || (uint32_t)m.texture_id >= s->n_textures)) {
            std::printf("texture id out of range\n");
            return NAN;
        }
    }
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const rt_texture& t = s->textures[i];
        acc += t.freq + t.color_a[2] + t.color_b[2] + t.layers;
        if (t.kind >= RT_TEX_VALUE_NOISE && (t.table_id < 0 || (uint32_t)t.table_id >= s->n_noise_tables)) {
            std::printf("table id out of range\n");
            return NAN;
        }
    }
    for (uint32_t i = 0; i < s->n_noise_tables; ++i)
        for (int k = 0; k < 512; ++k) acc += s->noise_tables[i].perms[k] + s->noise_tables[i].randoms[k & 255];
    return acc;
}

int main() {
    int n_worlds = 0;
    const char* const* names = rth_scene_names();
    for (int i = 0; names[i]; ++i) {
        for (uint64_t seed = 0; seed < 3; ++seed) {
            for (double aspect : {16.0 / 9.0, 1.0, 0.5}) {
                rth_world* w = nullptr;
                if (rth_world_create(names[i], 1, seed, aspect, &w) != RT_OK) {
                    std::printf("create %s failed: %s\n", names[i], rth_last_error());
                    return 1;
                }
                const double a = walk(rth_world_scene(w));
                const rt_camera* c = rth_world_camera(w);
                if (!std::isfinite(a) || !std::isfinite((double)c->origin[0] + c->horiz[0] + c->vert[1] + c->lower_left[2])) {
                    std::printf("%s: non-finite scene data\n", names[i]);
                    return 1;
                }
                (void)rth_world_render_seed(w);
                (void)rth_world_bvh_first(w);
                rth_world_destroy(w);
                ++n_worlds;
            }
        }
    }
    rth_world* w = nullptr;  // entropy seed, error paths
    if (rth_world_create("Random", 0, 0, 1.5, &w) != RT_OK) return 1;
    rth_world_destroy(w);
    if (rth_world_create("random", 1, 1, 1.5, &w) == RT_OK) { std::printf("wrong-case name accepted\n"); return 1; }
    if (rth_world_create(nullptr, 1, 1, 1.5, &w) == RT_OK) return 1;
    rth_world_destroy(nullptr);
    const double o[3] = {13, 2, 3}, l[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    rt_camera cam;
    if (rth_camera_build(o, l, up, 20, 1.5, 0.1, 10, 1, 0, 1, &cam) != RT_OK) return 1;
    if (rth_camera_build(o, l, nullptr, 20, 1.5, 0, 0, 0, 0, 0, &cam) != RT_OK) return 1;
    if (rth_camera_build(o, o, up, 20, 1.5, 0, 0, 0, 0, 0, &cam) == RT_OK) { std::printf("degenerate camera accepted\n"); return 1; }
    if (rth_camera_build(o, l, up, -1, 1.5, 0, 10, 0, 0, 0, &cam) == RT_OK) { std::printf("negative fov accepted\n"); return 1; }
    std::printf("host_scene sanitizer driver ok: %d worlds\n", n_worlds);
    return 0;
}
