// rtx_cli.cpp — the reference binary's command line on top of the C ABI.
//
// Mirrors `Config` (config.rs:25-69): -w/--width 1024, -h/--height 576, -n/--no-aa, -s/--samples 100
// (NonZeroU16), -m/--max-depth 100 (NonZeroU32), -r/--rng <seed>, positional scene (default Random, exact-case
// names, config.rs:71-79).  Flow follows main.rs:15-106: seed -> Scene::create -> render seed -> timed render ->
// "Rending time elapsed" on stderr -> Screen::encode.  The reference opens a minifb window (main.rs:97-105);
// there is no display on a GPU box, so the encoded 0RGB buffer is written as a binary PPM instead (-o, extension).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtx_hip.h"

static int usage(const char* argv0, const char* msg) {
    if (msg) std::fprintf(stderr, "error: %s\n", msg);
    std::fprintf(stderr,
                 "USAGE: %s [FLAGS] [OPTIONS] [scene]\n"
                 "  -w, --width <width>          Width of render [default: 1024]\n"
                 "  -h, --height <height>        Height of render [default: 576]\n"
                 "  -n, --no-aa                  Disable antialiasing\n"
                 "  -s, --samples <samples>      Number of samples per pixel [default: 100]\n"
                 "  -m, --max-depth <max-depth>  Maximum ray bounce depth [default: 100]\n"
                 "  -r, --rng <seed>             Use a specific seed for the rng.\n"
                 "  -o, --output <file.ppm>      Write the encoded image (extension; default render.ppm)\n"
                 "  -d, --device <ordinal>       HIP device (extension; default 0)\n"
                 "  -g, --gpus <n>               render on devices 0..n-1, samples split across them, one RCCL\n"
                 "                               reduce (extension; default: one device, no RCCL needed)\n"
                 "  <scene>                      The scene to render [default: Random]. Possible values:\n",
                 argv0);
    for (const char* const* n = rth_scene_names(); *n; ++n) std::fprintf(stderr, "      %s\n", *n);
    return 2;
}

static bool parse_u64(const char* s, uint64_t* out) {
    if (!s || !*s) return false;
    char* end = nullptr;
    unsigned long long v = std::strtoull(s, &end, 10);
    if (*end) return false;
    *out = v;
    return true;
}

int main(int argc, char** argv) {
    uint64_t width = 1024, height = 576, samples = 100, max_depth = 100, seed = 0, device = 0, gpus = 0;
    int has_seed = 0, antialias = 1;
    std::string scene = "Random", output = "render.ppm";
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](uint64_t* dst) { return i + 1 < argc && parse_u64(argv[++i], dst); };
        if (a == "-w" || a == "--width") { if (!val(&width)) return usage(argv[0], "bad --width"); }
        else if (a == "-h" || a == "--height") { if (!val(&height)) return usage(argv[0], "bad --height"); }
        else if (a == "-s" || a == "--samples") { if (!val(&samples)) return usage(argv[0], "bad --samples"); }
        else if (a == "-m" || a == "--max-depth") { if (!val(&max_depth)) return usage(argv[0], "bad --max-depth"); }
        else if (a == "-r" || a == "--rng") { if (!val(&seed)) return usage(argv[0], "bad --rng"); has_seed = 1; }
        else if (a == "-d" || a == "--device") { if (!val(&device)) return usage(argv[0], "bad --device"); }
        else if (a == "-g" || a == "--gpus") { if (!val(&gpus) || gpus == 0 || gpus > RT_MAX_DEVICES) return usage(argv[0], "bad --gpus"); }
        else if (a == "-n" || a == "--no-aa") antialias = 0;
        else if (a == "-o" || a == "--output") { if (i + 1 >= argc) return usage(argv[0], "bad --output"); output = argv[++i]; }
        else if (a == "--help") return usage(argv[0], nullptr);
        else if (!a.empty() && a[0] == '-') return usage(argv[0], ("unknown flag " + a).c_str());
        else scene = a;
    }
    // NonZeroUsize / NonZeroU16 / NonZeroU32 (config.rs:29,33,52,56)
    if (width == 0 || height == 0) return usage(argv[0], "width and height must be non-zero");
    if (samples == 0 || samples > 65535) return usage(argv[0], "samples must be in 1..=65535 (NonZeroU16)");
    if (max_depth == 0 || max_depth > 0xFFFFFFFFull) return usage(argv[0], "max-depth must be a non-zero u32");

    // everything below is released on every return path
    struct Handles {
        rth_world* world = nullptr;
        rt_ctx* ctx = nullptr;
        rt_multi* multi = nullptr;
        ~Handles() { rt_multi_destroy(multi); rt_destroy(ctx); rth_world_destroy(world); }
    } h;
    if (rth_world_create(scene.c_str(), has_seed, seed, (double)width / (double)height, &h.world) != RT_OK) {
        std::fprintf(stderr, "%s\n", rth_last_error());  // parse_scene's message (config.rs:71-79)
        return 2;
    }
    rth_world* world = h.world;
    if (rt_create((int)device, &h.ctx) != RT_OK) {  // (also used for the encode step of a multi-device render)
        std::fprintf(stderr, "rt_create: %s\n", rt_last_error(nullptr));
        return 1;
    }
    rt_ctx* ctx = h.ctx;
    if (gpus) {
        std::vector<int> ord(gpus);
        for (uint64_t i = 0; i < gpus; ++i) ord[i] = (int)i;
        if (rt_multi_create(ord.data(), (int)gpus, &h.multi) != RT_OK) {
            std::fprintf(stderr, "rt_multi_create: %s\n", rt_multi_last_error(nullptr));
            return 1;
        }
        if (rt_multi_upload_scene(h.multi, rth_world_scene(world)) != RT_OK) {
            std::fprintf(stderr, "rt_multi_upload_scene: %s\n", rt_multi_last_error(h.multi));
            return 1;
        }
    } else if (rt_upload_scene(ctx, rth_world_scene(world)) != RT_OK) {
        std::fprintf(stderr, "rt_upload_scene: %s\n", rt_last_error(ctx));
        return 1;
    }
    rt_params p{};
    p.width = (uint32_t)width; p.height = (uint32_t)height; p.spp = (uint32_t)samples;
    p.max_depth = (uint32_t)max_depth; p.antialias = antialias; p.spp_offset = 0;
    p.seed = rth_world_render_seed(world);  // main.rs:60
    p.t_min = 0.001f;                       // main.rs:116
    std::vector<float> rgb((size_t)width * height * 3);
    rt_stats st{};
    const auto t0 = std::chrono::steady_clock::now();  // main.rs:62
    if (gpus) {
        rt_multi_stats ms{};
        if (rt_multi_render(h.multi, rth_world_camera(world), &p, rgb.data(), &ms) != RT_OK) {
            std::fprintf(stderr, "rt_multi_render: %s\n", rt_multi_last_error(h.multi));
            return 1;
        }
        st.samples = ms.samples; st.rays_traced = ms.rays_traced; st.kernel_ms = ms.kernel_ms;
        std::fprintf(stderr, "%u device(s), RCCL reduce %.3f ms\n", ms.n_devices, ms.reduce_ms);
    } else if (rt_render(ctx, rth_world_camera(world), &p, rgb.data(), &st) != RT_OK) {
        std::fprintf(stderr, "rt_render: %s\n", rt_last_error(ctx));
        return 1;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::fprintf(stderr, "\nRending time elapsed: %.2f seconds\n", sec);  // main.rs:95 (sic)
    std::fprintf(stderr, "%.1f Msamples/s, %.1f Mrays/s, %.3f rays/sample, kernel %.2f ms\n",
                 (double)st.samples / sec / 1e6, (double)st.rays_traced / sec / 1e6,
                 (double)st.rays_traced / (double)st.samples, st.kernel_ms);

    std::vector<uint32_t> enc((size_t)width * height);
    uint32_t bad = 0;
    if (rt_encode(ctx, rgb.data(), (uint32_t)(width * height), enc.data(), &bad) != RT_OK) {
        std::fprintf(stderr, "rt_encode: %s\n", rt_last_error(ctx));
        return 1;
    }
    if (bad) {  // Screen::encode panics here (screen.rs:31-34)
        std::fprintf(stderr, "Invalid color in %u pixels\n", bad);
        return 101;
    }
    FILE* f = std::fopen(output.c_str(), "wb");
    if (!f) { std::perror(output.c_str()); return 1; }
    std::fprintf(f, "P6\n%llu %llu\n255\n", (unsigned long long)width, (unsigned long long)height);
    std::vector<unsigned char> row(width * 3);
    for (uint64_t y = 0; y < height; ++y) {
        for (uint64_t x = 0; x < width; ++x) {
            const uint32_t v = enc[y * width + x];  // 0RGB
            row[3 * x] = (unsigned char)(v >> 16); row[3 * x + 1] = (unsigned char)(v >> 8); row[3 * x + 2] = (unsigned char)v;
        }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return 0;
}
