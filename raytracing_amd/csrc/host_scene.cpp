// host_scene.cpp — host side ABOVE the C ABI: the reference's scene/camera API restated in
// C++ (the reference is compiled Rust; no Rust toolchain in this image).  It produces the
// flattened rt_scene / rt_camera that the HIP library consumes.  In a real deployment the
// Rust crate keeps config.rs / shape.rs / material.rs / screen.rs and flattens its own
// objects; this file is the tested stand-in for that caller.
//
// Follows: config.rs:85-404 (Scene enum, camera(), world()), screen.rs:112-147
// (CameraBuilder::build), screen.rs:150-198 (verify), material.rs:278-307 / :398-424 (noise
// tables), main.rs:19-26,60 (seeding order).
//
// RNG: the reference draws scene parameters from rand 0.7.3 `SmallRng` (= rand_pcg 0.2.1
// `Pcg64Mcg` on 64-bit targets).  That crate's source is not in the image; the generator and
// the Standard / Uniform conversions below are restated from their published algorithms and
// cannot be checked against the Rust binary here ("parity unpinned" for scene layouts).
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/rtx_hip.h"

namespace {

thread_local std::string g_err;

struct SmallRng {
    unsigned __int128 s;
    explicit SmallRng(uint64_t seed) {
        // rand_core 0.5.1 SeedableRng::seed_from_u64: PCG32 stream expands the u64 to 16 bytes.
        uint32_t w[4];
        for (uint32_t& x : w) {
            seed = seed * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xs = (uint32_t)(((seed >> 18) ^ seed) >> 27), rot = (uint32_t)(seed >> 59);
            x = (xs >> rot) | (xs << ((32u - rot) & 31u));
        }
        uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32), hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        s = ((((unsigned __int128)hi) << 64) | lo) | 1u;  // Mcg128Xsl64::new forces the state odd
    }
    uint64_t u64() {
        const unsigned __int128 M = (((unsigned __int128)0x2360ED051FC65DA4ull) << 64) | 0x4385DF649FCCF645ull;
        s *= M;
        uint64_t x = (uint64_t)(s >> 64) ^ (uint64_t)s;
        unsigned r = (unsigned)(s >> 122);
        return (x >> r) | (x << ((64u - r) & 63u));
    }
    double f64() { return (double)(u64() >> 11) * 0x1.0p-53; }  // Standard: 53 random bits
    double range(double lo, double hi) {                         // UniformFloat: [1,2) - 1, * scale + lo
        uint64_t b = (u64() >> 12) | 0x3FF0000000000000ull;
        double v;
        std::memcpy(&v, &b, 8);
        return (v - 1.0) * (hi - lo) + lo;
    }
    uint32_t below256() { return (uint32_t)(u64() >> 56); }  // Uniform::new(0usize, 256): hi word of v*256
    void axis() { (void)(uint32_t)u64(); }                   // Uniform::new(0u8,3): one u32 draw (vec3.rs:273)
};

struct V3 {
    double x, y, z;
};
V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
double len(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
bool unit(V3 a, V3* out) {  // Vec3::checked_normalized (vec3.rs:47-51)
    double n = len(a);
    if (n == 0.) return false;
    *out = {a.x / n, a.y / n, a.z / n};
    return true;
}

}  // namespace

struct rth_world {
    std::vector<float> c0x, c0y, c0z, dcx, dcy, dcz, rad;
    std::vector<uint32_t> mat_id;
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<rt_noise_table> tables;
    rt_scene scene{};
    rt_camera cam{};
    uint64_t render_seed = 0;
    int bvh_first = -1;

    int solid(double r, double g, double b) {
        rt_texture t{};
        t.kind = RT_TEX_SOLID;
        t.color_a[0] = (float)r; t.color_a[1] = (float)g; t.color_a[2] = (float)b;
        texs.push_back(t);
        return (int)texs.size() - 1;
    }
    int checker(double f, const double odd[3], const double even[3]) {  // Checkered::color (material.rs:147)
        rt_texture t{};
        t.kind = RT_TEX_CHECKER;
        t.freq = (float)f;
        for (int i = 0; i < 3; ++i) { t.color_a[i] = (float)odd[i]; t.color_b[i] = (float)even[i]; }
        texs.push_back(t);
        return (int)texs.size() - 1;
    }
    int lambert(int tex) {
        rt_material m{};
        m.type = RT_MAT_LAMBERTIAN;
        m.texture_id = tex;
        mats.push_back(m);
        return (int)mats.size() - 1;
    }
    int metal(double r, double g, double b, double fuzz) {
        rt_material m{};
        m.type = RT_MAT_METAL;
        m.albedo[0] = (float)r; m.albedo[1] = (float)g; m.albedo[2] = (float)b;
        m.fuzz = (float)std::fmin(fuzz, 1.0);  // Metal::new clamps (material.rs:52)
        m.texture_id = -1;
        mats.push_back(m);
        return (int)mats.size() - 1;
    }
    int glass(double ior) {
        rt_material m{};
        m.type = RT_MAT_DIELECTRIC;
        m.ior = (float)ior;
        m.texture_id = -1;
        mats.push_back(m);
        return (int)mats.size() - 1;
    }
    void sphere(V3 c, double r, int m) { moving(c, c, r, m); }
    void moving(V3 a, V3 b, double r, int m) {  // MovingSphere::new stores delta_c = c1 - c0 (shape.rs:82-89)
        c0x.push_back((float)a.x); c0y.push_back((float)a.y); c0z.push_back((float)a.z);
        dcx.push_back((float)(b.x - a.x)); dcy.push_back((float)(b.y - a.y)); dcz.push_back((float)(b.z - a.z));
        rad.push_back((float)r);
        mat_id.push_back((uint32_t)m);
    }
    // ValueNoise::new / PerlinNoise::new table construction (material.rs:284-299, :404-417).
    int table(bool with_randoms, int has_seed, uint64_t seed) {
        SmallRng g(has_seed ? seed : std::random_device{}() * 0x9E3779B97F4A7C15ull + (uint64_t)std::random_device{}());
        rt_noise_table t{};
        for (int i = 0; i < 256; ++i) {
            if (with_randoms) t.randoms[i] = (float)g.f64();
            t.perms[i] = (uint16_t)i;
        }
        for (int i = 0; i < 256; ++i) {
            uint32_t j = g.below256();
            std::swap(t.perms[i], t.perms[j]);
            t.perms[i + 256] = t.perms[i];  // copied mid-shuffle: second half goes stale
        }
        tables.push_back(t);
        return (int)tables.size() - 1;
    }
    int noise_tex(int kind, double freq, int tbl) {
        rt_texture t{};
        t.kind = kind;
        t.freq = (float)freq;
        t.op = RT_NOISE_PLAIN;
        t.table_id = tbl;
        texs.push_back(t);
        return (int)texs.size() - 1;
    }
    void seal() {
        scene.n_spheres = (uint32_t)rad.size();
        scene.c0x = c0x.data(); scene.c0y = c0y.data(); scene.c0z = c0z.data();
        scene.dcx = dcx.data(); scene.dcy = dcy.data(); scene.dcz = dcz.data();
        scene.radius = rad.data();
        scene.material_id = mat_id.data();
        scene.n_materials = (uint32_t)mats.size();
        scene.materials = mats.data();
        scene.n_textures = (uint32_t)texs.size();
        scene.textures = texs.data();
        scene.n_noise_tables = (uint32_t)tables.size();
        scene.noise_tables = tables.data();
    }
};

namespace {

const char* const kNames[] = {"Random",         "TwoSpheres",      "Balls",         "BirdsEyeView",     "ValueNoise1",
                              "ValueNoise2",    "ValueNoisePink",  "ValueTurbulence", "ValueMarbled",   "PerlinNoise1",
                              "PerlinNoise2",   "PerlinNoisePink", "PerlinTurbulence", "PerlinMarbled", "FinalStatic",
                              nullptr};

// fBm / turbulence / marbled builder calls (material.rs:184-263) recorded as data.
void set_fractal(rt_texture& t, int op, double lac, double gain, int layers) {
    t.op = op;
    t.lacunarity = (float)lac;
    t.gain = (float)gain;
    t.layers = layers;
}
void set_marbled(rt_texture& t, double sin_freq, double phase, double noise_gain) {
    t.marbled = 1;
    t.sin_freq = (float)sin_freq;
    t.phase = (float)phase;
    t.marble_gain = (float)((double)t.freq * noise_gain);  // material.rs:251
}

int bvh_axis_draws(size_t n) {  // BVH::inner_list draws one Axis per node with > 2 children (hit.rs:245-258)
    if (n <= 2) return 0;
    return 1 + bvh_axis_draws(n / 2) + bvh_axis_draws(n - n / 2);
}

struct CamSpec {
    V3 origin, look_at, up{0, 1, 0};
    double vfov = 60., aperture = 0., focus = -1., sh0 = 0., sh1 = 0.;
    int shutter = 0;
};

int build_camera(const CamSpec& s, double aspect, rt_camera* out) {
    V3 up, w, u;
    if (!unit(s.up, &up)) { g_err = "Camera's view_up vector has length 0"; return RT_ERR_BAD_ARG; }
    if (!unit(s.origin - s.look_at, &w)) { g_err = "Camera's origin and look_at vectors are the same"; return RT_ERR_BAD_ARG; }
    if (!unit(cross(up, w), &u)) { g_err = "Camera's look_at and view_up vectors are parallel"; return RT_ERR_BAD_ARG; }
    if (!(s.aperture >= 0.)) { g_err = "Camera's aperture is less than 0"; return RT_ERR_BAD_ARG; }
    if (!(s.vfov > 0.)) { g_err = "Camera's fov is less than or equal to 0"; return RT_ERR_BAD_ARG; }
    if (!(aspect > 0.)) { g_err = "Camera's aspect ratio is less than or equal to 0"; return RT_ERR_BAD_ARG; }
    double focus = s.focus > 0. ? s.focus : len(s.origin - s.look_at);  // screen.rs:119
    double theta = (s.vfov * (M_PI / 180.0)) / 2.;
    double hh = focus * std::tan(theta), hw = aspect * hh;
    V3 v = cross(w, u);
    V3 ll = s.origin - u * hw - v * hh - w * focus;  // screen.rs:132
    V3 horiz = u * 2. * hw, vert = v * 2. * hh;
    auto put = [](float* d, V3 a) { d[0] = (float)a.x; d[1] = (float)a.y; d[2] = (float)a.z; };
    std::memset(out, 0, sizeof(*out));
    put(out->origin, s.origin); put(out->lower_left, ll); put(out->horiz, horiz); put(out->vert, vert);
    put(out->u, u); put(out->v, v);
    out->lens_radius = (float)(s.aperture / 2.);
    out->has_shutter = s.shutter;
    out->shutter_open = (float)s.sh0;
    out->shutter_close = (float)s.sh1;
    return RT_OK;
}

void random_world(rth_world& w, SmallRng& rng, bool as_reference) {  // config.rs:150-200
    const double odd[3] = {0.2, 0.3, 0.1}, even[3] = {0.9, 0.9, 0.9};
    int ground_tex = as_reference ? w.checker(10., odd, even) : w.solid(0.5, 0.5, 0.5);
    w.sphere({0., -1000., 0.}, 1000., w.lambert(ground_tex));
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double x = 0.9 * rng.f64(), z = 0.9 * rng.f64();
            V3 c{a + x, 0.2, b + z};
            if (len(c - V3{4., 0.2, 0.}) <= 0.9) continue;
            double pick = rng.f64();
            if (pick < 0.8) {
                double l[3], r[3];
                for (double& q : l) q = rng.f64();  // Color::rand(rng) * Color::rand(rng), left first
                for (double& q : r) q = rng.f64();
                double dy = rng.range(0., 0.5);
                int m = w.lambert(w.solid(l[0] * r[0], l[1] * r[1], l[2] * r[2]));
                if (as_reference) w.moving(c, c + V3{0., dy, 0.}, 0.2, m);
                else w.sphere(c, 0.2, m);
            } else if (pick < 0.95) {
                double al[3];
                for (double& q : al) q = rng.range(0.5, 1.);
                double fuzz = rng.range(0., 0.5);
                w.sphere(c, 0.2, w.metal(al[0], al[1], al[2], fuzz));
            } else {
                w.sphere(c, 0.2, w.glass(1.5));
            }
        }
    w.sphere({0., 1., 0.}, 1., w.glass(1.5));
    w.sphere({-4., 1., 0.}, 1., w.lambert(w.solid(0.4, 0.2, 0.1)));
    w.sphere({4., 1., 0.}, 1., w.metal(0.7, 0.6, 0.5, 0.0));
    w.bvh_first = 1;  // everything but the ground goes through into_bvh (config.rs:197)
    for (int i = bvh_axis_draws(w.rad.size() - 1); i > 0; --i) rng.axis();
}

}  // namespace

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }
const char* const* rth_scene_names(void) { return kNames; }

int rth_camera_build(const double origin[3], const double look_at[3], const double view_up[3],
                     double vfov_degrees, double aspect_ratio, double aperture, double focus_dist,
                     int has_shutter, double shutter_open, double shutter_close, rt_camera* out) {
    if (!origin) { g_err = "Camera's origin wasn't provided."; return RT_ERR_BAD_ARG; }
    if (!look_at) { g_err = "Camera's look_at wasn't provided."; return RT_ERR_BAD_ARG; }
    if (!out) { g_err = "out is NULL"; return RT_ERR_BAD_ARG; }
    if (focus_dist < 0. || (focus_dist == 0. && std::signbit(focus_dist))) {
        g_err = "Camera's focus distance is less than or equal to 0";
        return RT_ERR_BAD_ARG;
    }
    CamSpec s;
    s.origin = {origin[0], origin[1], origin[2]};
    s.look_at = {look_at[0], look_at[1], look_at[2]};
    if (view_up) s.up = {view_up[0], view_up[1], view_up[2]};
    s.vfov = vfov_degrees; s.aperture = aperture; s.focus = focus_dist;
    s.shutter = has_shutter; s.sh0 = shutter_open; s.sh1 = shutter_close;
    return build_camera(s, aspect_ratio, out);
}

int rth_world_create(const char* name, int has_seed, uint64_t seed, double aspect, rth_world** out) {
    if (!name || !out) { g_err = "NULL argument"; return RT_ERR_BAD_ARG; }
    int which = -1;
    for (int i = 0; kNames[i]; ++i)
        if (std::strcmp(kNames[i], name) == 0) which = i;  // exact case only (config.rs:407-426)
    if (which < 0) {
        g_err = std::string("\"") + name + "\" isn't a Scene.";
        return RT_ERR_BAD_ARG;
    }
    uint64_t main_seed = has_seed ? seed
                                  : ((uint64_t)std::random_device{}() << 32 | std::random_device{}()) ^
                                        (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    SmallRng rng(main_seed);  // main.rs:19-22
    auto* w = new rth_world();
    CamSpec cs;
    cs.origin = {13., 2., 3.};
    cs.look_at = {0., 0., 0.};
    const std::string n = name;
    const double odd[3] = {0.2, 0.3, 0.1}, even[3] = {0.9, 0.9, 0.9};
    if (n == "Random" || n == "FinalStatic") {
        cs.vfov = 20.; cs.aperture = 0.1; cs.focus = 10.;
        if (n == "Random") { cs.shutter = 1; cs.sh0 = 0.; cs.sh1 = 1.; }
        random_world(*w, rng, n == "Random");
    } else if (n == "TwoSpheres") {
        cs.vfov = 20.; cs.focus = 10.;
        int t = w->checker(10., odd, even);
        w->sphere({0., -10., 0.}, 10., w->lambert(t));
        w->sphere({0., 10., 0.}, 10., w->lambert(t));
    } else if (n == "Balls") {
        cs.origin = {-2., 1.5, 1.}; cs.look_at = {-0.2, 0., -1.2}; cs.vfov = 40.;
        w->sphere({0., -100.5, -1.}, 100., w->lambert(w->solid(0.8, 0.8, 0.)));
        w->sphere({0., 0., -1.}, 0.5, w->glass(1.5));
        w->sphere({1.5, 0., -1.}, 0.5, w->metal(0.8, 0.6, 0.2, 0.));
        w->sphere({-1.05, 0., -1.}, 0.5, w->lambert(w->solid(0.1, 0.2, 0.5)));
        w->sphere({1.5, 0., -2.5}, 0.5, w->metal(0.8, 0.6, 0.2, 0.));
        w->sphere({-1.05, 0., -2.5}, 0.5, w->lambert(w->solid(0.1, 0.2, 0.5)));
    } else if (n == "BirdsEyeView") {
        cs.origin = {0., 20., 0.};
        double rads = 15. * (M_PI / 180.0) + M_PI / 2.;  // view_up_degrees(15, Axis::Y) (screen.rs:233-244)
        cs.up = {std::cos(rads), 0., -std::sin(rads)};
        const double o2[3] = {0.2, 0.3, 0.1}, e2[3] = {0.9, 0.9, 0.9};
        w->sphere({0., 0., 0.}, 10., w->lambert(w->checker(2.5, o2, e2)));
    } else {
        cs.vfov = 30.;
        bool value = n.rfind("Value", 0) == 0;
        int kind = value ? RT_TEX_VALUE_NOISE : RT_TEX_PERLIN_NOISE;
        int tbl = w->table(value, has_seed, seed);  // Noise::new(GLOBAL().seed, freq)
        std::string tail = n.substr(value ? 5 : 6);
        double freq = 4.;
        if (tail == "NoisePink") freq = 2.;
        else if (tail == "Turbulence") freq = 5.;
        else if (tail == "Marbled") freq = value ? 3. : 2.;
        int tex = w->noise_tex(kind, freq, tbl);
        rt_texture& t = w->texs[tex];
        if (tail == "NoisePink") set_fractal(t, RT_NOISE_FBM, 2., 0.5, 5);
        else if (tail == "Turbulence") {
            if (value) set_fractal(t, RT_NOISE_TURBULENCE, 1.8, 0.35, 5);
            else set_fractal(t, RT_NOISE_TURBULENCE, 2., 0.5, 7);
        } else if (tail == "Marbled") {
            if (value) { set_fractal(t, RT_NOISE_FBM, 2., 0.5, 2); set_marbled(t, 2., 2.5, 0.5); }
            else { set_fractal(t, RT_NOISE_FBM, 2., 0.5, 5); set_marbled(t, 2., 0., 2.0); }
        }
        w->sphere({0., -1000., 0.}, 1000., w->lambert(tex));
        if (tail == "Noise1") {
            w->sphere({0., 2., 0.}, 2., w->metal(0.8, 0.8, 0.8, 0.05));
            w->sphere({5., 2., -3.}, 2., w->lambert(w->solid(0.1, 0.2, 0.5)));
        } else {
            w->sphere({0., 2., 0.}, 2., w->lambert(tex));
        }
    }
    int rc = build_camera(cs, aspect, &w->cam);
    if (rc != RT_OK) { delete w; return rc; }
    w->render_seed = rng.u64();  // main.rs:60
    w->seal();
    *out = w;
    return RT_OK;
}

void rth_world_destroy(rth_world* w) { delete w; }
const rt_scene* rth_world_scene(const rth_world* w) { return w ? &w->scene : nullptr; }
const rt_camera* rth_world_camera(const rth_world* w) { return w ? &w->cam : nullptr; }
uint64_t rth_world_render_seed(const rth_world* w) { return w ? w->render_seed : 0; }
int rth_world_bvh_first(const rth_world* w) { return w ? w->bvh_first : -1; }

}  // extern "C"
