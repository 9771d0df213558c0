// oracle_capi.cpp — C entry points of the CPU oracle (ctypes-loadable).
// TEST INFRASTRUCTURE ONLY: see the header of rt_oracle.hpp.
#include "rt_oracle.hpp"

#include <cstdio>

using namespace orc;

extern "C" {

typedef struct orc_hit {
    int32_t hit, sphere_id, material_id, front_face;
    double t;
    double point[3];
    double normal[3];
    double margin;      // >= 1: every hit/miss/id decision is safely away from flipping in fp32 (see below)
    double margin_tie;  // the same WITHOUT term (a), the discriminant band: only the t_min boundary and near-ties
} orc_hit;

typedef struct orc_scatter {
    int32_t scattered, pad;
    double albedo[3], origin[3], dir[3], time;
} orc_scatter;

typedef struct orc_ray {
    double origin[3], dir[3], time;
} orc_ray;

// ---- the five Vec3 doctest operations (vec3.rs:27-33,41-46,64-70,87-93,98-104)
double orc_vec3_norm(const double v[3]) { return Vec3<double>(v[0], v[1], v[2]).norm(); }
void orc_vec3_normalized(const double v[3], double out[3]) {
    auto r = Vec3<double>::normalized({v[0], v[1], v[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_vec3_checked_normalized(const double v[3], double out[3]) {  // vec3.rs:47-51: Err on norm 0
    Vec3<double> a(v[0], v[1], v[2]);
    if (a.norm() == 0.) return -1;
    orc_vec3_normalized(v, out);
    return 0;
}
void orc_vec3_map_floor_times2(const double v[3], double out[3]) {  // the doctest's closure
    for (int i = 0; i < 3; ++i) out[i] = std::floor(v[i]) * 2.;
}
double orc_vec3_dot(const double a[3], const double b[3]) {
    return Vec3<double>(a[0], a[1], a[2]).dot({b[0], b[1], b[2]});
}
void orc_vec3_cross(const double a[3], const double b[3], double out[3]) {
    auto r = Vec3<double>(a[0], a[1], a[2]).cross({b[0], b[1], b[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_vec3_reflect(const double d[3], const double n[3], double out[3]) {
    auto r = Vec3<double>(d[0], d[1], d[2]).reflect({n[0], n[1], n[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_vec3_refract(const double d[3], const double n[3], double eta, double out[3]) {
    auto r = Vec3<double>(d[0], d[1], d[2]).refract({n[0], n[1], n[2]}, eta);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double orc_lerp(double t, double lo, double hi) { return lerp(t, lo, hi); }
double orc_smooth(double t) { return smooth(t); }
double orc_schlick(double c, double eta) { return schlick(c, eta); }

void orc_philox(uint64_t seed, const uint32_t* ctr4, uint32_t n, uint32_t* out4) {
    for (uint32_t i = 0; i < n; ++i)
        Philox::gen(seed, ctr4[4 * i], ctr4[4 * i + 1], ctr4[4 * i + 2], ctr4[4 * i + 3], out4 + 4 * i);
}
// rand 0.7.3 / rand_pcg restatement (unpinned): first n u64 of SmallRng::seed_from_u64(seed)
void orc_pcg64mcg(uint64_t seed, uint32_t n, uint64_t* out) {
    Pcg64Mcg g = Pcg64Mcg::seed_from_u64(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = g.next_u64();
}
void orc_map_unit_sphere(double u0, double u1, double out[3]) {
    auto v = map_unit_sphere<double>(u0, u1);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
void orc_map_unit_disk(double u0, double u1, double out[3]) {
    auto v = map_unit_disk<double>(u0, u1);
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
}
void orc_pcg_unit_sphere(uint64_t seed, uint32_t n, double* out3) {
    Pcg64Mcg g = Pcg64Mcg::seed_from_u64(seed);
    for (uint32_t i = 0; i < n; ++i) {
        auto v = pcg_unit_sphere<double>(g);
        out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
    }
}
void orc_pcg_unit_disk(uint64_t seed, uint32_t n, double* out3) {
    Pcg64Mcg g = Pcg64Mcg::seed_from_u64(seed);
    for (uint32_t i = 0; i < n; ++i) {
        auto v = pcg_unit_disk<double>(g);
        out3[3 * i] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
    }
}

}  // extern "C"

// Decision margin of one ray (f64), in units of "how far an evaluation is from flipping a decision"; >= 1 means
// every decision below is safe, < 1 means the ray is excluded from bit-exact checks (and counted).  The smallest of
//   (a) |disc/a| / max(|oc|^2, r^2) / (16 * 2^-24): the discriminant b^2 - a*c is a difference of two numbers
//       of size |oc|^2; 16 fp32 ulps of that cancellation is the band in which an FP32 sign is not trustworthy.
//       Only spheres that could change the answer count: not wholly behind t_min, not wholly beyond the
//       closest hit.  Reported in `margin` only: ROUND 1's bar.  The device decides hit/miss in f64 since round 2.
//   (b) |t_root / t_min - 1| / 1e-6 over the real roots of those spheres: a root sitting on the t_min boundary,
//   (c) (t_2nd - t_best) * |d| / max(t_best * |d|, r_best) / 1e-9: closest vs runner-up near-tie.
// *tie_only receives the margin without term (a): the bar for the device since round 3.  Its decisions (hit/miss, the
// t_min boundary, which of two spheres is closer, exact ties by world order) are all made in f64 on the same fp32
// inputs -- the fp32 scan only proposes candidates, conservatively -- so (b) and (c) are not fp32 bands any more: they
// stand for the last-bit differences between two f64 evaluations of the same formula (reciprocal-multiply instead of
// a division, a Newton square root, fused or unfused multiply-adds), which reach ~1e-13 relative for the r = 1000
// ground.  (Rounds 1-2 had 5 % of t_min and 1e-3 here: the device ordered candidates by fp32 keys.)
static double decision_margin(const Scene<double>& sc, const Ray<double>& ray, double t_min, double* tie_only) {
    const double kDisc = 16.0 / 16777216.0, kTmin = 1e-6, kTie = 1e-9;
    const double a = ray.dir.norm_squared(), len = std::sqrt(a);
    const double inf = std::numeric_limits<double>::infinity();
    double best = inf, second = inf, best_r = 1;
    struct Q { double hb, disc, r, oc2; };
    std::vector<Q> q(sc.spheres.size());
    for (size_t i = 0; i < sc.spheres.size(); ++i) {
        const auto& s = sc.spheres[i];
        Vec3<double> oc = ray.origin - s.center(ray.time);
        double hb = oc.dot(ray.dir), c = oc.norm_squared() - s.radius * s.radius;
        q[i] = {hb, hb * hb - a * c, s.radius, oc.norm_squared()};
        if (q[i].disc >= 0) {
            double root = std::sqrt(q[i].disc);
            double ts[2] = {(-hb - root) / a, (-hb + root) / a};
            double tv = -1;
            for (double t : ts)
                if (tv < 0 && t >= t_min) tv = t;
            if (tv >= 0) {
                if (tv < best) { second = best; best = tv; best_r = s.radius; }
                else if (tv < second) second = tv;
            }
        }
    }
    double m = inf, mt = inf;
    for (const Q& s : q) {
        const double tc = -s.hb / a, ext = 1.5 * s.r / len;  // closest approach +- a generous sphere extent
        if (tc + ext < t_min || tc - ext > best) continue;   // cannot produce a hit in [t_min, best]
        m = std::min(m, std::fabs(s.disc / a) / std::max(s.oc2, s.r * s.r) / kDisc);
        if (s.disc >= 0) {
            double root = std::sqrt(s.disc);
            for (double t : {(-s.hb - root) / a, (-s.hb + root) / a}) mt = std::min(mt, std::fabs(t / t_min - 1.0) / kTmin);
        }
    }
    if (std::isfinite(second)) mt = std::min(mt, (second - best) * len / std::max(best * len, best_r) / kTie);
    if (tie_only) *tie_only = mt;
    return std::min(m, mt);
}

template <class T>
static void intersect_impl(const rt_scene* flat, const rt_ray* rays, uint32_t n, double t_min, int bvh_first,
                           uint64_t bvh_seed, orc_hit* out) {
    Scene<T> sc = Scene<T>::from_flat(flat);
    Scene<double> scd = Scene<double>::from_flat(flat);
    if (bvh_first >= 0) sc.build_bvh(bvh_first, bvh_seed);
    for (uint32_t i = 0; i < n; ++i) {
        Ray<T> r{{T(rays[i].origin[0]), T(rays[i].origin[1]), T(rays[i].origin[2])},
                 {T(rays[i].dir[0]), T(rays[i].dir[1]), T(rays[i].dir[2])}, T(rays[i].time)};
        Hit<T> h;
        bool got = world_hit<T>(sc, r, T(t_min), std::numeric_limits<T>::infinity(), bvh_first >= 0, h);
        orc_hit& o = out[i];
        std::memset(&o, 0, sizeof(o));
        Ray<double> rd{{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]},
                       {rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]}, rays[i].time};
        o.margin = decision_margin(scd, rd, t_min, &o.margin_tie);
        if (!got) continue;
        o.hit = 1;
        o.sphere_id = h.sphere_id;
        o.material_id = (int32_t)sc.spheres[h.sphere_id].mat;
        o.front_face = h.front_face;
        o.t = (double)h.t;
        o.point[0] = h.point.x; o.point[1] = h.point.y; o.point[2] = h.point.z;
        o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
    }
}

extern "C" {

int orc_intersect_batch(const rt_scene* flat, const rt_ray* rays, uint32_t n, double t_min, int bvh_first,
                        uint64_t bvh_seed, int precision, orc_hit* out) {
    if (precision == 32) intersect_impl<float>(flat, rays, n, t_min, bvh_first, bvh_seed, out);
    else intersect_impl<double>(flat, rays, n, t_min, bvh_first, bvh_seed, out);
    return 0;
}

int orc_scatter_batch(const rt_scene* flat, const rt_ray* rays, const rt_hit* hits, const float* randoms4,
                      uint32_t n, orc_scatter* out) {
    Scene<double> sc = Scene<double>::from_flat(flat);
    for (uint32_t i = 0; i < n; ++i) {
        std::memset(&out[i], 0, sizeof(orc_scatter));
        if (!hits[i].hit) continue;
        Ray<double> r{{rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]},
                      {rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]}, rays[i].time};
        Hit<double> h;
        h.hit = true;
        h.sphere_id = hits[i].sphere_id;
        h.t = hits[i].t;
        h.front_face = hits[i].front_face != 0;
        h.point = {hits[i].point[0], hits[i].point[1], hits[i].point[2]};
        h.normal = {hits[i].normal[0], hits[i].normal[1], hits[i].normal[2]};
        const float* u = randoms4 + 4 * i;
        Scatter<double> s = scatter<double>(
            sc, r, h, [&] { return map_unit_sphere<double>(u[0], u[1]); }, [&] { return (double)u[0]; });
        out[i].scattered = 1;
        out[i].albedo[0] = s.albedo.x; out[i].albedo[1] = s.albedo.y; out[i].albedo[2] = s.albedo.z;
        out[i].origin[0] = s.ray.origin.x; out[i].origin[1] = s.ray.origin.y; out[i].origin[2] = s.ray.origin.z;
        out[i].dir[0] = s.ray.dir.x; out[i].dir[1] = s.ray.dir.y; out[i].dir[2] = s.ray.dir.z;
        out[i].time = s.ray.time;
    }
    return 0;
}

int orc_texture_batch(const rt_scene* flat, int32_t tex_id, const float* points3, uint32_t n, double* out3) {
    Scene<double> sc = Scene<double>::from_flat(flat);
    if (tex_id < 0 || tex_id >= (int)sc.textures.size()) return -1;
    for (uint32_t i = 0; i < n; ++i) {
        auto c = texture_value<double>(sc, tex_id, {points3[3 * i], points3[3 * i + 1], points3[3 * i + 2]});
        out3[3 * i] = c.x; out3[3 * i + 1] = c.y; out3[3 * i + 2] = c.z;
    }
    return 0;
}

int orc_camera_rays(const rt_camera* cam, const float* ij2, const float* randoms4, uint32_t n, orc_ray* out) {
    Camera<double> c = Camera<double>::from_flat(cam);
    for (uint32_t k = 0; k < n; ++k) {
        const float* u = randoms4 + 4 * k;
        Ray<double> r = c.get_ray(
            ij2[2 * k], ij2[2 * k + 1], [&] { return map_unit_disk<double>(u[0], u[1]); },
            [&] { return (double)u[2]; });
        out[k].origin[0] = r.origin.x; out[k].origin[1] = r.origin.y; out[k].origin[2] = r.origin.z;
        out[k].dir[0] = r.dir.x; out[k].dir[1] = r.dir.y; out[k].dir[2] = r.dir.z;
        out[k].time = r.time;
    }
    return 0;
}

// main.rs:64-92 on the CPU.  mode: 0 counter RNG (kernel-identical sampling), 1 reference
// samplers.  bvh_first >= 0 puts spheres [bvh_first, n) in a BVH as config.rs:197 does.
int orc_render(const rt_scene* flat, const rt_camera* cam, const rt_params* params, int mode, int bvh_first,
               uint64_t bvh_seed, int n_threads, int precision, uint32_t y_begin, uint32_t y_end,
               double* out_rgb, double* seconds, uint64_t* rays) {
    RenderResult rr;
    if (precision == 32) {
        Scene<float> sc = Scene<float>::from_flat(flat);
        if (bvh_first >= 0) sc.build_bvh(bvh_first, bvh_seed);
        rr = render<float>(sc, Camera<float>::from_flat(cam), *params, mode, bvh_first >= 0, n_threads, out_rgb,
                           y_begin, y_end);
    } else {
        Scene<double> sc = Scene<double>::from_flat(flat);
        if (bvh_first >= 0) sc.build_bvh(bvh_first, bvh_seed);
        rr = render<double>(sc, Camera<double>::from_flat(cam), *params, mode, bvh_first >= 0, n_threads,
                            out_rgb, y_begin, y_end);
    }
    if (seconds) *seconds = rr.seconds;
    if (rays) *rays = rr.rays;
    return 0;
}

uint32_t orc_encode(const double* rgb, uint64_t n_pixels, uint32_t* out) { return encode(rgb, n_pixels, out); }

int orc_bvh_stats(const rt_scene* flat, int bvh_first, uint64_t seed, int* n_nodes) {
    Scene<double> sc = Scene<double>::from_flat(flat);
    sc.build_bvh(bvh_first, seed);
    *n_nodes = (int)sc.nodes.size();
    return 0;
}

}  // extern "C"
