// rt_oracle.hpp — CPU restatement of the ArifRoktim/raytracing path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (raytracing_amd/, the
// C-ABI library, bench.py's GPU leg) may include, link or call this.  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
// checker / reported CPU baseline.
//
// Parity pinning: the reference is Rust and no Rust toolchain exists in the
// build image, so the reference itself cannot be run.  The
// reference's own tests pin only five Vec3 known answers (vec3.rs doctests);
// those are reproduced in tests/test_oracle_kats.py.  Everything else here is
// pinned by line-by-line review against the cited source: for intersection,
// scatter, noise, camera and image results this oracle is "parity unpinned" by
// any reference-held vector.  The third-party RNG (rand 0.7.3 SmallRng =
// rand_pcg 0.2.1 Pcg64Mcg, rand_distr 0.3.0 UnitSphere/UnitDisc) is restated
// from its published algorithm and is likewise unpinned.
//
// Templated on the scalar T: T = double is the parity oracle and CPU baseline
// (the reference is f64 end to end, vec3.rs:10-14); T = float is a comparison aid.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "../include/rtx_hip.h"  // POD scene/camera/ray layouts only (no product code)

namespace orc {

// ---------------------------------------------------------------- Vec3 (vec3.rs)
template <class T>
struct Vec3 {
    T x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
    T operator[](int a) const { return a == 0 ? x : (a == 1 ? y : z); }  // vec3.rs:136-146
    Vec3 operator-() const { return {-x, -y, -z}; }                     // vec3.rs:157-167
    Vec3 operator+(const Vec3& r) const { return {x + r.x, y + r.y, z + r.z}; }
    Vec3 operator-(const Vec3& r) const { return {x - r.x, y - r.y, z - r.z}; }
    Vec3 operator*(const Vec3& r) const { return {x * r.x, y * r.y, z * r.z}; }  // vec3.rs:208-218
    Vec3 operator*(T s) const { return {x * s, y * s, z * s}; }
    Vec3 operator/(T s) const { return {x / s, y / s, z / s}; }  // vec3.rs:246-256 (true division)
    Vec3& operator+=(const Vec3& r) { x += r.x; y += r.y; z += r.z; return *this; }
    Vec3& operator-=(const Vec3& r) { x -= r.x; y -= r.y; z -= r.z; return *this; }
    Vec3& operator*=(T s) { x *= s; y *= s; z *= s; return *this; }
    T norm_squared() const { return x * x + y * y + z * z; }  // vec3.rs:83-85
    T norm() const { return std::sqrt(norm_squared()); }      // vec3.rs:79-81
    T dot(const Vec3& r) const { return x * r.x + y * r.y + z * r.z; }  // vec3.rs:94-96
    Vec3 cross(const Vec3& r) const {                                    // vec3.rs:105-111
        return {y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x};
    }
    static Vec3 normalized(const Vec3& v) { return v / v.norm(); }  // vec3.rs:34-39
    // vec3.rs:113-116: reflect normalises `self` first.
    Vec3 reflect(const Vec3& n) const {
        Vec3 u = normalized(*this);
        return u - n * (T(2) * u.dot(n));
    }
    // vec3.rs:118-123.  Deviation (documented in DESIGN.md): sqrt argument clamped
    // at 0 so that rounding cannot produce NaN; identical otherwise.
    Vec3 refract(const Vec3& n, T eta) const {
        T cos_theta = (-*this).dot(n);
        Vec3 par = (*this + n * cos_theta) * eta;
        T k = T(1) - par.norm_squared();
        Vec3 perp = -n * std::sqrt(k > T(0) ? k : T(0));
        return par + perp;
    }
};
template <class T>
inline Vec3<T> operator*(T s, const Vec3<T>& v) { return v * s; }

template <class T>
struct Ray {  // lib.rs:20-38
    Vec3<T> origin, dir;
    T time;
    Vec3<T> at(T t) const { return origin + dir * t; }
};

template <class T> inline T lerp(T t, T low, T high) { return low * (T(1) - t) + high * t; }  // lib.rs:46-48
template <class T> inline T smooth(T t) { return (t * t * t) * (t * (T(6) * t - T(15)) + T(10)); }  // lib.rs:50-53

// ------------------------------------------------------- counter RNG: Philox4x32-10
// (Salmon et al., SC'11; Random123 reference constants.)  This REPLACES the
// reference's sequential SmallRng for the hot path (north_star); the HIP kernel
// holds an independent implementation that must agree bit for bit.
struct Philox {
    static inline void round(uint32_t c[4], uint32_t k0, uint32_t k1) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static inline void gen(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t out[4]) {
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        uint32_t c[4] = {c0, c1, c2, c3};
        for (int r = 0; r < 10; ++r) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
    // 24-bit uniform in [0,1): exactly representable in float and double.
    static inline double u01(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }
};
// Stream layout shared with the kernel: counter = (pixel index, sample index, dim, 0)
//   dim 0: (jitter_i, jitter_j, lens_u0, lens_u1)   dim 1: (shutter_u, -, -, -)
//   dim 2+b: the four uniforms of bounce b.
enum { DIM_PIXEL = 0, DIM_TIME = 1, DIM_BOUNCE0 = 2 };

// ---------------------------- rand 0.7.3 / rand_pcg 0.2.1 restatement (UNPINNED)
// SmallRng on 64-bit targets = Pcg64Mcg (Mcg128Xsl64).  Used only by the
// "reference sampler" render mode (CPU baseline / statistical cross-check) and
// for BVH axis draws (hit.rs:249).
struct Pcg64Mcg {
    unsigned __int128 state;
    static unsigned __int128 mult() {
        return ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;
    }
    static Pcg64Mcg seed_from_u64(uint64_t s) {  // rand_core 0.5.1 SeedableRng::seed_from_u64
        uint8_t seed[16];
        for (int i = 0; i < 4; ++i) {
            s = s * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xs = (uint32_t)(((s >> 18) ^ s) >> 27);
            uint32_t rot = (uint32_t)(s >> 59);
            uint32_t x = (xs >> rot) | (xs << ((32 - rot) & 31));
            std::memcpy(seed + 4 * i, &x, 4);
        }
        uint64_t lo, hi;
        std::memcpy(&lo, seed, 8);
        std::memcpy(&hi, seed + 8, 8);
        Pcg64Mcg r;
        r.state = (((unsigned __int128)hi << 64) | lo) | 1;
        return r;
    }
    uint64_t next_u64() {
        state *= mult();
        uint32_t rot = (uint32_t)(state >> 122);
        uint64_t xsl = (uint64_t)(state >> 64) ^ (uint64_t)state;
        return (xsl >> rot) | (xsl << ((64 - rot) & 63));
    }
    uint32_t next_u32() { return (uint32_t)next_u64(); }
    double gen_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }  // Standard f64
    // Uniform<f64>::sample / sample_single: value0_1 * scale + low, value0_1 from 52 bits.
    double uniform(double low, double high) {
        uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ull;
        double v12;
        std::memcpy(&v12, &bits, 8);
        return (v12 - 1.0) * (high - low) + low;
    }
    uint32_t gen_axis() {  // Uniform::new(0u8, 3) (vec3.rs:271-280): widening-multiply rejection on u32
        const uint32_t zone = 4294967294u;
        for (;;) {
            uint64_t m = (uint64_t)next_u32() * 3u;
            if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
        }
    }
};
// rand_distr 0.3.0 UnitSphere (Marsaglia 1972) / UnitDisc (rejection) restated.
template <class T>
inline Vec3<T> pcg_unit_sphere(Pcg64Mcg& g) {
    for (;;) {
        double x1 = g.uniform(-1., 1.), x2 = g.uniform(-1., 1.);
        double s = x1 * x1 + x2 * x2;
        if (s >= 1.) continue;
        double f = 2. * std::sqrt(1. - s);
        return {T(x1 * f), T(x2 * f), T(1. - 2. * s)};
    }
}
template <class T>
inline Vec3<T> pcg_unit_disk(Pcg64Mcg& g) {
    for (;;) {
        double x1 = g.uniform(-1., 1.), x2 = g.uniform(-1., 1.);
        if (x1 * x1 + x2 * x2 <= 1.) return {T(x1), T(x2), T(0)};
    }
}

// ------------------------------------------ direct sampling maps (explicit uniforms)
// Same distributions as rand_distr::UnitSphere / UnitDisc (vec3.rs:53-62), written
// as deterministic maps of two uniforms so oracle and kernel can be fed identical
// numbers.  The kernel uses the same maps.
static constexpr double kPi = 3.14159265358979323846;
template <class T>
inline Vec3<T> map_unit_sphere(T u0, T u1) {
    T z = T(1) - T(2) * u0;
    T r2 = T(1) - z * z;
    T r = std::sqrt(r2 > T(0) ? r2 : T(0));
    T phi = T(2) * T(kPi) * u1;
    return {r * std::cos(phi), r * std::sin(phi), z};
}
template <class T>
inline Vec3<T> map_unit_disk(T u0, T u1) {
    T r = std::sqrt(u0);
    T phi = T(2) * T(kPi) * u1;
    return {r * std::cos(phi), r * std::sin(phi), T(0)};
}

// ------------------------------------------------------------------ scene (widened)
template <class T>
struct SphereT {  // shape.rs:16-21, :75-80
    Vec3<T> c0, dc;
    T radius;
    uint32_t mat;
    bool moving;
    Vec3<T> center(T time) const { return moving ? c0 + dc * time : c0; }  // shape.rs:96-98
};

template <class T>
struct Hit {  // hit.rs:11-23 (u,v omitted: never read)
    bool hit = false;
    int sphere_id = -1;
    Vec3<T> point, normal;
    T t = 0;
    bool front_face = false;
};

template <class T>
struct AABB {  // hit.rs:137-189
    Vec3<T> mn, mx;
    AABB surrounding(const AABB& o) const {
        return {{std::min(mn.x, o.mn.x), std::min(mn.y, o.mn.y), std::min(mn.z, o.mn.z)},
                {std::max(mx.x, o.mx.x), std::max(mx.y, o.mx.y), std::max(mx.z, o.mx.z)}};
    }
    bool hit(const Ray<T>& ray, T t0r, T t1r) const {  // hit.rs:163-184
        T start = t0r, end = t1r;
        for (int a = 0; a < 3; ++a) {
            T inv = T(1) / ray.dir[a];
            T t0 = (mn[a] - ray.origin[a]) * inv;
            T t1 = (mx[a] - ray.origin[a]) * inv;
            if (inv < T(0)) std::swap(t0, t1);
            start = std::fmax(start, t0);  // Rust f64::max ignores NaN, like fmax
            end = std::fmin(end, t1);
            if (!(end > start)) return false;
        }
        return true;
    }
};

template <class T>
struct Scene {
    std::vector<SphereT<T>> spheres;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_noise_table> tables;
    // BVH (hit.rs:191-288) over spheres [bvh_first, n); spheres before it are a plain list.
    struct Node {
        AABB<T> box;
        int left, right;  // >=0: node index; <= -2: sphere id -(v+2); -1: Dummy (shape.rs:150-161)
    };
    std::vector<Node> nodes;
    int bvh_root = -1;
    int bvh_first = -1;

    static Scene from_flat(const rt_scene* s) {
        Scene sc;
        sc.spheres.resize(s->n_spheres);
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            auto& sp = sc.spheres[i];
            sp.c0 = {T(s->c0x[i]), T(s->c0y[i]), T(s->c0z[i])};
            sp.dc = {T(s->dcx[i]), T(s->dcy[i]), T(s->dcz[i])};
            sp.radius = T(s->radius[i]);
            sp.mat = s->material_id[i];
            sp.moving = (s->dcx[i] != 0.f || s->dcy[i] != 0.f || s->dcz[i] != 0.f);
        }
        sc.materials.assign(s->materials, s->materials + s->n_materials);
        sc.textures.assign(s->textures, s->textures + s->n_textures);
        sc.tables.assign(s->noise_tables, s->noise_tables + s->n_noise_tables);
        return sc;
    }

    AABB<T> sphere_box(int id) const {  // shape.rs:67-70, :135-145 with shutter 0..1 (config.rs:197)
        const auto& s = spheres[id];
        Vec3<T> rad(s.radius, s.radius, s.radius);
        AABB<T> b0{s.center(T(0)) - rad, s.center(T(0)) + rad};
        if (!s.moving) return b0;
        AABB<T> b1{s.center(T(1)) - rad, s.center(T(1)) + rad};
        return b0.surrounding(b1);
    }
    AABB<T> child_box(int c) const { return c >= 0 ? nodes[c].box : sphere_box(-(c + 2)); }

    int build_rec(std::vector<int> ids, Pcg64Mcg& rng) {  // hit.rs:228-261
        Node n;
        if (ids.size() == 1) {
            n.left = -1;  // Dummy
            n.right = -(ids[0] + 2);
            n.box = sphere_box(ids[0]);
        } else {
            if (ids.size() == 2) {
                n.left = -(ids[1] + 2);  // pop() takes the last first
                n.right = -(ids[0] + 2);
            } else {
                uint32_t axis = rng.gen_axis();
                // sort_unstable_by(min[axis]); tie order is implementation-defined in
                // the reference (pdqsort) — stable sort here; affects tree shape only.
                std::stable_sort(ids.begin(), ids.end(), [&](int a, int b) {
                    return sphere_box(a).mn[axis] < sphere_box(b).mn[axis];
                });
                size_t h = ids.size() / 2;
                std::vector<int> first(ids.begin(), ids.begin() + h), second(ids.begin() + h, ids.end());
                n.left = build_rec(first, rng);
                n.right = build_rec(second, rng);
            }
            n.box = child_box(n.left).surrounding(child_box(n.right));
        }
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }
    void build_bvh(int first, uint64_t seed) {
        nodes.clear();
        bvh_root = -1;
        bvh_first = first;
        if (first < 0 || first >= (int)spheres.size()) { bvh_first = -1; return; }
        std::vector<int> ids;
        for (int i = first; i < (int)spheres.size(); ++i) ids.push_back(i);
        Pcg64Mcg rng = Pcg64Mcg::seed_from_u64(seed);
        bvh_root = build_rec(ids, rng);
    }
};

// shape.rs:35-65 and :101-133 — the half-b quadratic with an UNNORMALISED direction.
// Range semantics: Rust `Range::contains` = t_min <= t < t_max.
template <class T>
inline bool sphere_hit(const Scene<T>& sc, int id, const Ray<T>& ray, T t_min, T t_max, Hit<T>& out) {
    const SphereT<T>& s = sc.spheres[id];
    Vec3<T> center = s.center(ray.time);
    Vec3<T> oc = ray.origin - center;
    T a = ray.dir.norm_squared();
    T half_b = oc.dot(ray.dir);
    T c = oc.norm_squared() - s.radius * s.radius;
    T disc = half_b * half_b - a * c;
    if (disc >= T(0)) {
        T root = std::sqrt(disc);
        T t = (-half_b - root) / a;
        bool ok = (t >= t_min && t < t_max);
        if (!ok) {
            t = (-half_b + root) / a;
            ok = (t >= t_min && t < t_max);
        }
        if (ok) {
            out.hit = true;
            out.sphere_id = id;
            out.t = t;
            out.point = ray.at(t);
            Vec3<T> n = (out.point - center) / s.radius;
            out.front_face = ray.dir.dot(n) < T(0);  // hit.rs:55
            out.normal = out.front_face ? n : -n;    // hit.rs:57-59
            return true;
        }
    }
    return false;
}

template <class T>
inline bool bvh_hit(const Scene<T>& sc, int node, const Ray<T>& ray, T t_min, T t_max, Hit<T>& out) {  // hit.rs:264-283
    const auto& n = sc.nodes[node];
    if (!n.box.hit(ray, t_min, t_max)) return false;
    auto child = [&](int c, T tmax, Hit<T>& h) -> bool {
        if (c == -1) return false;
        if (c <= -2) return sphere_hit(sc, -(c + 2), ray, t_min, tmax, h);
        return bvh_hit(sc, c, ray, t_min, tmax, h);
    };
    Hit<T> hl, hr;
    T end = t_max;
    bool got_l = child(n.left, end, hl);
    if (got_l) end = hl.t;
    if (child(n.right, end, hr)) { out = hr; return true; }
    if (got_l) { out = hl; return true; }
    return false;
}

// hit.rs:100-110 — closest hit over the world; strictly-closer only, first wins ties.
template <class T>
inline bool world_hit(const Scene<T>& sc, const Ray<T>& ray, T t_min, T t_max, bool use_bvh, Hit<T>& out) {
    bool any = false;
    T end = t_max;
    int n = (int)sc.spheres.size();
    int lin_end = (use_bvh && sc.bvh_root >= 0) ? sc.bvh_first : n;
    Hit<T> h;
    for (int i = 0; i < lin_end; ++i)
        if (sphere_hit(sc, i, ray, t_min, end, h)) { end = h.t; out = h; any = true; }
    if (lin_end < n && bvh_hit(sc, sc.bvh_root, ray, t_min, end, h)) { out = h; any = true; }
    return any;
}

// ------------------------------------------------------------ textures (material.rs)
template <class T>
inline int noise_hash(const rt_noise_table& tb, long long x, long long y, long long z) {  // material.rs:309-313
    uint16_t pxy = (uint16_t)(tb.perms[x] + (uint16_t)y);
    uint16_t pz = (uint16_t)(tb.perms[pxy] + (uint16_t)z);
    return tb.perms[pz];
}
template <class T>
inline T value_noise(const rt_noise_table& tb, T freq, Vec3<T> p) {  // material.rs:322-361
    p *= freq;
    Vec3<T> fl(std::floor(p.x), std::floor(p.y), std::floor(p.z));
    Vec3<T> t = p - fl;
    Vec3<T> s(smooth(t.x), smooth(t.y), smooth(t.z));
    long long rx0 = (long long)fl.x & 255, ry0 = (long long)fl.y & 255, rz0 = (long long)fl.z & 255;
    long long rx1 = (rx0 + 1) & 255, ry1 = (ry0 + 1) & 255, rz1 = (rz0 + 1) & 255;
    auto R = [&](long long x, long long y, long long z) { return T(tb.randoms[noise_hash<T>(tb, x, y, z)]); };
    T c000 = R(rx0, ry0, rz0), c100 = R(rx1, ry0, rz0), c010 = R(rx0, ry1, rz0), c110 = R(rx1, ry1, rz0);
    T c001 = R(rx0, ry0, rz1), c101 = R(rx1, ry0, rz1), c011 = R(rx0, ry1, rz1), c111 = R(rx1, ry1, rz1);
    T x00 = lerp(s.x, c000, c100), x10 = lerp(s.x, c010, c110);
    T x01 = lerp(s.x, c001, c101), x11 = lerp(s.x, c011, c111);
    T y0 = lerp(s.y, x00, x10), y1 = lerp(s.y, x01, x11);
    return lerp(s.z, y0, y1);
}
template <class T>
inline T dot_gradient(int perm, T x, T y, T z) {  // material.rs:427-451: the TABLE, not its comments
    switch (perm & 15) {
        case 0: return x + y;
        case 1: return -x + y;
        case 2: return x - y;
        case 3: return -x - y;
        case 4: return x + z;
        case 5: return -x + z;
        case 6: return x - z;
        case 7: return -x - z;
        case 8: return y + z;
        case 9: return -y - z;
        case 10: return y - z;
        case 11: return -y - z;
        case 12: return x + y;
        case 13: return -x + y;
        case 14: return -y + z;
        default: return -y - z;
    }
}
template <class T>
inline T perlin_noise(const rt_noise_table& tb, T freq, Vec3<T> p) {  // material.rs:466-521
    p *= freq;
    Vec3<T> fl(std::floor(p.x), std::floor(p.y), std::floor(p.z));
    Vec3<T> t = p - fl;
    Vec3<T> s(smooth(t.x), smooth(t.y), smooth(t.z));
    long long rx0 = (long long)fl.x & 255, ry0 = (long long)fl.y & 255, rz0 = (long long)fl.z & 255;
    long long rx1 = (rx0 + 1) & 255, ry1 = (ry0 + 1) & 255, rz1 = (rz0 + 1) & 255;
    T x0 = t.x, y0 = t.y, z0 = t.z, x1 = x0 - T(1), y1 = y0 - T(1), z1 = z0 - T(1);
    auto G = [&](long long hx, long long hy, long long hz, T x, T y, T z) {
        return dot_gradient<T>(noise_hash<T>(tb, hx, hy, hz), x, y, z);
    };
    T v000 = G(rx0, ry0, rz0, x0, y0, z0), v100 = G(rx1, ry0, rz0, x1, y0, z0);
    T v010 = G(rx0, ry1, rz0, x0, y1, z0), v110 = G(rx1, ry1, rz0, x1, y1, z0);
    T v001 = G(rx0, ry0, rz1, x0, y0, z1), v101 = G(rx1, ry0, rz1, x1, y0, z1);
    T v011 = G(rx0, ry1, rz1, x0, y1, z1), v111 = G(rx1, ry1, rz1, x1, y1, z1);
    T x00 = lerp(s.x, v000, v100), x10 = lerp(s.x, v010, v110);
    T x01 = lerp(s.x, v001, v101), x11 = lerp(s.x, v011, v111);
    T ya = lerp(s.y, x00, x10), yb = lerp(s.y, x01, x11);
    T n = lerp(s.z, ya, yb);
    return (n + T(1)) * T(0.5);
}
template <class T>
inline T powi(T b, int n) {  // f64::powi: repeated multiplication
    T r = T(1);
    for (int i = 0; i < n; ++i) r *= b;
    return r;
}
// ValueNoise::eval / PerlinNoise::eval (material.rs:315-320, :459-464) with the
// NoiseAdapter callbacks (material.rs:184-263) expressed as a descriptor.
template <class T>
inline T noise_eval(const Scene<T>& sc, const rt_texture& tx, Vec3<T> p) {
    const rt_noise_table& tb = sc.tables[tx.table_id];
    T freq = T(tx.freq);
    auto base = [&](Vec3<T> q) {
        return tx.kind == RT_TEX_VALUE_NOISE ? value_noise<T>(tb, freq, q) : perlin_noise<T>(tb, freq, q);
    };
    T v;
    if (tx.op == RT_NOISE_PLAIN) {
        v = base(p);
    } else {
        T gain = T(tx.gain), lac = T(tx.lacunarity);
        T mx = (T(1) - powi(gain, tx.layers)) / (T(1) - gain);  // material.rs:189,216
        T sum = 0, amp = 1;
        Vec3<T> q = p;
        for (int l = 0; l < tx.layers; ++l) {
            T nz = base(q);
            if (tx.op == RT_NOISE_TURBULENCE) nz = std::fabs(T(2) * nz - T(1));  // material.rs:224-226
            sum += nz * amp;
            q *= lac;
            amp *= gain;
        }
        v = sum / mx;
    }
    if (tx.marbled) {  // material.rs:252-259
        T n = T(tx.marble_gain) * v + p.z;
        n = T(tx.sin_freq) * n + T(tx.phase);
        v = (std::sin(n) + T(1)) * T(0.5);
    }
    return v;
}
template <class T>
inline Vec3<T> texture_value(const Scene<T>& sc, int tex_id, const Vec3<T>& p) {
    const rt_texture& tx = sc.textures[tex_id];
    switch (tx.kind) {
        case RT_TEX_SOLID: return {T(tx.color_a[0]), T(tx.color_a[1]), T(tx.color_a[2])};  // color.rs:31-35
        case RT_TEX_CHECKER: {  // material.rs:155-165
            T f = T(tx.freq);
            bool parity = std::sin(p.x * f) < T(0);
            parity ^= std::sin(p.y * f) < T(0);
            parity ^= std::sin(p.z * f) < T(0);
            const float* c = parity ? tx.color_a : tx.color_b;
            return {T(c[0]), T(c[1]), T(c[2])};
        }
        default: {
            T v = noise_eval<T>(sc, tx, p);  // material.rs:369-373, :529-533: white * eval
            return {v, v, v};
        }
    }
}

// ---------------------------------------------------------- materials (material.rs)
template <class T>
struct Scatter {
    Vec3<T> albedo;
    Ray<T> ray;
};
// Dielectric decision needs its uniform only when not TIR (short-circuit,
// material.rs:101-102): `need_u` tells a sequential-RNG caller whether to draw.
template <class T>
inline T schlick(T cos, T eta) {  // material.rs:84-88
    T r0 = (T(1) - eta) / (T(1) + eta);
    r0 = r0 * r0;
    T m = T(1) - cos;
    return r0 + (T(1) - r0) * (m * m * m * m * m);
}
// Guard shared with the kernel (documented deviation): a Lambertian direction
// n + S shorter than 1e-6 is replaced by n, so `normalized` cannot divide by 0
// (the reference has no guard, material.rs:37 / vec3.rs:34-39).
static constexpr double kDegenerateDir2 = 1e-12;

template <class T, class SphereSampler, class UniformSampler>
inline Scatter<T> scatter(const Scene<T>& sc, const Ray<T>& ray, const Hit<T>& hit,
                          SphereSampler&& unit_sphere, UniformSampler&& uniform) {
    const rt_material& m = sc.materials[sc.spheres[hit.sphere_id].mat];
    Scatter<T> out;
    out.ray.origin = hit.point;
    out.ray.time = ray.time;
    if (m.type == RT_MAT_LAMBERTIAN) {  // material.rs:36-41
        Vec3<T> dir = hit.normal + unit_sphere();
        if (dir.norm_squared() < T(kDegenerateDir2)) dir = hit.normal;
        out.ray.dir = dir;
        out.albedo = texture_value<T>(sc, m.texture_id, hit.point);
    } else if (m.type == RT_MAT_METAL) {  // material.rs:61-72
        Vec3<T> fuzz = unit_sphere() * T(m.fuzz);
        Vec3<T> refl = ray.dir.reflect(hit.normal) + fuzz;
        if (refl.dot(hit.normal) <= T(0)) refl -= fuzz * T(2);
        out.ray.dir = refl;
        out.albedo = {T(m.albedo[0]), T(m.albedo[1]), T(m.albedo[2])};
    } else {  // material.rs:91-111
        T eta = hit.front_face ? T(1) / T(m.ior) : T(m.ior);
        Vec3<T> ud = Vec3<T>::normalized(ray.dir);
        T cos_theta = std::min((-ud).dot(hit.normal), T(1));
        T sin_theta = std::sqrt(T(1) - cos_theta * cos_theta);
        Vec3<T> dir;
        if (eta * sin_theta > T(1) || uniform() < schlick<T>(cos_theta, eta))
            dir = ud.reflect(hit.normal);
        else
            dir = ud.refract(hit.normal, eta);
        out.ray.dir = dir;
        out.albedo = {T(1), T(1), T(1)};
    }
    return out;
}

// ------------------------------------------------------------- camera (screen.rs)
template <class T>
struct Camera {
    Vec3<T> origin, lower_left, horiz, vert, u, v;
    T lens_radius, sh_open, sh_close;
    bool has_shutter;
    static Camera from_flat(const rt_camera* c) {
        auto V = [](const float* f) { return Vec3<T>(T(f[0]), T(f[1]), T(f[2])); };
        Camera k;
        k.origin = V(c->origin); k.lower_left = V(c->lower_left); k.horiz = V(c->horiz);
        k.vert = V(c->vert); k.u = V(c->u); k.v = V(c->v);
        k.lens_radius = T(c->lens_radius);
        k.sh_open = T(c->shutter_open); k.sh_close = T(c->shutter_close);
        k.has_shutter = c->has_shutter != 0;
        return k;
    }
    // screen.rs:79-94 with the two samplers supplied by the caller.
    template <class Disk, class Uni>
    Ray<T> get_ray(T i, T j, Disk&& unit_disk, Uni&& uniform01) const {
        Vec3<T> o = origin;
        if (lens_radius != T(0)) {
            Vec3<T> rd = unit_disk() * lens_radius;
            o = origin + (u * rd.x + v * rd.y);
        }
        T time = has_shutter ? uniform01() * (sh_close - sh_open) + sh_open : T(0);
        return {o, lower_left + horiz * i + vert * j - o, time};
    }
};

// ------------------------------------------------- integrator (main.rs:110-139)
template <class T, class SphereSampler, class UniformSampler>
inline Vec3<T> ray_color(const Scene<T>& sc, Ray<T> ray, uint32_t max_depth, T t_min, bool use_bvh,
                         uint64_t& rays, SphereSampler&& sphere_at, UniformSampler&& uniform_at) {
    Vec3<T> color(1, 1, 1);
    uint32_t bounces = max_depth;
    uint32_t b = 0;
    Hit<T> hit;
    for (;;) {
        ++rays;
        hit = Hit<T>();
        if (!world_hit<T>(sc, ray, t_min, std::numeric_limits<T>::infinity(), use_bvh, hit)) break;
        Scatter<T> s = scatter<T>(sc, ray, hit, [&] { return sphere_at(b); }, [&] { return uniform_at(b); });
        color = color * s.albedo;
        ray = s.ray;
        ++b;
        if (--bounces == 0) {
            color = color * T(0);
            break;
        }
    }
    Vec3<T> ud = Vec3<T>::normalized(ray.dir);
    T t = T(0.5) * (ud.y + T(1));
    Vec3<T> sky = Vec3<T>(1, 1, 1) * (T(1) - t) + Vec3<T>(T(0.5), T(0.7), T(1)) * t;
    return sky * color;
}

enum RenderMode {
    MODE_COUNTER = 0,   // Philox streams + direct maps: the kernel's sampling, path for path
    MODE_REFERENCE = 1  // sequential per-row Pcg64Mcg + rejection samplers, as main.rs:71-87
};

struct RenderResult {
    double seconds = 0;
    uint64_t rays = 0;
};

// main.rs:64-92.  Rows are handed to worker threads dynamically, like rayon's
// par_chunks_exact_mut (screen.rs:51-53).
template <class T>
RenderResult render(const Scene<T>& sc, const Camera<T>& cam, const rt_params& p, int mode, bool use_bvh,
                    int n_threads, double* out_rgb /* W*H*3 means */, uint32_t y_begin = 0,
                    uint32_t y_end = 0xFFFFFFFFu) {
    const uint32_t W = p.width, H = p.height;
    if (y_end > H) y_end = H;
    std::atomic<uint32_t> next_row{y_begin};
    std::atomic<uint64_t> total_rays{0};
    auto work = [&]() {
        uint64_t rays = 0;
        for (;;) {
            uint32_t y = next_row.fetch_add(1);
            if (y >= y_end) break;
            Pcg64Mcg rng = Pcg64Mcg::seed_from_u64((p.seed + 1) * (uint64_t)y);  // main.rs:71
            for (uint32_t x = 0; x < W; ++x) {
                Vec3<T> avg(0, 0, 0);
                for (uint32_t s = 0; s < p.spp; ++s) {
                    Vec3<T> c;
                    if (mode == MODE_COUNTER) {
                        uint32_t pix = y * W + x, smp = p.spp_offset + s;
                        uint32_t r0[4], r1[4];
                        Philox::gen(p.seed, pix, smp, DIM_PIXEL, 0, r0);
                        T ri = p.antialias ? T(Philox::u01(r0[0])) : T(0);
                        T rj = p.antialias ? T(Philox::u01(r0[1])) : T(0);
                        T i = (T(x) + ri) / (T(W) - T(1));
                        T j = T(1) - (T(y) + rj) / (T(H) - T(1));
                        Ray<T> ray = cam.get_ray(
                            i, j, [&] { return map_unit_disk<T>(T(Philox::u01(r0[2])), T(Philox::u01(r0[3]))); },
                            [&] { Philox::gen(p.seed, pix, smp, DIM_TIME, 0, r1); return T(Philox::u01(r1[0])); });
                        uint32_t cached_b = 0xFFFFFFFFu, rb[4] = {0, 0, 0, 0};
                        auto draw = [&](uint32_t b) {
                            if (b != cached_b) { Philox::gen(p.seed, pix, smp, DIM_BOUNCE0 + b, 0, rb); cached_b = b; }
                        };
                        c = ray_color<T>(
                            sc, ray, p.max_depth, T(p.t_min), use_bvh, rays,
                            [&](uint32_t b) { draw(b); return map_unit_sphere<T>(T(Philox::u01(rb[0])), T(Philox::u01(rb[1]))); },
                            [&](uint32_t b) { draw(b); return T(Philox::u01(rb[0])); });
                    } else {
                        T ri = 0, rj = 0;
                        if (p.antialias) { ri = T(rng.gen_f64()); rj = T(rng.gen_f64()); }  // main.rs:76-80
                        T i = (T(x) + ri) / (T(W) - T(1));
                        T j = T(1) - (T(y) + rj) / (T(H) - T(1));
                        Ray<T> ray = cam.get_ray(
                            i, j, [&] { return pcg_unit_disk<T>(rng); }, [&] { return T(rng.gen_f64()); });
                        c = ray_color<T>(
                            sc, ray, p.max_depth, T(p.t_min), use_bvh, rays,
                            [&](uint32_t) { return pcg_unit_sphere<T>(rng); },
                            [&](uint32_t) { return T(rng.gen_f64()); });
                    }
                    avg += c;
                }
                avg = avg / T(p.spp);  // main.rs:88
                double* o = out_rgb + 3 * ((size_t)y * W + x);
                o[0] = (double)avg.x; o[1] = (double)avg.y; o[2] = (double)avg.z;
            }
        }
        total_rays += rays;
    };
    auto t0 = std::chrono::steady_clock::now();  // main.rs:62
    if (n_threads <= 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < n_threads; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    RenderResult r;
    r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();  // main.rs:93
    r.rays = total_rays.load();
    return r;
}

// screen.rs:26-45 — gamma 2, x255.99, truncate, 0RGB.  The reference panics on
// NaN / out-of-[0,1]; here such pixels are counted and clamped.
inline uint32_t encode(const double* rgb, size_t n_pixels, uint32_t* out) {
    uint32_t invalid = 0;
    for (size_t i = 0; i < n_pixels; ++i) {
        double c[3] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        bool bad = false;
        for (double& v : c)
            if (!(v >= 0.0 && v <= 1.0)) { bad = true; v = (v > 1.0) ? 1.0 : 0.0; }
        invalid += bad;
        uint32_t r = (uint32_t)(255.99 * std::sqrt(c[0]));
        uint32_t g = (uint32_t)(255.99 * std::sqrt(c[1]));
        uint32_t b = (uint32_t)(255.99 * std::sqrt(c[2]));
        out[i] = (r << 16) | (g << 8) | b;
    }
    return invalid;
}

}  // namespace orc
