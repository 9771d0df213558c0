// rt_device.h — gfx950 device code of the path-tracing hot path (included by rt_lib.hip only).
//
// Written for CDNA4 directly: 64-wide wavefronts, two rays per lane in packed fp32, the sphere lists fetched with
// scalar loads and used from SGPRs, fp32 VALU for the N-sphere scan, f64 VALU
// for the one winning sphere per bounce, no MFMA (nothing here is a dense contraction).
//
// Reference behaviour restated (file:line in /root/reference/src):
//   closest hit     hit.rs:100-110 (HitList::hit), shape.rs:35-65, :101-133 (sphere quadratics)
//   hit record      hit.rs:45-61 (front_face / normal flip)
//   materials       material.rs:36-41 (Lambertian), :61-72 (Metal), :91-111 (Dielectric), :84-88 (schlick)
//   vector helpers  vec3.rs:113-123 (reflect normalises first; refract)
//   textures        color.rs:31-35, material.rs:155-165 (checker), :322-361 (value noise),
//                   :427-521 (Perlin), :184-263 (fBm / turbulence / marbled)
//   camera          screen.rs:79-94
//   integrator      main.rs:110-139, pixel/sample loop main.rs:75-89
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/rtx_hip.h"
#include "scan_layout.h"  // kChunk, kGroup, kLevels, bound_level_*: shared with the device-free layout code

// Census builds (-DRT_CENSUS -S): comments in the ISA that split a pass into regions for tools/isa_census.py
#ifdef RT_CENSUS
#define RT_MARK(name) asm volatile("; RT_MARK " name)
#else
#define RT_MARK(name)
#endif

namespace rtd {

#ifndef RT_TAILCOND2
#define RT_TAILCOND2 __builtin_amdgcn_ballot_w64(hit >= 0) != 0
#endif
#ifndef RT_TAILCOND
#define RT_TAILCOND __builtin_expect(__builtin_amdgcn_ballot_w64(any >= 0) != 0, 0)
#endif
#ifndef RT_SPARSE_RAYS
#define RT_SPARSE_RAYS 16
#endif
constexpr int kSparseRays = RT_SPARSE_RAYS;
constexpr uint32_t kInfBits = 0x7F800000u;
constexpr float kTwoPow32 = 4294967296.0f;

struct DevTexture {
    rt_texture t;
    float inv_fractal_max;  // (1 - gain) / (1 - gain^layers), material.rs:189,216 (computed in f64 on upload)
    float freq_turns;       // freq / 2pi (computed in f64 on upload): the checker needs only the SIGN of sin(freq * x)
    int32_t fused_id;       // noise textures: index of the FusedNoise image (table, kind) this texture reads
    int32_t pad;
};

// What the noise kernels read, built on upload from an rt_noise_table for ONE noise kind and staged in LDS.
// hash(x, y, z) = perms[perms[perms[x] + y] + z] (material.rs:309-313) followed by the corner term of that hash
// value; the last lookup and the corner term are fused into one 16-byte entry indexed by i = perms[perms[x]+y]+z:
//   Perlin  (material.rs:427-451): g[i] = {gx, gy, gz, 0}, the gradient of case perms[i] & 15 (TABLE, incl. its
//                                  duplicated cases 9/11/15 = -y-z and 14 = -y+z), corner = g . (tx, ty, tz)
//   value   (material.rs:339-347): g[i] = {0, 0, 0, randoms[perms[i]]}
// so a corner costs one ds_read_b128 and three FMAs for either kind (the 16-way gradient switch was ~12 VALU
// instructions per corner and a third dependent LDS level).
struct FusedNoise {
    uint16_t perms[512];  // all 512 entries are data (the second half is NOT perms[i & 255], material.rs:294-299)
    float4 g[512];
};
static_assert(sizeof(FusedNoise) == 9216, "FusedNoise layout");
constexpr int kMaxFused = 6;  // 6 x 9 KiB + the per-wave accumulators stay under the 64 KiB dynamic-LDS limit

// Device view of an uploaded scene.
// fp32 scan lists (read with scalar loads: wave-uniform index)
// All centres are relative to `origin` (the middle of the small spheres' bounding box, chosen on upload): the
// expanded form of the static test cancels |c|^2 + |o|^2 against 2 c.o, so coordinates are kept small.
struct ScanLists {
    const float4* stat;     // ns_pad x {cx, cy, cz, |c|^2 - r^2}; pads have w = +inf (never nominated)
    const float* stat_nr2;  // ns_pad x -r^2 (read only by the root-extraction side)
    // bounding spheres {bx, by, bz, |b|^2 - R^2} (w = +inf: nothing inside), kLevels levels one after another:
    // level 0 = every group of kGroup spheres, level L >= 1 = every run of 4^(L-1) chunks (16 * 4^(L-1) spheres)
    const float4* stat_bound;
    // The PLANAR static list: static small spheres whose centres share one coordinate exactly (objects of one size
    // resting on a floor: the reference's final scenes), laid out like the static list.  The shared coordinate's term
    // drops out of every test as a per-ray constant (Pair::nod_p, qb_p): 6 packed ops per member instead of 8, 10 VALU
    // per bound instead of 12.  Bounds as for the static list (their shared coordinate is plan_val too).  Scan slots
    // [ns_pad, ns_pad + np_pad).
    const float4* plan;
    const float* plan_nr2;
    const float4* plan_bound;
    int32_t np_pad;
    int32_t plan_axis;      // which axis the shared coordinate is on: 0, 1, 2
    float plan_val;         // the shared coordinate, in the scan frame
    int32_t plan_levels, plan_levels_sparse;
    const float4* mov_a;    // nm_pad x {c0x, c0y, c0z, -r^2}; pads -r^2 = +inf
    const float4* mov_b;    // nm_pad x {dcx, dcy, dcz, 0}
    const float* mov_dy;    // nm_pad x dcy: all the "y only" walk reads of the motion (one s_load_dwordx4 per group instead of a dwordx16)
    const float4* mov_bound;  // the same levels for the moving list, over the whole motion for time in [0, 1]
    int32_t ns_pad, nm_pad;
    float ox, oy, oz;       // origin of the scan frame (world coordinates)
    float cmax;             // max |c| over the static list, in the scan frame
    float key_eps0;         // absolute part of a key's error bound: kKeyEps0 * max(|c| + r) over the scan lists (scan frame)
    int32_t mov_axes;       // bit k set: some moving sphere has dc[k] != 0 (terms_moving skips the other axes)
    int32_t stat_levels, mov_levels;  // bit L >= 1 set: bound level L is worth testing for this list (upload decides)
    // the same for a wave with at most kSparseRays live rays (the end of a launch, when the last long paths are traced
    // a pass at a time): few rays meet few nodes, so every level with at least two nodes is worth its tests
    int32_t stat_levels_sparse, mov_levels_sparse;
};
// A "big" sphere as the per-ray f64 test reads it: the exact f64 images of the caller's fp32 numbers (r2 = r * r is
// exact in f64), fetched with ONE scalar load per sphere (wave-uniform index) and used straight from SGPRs.
struct BigSphere {
    double cx, cy, cz, r2;
    double dcx, dcy, dcz, inv_r;  // inv_r: 1/r rounded to fp32, as in DevScene::rec_a
};
static_assert(sizeof(BigSphere) == 64, "BigSphere = one s_load_dwordx16");
// What scatter reads for the sphere a path hit, flattened per sphere (48 B, three 16-byte loads in ONE round trip
// instead of the chain sphere -> material id -> material -> texture): the material and, for a Lambertian, its solid /
// checker texture (material.rs:125-165).  A Lambertian with a noise texture carries the texture's id instead.
struct ShadeRec {
    int32_t type;       // RT_MAT_*
    float fuzz, ior;
    int32_t noise_tex;  // >= 0: Lambertian whose albedo is noise texture `noise_tex` (white times its value)
    float a[3];         // Metal: albedo; Lambertian: colour of a solid texture / odd colour of a checker
    int32_t kind;       // RT_TEX_SOLID or RT_TEX_CHECKER
    float b[3];         // even colour of a checker
    float freq_turns;   // checker frequency / 2 pi
};
static_assert(sizeof(ShadeRec) == 48, "ShadeRec layout");
struct DevScene {
    ScanLists lists;
    const int32_t* scan_id;  // scan slot (static slots first, then planar, then moving) -> sphere id
    // spheres too large for the fp32 scan (|c|,r >> 1: e.g. the r = 1000 ground): f64 test
    const int32_t* big_id;
    const BigSphere* big;  // n_big records, in big_id order
    int32_t n_big;
    // the nominee's f64 decision reads these in SCAN-SLOT order (no id indirection): {c0x, c0y, c0z, r}, {dcx, dcy, dcz, 0}
    const float4* slot_a;
    const float4* slot_dc;
    // the f64 hit record of the winning sphere reads these in world order: {c0x, c0y, c0z, 1/r rounded to fp32}, {dcx, dcy, dcz, 0}
    const float4* rec_a;
    const float4* rec_dc;
    const uint32_t* mat_id;
    const ShadeRec* shade;   // world order
    const DevTexture* texs;
    const FusedNoise* fused;  // n_fused images, copied to LDS by every workgroup that evaluates textures
    int32_t n_fused;
    int32_t n_spheres;
    int32_t n_textures;
    int32_t any_moving;  // some sphere has dc != 0 (else the f64 tests skip the dc gathers: c0 + time * 0 = c0 exactly)
};

// Kernel arguments are re-read from the KERNARG segment (constant address space => scalar loads) at chosen points
// instead of living in SGPRs for the whole kernel: see render_kernel.  `launder` makes the pointer opaque to the
// optimiser, so loads through the result cannot be hoisted above that point.
#define RT_AS4 __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const RT_AS4 T* launder(const RT_AS4 T* p) {
    asm volatile("" : "+s"(p));
    return p;
}

// LDS image: only the fused noise tables (gathered per lane: 14 lookups per noise call).  The sphere
// lists are NOT staged in LDS: they are wave-uniform and go through the scalar cache into SGPRs (see the scan).
struct LdsScene {
    const FusedNoise* fused;
};

__host__ __device__ inline size_t lds_scene_bytes(int n_fused) { return (size_t)n_fused * sizeof(FusedNoise); }

__device__ inline LdsScene stage_scene(const DevScene& S, unsigned char* lds, int tid, int nthreads) {
    uint4* tab = reinterpret_cast<uint4*>(lds);
    const uint4* gt = reinterpret_cast<const uint4*>(S.fused);
    const int nt = S.n_fused * (int)(sizeof(FusedNoise) / 16);
    for (int i = tid; i < nt; i += nthreads) tab[i] = gt[i];
    __syncthreads();
    return {reinterpret_cast<const FusedNoise*>(tab)};
}

// ------------------------------------------------------------------ counter RNG
// Philox4x32-10 (Salmon et al. 2011).  Independent of the oracle's copy; tests compare the two
// bit for bit and against the Random123 known-answer vectors.
__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                                              uint32_t c3, uint32_t out[4]) {
#ifndef RT_PHILOX_ROUNDS
#define RT_PHILOX_ROUNDS 10  // anything else is a timing experiment, not Philox4x32-10
#endif
#pragma unroll
    for (int r = 0; r < RT_PHILOX_ROUNDS; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        c1 = (uint32_t)p1;
        c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// counter = (pixel, sample, dim, 0); dims as in oracle/rt_oracle.hpp
enum { DIM_PIXEL = 0, DIM_TIME = 1, DIM_BOUNCE0 = 2 };

// ------------------------------------------------------------------ small vector helpers
struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// Shading uses the 1-ulp hardware v_rsq_f32 / v_rcp_f32 / v_sqrt_f32 (not the IEEE-correct multi-instruction
// expansions): the parity bar for directions is 1e-5 relative, these are ~1e-7.
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ f3 normalize(f3 a) {
    const float inv = fast_rsq(dot(a, a));
    return {a.x * inv, a.y * inv, a.z * inv};
}

struct Ray {
    f3 o, d;
    float time;
};
struct HitRec {
    int32_t sphere;  // -1 = miss
    int32_t front;
    float t;
    f3 p, n;
};

// sin / cos of u TURNS (2 pi u), u in [0, 1): v_sin_f32 / v_cos_f32 take their argument in turns and are within
// 1.3e-7 (absolute) of the exact value on that range (measured over all 2^24 multiples of 2^-24), the same as the
// 43-instruction sincospif expansion they replace.
__device__ __forceinline__ float turn_sin(float u) { return __builtin_amdgcn_sinf(u); }
__device__ __forceinline__ float turn_cos(float u) { return __builtin_amdgcn_cosf(u); }
// uniform on the unit sphere surface / in the unit disc from two uniforms (same maps as the
// oracle; same distributions as rand_distr::UnitSphere / UnitDisc used at vec3.rs:53-62)
__device__ __forceinline__ f3 map_unit_sphere(float u0, float u1) {
    const float z = 1.0f - 2.0f * u0;
    const float r = fast_sqrt(fmaxf(0.0f, 1.0f - z * z));
    const float s = turn_sin(u1), c = turn_cos(u1);
    return {r * c, r * s, z};
}
__device__ __forceinline__ f3 map_unit_disk(float u0, float u1) {
    const float r = fast_sqrt(u0);
    const float s = turn_sin(u1), c = turn_cos(u1);
    return {r * c, r * s, 0.0f};
}

// ------------------------------------------------------------------ the fp32 sphere scan
// One sphere against a ray, in a frame where the ray starts at o' = o + t_min*d and has a UNIT direction d^.
// The key is the bit pattern of t' = (t - t_min)*|d| for the root the reference would take, or a pattern
// that loses every unsigned min (negative -> sign bit set, miss -> NaN):
//   oc = c - o'     b = oc.d^     q = |oc|^2 - r^2  (< 0: o' inside the sphere)     disc = b^2 - q
//   roots t' = b -+ sqrt(disc); their product is q, so exactly one root is positive when o' is inside
//   (take b + s) and both have the sign of b when outside (take b - s): this is shape.rs:53-61's
//   "near root if in range else far root" with the range start folded into o'.
//
// How it is mapped to gfx950 (measured, tools/ubench*.hip; DESIGN.md section 4 has the numbers):
//  * A wave64 VALU instruction issues in ~4.4 cycles whether or not its operands are SGPRs, and
//    v_pk_{add,mul,fma}_f32 cost the same as their scalar forms while doing two lanes' worth: the 157 TF fp32
//    peak exists only for packed math.  So EVERY LANE CARRIES TWO RAYS (A and B) and the ten arithmetic ops of a
//    test up to disc are packed {A,B}.
//  * The sphere is wave-uniform: it is fetched with SCALAR loads (s_load_dwordx16 = 4 spheres, constant
//    address space, served by the scalar cache) and used straight from SGPRs, so the loop has no LDS or
//    vector-memory instruction and holds no sphere in VGPRs.
//  * f32 MFMA does not help: v_mfma_f32_32x32x2_f32 and VALU fp32 do NOT overlap on this part (65 + 38 cycles
//    measured 106 together), they share the fp32 datapath.
//  * v_sqrt_f32 (~8 cycles), v_bfi (copysign), the subtract and the min/select are un-packable: they run lazily,
//    only for groups of spheres that some lane's line meets (see scan_spheres).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v4f* cv4p;  // constant address space => scalar loads
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ cv4p as_scalar(const float4* p) { return (cv4p)(uintptr_t)p; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// the two rays of a lane, in the scan frame (component-wise pairs {A, B})
struct Pair {
    f2 m2ox, m2oy, m2oz;  // -2 o', o' = o + t_min*d - origin (the direct forms recover o' = -0.5 * m2o exactly)
    f2 dx, dy, dz;        // unit direction (x kScanInflate)
    f2 time;
    // per-ray terms of the expanded static test
    f2 nod;               // -(o' . d^)
    f2 qb;                // |o'|^2 - E, E = kExpandErr (|o'| + cmax)^2: the error bound of the expanded disc
    // the planar list's shared coordinate folded into the per-ray constants: nod_p = nod + plan_val * d_k,
    // qb_p = qb + plan_val * (-2 o'_k)
    f2 nod_p, qb_p;
};

// STATIC spheres, every sphere: the EXPANDED form, 8 packed ops for both rays (the direct form below needs 10):
//   b = c.d^ - o'.d^          3 fma   (sphere components are SGPR broadcasts; nod = -(o'.d^) per ray)
//   q = (|c|^2 - r^2) + |o'|^2 - 2 c.o'    1 add + 3 fma   (w = |c|^2 - r^2 per sphere in f64 on upload)
//   disc = b^2 - q            1 fma
// It cancels numbers of size (|c| + |o'|)^2 =: M^2, so it is used ONLY as a conservative pre-filter: its rounding
// error is below 26 ulp of M^2 (3 roundings in b, squared: 8 |b| M; w, |o'|^2, their sum and three fma in q: 6 M^2;
// b^2 and disc: 2 M^2; the rounding of c and o' into the scan frame: 4 M^2; the normalisation of d^: 5 M^2), and
// E = 32 ulp of M^2 is subtracted from q through the per-ray term qb, so a sphere the exact line meets always has
// disc >= 0 here.  A sphere that passes is tested again in the DIRECT form (terms_static_direct) on the
// root-extraction side, which is as exact as fp32 gets (error ~ulp of |oc|^2, covered by kScanInflate) and
// yields the key; the pre-filter's false positives (a band of ~E / 2r around the limb: 1-2 % of r for the
// final scene seen from its camera) die there as disc < 0 without ever being nominated.
constexpr float kExpandErr = 32.0f / 16777216.0f;
__device__ __forceinline__ f2 disc_static_expanded(v4f s, const Pair& r) {
    const f2 b = fma2(f2{s.z, s.z}, r.dz, fma2(f2{s.y, s.y}, r.dy, fma2(f2{s.x, s.x}, r.dx, r.nod)));
    const f2 q = fma2(f2{s.z, s.z}, r.m2oz, fma2(f2{s.y, s.y}, r.m2oy, fma2(f2{s.x, s.x}, r.m2ox, s.w + r.qb)));
    return fma2(b, b, -q);
}
// The bound of a group: the same expanded test plus "not wholly behind the ray".  A line test alone passes for
// spheres behind the origin (measured: ~60 % of the group bounds pass it, 34 % once those are excluded).  Behind
// means b < 0 (centre behind) and q > 0 (origin outside); -q is what the chain computes here (negations are free
// source modifiers), so per ray one v_and_or_b32 folds the three sign bits: sign set <=> this ray cannot meet the
// bound.  q carries -E, so "outside" is decided conservatively; b's error (~1e-5 of the scene size) only matters
// for origins within that distance of the bound's surface, where disc < 0 rejects anyway.
__device__ __forceinline__ int bound_reject_bits(v4f s, const Pair& r) {
    const f2 b = fma2(f2{s.z, s.z}, r.dz, fma2(f2{s.y, s.y}, r.dy, fma2(f2{s.x, s.x}, r.dx, r.nod)));
    const f2 nq = fma2(f2{s.z, s.z}, -r.m2oz, fma2(f2{s.y, s.y}, -r.m2oy, fma2(f2{s.x, s.x}, -r.m2ox, -s.w - r.qb)));
    const f2 disc = fma2(b, b, nq);
    const int ux = (__float_as_int(b.x) & __float_as_int(nq.x)) | __float_as_int(disc.x);
    const int uy = (__float_as_int(b.y) & __float_as_int(nq.y)) | __float_as_int(disc.y);
    return ux & uy;  // sign set <=> neither ray of this lane can meet the bound
}
// The planar list's versions, kAxis = the shared coordinate's axis: its terms are in nod_p / qb_p and are left out.
template <int kAxis>
__device__ __forceinline__ void planar_b_q(v4f s, const Pair& r, f2 q0, f2& b, f2& q) {
    b = r.nod_p;
    q = q0;
    if constexpr (kAxis != 0) { b = fma2(f2{s.x, s.x}, r.dx, b); q = fma2(f2{s.x, s.x}, r.m2ox, q); }
    if constexpr (kAxis != 1) { b = fma2(f2{s.y, s.y}, r.dy, b); q = fma2(f2{s.y, s.y}, r.m2oy, q); }
    if constexpr (kAxis != 2) { b = fma2(f2{s.z, s.z}, r.dz, b); q = fma2(f2{s.z, s.z}, r.m2oz, q); }
}
template <int kAxis>
__device__ __forceinline__ f2 disc_planar_expanded(v4f s, const Pair& r) {
    f2 b, q;
    planar_b_q<kAxis>(s, r, s.w + r.qb_p, b, q);
    return fma2(b, b, -q);
}
template <int kAxis>
__device__ __forceinline__ int bound_reject_bits_planar(v4f s, const Pair& r) {
    f2 b = r.nod_p, nq = -s.w - r.qb_p;  // -q is what the chain computes (negations are free source modifiers)
    if constexpr (kAxis != 0) { b = fma2(f2{s.x, s.x}, r.dx, b); nq = fma2(f2{s.x, s.x}, -r.m2ox, nq); }
    if constexpr (kAxis != 1) { b = fma2(f2{s.y, s.y}, r.dy, b); nq = fma2(f2{s.y, s.y}, -r.m2oy, nq); }
    if constexpr (kAxis != 2) { b = fma2(f2{s.z, s.z}, r.dz, b); nq = fma2(f2{s.z, s.z}, -r.m2oz, nq); }
    const f2 disc = fma2(b, b, nq);
    const int ux = (__float_as_int(b.x) & __float_as_int(nq.x)) | __float_as_int(disc.x);
    const int uy = (__float_as_int(b.y) & __float_as_int(nq.y)) | __float_as_int(disc.y);
    return ux & uy;
}
// b and q of one sphere for both rays, DIRECT form.  static sphere: centre s.xyz, nr2 = -r^2
__device__ __forceinline__ void terms_static_direct(v4f s, float nr2, const Pair& r, f2& b, f2& q) {
    const f2 h = splat2(0.5f);
    const f2 ocx = fma2(r.m2ox, h, splat2(s.x)), ocy = fma2(r.m2oy, h, splat2(s.y)), ocz = fma2(r.m2oz, h, splat2(s.z));  // c - o'
    b = fma2(ocz, r.dz, fma2(ocy, r.dy, ocx * r.dx));
    q = fma2(ocz, ocz, fma2(ocy, ocy, fma2(ocx, ocx, f2{nr2, nr2})));
}
// moving sphere a = {c0, -r^2}, m = {dc, 0}: center(time) = c0 + time*dc (shape.rs:96-98)
// kAxes: bit k set when SOME moving sphere of the scene may move along axis k.  An axis no sphere moves along has
// dc = 0 for every sphere and time * 0 + x = x exactly, so its fma is left out: the scan is instantiated for "any
// axes" (7) and for "y only" (2: the reference's motion-blur scene bounces its spheres along y, config.rs:170-176;
// 12 packed ops per test instead of 14) and upload tells which one a scene may use (ScanLists::mov_axes).  One
// instantiation per single axis measured 1 % faster still on that scene but 2-3 % slower on the noise scenes, which
// never run this code (a side effect of the larger kernel on the layout of the rest; not instruction-cache misses:
// SQC_ICACHE_MISSES is 5e3 per 3e10 requests either way), so x-only and z-only motion take the general form.
template <int kAxes = 7>
__device__ __forceinline__ void terms_moving(v4f a, v4f m, const Pair& r, f2& b, f2& q) {
    constexpr int axes = kAxes;
    const f2 h = splat2(0.5f);
    f2 ocx = fma2(r.m2ox, h, splat2(a.x)), ocy = fma2(r.m2oy, h, splat2(a.y)), ocz = fma2(r.m2oz, h, splat2(a.z));
    if constexpr ((axes & 1) != 0) ocx = fma2(r.time, f2{m.x, m.x}, ocx);
    if constexpr ((axes & 2) != 0) ocy = fma2(r.time, f2{m.y, m.y}, ocy);
    if constexpr ((axes & 4) != 0) ocz = fma2(r.time, f2{m.z, m.z}, ocz);
    b = fma2(ocz, r.dz, fma2(ocy, r.dy, ocx * r.dx));
    q = fma2(ocz, ocz, fma2(ocy, ocy, fma2(ocx, ocx, f2{a.w, a.w})));
}
// Root extraction for one ray of one sphere whose line test passed somewhere in the wave.  Strict <: the
// first sphere in list order keeps a tie (hit.rs:100-110 narrows t_max with a half-open range; the lists are
// stably sorted, so coincident spheres keep their world order).
// kFloor (retrace after a rejected nominee, see hit_end): candidates at or before the rejected one in the scan's
// nomination order are skipped.  That order is (key, slot): `lo` is the rejected key and `ls` its slot, so a
// candidate passes when key > lo, or key == lo at a later slot (two spheres CAN produce the same fp32 key; strict <
// below nominates the first of them).  lo = -1 skips nothing (keys of valid roots are bit patterns of non-negative
// floats: non-negative as ints).  The scan is instantiated both ways and the floor costs nothing in passes where no
// lane of the wave retraces.
struct Floor {
    int lo_a, lo_b, ls_a, ls_b;
};
// The root: "near root if it is in range, else the far root" (shape.rs:53-61) with the range start folded into
// o': as unsigned bit patterns a negative float loses to any non-negative one, so the root taken is
// umin(bits(b - s), bits(b + s)) -- as take_root refines it.
// `second` is the runner-up KEY (no slot): the smallest key among the candidates that did not become `best`
// (v_med3_u32 keeps best <= second).  hit_end compares it with the exact (f64) depth of the accepted nominee: when
// some other candidate's key does not lie safely beyond it, the ray is traced again from that nominee on and the next
// candidate is decided in f64 as well, so near-ties are ordered by their f64 depths, not by fp32 keys.
// The float part (two roots and the median) is packed for the two rays of a lane; the integer part is per ray.
__device__ __forceinline__ f2 root_keys2(f2 b, f2 disc) {
    const f2 s = f2{__builtin_amdgcn_sqrtf(disc.x), __builtin_amdgcn_sqrtf(disc.y)};  // NaN for disc < 0
    // A key must be a LOWER bound of the exact depth (hit_end stops looking once the next key exceeds the exact best).
    // The near root b - s is one (the inflated direction and the widened radius both lower it); the far root b + s is
    // not -- the widened radius raises it, by far more than the keys' error bound when the ray runs tangent just inside
    // the surface -- so a ray that starts inside a sphere (near root negative, far root ahead) gets the key +0 for it:
    // the median of (b - s, b + s, z) with z = disc * 0 (+0, or -0 when disc < 0) is the near root when that is ahead,
    // +0 when only the far root is, and the (negative: loses as an unsigned pattern) far root when both are behind;
    // for disc < 0 the two roots are NaNs, v_med3_f32 then returns the minimum of its non-NaN operands: -0, which loses.
    const f2 k1 = b - s, k2 = b + s, z = disc * splat2(0.0f);
#ifdef RT_X_OLDKEY
    return f2{__uint_as_float(min(__float_as_uint(k1.x), __float_as_uint(k2.x))), __uint_as_float(min(__float_as_uint(k1.y), __float_as_uint(k2.y)))};
#else
    return f2{__builtin_amdgcn_fmed3f(k1.x, k2.x, z.x), __builtin_amdgcn_fmed3f(k1.y, k2.y, z.y)};
#endif
}
template <bool kFloor>
__device__ __forceinline__ void take_root(float key_f, int slot, int lo, int ls, uint32_t& best, uint32_t& second, int& idx) {
    uint32_t key = __float_as_uint(key_f);
    if (kFloor) key = ((int)key > lo - (slot > ls ? 1 : 0)) ? key : 0xFFFFFFFFu;  // at or before the floor: decided already
#ifndef RT_X_NOSECOND
    second = max(min(key, best), min(max(key, best), second));  // v_med3_u32
#endif
    const uint32_t nb = min(key, best);
    idx = nb != best ? slot : idx;  // strictly closer: the first of two equal keys stays (the other is `second`)
    best = nb;
}

#ifdef RT_STATS
__device__ unsigned long long g_dbg[8];  // 0 groups, 1 groups taken, 2 spheres taken, 3 ray-roots useful (disc>=0), 4 ray-roots in front
#define RT_STAT(i, n) dbg[i] += (n)
#else
#define RT_STAT(i, n)
#endif
// The root-extraction side of a group: second level, per sphere, again wave-uniform (about one sphere in four
// of a group that got here is met by some lane).  `filter` holds the discriminants that sent the wave here (the
// expanded pre-filter for the static list, the direct ones for the moving list); `direct(k, b, disc)` yields the
// direct-form b and disc of sphere k for both rays.
template <bool kFloor, class Direct>
__device__ __forceinline__ void roots_of_group(const f2 (&filter)[kGroup], Direct&& direct, int slot0, const Floor& fl,
                                               uint32_t& best_a, uint32_t& best_b, uint32_t& second_a, uint32_t& second_b,
                                               int& idx_a, int& idx_b, unsigned* dbg) {
#pragma unroll
    for (int k = 0; k < kGroup; ++k) {
        const int hit = max(__float_as_int(filter[k].x), __float_as_int(filter[k].y));
        if (RT_TAILCOND2) {
            f2 b, disc;
            direct(k, b, disc);
            RT_STAT(2, 1);
            RT_STAT(3, __popcll(__builtin_amdgcn_ballot_w64(disc.x >= 0)) + __popcll(__builtin_amdgcn_ballot_w64(disc.y >= 0)));
            const f2 key = root_keys2(b, disc);
            take_root<kFloor>(key.x, slot0 + k, fl.lo_a, fl.ls_a, best_a, second_a, idx_a);
            take_root<kFloor>(key.y, slot0 + k, fl.lo_b, fl.ls_b, best_b, second_b, idx_b);
        }
    }
}

// Scan both lists for both rays.  best[] enter as the bounds from the big-sphere test (or +inf bits).  On
// return idx[i] is the scan slot (static: [0, ns_pad), moving: ns_pad + [0, nm_pad)) of ray i's closest small
// sphere, or stays -1 when none beat the incoming bound.
//
// The packed arithmetic up to disc = b^2 - q (the reference's 17 / 23 flops, shape.rs:36-40,102-108) runs for
// every sphere; the un-packable root extraction (v_sqrt, v_bfi, subtract, compare, two selects: 12 instructions
// for the two rays) runs only for a GROUP of kGroup spheres in which some lane's line meets a sphere
// (disc >= 0 <=> sign bit clear; one v_max3_i32 per sphere folds the 2*kGroup sign bits, one compare and one
// wave-uniform branch per group decide).  Upload orders the lists along a Morton curve so the spheres of a group
// are neighbours and the groups a wave's rays meet are few.
template <bool kFloor, bool kPlanar>
__device__ __forceinline__ void scan_spheres(const ScanLists& S, const Pair& r, const Floor& fl, uint32_t& best_a,
                                             uint32_t& best_b, uint32_t& second_a, uint32_t& second_b, int& idx_a,
                                             int& idx_b) {
    unsigned dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const cv4p st = as_scalar(S.stat);
        auto group = [&](const v4f (&sp)[kGroup], int base) {
            f2 disc[kGroup];
            int any = (int)0x80000000;
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                disc[k] = disc_static_expanded(sp[k], r);
                any = max(max(any, __float_as_int(disc[k].x)), __float_as_int(disc[k].y));  // one v_max3_i32
            }
            RT_STAT(0, 1);
#ifdef RT_STATS
            {  // would a bounding sphere of the group (and of the 8 / 16 spheres starting at an aligned base) be met by any lane?
                auto met = [&](int n, const v4f* g) {
                    float cx = 0, cy = 0, cz = 0;
                    for (int k = 0; k < n; ++k) { cx += g[k].x / n; cy += g[k].y / n; cz += g[k].z / n; }
                    float R = 0;
                    for (int k = 0; k < n; ++k) {
                        const float dx = g[k].x - cx, dy = g[k].y - cy, dz = g[k].z - cz;
                        const float rr = sqrtf(fmaxf(0.f, g[k].x * g[k].x + g[k].y * g[k].y + g[k].z * g[k].z - g[k].w));
                        if (g[k].w < 1e30f) R = fmaxf(R, sqrtf(dx * dx + dy * dy + dz * dz) + rr);
                    }
                    bool any_lane = false;
                    for (int h = 0; h < 2; ++h) {
                        const float ox = -0.5f * r.m2ox[h] , oy = -0.5f * r.m2oy[h], oz = -0.5f * r.m2oz[h];
                        const float ex = cx - ox, ey = cy - oy, ez = cz - oz;
                        const float bb = ex * r.dx[h] + ey * r.dy[h] + ez * r.dz[h];
                        const float pp = ex * ex + ey * ey + ez * ez - bb * bb;
                        const bool inside = ex * ex + ey * ey + ez * ez < R * R;
                        any_lane |= (pp <= R * R && (bb > 0 || inside)) && (r.dx[h] != 0.f || r.dy[h] != 0.f || r.dz[h] != 0.f);
                    }
                    return __builtin_amdgcn_ballot_w64(any_lane) != 0;
                };
                RT_STAT(4, met(4, sp));
            }
#endif
            if (RT_TAILCOND) {
                RT_STAT(1, 1);
                const v4f nr2 = *(cv4p)(uintptr_t)(S.stat_nr2 + base);  // -r^2 of the group: one scalar load, this side only
                roots_of_group<kFloor>(disc, [&](int k, f2& b, f2& d) {
                    f2 q;
                    terms_static_direct(sp[k], nr2[k], r, b, q);
                    d = fma2(b, b, -q);
                }, base, fl, best_a, best_b, second_a, second_b, idx_a, idx_b, dbg);
            }
        };
    // HIERARCHICAL CULLING.  Upload stores a bounding sphere around each group of kGroup list neighbours (Morton
    // order makes them spatial neighbours) and around every run of 16, 64, 256 and 1024 spheres.  A bound goes
    // through the expanded test (plus "not wholly behind": bound_reject_bits), and what it covers is fetched and
    // tested only when some lane's ray can meet it.  The walk is wave-uniform (scalar control flow): at a chunk
    // index aligned to a level's span that level's bound is tested first and a rejection skips the whole span, so
    // the cost is sub-linear in the sphere count wherever the rays of a wave leave most of the scene aside
    // (measured on configs[1]: 34 % of the group bounds pass; a wave's 128 rays leave from a few hit points in all
    // directions, so only distant groups are skipped).  One s_load_dwordx16 brings the bounds of a chunk's 4 groups.
    static_assert(kGroup == 4 && kChunk == 16, "one s_load_dwordx16 = the bounds of 4 groups = one chunk");
    auto met3 = [&](v4f bound) {
        RT_STAT(5, 1);
        return __builtin_amdgcn_ballot_w64(bound_reject_bits(bound, r) >= 0) != 0ull;
    };
    // the planar list (see ScanLists::plan), instantiated per axis of the shared coordinate
    const cv4p pl = as_scalar(S.plan);
    auto planar_walk = [&](auto axis_tag, auto&& walk_fn) {
        constexpr int kAxis = decltype(axis_tag)::value;
        auto metp = [&](v4f bound) {
            RT_STAT(5, 1);
            return __builtin_amdgcn_ballot_w64(bound_reject_bits_planar<kAxis>(bound, r) >= 0) != 0ull;
        };
        walk_fn(as_scalar(S.plan_bound), S.np_pad / kChunk, S.plan_levels, true, metp, [&](int g) {
            const int base = g * kGroup;
            v4f sp[kGroup];
            f2 disc[kGroup];
            int any = (int)0x80000000;
#pragma unroll
            for (int k = 0; k < kGroup; ++k) sp[k] = pl[base + k];
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                disc[k] = disc_planar_expanded<kAxis>(sp[k], r);
                any = max(max(any, __float_as_int(disc[k].x)), __float_as_int(disc[k].y));
            }
            RT_STAT(0, 1);
            if (RT_TAILCOND) {
                RT_STAT(1, 1);
                const v4f nr2 = *(cv4p)(uintptr_t)(S.plan_nr2 + base);
                roots_of_group<kFloor>(disc, [&](int k, f2& b, f2& d) {
                    f2 q;
                    terms_static_direct(sp[k], nr2[k], r, b, q);
                    d = fma2(b, b, -q);
                }, S.ns_pad + base, fl, best_a, best_b, second_a, second_b, idx_a, idx_b, dbg);
            }
        });
    };
    auto chunk = [&](cv4p bd, int c, bool cull, auto&& met, auto&& members) {  // the 4 groups of chunk c
        v4f bound[kChunk / kGroup];
#pragma unroll
        for (int g = 0; g < kChunk / kGroup; ++g) bound[g] = bd[c * (kChunk / kGroup) + g];
#pragma unroll
        for (int g = 0; g < kChunk / kGroup; ++g)
            if (!cull || met(bound[g])) { RT_STAT(6, 1); members(c * (kChunk / kGroup) + g); }
    };
    auto walk = [&](cv4p bd, int n_chunks, int levels, bool cull, auto&& met, auto&& members) {
        if (!cull || levels == 0) {  // small scenes: the group bounds only, no bookkeeping in the loop
            for (int c = 0; c < n_chunks; ++c) chunk(bd, c, cull, met, members);
            return;
        }
        int off[kLevels];
#pragma unroll
        for (int L = 0; L < kLevels; ++L) off[L] = bound_level_offset(n_chunks, L);
        for (int c = 0; c < n_chunks;) {
            bool skipped = false;
#pragma unroll
            for (int L = kLevels - 1; L >= 1; --L) {
                const int span = 1 << (2 * (L - 1));  // chunks under a node of this level
                if (!skipped && ((levels >> L) & 1) && (c & (span - 1)) == 0 && !met(bd[off[L] + (c >> (2 * (L - 1)))])) {
                    c += span;
                    skipped = true;
                }
            }
            if (skipped) continue;
            chunk(bd, c, true, met, members);
            ++c;
        }
    };
#ifndef RT_NOSCAN
    walk(as_scalar(S.stat_bound), S.ns_pad / kChunk, S.stat_levels, true, met3, [&](int g) {
        v4f sp[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; ++k) sp[k] = st[g * kGroup + k];
        group(sp, g * kGroup);
    });
    if (kPlanar && S.np_pad > 0) {  // wave-uniform
        if (S.plan_axis == 0) planar_walk(std::integral_constant<int, 0>{}, walk);
        else if (S.plan_axis == 1) planar_walk(std::integral_constant<int, 1>{}, walk);
        else planar_walk(std::integral_constant<int, 2>{}, walk);
    }
#endif
    // The moving list: a bound covers its members' whole motion for time in [0, 1] (the segments c0 .. c0 + dc).  A
    // ray outside that time range (the fixed-batch entry point takes any) switches the culling off for its wave.
    const cv4p ma = as_scalar(S.mov_a), mb = as_scalar(S.mov_b);
    const bool cull_moving = S.nm_pad > 0 && __builtin_amdgcn_ballot_w64(!(r.time.x >= 0.0f && r.time.x <= 1.0f) ||
                                                                        !(r.time.y >= 0.0f && r.time.y <= 1.0f)) == 0ull;
    auto moving_members = [&](auto axes_tag) {
        constexpr int kAxes = decltype(axes_tag)::value;
        walk(as_scalar(S.mov_bound), S.nm_pad / kChunk, S.mov_levels, cull_moving, met3, [&](int g) {
            const int base = g * kGroup;
            f2 b[kGroup], q[kGroup], disc[kGroup];
            int any = (int)0x80000000;
            v4f dy = {0, 0, 0, 0};
            if constexpr (kAxes == 2) dy = *(cv4p)(uintptr_t)(S.mov_dy + base);
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                if constexpr (kAxes == 2) terms_moving<kAxes>(ma[base + k], v4f{0.0f, dy[k], 0.0f, 0.0f}, r, b[k], q[k]);
                else terms_moving<kAxes>(ma[base + k], mb[base + k], r, b[k], q[k]);
                disc[k] = fma2(b[k], b[k], -q[k]);
                any = max(max(any, __float_as_int(disc[k].x)), __float_as_int(disc[k].y));
            }
            RT_STAT(0, 1);
            if (RT_TAILCOND) {
                RT_STAT(1, 1);
                roots_of_group<kFloor>(disc, [&](int k, f2& bb, f2& d) { bb = b[k]; d = disc[k]; }, S.ns_pad + S.np_pad + base, fl,
                                       best_a, best_b, second_a, second_b, idx_a, idx_b, dbg);
            }
        });
    };
    if (S.mov_axes == 2) moving_members(std::integral_constant<int, 2>{});  // wave-uniform
    else moving_members(std::integral_constant<int, 7>{});
#ifdef RT_STATS
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_dbg[i], (unsigned long long)dbg[i]);
#endif
    (void)dbg;
}

// ------------------------------------------------------------------ the f64 sphere test
// shape.rs:35-65 / :101-133 verbatim in f64 on the fp32 inputs, unnormalised direction:
// near root if t_min <= t < t_max else far root.  The fp32 scan only NOMINATES a sphere; this f64 test DECIDES
// it, so a grazing or far-origin false positive of the scan is rejected here.
// f64 reciprocal and square root without the library's scaling and special cases (12 and 22 VALU instructions as
// compiled from 1.0 / a and sqrt(x)): the arguments here are |d|^2 of an fp32 direction and a discriminant of fp32
// inputs, normal f64 numbers or exactly 0.  v_rcp_f64 / v_rsq_f64 seed two Newton steps (rcp: <= 1 ulp) or one
// Goldschmidt step and one correction (sqrt: ~1e-16 relative; the result is narrowed to fp32 or compared with
// another root, nothing depends on its last bit).
__device__ __forceinline__ double rcp_f64(double a) {
    double y = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-a, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double sqrt_f64(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? 0.0 : g;  // rsq(0) = inf
}
// centre at the ray's time, shape.rs:96-98 (dc = 0 for a static sphere), from one 16-byte record {c0, .} (+ {dc, .})
template <class SceneT>
__device__ __forceinline__ void sphere_centre_f64(const SceneT& S, const float4* c0, const float4* dc, int i, float time,
                                                  double& cx, double& cy, double& cz, float& w) {
    const v4f a = *reinterpret_cast<const v4f*>(c0 + i);
    cx = (double)a.x; cy = (double)a.y; cz = (double)a.z; w = a.w;
    if (S.any_moving) {  // wave-uniform
        const v4f m = *reinterpret_cast<const v4f*>(dc + i);
        const double tm = (double)time;
        cx += tm * (double)m.x; cy += tm * (double)m.y; cz += tm * (double)m.z;
    }
}
__device__ __forceinline__ bool sphere_roots_f64(double cx, double cy, double cz, double r2, const Ray& r, double a,
                                                 double inv_a, double t_min, double t_max, double& t_out) {
    const double ocx = (double)r.o.x - cx, ocy = (double)r.o.y - cy, ocz = (double)r.o.z - cz;
    const double dx = r.d.x, dy = r.d.y, dz = r.d.z;
    const double hb = ocx * dx + ocy * dy + ocz * dz;
    const double c = ocx * ocx + ocy * ocy + ocz * ocz - r2;
    const double disc = hb * hb - a * c;
    if (!(disc >= 0.0)) return false;  // shape.rs:41
    const double root = sqrt_f64(disc);
    double t = (-hb - root) * inv_a;  // one f64 reciprocal per ray instead of a division per root
    bool ok = (t >= t_min) && (t < t_max);
    if (!ok) {
        t = (-hb + root) * inv_a;
        ok = (t >= t_min) && (t < t_max);
    }
    if (!ok) return false;
    t_out = t;
    return true;
}
// HitList::hit keeps the FIRST object in world order among those hit at exactly the same t (hit.rs:100-110: the
// range is narrowed to the closest t so far and is half-open).  The device meets candidates in another order (big
// spheres, then fp32-key order), so "closer" is (t, id) lexicographic.
__device__ __forceinline__ bool closer_hit(double t, int id, double t_best, int id_best) {
    return t < t_best || (t == t_best && id < id_best);
}
// Per-ray state kept across the packed scan: the f64 big-sphere result before it, the f64 hit record after it.
// Kept small (7 VGPRs per ray): everything else is recomputed after the scan from the ray and the winner's id.
struct HitWork {
    double t_best;
    double inv_a;  // 1 / |d|^2, f64
    int id_best;
    float key_scale;  // |d| * kScanInflate: t -> key units (see hit_key)
};
// The scan is CONSERVATIVE and the f64 test decides: the scan direction is the unit direction times (1 + 8 ulp),
// which inflates b^2 by 16 ulp -- more than the worst-case rounding of disc = b^2 - (|oc|^2 - r^2), ~15 ulp of
// |oc|^2 (three roundings in b squared, three in q, one in disc, the normalisation of d), and b^2 ~ |oc|^2 for
// every sphere the line passes near.  So a sphere the exact line meets always has disc >= 0 in fp32, whatever
// |oc| / r is (a radius-0.1 sphere 300 away is hit or missed at random by a plain fp32 discriminant); the price is
// a band of false nominations, which the f64 test of the nominee rejects (HitRec::kRetry below).
constexpr float kScanInflate = 1.0f + 8.0f / 16777216.0f;
// How far a KEY can lie above the exact depth of its sphere, in key units (distance along the ray from the scan's
// start).  The inflation biases every near-root key downwards (d(b - s)/db = -(b - s)/s < 0 outside the sphere), a
// far-root key (origin inside) upwards by <= 16 ulp; what is left is ABSOLUTE: centres and origin are rounded into
// the scan frame, 3 half-ulps of the frame coordinates per component of oc, which b and s inherit: <= 16 ulp of
// (|o'| + |c|).  For an origin farther than 3 C from the frame's origin (C = max(|c| + r) over the lists) every
// depth exceeds 2/3 |o'| and the bound is < 32 ulp of the depth; nearer, it is < 64 ulp of C.  So
//   key <= depth * (1 + kKeyRel) + kKeyEps0 * C        (with room for the roundings of hit_key itself)
// needs no per-ray term: ScanLists::key_eps0 = kKeyEps0 * C is a scene constant.
constexpr float kKeyRel = 128.0f / 16777216.0f;
// The scan does not start at t_min but EARLIER, by kStartShift of t_min plus, along the ray, the keys' absolute
// error near the scene (half of ScanLists::key_eps0; a distance, so key_eps0 / 2|d| in units of t): a root at or just above t_min
// (which the reference takes, shape.rs:54 `t_min <= t`) must not come out of the fp32 arithmetic with a negative key
// and be lost.  The distance term is what protects rays with a tiny direction (a Lambertian n + S that nearly
// cancels: |d| ~ 1e-3 stretches t_min = 1e-3 to a distance of 1e-6, below the keys' resolution).  Roots inside the
// shifted-in sliver are nominated, rejected by the f64 test (which uses the true t_min) and passed over (kRetry).
constexpr float kStartShift = 1.0f / 32.0f;
__host__ __device__ __forceinline__ float scan_start(float t_min) { return t_min - t_min * kStartShift; }
// The f64 constants of the closest-hit steps.  The render kernel gets them ready-made in its kernarg segment and reads
// them where they are used (scalar loads -> SGPR pairs, used straight as f64 operands): left to the compiler, the
// conversions (double)t_min, (double)origin ... are hoisted out of the persistent loop into ten VGPRs that stay live
// across all of it (measured: the difference between 2 and 29 spilled VGPRs).
struct HitConsts {
    double t_min, t_start;  // the reference's range start (main.rs:116) and where the scan starts
    double ox, oy, oz;      // ScanLists::ox, oy, oz
    float key_eps0;         // ScanLists::key_eps0
    float t_start_f;        // t_start as computed in fp32 (scan_start)
};
__host__ __device__ inline HitConsts make_hit_consts(const ScanLists& L, float t_min) {
    return {(double)t_min, (double)scan_start(t_min), (double)L.ox, (double)L.oy, (double)L.oz, L.key_eps0, scan_start(t_min)};
}
__device__ __forceinline__ double dir_len2(const Ray& r) {
    return (double)r.d.x * (double)r.d.x + (double)r.d.y * (double)r.d.y + (double)r.d.z * (double)r.d.z;
}
// The exact best depth so far as an UPPER bound in the scan's key units: a candidate whose key is not below it
// cannot be closer than that hit.
__device__ __forceinline__ uint32_t hit_key(const HitWork& w, const HitConsts& C) {
    if (w.id_best < 0) return kInfBits;
    // distance from the scan's start: (t - t_start) |d| + the distance part of the shift (x (1 + 32 ulp): the
    // inflation, squared, and the roundings of closest_begin's t_start - 2 key_eps0 / |d|)
    const float tb = fmaf((float)(w.t_best - C.t_start), w.key_scale, C.key_eps0 * (0.5f + 16.0f / 16777216.0f));  // >= 0: t_best >= t_min > t_start
    return __float_as_uint(fmaf(tb, 1.0f + kKeyRel, C.key_eps0));
}
// The hit record of the winner (shape.rs:45-50, hit.rs:45-61) in f64, then narrowed; a miss is all zeros.
template <class SceneT>
__device__ __forceinline__ HitRec hit_record(const SceneT& S, const Ray& r, const HitWork& w) {
    HitRec h;
    h.sphere = w.id_best;
    if (w.id_best < 0) {
        h.front = 0; h.t = 0.0f; h.p = {0, 0, 0}; h.n = {0, 0, 0};
        return h;
    }
    const int id = w.id_best;
    double cx, cy, cz;
    float inv_rad;  // 1/r rounded to fp32: |n| = 1 +- 6e-8
    sphere_centre_f64(S, S.rec_a, S.rec_dc, id, r.time, cx, cy, cz, inv_rad);
    const double px = (double)r.o.x + w.t_best * (double)r.d.x;
    const double py = (double)r.o.y + w.t_best * (double)r.d.y;
    const double pz = (double)r.o.z + w.t_best * (double)r.d.z;
    const double ir = (double)inv_rad;
    double nx = (px - cx) * ir, ny = (py - cy) * ir, nz = (pz - cz) * ir;
    const bool front = ((double)r.d.x * nx + (double)r.d.y * ny + (double)r.d.z * nz) < 0.0;
    if (!front) { nx = -nx; ny = -ny; nz = -nz; }
    h.front = front ? 1 : 0;
    h.t = (float)w.t_best;
    h.p = {(float)px, (float)py, (float)pz};
    h.n = {(float)nx, (float)ny, (float)nz};
    return h;
}
// HitRec::sphere == kRetry: the trace of this ray segment is NOT finished.  Either the scan's nominee was a false
// positive of the conservative fp32 test, or it was accepted but another candidate's key (Hit2::second) lies below
// its exact depth's bound, so that candidate must be decided in f64 too.  The ray is traced AGAIN with the nominee's
// (key, slot) as the floor of the scan -- candidates at or before it are skipped -- and with the exact best so far
// carried along, until no undecided candidate can be closer.  Near-ties are therefore ordered by f64 depths and exact
// ties by world order, as the reference does; the fp32 keys only decide in which order candidates are looked at.
// The record of an unfinished trace: sphere = kRetry, t = the floor's key bits, front = the floor's slot, and the exact
// best so far (a scan candidate accepted by an earlier pass; a big sphere's hit is found again by every pass anyway) in
// the fields a finished record uses for the hit point: p.x = its sphere id (bits; -1: none), p.y / p.z = its f64 depth.
constexpr int kRetry = -2;
__device__ __forceinline__ int retry_key(const HitRec& h) { return h.sphere == kRetry ? __float_as_int(h.t) : -1; }
__device__ __forceinline__ int retry_slot(const HitRec& h) { return h.sphere == kRetry ? h.front : 0x7FFFFFFF; }
__device__ __forceinline__ void merge_carry(HitWork& w, const HitRec& h) {
    if (h.sphere == kRetry) {
        const int id = __float_as_int(h.p.x);
        const double t = __hiloint2double(__float_as_int(h.p.z), __float_as_int(h.p.y));
        if (id >= 0 && closer_hit(t, id, w.t_best, w.id_best)) { w.t_best = t; w.id_best = id; }
    }
}
template <class SceneT>
__device__ __forceinline__ HitRec hit_end(const SceneT& S, const Ray& r, const HitConsts& C, HitWork& w, uint32_t best,
                                          uint32_t second, int slot) {
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    HitRec h;
    if (slot >= 0 && best < kInfBits) {  // the scan's nominee, decided in f64
        const int id = S.scan_id[slot];
        double cx, cy, cz, t;
        float rad;
        sphere_centre_f64(S, S.slot_a, S.slot_dc, slot, r.time, cx, cy, cz, rad);
        if (id >= 0 && sphere_roots_f64(cx, cy, cz, (double)rad * (double)rad, r, dir_len2(r), w.inv_a, C.t_min, inf, t) &&
            closer_hit(t, id, w.t_best, w.id_best)) {
            w.t_best = t; w.id_best = id;
        }
        // every candidate not decided yet has a key >= second: final unless that may still be closer than the best
#ifdef RT_X_NORETRY  // tolerance-sensitivity experiment (tools/tolerance_sensitivity.py): round 1's behaviour, a rejected nominee is final
        if (false) {
#else
        if (second < hit_key(w, C)) {
#endif
            h.sphere = kRetry; h.front = slot; h.t = __uint_as_float(best);
            h.p = {__int_as_float(w.id_best), __int_as_float(__double2loint(w.t_best)), __int_as_float(__double2hiint(w.t_best))};
            h.n = {0, 0, 0};
            return h;
        }
    }
    return hit_record(S, r, w);
}

// The spheres of the f64 list against the two rays of a lane, world order: one scalar load per sphere serves both.
// `tie_by_id`: a hit of a LATER sphere may already be in w (carried over by a retrace), then (t, id) decides.
template <class SceneT>
__device__ __forceinline__ void big_spheres_f64(const SceneT& S, const Ray& ra, const Ray& rb, bool do_a, bool do_b,
                                                const HitConsts& C, bool tie_by_id, double aa, double ab, HitWork& wa,
                                                HitWork& wb) {
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    typedef double d8 __attribute__((ext_vector_type(8)));
    const RT_AS4 d8* big = (const RT_AS4 d8*)(uintptr_t)S.big;
    const RT_AS4 int32_t* big_id = (const RT_AS4 int32_t*)(uintptr_t)S.big_id;
    const int n_big = S.n_big;
    const bool moving = S.any_moving;
    for (int k = 0; k < n_big; ++k) {
        const d8 B = big[k];  // {cx, cy, cz, r2, dcx, dcy, dcz, inv_r}
        const int id = big_id[k];
        auto test = [&](const Ray& r, double a, HitWork& w) {
            double cx = B[0], cy = B[1], cz = B[2], t;
            if (moving) { const double tm = (double)r.time; cx += tm * B[4]; cy += tm * B[5]; cz += tm * B[6]; }
            // world order: strictly closer (hit.rs:100-110)
            if (sphere_roots_f64(cx, cy, cz, B[3], r, a, w.inv_a, C.t_min, inf, t) &&
                (t < w.t_best || (tie_by_id && closer_hit(t, id, w.t_best, w.id_best)))) {
                w.t_best = t; w.id_best = id;
            }
        };
        if (do_a) test(ra, aa, wa);
        if (do_b) test(rb, ab, wb);
    }
}
// A scene WITHOUT scan lists (upload sends every sphere of a scene of at most kDirectMax spheres besides the big
// ones to the f64 list: the reference's noise scenes are a ground and one ball): HitList::hit is the f64 loop and the
// hit record, nothing else -- no scan frame, no nomination, no decision, no retrace.
template <class SceneT>
__device__ __forceinline__ void closest_direct(const SceneT& S, const Ray& ra, const Ray& rb, bool do_a, bool do_b,
                                               const HitConsts& C, HitRec& ha, HitRec& hb) {
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    HitWork wa, wb;
    wa.id_best = wb.id_best = -1;
    wa.t_best = wb.t_best = inf;
    wa.key_scale = wb.key_scale = 0.0f;
    const double aa = dir_len2(ra), ab = dir_len2(rb);
    wa.inv_a = rcp_f64(aa);
    wb.inv_a = rcp_f64(ab);
    big_spheres_f64(S, ra, rb, do_a, do_b, C, false, aa, ab, wa, wb);
    ha.sphere = hb.sphere = -1;
    if (do_a) ha = hit_record(S, ra, wa);
    if (do_b) hb = hit_record(S, rb, wb);
}
// HitList::hit(ray, t_min..inf) (hit.rs:100-110) for the TWO rays of a lane: closest hit over the whole world, in
// three steps so that a caller can re-read its scene arguments between them (render_kernel):
//   closest_begin: f64 big-sphere pass + the scan frame;  scan_spheres;  closest_end: f64 decision + hit records.
// do_a / do_b: whether that ray is live (a dead one still rides through the packed scan, its result is dropped).
struct Hit2 {
    HitWork wa, wb;
    Pair p;
    uint32_t best_a, best_b;      // closest key so far (enters the scan as the bound from the f64 results so far)
    uint32_t second_a, second_b;  // runner-up key (take_root)
    int slot_a, slot_b;
};
// the scan with or without the floor, chosen per wave
template <bool kPlanar>
__device__ __forceinline__ void scan_with_floor(const ScanLists& L, const Floor& fl, Hit2& x) {
#ifdef RT_X_ALWAYSFLOOR  // what a pass with a retracing lane costs: every pass takes the floor variant
    if (true)
#else
    if (__ballot(fl.lo_a >= 0 || fl.lo_b >= 0) != 0ull)
#endif
        scan_spheres<true, kPlanar>(L, x.p, fl, x.best_a, x.best_b, x.second_a, x.second_b, x.slot_a, x.slot_b);
    else
        scan_spheres<false, kPlanar>(L, x.p, fl, x.best_a, x.best_b, x.second_a, x.second_b, x.slot_a, x.slot_b);
}
// prev_a / prev_b: the rays' records of the previous trace; `carry` (wave-uniform) says whether any of them may be a
// kRetry record whose exact best so far has to be taken over (passes without a retracing lane skip that code).
template <bool kPlanar, class SceneT>
__device__ __forceinline__ void closest_begin(const SceneT& S, const Ray& ra, const Ray& rb, bool do_a, bool do_b,
                                              const HitConsts& C, bool carry, const HitRec& prev_a, const HitRec& prev_b,
                                              Hit2& x) {
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    x.best_a = x.best_b = kInfBits;
    x.second_a = x.second_b = 0xFFFFFFFFu;
    x.wa.id_best = x.wb.id_best = -1;
    x.wa.t_best = x.wb.t_best = inf;
    const double aa = dir_len2(ra), ab = dir_len2(rb);
    x.wa.inv_a = rcp_f64(aa);
    x.wb.inv_a = rcp_f64(ab);
    if (carry) {  // a retracing lane's best candidate so far, decided again
        RT_MARK("carry");
        if (do_a) merge_carry(x.wa, prev_a);
        if (do_b) merge_carry(x.wb, prev_b);
    }
    RT_MARK("begin");
    big_spheres_f64(S, ra, rb, do_a, do_b, C, carry, aa, ab, x.wa, x.wb);
    const float la2 = (float)aa, lb2 = (float)ab;
    const float inv_la = fast_rsq(la2) * kScanInflate, inv_lb = fast_rsq(lb2) * kScanInflate;
    x.wa.key_scale = la2 * inv_la;
    x.wb.key_scale = lb2 * inv_lb;
    Pair& p = x.p;
    p.dx = f2{ra.d.x * inv_la, rb.d.x * inv_lb};
    p.dy = f2{ra.d.y * inv_la, rb.d.y * inv_lb};
    p.dz = f2{ra.d.z * inv_la, rb.d.z * inv_lb};
    // o' = o + t_start_ray*d - origin in f64, rounded once: its error is an ulp of the SCAN-FRAME coordinate
    const auto& L = S.lists;
#ifdef RT_X_NOABSSHIFT
    const double tsa = C.t_start, tsb = C.t_start;
#else
    const double tsa = (double)fmaf(-0.5f * C.key_eps0, inv_la, C.t_start_f), tsb = (double)fmaf(-0.5f * C.key_eps0, inv_lb, C.t_start_f);
#endif
    auto frame = [&](float o, float d, double ts, double org) { return (float)(((double)o + ts * (double)d) - org); };
    f2 ox = f2{frame(ra.o.x, ra.d.x, tsa, C.ox), frame(rb.o.x, rb.d.x, tsb, C.ox)};
    f2 oy = f2{frame(ra.o.y, ra.d.y, tsa, C.oy), frame(rb.o.y, rb.d.y, tsb, C.oy)};
    f2 oz = f2{frame(ra.o.z, ra.d.z, tsa, C.oz), frame(rb.o.z, rb.d.z, tsb, C.oz)};
    p.time = f2{ra.time, rb.time};
    // a dead ray rides through the packed scan: park it (d = 0, far origin => disc = -|oc|^2 - ... < 0 for every
    // sphere, in either form) so it never sends its wave into a root-extraction group
    if (!do_a) { p.dx.x = p.dy.x = p.dz.x = 0.0f; ox.x = 1e18f; oy.x = oz.x = 0.0f; p.time.x = 0.0f; }
    if (!do_b) { p.dx.y = p.dy.y = p.dz.y = 0.0f; ox.y = 1e18f; oy.y = oz.y = 0.0f; p.time.y = 0.0f; }
    // the per-ray terms of the expanded static test
    p.m2ox = ox * -2.0f; p.m2oy = oy * -2.0f; p.m2oz = oz * -2.0f;
    p.nod = -fma2(oz, p.dz, fma2(oy, p.dy, ox * p.dx));
    const f2 oo = fma2(oz, oz, fma2(oy, oy, ox * ox));
    const f2 m = f2{fast_sqrt(oo.x), fast_sqrt(oo.y)} + L.cmax;
    p.qb = fma2(m * m, splat2(-kExpandErr), oo);
    // the bound a ray brings into the scan: its exact best so far (big spheres, carried candidates)
    if (do_a) x.best_a = hit_key(x.wa, C);
    if (do_b) x.best_b = hit_key(x.wb, C);
    if (kPlanar) {  // the planar list's shared coordinate, folded (wave-uniform selects)
        const int ax = L.plan_axis;
        const f2 dk = ax == 0 ? p.dx : ax == 1 ? p.dy : p.dz, mk = ax == 0 ? p.m2ox : ax == 1 ? p.m2oy : p.m2oz;
        p.nod_p = fma2(splat2(L.plan_val), dk, p.nod);
        p.qb_p = fma2(splat2(L.plan_val), mk, p.qb);
    }
    x.slot_a = x.slot_b = -1;
}
template <class SceneT>
__device__ __forceinline__ void closest_end(const SceneT& S, const Ray& ra, const Ray& rb, bool do_a, bool do_b,
                                            const HitConsts& C, Hit2& x, HitRec& ha, HitRec& hb) {
    ha.sphere = -1;
    hb.sphere = -1;
    if (do_a) ha = hit_end(S, ra, C, x.wa, x.best_a, x.second_a, x.slot_a);
    if (do_b) hb = hit_end(S, rb, C, x.wb, x.best_b, x.second_b, x.slot_b);
}
// one-shot form for the batch kernel: retraces on the spot while any lane of the wave holds an unfinished trace
__device__ __forceinline__ void closest_hit2(const DevScene& S, const Ray& ra, const Ray& rb, bool do_a, bool do_b,
                                             float t_min, HitRec& ha, HitRec& hb) {
    ha.sphere = hb.sphere = -1;
    Floor fl = {-1, -1, 0x7FFFFFFF, 0x7FFFFFFF};
    const HitConsts C = make_hit_consts(S.lists, t_min);
    if (S.lists.ns_pad + S.lists.np_pad + S.lists.nm_pad == 0) {  // (wave-uniform) the render kernel's direct variant
        closest_direct(S, ra, rb, do_a, do_b, C, ha, hb);
        return;
    }
    while (__ballot(do_a || do_b) != 0ull) {  // wave-uniform: every lane runs the packed scan together
        Hit2 x;
        HitRec na, nb;
        closest_begin<true>(S, ra, rb, do_a, do_b, C, true, ha, hb, x);
        scan_with_floor<true>(S.lists, fl, x);
        closest_end(S, ra, rb, do_a, do_b, C, x, na, nb);
        if (do_a) { ha = na; fl.lo_a = retry_key(na); fl.ls_a = retry_slot(na); do_a = na.sphere == kRetry; }
        if (do_b) { hb = nb; fl.lo_b = retry_key(nb); fl.ls_b = retry_slot(nb); do_b = nb.sphere == kRetry; }
    }
}

// ------------------------------------------------------------------ textures
// Noise is evaluated for the TWO path slots of a lane at once, packed {slot 0, slot 1} like the scan: the float
// arithmetic (frequency scale, fractions, quintic fades, the seven lerps, octave sums) is v_pk_*_f32, and the 14
// dependent LDS lookups of the two slots are issued side by side, so their latencies overlap.  The texture is
// wave-uniform inside noise_eval2 (callers loop over the distinct textures of a wave): its parameters are scalars.
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 smooth2(f2 t) {  // lib.rs:50: t^3 (t (6t - 15) + 10)
    const f2 a = fma2(t, fma2(t, splat(6.0f), splat(-15.0f)), splat(10.0f));
    return (t * t * t) * a;
}
__device__ __forceinline__ f2 lerp2(f2 t, f2 lo, f2 hi) { return fma2(t, hi - lo, lo); }  // lib.rs:46

// the eight corner terms of one slot: hash chain (material.rs:309-313) + fused corner entry
__device__ __forceinline__ void noise_corners(const FusedNoise& tb, float fx, float fy, float fz, float tx, float ty,
                                              float tz, float (&c)[8]) {
    const int x0 = (int)fx & 255, y0 = (int)fy & 255, z0 = (int)fz & 255;  // material.rs:330-335
    const int x1 = (x0 + 1) & 255, y1 = (y0 + 1) & 255, z1 = (z0 + 1) & 255;
    const int a0 = tb.perms[x0], a1 = tb.perms[x1];
    const int b00 = tb.perms[a0 + y0], b10 = tb.perms[a1 + y0], b01 = tb.perms[a0 + y1], b11 = tb.perms[a1 + y1];
    const float ux = tx - 1.0f, uy = ty - 1.0f, uz = tz - 1.0f;  // material.rs:478-479
    auto corner = [&](int i, float x, float y, float z) {
        const float4 g = tb.g[i];
        return fmaf(g.z, z, fmaf(g.y, y, fmaf(g.x, x, g.w)));
    };
    c[0] = corner(b00 + z0, tx, ty, tz); c[1] = corner(b10 + z0, ux, ty, tz);   // 000 100
    c[2] = corner(b01 + z0, tx, uy, tz); c[3] = corner(b11 + z0, ux, uy, tz);   // 010 110
    c[4] = corner(b00 + z1, tx, ty, uz); c[5] = corner(b10 + z1, ux, ty, uz);   // 001 101
    c[6] = corner(b01 + z1, tx, uy, uz); c[7] = corner(b11 + z1, ux, uy, uz);   // 011 111
}
// noise(p) of material.rs:322-361 (value) / :466-521 (Perlin) for both slots; `perlin` is wave-uniform
__device__ __forceinline__ f2 noise_base2(const FusedNoise& tb, float freq, bool perlin, f2 px, f2 py, f2 pz) {
    px = px * freq; py = py * freq; pz = pz * freq;
    const f2 fx = {floorf(px.x), floorf(px.y)}, fy = {floorf(py.x), floorf(py.y)}, fz = {floorf(pz.x), floorf(pz.y)};
    const f2 tx = px - fx, ty = py - fy, tz = pz - fz;
    float c0[8], c1[8];
    noise_corners(tb, fx.x, fy.x, fz.x, tx.x, ty.x, tz.x, c0);
    noise_corners(tb, fx.y, fy.y, fz.y, tx.y, ty.y, tz.y, c1);
    const f2 sx = smooth2(tx), sy = smooth2(ty), sz = smooth2(tz);
    const f2 x00 = lerp2(sx, f2{c0[0], c1[0]}, f2{c0[1], c1[1]}), x10 = lerp2(sx, f2{c0[2], c1[2]}, f2{c0[3], c1[3]});
    const f2 x01 = lerp2(sx, f2{c0[4], c1[4]}, f2{c0[5], c1[5]}), x11 = lerp2(sx, f2{c0[6], c1[6]}, f2{c0[7], c1[7]});
    const f2 n = lerp2(sz, lerp2(sy, x00, x10), lerp2(sy, x01, x11));
    return perlin ? fma2(n, splat(0.5f), splat(0.5f)) : n;  // material.rs:520: (noise + 1) / 2
}
// eval(p): the NoiseAdapter callbacks as data (material.rs:184-263); T is wave-uniform
template <class TexT>
__device__ __forceinline__ f2 noise_eval2(const TexT& T, const FusedNoise* lds_fused, f2 px, f2 py, f2 pz) {
    const FusedNoise& tb = lds_fused[T.fused_id];
    const bool perlin = T.t.kind == RT_TEX_PERLIN_NOISE;
    const float freq = T.t.freq;
    f2 v;
    if (T.t.op == RT_NOISE_PLAIN) {
        v = noise_base2(tb, freq, perlin, px, py, pz);
    } else {  // material.rs:192-203, :219-233
        const bool turb = T.t.op == RT_NOISE_TURBULENCE;
        const float lac = T.t.lacunarity, gain = T.t.gain;
        f2 sum = splat(0.0f), qx = px, qy = py, qz = pz;
        float amp = 1.0f;
        for (int l = 0; l < T.t.layers; ++l) {
            f2 nz = noise_base2(tb, freq, perlin, qx, qy, qz);
            if (turb) { nz = fma2(nz, splat(2.0f), splat(-1.0f)); nz = f2{fabsf(nz.x), fabsf(nz.y)}; }
            sum = fma2(nz, splat(amp), sum);
            qx = qx * lac; qy = qy * lac; qz = qz * lac;
            amp *= gain;
        }
        v = sum * T.inv_fractal_max;
    }
    if (T.t.marbled) {  // material.rs:252-259: (sin(sf * (g * v + p.z) + ph) + 1) / 2
        f2 n = fma2(splat(T.t.marble_gain), v, pz);
        n = fma2(splat(T.t.sin_freq), n, splat(T.t.phase));
        // v_sin_f32 takes turns and wants them reduced: fract first.  ~1e-6 absolute, against a value that is
        // no better known than fp32 knows sf * (...) (6e-8 relative of up to thousands of radians)
        n = n * 0.15915494309189535f;
        const float s0 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(n.x));
        const float s1 = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(n.y));
        v = fma2(f2{s0, s1}, splat(0.5f), splat(0.5f));
    }
    return v;
}
// The noise values of a lane's two texture requests (tex < 0: none).  Lanes of a wave may ask for different
// textures: the loop runs once per DISTINCT texture of the wave (every reference scene has one), with that
// texture's descriptor read through the scalar cache.
template <class SceneT>
__device__ __forceinline__ void noise_pair(const SceneT& S, const FusedNoise* lds_fused, int tex0, int tex1, f3 p0,
                                           f3 p1, float& v0, float& v1) {
    typedef const RT_AS4 DevTexture* TexPtr;
    const TexPtr texs = (TexPtr)(uintptr_t)S.texs;
    for (;;) {
        const int want = tex0 >= 0 ? tex0 : tex1;
        const unsigned long long m = __ballot(want >= 0);
        if (m == 0ull) break;
        const int cur = __builtin_amdgcn_readlane(want, (int)__ffsll((long long)m) - 1);
        const bool m0 = tex0 == cur, m1 = tex1 == cur;
        if (m0 || m1) {
            const TexPtr T = texs + cur;
            const f2 px = {m0 ? p0.x : 0.0f, m1 ? p1.x : 0.0f}, py = {m0 ? p0.y : 0.0f, m1 ? p1.y : 0.0f},
                     pz = {m0 ? p0.z : 0.0f, m1 ? p1.z : 0.0f};
            const f2 v = noise_eval2(*T, lds_fused, px, py, pz);
            if (m0) { v0 = v.x; tex0 = -1; }
            if (m1) { v1 = v.y; tex1 = -1; }
        }
    }
}
// Texture::value (material.rs:125-127) without the noise kinds: returns true and the colour for a solid or a
// checker, false for a noise texture (the caller batches those through noise_pair).
__device__ __forceinline__ bool texture_simple(const DevScene& S, int tex_id, f3 p, f3& out) {
    // straight-line on purpose (every lane reads both colours of its descriptor and selects): no per-lane pointer
    // selects, no divergent control flow in the middle of scatter
    const DevTexture& T = S.texs[tex_id];
    const int kind = T.t.kind;
    // material.rs:155-165.  sin(f x) < 0  <=>  fract(f x / 2pi) > 1/2 : one multiply and v_fract_f32 per axis
    // instead of three libm sinf (range reduction + polynomial); the argument f*x is no better known in fp32 than
    // this phase is
    const float ft = T.freq_turns;
    bool parity = __builtin_amdgcn_fractf(p.x * ft) > 0.5f;
    parity ^= __builtin_amdgcn_fractf(p.y * ft) > 0.5f;
    parity ^= __builtin_amdgcn_fractf(p.z * ft) > 0.5f;
    const bool a = kind == RT_TEX_SOLID || parity;  // SOLID: color_a; CHECKER: odd (color_a) / even (color_b)
    out = {a ? T.t.color_a[0] : T.t.color_b[0], a ? T.t.color_a[1] : T.t.color_b[1], a ? T.t.color_a[2] : T.t.color_b[2]};
    return kind == RT_TEX_SOLID || kind == RT_TEX_CHECKER;
}
// one texture for one point (the batch kernels); tex_id < 0: no texture (returns 0).  Every lane of the wave
// must call it (noise_pair is a wave-level loop).
__device__ inline f3 texture_value(const DevScene& S, const FusedNoise* lds_fused, int tex_id, f3 p) {
    f3 c = {0.0f, 0.0f, 0.0f};
    int noise_tex = -1;
    if (tex_id >= 0 && !texture_simple(S, tex_id, p, c)) noise_tex = tex_id;
    float v = 0.0f, unused = 0.0f;
    noise_pair(S, lds_fused, noise_tex, -1, p, p, v, unused);
    return noise_tex >= 0 ? f3{v, v, v} : c;  // colour = white * value (material.rs:369-373, :529-533)
}

// ------------------------------------------------------------------ materials
__device__ __forceinline__ f3 reflect_unit(f3 ud, f3 n) {  // vec3.rs:113-116 after its normalisation
    const float k = 2.0f * dot(ud, n);
    return {ud.x - k * n.x, ud.y - k * n.y, ud.z - k * n.z};
}
__device__ __forceinline__ float schlick(float cosv, float eta) {  // material.rs:84-88
    float r0 = (1.0f - eta) * fast_rcp(1.0f + eta);
    r0 *= r0;
    const float m = 1.0f - cosv;
    return r0 + (1.0f - r0) * (m * m * m * m * m);
}
// Material::scatter with explicit uniforms u0,u1.  Returns the albedo; `r` becomes the scattered ray.
// Lanes of a wave hold different materials, so the three branches usually all run: what they share (the unit
// incident direction, the unit-sphere sample) is computed once outside them.
// A Lambertian whose texture is a noise texture returns white and its texture id in `noise_tex` (else -1): the
// caller evaluates the noise at r.o (= the hit point) for both path slots together (noise_pair) and multiplies.
// kNoise = false compiles that out: the render kernel is instantiated both ways and the scenes without a noise
// texture (upload knows) run the variant whose register budget is not set by the noise code.
template <bool kNoise = true, class SceneT>
__device__ inline f3 scatter(const SceneT& S, const HitRec& h, Ray& r, float u0, float u1, int& noise_tex) {
    const ShadeRec m = S.shade[h.sphere];
    const f3 ud = normalize(r.d);
    const float dn = dot(ud, h.n);
    f3 albedo, dir;
    noise_tex = -1;
    if (m.type == RT_MAT_DIELECTRIC) {  // material.rs:91-111
        const float eta = h.front ? fast_rcp(m.ior) : m.ior;
        const float cos_t = fminf(-dn, 1.0f);
        const float sin_t = fast_sqrt(fmaxf(0.0f, 1.0f - cos_t * cos_t));
        if (eta * sin_t > 1.0f || u0 < schlick(cos_t, eta)) {
            dir = reflect_unit(ud, h.n);  // reflect() re-normalises a unit vector: identity up to rounding
        } else {  // vec3.rs:118-123; cos re-derived un-clamped as the reference does
            const f3 par = (ud + h.n * (-dn)) * eta;
            const float k = fast_sqrt(fmaxf(0.0f, 1.0f - dot(par, par)));
            dir = par - h.n * k;
        }
        albedo = {1.0f, 1.0f, 1.0f};
    } else {
        const f3 sph = map_unit_sphere(u0, u1);
        if (m.type == RT_MAT_LAMBERTIAN) {  // material.rs:36-41
            dir = h.n + sph;
            if (dot(dir, dir) < 1e-12f) dir = h.n;  // degenerate-direction guard (deviations)
            // Texture::value for a solid or a checker (material.rs:155-165; see texture_simple), straight-line
            const float ft = m.freq_turns;
            bool parity = __builtin_amdgcn_fractf(h.p.x * ft) > 0.5f;
            parity ^= __builtin_amdgcn_fractf(h.p.y * ft) > 0.5f;
            parity ^= __builtin_amdgcn_fractf(h.p.z * ft) > 0.5f;
            const bool a = m.kind == RT_TEX_SOLID || parity;
            albedo = {a ? m.a[0] : m.b[0], a ? m.a[1] : m.b[1], a ? m.a[2] : m.b[2]};
            if (m.noise_tex >= 0) {
                albedo = {1.0f, 1.0f, 1.0f};
                if (kNoise) noise_tex = m.noise_tex;  // unreachable otherwise: that variant only runs noise-free scenes
            }
        } else {  // material.rs:61-72
            const f3 fz = sph * m.fuzz;
            dir = reflect_unit(ud, h.n) + fz;
            if (dot(dir, h.n) <= 0.0f) dir = dir - fz * 2.0f;
            albedo = {m.a[0], m.a[1], m.a[2]};
        }
    }
    r.o = h.p;
    r.d = dir;
    return albedo;
}

// Camera::get_ray (screen.rs:79-94) with explicit uniforms.
template <class CameraT>
__device__ __forceinline__ Ray camera_ray(const CameraT& c, float i, float j, float u0, float u1, float u2) {
    f3 o = {c.origin[0], c.origin[1], c.origin[2]};
    if (c.lens_radius != 0.0f) {
        const f3 rd = map_unit_disk(u0, u1) * c.lens_radius;
        o = {o.x + (rd.x * c.u[0] + rd.y * c.v[0]), o.y + (rd.x * c.u[1] + rd.y * c.v[1]),
             o.z + (rd.x * c.u[2] + rd.y * c.v[2])};
    }
    Ray r;
    r.o = o;
    r.d = {c.lower_left[0] + i * c.horiz[0] + j * c.vert[0] - o.x, c.lower_left[1] + i * c.horiz[1] + j * c.vert[1] - o.y,
           c.lower_left[2] + i * c.horiz[2] + j * c.vert[2] - o.z};
    r.time = c.has_shutter ? u2 * (c.shutter_close - c.shutter_open) + c.shutter_open : 0.0f;
    return r;
}

// sky gradient of main.rs:133-138
__device__ __forceinline__ f3 sky(f3 d) {
    const f3 ud = normalize(d);
    const float t = 0.5f * (ud.y + 1.0f);
    return {(1.0f - t) + t * 0.5f, (1.0f - t) + t * 0.7f, (1.0f - t) + t};
}

}  // namespace rtd
