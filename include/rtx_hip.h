/* rtx_hip.h — C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * This is the drop-in boundary for ONE path of ArifRoktim/raytracing: the
 * per-pixel-sample path trace, reference `src/main.rs:64-92` (pixel/sample
 * loop) + `src/main.rs:110-139` (`ray_color`) and everything they call.  The
 * reference has no FFI today (one Rust crate, trait objects); its `main()`
 * would call `rt_render` in place of the rayon block, after `Scene::create`
 * (`main.rs:26`) and before `Screen::encode` (`main.rs:100`).  INTEGRATION.md
 * shows the Rust-side `ffi.rs` a maintainer would add.
 *
 * Plain C: POD structs, plain pointers and sizes.  No HIP/torch types.
 * Caller owns every input struct and output buffer; the library copies what it
 * needs during the call and keeps no caller pointer after returning.
 * All entry points are synchronous (return after the stream has drained) and
 * return 0 on success or a negative rt_status; text via rt_last_error().
 * An rt_ctx is bound to one device and is not thread-safe (the reference does
 * one render per process: `main.rs:15-106`).
 */
#ifndef RTX_HIP_H
#define RTX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 1

/* Limits.  The sphere lists are streamed through the scalar cache (not LDS), so the sphere limit is a sanity bound,
 * not a capacity: every reference scene has < 500.  Cost per ray segment grows slower than the count (bounding
 * spheres over runs of 4 .. 1024 list neighbours let a wave skip what none of its rays can meet).
 * Noise tables (2 KiB each) are staged in LDS.  Images: 2 <= width, height <= 65536 and at most 2^28 pixels; batch entry
 * points take at most 2^30 elements per call. */
#define RT_MAX_SPHERES 16384
#define RT_MAX_MATERIALS 16384
#define RT_MAX_TEXTURES 16384
#define RT_MAX_NOISE_TABLES 4
#define RT_MAX_DEVICES 16 /* devices of one rt_multi (one node) */
/* Largest image (pixels) of rt_render_accum_device_noalloc: its workspace is a static device array of
 * RT_NOALLOC_MAX_PIXELS * 24 bytes (24 MiB) per GPU.  1200x800 (BASELINE configs[1]-[3]) fits; 3840x2160 does not. */
#define RT_NOALLOC_MAX_PIXELS (1u << 20)

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_BAD_ARG = -1,
    RT_ERR_HIP = -2,
    RT_ERR_SCENE_TOO_LARGE = -3,
    RT_ERR_NO_SCENE = -4,
    RT_ERR_NO_DEVICE = -5,
    RT_ERR_ALLOC = -6,
    RT_ERR_RCCL = -7 /* librccl missing, or an RCCL call failed (multi-device entry points only) */
} rt_status;

/* `impl Material for ...` in reference src/material.rs:35-112 */
typedef enum rt_material_type {
    RT_MAT_LAMBERTIAN = 0, /* material.rs:25-42  */
    RT_MAT_METAL = 1,      /* material.rs:44-73  */
    RT_MAT_DIELECTRIC = 2  /* material.rs:75-112 */
} rt_material_type;

/* `impl Texture for ...`: color.rs:31-35, material.rs:155-166, :369-373, :529-533 */
typedef enum rt_texture_kind {
    RT_TEX_SOLID = 0,
    RT_TEX_CHECKER = 1,
    RT_TEX_VALUE_NOISE = 2,
    RT_TEX_PERLIN_NOISE = 3
} rt_texture_kind;

/* NoiseAdapter callbacks (material.rs:184-237) as data: closures cannot cross an FFI. */
typedef enum rt_noise_op {
    RT_NOISE_PLAIN = 0,     /* eval() with no callback: noise(p)         */
    RT_NOISE_FBM = 1,       /* NoiseAdapter::fBm        material.rs:184  */
    RT_NOISE_TURBULENCE = 2 /* NoiseAdapter::turbulence material.rs:211  */
} rt_noise_op;

/* ValueNoise{randoms,perms} material.rs:266-272 / PerlinNoise{perms} material.rs:388-392.
 * All 512 perms are data: the second half is copied during the shuffle and is
 * NOT perms[i & 255] (material.rs:294-299). */
typedef struct rt_noise_table {
    uint16_t perms[512];
    float randoms[256]; /* unused by Perlin noise */
} rt_noise_table;

typedef struct rt_texture {
    int32_t kind;     /* rt_texture_kind */
    float color_a[3]; /* SOLID: the colour. CHECKER: `odd`  (material.rs:138) */
    float color_b[3]; /* CHECKER: `even` (material.rs:139) */
    float freq;       /* CHECKER freq (material.rs:137) / noise freq (material.rs:270,390) */
    int32_t op;       /* rt_noise_op */
    float lacunarity; /* fBm / turbulence */
    float gain;
    int32_t layers;
    int32_t marbled;   /* != 0: wrap with NoiseAdapter::marbled (material.rs:241-263) */
    float sin_freq;    /* marbled: (sin_freq, phase_shift) */
    float phase;
    float marble_gain; /* marbled: `gain` = freq * noise_gain (material.rs:251) */
    int32_t table_id;  /* index into rt_scene.noise_tables */
} rt_texture;

typedef struct rt_material {
    int32_t type;       /* rt_material_type */
    float albedo[3];    /* METAL albedo (material.rs:46) */
    float fuzz;         /* METAL fuzz, already clamped to <= 1 (material.rs:52) */
    float ior;          /* DIELECTRIC ref_index (material.rs:77) */
    int32_t texture_id; /* LAMBERTIAN albedo texture (material.rs:28) */
    int32_t reserved;
} rt_material;

/* Flattened `HitList` (hit.rs:83-110) of `Sphere` (shape.rs:16-21) and
 * `MovingSphere` (shape.rs:75-80): SoA, world order (index = sphere id, list
 * order decides exact-t ties: first wins).  A static sphere has dc = 0.
 * center(time) = c0 + time * dc (shape.rs:96-98). */
typedef struct rt_scene {
    uint32_t n_spheres;
    const float* c0x;
    const float* c0y;
    const float* c0z;
    const float* dcx;
    const float* dcy;
    const float* dcz;
    const float* radius;
    const uint32_t* material_id;
    uint32_t n_materials;
    const rt_material* materials;
    uint32_t n_textures;
    const rt_texture* textures;
    uint32_t n_noise_tables;
    const rt_noise_table* noise_tables;
} rt_scene;

/* Exactly the fields `Camera::get_ray` reads (screen.rs:79-94). */
typedef struct rt_camera {
    float origin[3];
    float lower_left[3];
    float horiz[3];
    float vert[3];
    float u[3];
    float v[3];
    float lens_radius;   /* 0 disables depth of field (screen.rs:80) */
    float shutter_open;  /* Uniform range of ray.time (screen.rs:87) */
    float shutter_close;
    int32_t has_shutter; /* 0: time = 0 */
} rt_camera;

/* Mirrors `Config` (config.rs:25-69). */
typedef struct rt_params {
    uint32_t width;
    uint32_t height;
    uint32_t spp;        /* samples rendered by THIS call (config.rs:52) */
    uint32_t max_depth;  /* config.rs:56, default 100 */
    int32_t antialias;   /* config.rs:48 */
    uint32_t spp_offset; /* index of the first sample: disjoint RNG streams for
                            multi-GPU spp partitioning / progressive rendering */
    uint64_t seed;       /* counter-RNG key (replaces main.rs:60,71) */
    float t_min;         /* 0.001 in the reference (main.rs:116) */
    uint32_t reserved;   /* must be 0 (RT_ERR_BAD_ARG otherwise) */
} rt_params;

typedef struct rt_stats {
    uint64_t rays_traced; /* ray segments: primary + every scattered ray */
    uint64_t samples;     /* width*height*spp */
    float kernel_ms;      /* HIP-event time of the path-trace kernel(s) on the ctx stream */
    float total_ms;       /* memset + kernels + finalize (+ D2H for rt_render) */
    uint32_t n_launches;  /* path-trace kernel launches in this call */
    uint32_t reserved;
} rt_stats;

typedef struct rt_ray {
    float origin[3];
    float dir[3]; /* NOT normalised, as in the reference (lib.rs:21-25) */
    float time;
} rt_ray;

/* `Hit` (hit.rs:11-23) without u,v (dead: no texture reads them) and with the
 * material reference as an id. */
typedef struct rt_hit {
    int32_t hit; /* 0 = miss (then every other field is 0) */
    int32_t sphere_id;
    int32_t material_id;
    int32_t front_face;
    float t;
    float point[3];
    float normal[3]; /* unit, faces the ray (hit.rs:55-59) */
} rt_hit;

/* `Scatter` (material.rs:10-13). */
typedef struct rt_scatter {
    int32_t scattered; /* always 1 for the three materials */
    float albedo[3];
    float origin[3];
    float dir[3];
    float time;
} rt_scatter;

/* ---- context ------------------------------------------------------------ */
typedef struct rt_ctx rt_ctx;

/* Create a context on HIP device `device_ordinal`.  Replaces nothing in the
 * reference (it has no device); it is the handle `main()` (main.rs:15) holds. */
int rt_create(int device_ordinal, rt_ctx** out);
void rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx); /* ctx may be NULL: last create error */
int rt_abi_version(void);

/* Copy a flattened world to the device.  Replaces the `&HitList` borrowed by
 * the rayon closure (main.rs:67,85).  Validates ids and sizes. */
int rt_upload_scene(rt_ctx* ctx, const rt_scene* scene);

/* ---- the hot path --------------------------------------------------------- */
/* Replaces main.rs:64-92 (+ ray_color main.rs:110-139).  out_rgb: HOST buffer,
 * width*height*3 floats, row-major, y = 0 is the top row, linear radiance MEAN
 * over params->spp samples (what `*pix = avg` holds, main.rs:88-89). */
int rt_render(rt_ctx* ctx, const rt_camera* cam, const rt_params* params,
              float* out_rgb, rt_stats* stats);

/* Same, but writes the per-pixel radiance SUM (not mean) over this call's
 * samples into a DEVICE buffer of width*height*3 floats on the ctx's device —
 * the fp32 accumulator that multi-GPU callers sum with an RCCL reduce. */
int rt_render_accum_device(rt_ctx* ctx, const rt_camera* cam, const rt_params* params,
                           float* d_sum_rgb, rt_stats* stats);

/* rt_render_accum_device for callers that CANNOT ALLOCATE (tight serving loops, harnesses that forbid hipMalloc,
 * code that must not disturb an allocator while a HIP graph is being captured elsewhere).  No counterpart in the
 * reference (it has no device, main.rs:64-92 allocates nothing either).  Same arguments, same kernels and
 * bit-identical output, but the fixed-point radiance accumulators and the launch counters live in a static device
 * array inside the library instead of a buffer the ctx hipMallocs on demand: this entry point allocates no device
 * or host memory (no hipMalloc / hipFree or other allocating routine).  Like every entry point it runs SYNCHRONOUSLY
 * on the ctx's own stream and returns after that stream has drained: it is not capturable into a caller's graph.
 * Limit: width*height <= RT_NOALLOC_MAX_PIXELS, larger images return RT_ERR_BAD_ARG.  The array is per GPU, not
 * per ctx, and keeps the previous call's contents (each call clears what it uses): two such calls that overlap
 * on one GPU (two ctxs, two threads) race; calls on different GPUs do not. */
int rt_render_accum_device_noalloc(rt_ctx* ctx, const rt_camera* cam, const rt_params* params,
                                   float* d_sum_rgb, rt_stats* stats);

/* ---- fixed-batch entry points (parity tests; host buffers) -------------------- */
/* `HitList::hit(ray, t_min..inf)` (hit.rs:100-110) for n rays. */
int rt_intersect_batch(rt_ctx* ctx, const rt_ray* rays, uint32_t n, float t_min,
                       rt_hit* out);
/* `Material::scatter` (material.rs:36,61,91) with explicit uniforms u[4] in [0,1)
 * per ray: Lambertian/Metal use u[0],u[1] (unit sphere), Dielectric u[0]. */
int rt_scatter_batch(rt_ctx* ctx, const rt_ray* rays, const rt_hit* hits,
                     const float* randoms4, uint32_t n, rt_scatter* out);
/* `Texture::value(_, _, point)` (material.rs:125-127) for n points. */
int rt_texture_batch(rt_ctx* ctx, int32_t texture_id, const float* points3,
                     uint32_t n, float* out_rgb3);
/* `Camera::get_ray(i, j, rng)` (screen.rs:79-94) with explicit uniforms u[4]:
 * u[0],u[1] lens disc, u[2] shutter time. ij2: (i, j) pairs. */
int rt_camera_rays(rt_ctx* ctx, const rt_camera* cam, const float* ij2,
                   const float* randoms4, uint32_t n, rt_ray* out);
/* The counter RNG itself: Philox4x32-10, key = seed, counter = (c0,c1,c2,c3). */
int rt_rng_batch(rt_ctx* ctx, uint64_t seed, const uint32_t* counters4, uint32_t n,
                 uint32_t* out4);

/* ---- multi-device (one process, the GPUs of one node) ------------------------------------------------------
 * The reference's only parallelism is rayon over rows (main.rs:64-67); on N GPUs the samples-per-pixel are split:
 * device g renders sample indices [first_g, first_g + count_g) of every pixel (rt_partition_spp), so the counter-RNG
 * streams are disjoint and the N-device image equals the 1-device image up to fp32 summation order.  The scene is
 * replicated; the only exchange step is ONE ncclReduce(sum, float, width*height*3) onto the first device over
 * xGMI.  RCCL is loaded at run time by rt_multi_create; if it is missing or a communicator cannot be formed, the
 * parts are copied to the first device (hipMemcpyPeerAsync) and added there instead: same sums, n - 1 copies
 * (rt_multi_stats.exchange tells which). */
typedef struct rt_multi rt_multi;

typedef struct rt_multi_stats {
    uint64_t rays_traced; /* all devices */
    uint64_t samples;     /* width*height*spp */
    float kernel_ms;      /* slowest device's path-trace kernel time */
    float reduce_ms;      /* the exchange step, events on the root device's stream */
    uint32_t n_devices;
    uint32_t exchange;    /* how the sums were combined: 0 = one ncclReduce, 1 = copied to the root device and added
                             there (no communicator could be formed: librccl missing or ncclCommInitAll failed) */
    float device_kernel_ms[RT_MAX_DEVICES];
} rt_multi_stats;

/* Contiguous split of sample indices [0, total_spp) into n_parts: the remainder goes to the lowest parts; a part
 * may get 0 samples when total_spp < n_parts.  Pure arithmetic (no device needed). */
int rt_partition_spp(uint32_t total_spp, uint32_t n_parts, uint32_t part, uint32_t* first, uint32_t* count);

/* One rt_ctx per listed device + one RCCL communicator over them (ncclCommInitAll); device_ordinals[0] is the
 * reduce root.  Replaces nothing in the reference. */
int rt_multi_create(const int* device_ordinals, int n_devices, rt_multi** out);
void rt_multi_destroy(rt_multi* m);
const char* rt_multi_last_error(const rt_multi* m); /* m may be NULL: last create error */
int rt_multi_upload_scene(rt_multi* m, const rt_scene* scene);
/* rt_render over all devices: out_rgb as for rt_render (HOST, linear radiance mean over params->spp samples). */
int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_params* params, float* out_rgb,
                    rt_multi_stats* stats);

/* ---- adjacent: `Screen::encode` screen.rs:26-45 ----------- */
/* Gamma-2 + 0RGB packing of a HOST rgb buffer on the device.  Returns the number
 * of pixels the reference would panic on (NaN / outside [0,1]) in *n_invalid;
 * such pixels are clamped. */
int rt_encode(rt_ctx* ctx, const float* rgb, uint32_t n_pixels, uint32_t* out_0rgb,
              uint32_t* n_invalid);

/* ---- host side above the ABI: restates the reference's scene/camera API ------ */
/* (C++ in raytracing_amd/csrc/host_scene.cpp; the Rust crate keeps its own.) */
typedef struct rth_world rth_world; /* owns the arrays an rt_scene points into */

/* `Scene::from_str` + `Scene::create` (config.rs:85-107,146-404).  `name` is one
 * of the 14 reference scene names, exact case (config.rs:407-426), or the
 * extension "FinalStatic" (RTIOW book-1 final scene: Random with dc = 0, grey
 * ground, no shutter; consumes the same RNG draws).  has_seed = 0 mirrors
 * `--rng` absent (entropy).  aspect = width / height (screen.rs:263-270). */
int rth_world_create(const char* name, int has_seed, uint64_t seed, double aspect,
                     rth_world** out);
void rth_world_destroy(rth_world* w);
const rt_scene* rth_world_scene(const rth_world* w);
const rt_camera* rth_world_camera(const rth_world* w);
/* `let seed: u64 = rng.gen()` (main.rs:60): the render seed drawn after the scene. */
uint64_t rth_world_render_seed(const rth_world* w);
/* Index of the first sphere the reference wraps in its BVH (config.rs:197), or -1:
 * the GPU path ignores it (flat list); the CPU oracle uses it to time the real algorithm. */
int rth_world_bvh_first(const rth_world* w);
const char* rth_last_error(void);
/* `Scene::VARIANTS` (config.rs:76): NULL-terminated list of accepted names. */
const char* const* rth_scene_names(void);

/* `CameraBuilder::build` (screen.rs:112-147); view_up NULL = UNIT_Y,
 * focus_dist <= 0 = |origin - look_at|, has_shutter 0 = None.
 * Returns RT_ERR_BAD_ARG where `verify` (screen.rs:150-198) would fail. */
int rth_camera_build(const double origin[3], const double look_at[3],
                     const double view_up[3], double vfov_degrees, double aspect_ratio,
                     double aperture, double focus_dist, int has_shutter,
                     double shutter_open, double shutter_close, rt_camera* out);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HIP_H */
