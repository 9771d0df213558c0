//! ffi.rs — `#[repr(C)]` mirror of include/rtx_hip.h (ABI version 1).  UNVERIFIED: no Rust toolchain in the build
//! image; field order and types follow the header line by line.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int};

pub const RT_OK: c_int = 0;
pub const RT_ERR_BAD_ARG: c_int = -1;
pub const RT_ERR_HIP: c_int = -2;
pub const RT_ERR_SCENE_TOO_LARGE: c_int = -3;

pub const RT_MAT_LAMBERTIAN: i32 = 0;
pub const RT_MAT_METAL: i32 = 1;
pub const RT_MAT_DIELECTRIC: i32 = 2;
pub const RT_TEX_SOLID: i32 = 0;
pub const RT_TEX_CHECKER: i32 = 1;
pub const RT_TEX_VALUE_NOISE: i32 = 2;
pub const RT_TEX_PERLIN_NOISE: i32 = 3;
pub const RT_NOISE_PLAIN: i32 = 0;
pub const RT_NOISE_FBM: i32 = 1;
pub const RT_NOISE_TURBULENCE: i32 = 2;

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_noise_table {
    pub perms: [u16; 512],
    pub randoms: [f32; 256],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_texture {
    pub kind: i32,
    pub color_a: [f32; 3], // solid colour / checker `odd`
    pub color_b: [f32; 3], // checker `even`
    pub freq: f32,
    pub op: i32,
    pub lacunarity: f32,
    pub gain: f32,
    pub layers: i32,
    pub marbled: i32,
    pub sin_freq: f32,
    pub phase: f32,
    pub marble_gain: f32, // freq * noise_gain (material.rs:251)
    pub table_id: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_material {
    pub type_: i32,
    pub albedo: [f32; 3],
    pub fuzz: f32,
    pub ior: f32,
    pub texture_id: i32,
    pub reserved: i32,
}

#[repr(C)]
pub struct rt_scene {
    pub n_spheres: u32,
    pub c0x: *const f32,
    pub c0y: *const f32,
    pub c0z: *const f32,
    pub dcx: *const f32,
    pub dcy: *const f32,
    pub dcz: *const f32,
    pub radius: *const f32,
    pub material_id: *const u32,
    pub n_materials: u32,
    pub materials: *const rt_material,
    pub n_textures: u32,
    pub textures: *const rt_texture,
    pub n_noise_tables: u32,
    pub noise_tables: *const rt_noise_table,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub origin: [f32; 3],
    pub lower_left: [f32; 3],
    pub horiz: [f32; 3],
    pub vert: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub lens_radius: f32,
    pub shutter_open: f32,
    pub shutter_close: f32,
    pub has_shutter: i32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_params {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub max_depth: u32,
    pub antialias: i32,
    pub spp_offset: u32,
    pub seed: u64,
    pub t_min: f32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub rays_traced: u64,
    pub samples: u64,
    pub kernel_ms: f32,
    pub total_ms: f32,
    pub n_launches: u32,
    pub reserved: u32,
}

pub const RT_MAX_DEVICES: usize = 16;
pub const RT_NOALLOC_MAX_PIXELS: u32 = 1 << 20;
pub const RT_ERR_RCCL: c_int = -7;

#[repr(C)]
pub struct rt_multi {
    _p: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_multi_stats {
    pub rays_traced: u64,
    pub samples: u64,
    pub kernel_ms: f32, // slowest device
    pub reduce_ms: f32, // the exchange step (one ncclReduce over xGMI, or the copy-and-add fallback)
    pub n_devices: u32,
    pub exchange: u32, // 0 = ncclReduce, 1 = parts copied to the first device and added there
    pub device_kernel_ms: [f32; RT_MAX_DEVICES],
}

extern "C" {
    pub fn rt_abi_version() -> c_int;
    pub fn rt_create(device: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_destroy(ctx: *mut rt_ctx);
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_upload_scene(ctx: *mut rt_ctx, scene: *const rt_scene) -> c_int;
    pub fn rt_render(
        ctx: *mut rt_ctx,
        cam: *const rt_camera,
        params: *const rt_params,
        out_rgb: *mut f32, // width*height*3, row-major, y = 0 is the top row, linear radiance means
        stats: *mut rt_stats,
    ) -> c_int;
    // per-pixel radiance SUMS into a DEVICE buffer (callers that manage device memory themselves)
    pub fn rt_render_accum_device(
        ctx: *mut rt_ctx,
        cam: *const rt_camera,
        params: *const rt_params,
        d_sum_rgb: *mut f32,
        stats: *mut rt_stats,
    ) -> c_int;
    // the same for callers that cannot allocate (workspace in a static device array, <= RT_NOALLOC_MAX_PIXELS pixels)
    pub fn rt_render_accum_device_noalloc(
        ctx: *mut rt_ctx,
        cam: *const rt_camera,
        params: *const rt_params,
        d_sum_rgb: *mut f32,
        stats: *mut rt_stats,
    ) -> c_int;

    // ---- one process, the GPUs of one node: spp-partitioned render + one RCCL sum-reduce onto devices[0]
    pub fn rt_partition_spp(total_spp: u32, n_parts: u32, part: u32, first: *mut u32, count: *mut u32) -> c_int;
    pub fn rt_multi_create(device_ordinals: *const c_int, n_devices: c_int, out: *mut *mut rt_multi) -> c_int;
    pub fn rt_multi_destroy(m: *mut rt_multi);
    pub fn rt_multi_last_error(m: *const rt_multi) -> *const c_char;
    pub fn rt_multi_upload_scene(m: *mut rt_multi, scene: *const rt_scene) -> c_int;
    pub fn rt_multi_render(
        m: *mut rt_multi,
        cam: *const rt_camera,
        params: *const rt_params,
        out_rgb: *mut f32,
        stats: *mut rt_multi_stats,
    ) -> c_int;
}
