//! flatten.rs — turn the crate's trait-object scene into the POD arrays of include/rtx_hip.h and render it on the GPU.
//! UNVERIFIED: written against the reference's public items (file:line cited), never compiled (no toolchain here).
//!
//! Why a hook is needed: `HitList` is `Vec<Box<dyn Hittable>>` (hit.rs:84), materials are generic parameters of
//! the shapes (shape.rs:17-21, :75-80), noise adapters are boxed closures (material.rs:169, :184-263).  None of
//! that can cross an FFI, so every trait gets ONE method that writes plain data.
//!
//! ---- additions to existing traits (each type's impl is below) -------------------------------------------------
//!   hit.rs:64       trait Hittable  { ...; fn flatten(&self, out: &mut FlatScene) -> Result<(), FlattenError>; }
//!   material.rs:20  trait Material  { ...; fn flatten(&self, out: &mut FlatScene) -> Result<u32, FlattenError>; } // material id
//!   material.rs:125 trait Texture   { ...; fn flatten(&self, out: &mut FlatScene) -> Result<i32, FlattenError>; } // texture id
//!   material.rs:172 trait NoiseAdapter { ...; fn desc(&mut self) -> &mut NoiseDesc; }
//!   material.rs:266 struct ValueNoise  { ..., desc: NoiseDesc }   // = NoiseDesc::default() in `new` (material.rs:278)
//!   material.rs:388 struct PerlinNoise { ..., desc: NoiseDesc }   // = NoiseDesc::default() in `new` (material.rs:398)
//!   screen.rs:56    struct Camera { ..., pub shutter: Option<(f64, f64)> } // `Uniform<f64>` (screen.rs:65) hides its
//!                   bounds; CameraBuilder::build (screen.rs:112-147) stores the (open, close) pair it was given
//!
//! In each existing `impl Hittable / Material / Texture for X` block the new method forwards to the inherent
//! `X::flatten_impl` defined below:  fn flatten(&self, out: &mut FlatScene) -> ... { self.flatten_impl(out) }
//!
//! ---- three one-line additions inside the adapters (material.rs) -----------------------------------------------
//!   fBm        (:184)  *self.desc() = NoiseDesc { op: RT_NOISE_FBM,        lacunarity, gain, layers, ..*self.desc() };
//!   turbulence (:211)  *self.desc() = NoiseDesc { op: RT_NOISE_TURBULENCE, lacunarity, gain, layers, ..*self.desc() };
//!   marbled    (:241)  let d = self.desc(); d.marbled = true; d.sin_freq = sin_freq; d.phase = phase_shift;
//!                      d.marble_gain = gain;   // `gain` = self.freq() * noise_gain, material.rs:251
use std::ffi::CStr;

use anyhow::{anyhow, Result};

use crate::ffi::*;
use crate::hit::{HitList, Hittable, BVH};
use crate::material::{Checkered, Dielectric, Lambertian, Material, Metal, PerlinNoise, Texture, ValueNoise};
use crate::screen::Camera;
use crate::shape::{MovingSphere, Sphere};
use crate::{Color, Vec3};

#[derive(Debug)]
pub enum FlattenError {
    /// A checker whose children are not plain colours, `DbgBlack`, `Dummy`, ...
    Unsupported(&'static str),
}

/// What fBm / turbulence / marbled captured in their closures (material.rs:184-263), as data.
#[derive(Clone, Copy, Debug)]
pub struct NoiseDesc {
    pub op: i32,
    pub lacunarity: f64,
    pub gain: f64,
    pub layers: usize,
    pub marbled: bool,
    pub sin_freq: f64,
    pub phase: f64,
    pub marble_gain: f64,
}
impl Default for NoiseDesc {
    fn default() -> Self {
        Self { op: RT_NOISE_PLAIN, lacunarity: 1., gain: 0.5, layers: 1, marbled: false, sin_freq: 0., phase: 0., marble_gain: 0. }
    }
}

/// The scene as arrays, spheres in visiting order (= sphere id; the ground is first in every scene, config.rs:153).
#[derive(Default)]
pub struct FlatScene {
    pub c0: [Vec<f32>; 3],
    pub dc: [Vec<f32>; 3],
    pub radius: Vec<f32>,
    pub material_id: Vec<u32>,
    pub materials: Vec<rt_material>,
    pub textures: Vec<rt_texture>,
    pub tables: Vec<rt_noise_table>,
}
impl FlatScene {
    fn push_sphere(&mut self, c0: Vec3, dc: Vec3, radius: f64, material: u32) {
        for (k, v) in [c0.x, c0.y, c0.z].iter().enumerate() {
            self.c0[k].push(*v as f32);
        }
        for (k, v) in [dc.x, dc.y, dc.z].iter().enumerate() {
            self.dc[k].push(*v as f32);
        }
        self.radius.push(radius as f32);
        self.material_id.push(material);
    }
    /// Borrowed view for `rt_upload_scene` (the library copies everything during the call).
    pub fn as_rt_scene(&self) -> rt_scene {
        rt_scene {
            n_spheres: self.radius.len() as u32,
            c0x: self.c0[0].as_ptr(), c0y: self.c0[1].as_ptr(), c0z: self.c0[2].as_ptr(),
            dcx: self.dc[0].as_ptr(), dcy: self.dc[1].as_ptr(), dcz: self.dc[2].as_ptr(),
            radius: self.radius.as_ptr(),
            material_id: self.material_id.as_ptr(),
            n_materials: self.materials.len() as u32, materials: self.materials.as_ptr(),
            n_textures: self.textures.len() as u32, textures: self.textures.as_ptr(),
            n_noise_tables: self.tables.len() as u32, noise_tables: self.tables.as_ptr(),
        }
    }
}

fn rgb(c: Color) -> [f32; 3] {
    [c.r as f32, c.g as f32, c.b as f32]
}

// ---- Hittable --------------------------------------------------------------------------------------------------
impl<T: Material> Sphere<T> {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<(), FlattenError> {
        let m = self.material.flatten(out)?;
        out.push_sphere(self.center, Vec3::new(0., 0., 0.), self.radius, m); // shape.rs:17-21
        Ok(())
    }
}
impl<T: Material> MovingSphere<T> {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<(), FlattenError> {
        let m = self.material.flatten(out)?;
        out.push_sphere(self.center_0, self.delta_c, self.radius, m); // center(t) = c0 + t*dc, shape.rs:96-98
        Ok(())
    }
}
impl HitList {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<(), FlattenError> {
        self.0.iter().try_for_each(|h| h.flatten(out)) // list order = sphere id (hit.rs:100-110 tie rule)
    }
}
impl BVH {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<(), FlattenError> {
        self.left.flatten(out)?; // the GPU scans a flat list: the tree (hit.rs:191-288) only fixes a visiting order
        self.right.flatten(out)
    }
}
// `Dummy` (shape.rs:150): fn flatten(..) { Err(FlattenError::Unsupported("Dummy")) }

// ---- Material --------------------------------------------------------------------------------------------------
impl<T: Texture> Lambertian<T> {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<u32, FlattenError> {
        let texture_id = self.albedo.flatten(out)?; // material.rs:27-29
        out.materials.push(rt_material { type_: RT_MAT_LAMBERTIAN, texture_id, ..Default::default() });
        Ok(out.materials.len() as u32 - 1)
    }
}
impl Metal {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<u32, FlattenError> {
        out.materials.push(rt_material {
            type_: RT_MAT_METAL, albedo: rgb(self.albedo), fuzz: self.fuzz as f32, texture_id: -1, ..Default::default()
        }); // material.rs:45-49 (fuzz already clamped by Metal::new, :51-58)
        Ok(out.materials.len() as u32 - 1)
    }
}
impl Dielectric {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<u32, FlattenError> {
        out.materials.push(rt_material {
            type_: RT_MAT_DIELECTRIC, albedo: [1., 1., 1.], ior: self.ref_index as f32, texture_id: -1, ..Default::default()
        }); // material.rs:76-78; attenuation is white (material.rs:92)
        Ok(out.materials.len() as u32 - 1)
    }
}
// `DbgBlack` (material.rs:116): fn flatten(..) { Err(FlattenError::Unsupported("DbgBlack")) }

// ---- Texture ---------------------------------------------------------------------------------------------------
impl Color {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<i32, FlattenError> {
        out.textures.push(rt_texture { kind: RT_TEX_SOLID, color_a: rgb(*self), color_b: rgb(*self), ..Default::default() });
        Ok(out.textures.len() as i32 - 1) // color.rs:31-35
    }
}
impl Checkered<Color, Color> {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<i32, FlattenError> {
        out.textures.push(rt_texture {
            kind: RT_TEX_CHECKER, color_a: rgb(self.odd), color_b: rgb(self.even), freq: self.freq as f32, ..Default::default()
        }); // parity true -> odd (material.rs:157-164)
        Ok(out.textures.len() as i32 - 1)
    }
}
// any other Checkered<O, E>: Err(FlattenError::Unsupported("Checkered of textures"))
// `impl Texture for Arc<T>` (material.rs:128-133) forwards `value`; it forwards `flatten` the same way:
//     fn flatten(&self, out: &mut FlatScene) -> Result<i32, FlattenError> { <T as Texture>::flatten(self, out) }

fn noise_texture(kind: i32, freq: f64, d: &NoiseDesc, table_id: i32) -> rt_texture {
    rt_texture {
        kind, freq: freq as f32, op: d.op, lacunarity: d.lacunarity as f32, gain: d.gain as f32, layers: d.layers as i32,
        marbled: d.marbled as i32, sin_freq: d.sin_freq as f32, phase: d.phase as f32, marble_gain: d.marble_gain as f32,
        table_id, ..Default::default()
    }
}
impl ValueNoise {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<i32, FlattenError> {
        let mut t = rt_noise_table { perms: [0; 512], randoms: [0.; 256] };
        t.perms.copy_from_slice(&self.perms); // all 512 entries, second half as built (material.rs:289-306)
        for (dst, src) in t.randoms.iter_mut().zip(self.randoms.iter()) {
            *dst = *src as f32; // material.rs:280-287
        }
        out.tables.push(t);
        out.textures.push(noise_texture(RT_TEX_VALUE_NOISE, self.freq, &self.desc, out.tables.len() as i32 - 1));
        Ok(out.textures.len() as i32 - 1)
    }
}
impl PerlinNoise {
    pub fn flatten_impl(&self, out: &mut FlatScene) -> Result<i32, FlattenError> {
        let mut t = rt_noise_table { perms: [0; 512], randoms: [0.; 256] };
        t.perms.copy_from_slice(&self.perms); // material.rs:400-424
        out.tables.push(t);
        out.textures.push(noise_texture(RT_TEX_PERLIN_NOISE, self.freq, &self.desc, out.tables.len() as i32 - 1));
        Ok(out.textures.len() as i32 - 1)
    }
}

// ---- Camera: exactly the fields `get_ray` reads (screen.rs:79-94), narrowed to f32 --------------------------------
pub fn rt_camera_from(c: &Camera) -> rt_camera {
    let v3 = |v: Vec3| [v.x as f32, v.y as f32, v.z as f32];
    let (open, close, has) = match c.shutter {
        Some((o, cl)) => (o as f32, cl as f32, 1),
        None => (0., 0., 0),
    };
    rt_camera {
        origin: v3(c.origin), lower_left: v3(c.lower_left), horiz: v3(c.horiz), vert: v3(c.vert), u: v3(c.u), v: v3(c.v),
        lens_radius: c.lens_radius as f32, shutter_open: open, shutter_close: close, has_shutter: has,
    }
}

// ---- the replacement for main.rs:64-92 --------------------------------------------------------------------------
/// Renders `world` as seen by `camera` into `rgb` (width*height*3 linear radiance means, y = 0 top row).
pub fn render_hip(
    world: &dyn Hittable, camera: &Camera, width: usize, height: usize, samples: u32, max_depth: u32, antialias: bool,
    seed: u64, rgb: &mut [f32],
) -> Result<rt_stats> {
    assert_eq!(rgb.len(), width * height * 3);
    let mut flat = FlatScene::default();
    world.flatten(&mut flat).map_err(|e| anyhow!("scene cannot be flattened: {:?}", e))?;
    let scene = flat.as_rt_scene();
    let cam = rt_camera_from(camera);
    let params = rt_params {
        width: width as u32, height: height as u32, spp: samples, max_depth, antialias: antialias as i32, spp_offset: 0,
        seed, t_min: 0.001 /* main.rs:116 */, reserved: 0,
    };
    let mut stats = rt_stats::default();
    unsafe {
        let mut ctx: *mut rt_ctx = std::ptr::null_mut();
        let check = |rc: i32, ctx: *const rt_ctx| -> Result<()> {
            if rc == RT_OK {
                Ok(())
            } else {
                Err(anyhow!("rtx_hip error {}: {}", rc, CStr::from_ptr(rt_last_error(ctx)).to_string_lossy()))
            }
        };
        check(rt_create(0, &mut ctx), std::ptr::null())?;
        let r = check(rt_upload_scene(ctx, &scene), ctx)
            .and_then(|_| check(rt_render(ctx, &cam, &params, rgb.as_mut_ptr(), &mut stats), ctx));
        rt_destroy(ctx);
        r?;
    }
    Ok(stats)
}

// main.rs after the change (lines 26, 60, 62, 95, 100 of the reference stay as they are):
//
//     let (camera, world) = CFG.scene.create(&mut rng);                     // main.rs:26
//     let seed: u64 = rng.gen();                                            // main.rs:60
//     let time = Instant::now();                                            // main.rs:62
//     let mut rgb = vec![0f32; width * height * 3];
//     raytracing::flatten::render_hip(&world, &camera, width, height, CFG.samples.get() as u32,
//                                     CFG.max_depth.get(), CFG.antialias, seed, &mut rgb)?;   // replaces main.rs:64-92
//     eprintln!("\nRending time elapsed: {:.2} seconds", time.elapsed().as_secs_f64());         // main.rs:95
//     for (pix, c) in screen.pixels_mut().zip(rgb.chunks_exact(3)) {
//         *pix = Color::new(c[0] as f64, c[1] as f64, c[2] as f64);
//     }
//     let buffer = screen.encode();                                         // main.rs:100
