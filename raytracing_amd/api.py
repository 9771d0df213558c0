"""ctypes binding of the C ABI in include/rtx_hip.h (librtx_hip.so).

Thin on purpose: the product is the HIP library; Python only loads it for tests and bench.py.
There is NO CPU fallback here: if the shared library is missing, or no gfx950 device is
visible when a context is created, this raises.

Reference interface mirrored (names / argument meaning): `Scene` (config.rs:85-107),
`Camera` / `CameraBuilder` (screen.rs:56-147), `Config` (config.rs:25-69), the hot loop
main.rs:64-92 + `ray_color` main.rs:110-139.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librtx_hip.so")  # the in-tree build; see allow_lib_override()
_override_allowed = False

RT_OK = 0
RT_ERR_BAD_ARG = -1
RT_ERR_HIP = -2
RT_ERR_SCENE_TOO_LARGE = -3
RT_ERR_NO_SCENE = -4
RT_ERR_NO_DEVICE = -5
RT_ERR_ALLOC = -6
RT_ERR_RCCL = -7
RT_MAX_DEVICES = 16

RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC = 0, 1, 2
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_VALUE_NOISE, RT_TEX_PERLIN_NOISE = 0, 1, 2, 3
RT_NOISE_PLAIN, RT_NOISE_FBM, RT_NOISE_TURBULENCE = 0, 1, 2

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class NoiseTable(C.Structure):
    _fields_ = [("perms", C.c_uint16 * 512), ("randoms", C.c_float * 256)]


class Texture(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("color_a", C.c_float * 3), ("color_b", C.c_float * 3), ("freq", C.c_float),
        ("op", C.c_int32), ("lacunarity", C.c_float), ("gain", C.c_float), ("layers", C.c_int32),
        ("marbled", C.c_int32), ("sin_freq", C.c_float), ("phase", C.c_float), ("marble_gain", C.c_float),
        ("table_id", C.c_int32),
    ]


class Material(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("albedo", C.c_float * 3), ("fuzz", C.c_float), ("ior", C.c_float),
        ("texture_id", C.c_int32), ("reserved", C.c_int32),
    ]


class Scene(C.Structure):
    _fields_ = [
        ("n_spheres", C.c_uint32),
        ("c0x", f32p), ("c0y", f32p), ("c0z", f32p), ("dcx", f32p), ("dcy", f32p), ("dcz", f32p),
        ("radius", f32p), ("material_id", u32p),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
        ("n_noise_tables", C.c_uint32), ("noise_tables", C.POINTER(NoiseTable)),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3), ("lower_left", C.c_float * 3), ("horiz", C.c_float * 3),
        ("vert", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3),
        ("lens_radius", C.c_float), ("shutter_open", C.c_float), ("shutter_close", C.c_float),
        ("has_shutter", C.c_int32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
        ("antialias", C.c_int32), ("spp_offset", C.c_uint32), ("seed", C.c_uint64), ("t_min", C.c_float),
        ("reserved", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_traced", C.c_uint64), ("samples", C.c_uint64), ("kernel_ms", C.c_float), ("total_ms", C.c_float),
        ("n_launches", C.c_uint32), ("reserved", C.c_uint32),
    ]


class MultiStats(C.Structure):
    _fields_ = [
        ("rays_traced", C.c_uint64), ("samples", C.c_uint64), ("kernel_ms", C.c_float), ("reduce_ms", C.c_float),
        ("n_devices", C.c_uint32), ("exchange", C.c_uint32), ("device_kernel_ms", C.c_float * RT_MAX_DEVICES),
    ]


# numpy views of the batch PODs
ray_dtype = np.dtype([("origin", "<f4", 3), ("dir", "<f4", 3), ("time", "<f4")])
hit_dtype = np.dtype([("hit", "<i4"), ("sphere_id", "<i4"), ("material_id", "<i4"), ("front_face", "<i4"),
                      ("t", "<f4"), ("point", "<f4", 3), ("normal", "<f4", 3)])
scatter_dtype = np.dtype([("scattered", "<i4"), ("albedo", "<f4", 3), ("origin", "<f4", 3), ("dir", "<f4", 3),
                          ("time", "<f4")])
assert ray_dtype.itemsize == 28 and hit_dtype.itemsize == 44 and scatter_dtype.itemsize == 44

# every symbol include/rtx_hip.h declares (tests check the library exports them all)
ABI_SYMBOLS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_abi_version", "rt_upload_scene", "rt_render",
    "rt_render_accum_device", "rt_render_accum_device_noalloc", "rt_intersect_batch", "rt_scatter_batch", "rt_texture_batch", "rt_camera_rays",
    "rt_rng_batch", "rt_encode", "rt_partition_spp", "rt_multi_create", "rt_multi_destroy", "rt_multi_last_error",
    "rt_multi_upload_scene", "rt_multi_render", "rth_world_create", "rth_world_destroy", "rth_world_scene", "rth_world_camera",
    "rth_world_render_seed", "rth_world_bvh_first", "rth_last_error", "rth_scene_names", "rth_camera_build",
    # include/rtx_hip_debug.h (test / tuning hooks, not part of the drop-in ABI)
    "rt_debug_set_work_units", "rt_debug_set_scan_order", "rt_debug_set_tile_buffers", "rt_debug_set_planar_list",
    "rt_debug_scan_layout", "rt_debug_set_direct", "rt_debug_multi", "rt_debug_set_tail_order", "rt_debug_tail_order_stats",
]

_lib = None


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


def allow_lib_override() -> None:
    """A/B tooling only (bench.py --allow-lib-override, tools/): let the environment variable RTX_HIP_LIB name the
    library file to load instead of the in-tree build.  Without this call a set RTX_HIP_LIB is an ERROR, so that no
    test or measurement can silently run another file than the one in the tree."""
    global _override_allowed
    _override_allowed = True


def lib_provenance() -> dict:
    """Absolute path and sha256 of the library file actually loaded (bench.py prints it in its JSON line)."""
    import hashlib
    load()
    with open(_lib_path, "rb") as f:
        digest = hashlib.sha256(f.read()).hexdigest()
    return {"path": _lib_path, "sha256": digest, "in_tree": os.path.realpath(_lib_path) == os.path.realpath(LIB_PATH)}


_lib_path = None


def load() -> C.CDLL:
    """Load librtx_hip.so (built by raytracing_amd/csrc/Makefile). Raises if it is missing."""
    global _lib, _lib_path
    if _lib is not None:
        return _lib
    path = LIB_PATH
    env = os.environ.get("RTX_HIP_LIB")
    if env:
        if not _override_allowed:
            raise RuntimeError("RTX_HIP_LIB is set but library overrides are not allowed here "
                               "(bench.py: --allow-lib-override; tools: api.allow_lib_override())")
        path = os.path.abspath(env)
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64. Two HIP runtimes in one process
    # cannot both own the GPU ("No HIP GPUs are available" in whichever initialises second), so when torch is
    # installed let it load its runtime FIRST: our DT_NEEDED libamdhip64.so.7 then binds to that same copy.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `make -C raytracing_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(path)
    _lib_path = path
    vp = C.c_void_p
    lib.rt_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_destroy.restype = None
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_upload_scene.argtypes = [vp, C.POINTER(Scene)]
    lib.rt_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
    lib.rt_render_accum_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
    lib.rt_render_accum_device_noalloc.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(Stats)]
    lib.rt_debug_set_work_units.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32]
    lib.rt_debug_set_scan_order.argtypes = [vp, C.c_int]
    lib.rt_debug_set_tile_buffers.argtypes = [vp, C.c_int]
    if hasattr(lib, "rt_debug_multi"):
        lib.rt_debug_multi.argtypes = [C.c_int]
    if hasattr(lib, "rt_debug_set_direct"):
        lib.rt_debug_set_direct.argtypes = [vp, C.c_int]
    if hasattr(lib, "rt_debug_set_tail_order"):
        lib.rt_debug_set_tail_order.argtypes = [vp, C.c_int]
        lib.rt_debug_tail_order_stats.argtypes = [vp, C.POINTER(C.c_uint32 * 3)]
    if hasattr(lib, "rt_debug_set_planar_list"):  # (absent from older builds that tools/ab*.sh compare against)
        lib.rt_debug_set_planar_list.argtypes = [vp, C.c_int]
        lib.rt_debug_scan_layout.argtypes = [vp, C.POINTER(C.c_int32 * 8)]
    lib.rt_intersect_batch.argtypes = [vp, vp, C.c_uint32, C.c_float, vp]
    lib.rt_scatter_batch.argtypes = [vp, vp, vp, vp, C.c_uint32, vp]
    lib.rt_texture_batch.argtypes = [vp, C.c_int32, vp, C.c_uint32, vp]
    lib.rt_camera_rays.argtypes = [vp, C.POINTER(Camera), vp, vp, C.c_uint32, vp]
    lib.rt_rng_batch.argtypes = [vp, C.c_uint64, vp, C.c_uint32, vp]
    lib.rt_encode.argtypes = [vp, vp, C.c_uint32, vp, u32p]
    lib.rt_partition_spp.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    lib.rt_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    lib.rt_multi_destroy.argtypes = [vp]
    lib.rt_multi_destroy.restype = None
    lib.rt_multi_last_error.argtypes = [vp]
    lib.rt_multi_last_error.restype = C.c_char_p
    lib.rt_multi_upload_scene.argtypes = [vp, C.POINTER(Scene)]
    lib.rt_multi_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Params), vp, C.POINTER(MultiStats)]
    lib.rth_world_create.argtypes = [C.c_char_p, C.c_int, C.c_uint64, C.c_double, C.POINTER(vp)]
    lib.rth_world_destroy.argtypes = [vp]
    lib.rth_world_destroy.restype = None
    lib.rth_world_scene.argtypes = [vp]
    lib.rth_world_scene.restype = C.POINTER(Scene)
    lib.rth_world_camera.argtypes = [vp]
    lib.rth_world_camera.restype = C.POINTER(Camera)
    lib.rth_world_render_seed.argtypes = [vp]
    lib.rth_world_render_seed.restype = C.c_uint64
    lib.rth_world_bvh_first.argtypes = [vp]
    lib.rth_last_error.restype = C.c_char_p
    lib.rth_scene_names.restype = C.POINTER(C.c_char_p)
    d3 = C.POINTER(C.c_double)
    lib.rth_camera_build.argtypes = [d3, d3, d3, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                     C.c_double, C.c_double, C.POINTER(Camera)]
    _lib = lib
    return lib


def scene_names() -> list:
    """`Scene::VARIANTS` (config.rs:76) plus the FinalStatic extension."""
    lib = load()
    arr = lib.rth_scene_names()
    out, i = [], 0
    while arr[i]:
        out.append(arr[i].decode())
        i += 1
    return out


class World:
    """`Scene::create` (config.rs:104-107): camera + world for a named scene."""

    def __init__(self, name: str, seed: Optional[int] = 1, aspect: float = 1024 / 576):
        lib = load()
        h = C.c_void_p()
        rc = lib.rth_world_create(name.encode(), 0 if seed is None else 1, 0 if seed is None else seed,
                                  float(aspect), C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, lib.rth_last_error().decode())
        self._h = h
        self.name = name

    # The structs returned below are VIEWS into memory the rth_world owns: each view keeps a reference to this
    # World, so `ctx.upload(api.World(..).scene)` cannot free the world before the upload has read it.
    @property
    def scene(self) -> Scene:
        s = load().rth_world_scene(self._h).contents
        s._world = self
        return s

    @property
    def camera(self) -> Camera:
        c = load().rth_world_camera(self._h).contents
        c._world = self
        return c

    @property
    def render_seed(self) -> int:
        return int(load().rth_world_render_seed(self._h))

    @property
    def bvh_first(self) -> int:
        return int(load().rth_world_bvh_first(self._h))

    def arrays(self) -> dict:
        s = self.scene
        n = s.n_spheres
        get = lambda p, t: np.ctypeslib.as_array(p, shape=(n,)).astype(t).copy()
        return {
            "c0": np.stack([get(s.c0x, np.float32), get(s.c0y, np.float32), get(s.c0z, np.float32)], 1),
            "dc": np.stack([get(s.dcx, np.float32), get(s.dcy, np.float32), get(s.dcz, np.float32)], 1),
            "radius": get(s.radius, np.float32),
            "material_id": get(s.material_id, np.uint32),
        }

    def close(self):
        if self._h:
            load().rth_world_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FlatScene:
    """A caller-built rt_scene (the Python twin of INTEGRATION.md's `FlatScene::as_rt_scene`): spheres in world order
    as arrays, materials as (type, albedo, fuzz, ior, texture_id) tuples, textures as `Texture` structs. Keeps the
    arrays alive for as long as `.scene` is used."""

    def __init__(self, c0, radius, material_id, materials, textures=(), dc=None, noise_tables=()):
        c0 = np.asarray(c0, np.float32).reshape(-1, 3)
        n = c0.shape[0]
        dc = np.zeros((n, 3), np.float32) if dc is None else np.asarray(dc, np.float32).reshape(n, 3)
        self._cols = [np.ascontiguousarray(c0[:, k]) for k in range(3)] + \
                     [np.ascontiguousarray(dc[:, k]) for k in range(3)]
        self._radius = np.ascontiguousarray(radius, np.float32).reshape(n)
        self._mid = np.ascontiguousarray(material_id, np.uint32).reshape(n)
        self._mats = (Material * max(len(materials), 1))()
        for i, (typ, albedo, fuzz, ior, tex) in enumerate(materials):
            self._mats[i] = Material(int(typ), (C.c_float * 3)(*albedo), float(fuzz), float(ior), int(tex), 0)
        self._texs = (Texture * max(len(textures), 1))(*textures)
        self._tabs = (NoiseTable * max(len(noise_tables), 1))(*noise_tables)
        ptr = lambda a: a.ctypes.data_as(f32p)
        self.scene = Scene(n, *[ptr(a) for a in self._cols], ptr(self._radius), self._mid.ctypes.data_as(u32p),
                           len(materials), self._mats, len(textures), self._texs, len(noise_tables), self._tabs)


def solid_texture(rgb) -> Texture:
    t = Texture()
    t.kind = RT_TEX_SOLID
    t.color_a = (C.c_float * 3)(*rgb)
    t.color_b = (C.c_float * 3)(*rgb)
    return t


def camera_build(origin, look_at, view_up=None, vfov_degrees=60.0, aspect_ratio=1024 / 576, aperture=0.0,
                 focus_dist=0.0, shutter=None) -> Camera:
    """`CameraBuilder::build` (screen.rs:112-147)."""
    lib = load()
    d3 = C.c_double * 3
    cam = Camera()
    o = d3(*origin) if origin is not None else None
    l = d3(*look_at) if look_at is not None else None
    up = d3(*view_up) if view_up is not None else None
    rc = lib.rth_camera_build(o, l, up, vfov_degrees, aspect_ratio, aperture, focus_dist,
                              0 if shutter is None else 1, 0.0 if shutter is None else shutter[0],
                              0.0 if shutter is None else shutter[1], C.byref(cam))
    if rc != RT_OK:
        raise RtError(rc, lib.rth_last_error().decode())
    return cam


def make_params(width, height, spp, max_depth=100, antialias=True, seed=0, spp_offset=0, t_min=1e-3) -> Params:
    """`Config` defaults (config.rs:25-69): max_depth 100, antialias on."""
    return Params(width, height, spp, max_depth, 1 if antialias else 0, spp_offset, seed, t_min, 0)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def partition_spp(total_spp: int, n_parts: int, part: int):
    """rt_partition_spp: (first sample index, count) of `part` (pure arithmetic, no device)."""
    first, count = C.c_uint32(0), C.c_uint32(0)
    rc = load().rt_partition_spp(total_spp, n_parts, part, C.byref(first), C.byref(count))
    if rc != RT_OK:
        raise RtError(rc, "bad partition arguments")
    return int(first.value), int(count.value)


def debug_multi(flags: int = 0) -> None:
    """include/rtx_hip_debug.h rt_debug_multi: bit 0 = a device may be listed several times, bit 1 = no RCCL (the
    copy-and-add exchange step); process-wide, read by the next MultiContext."""
    rc = load().rt_debug_multi(int(flags))
    if rc != RT_OK:
        raise RtError(rc, "bad rt_debug_multi flags")


class MultiContext:
    """An rt_multi: one process, several GPUs of one node, spp-partitioned render + one RCCL reduce."""

    def __init__(self, devices: Sequence[int]):
        self.lib = load()
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        rc = self.lib.rt_multi_create(arr, len(devices), C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_multi_last_error(None).decode())
        self._h = h

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_multi_last_error(self._h).decode())

    def upload(self, scene: Scene):
        self._check(self.lib.rt_multi_upload_scene(self._h, C.byref(scene)))

    def render(self, cam: Camera, params: Params):
        out = np.empty((params.height, params.width, 3), np.float32)
        st = MultiStats()
        self._check(self.lib.rt_multi_render(self._h, C.byref(cam), C.byref(params), _ptr(out), C.byref(st)))
        return out, st

    def close(self):
        if self._h:
            self.lib.rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """An rt_ctx on one HIP device. Raises if the device or library is missing."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.rt_create(device, C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_last_error(None).decode())
        self._h = h
        self.device = device

    def _check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.lib.rt_last_error(self._h).decode())

    def upload(self, scene: Scene):
        self._check(self.lib.rt_upload_scene(self._h, C.byref(scene)))

    def render(self, cam: Camera, params: Params):
        """rt_render: returns (H, W, 3) float32 linear radiance means and Stats."""
        out = np.empty((params.height, params.width, 3), np.float32)
        st = Stats()
        self._check(self.lib.rt_render(self._h, C.byref(cam), C.byref(params), _ptr(out), C.byref(st)))
        return out, st

    def render_accum_device(self, cam: Camera, params: Params, device_ptr: int) -> Stats:
        """rt_render_accum_device: per-pixel radiance SUMS into a device fp32 buffer (e.g. a torch tensor)."""
        st = Stats()
        self._check(self.lib.rt_render_accum_device(self._h, C.byref(cam), C.byref(params),
                                                    C.c_void_p(device_ptr), C.byref(st)))
        return st

    def render_accum_device_noalloc(self, cam: Camera, params: Params, device_ptr: int) -> Stats:
        """rt_render_accum_device_noalloc: the same, with the workspace in the library's static device array."""
        st = Stats()
        self._check(self.lib.rt_render_accum_device_noalloc(self._h, C.byref(cam), C.byref(params),
                                                            C.c_void_p(device_ptr), C.byref(st)))
        return st

    def debug_set_work_units(self, unit_a: int = 0, unit_b: int = 0, tail_span: int = -1):
        """include/rtx_hip_debug.h: force the work-unit schedule (0, 0, -1 = defaults)."""
        self._check(self.lib.rt_debug_set_work_units(self._h, unit_a, unit_b, tail_span))

    def debug_set_planar_list(self, on: bool = True):
        """include/rtx_hip_debug.h: static spheres with a shared centre coordinate in their own scan list (default) or
        in the general one, from the next upload on."""
        self._check(self.lib.rt_debug_set_planar_list(self._h, 1 if on else 0))

    def debug_set_tail_order(self, on: bool = True):
        """include/rtx_hip_debug.h: the small units at the end of a launch dealt long-path tiles first (default) or in
        tile-index order."""
        self._check(self.lib.rt_debug_set_tail_order(self._h, 1 if on else 0))

    def debug_tail_order_stats(self) -> dict:
        """include/rtx_hip_debug.h: whether the last launch built a tail order and how many tiles it dealt first."""
        out = (C.c_uint32 * 3)()
        self._check(self.lib.rt_debug_tail_order_stats(self._h, C.byref(out)))
        return {"built": int(out[0]), "marked_tiles": int(out[1]), "valid": int(out[2])}

    def debug_set_direct(self, on: bool = True):
        """include/rtx_hip_debug.h: scenes of a handful of spheres entirely on the f64 list and rendered by the kernel
        variant without a scan (default), or laid out like any other scene, from the next upload on."""
        self._check(self.lib.rt_debug_set_direct(self._h, 1 if on else 0))

    def debug_scan_layout(self) -> dict:
        """include/rtx_hip_debug.h: how the last upload laid the world out for the scan.  `scan` is the render
        kernel's third template argument: 0 general lists, 1 with the planar list, 2 no scan lists at all."""
        out = (C.c_int32 * 8)()
        self._check(self.lib.rt_debug_scan_layout(self._h, C.byref(out)))
        keys = ("n_big", "static_slots", "planar_slots", "moving_slots", "planar_axis", "moving_axes", "tile_buffers", "noise")
        d = dict(zip(keys, (int(v) for v in out)))
        d["scan"] = 1 if d["planar_slots"] > 0 else 2 if d["static_slots"] + d["moving_slots"] == 0 else 0
        return d

    def debug_set_tile_buffers(self, n: int = 0):
        """include/rtx_hip_debug.h: LDS tile buffers per wave from the next upload on (0 = chosen by the library)."""
        self._check(self.lib.rt_debug_set_tile_buffers(self._h, n))

    def debug_set_scan_order(self, world_order: bool):
        self._check(self.lib.rt_debug_set_scan_order(self._h, 1 if world_order else 0))

    def intersect_batch(self, rays: np.ndarray, t_min: float = 1e-3) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=ray_dtype)
        out = np.zeros(rays.shape[0], hit_dtype)
        self._check(self.lib.rt_intersect_batch(self._h, _ptr(rays), rays.shape[0], t_min, _ptr(out)))
        return out

    def scatter_batch(self, rays: np.ndarray, hits: np.ndarray, randoms4: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=ray_dtype)
        hits = np.ascontiguousarray(hits, dtype=hit_dtype)
        u = np.ascontiguousarray(randoms4, dtype=np.float32).reshape(-1, 4)
        assert u.shape[0] == rays.shape[0] == hits.shape[0]
        out = np.zeros(rays.shape[0], scatter_dtype)
        self._check(self.lib.rt_scatter_batch(self._h, _ptr(rays), _ptr(hits), _ptr(u), rays.shape[0], _ptr(out)))
        return out

    def texture_batch(self, texture_id: int, points: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        out = np.zeros_like(p)
        self._check(self.lib.rt_texture_batch(self._h, texture_id, _ptr(p), p.shape[0], _ptr(out)))
        return out

    def camera_rays(self, cam: Camera, ij: np.ndarray, randoms4: np.ndarray) -> np.ndarray:
        ij = np.ascontiguousarray(ij, dtype=np.float32).reshape(-1, 2)
        u = np.ascontiguousarray(randoms4, dtype=np.float32).reshape(-1, 4)
        out = np.zeros(ij.shape[0], ray_dtype)
        self._check(self.lib.rt_camera_rays(self._h, C.byref(cam), _ptr(ij), _ptr(u), ij.shape[0], _ptr(out)))
        return out

    def rng_batch(self, seed: int, counters4: np.ndarray) -> np.ndarray:
        c = np.ascontiguousarray(counters4, dtype=np.uint32).reshape(-1, 4)
        out = np.zeros_like(c)
        self._check(self.lib.rt_rng_batch(self._h, seed, _ptr(c), c.shape[0], _ptr(out)))
        return out

    def encode(self, rgb: np.ndarray):
        """`Screen::encode` (screen.rs:26-45): returns (0RGB uint32 array, n_invalid)."""
        a = np.ascontiguousarray(rgb, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(a.shape[0], np.uint32)
        bad = C.c_uint32(0)
        self._check(self.lib.rt_encode(self._h, _ptr(a), a.shape[0], _ptr(out), C.byref(bad)))
        return out, int(bad.value)

    def close(self):
        if self._h:
            self.lib.rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
