//! Raw declarations of `include/rrt.h` and `include/rrt_host.h` (librrt.so).
//! UNVERIFIED SOURCE: the build image has no Rust toolchain (see ../README.md).
//! Layouts mirror the C structs field for field; keep them in step with the headers.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

/// Opaque scene handle (`typedef struct rrt_scene rrt_scene`).
#[repr(C)]
pub struct rrt_scene {
    _private: [u8; 0],
}

/// Opaque communicator of the tile gather (`typedef struct rrt_comm rrt_comm`).
#[repr(C)]
pub struct rrt_comm {
    _private: [u8; 0],
}

/// `rrt_camera`: `RayTracer::new_default_renderer` + the lens set-up of `render`
/// (src/tracer.rs:38-51, 84-110).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_camera {
    pub eye: [f64; 3],
    pub lens_z: f64,
    pub lens_w: f64,
    pub lens_h: f64,
}

/// `rrt_settings`: `RenderSettings` (src/tracer.rs:16-18) plus the extensions.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_settings {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub max_bounce: u32,
    pub seed: u64,
    pub prune_zero_mix: u32,
    pub tile_rank: u32,
    pub tile_world: u32,
    pub batch_spp: u32,
    pub tile_deal: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rrt_stats {
    pub rays_traced: u64,
    pub primary_rays: u64,
    pub paths: u64,
    pub traversal_overflow: u64,
    pub gpu_ms: f64,
    pub primary_kernel_ms: f64,
    pub bounce_kernel_ms: f64,
    pub primary_kernel_launches: u32,
    pub bounce_kernel_launches: u32,
    pub n_batches: u32,
    pub reserved: u32,
    pub level_rays: [u64; 4],
    pub cull_violations: u64,
    pub zero_mix_paths: u64,
    pub sky_kernel_ms: f64,
    pub heavy_tiles: u32,
    pub sky_tiles: u32,
}

pub const RRT_TILE_W: u32 = 32;
pub const RRT_TILE_H: u32 = 8;

extern "C" {
    pub fn rrt_last_error() -> *const c_char;
    pub fn rrt_version() -> *const c_char;
    pub fn rrt_set_device(device: c_int) -> c_int;
    pub fn rrt_device_image_support() -> c_int;
    pub fn rrt_default_camera(cam: *mut rrt_camera);
    pub fn rrt_default_settings(st: *mut rrt_settings, w: u32, h: u32);

    pub fn rrt_scene_create() -> *mut rrt_scene;
    pub fn rrt_scene_destroy(s: *mut rrt_scene);
    pub fn rrt_add_material(s: *mut rrt_scene, color: *const f64, albedo: f64, lambert_mix: f64,
                            reflective_mix: f64, phong_mix: f64, alpha: i32) -> c_int;
    /// Extension lobes (no reference counterpart): 0 reflection, 1 refraction(ior), 2 diffuse bounce.
    pub fn rrt_add_material_ext(s: *mut rrt_scene, color: *const f64, albedo: f64, lambert_mix: f64,
                                lobe_mix: f64, phong_mix: f64, alpha: i32, lobe: c_int, ior: f64) -> c_int;
    pub fn rrt_add_sphere(s: *mut rrt_scene, center: *const f64, radius: f64, material: c_int) -> c_int;
    pub fn rrt_add_mesh(s: *mut rrt_scene, verts: *const f64, normals: *const f64, n_triangles: i64,
                        material: c_int) -> c_int;
    /// Triangle::new + push: a bare triangle as an entity of its own (triangle.rs:19-58), no kd-tree around it.
    pub fn rrt_add_triangle(s: *mut rrt_scene, v0: *const f64, v1: *const f64, v2: *const f64, normal: *const f64,
                            material: c_int) -> c_int;
    pub fn rrt_add_group(s: *mut rrt_scene, members: *const i32, n_members: i32) -> c_int;
    pub fn rrt_set_sky_face(s: *mut rrt_scene, face: c_int, rgb8: *const u8, w: u32, h: u32) -> c_int;
    pub fn rrt_set_directional_light(s: *mut rrt_scene, direction: *const f64, color: *const f64,
                                     intensity: f64) -> c_int;
    /// DirectionalLight::new + world.light_sources.push: appends (the set_ variant makes the list [this light]).
    pub fn rrt_add_directional_light(s: *mut rrt_scene, direction: *const f64, color: *const f64,
                                     intensity: f64) -> c_int;
    pub fn rrt_add_point_light(s: *mut rrt_scene, position: *const f64, color: *const f64, brightness: f64,
                               attenuation: f64) -> c_int;
    pub fn rrt_set_shadow_mode(s: *mut rrt_scene, mode: c_int) -> c_int;
    pub fn rrt_translate_entity(s: *mut rrt_scene, entity: c_int, delta: *const f64, refit: c_int) -> c_int;
    pub fn rrt_occluded_rays(s: *mut rrt_scene, rays: *const f64, max_dist2: *const f64, n: i64, out: *mut u8) -> c_int;
    pub fn rrt_set_ambient(s: *mut rrt_scene, ambient: f64) -> c_int;
    pub fn rrt_set_work_budget(s: *mut rrt_scene, bytes: u64) -> c_int;
    pub fn rrt_set_cull_mode(s: *mut rrt_scene, mode: c_int) -> c_int;
    pub fn rrt_set_timing(s: *mut rrt_scene, level: c_int) -> c_int;
    pub fn rrt_set_sky_stream(s: *mut rrt_scene, mode: c_int) -> c_int;
    /// Inspection of a primary-ray coverage mask: one byte per pixel (1 = some triangle of the mesh may accept a ray
    /// through it) and the "every pixel counts as set" flag.
    pub fn rrt_cover_mask(s: *mut rrt_scene, cam: *const rrt_camera, width: u32, height: u32, entity: c_int,
                          pixels: *mut u8, every: *mut c_int) -> c_int;
    pub fn rrt_scene_commit(s: *mut rrt_scene) -> c_int;

    pub fn rrt_render(s: *mut rrt_scene, cam: *const rrt_camera, st: *const rrt_settings, out_rgb8: *mut u8,
                      out_ids: *mut i32, out_t: *mut f64, out_color: *mut f64) -> c_int;
    pub fn rrt_render_device(s: *mut rrt_scene, cam: *const rrt_camera, st: *const rrt_settings,
                             d_rgb8: *mut c_void, d_ids: *mut c_void, d_t: *mut c_void, d_color: *mut c_void,
                             stream: *mut c_void) -> c_int;
    pub fn rrt_render_device_noalloc(s: *mut rrt_scene, cam: *const rrt_camera, st: *const rrt_settings,
                                     d_rgb8: *mut c_void, d_ids: *mut c_void, d_t: *mut c_void, d_color: *mut c_void,
                                     stream: *mut c_void) -> c_int;
    pub fn rrt_noalloc_scratch_bytes() -> u64;
    /// Page-locked host memory: rrt_render fills such a destination band by band while the frame is being resolved.
    pub fn rrt_alloc_host_pinned(bytes: u64) -> *mut c_void;
    pub fn rrt_free_host_pinned(p: *mut c_void);
    /// Gives back the work buffers the scene keeps for `stream` (call before destroying a stream it rendered on).
    pub fn rrt_release_stream(s: *mut rrt_scene, stream: *mut c_void) -> c_int;
    pub fn rrt_tile_skew(width: u32, tile_world: u32) -> u32;
    pub fn rrt_comm_unique_id(id128: *mut c_void) -> c_int;
    pub fn rrt_comm_init(rank: c_int, world: c_int, id128: *const c_void, out: *mut *mut rrt_comm) -> c_int;
    pub fn rrt_comm_destroy(c: *mut rrt_comm);
    pub fn rrt_gather_tiles(c: *mut rrt_comm, d_local: *const c_void, d_gathered: *mut c_void, width: u32, height: u32,
                            root: c_int, stream: *mut c_void) -> c_int;
    pub fn rrt_tile_slots(width: u32, height: u32, tile_world: u32) -> i64;
    /// Bytes of a rank's compact tile buffer (with tile_deal: + its int32 slot -> tile table).
    pub fn rrt_tile_buffer_bytes(width: u32, height: u32, tile_world: u32, tile_deal: u32) -> u64;
    pub fn rrt_gather_tiles_dealt(c: *mut rrt_comm, d_local: *const c_void, d_gathered: *mut c_void, width: u32, height: u32,
                                  root: c_int, stream: *mut c_void) -> c_int;
    pub fn rrt_assemble_tiles_dealt(d_gathered: *const c_void, width: u32, height: u32, tile_world: u32,
                                    d_rgb8: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rrt_assemble_tiles(d_gathered: *const c_void, width: u32, height: u32, tile_world: u32,
                              d_rgb8: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rrt_trace_rays(s: *mut rrt_scene, rays: *const f64, n: i64, ent: *mut i32, tri: *mut i32,
                          t: *mut f64, pos: *mut f64, normal: *mut f64) -> c_int;
    pub fn rrt_cast_rays(s: *mut rrt_scene, rays: *const f64, n: i64, max_bounce: u32, prune_zero_mix: u32,
                         color: *mut f64) -> c_int;
    pub fn rrt_get_stats(s: *const rrt_scene, out: *mut rrt_stats) -> c_int;

    // include/rrt_host.h: the asset formats either side of the path
    pub fn rrt_host_load_model(obj_path: *const c_char, position: *const f64, scale: *const f64,
                               verts: *mut *mut f64, normals: *mut *mut f64, n_triangles: *mut i64) -> c_int;
    pub fn rrt_host_add_model(s: *mut rrt_scene, obj_path: *const c_char, material: c_int,
                              position: *const f64, scale: *const f64) -> c_int;
    pub fn rrt_host_load_model_ex(obj_path: *const c_char, position: *const f64, scale: *const f64, flags: u32,
                                  verts: *mut *mut f64, normals: *mut *mut f64, n_triangles: *mut i64) -> c_int;
    pub fn rrt_host_add_model_ex(s: *mut rrt_scene, obj_path: *const c_char, material: c_int,
                                 position: *const f64, scale: *const f64, flags: u32) -> c_int;
    pub fn rrt_host_load_png_rgb8(path: *const c_char, rgb8: *mut *mut u8, w: *mut u32, h: *mut u32) -> c_int;
    pub fn rrt_host_save_png_rgb8(path: *const c_char, rgb8: *const u8, w: u32, h: u32) -> c_int;
    pub fn rrt_host_load_cubemap(s: *mut rrt_scene, folder: *const c_char) -> c_int;
    pub fn rrt_host_free(p: *mut c_void);
}
