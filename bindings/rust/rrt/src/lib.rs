//! Safe wrapper over `rrt-sys` that keeps the reference's operator API for the tracing path.
//! UNVERIFIED SOURCE: the build image has no Rust toolchain (see ../README.md).
//!
//! The reference's `World` holds `Vec<Box<dyn Entity>>` (src/common.rs:10-15) and every material
//! a `Vec<Arc<dyn RayBehavior>>` (src/material.rs:13-18): open sets a GPU cannot run.  Here the
//! same constructors build plain descriptions (`enum Entity`, what `entity_enum!` /
//! `#[derive(Entity)]` generate for `Scene<T>`, derive_entity/src/lib.rs:7-51), and
//! `RayTracer::render` hands them to librrt.so in `world.entities` push order, so entity ids in
//! the AOV equal the reference's indices.
//!
//! `main.rs:22-65` against this crate:
//! ```ignore
//! let rt = RayTracer::new_default_renderer((3840, 2160));
//! let mut world = RayTracer::new_empty_world("hd_blue_sunset");
//! let mat = Material::new_lambert_material(color_vec(100.0, 100.0, 200.0), 0.8, 1.0, 0.01, 0.1, 20);
//! let mat2 = Material::new_lambert_material(color_vec(0.0, 0.0, 0.0), 0.8, 0.0, 1.0, 0.1, 20);
//! let ufo_mat = Material::new_lambert_material(color_vec(100.0, 100.0, 50.0), 1.0, 1.0, 0.0, 0.3, 20);
//! world.entities.push(Entity::Model(Model::new("./obj/ufo_fix.obj", ufo_mat, [0.0, 30.0, 70.0], [1.0, -1.0, 1.0])));
//! world.entities.push(Entity::Sphere(Sphere::new([-3.0, 0.0, 5.0], 1.0, mat)));
//! world.entities.push(Entity::Sphere(Sphere::new([2.0, 0.0, 8.0], 1.0, mat2)));
//! rt.render("bruh.png".to_string(), world).unwrap();
//! ```
use std::ffi::{CStr, CString};
use std::ptr;

use rrt_sys as sys;

pub type Vec3 = [f64; 3];

/// `color_vec` (src/common.rs:88-93): 8-bit colour to the unit cube.
pub fn color_vec(r: f64, g: f64, b: f64) -> Vec3 {
    [r / 255.0, g / 255.0, b / 255.0]
}

#[derive(Debug)]
pub struct Error(pub i32, pub String);

fn check(rc: i32) -> Result<i32, Error> {
    if rc < 0 {
        let msg = unsafe { CStr::from_ptr(sys::rrt_last_error()) }.to_string_lossy().into_owned();
        Err(Error(rc, msg))
    } else {
        Ok(rc)
    }
}

/// The continuation lobe in the Reflection slot of `new_lambert_material`'s shader list.
#[derive(Clone, Copy, Debug, PartialEq)]
pub enum Lobe {
    /// `ReflectionBehavior` (src/behavior/reflection.rs:19-44): the reference.
    Reflection,
    /// Extension: Snell refraction with index `ior`.
    Refraction(f64),
    /// Extension: stochastic diffuse bounce (counter-based RNG).
    DiffuseBounce,
}

/// `Material` (src/material.rs:13-43), closed over the shaders `new_lambert_material` installs.
#[derive(Clone, Copy, Debug)]
pub struct Material {
    pub color: Vec3,
    pub albedo: f64,
    pub lambert_mix: f64,
    pub reflective_mix: f64,
    pub phong_mix: f64,
    pub alpha: i32,
    pub lobe: Lobe,
}

impl Material {
    /// Same argument order as src/material.rs:20.
    pub fn new_lambert_material(color: Vec3, albedo: f64, lambert_mix: f64, reflective_mix: f64, phong_mix: f64,
                                alpha: i32) -> Material {
        Material { color, albedo, lambert_mix, reflective_mix, phong_mix, alpha, lobe: Lobe::Reflection }
    }

    pub fn with_lobe(mut self, lobe: Lobe) -> Material {
        self.lobe = lobe;
        self
    }

    fn add_to(&self, s: *mut sys::rrt_scene) -> Result<i32, Error> {
        let (lobe, ior) = match self.lobe {
            Lobe::Reflection => (0, 1.0),
            Lobe::Refraction(ior) => (1, ior),
            Lobe::DiffuseBounce => (2, 1.0),
        };
        check(unsafe {
            if lobe == 0 {
                sys::rrt_add_material(s, self.color.as_ptr(), self.albedo, self.lambert_mix, self.reflective_mix,
                                      self.phong_mix, self.alpha)
            } else {
                sys::rrt_add_material_ext(s, self.color.as_ptr(), self.albedo, self.lambert_mix,
                                          self.reflective_mix, self.phong_mix, self.alpha, lobe, ior)
            }
        })
    }
}

/// `Sphere::new(center, radius, material)` (src/geometry/sphere.rs:17-24).
#[derive(Clone, Copy, Debug)]
pub struct Sphere {
    pub center: Vec3,
    pub radius: f64,
    pub material: Material,
}

impl Sphere {
    pub fn new(center: Vec3, radius: f64, material: Material) -> Sphere {
        Sphere { center, radius, material }
    }
}

/// `Model::new(path, material, position, scale)` (src/geometry/model.rs:24-56).  The OBJ is read
/// by the library's host side with obj-rs semantics (`rrt_host_add_model`); the kd-tree is built
/// inside with the reference's algorithm.
#[derive(Clone, Debug)]
pub struct Model {
    pub path: String,
    pub material: Material,
    pub position: Vec3,
    pub scale: Vec3,
}

impl Model {
    pub fn new(path: &str, material: Material, position: Vec3, scale: Vec3) -> Model {
        Model { path: path.to_string(), material, position, scale }
    }
}

/// `Scene::new(models, position)` (src/geometry/scene.rs:31-47): a kd-tree over whole entities.
/// Members are spheres and models (groups do not nest on the GPU path).  `position` is kept as the
/// reference keeps it (scene.rs:45: stored, never read by `collide`).
#[derive(Clone, Debug)]
pub struct Scene {
    pub members: Vec<Entity>,
    pub position: Vec3,
}

impl Scene {
    pub fn new(members: Vec<Entity>, position: Vec3) -> Scene {
        Scene { members, position }
    }
}

/// `entity_enum!(Name, Type1, Type2, ...)` (src/geometry/scene.rs:12-21).  The reference generates an enum
/// with one variant per listed type and derives `Entity` for it (derive_entity/src/lib.rs:7-51: every method
/// forwards to the variant).  The GPU path has one closed set, so the name becomes an alias of it and the
/// listed types are only checked to be variants.
#[macro_export]
macro_rules! entity_enum {
    ($name:ident, $($x:ident),+) => {
        #[allow(dead_code)]
        type $name = $crate::Entity;
        $( const _: fn($crate::$x) -> $crate::Entity = $crate::Entity::$x; )+
    };
}

/// `scene!(EnumType, position, Type1(obj1), Type2(obj2), ...)` (src/geometry/scene.rs:23-29), so that
/// main.rs:57-59 reads as it does in the reference.
#[macro_export]
macro_rules! scene {
    ($type_name:ident, $position:expr, $($x:ident($y:ident)),+) => {
        $crate::Scene::new(vec![$($type_name::$x($y)),+], $position)
    };
}

/// The closed entity set (`entity_enum!` in the reference, src/geometry/scene.rs:54-75).
#[derive(Clone, Debug)]
pub enum Entity {
    Sphere(Sphere),
    Model(Model),
    Scene(Scene),
}

/// `DirectionalLight::new(direction, color, intensity)` (src/lighting.rs:25-32).
#[derive(Clone, Copy, Debug)]
pub struct DirectionalLight {
    pub direction: Vec3,
    pub color: Vec3,
    pub intensity: f64,
}

/// `PointLight::new(position, color, brightness, radius)` (src/lighting.rs:52-68).  `render` never reaches
/// it in the reference (World::light_sources holds DirectionalLights only); here it is pushed behind the
/// directional light (`rrt_add_point_light`), with `World::shadow_mode` choosing `visible()` as written
/// (0, lighting.rs:70-84) or the corrected occlusion test (1).
#[derive(Clone, Copy, Debug)]
pub struct PointLight {
    pub position: Vec3,
    pub color: Vec3,
    pub brightness: f64,
    pub radius: f64,
}

impl PointLight {
    pub fn new(position: Vec3, color: Vec3, brightness: f64, radius: f64) -> PointLight {
        PointLight { position, color, brightness, radius }
    }
}

/// `World` (src/common.rs:10-15).
#[derive(Clone, Debug)]
pub struct World {
    pub entities: Vec<Entity>,
    pub light_sources: Vec<DirectionalLight>,
    pub point_lights: Vec<PointLight>,
    pub shadow_mode: i32,
    /// Folder of the six cubemap PNGs (`Material::new_sky_material`, src/material.rs:35-42).
    pub sky: String,
    pub ambient: f64,
}

/// Extensions of `RenderSettings`; the defaults reproduce the reference.
#[derive(Clone, Copy, Debug)]
pub struct RenderSettings {
    pub image_size: (u32, u32),
    pub spp: u32,
    pub max_bounce: u32,
    pub seed: u64,
}

pub struct RayTracer {
    pub settings: RenderSettings,
    pub camera: sys::rrt_camera,
}

struct SceneHandle(*mut sys::rrt_scene);

impl Drop for SceneHandle {
    fn drop(&mut self) {
        unsafe { sys::rrt_scene_destroy(self.0) }
    }
}

impl RayTracer {
    /// src/tracer.rs:38-51.
    pub fn new_default_renderer(size: (u32, u32)) -> RayTracer {
        let mut cam = sys::rrt_camera { eye: [0.0; 3], lens_z: 0.0, lens_w: 0.0, lens_h: 0.0 };
        unsafe { sys::rrt_default_camera(&mut cam) };
        RayTracer { settings: RenderSettings { image_size: size, spp: 1, max_bounce: 2, seed: 0 }, camera: cam }
    }

    /// src/tracer.rs:53-75: one directional light (1,-1,1), colour 230/255, intensity 2.0; ambient 0.15.
    pub fn new_empty_world(skybox: &str) -> World {
        World {
            entities: Vec::new(),
            light_sources: vec![DirectionalLight {
                direction: [1.0, -1.0, 1.0],
                color: color_vec(230.0, 230.0, 230.0),
                intensity: 2.0,
            }],
            point_lights: Vec::new(),
            shadow_mode: 0,
            sky: format!("./cubemaps/{}", skybox),
            ambient: 0.15,
        }
    }

    fn add_entity(s: *mut sys::rrt_scene, e: &Entity, in_group: bool) -> Result<i32, Error> {
        match e {
            Entity::Sphere(sp) => {
                let m = sp.material.add_to(s)?;
                check(unsafe { sys::rrt_add_sphere(s, sp.center.as_ptr(), sp.radius, m) })
            }
            Entity::Model(md) => {
                let m = md.material.add_to(s)?;
                let path = CString::new(md.path.as_str()).map_err(|_| Error(-1, "path contains NUL".into()))?;
                check(unsafe { sys::rrt_host_add_model(s, path.as_ptr(), m, md.position.as_ptr(), md.scale.as_ptr()) })
            }
            Entity::Scene(sc) => {
                if in_group {
                    return Err(Error(-1, "Scene<Scene<..>> is not supported: groups do not nest".into()));
                }
                let mut ids: Vec<i32> = Vec::with_capacity(sc.members.len());
                for m in &sc.members {
                    ids.push(Self::add_entity(s, m, true)?);
                }
                // the members become the group's; the group is the world entity (main.rs:57-59)
                check(unsafe { sys::rrt_add_group(s, ids.as_ptr(), ids.len() as i32) })
            }
        }
    }

    fn build(world: &World) -> Result<SceneHandle, Error> {
        let h = SceneHandle(unsafe { sys::rrt_scene_create() });
        if h.0.is_null() {
            return Err(Error(-1, "rrt_scene_create failed".into()));
        }
        let sky = CString::new(world.sky.as_str()).map_err(|_| Error(-1, "path contains NUL".into()))?;
        check(unsafe { sys::rrt_host_load_cubemap(h.0, sky.as_ptr()) })?;
        if let Some(l) = world.light_sources.first() {
            check(unsafe { sys::rrt_set_directional_light(h.0, l.direction.as_ptr(), l.color.as_ptr(), l.intensity) })?;
        }
        check(unsafe { sys::rrt_set_ambient(h.0, world.ambient) })?;
        for l in &world.point_lights {
            check(unsafe { sys::rrt_add_point_light(h.0, l.position.as_ptr(), l.color.as_ptr(), l.brightness, l.radius) })?;
        }
        check(unsafe { sys::rrt_set_shadow_mode(h.0, world.shadow_mode) })?;
        for e in &world.entities {
            Self::add_entity(h.0, e, false)?;
        }
        check(unsafe { sys::rrt_scene_commit(h.0) })?;
        Ok(h)
    }

    /// `render(self, output, world)` (src/tracer.rs:77-175): consumes both, writes the PNG.
    pub fn render(self, output: String, world: World) -> Result<(), Error> {
        let (w, h) = self.settings.image_size;
        let scene = Self::build(&world)?;
        let mut st = sys::rrt_settings {
            width: 0, height: 0, spp: 0, max_bounce: 0, seed: 0, prune_zero_mix: 0, tile_rank: 0, tile_world: 0,
            batch_spp: 0, tile_deal: 0, reserved: 0,
        };
        unsafe { sys::rrt_default_settings(&mut st, w, h) };
        st.spp = self.settings.spp;
        st.max_bounce = self.settings.max_bounce;
        st.seed = self.settings.seed;
        let mut rgb = vec![0u8; w as usize * h as usize * 3];
        check(unsafe {
            sys::rrt_render(scene.0, &self.camera, &st, rgb.as_mut_ptr(), ptr::null_mut(), ptr::null_mut(),
                            ptr::null_mut())
        })?;
        let out = CString::new(output).map_err(|_| Error(-1, "path contains NUL".into()))?;
        check(unsafe { sys::rrt_host_save_png_rgb8(out.as_ptr(), rgb.as_ptr(), w, h) })?;
        Ok(())
    }
}
