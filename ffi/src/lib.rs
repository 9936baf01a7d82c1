//! raytracer-amd-sys — the binding a Rust host (doxxx/raytracer) adds to call the MI355X rendering core.
//!
//! The reference has no FFI; the seam is `system::render(options, scene, &mut progress)`
//! (`src/main.rs:106` -> `src/system.rs:263-307`).  This crate declares `include/rt_api.h` in Rust
//! (`#[repr(C)]` structs, one `extern "C"` block), wraps scene handles and render calls in safe types, and
//! defines the one trait method the host's `Shape` / `Material` implementors gain (`Flatten`), because their
//! fields are private (`src/shapes/*.rs`, `src/materials/*.rs`).
//!
//! NOT COMPILED IN THIS REPOSITORY'S IMAGE (no cargo / rustc).  `tests/test_host_logic.py` parses the
//! `#[repr(C)]` declarations below and checks every size and field offset against what the C compiler makes
//! of `include/rt_api.h`.

#![allow(non_camel_case_types)]

use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::ptr;

// ------------------------------------------------------------------------------------------------ status
pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_PARSE: c_int = -2;
pub const RT_ERR_IO: c_int = -3;
pub const RT_ERR_DEVICE: c_int = -4;
pub const RT_ERR_UNSUPPORTED: c_int = -5;

// rt_shape_kind
pub const RT_SHAPE_SPHERE: i32 = 0;
pub const RT_SHAPE_PLANE: i32 = 1;
pub const RT_SHAPE_XYRECT: i32 = 2;
pub const RT_SHAPE_XZRECT: i32 = 3;
pub const RT_SHAPE_ZYRECT: i32 = 4;
pub const RT_SHAPE_CUBE: i32 = 5;
pub const RT_SHAPE_CYLINDER: i32 = 6;
pub const RT_SHAPE_TORUS: i32 = 7;
pub const RT_SHAPE_UNION: i32 = 8;
pub const RT_SHAPE_INTERSECTION: i32 = 9;
pub const RT_SHAPE_DIFFERENCE: i32 = 10;
pub const RT_SHAPE_COMPOSITE: i32 = 11;
pub const RT_SHAPE_MESH: i32 = 12;
pub const RT_SHAPE_MEDIUM: i32 = 13;

// rt_material_kind
pub const RT_MAT_LAMBERTIAN: i32 = 0;
pub const RT_MAT_METAL: i32 = 1;
pub const RT_MAT_DIELECTRIC: i32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: i32 = 3;
pub const RT_MAT_ISOTROPIC: i32 = 4;

// rt_texture_kind
pub const RT_TEX_SOLID: i32 = 0;
pub const RT_TEX_CHECKERBOARD: i32 = 1;
pub const RT_TEX_IMAGE: i32 = 2;

// rt_options.flags
pub const RT_FLAG_ROW_MAJOR: u32 = 1;
pub const RT_FLAG_STAGE_TIMES: u32 = 2;

pub const RT_NOALLOC_MAX_SLOTS: u32 = 84 * 4096;
pub const RT_NOALLOC_MAX_DEPTH: u32 = 64;

// ------------------------------------------------------------------------------------------------ PODs
// Field for field include/rt_api.h; the layout test reads these declarations.

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_shape_node {
    pub kind: i32,
    pub child_a: i32,
    pub child_b: i32,
    pub flags: i32,
    pub params: [f64; 8],
    pub o2w: [f64; 16],
    pub w2o: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_material {
    pub kind: i32,
    pub texture: i32,
    pub fuzz: f64,
    pub ior: f64,
    pub intensity: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_texture {
    pub kind: i32,
    pub image: i32,
    pub c1: [f64; 3],
    pub c2: [f64; 3],
    pub scale: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_image {
    pub width: u32,
    pub height: u32,
    pub offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_mesh {
    pub first_vertex: u32,
    pub n_vertices: u32,
    pub first_normal: u32,
    pub n_normals: u32,
    pub first_triangle: u32,
    pub n_triangles: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_object {
    pub shape: i32,
    pub material: i32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_camera {
    pub fov_factor: f64,
    pub camera_to_world: [f64; 16],
}

#[repr(C)]
pub struct rt_scene_desc {
    pub camera: rt_camera,
    pub background: [f64; 3],
    pub n_objects: u32,
    pub objects: *const rt_object,
    pub n_nodes: u32,
    pub nodes: *const rt_shape_node,
    pub n_child_list: u32,
    pub child_list: *const i32,
    pub n_materials: u32,
    pub materials: *const rt_material,
    pub n_textures: u32,
    pub textures: *const rt_texture,
    pub n_images: u32,
    pub images: *const rt_image,
    pub n_image_bytes: u64,
    pub image_data: *const u8,
    pub n_meshes: u32,
    pub meshes: *const rt_mesh,
    pub n_vertices: u32,
    pub vertices: *const f64,
    pub n_normals: u32,
    pub normals: *const f64,
    pub n_triangles: u32,
    pub tri_v: *const u32,
    pub tri_n: *const u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_options {
    pub width: u32,
    pub height: u32,
    pub samples: u32,
    pub max_depth: u32,
    pub bias: f64,
    pub seed: u64,
    pub tile_size: u32,
    pub shard_rank: u32,
    pub shard_count: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub rays: u64,
    pub primary_rays: u64,
    pub strata: u32,
    pub bounce_passes: u32,
    pub guards: [u32; 5],
    pub interval_overflow: u32,
    pub ms_device: f64,
    pub ms_raygen: f64,
    pub ms_intersect: f64,
    pub ms_shade: f64,
    pub ms_mesh: f64,
    pub ms_tail: f64,
    pub n_intersect_launches: u32,
    pub n_mesh_launches: u32,
    pub mesh_rays: u64,
    pub intersect_rays: u64,
    pub mesh_records: u64,
    pub mesh_one_rays: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_hit {
    pub t: f64,
    pub n: [f64; 3],
    pub uv: [f64; 2],
    pub object: i32,
    pub pad: i32,
}

pub type rt_progress_cb =
    Option<unsafe extern "C" fn(user: *mut c_void, strata_done: u32, strata_total: u32, rgb_sum: *const f64) -> c_int>;
pub type rt_stratum_cb = Option<unsafe extern "C" fn(user: *mut c_void, strata_done: u32, strata_total: u32) -> c_int>;
pub type rt_snapshot_cb = Option<
    unsafe extern "C" fn(user: *mut c_void, strata_done: u32, strata_total: u32, rgb8: *const u8, n_pixels: u64) -> c_int,
>;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rt_progress {
    pub on_stratum: rt_stratum_cb,
    pub on_snapshot: rt_snapshot_cb,
    pub min_interval_ms: u32,
    pub reserved: u32,
    pub user: *mut c_void,
}

/// Opaque `rt_scene`.
#[repr(C)]
pub struct rt_scene {
    _private: [u8; 0],
}

extern "C" {
    pub fn rt_scene_create(desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene);
    pub fn rt_render(
        scene: *mut rt_scene,
        opts: *const rt_options,
        rgb_sum: *mut f64,
        primary_ids: *mut i32,
        stats: *mut rt_stats,
        cb: rt_progress_cb,
        user: *mut c_void,
    ) -> c_int;
    pub fn rt_render_device(
        scene: *mut rt_scene,
        opts: *const rt_options,
        d_rgb_sum: *mut f64,
        d_primary_ids: *mut i32,
        stats: *mut rt_stats,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_render_device_progress(
        scene: *mut rt_scene,
        opts: *const rt_options,
        d_rgb_sum: *mut f64,
        d_primary_ids: *mut i32,
        stats: *mut rt_stats,
        stream: *mut c_void,
        progress: *const rt_progress,
    ) -> c_int;
    pub fn rt_render_multi(
        scenes: *const *mut rt_scene,
        n_devices: u32,
        opts: *const rt_options,
        rgb_sum: *mut f64,
        primary_ids: *mut i32,
        stats: *mut rt_stats,
        progress: *const rt_progress,
    ) -> c_int;
    pub fn rt_unshard_rgb8_device(
        opts: *const rt_options,
        shard_rank: u32,
        d_shard_rgb8: *const u8,
        d_full_rgb8: *mut u8,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_render_rgb8(
        scene: *mut rt_scene,
        opts: *const rt_options,
        rgb8: *mut u8,
        primary_ids: *mut i32,
        stats: *mut rt_stats,
        progress: *const rt_progress,
    ) -> c_int;
    pub fn rt_render_multi_rgb8(
        scenes: *const *mut rt_scene,
        n_devices: u32,
        opts: *const rt_options,
        rgb8: *mut u8,
        primary_ids: *mut i32,
        stats: *mut rt_stats,
        progress: *const rt_progress,
    ) -> c_int;
    pub fn rt_pinned_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn rt_pinned_free(p: *mut c_void);
    pub fn rt_render_device_noalloc(
        scene: *mut rt_scene,
        opts: *const rt_options,
        d_rgb_sum: *mut f64,
        d_primary_ids: *mut i32,
        stats: *mut rt_stats,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_shard_pixels(opts: *const rt_options) -> u64;
    pub fn rt_shard_map(opts: *const rt_options, pixel_index: *mut u32) -> c_int;
    pub fn rt_unshard_device(
        opts: *const rt_options,
        shard_rank: u32,
        d_shard_rgb: *const f64,
        d_full_rgb: *mut f64,
        d_shard_ids: *const i32,
        d_full_ids: *mut i32,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rt_trace_rays(
        scene: *mut rt_scene,
        seed: u64,
        n: u64,
        origins: *const f64,
        directions: *const f64,
        hits: *mut rt_hit,
        stats: *mut rt_stats,
    ) -> c_int;
    pub fn rt_tonemap(rgb_sum: *const f64, n_pixels: u64, divisor: f64, rgb8: *mut u8) -> c_int;
    pub fn rt_tonemap_device(d_rgb_sum: *const f64, n_pixels: u64, divisor: f64, d_rgb8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rt_write_png(path: *const c_char, rgb8: *const u8, width: u32, height: u32) -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_sdl_parse(text: *const c_char, base_dir: *const c_char, width: u32, height: u32, out: *mut *mut rt_scene_desc) -> c_int;
    pub fn rt_sdl_parse_file(path: *const c_char, base_dir: *const c_char, width: u32, height: u32, out: *mut *mut rt_scene_desc) -> c_int;
    pub fn rt_scene_desc_free(desc: *mut rt_scene_desc);
}

// ------------------------------------------------------------------------------------------------ errors
/// A non-zero `rt_status` with the library's thread-local message.
#[derive(Debug, Clone)]
pub struct Error {
    pub status: c_int,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "librt_amd status {}: {}", self.status, self.message)
    }
}

impl std::error::Error for Error {}

fn check(status: c_int) -> Result<(), Error> {
    if status == RT_OK {
        return Ok(());
    }
    let message = unsafe {
        let p = rt_last_error();
        if p.is_null() {
            String::new()
        } else {
            CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    };
    Err(Error { status, message })
}

// ------------------------------------------------------------------------------------------------ flatten
/// What the host's scene walk fills: growable pools in the layout `rt_scene_desc` points into.
/// Object index = position in `objects` = primary-hit id (the reference's `scene.objects` order).
#[derive(Default)]
pub struct FlatScene {
    pub camera_fov_factor: f64,
    pub camera_to_world: [f64; 16],
    pub background: [f64; 3],
    pub objects: Vec<rt_object>,
    pub nodes: Vec<rt_shape_node>,
    pub child_list: Vec<i32>,
    pub materials: Vec<rt_material>,
    pub textures: Vec<rt_texture>,
    pub images: Vec<rt_image>,
    pub image_data: Vec<u8>,
    pub meshes: Vec<rt_mesh>,
    pub vertices: Vec<f64>,
    pub normals: Vec<f64>,
    pub tri_v: Vec<u32>,
    pub tri_n: Vec<u32>,
}

impl FlatScene {
    /// Appends a node and returns its index.  `o2w` / `w2o` are `Transformation::object_to_world` /
    /// `world_to_object` (`src/object.rs:7-26`) as row-major `[f64; 16]`.
    pub fn push_node(&mut self, kind: i32, params: &[f64], child_a: i32, child_b: i32, flags: i32, o2w: [f64; 16], w2o: [f64; 16]) -> i32 {
        let mut p = [0.0f64; 8];
        p[..params.len()].copy_from_slice(params);
        self.nodes.push(rt_shape_node { kind, child_a, child_b, flags, params: p, o2w, w2o });
        (self.nodes.len() - 1) as i32
    }

    /// RGB8 texels of a `Texture::Image` (`image.to_rgb()`), rows top to bottom; returns the image index.
    pub fn push_image(&mut self, width: u32, height: u32, rgb8: &[u8]) -> i32 {
        assert_eq!(rgb8.len(), (width as usize) * (height as usize) * 3);
        self.images.push(rt_image { width, height, offset: self.image_data.len() as u64 });
        self.image_data.extend_from_slice(rgb8);
        (self.images.len() - 1) as i32
    }

    /// One `Mesh` (`src/shapes/mesh.rs:11-52`): vertices, normals, and per triangle three vertex and three
    /// normal indices relative to this mesh; returns the mesh index for the node's `child_a`.
    pub fn push_mesh(&mut self, vertices: &[[f64; 3]], normals: &[[f64; 3]], tri_v: &[[u32; 3]], tri_n: &[[u32; 3]]) -> i32 {
        assert_eq!(tri_v.len(), tri_n.len());
        let m = rt_mesh {
            first_vertex: (self.vertices.len() / 3) as u32,
            n_vertices: vertices.len() as u32,
            first_normal: (self.normals.len() / 3) as u32,
            n_normals: normals.len() as u32,
            first_triangle: (self.tri_v.len() / 3) as u32,
            n_triangles: tri_v.len() as u32,
        };
        for v in vertices {
            self.vertices.extend_from_slice(v);
        }
        for n in normals {
            self.normals.extend_from_slice(n);
        }
        for t in tri_v {
            self.tri_v.extend_from_slice(t);
        }
        for t in tri_n {
            self.tri_n.extend_from_slice(t);
        }
        self.meshes.push(m);
        (self.meshes.len() - 1) as i32
    }

    /// The borrowed view the C ABI takes.  Valid while `self` is neither moved nor mutated.
    pub fn desc(&self) -> rt_scene_desc {
        fn p<T>(v: &Vec<T>) -> *const T {
            if v.is_empty() {
                ptr::null()
            } else {
                v.as_ptr()
            }
        }
        rt_scene_desc {
            camera: rt_camera { fov_factor: self.camera_fov_factor, camera_to_world: self.camera_to_world },
            background: self.background,
            n_objects: self.objects.len() as u32,
            objects: p(&self.objects),
            n_nodes: self.nodes.len() as u32,
            nodes: p(&self.nodes),
            n_child_list: self.child_list.len() as u32,
            child_list: p(&self.child_list),
            n_materials: self.materials.len() as u32,
            materials: p(&self.materials),
            n_textures: self.textures.len() as u32,
            textures: p(&self.textures),
            n_images: self.images.len() as u32,
            images: p(&self.images),
            n_image_bytes: self.image_data.len() as u64,
            image_data: p(&self.image_data),
            n_meshes: self.meshes.len() as u32,
            meshes: p(&self.meshes),
            n_vertices: (self.vertices.len() / 3) as u32,
            vertices: p(&self.vertices),
            n_normals: (self.normals.len() / 3) as u32,
            normals: p(&self.normals),
            n_triangles: (self.tri_v.len() / 3) as u32,
            tri_v: p(&self.tri_v),
            tri_n: p(&self.tri_n),
        }
    }
}

/// The one method the host's `Shape` implementors gain (`src/shapes/mod.rs:57-60` stays as it is otherwise):
/// append yourself to `out` and return your node index.  Per implementor:
///
/// | type | kind | params | children |
/// |---|---|---|---|
/// | `Sphere` | `RT_SHAPE_SPHERE` | origin xyz, radius | |
/// | `Plane` | `RT_SHAPE_PLANE` | origin xyz, normal xyz | |
/// | `XYRectangle` / `XZRectangle` / `ZYRectangle` | `RT_SHAPE_XYRECT` .. | origin xyz, width, height; `flags` bit 0 = reverse | |
/// | `Cube` | `RT_SHAPE_CUBE` | p1 xyz, p2 xyz | |
/// | `Cylinder` | `RT_SHAPE_CYLINDER` | radius, height | |
/// | `Torus` | `RT_SHAPE_TORUS` | radius1, radius2 | |
/// | `CSGUnion` / `CSGIntersection` / `CSGDifference` | `RT_SHAPE_UNION` .. | | `child_a = a.flatten(out)`, `child_b = b.flatten(out)` |
/// | `Composite` | `RT_SHAPE_COMPOSITE` | | children's indices appended to `out.child_list`; `child_a` = first entry, `child_b` = count |
/// | `Mesh` | `RT_SHAPE_MESH` | | `child_a = out.push_mesh(..)`; `flags` bit 0 = `smooth_shading` |
/// | `HomogenousMedium` | `RT_SHAPE_MEDIUM` | density | `child_a = boundary.flatten(out)` |
///
/// `o2w` / `w2o` are the implementor's own `self.tx` matrices, untouched: the library composes them.
pub trait Flatten {
    fn flatten(&self, out: &mut FlatScene) -> i32;
}

/// The same for `Material` implementors: push an `rt_material` (and its `rt_texture`), return its index.
pub trait FlattenMaterial {
    fn flatten_material(&self, out: &mut FlatScene) -> i32;
}

// ------------------------------------------------------------------------------------------------ safe wrappers
/// `Options` (`src/system.rs:18-26`) plus the seed of the counter-based random numbers.
#[derive(Clone, Copy, Debug)]
pub struct Options {
    pub width: u32,
    pub height: u32,
    pub samples: u32,
    pub max_depth: u32,
    pub bias: f64,
    pub seed: u64,
}

impl Options {
    /// `main.rs:81-88`: bias 1e-4, max_depth 50.
    pub fn new(width: u32, height: u32, samples: u32) -> Options {
        Options { width, height, samples, max_depth: 50, bias: 1e-4, seed: 1 }
    }

    fn to_c(&self, shard_rank: u32, shard_count: u32, flags: u32) -> rt_options {
        rt_options {
            width: self.width,
            height: self.height,
            samples: self.samples,
            max_depth: self.max_depth,
            bias: self.bias,
            seed: self.seed,
            tile_size: 64,
            shard_rank,
            shard_count,
            flags,
        }
    }
}

/// `RenderProgress` (`src/system.rs:223-227`) as the host already implements it, over row-major sums.
pub trait RenderProgress {
    fn render_started(&mut self, options: &Options);
    /// `renderbuf`: `height * width * 3` running sums, row-major.
    fn sample_finished(&mut self, options: &Options, renderbuf: &[f64]);
    fn render_finished(&mut self, options: &Options, renderbuf: &[f64]);
}

/// A scene resident on the current HIP device.
pub struct Scene {
    handle: *mut rt_scene,
}

// One rt_scene may be rendered from one thread at a time (rt_api.h); moving it between threads is fine.
unsafe impl Send for Scene {}

impl Scene {
    pub fn new(flat: &FlatScene) -> Result<Scene, Error> {
        let desc = flat.desc();
        let mut handle: *mut rt_scene = ptr::null_mut();
        check(unsafe { rt_scene_create(&desc, &mut handle) })?;
        Ok(Scene { handle })
    }

    /// `system::render(options, scene, &mut progress)` on one GPU: returns the row-major sums.
    pub fn render(&mut self, options: &Options, progress: &mut dyn RenderProgress) -> Result<(Vec<f64>, rt_stats), Error> {
        struct Ctx<'a> {
            options: Options,
            progress: &'a mut dyn RenderProgress,
            n: usize,
        }
        unsafe extern "C" fn on_stratum(user: *mut c_void, _done: u32, _total: u32, rgb_sum: *const f64) -> c_int {
            let ctx = &mut *(user as *mut Ctx);
            let buf = std::slice::from_raw_parts(rgb_sum, ctx.n);
            ctx.progress.sample_finished(&ctx.options, buf);
            0
        }
        let n = (options.width as usize) * (options.height as usize) * 3;
        let mut buf = vec![0f64; n];
        let mut stats = rt_stats::default();
        let opts = options.to_c(0, 1, 0);
        progress.render_started(options);
        let mut ctx = Ctx { options: *options, progress, n };
        check(unsafe {
            rt_render(
                self.handle,
                &opts,
                buf.as_mut_ptr(),
                ptr::null_mut(),
                &mut stats,
                Some(on_stratum),
                &mut ctx as *mut Ctx as *mut c_void,
            )
        })?;
        ctx.progress.render_finished(options, &buf);
        Ok((buf, stats))
    }

    /// `system::render` + `write_render_result_to_file`'s pixel math (`src/main.rs:106`, `201-222`): the frame as
    /// row-major RGB8, tone-mapped on the GPU by strata + 1; 3 bytes per pixel cross PCIe instead of 24.
    pub fn render_rgb8(&mut self, options: &Options) -> Result<(Vec<u8>, rt_stats), Error> {
        let n = (options.width as usize) * (options.height as usize) * 3;
        let mut buf = vec![0u8; n];
        let mut stats = rt_stats::default();
        let opts = options.to_c(0, 1, 0);
        check(unsafe { rt_render_rgb8(self.handle, &opts, buf.as_mut_ptr(), ptr::null_mut(), &mut stats, ptr::null()) })?;
        Ok((buf, stats))
    }
}

/// Page-locked host memory (`rt_pinned_alloc`) for frames that leave the GPU at the link's speed.
pub struct PinnedBuf {
    ptr: *mut c_void,
    bytes: usize,
}

impl PinnedBuf {
    pub fn new(bytes: usize) -> Result<PinnedBuf, Error> {
        let mut ptr: *mut c_void = ptr::null_mut();
        check(unsafe { rt_pinned_alloc(bytes, &mut ptr) })?;
        Ok(PinnedBuf { ptr, bytes })
    }
    pub fn as_mut_ptr<T>(&mut self) -> *mut T {
        self.ptr as *mut T
    }
    pub fn len(&self) -> usize {
        self.bytes
    }
}

impl Drop for PinnedBuf {
    fn drop(&mut self) {
        unsafe { rt_pinned_free(self.ptr) }
    }
}

impl Drop for Scene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.handle) }
    }
}

/// One frame over several GPUs of this process (`rt_render_multi`): `scenes[k]` was created with GPU k current.
pub fn render_multi(scenes: &mut [Scene], options: &Options) -> Result<(Vec<f64>, rt_stats), Error> {
    let handles: Vec<*mut rt_scene> = scenes.iter().map(|s| s.handle).collect();
    let n = (options.width as usize) * (options.height as usize) * 3;
    let mut buf = vec![0f64; n];
    let mut stats = rt_stats::default();
    let opts = options.to_c(0, 1, 0);
    check(unsafe {
        rt_render_multi(handles.as_ptr(), handles.len() as u32, &opts, buf.as_mut_ptr(), ptr::null_mut(), &mut stats, ptr::null())
    })?;
    Ok((buf, stats))
}

/// `write_render_result_to_file` (`src/main.rs:219-227`): divisor = strata + 1 is the reference's own count.
pub fn write_png(path: &str, rgb_sum: &[f64], width: u32, height: u32, divisor: f64) -> Result<(), Error> {
    let n_pixels = (width as u64) * (height as u64);
    let mut rgb8 = vec![0u8; (n_pixels * 3) as usize];
    check(unsafe { rt_tonemap(rgb_sum.as_ptr(), n_pixels, divisor, rgb8.as_mut_ptr()) })?;
    let c_path = CString::new(path).map_err(|_| Error { status: RT_ERR_INVALID, message: "path contains NUL".into() })?;
    check(unsafe { rt_write_png(c_path.as_ptr(), rgb8.as_ptr(), width, height) })
}
