/*
 * rt_api.h — C ABI of the MI355X rendering core (librt_amd.so).
 *
 * The reference (doxxx/raytracer, Rust) has no FFI or plugin interface; the
 * seam this library replaces is the call
 *
 *     system::render(options, scene, &mut progress)
 *         reference src/main.rs:106  ->  src/system.rs:263-307
 *
 * A Rust host keeps its parser and scene API and adds one `flatten` pass that
 * fills `rt_scene_desc` (INTEGRATION.md shows the `extern "C"` block and the
 * build.rs a maintainer would add).  Everything crossing the boundary is plain
 * pointers, sizes and PODs.  All functions return 0 on success or a negative
 * `rt_status`; nothing unwinds or aborts across the boundary.  Pointers are
 * borrowed for the duration of the call.  One `rt_scene` may be rendered from
 * one thread at a time.
 *
 * Conventions follow the reference: f64 everywhere; points/directions are row
 * vectors, `p * M` with translation in row 3 (src/matrix.rs:29-36,198-225);
 * matrices are row-major double[16].
 */
#ifndef RT_API_H
#define RT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_API_VERSION 1

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / malformed scene description */
    RT_ERR_PARSE = -2,       /* scene text does not match the grammar       */
    RT_ERR_IO = -3,          /* file referenced by the scene not readable   */
    RT_ERR_DEVICE = -4,      /* HIP runtime error (no GPU, OOM, launch)      */
    RT_ERR_UNSUPPORTED = -5, /* scene exceeds a compiled-in device limit     */
} rt_status;

/* ---- scene description (what a host `flatten` pass produces) ------------ */

/* Files under src/shapes: one node per `Shape` value, file order, children first or
 * last does not matter (indices). */
typedef enum rt_shape_kind {
    RT_SHAPE_SPHERE = 0,        /* params: origin xyz, radius              (sphere.rs:12-25)   */
    RT_SHAPE_PLANE = 1,         /* params: origin xyz, normal xyz          (plane.rs:34-56)    */
    RT_SHAPE_XYRECT = 2,        /* params: origin xyz, width, height; flags bit0 = reverse (plane.rs:111-132) */
    RT_SHAPE_XZRECT = 3,        /* same                                    (plane.rs:156-177)  */
    RT_SHAPE_ZYRECT = 4,        /* same                                    (plane.rs:201-222)  */
    RT_SHAPE_CUBE = 5,          /* params: p1 xyz, p2 xyz                  (cube.rs:20-37)     */
    RT_SHAPE_CYLINDER = 6,      /* params: radius, height                  (cylinder.rs:12-25) */
    RT_SHAPE_TORUS = 7,         /* params: radius1 (ring), radius2 (tube)  (torus.rs:13-26)    */
    RT_SHAPE_UNION = 8,         /* child_a, child_b                        (csg.rs:7-21)       */
    RT_SHAPE_INTERSECTION = 9,  /* child_a, child_b                        (csg.rs:112-126)    */
    RT_SHAPE_DIFFERENCE = 10,   /* child_a, child_b                        (csg.rs:190-204)    */
    RT_SHAPE_COMPOSITE = 11,    /* child_a = first entry in child_list, child_b = count (composite.rs:6-18) */
    RT_SHAPE_MESH = 12,         /* child_a = mesh index; flags bit0 = smooth_shading (mesh.rs:11-52) */
    RT_SHAPE_MEDIUM = 13,       /* child_a = boundary node; params[0] = density (homogenous_medium.rs:13-27) */
} rt_shape_kind;

typedef struct rt_shape_node {
    int32_t kind;
    int32_t child_a;
    int32_t child_b;
    int32_t flags;
    double params[8];
    double o2w[16]; /* Transformation::object_to_world (object.rs:7-26) */
    double w2o[16]; /* Transformation::world_to_object = o2w.inverse()  */
} rt_shape_node;

typedef enum rt_material_kind {
    RT_MAT_LAMBERTIAN = 0,    /* texture                (materials/lambertian.rs)    */
    RT_MAT_METAL = 1,         /* fuzz, texture          (materials/metal.rs)         */
    RT_MAT_DIELECTRIC = 2,    /* ior, fuzz              (materials/dielectric.rs)    */
    RT_MAT_DIFFUSE_LIGHT = 3, /* intensity, texture     (materials/diffuse_light.rs) */
    RT_MAT_ISOTROPIC = 4,     /* texture                (materials/isotropic.rs)     */
} rt_material_kind;

typedef struct rt_material {
    int32_t kind;
    int32_t texture; /* index into textures, -1 for dielectric */
    double fuzz;
    double ior;
    double intensity;
} rt_material;

typedef enum rt_texture_kind {
    RT_TEX_SOLID = 0,        /* c1                       (texture.rs:15)     */
    RT_TEX_CHECKERBOARD = 1, /* c1, c2, scale            (texture.rs:72-90)  */
    RT_TEX_IMAGE = 2,        /* image, scale             (texture.rs:59-67)  */
} rt_texture_kind;

typedef struct rt_texture {
    int32_t kind;
    int32_t image; /* index into images */
    double c1[3];
    double c2[3];
    double scale;
} rt_texture;

typedef struct rt_image {
    uint32_t width;
    uint32_t height;
    uint64_t offset; /* byte offset of the top-left RGB8 texel in image_data */
} rt_image;

/* One per wavefront `o` group (sdl.rs:65-92).  Indices in tri_v / tri_n are
 * relative to first_vertex / first_normal. */
typedef struct rt_mesh {
    uint32_t first_vertex, n_vertices;
    uint32_t first_normal, n_normals;
    uint32_t first_triangle, n_triangles;
} rt_mesh;

typedef struct rt_object {
    int32_t shape;    /* root node */
    int32_t material; /* index into materials */
} rt_object;

/* Camera::new output (system.rs:37-55). */
typedef struct rt_camera {
    double fov_factor;          /* tan(fov/2) */
    double camera_to_world[16]; /* rows: xaxis, yaxis, zaxis, origin */
} rt_camera;

typedef struct rt_scene_desc {
    rt_camera camera;
    double background[3]; /* SceneOptions::background_color (sdl.rs:24-34) */

    uint32_t n_objects;
    const rt_object *objects; /* file order: index = primary-hit id */
    uint32_t n_nodes;
    const rt_shape_node *nodes;
    uint32_t n_child_list;
    const int32_t *child_list;
    uint32_t n_materials;
    const rt_material *materials;
    uint32_t n_textures;
    const rt_texture *textures;
    uint32_t n_images;
    const rt_image *images;
    uint64_t n_image_bytes;
    const uint8_t *image_data;
    uint32_t n_meshes;
    const rt_mesh *meshes;
    uint32_t n_vertices;
    const double *vertices; /* xyz */
    uint32_t n_normals;
    const double *normals; /* xyz */
    uint32_t n_triangles;
    const uint32_t *tri_v; /* 3 per triangle */
    const uint32_t *tri_n; /* 3 per triangle */
} rt_scene_desc;

/* ---- render options ------------------------------------------------------ */

/* `Options` (system.rs:18-26) plus what the reference leaves implicit. */
typedef struct rt_options {
    uint32_t width;
    uint32_t height;
    uint32_t samples;   /* `-s`; floor(sqrt(samples))^2 strata are rendered (system.rs:276-286) */
    uint32_t max_depth; /* 50 in main.rs:86 */
    double bias;        /* 1e-4 in main.rs:85 */
    uint64_t seed;      /* rt_rng.h */
    /* Tile sharding (multi-GPU): the image is cut into tile_size x tile_size
     * tiles in row-major tile order; tile k belongs to shard k % shard_count. */
    uint32_t tile_size;   /* multiple of 8; 0 -> 64 */
    uint32_t shard_rank;  /* 0 .. shard_count-1 */
    uint32_t shard_count; /* 0 -> 1 */
    uint32_t flags;       /* RT_FLAG_* */
} rt_options;

/* rt_options.flags */
#define RT_FLAG_ROW_MAJOR 1u /* shard_count == 1 only: outputs are indexed y * width + x instead of by shard
                             * slot (width * height elements), so a whole image needs no rt_unshard_device */

#define RT_FLAG_STAGE_TIMES 2u /* fill rt_stats.ms_raygen .. ms_tail: four HIP events per bounce pass on the launch
                               * stream, which cost about 3 % of a 4K frame; without the flag those fields are 0 */

typedef struct rt_stats {
    uint64_t rays;         /* Ray::trace invocations, primary + secondary  */
    uint64_t primary_rays; /* of which depth 0                             */
    uint32_t strata;       /* strata rendered = floor(sqrt(samples))^2     */
    uint32_t bounce_passes; /* path depths that had rays                    */
    /* Situations in which the reference panics; the device
     * treats them as a miss and counts them: cube >2 face hits, torus
     * three-root mismatch, CSG "unanticipated difference case", cylinder
     * asserts, NaN ordering. */
    uint32_t guards[5];
    uint32_t interval_overflow; /* CSG interval list exceeded device capacity */
    double ms_device;           /* GPU time of the call, HIP events on its stream */
    /* per-kernel GPU time, HIP events on the same stream, summed over launches */
    double ms_raygen;
    double ms_intersect; /* k_intersect: nearest hit over every object except the triangles, and the
                          * shading of the rays that have no mesh to visit                            */
    double ms_shade;     /* k_shade: the shading of the rays that went through the mesh record queue  */
    double ms_mesh;      /* k_mesh / k_mesh_brute: the rays queued for a mesh   */
    double ms_tail;      /* k_tail: the last few paths, all remaining bounces   */
    /* launches and rays per stage: always filled (the times above only with RT_FLAG_STAGE_TIMES) */
    uint32_t n_intersect_launches; /* k_intersect launches that had rays        */
    uint32_t n_mesh_launches;      /* k_mesh (grid walk) launches               */
    uint64_t mesh_rays;            /* rays those k_mesh launches walked         */
    uint64_t intersect_rays;       /* rays those k_intersect launches took      */
    uint64_t mesh_records;         /* rays that went through the mesh record queue (k_mesh or k_mesh_brute, then k_shade) */
    uint64_t mesh_one_rays;        /* of mesh_rays: walked by k_mesh_one (scenes with one mesh job), which leaves the merge
                                    * of its triangle into the record's hit to k_shade */
} rt_stats;

/* Nearest-hit record, one per ray: RayHit (system.rs:154-171). */
typedef struct rt_hit {
    double t; /* < 0 -> miss */
    double n[3];
    double uv[2];
    int32_t object; /* -1 -> miss */
    int32_t pad;
} rt_hit;

typedef struct rt_scene rt_scene;

/* RenderProgress::sample_finished (system.rs:223-227): called on the calling
 * thread after each stratum with the running-sum buffer (host, row-major
 * [H][W][3]).  Return non-zero to stop early. */
typedef int (*rt_progress_cb)(void *user, uint32_t strata_done, uint32_t strata_total, const double *rgb_sum);

/* ---- device path --------------------------------------------------------- */

/* Replaces the construction of RenderContext (system.rs:272-278): deep-copies
 * the description onto the current HIP device, builds the per-mesh grid and
 * the flattened per-object CSG programs. */
int rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

/* Replaces system::render (system.rs:263-307) for a whole image on one GPU.
 * rgb_sum: H*W*3 doubles, row-major, SUM over strata (not divided).
 * primary_ids: H*W or NULL; object index of the primary hit in stratum 0,
 * -1 for a miss.  stats, cb may be NULL; without cb nothing leaves the GPU
 * before the frame is done.  The scene keeps the device-side frame between
 * calls (grown on demand). */
int rt_render(rt_scene *scene, const rt_options *opts, double *rgb_sum, int32_t *primary_ids, rt_stats *stats,
              rt_progress_cb cb, void *user);

/* Same, for one shard of the tiles, on caller-provided DEVICE buffers and a
 * caller-provided stream (hipStream_t, NULL = default).  Outputs are in shard
 * layout: rt_shard_pixels(opts) entries, tile after tile (see
 * rt_unshard_device).  All work is queued on `stream`; the call returns when it has
 * finished (the host follows the queue lengths from pass to pass and fills `stats`),
 * so the outputs are complete on return. */
int rt_render_device(rt_scene *scene, const rt_options *opts, double *d_rgb_sum, int32_t *d_primary_ids,
                     rt_stats *stats, void *stream);

/* RenderProgress (system.rs:223-227) for renders whose frame stays on the device.  The reference's
 * CliRenderProgress counts strata in sample_finished and rewrites out.png from the running sum at most
 * every 5 s, dividing by strata_done + 1 (main.rs:174-185,219-222).  Handing 24 bytes per pixel to the host
 * after every stratum is what that costs; here the cheap callback carries no image, and the snapshot
 * callback is only called when due, with the running sum already tone-mapped on the device (3 bytes per
 * pixel cross PCIe).  Either callback may be NULL; a non-zero return stops the render after that stratum. */
typedef int (*rt_stratum_cb)(void *user, uint32_t strata_done, uint32_t strata_total);
/* rgb8: n_pixels * 3 bytes of host memory owned by the library, valid during the call.  Layout: that of the
 * render's outputs - shard slots for rt_render_device_progress (rt_shard_map gives the pixel of each slot;
 * row-major with RT_FLAG_ROW_MAJOR), the row-major image for rt_render_multi. */
typedef int (*rt_snapshot_cb)(void *user, uint32_t strata_done, uint32_t strata_total, const uint8_t *rgb8, uint64_t n_pixels);
typedef struct rt_progress {
    rt_stratum_cb on_stratum;   /* after every stratum (sample_finished's count)                       */
    rt_snapshot_cb on_snapshot; /* after a stratum that is not the last, when min_interval_ms have passed
                                 * since the render began or since the last snapshot                     */
    uint32_t min_interval_ms;   /* 5000 in main.rs:178; 0 = after every stratum                         */
    uint32_t reserved;
    void *user;
} rt_progress;

/* rt_render_device with progress reporting (progress may be NULL). */
int rt_render_device_progress(rt_scene *scene, const rt_options *opts, double *d_rgb_sum, int32_t *d_primary_ids,
                              rt_stats *stats, void *stream, const rt_progress *progress);

/* system::render for a whole image over n_devices GPUs of this process.  The reference's one parallel step
 * spreads strata over threads and merges them under a mutex (system.rs:288-299); here the tiles of the image
 * are spread over the GPUs (tile k to scenes[k % n_devices], opts->shard_* are ignored), each GPU runs its
 * wavefront loop from its own host thread, and the finished tiles are gathered over xGMI to scenes[0]'s GPU,
 * which scatters them into the row-major image (combine_renderbuf's counterpart) and hands it to the host.
 * Every peer pushes its shard on its own stream into its own region of the first GPU's buffer, so the copies
 * run over the peers' links at the same time; one launch then scatters all regions.
 * scenes[k]: the same description created with GPU k current (rt_scene_create); the same device may appear
 * more than once, each time through an rt_scene of its own (one scene serves one render at a time: the same
 * pointer twice is RT_ERR_INVALID).  rgb_sum: H*W*3 doubles, host (page-locked memory, rt_pinned_alloc, takes
 * the frame at the link's speed); primary_ids: H*W or NULL; stats: rays and counters summed over the devices,
 * ms_device of the slowest; progress (may be NULL): on_stratum when every GPU has finished that stratum,
 * on_snapshot with the row-major image assembled from all shards.  A device that fails (its render, a progress
 * snapshot, its copy) makes the call return that device's code with its message in rt_last_error().  Each
 * scene keeps a stream and a host thread for these calls.  The calling thread drives scenes[0]'s GPU. */
int rt_render_multi(rt_scene *const *scenes, uint32_t n_devices, const rt_options *opts, double *rgb_sum,
                    int32_t *primary_ids, rt_stats *stats, const rt_progress *progress);

/* system::render followed by write_render_result_to_file's pixel math (main.rs:106, 201-222), for a host that
 * wants the image and not the sums: the frame is tone-mapped on the GPU by strata + 1 (the reference's divisor,
 * SURVEY Q1) and handed over as H*W*3 bytes, row-major RGB8 - 3 bytes per pixel across PCIe instead of 24.
 * primary_ids, stats, progress may be NULL; progress snapshots are row-major RGB8 as well. */
int rt_render_rgb8(rt_scene *scene, const rt_options *opts, uint8_t *rgb8, int32_t *primary_ids, rt_stats *stats,
                   const rt_progress *progress);
/* The same over n_devices GPUs (rt_render_multi's arguments): every GPU tone-maps its own tiles before it pushes them,
 * so 3 bytes per pixel cross the links as well. */
int rt_render_multi_rgb8(rt_scene *const *scenes, uint32_t n_devices, const rt_options *opts, uint8_t *rgb8,
                         int32_t *primary_ids, rt_stats *stats, const rt_progress *progress);

/* Page-locked host memory for the outputs of rt_render / rt_render_rgb8 / rt_render_multi (hipHostMalloc): the
 * copy out of the GPU then runs at the link's speed instead of through a staging buffer.  No counterpart in the
 * reference (its frame buffer never leaves the process).  Free with rt_pinned_free. */
int rt_pinned_alloc(size_t bytes, void **out);
void rt_pinned_free(void *p);

/* rt_render_device for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc,
 * processes whose device memory is spoken for.  No counterpart in the reference.  Same arguments, same
 * outputs (bit for bit), but the wavefront state (ray queues, hit records, mesh queue, counters) lives in a
 * static device array of the library (99.8 MiB per GPU the library runs on, held until the process ends)
 * instead of a workspace the scene allocates and grows, and the call makes no hipMalloc / hipFree /
 * hipHostMalloc / hipEventCreate: everything else it needs was set up by rt_scene_create.  Like
 * rt_render_device it drives the bounce passes from the host: it reads queue lengths back and synchronises
 * `stream` before it returns, so it cannot run under HIP stream capture.  Limits: a shard of at most
 * RT_NOALLOC_MAX_SLOTS ray slots (rt_shard_pixels(opts); 640 x 480 fits) and max_depth <=
 * RT_NOALLOC_MAX_DEPTH, beyond which it returns RT_ERR_UNSUPPORTED.  Every call on a GPU works in that GPU's
 * one array: a call made while another is running on the same GPU (two streams, two scenes, two threads)
 * returns RT_ERR_INVALID; calls from other processes are not seen and race. */
#define RT_NOALLOC_MAX_SLOTS (84u * 4096u)
#define RT_NOALLOC_MAX_DEPTH 64u
int rt_render_device_noalloc(rt_scene *scene, const rt_options *opts, double *d_rgb_sum, int32_t *d_primary_ids,
                             rt_stats *stats, void *stream);

/* Number of pixel slots (padded to whole tiles) that shard `opts->shard_rank`
 * owns. */
uint64_t rt_shard_pixels(const rt_options *opts);

/* Host-side view of the same layout: pixel_index[slot] = y * width + x, or
 * 0xFFFFFFFF for padding beyond the image edge; rt_shard_pixels(opts) entries. */
int rt_shard_map(const rt_options *opts, uint32_t *pixel_index);

/* combine_renderbuf's counterpart after an RCCL gather: scatter one shard's
 * buffers (device) into the full row-major image buffers (device). Either
 * pair may be NULL. */
int rt_unshard_device(const rt_options *opts, uint32_t shard_rank, const double *d_shard_rgb, double *d_full_rgb,
                      const int32_t *d_shard_ids, int32_t *d_full_ids, void *stream);

/* The same for a shard that was tone-mapped where it was rendered (rt_tonemap_device on the shard's buffer: the pixel
 * math of main.rs:201-222 is per pixel, so it commutes with the gather): 3 bytes per slot cross the links instead of
 * 24.  d_shard_rgb8: rt_shard_pixels(opts) x 3 bytes; d_full_rgb8: width x height x 3 bytes, row-major. */
int rt_unshard_rgb8_device(const rt_options *opts, uint32_t shard_rank, const uint8_t *d_shard_rgb8, uint8_t *d_full_rgb8,
                           void *stream);

/* Ray::trace (system.rs:122-133) for a batch of world-space rays (host
 * arrays, xyz interleaved).  Directions must have unit length, as every ray of
 * the reference has (camera rays, scattered rays and Ray::transform all
 * normalise): a direction off by more than 1e-9 is RT_ERR_INVALID.  Ray i draws
 * its fog random numbers as pixel i, stratum 0, depth 0. */
int rt_trace_rays(rt_scene *scene, uint64_t seed, uint64_t n, const double *origins, const double *directions,
                  rt_hit *hits, rt_stats *stats);

/* write_render_result_to_file's pixel math (main.rs:201-222): c = sum /
 * divisor, sqrt, * 255, min 255, truncate.  The reference passes divisor =
 * strata + 1.  Host buffers. */
int rt_tonemap(const double *rgb_sum, uint64_t n_pixels, double divisor, uint8_t *rgb8);

/* The same pixel math as a kernel on `stream` (hipStream_t or NULL), device
 * buffers in and out: a frame that is only wanted as an image leaves the GPU
 * as 3 bytes per pixel instead of 24. */
int rt_tonemap_device(const double *d_rgb_sum, uint64_t n_pixels, double divisor, uint8_t *d_rgb8, void *stream);

/* The file half of write_render_result_to_file (main.rs:219-227): an 8-bit
 * RGB PNG of a tone-mapped frame, rows top to bottom as in the render buffer.
 * (The pixels are the reference's; the compressed bytes depend on the encoder
 * and are not comparable with image 0.14's.) */
int rt_write_png(const char *path, const uint8_t *rgb8, uint32_t width, uint32_t height);

/* Thread-local message for the last non-zero status. */
const char *rt_last_error(void);

/* ---- host plumbing: scene text -> rt_scene_desc -------------------------- */

/* sdl::parse (sdl.rs:36-38, sdl_grammar.rs).  `base_dir` resolves mesh and
 * image paths (the reference resolves them against the CWD).  Width/height
 * feed Camera::new.  The result owns its arrays; free with
 * rt_scene_desc_free. */
int rt_sdl_parse(const char *text, const char *base_dir, uint32_t width, uint32_t height, rt_scene_desc **out);
int rt_sdl_parse_file(const char *path, const char *base_dir, uint32_t width, uint32_t height, rt_scene_desc **out);
void rt_scene_desc_free(rt_scene_desc *desc);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */
