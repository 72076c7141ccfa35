/* =============================================================================
 * rrt.h -- C ABI of the MI355X (gfx950) tracing core for rustraytracer scenes.
 *
 * The reference (rishabh-bector/rustraytracer) has no FFI layer; its operator
 * API is the Rust traits Entity / RayBehavior / LightSource plus the
 * constructors that main.rs calls.  The cut point is
 *     RayTracer::render(self, output, world)        src/tracer.rs:77
 * Everything above it (scene set-up, file I/O) stays on the host; everything
 * below it (ray generation, closest hit, shading, sky lookup, quantisation)
 * is one blocking call into this library.  Because `dyn` traits are open sets
 * and a GPU needs closed ones, the ABI is a scene-description API over the
 * closed set the reference ships: entity kinds {sphere, triangle mesh + rope
 * kd-tree, bare triangle, Scene<T> group}, behaviour kinds {lambert, reflection,
 * phong, cubemap}, light kinds {directional, point}.  Each entry point cites the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes, no C++/torch types.  All functions
 * returning int return RRT_OK (0) or a negative RRT_E_* code; the message is
 * available from rrt_last_error() (thread-local).  Nothing throws or aborts
 * across the boundary (the reference panics: cubemap.rs:19-44, model.rs:26-27).
 * The library copies every input at the call (the caller may free at once);
 * the caller owns all output buffers.  One scene handle is not re-entrant;
 * distinct handles may be used from distinct threads.
 * ========================================================================== */
#ifndef RRT_H
#define RRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_OK 0
#define RRT_E_INVALID (-1)   /* bad argument / handle / id                      */
#define RRT_E_HIP (-2)       /* a HIP runtime call failed                       */
#define RRT_E_STATE (-3)     /* scene incomplete (no sky set, no device, ...)   */
#define RRT_E_NOMEM (-4)     /* work buffers do not fit the configured budget   */

typedef struct rrt_scene rrt_scene;

/* Camera of RayTracer::new_default_renderer + the lens set-up at the top of
 * render (src/tracer.rs:38-51, 84-110): eye, lens plane at eye.z + lens_z,
 * lens of lens_w x lens_h world units centred on the eye's x/y.  The
 * reference's values are {0,0,0}, 75, 160, 90 (rrt_default_camera). */
typedef struct rrt_camera {
    double eye[3];
    double lens_z;
    double lens_w;
    double lens_h;
} rrt_camera;

/* RenderSettings (src/tracer.rs:16-18) plus the build's extensions.
 * Reference behaviour = {W, H, spp 1, max_bounce 2, seed any, prune 0}.
 *   spp          samples per pixel.  Sample 0 is always the reference's ray
 *                through the pixel's top-left corner (tracer.rs:128-131);
 *                samples s>=1 add a jitter in [0,1)^2 pixel units drawn from
 *                Philox4x32-10(counter = {pixel, s, 0, 0}, key = seed).
 *                The pixel is quantised from (sum over s in order) / spp.
 *   max_bounce   ReflectionBehavior's literal `2` (reflection.rs:26): a
 *                reflection ray is spawned while ray.bounce <= max_bounce.
 *   prune_zero_mix  1 = do not trace reflection rays whose mix is exactly 0.0
 *                (value-identical; changes the ray count).
 *   tile_rank/tile_world  image-tile sharding: 32x8-pixel tiles; tile t belongs
 *                to rank t % tile_world and sits in row ty = t / tiles_x of the
 *                tile grid, column (t % tiles_x + rrt_tile_skew() * ty) % tiles_x
 *                (each row of tiles rotated, so that neighbouring tiles of a row
 *                AND of a column go to different ranks).  tile_world = 1 renders
 *                the whole frame.
 *   batch_spp    samples per pixel per wavefront batch; 0 = choose from the
 *                work-memory budget (rrt_set_work_budget).
 *   tile_deal    (tile_world > 1) 1 = deal the tiles to the ranks BY COST instead of by position: every rank sorts
 *                the image's tiles into [a mesh can be met, only spheres can, only sky] from the frame's tile bins
 *                (each class in image order; the same list on every rank), and position p of that list is slot
 *                p / tile_world of rank p % tile_world -- each rank gets its share of every class to within one
 *                tile, so the ranks' frames cost the same.  The compact tile buffer then carries its slot -> tile
 *                table behind the pixels (rrt_tile_buffer_bytes, rrt_gather_tiles_dealt, rrt_assemble_tiles_dealt).
 *                0 = the positional map above.  The image does not depend on it. */
typedef struct rrt_settings {
    uint32_t width;
    uint32_t height;
    uint32_t spp;
    uint32_t max_bounce;
    uint64_t seed;
    uint32_t prune_zero_mix;
    uint32_t tile_rank;
    uint32_t tile_world;
    uint32_t batch_spp;
    uint32_t tile_deal;
    uint32_t reserved;        /* 0 (keeps the struct free of hidden padding) */
} rrt_settings;

/* Filled by rrt_get_stats after a render. */
typedef struct rrt_stats {
    uint64_t rays_traced;      /* closest-hit queries (= cast() invocations, tracer.rs:177) executed */
    uint64_t primary_rays;     /* of which primary                                                    */
    uint64_t paths;            /* pixel samples                                                       */
    uint64_t traversal_overflow; /* rays that exceeded the leaf-visit safety bound (expected 0)       */
    double gpu_ms;             /* hipEvent span of the last render's kernels                          */
    double primary_kernel_ms;  /* summed hipEvent duration of the primary-ray kernel launches         */
    double bounce_kernel_ms;   /* hipEvent span of the bounce-queue kernel launches (one pair per batch around its levels) */
    uint32_t primary_kernel_launches;
    uint32_t bounce_kernel_launches;
    uint32_t n_batches;
    uint32_t reserved;
    uint64_t level_rays[4];    /* rays cast at bounce level 0,1,2,3 (level 0 = primary)               */
    uint64_t cull_violations;  /* cull mode 2 only: entries the cull skipped but the exact test accepts (must be 0) */
    uint64_t zero_mix_paths;   /* primary hits whose reflection ray was cast with mix() == 0.0           */
    double sky_kernel_ms;      /* summed hipEvent duration of the sky-only kernel launches (they run on a side stream, beside the bounce levels) */
    uint32_t heavy_tiles;      /* tiles of this rank whose primary rays went through the wavefront kernels ...         */
    uint32_t sky_tiles;        /* ... and tiles behind which there is only sky (0 / 0 for a frame without tile lists)  */
} rrt_stats;

#define RRT_TILE_W 32
#define RRT_TILE_H 8

const char* rrt_last_error(void);
const char* rrt_version(void);

/* Selects the HIP device used by scenes created afterwards on this thread
 * (one process per GPU: pass LOCAL_RANK). */
int rrt_set_device(int device);

/* 1 if the device reports HIP image/texture support, 0 if not, <0 on error. */
int rrt_device_image_support(void);

void rrt_default_camera(rrt_camera* cam);                       /* tracer.rs:38-51 */
void rrt_default_settings(rrt_settings* st, uint32_t w, uint32_t h);

/* World (src/common.rs:10-15) ------------------------------------------------ */
rrt_scene* rrt_scene_create(void);                              /* RayTracer::new_empty_world's empty World */
void rrt_scene_destroy(rrt_scene* s);

/* Material::new_lambert_material(color, albedo, lambert, reflective, phong,
 * alpha) (src/material.rs:20-33): shaders [Lambert, Reflection, Phong].
 * Returns the material id (>= 0) or a negative error. */
int rrt_add_material(rrt_scene* s, const double color[3], double albedo, double lambert_mix,
                     double reflective_mix, double phong_mix, int32_t alpha);

/* EXTENSION (no reference counterpart; BASELINE configs[2..4] ask for refraction and stochastic
 * diffuse bounces, which the reference lacks -- SURVEY §0).  Same three shader slots, but slot [1],
 * the continuation lobe, is chosen: lobe 0 = Reflection (identical to rrt_add_material),
 * 1 = Refraction with index `ior` (Snell; total internal reflection mirrors; a ray refracted into
 * a sphere travels "inside" it and meets that sphere's FAR root, because sphere.rs:28-47 only
 * returns the near one), 2 = DiffuseBounce: direction normalize(n + uniform sphere point) drawn
 * from Philox4x32-10(counter = {pixel, sample, bounce, 1+attempt}, key = seed) by Marsaglia's
 * trig-free rejection.  colour += cast(continuation) * lobe_mix, under the same
 * `bounce <= max_bounce` rule and summation order as the Reflection slot (tracer.rs:194-202).
 * One continuation lobe per material keeps every path a non-branching chain. */
int rrt_add_material_ext(rrt_scene* s, const double color[3], double albedo, double lambert_mix, double lobe_mix,
                         double phong_mix, int32_t alpha, int32_t lobe, double ior);

/* Sphere::new(position, radius, material) + world.entities.push
 * (src/geometry/sphere.rs:17-24, src/main.rs:62).  Returns the entity id =
 * insertion order, as the reference's entity list. */
int rrt_add_sphere(rrt_scene* s, const double center[3], double radius, int material);

/* The Triangle list Model::new produces (src/geometry/model.rs:31-45) + the
 * KDTree::new it builds (src/geometry/kdtree.rs:46-51) + world.entities.push.
 * verts: 9 doubles per triangle (v0,v1,v2 after the model transform);
 * normals: 3 doubles per triangle (the flat averaged normal).  The kd-tree is
 * built inside with the reference's algorithm (kdtree.rs:53-190) including
 * the tie order of its sort_unstable_by.  Returns the entity id. */
int rrt_add_mesh(rrt_scene* s, const double* verts, const double* normals, int64_t n_triangles, int material);

/* Triangle::new(v0, v1, v2, normal, material) + world.entities.push (src/geometry/triangle.rs:19-23 is `pub`, and
 * Triangle implements Entity, triangle.rs:25-83): a bare triangle as an entity of its own.  Entity::collide is
 * Triangle::collide (triangle.rs:25-58) and nothing else -- rrt_add_mesh with one triangle is NOT equivalent: that
 * goes through Model's kd-tree, whose root-box test (aabb.rs:50-100 with the +0.001 nudge of :89) rejects every ray
 * for an axis-aligned (flat-box) triangle.  `normal` is stored and returned as given (triangle.rs:54; not normalised,
 * not derived from the vertices).  May also be a member of a group (rrt_add_group).  The id AOV reports the entity id
 * and triangle id -1 ("not part of a mesh").  Returns the entity id. */
int rrt_add_triangle(rrt_scene* s, const double v0[3], const double v1[3], const double v2[3], const double normal[3],
                     int material);

/* Scene::new(models, position) + world.entities.push(scene) (src/geometry/scene.rs:39-47; the
 * scene!/entity_enum! macros scene.rs:12-29 and #[derive(Entity)] derive_entity/src/lib.rs:7-51 are
 * the enum plumbing around it; usage commented out at src/main.rs:57-59).  `members` are ids of
 * spheres / meshes / bare triangles already added: they leave the world's entity list and are reached only through
 * the group's kd-tree over their bounding boxes (KDTree<T>::new over entities, kdtree.rs:46-51),
 * whose collide (kdtree.rs:192-235) answers with the closest member of the first leaf in which
 * any member is hit -- after the tree's root box passed aabb.rs:50-100, so results can differ from
 * the flat entity loop exactly as they do in the reference.  A member may itself be a group (Scene<T> is generic over
 * any Entity, Scene<...> included, scene.rs:32,50-62: its bounding box is its own tree's root box, its collide its own
 * walk) -- one level deep: a member group may not hold groups (RRT_E_INVALID).  Entity ids
 * reported in the id AOV remain those of the innermost members.  The group's `position` argument is stored
 * by the reference but never read by render; it is not part of this call.
 * Returns the group's entity id. */
int rrt_add_group(rrt_scene* s, const int32_t* members, int32_t n_members);

/* Material::new_sky_material / CubemapBehavior::new (src/material.rs:35-42,
 * src/behavior/cubemap.rs:16-52) after the PNGs are decoded to RGB8 by the
 * host.  face index = position in the reference's maps[] array
 * [left,right,down,up,front,back] (cubemap.rs:47-48). rgb8: h rows of w texels. */
int rrt_set_sky_face(rrt_scene* s, int face, const uint8_t* rgb8, uint32_t w, uint32_t h);

/* world.light_sources is a LIST (src/common.rs:12: Vec<Box<dyn LightSource>>): Lambert and Phong sum over it in
 * push order (src/behavior/lambert.rs:34-43, phong.rs:37-45), and with f64 the order of a sum is part of the result.
 * rrt_add_directional_light = DirectionalLight::new(direction, color, intensity) (src/lighting.rs:25-32; the
 * direction is normalised inside, as there) + world.light_sources.push: appends, returns the light's index in the
 * list.  rrt_add_point_light (below) appends to the SAME list, so directional and point lights interleave in the
 * order the host pushes them.  At most 32 lights.
 * rrt_set_directional_light = "the list becomes [this light]": what RayTracer::new_empty_world leaves behind
 * (tracer.rs:55-65); lights pushed before it are dropped. */
int rrt_add_directional_light(rrt_scene* s, const double direction[3], const double color[3], double intensity);
int rrt_set_directional_light(rrt_scene* s, const double direction[3], const double color[3], double intensity);

int rrt_set_ambient(rrt_scene* s, double ambient);             /* World.ambient, tracer.rs:73 */

/* PointLight{position, color, brightness, attenuation} pushed to world.light_sources
 * (src/lighting.rs:52-89).  The reference gives the struct no constructor and main.rs never builds
 * one (dead code there), so this is PARITY UNPINNED against the reference and
 * restated from the impl block alone: illuminate = {power: brightness / (attenuation * |pos-light|^2),
 * direction: normalize(pos - light)} (lighting.rs:60-68), summed by Lambert and Phong at its place in
 * world.light_sources (push order across light kinds, lambert.rs:34-43, phong.rs:37-45).  At most 16.  Its visible()
 * casts a SHADOW RAY through the world -- an any-hit walk of the same entities / kd-trees (the
 * first accepted triangle that is close enough ends it) -- see rrt_set_shadow_mode.
 * Returns the light's index in world.light_sources. */
int rrt_add_point_light(rrt_scene* s, const double position[3], const double color[3], double brightness,
                        double attenuation);

/* How PointLight::visible is answered.
 *   0 (default) = src/lighting.rs:70-84 exactly as written: the ray starts ON the surface, its
 *       direction (light - pos) is not normalised, and the loop answers "visible" as soon as ONE
 *       entity of the world is missed, "not visible" only when every entity is hit (inverted with
 *       respect to a shadow test; an empty world is never lit).  Restated because it is the code.
 *   1 = EXTENSION, the corrected test: the surface must face the light (n . (pos - light) < 0, as
 *       DirectionalLight::visible lighting.rs:43-45), the ray starts at pos + normal*0.3 (the offset
 *       reflection.rs:30 uses) with a normalised direction, and the light is visible iff no entity
 *       answers Entity::collide closer than the light. */
int rrt_set_shadow_mode(rrt_scene* s, int mode);

/* Entity::translate(vec) (src/geometry/sphere.rs:68-70; model.rs:80-86 = Triangle::translate
 * triangle.rs:79-83 on every triangle + KDTree::translate_nodes kdtree.rs:35-42; scene.rs:68-74 =
 * every member + the group's own nodes).  Never called by the reference's render.
 *   refit = 0: as written -- vertices and split planes move, the nodes' boxes (and the model's
 *       aa_bb) stay where they were, because translate_nodes forgets them: rays are then walked
 *       through the stale boxes, bit for bit as the Rust code would.
 *   refit = 1: EXTENSION -- every box of the tree moves as well (AABB::translate aabb.rs:116-119 on
 *       each node), i.e. a rigid translation: the result equals a scene built at the new place
 *       whenever the additions are exact.
 * The scene is re-uploaded by the next render / rrt_scene_commit. */
int rrt_translate_entity(rrt_scene* s, int entity, const double delta[3], int refit);

/* Upper bound in bytes for per-render wavefront work buffers (queues, path
 * records, sample buffer).  Default 48 GiB (per stream the scene renders on). */
int rrt_set_work_budget(rrt_scene* s, uint64_t bytes);

/* Conservative triangle cull inside the kd leaves (no reference counterpart; results are
 * bit-identical by construction, see rrt_kernels.hip): 0 = off (every triangle of a visited leaf
 * gets the exact f64 test, as kdtree.rs:203-212), 1 = on (default), 2 = verify (run both, count
 * disagreements in rrt_stats.cull_violations). */
int rrt_set_cull_mode(rrt_scene* s, int mode);

/* hipEvents a render records for rrt_get_stats (no reference counterpart; the reference times the whole
 * render with one Instant, tracer.rs:79,173): 2 (default) = frame begin/end and a pair around the primary
 * kernel and around the bounce kernels of every batch; 1 = frame begin/end only; 0 = none (the *_ms fields
 * of rrt_stats stay 0).  An event between two launches idles the GPU for ~3 us: a caller that renders
 * sub-millisecond frames back to back (one rank of a tile-sharded frame) turns them off. */
int rrt_set_timing(rrt_scene* s, int level);

/* Where the kernel of the sky-only tiles runs (no reference counterpart; scheduling only, the image is the same):
 * 1 (default) = on a side stream of the frame's stream, forked off behind the primary kernel, so that it fills what
 * the bounce levels leave idle -- what a caller that renders one frame at a time wants (1080p/64 spp: 3.56 ms
 * against 3.9); 0 = on the frame's own stream, in front of the primary kernel.  With several frames in flight on
 * several streams either can be the faster one by 4-18 %: the runtime deals streams to a few hardware queues, and
 * which streams share one depends on every stream the process has created (an RCCL communicator made before the
 * first frame turns the order round).  A caller that pipelines measures both once (bench.py, TilePipeline.autotune). */
int rrt_set_sky_stream(rrt_scene* s, int mode);

/* Uploads the scene to the device (idempotent; render calls it if needed). */
int rrt_scene_commit(rrt_scene* s);

/* RayTracer::render (src/tracer.rs:77-175) minus the PNG save: blocking.
 * out_rgb8  [H][W][3] bytes, vec_rgb quantisation (common.rs:75-81)   (may be NULL)
 * out_ids   [H][W][2] int32 {entity id, triangle id} of the sample-0 primary
 *           hit; entity -1 = sky, triangle -1 = not a mesh            (may be NULL)
 * out_t     [H][W] f64 hit parameter t of that hit (0 for sky)        (may be NULL)
 * out_color [H][W][3] f64 un-quantised mean colour                    (may be NULL)
 * Requires tile_world == 1.  The frame runs on a stream of the scene's own and the call waits for that stream only
 * (not for the device).  Destinations in page-locked memory (rrt_alloc_host_pinned) are filled band by band while
 * the frame is still being resolved; pageable destinations by one copy after it. */
int rrt_render(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, uint8_t* out_rgb8, int32_t* out_ids,
               double* out_t, double* out_color);

/* Page-locked host memory for rrt_render's outputs (hipHostMalloc; no reference counterpart: the reference's image
 * is a Vec inside `render`, src/tracer.rs:112).  rrt_render recognises ANY page-locked destination (these, a
 * hipHostMalloc of the host's own, a hipHostRegister-ed range): its RGB8 frame is then resolved in four bands of tile
 * rows, each copied into out_rgb8 by the copy engine while the next is resolved, instead of one staged copy of
 * pageable memory after the frame (1080p: ~0.1 ms against ~0.4 ms on top of the kernels).  NULL on failure. */
void* rrt_alloc_host_pinned(uint64_t bytes);
void rrt_free_host_pinned(void* p);

/* Same with DEVICE output pointers, asynchronous on `stream` (a hipStream_t,
 * NULL = default stream).  With tile_world > 1 d_rgb8 is this rank's compact
 * tile buffer [rrt_tile_slots()][RRT_TILE_H][RRT_TILE_W][3] (slot j = tile
 * j*tile_world + tile_rank) and the AOV pointers must be NULL.
 * Frames in flight: the scene keeps one set of work buffers (ray queues, level records, samples; each up to
 * the work budget) per stream it has rendered on, at most 8.  Calls on ONE stream are ordered by it; calls on
 * DIFFERENT streams may overlap on the GPU -- the thin end of one frame's launch chain (bounce levels,
 * accumulate) then runs under the next frame's primary kernel: measured 1.76 -> 1.49 ms per 1080p/16spp frame
 * with three streams, 1.02 -> 0.75 ms for one rank's eighth of the 64-spp frame.  A ninth stream returns
 * RRT_E_STATE.  rrt_get_stats reports the frame enqueued last.  Do not change or re-commit the scene while
 * frames are in flight. */
int rrt_render_device(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, void* d_rgb8, void* d_ids,
                      void* d_t, void* d_color, void* stream);

/* Gives back the work buffers the scene keeps for `stream` (see "frames in flight" above; no reference counterpart).
 * A long-lived host that creates and destroys streams calls this before it destroys one: otherwise the scene keeps
 * that stream's buffers (up to the work budget EACH) until rrt_scene_destroy, the ninth distinct stream is refused,
 * and a recycled handle value would inherit the old set.  Waits for the stream's frames first.  RRT_E_INVALID if the
 * scene has never rendered on that stream. */
int rrt_release_stream(rrt_scene* s, void* stream);

/* NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments, outputs and
 * results (bit for bit) as rrt_render_device, but the call allocates nothing: no hipMalloc /
 * hipFree, no event creation, no host container growth.  The wavefront work buffers (ray queues,
 * level records, sample colours, running sums, queue counters, statistics) are carved out of a
 * file-scope `static __device__` array of rrt_noalloc_scratch_bytes() = 256 MiB that every GPU the
 * library runs on holds until the process ends.
 *   - the scene must be committed (rrt_scene_commit) and unchanged since: otherwise RRT_E_STATE;
 *   - the frame must fit: local pixels x (24 + bytes per path of ONE sample) <= the scratch, i.e.
 *     about 800 k pixels per rank at max_bounce 2 (960x540 whole; 1080p from tile_world 3 up);
 *     larger frames return RRT_E_NOMEM with the sizes in rrt_last_error().  spp is not limited:
 *     samples are processed in as many batches as the scratch requires;
 *   - all calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 *     threads, two scenes) race.  Serialise them.  What must start at zero is cleared by the call;
 *   - the frame is not timed: rrt_get_stats reports 0 for the *_ms fields (counters are valid). */
int rrt_render_device_noalloc(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, void* d_rgb8, void* d_ids,
                              void* d_t, void* d_color, void* stream);
uint64_t rrt_noalloc_scratch_bytes(void);

/* The per-row rotation of the tile grid used for tile_world ranks (0 for one rank), see rrt_settings. */
uint32_t rrt_tile_skew(uint32_t width, uint32_t tile_world);

/* Tiles each rank owns (padded to equal size so the gather is regular). */
int64_t rrt_tile_slots(uint32_t width, uint32_t height, uint32_t tile_world);

/* Bytes of a rank's compact tile buffer: rrt_tile_slots() x 768 pixels bytes, and with tile_deal the int32 slot ->
 * tile table (image tile index ty * tiles_x + tx, -1 = padding) behind them. */
uint64_t rrt_tile_buffer_bytes(uint32_t width, uint32_t height, uint32_t tile_world, uint32_t tile_deal);

/* Root-side de-interleave after the RCCL gather: d_gathered is
 * [tile_world][slots][RRT_TILE_H][RRT_TILE_W][3]; writes row-major d_rgb8. */
int rrt_assemble_tiles(const void* d_gathered, uint32_t width, uint32_t height, uint32_t tile_world, void* d_rgb8,
                       void* stream);
/* The same for frames rendered with tile_deal = 1: d_gathered is [tile_world][rrt_tile_buffer_bytes(.., 1)]; every
 * rank's buffer names its tiles itself, so the root needs nothing but the gathered bytes. */
int rrt_assemble_tiles_dealt(const void* d_gathered, uint32_t width, uint32_t height, uint32_t tile_world, void* d_rgb8,
                             void* stream);

/* The frame's one exchange step for hosts that are not Python (no reference counterpart: the reference
 * is one process, src/tracer.rs:117-158).  One process per GPU; RCCL over xGMI inside the library,
 * resolved at run time (a process that already holds an RCCL shares it; a single-GPU host never loads it).
 *   rrt_comm_unique_id   one rank creates the 128-byte id and hands it to the others by whatever means
 *                        the host has (a file, an environment variable, MPI, a socket);
 *   rrt_comm_init        every rank, after rrt_set_device(LOCAL_RANK): ncclCommInitRank;
 *   rrt_gather_tiles     d_local = this rank's compact tile buffer (rrt_render_device with tile_world > 1),
 *                        d_gathered = [tile_world][rrt_tile_slots()][8][32][3] on the receiving rank(s);
 *                        root >= 0: every peer sends straight to that rank (xGMI is point to point: the
 *                        messages arrive side by side), root = -1: all-gather.  Asynchronous on `stream`;
 *                        follow it with rrt_assemble_tiles on the same stream. */
#define RRT_COMM_ID_BYTES 128
typedef struct rrt_comm rrt_comm;
int rrt_comm_unique_id(void* id128);
int rrt_comm_init(int rank, int world, const void* id128, rrt_comm** out);
void rrt_comm_destroy(rrt_comm* c);
int rrt_gather_tiles(rrt_comm* c, const void* d_local, void* d_gathered, uint32_t width, uint32_t height, int root,
                     void* stream);
/* ... of buffers rendered with tile_deal = 1 (rrt_tile_buffer_bytes(.., 1) bytes per rank); follow it with
 * rrt_assemble_tiles_dealt. */
int rrt_gather_tiles_dealt(rrt_comm* c, const void* d_local, void* d_gathered, uint32_t width, uint32_t height, int root,
                           void* stream);

/* The closest-hit loop of RayTracer::cast over world.entities
 * (src/tracer.rs:178-190 calling Entity::collide, common.rs:42) for caller
 * supplied rays (6 doubles each: origin, direction).  Host pointers; outputs
 * may be NULL.  ent -1 = miss. */
int rrt_trace_rays(rrt_scene* s, const double* rays, int64_t n, int32_t* ent, int32_t* tri, double* t, double* pos,
                   double* normal);

/* The shadow-ray question for caller supplied rays (the loop of PointLight::visible,
 * src/lighting.rs:77-82, in its corrected sense): out[i] = 1 iff some entity of world.entities
 * answers Entity::collide for ray i with a position closer to the origin than max_dist2[i]
 * (squared; pass DBL_MAX for "any hit").  An any-hit walk: entities in order, each kd-tree left at
 * the first accepted triangle that is close enough.  Host pointers. */
int rrt_occluded_rays(rrt_scene* s, const double* rays, const double* max_dist2, int64_t n, uint8_t* out);

/* RayTracer::cast (src/tracer.rs:177-211) for caller supplied bounce-0 rays:
 * full shading incl. reflection chain; color: 3 doubles per ray. */
int rrt_cast_rays(rrt_scene* s, const double* rays, int64_t n, uint32_t max_bounce, uint32_t prune_zero_mix,
                  double* color);

int rrt_get_stats(const rrt_scene* s, rrt_stats* out);

/* Inspection of a primary-ray coverage mask (no counterpart in the reference; the masks are an acceleration of
 * the primary rays of RayTracer::render, src/tracer.rs:117-158, that never changes a result).  Per frame and per
 * mesh of at least 256 stream entries (the first four such meshes of a scene) the renderer projects every triangle,
 * dilated by the acceptance bound of Triangle::collide (src/geometry/triangle.rs:25-58) that the conservative cull
 * uses too, through the eye onto the image and sets the pixels of its bounding rectangle; a primary ray through a
 * clear pixel cannot be accepted by any triangle of that mesh, so the mesh's Entity::collide is answered "miss"
 * without being run.  cull_mode 0 switches the masks off with the cull, cull_mode 2 verifies them (a masked ray the
 * walk answers counts in rrt_stats.cull_violations).
 * This call fills the mask of mesh entity `entity` for (cam, width, height) exactly as a render would and copies
 * it to the host: pixels[y * width + x] = 1 if pixel (x, y) is set (the mask keeps one flag per cell of 4 x 1 pixels,
 * so set pixels come in aligned runs of four), *every = 1 when some triangle reaches the eye plane and every pixel
 * counts as set whatever `pixels` says.  RRT_E_STATE if that entity carries no mask. */
int rrt_cover_mask(rrt_scene* s, const rrt_camera* cam, uint32_t width, uint32_t height, int entity, uint8_t* pixels, int* every);

/* Flattened kd-tree of a mesh entity in pre-order, for inspection and tests.
 * boxes 6/node, split 1/node, axis 1/node (-1 = leaf), child 2/node (interior:
 * left,right; leaf: first,count into leaf_refs), ropes 6/node (-1 = none;
 * index i: 0..2 = -x,-y,-z faces, 3..5 = +x,+y,+z; kdtree.rs:139-190). */
int rrt_mesh_tree_sizes(const rrt_scene* s, int entity, int32_t* n_nodes, int32_t* n_leaf_refs);
int rrt_mesh_tree_export(const rrt_scene* s, int entity, double* boxes, double* split, int32_t* axis, int32_t* child,
                         int32_t* ropes, int32_t* leaf_refs);

/* Device arithmetic self-test used by the parity tests: out[i] = a[i] op b[i]
 * computed on the GPU in f64 (op 0 = div, 1 = sqrt(a), 2 = a*b+c unfused with
 * c = b, 3 = powi(a, (int)b)).  Host pointers. */
int rrt_selftest_math(int op, const double* a, const double* b, int64_t n, double* out);

/* Extension RNG, exposed for known-answer tests (host-side evaluation of the
 * same function the kernels use). */
void rrt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
