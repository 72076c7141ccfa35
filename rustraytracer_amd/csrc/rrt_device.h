// Shared host/device plain-data layouts (HBM-resident scene + wavefront work
// buffers) and the launcher entry points implemented in rrt_kernels.hip.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/rrt.h"

namespace rrt {

// Safety bound on leaves visited per ray (the reference loops unbounded,
// kdtree.rs:199-233; a ray that exceeds it is a miss and is counted).
constexpr uint32_t kMaxLeafVisits = 1u << 16;
constexpr int kMaxLevels = 18;  // max_bounce + 2 casts per path, max_bounce <= 16

// kd-tree in three arrays (pre-order node numbering, left child = index + 1):
//   DNodeI[node]  16 B  what find_point's descent reads (kdtree.rs:27-31): split, axis, and the
//                       right child (interior) or the leaf number (axis < 0).  Small enough
//                       (ufo: 27 KB) to be staged whole in LDS by the trace kernels.
//   DBox[node]    48 B  every node's box: find_point tests the box of the node it starts from,
//                       and rope targets may be interior nodes (kdtree.rs:23-24, 139-161).
//   DLeaf[leaf]  128 B  box again (exit test + get_plane), ropes, range in the triangle stream, content box.
struct DNodeI {
    union {
        double split;    // interior node
        int32_t run[2];  // leaf: first entry in the triangle stream, entry count (= DLeaf::first/count)
    };
    int32_t axis;    // 0..2, or -1 for a leaf
    int32_t right;   // interior: right child node; leaf: index into DLeaf[]
};
static_assert(sizeof(DNodeI) == 16, "DNodeI layout");

struct DBox {
    double mn[3];
    double mx[3];
};

struct DBox32 {
    float v[6];   // mn.x, mx.x, mn.y, mx.y, mn.z, mx.z  (axis pairs, as the packed-f32 slab test reads them)
};

// One 128-byte line per leaf: everything a visit reads -- the content box on arrival, box + ropes on the way out.
struct alignas(128) DLeaf {
    double mn[3];
    double mx[3];
    int32_t rope[6];   // node index or -1; 0..2 = -x,-y,-z faces, 3..5 = +x,+y,+z
    int32_t first;     // first entry in the triangle stream (multiple of kGroup)
    int32_t count;
    DBox32 content;    // union of the cull boxes of the leaf's entries (meshes; unused in a Scene<T> group's tree):
                       // a ray that misses it cannot be accepted by any triangle of the leaf
    int32_t pad[6];
};
static_assert(sizeof(DLeaf) == 128 && offsetof(DLeaf, content) == 80, "DLeaf layout");

// Per triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (the two subtractions of
// triangle.rs:30-31 hoisted to scene build; same IEEE results), 72 bytes.
struct DTri {
    double v0[3];
    double e1[3];
    double e2[3];
};

// DBox32 (above): f32 bounding box of one triangle-stream entry, rounded outward (cull test only).

// Inside a leaf the stream is spatially sorted (Morton order of the triangles' centroids), so that the union box
// of four consecutive entries is tight; the reference walks the leaf in ITS list order and keeps the first of equally
// distant hits (kdtree.rs:207, strict <), so that position travels with the entry and decides ties.
struct DRef {
    int32_t tri;     // triangle id (index into Model.triangles)
    int32_t rank;    // position in the reference's leaf list
};

struct DMesh {
    const DNodeI* nodes;
    const DBox* boxes;
    const DLeaf* leaves;
    const DRef* leaf_refs;      // stream entry j: its triangle id and its position in the reference's leaf list
    const DTri* tris;           // leaf-ordered stream: entry j = the triangle leaf_refs[j].tri (duplicated per
                                // leaf, so a leaf's triangles are contiguous and the next one can be prefetched)
    const double* normals;  // 3 per triangle (indexed by triangle id)
    const DBox32* cull;     // per stream entry, see "conservative cull" in rrt_kernels.hip
    const DBox32* grp;      // per group of kGroup consecutive stream entries: union of their cull boxes
                            // (tris and grp carry kStreamPad extra entries for the loops to run into)
    int32_t material;
    int32_t flags;          // kTreeStaleBoxes | (coverage-mask slot + 1) << kCoverShift (0 = the mesh has no mask)
    // constants of the conservative cull (host-computed, f64)
    double center[3];       // root box centre
    double radius;          // root box half diagonal
    double edge_max;        // S = max |e1|,|e2| over the mesh
    double abs_max;         // largest |coordinate| of the root box
};

struct DSphere {
    double c[3];
    double r2;
    int32_t material;
    int32_t pad;
};

// Material::new_lambert_material's three shaders (material.rs:28-32) in fixed slots.  Slot [1] is
// the continuation lobe: 0 = Reflection (the reference), 1 = Refraction(ior), 2 = DiffuseBounce
// (extensions, see include/rrt.h); mixR is that lobe's mix().
struct DMaterial {
    double color[3];
    double albedo;
    double mixL, mixR, mixP;
    int32_t alpha;
    int32_t lobe;
    double ior;
};

// One entry of world.light_sources (common.rs:12: Vec<Box<dyn LightSource>>), in push order -- the order Lambert and
// Phong sum them in (lambert.rs:34-43, phong.rs:37-45).
//   kind 0 DirectionalLight{direction (normalised by ::new, lighting.rs:26), color, intensity}  (lighting.rs:19-50)
//   kind 1 PointLight{position, color, brightness, attenuation}                                  (lighting.rs:52-89)
struct DLight {
    int32_t kind;
    int32_t pad;
    double v[3];       // direction / position
    double color[3];
    double a, b;       // intensity, - / brightness, attenuation
};
constexpr int32_t kMaxPointLights = 16;
constexpr int32_t kMaxLights = 32;

// A bare Triangle pushed to world.entities (or into a Scene<T>) as an Entity of its own (triangle.rs:19-58): collided by
// Triangle::collide directly -- no kd-tree around it, hence no root-box gate (aabb.rs:89) in front of the test.
struct DTriEnt {
    double v0[3], e1[3], e2[3];   // as DTri
    double normal[3];             // the stored normal, returned as is (triangle.rs:54)
    int32_t material;
    int32_t pad;
};

struct DEntity {
    int32_t kind;   // 0 sphere, 1 mesh, 2 group (Scene<T>), 3 bare triangle; | kEntityMember when it lives inside a group
    int32_t index;
};
constexpr int32_t kEntityMember = 0x100;

// Scene<T> (src/geometry/scene.rs:31-47): a kd-tree over whole entities, same node/box/leaf arrays as
// a mesh's tree; a leaf's run indexes `members` (entity ids).
struct DGroup {
    const DNodeI* nodes;
    const DBox* boxes;
    const DLeaf* leaves;
    const int32_t* members;
    int32_t flags;          // kTreeStaleBoxes
    int32_t pad;
};
// The tree went through a literal rrt_translate_entity (refit = 0; kdtree.rs:35-42 moves the split planes and forgets
// the boxes): a leaf's box need not contain the point find_point arrived with any more, so the leaf exit has to be
// the reference's full AABB::collide (aabb.rs:50-100, negative -> position (0,0,0), kdtree.rs:221-222), not its
// inside arm alone.
constexpr int32_t kTreeStaleBoxes = 1;

// Primary-ray coverage masks.  Primary rays share the eye, so "can a ray through pixel (x, y) be accepted by any
// triangle of this mesh?" has a conservative per-frame answer: k_cover projects every triangle (dilated by the same
// acceptance bound the conservative cull uses) onto the image and sets the pixels of its bounding rectangle; a
// primary ray whose pixel is clear skips Entity::collide of that mesh -- root box, set-up, walk -- with the miss
// the walk would have returned.  One BYTE per cell of kCoverCell x 1 pixels (bytes, so that k_cover can set them with
// plain stores -- every writer writes the same 1 -- instead of read-modify-write atomics; four pixels of a row, which
// is what a wave of the primary kernel covers at 16 samples a pixel), rows of DFrame::cover_stride bytes (a multiple
// of 8: a 32-pixel tile row is one aligned 8-byte word), then 8 trailing bytes whose first says "treat every cell as
// set" (a triangle at or behind the eye plane); up to kMaxCover meshes of a scene carry one.
constexpr int32_t kCoverShift = 8;
constexpr int32_t kMaxCover = 4;
constexpr int32_t kCoverCell = 4;
constexpr int32_t kCoverMinEntries = 256;   // smaller meshes are not worth a mask (their walk is a handful of leaves)

struct DScene {
    const DEntity* ents;
    const DSphere* spheres;
    const DMesh* meshes;
    const DMaterial* mats;
    const DGroup* groups;
    const DTriEnt* tri_ents;
    int32_t n_groups;
    int32_t lds_mats;        // materials the trace kernels keep in LDS: all of them (<= kMaxLdsMats), or 0

    int32_t n_ents;
    int32_t has_light;
    double light_dir[3];
    double light_color[3];
    double light_intensity;
    double ambient;
    const uint8_t* sky[6];   // RGB8 rows
    uint32_t sky_w[6];
    uint32_t sky_h[6];
    // Descent arrays the trace kernels keep in LDS (stage_scene copies st_cnt[k] nodes from st_src[k] to node
    // offset st_off[k]): the whole pre-order DNodeI array of every mesh that fits (kind 0), and for a mesh too
    // large for that the top levels of its tree in breadth-first order (kind 1, see KdTop in rrt_api.cpp).
    int32_t n_staged;
    int32_t lds_nodes;       // total nodes staged (LDS bytes = 16 * lds_nodes)
    int32_t st_off[4], st_cnt[4];
    const DNodeI* st_src[4];
    const int32_t* mesh_stage;   // [n_meshes]: +-(LDS node offset + 1), negative for a breadth-first top; 0 = not staged
    int32_t cull_mode;       // 0 = off, 1 = on, 2 = verify (run both, count cull-vs-exact disagreements)
    int32_t lds_ents;        // 1: the trace kernels keep the entity table in LDS (n_ents <= kMaxLdsEnts)
    // world.light_sources in push order.  The reference's own world holds exactly one DirectionalLight (tracer.rs:55-65):
    // then (and for a world without lights) has_light / light_* above are all the kernels read and general_lights is 0;
    // any other list -- several directional lights, point lights, any order -- sets general_lights and the shading walks
    // `lights` (the PL instantiations of the trace kernels).
    const DLight* lights;
    int32_t n_lights;
    int32_t general_lights;
    int32_t n_tri_ents;      // bare Triangle entities (DTriEnt): such worlds take the GROUPS ("general scene") kernels
    int32_t pad_;
    int32_t shadow_mode;     // PointLight::visible: 0 = as the reference wrote it, 1 = corrected (rrt_set_shadow_mode)
};

constexpr int32_t kGroup = 4;            // a leaf's run in the stream starts at a multiple of kGroup and is padded to one
constexpr int32_t kStreamPad = 4;        // readable entries past the end of DMesh::tris / DMesh::grp
constexpr int32_t kMaxLdsEnts = 64;      // 128 B of LDS per entity
constexpr int32_t kMaxLdsMats = 64;      // 72 B of LDS per material
constexpr int32_t kMaxLdsNodes = 3072;   // 48 KB of LDS per 256-thread block at most, all staged arrays together
constexpr int32_t kMaxStaged = 4;
constexpr int32_t kTopNodes = 1023;      // breadth-first top of a large tree: 10 levels, 16 KB
constexpr int32_t kTopLink = -2;         // DNodeI::axis of a top-array entry that hands over to global node `right`

// Wavefront work buffers for one batch of P paths.  Records are AoS, kRecDoubles f64 (56 B) each:
//   queue entry  {o.xyz, d.xyz, (u32 path, i32 medium)}   medium = sphere entity the ray travels inside, -1 = none
//   level record {A.xyz = color*ambient + L*mixL, Pm.xyz = P*mixP, mixR}
constexpr int kRecDoubles = 7;
constexpr uint32_t kMaxSegs = 32768;     // queue segments = blocks of the primary kernel's grid at most
struct uint2_ { uint32_t x, y; };        // (uint2 without the HIP headers: this file is also host C++)
struct DWork {
    double* q[2];        // [P][7]          ray queues (read / written alternately by the bounce levels)
    unsigned long long* q_count;   // [kMaxLevels + 1] rays queued for each bounce level (one atomic per block): a level without rays leaves on one load
    uint2_* seg_n;       // [2][kMaxSegs] per queue segment {live rays, dead rays} of the level about to be cast (parity = level & 1)
    double* rec;         // [levels][P][7]  shading records of the continuing casts
    double* samp;        // [P][3]          finished per-sample colour ...
    uint32_t* samp32;    // [P]             ... unless this says kSampTexel | sky texel (a primary ray that ended in the sky)
    double* accum;       // [3][npx_local]  running sum over samples
    unsigned long long* stats;  // [8]: 0 bounce rays traced, 1 traversal overflows, 2 cull violations (mode 2), 3 zero-mix paths, 4+l rays of level l
    uint64_t cap;        // P
};

struct DFrame {
    rrt_camera cam;
    uint32_t width, height;
    uint32_t spp, max_bounce;
    uint32_t seed_lo, seed_hi;
    uint32_t prune;
    uint32_t tile_rank, tile_world;
    uint32_t tiles_x, tiles_y;
    uint32_t n_slots;        // tiles owned (padded)
    uint32_t npx_local;      // n_slots * 256
    uint32_t s0, ns;         // first sample / samples in this batch
    uint32_t tile_skew;      // tile_skew(tiles_x, tile_world): row ty of the tile grid is rotated by tile_skew * ty columns
    uint32_t pack_log2;      // a wave of the primary kernel = 64 >> pack_log2 pixels x (1 << pack_log2) samples of this batch
    // coverage masks of this frame (see kMaxCover): n_cover masks of cover_bytes bytes each, or n_cover = 0
    const uint8_t* cover;
    uint32_t n_cover, cover_stride, cover_bytes;
    // Tile bins of this frame, or nullptr: one 64-bit word per 32x8-pixel tile of the image (row-major over the tile
    // grid), bit e clear = no primary ray through that tile can be answered by entity e (a sphere whose projected
    // bounds miss the tile, a mesh whose coverage mask is clear over it).  Scenes of at most 64 entities.
    const uint64_t* tile_bins;
    const uint8_t* tile_cost;    // beside the bins: a cost estimate per tile (0..127: cells of 4 pixels in which something can be met), k_bins
    // Tile lists of this frame (k_lists; present whenever tile_bins is).  `order`: the rank's tile slots in processing
    // order -- first the list_n[0] slots whose tile a primary ray can meet an entity in ("heavy": k_wave_primary), then,
    // up to list_n[1], the slots behind which there is only sky (k_sky); padding slots last.  `slot_tile` (dealt mode,
    // rrt_settings::tile_deal): the image tile (ty * tiles_x + tx) of slot j, -1 = padding -- the tiles are dealt to
    // the ranks by cost class from the bins instead of by position, see k_lists; nullptr = the arithmetic map
    // (tile_skew below).  list_n = {heavy slots, valid slots, valid pixels of this rank, -}.
    const int32_t* slot_tile;
    const int32_t* order;
    const uint32_t* list_n;
    uint32_t chunks_per_block;   // 256-path chunks a persistent block of the primary kernel is sized for
    uint32_t bounce_rays_per_block;   // a bounce level's rays per working block (k_wave_bounce picks how many of its blocks work)
    uint32_t seg_grid;           // grid the batch's primary kernel is launched with (primary_grid): the queue segments follow from it
};

// Tile t (t % tile_world = owner, t / tile_world = slot) is the tile in row ty = t / tiles_x of the image, column
// (t % tiles_x + skew * ty) % tiles_x.  Plain t % world with tiles_x = 60 and world = 8 walks down a column of tiles
// with a stride of 60 % 8 = 4: two ranks share each column, and whatever is tall and narrow in the image (the UFO)
// lands on four ranks twice as hard (measured: 1.03 ms for the lightest rank of eight, 1.16 ms for the heaviest).
// The rotation makes the stride down a column g, coprime to the world size and near 0.38 of it (3 for 8 ranks),
// so any 8 neighbouring tiles of a row AND of a column go to 8 different ranks.  0 for a single rank.
inline uint32_t tile_skew(uint32_t tiles_x, uint32_t world) {
    if (world <= 1 || tiles_x == 0) return 0;
    uint32_t best = 1;
    double best_d = 1e9;
    for (uint32_t g = 1; g < world; ++g) {
        uint32_t a = g, b = world;
        while (b) { const uint32_t t = a % b; a = b; b = t; }
        const double dd = (double)g - 0.382 * (double)world;
        if (a == 1 && (dd < 0 ? -dd : dd) < best_d) { best = g; best_d = dd < 0 ? -dd : dd; }
    }
    return (tiles_x % world + world - best) % world;      // (tiles_x - skew) % world == best
}

// ---- launchers (rrt_kernels.hip); all asynchronous on `stream` ------------
// d_sc = the same DScene in device memory (what the kernels read when built with RRT_SC_BY_PTR)
void launch_primary(const DScene& sc, const DScene* d_sc, const DWork& wk, const DFrame& fr, int32_t* d_ids, double* d_t, void* stream);
void launch_bounce(const DScene& sc, const DScene* d_sc, const DWork& wk, const DFrame& fr, uint32_t bounce, void* stream);
void launch_accumulate(const DWork& wk, const DFrame& fr, bool final_batch, uint8_t* d_rgb8, double* d_color, void* stream,
                       uint32_t tile0 = 0, uint32_t n_tiles = 0);
// Fills coverage mask `slot` of fr (zeroed by the caller) from the mesh's triangle stream (n entries).
void launch_cover(const DMesh& mesh, int32_t n_entries, const DFrame& fr, uint32_t slot, void* stream);
// Fills fr.tile_bins from the scene's spheres and the frame's coverage masks (after launch_cover).
void launch_bins(const DScene& sc, const DFrame& fr, void* stream);
// Fills fr.order / fr.list_n (and fr.slot_tile when `dealt`) from fr.tile_bins and fr.tile_cost.
void launch_lists(const DFrame& fr, bool dealt, bool all_heavy, void* stream);
uint32_t primary_chunks_per_block();
uint32_t primary_grid(const DFrame& fr);
// The sky-only slots of the batch (fr.order[list_n[0] .. list_n[1])): texel words into wk.samp32, AOVs of sample 0.
void launch_sky(const DScene& sc, const DWork& wk, const DFrame& fr, int32_t* d_ids, double* d_t, void* stream);
// Dealt mode: [world][slots x 768 B pixels, slots x int32 tile table] -> row-major frame.
void launch_assemble_dealt(const uint8_t* d_gathered, uint32_t w, uint32_t h, uint32_t world, uint8_t* d_rgb8, void* stream);
void launch_assemble(const uint8_t* d_gathered, uint32_t w, uint32_t h, uint32_t world, uint8_t* d_rgb8, void* stream);
void launch_trace_rays(const DScene& sc, const double* d_rays, int64_t n, int32_t* ent, int32_t* tri, double* t,
                       double* pos, double* nrm, unsigned long long* stats, void* stream);
void launch_cast_rays(const DScene& sc, const double* d_rays, int64_t n, uint32_t max_bounce, uint32_t prune,
                      double* color, unsigned long long* stats, void* stream);
void launch_occluded_rays(const DScene& sc, const double* d_rays, const double* d_lim2, int64_t n, uint8_t* out,
                          unsigned long long* stats, void* stream);
void launch_selftest(int op, const double* a, const double* b, int64_t n, double* out, void* stream);

// Scratch of the entry point that allocates nothing (rrt_render_device_noalloc): a file-scope
// `static __device__` array in rrt_kernels.hip; every GPU the library runs on holds its own copy
// until the process ends.  scratch_base resolves the current device's copy ONCE per device
// (hipGetSymbolAddress, device `device` must be current) and returns 0 or the hipError_t.
constexpr uint64_t kScratchBytes = 256ull << 20;
int scratch_base(int device, unsigned char** ptr);

void debug_phases(unsigned long long out[32]);   // zeros unless built with -DRRT_PROBE=1
void philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

}  // namespace rrt
