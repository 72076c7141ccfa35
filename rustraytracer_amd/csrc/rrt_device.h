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

// One kd node, 96 bytes.  Interior: axis 0..2, split, c0/c1 = children.
// Leaf: axis -1, c0/c1 = first/count into leaf_refs, rope[6].  The box is kept
// for every node because find_point tests the rope target's box, and rope
// targets may be interior nodes (kdtree.rs:23-24, 139-161).
struct DNode {
    double mn[3];
    double mx[3];
    double split;
    int32_t axis;
    int32_t c0;
    int32_t c1;
    int32_t rope[6];
    int32_t pad;
};
static_assert(sizeof(DNode) == 96, "DNode layout");

// Per triangle: v0, e1 = v1 - v0, e2 = v2 - v0 (the two subtractions of
// triangle.rs:30-31 hoisted to scene build; same IEEE results), 72 bytes.
struct DTri {
    double v0[3];
    double e1[3];
    double e2[3];
};

struct DMesh {
    const DNode* nodes;
    const int32_t* leaf_refs;
    const DTri* tris;
    const double* normals;  // 3 per triangle
    int32_t material;
    int32_t n_nodes;
};

struct DSphere {
    double c[3];
    double r2;
    int32_t material;
    int32_t pad;
};

struct DMaterial {
    double color[3];
    double albedo;
    double mixL, mixR, mixP;
    int32_t alpha;
    int32_t pad;
};

struct DEntity {
    int32_t kind;   // 0 sphere, 1 mesh
    int32_t index;
};

struct DScene {
    const DEntity* ents;
    const DSphere* spheres;
    const DMesh* meshes;
    const DMaterial* mats;
    int32_t n_ents;
    int32_t has_light;
    double light_dir[3];
    double light_color[3];
    double light_intensity;
    double ambient;
    const uint8_t* sky[6];   // RGB8 rows
    uint32_t sky_w[6];
    uint32_t sky_h[6];
};

// Ray queue, SoA.  cap entries each.
struct DQueue {
    double* o;        // [3][cap]
    double* d;        // [3][cap]
    uint32_t* path;   // [cap]
};

// Wavefront work buffers for one batch of P paths.
struct DWork {
    DQueue q[2];
    uint32_t* q_count;   // [kMaxLevels + 1] entries used per bounce level
    double* recA;        // [levels][3][P]  color*ambient + L*mixL
    double* recP;        // [levels][3][P]  P*mixP
    double* recR;        // [levels][P]     mixR
    double* samp;        // [3][P]          finished per-sample colour
    double* accum;       // [3][npx_local]  running sum over samples
    unsigned long long* stats;  // [0] rays traced, [1] traversal overflows
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
};

// ---- launchers (rrt_kernels.hip); all asynchronous on `stream` ------------
void launch_primary(const DScene& sc, const DWork& wk, const DFrame& fr, int32_t* d_ids, double* d_t, void* stream);
void launch_bounce(const DScene& sc, const DWork& wk, const DFrame& fr, uint32_t bounce, void* stream);
void launch_accumulate(const DWork& wk, const DFrame& fr, void* stream);
void launch_resolve(const DWork& wk, const DFrame& fr, uint8_t* d_rgb8, double* d_color, void* stream);
void launch_assemble(const uint8_t* d_gathered, uint32_t w, uint32_t h, uint32_t world, uint8_t* d_rgb8, void* stream);
void launch_trace_rays(const DScene& sc, const double* d_rays, int64_t n, int32_t* ent, int32_t* tri, double* t,
                       double* pos, double* nrm, unsigned long long* stats, void* stream);
void launch_cast_rays(const DScene& sc, const double* d_rays, int64_t n, uint32_t max_bounce, uint32_t prune,
                      double* color, unsigned long long* stats, void* stream);
void launch_selftest(int op, const double* a, const double* b, int64_t n, double* out, void* stream);

void philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

}  // namespace rrt
