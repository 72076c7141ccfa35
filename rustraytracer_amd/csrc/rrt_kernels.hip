// =============================================================================
// gfx950 (MI355X) kernels of the tracing hot path.  Built with
// -ffp-contract=off: the reference is rustc-compiled f64 with no FMA
// contraction and hit selection has to be bit-exact.
//
// Structure: a wavefront tracer.  One "path" = one pixel sample.  k_wave<true>
// generates primary rays, k_wave<false> consumes the ray queue of the previous
// bounce; each traces the world-level closest hit, shades, and either appends
// the reflection ray to the next queue (wave-aggregated atomic) or folds the
// path's per-bounce records back to front (tracer.rs:194-203 nests the
// reflection inside the shading sum, so the fold order is fixed) and writes the
// sample colour.  No host round trip inside a frame: bounce kernels read their
// queue length from device memory and grid-stride over it.
// =============================================================================
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "rrt_device.h"

namespace rrt {
namespace {

#define RRT_DEV __device__ __forceinline__

// Diagnostic build only (-DRRT_PROBE=1): wave-level event counters.  CNT(i) adds, once per wave
// that reaches it, 1 to g_phase[2i] and the number of active lanes to g_phase[2i+1]; with the
// static instruction counts of the blocks in the ISA this gives the issue budget of a frame
// (tools/phase_probe.py).  The shipped build (RRT_PROBE 0) contains none of this.
// The library is built from this file as TWO translation units (make -j halves the five minutes of hipcc): RRT_TU 1 =
// the bounce kernels' launcher, and so their 18 instantiations; RRT_TU 0 = every other launcher and the static scratch.
// Undefined = everything in one unit (the probe, variant and listing builds).
#ifndef RRT_TU
#define RRT_TU -1
#endif
#define RRT_TU_HAS(x) (RRT_TU < 0 || RRT_TU == (x))
// Switch (off): the two wavefront kernels take the scene block by pointer (device memory, read through scalar loads)
// instead of 500 B by value in the kernarg segment.  Measured: SGPR spills 108 -> 107 / 135 -> 127, frame times
// within noise -- kernargs are fetched by the same s_loads on demand, so nothing but the base address changes.
#ifndef RRT_SC_BY_PTR
#define RRT_SC_BY_PTR 0
#endif
#if RRT_SC_BY_PTR
#define RRT_SC_PARAM const DScene* __restrict__ scp
#define RRT_SC_BIND const DScene& sc = *scp;
#define RRT_SC_ARG(host, dev) (dev)
#else
#define RRT_SC_PARAM DScene sc
#define RRT_SC_BIND
#define RRT_SC_ARG(host, dev) (host)
#endif
#ifndef RRT_WAVES
#define RRT_WAVES 4   // waves per SIMD the trace kernels are register-bounded for
#endif
#ifndef RRT_BWAVES
#define RRT_BWAVES RRT_WAVES   // the same bound for the bounce kernels (5 and 6 measured slower: spills)
#endif
#ifndef RRT_PROBE
#define RRT_PROBE 0
#endif
#ifndef RRT_ABL
#define RRT_ABL 0
#endif
#ifndef RRT_PHILOX_VKEYS
#define RRT_PHILOX_VKEYS 1
#endif
#ifndef RRT_XCD_ORDER
#define RRT_XCD_ORDER 1
#endif
#if RRT_PROBE
__device__ unsigned long long g_phase[32];
#if RRT_PROBE == 2      // timing only: the event counters' atomics would be what is timed
#define CNT(i) do { } while (0)
#else
#define CNT(i)                                                                          \
    do {                                                                                \
        const unsigned long long m_ = __ballot(1);                                      \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) {                           \
            atomicAdd(&g_phase[2 * (i)], 1ull);                                         \
            atomicAdd(&g_phase[2 * (i) + 1], (unsigned long long)__popcll(m_));         \
        }                                                                               \
    } while (0)
#endif
// TSTAMP: cycle counter after everything outstanding has landed; TADD(i, dt): g_phase[24 + i] += dt by the wave's first lane
#define TSTAMP() (__builtin_amdgcn_s_waitcnt(0), (unsigned long long)__builtin_readcyclecounter())
#define TADD(i, dt)                                                                     \
    do {                                                                                \
        const unsigned long long m_ = __ballot(1);                                      \
        if ((int)__lane_id() == __ffsll((long long)m_) - 1) atomicAdd(&g_phase[24 + (i)], (unsigned long long)(dt)); \
    } while (0)
#else
#define CNT(i) do { } while (0)
#define TSTAMP() 0ull
#define TADD(i, dt) do { } while (0)
#endif

struct V3 {
    double x, y, z;
};
RRT_DEV V3 mk(double x, double y, double z) { return V3{x, y, z}; }
RRT_DEV V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
RRT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RRT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RRT_DEV V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
RRT_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// cgmath 0.17: dot = (x*x' + y*y') + z*z'
RRT_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RRT_DEV double mag2(V3 a) { return dot(a, a); }
RRT_DEV V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// cgmath normalize: v * (1 / |v|)
RRT_DEV V3 normalize(V3 a) { return a * (1.0 / sqrt(mag2(a))); }
RRT_DEV double comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

constexpr double kF64Max = 1.7976931348623157e308;

// f64::powi with run-time exponent == compiler-rt __powidf2 (phong.rs:42)
RRT_DEV double powi(double a, int b) {
    const bool recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

// Rust `as u8` / `as u32`
RRT_DEV uint32_t sat_u8(double v) {
    if (!(v == v) || v <= 0.0) return 0u;
    if (v >= 255.0) return 255u;
    return (uint32_t)v;
}
RRT_DEV uint32_t sat_u32(double v) {
    if (!(v == v) || v <= 0.0) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#if defined(__HIP_DEVICE_COMPILE__) && RRT_PHILOX_VKEYS
    // The key schedule is wave-uniform (seed + round * constant), so the compiler keeps all twenty round keys in SGPRs --
    // which this kernel does not have: they were spilled to VGPR lanes and read back with v_readlane (+ hazard nops)
    // in every round.  Made opaque per-lane values here, a round's keys are two v_add_u32 with a literal.
    asm volatile("" : "+v"(k0), "+v"(k1));
#endif
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------
// AABB (src/geometry/aabb.rs)
// ---------------------------------------------------------------------------
RRT_DEV bool box_contains(V3 mn, V3 mx, V3 p) {   // aabb.rs:41-46
    if (p.x > mx.x || p.x < mn.x) return false;
    if (p.y > mx.y || p.y < mn.y) return false;
    if (p.z > mx.z || p.z < mn.z) return false;
    return true;
}

RRT_DEV double axis_time(double cand, double inv) {   // aabb.rs:76-80
    const bool fin = fabs(inv) <= kF64Max;            // is_finite (false for NaN)
    return fin ? cand * inv : (cand == 0. ? -1. : kF64Max);
}

// AABB::collide (aabb.rs:50-100), general form.  inv = 1/d per component.
RRT_DEV bool box_collide(V3 mn, V3 mx, V3 o, V3 d, V3 inv, V3* pos) {
    const bool inside = box_contains(mn, mx, o);
    double cand[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double oi = comp(o, i), di = comp(d, i), lo = comp(mn, i), hi = comp(mx, i);
        if (oi < lo) {
            cand[i] = lo - oi;
            if (di < 0.) return false;
        } else if (oi > hi) {
            cand[i] = hi - oi;
            if (di > 0.) return false;
        } else if (inside) {
            cand[i] = (di > 0.) ? hi - oi : lo - oi;
        } else {
            cand[i] = -di;
        }
    }
    const double t0 = axis_time(cand[0], inv.x), t1 = axis_time(cand[1], inv.y), t2 = axis_time(cand[2], inv.z);
    if (inside) {
        double t = t0;                 // min_by: first of equal minima
        if (t1 < t) t = t1;
        if (t2 < t) t = t2;
        *pos = o + d * t;
        return true;
    }
    double t = t0;                     // max_by: last of equal maxima
    if (t1 >= t) t = t1;
    if (t2 >= t) t = t2;
    const V3 hp = o + d * t;
    if (!box_contains(mn, mx, hp + mk(0.001, 0.001, 0.001))) return false;
    *pos = hp;
    return true;
}

// AABB::collide for an origin known to satisfy contains(): only the
// `else if inside` arm of aabb.rs:56-72 can be taken, so this is the same
// function, not an approximation.  (The traversal's point always satisfies
// contains(leaf box): find_point only descends inside the box it tested.)
RRT_DEV V3 box_exit(V3 mn, V3 mx, V3 o, V3 d, V3 inv) {
    const double cx = (d.x > 0.) ? mx.x - o.x : mn.x - o.x;
    const double cy = (d.y > 0.) ? mx.y - o.y : mn.y - o.y;
    const double cz = (d.z > 0.) ? mx.z - o.z : mn.z - o.z;
    const double t0 = axis_time(cx, inv.x), t1 = axis_time(cy, inv.y), t2 = axis_time(cz, inv.z);
    double t = t0;
    if (t1 < t) t = t1;
    if (t2 < t) t = t2;
    return o + d * t;
}

// The leaf exit of kdtree.rs:221-222 when nothing is known about where the point lies (trees whose boxes went
// stale through a literal translate): the full AABB::collide; its collision flag is ignored by the caller and a
// negative result carries the position (0,0,0) (ColliderResult::negative, common.rs:57-72).
#ifndef RRT_COLD
#define RRT_COLD __device__ __forceinline__
#endif
RRT_COLD V3 leaf_exit_general(V3 mn, V3 mx, V3 o, V3 d, V3 inv) {
    V3 pos;
    if (!box_collide(mn, mx, o, d, inv, &pos)) pos = mk(0., 0., 0.);
    return pos;
}

// ---------------------------------------------------------------------------
// Triangle::collide (triangle.rs:25-58) on the pre-subtracted edges.
// ---------------------------------------------------------------------------
// Scene pointers are read out of structs in memory, so the compiler only knows them as generic
// ("flat") pointers; these casts give it global_load instead of flat_load on the hot arrays.
#define RRT_GLOBAL __attribute__((address_space(1)))
template <class T>
RRT_DEV const RRT_GLOBAL T* as_global(const T* p) { return (const RRT_GLOBAL T*)p; }
RRT_DEV V3 ld3g(const RRT_GLOBAL double* p) { return V3{p[0], p[1], p[2]}; }

struct TriData {
    V3 v0, e1, e2;
};
RRT_DEV TriData ld_tri(const RRT_GLOBAL DTri* T) { return TriData{ld3g(T->v0), ld3g(T->e1), ld3g(T->e2)}; }

RRT_DEV bool tri_collide(const TriData& T, V3 o, V3 d, double* tout) {
    const double EPS = 0.0000001;
    const V3 e1 = T.e1, e2 = T.e2, v0 = T.v0;
    const V3 h = cross(d, e2);
    const double a = dot(e1, h);
    if (a > -EPS && a < EPS) return false;
    const double f = 1.0 / a;
    const V3 s = o - v0;
    const double u = f * dot(s, h);
    if (!(0.0 <= u && u <= 1.0)) return false;
    const V3 q = cross(s, e1);
    const double v = f * dot(d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    const double t = f * dot(e2, q);
    if (t > EPS) {
        *tout = t;
        return true;
    }
    return false;
}

struct Hit {
    bool hit;
    bool sphere;     // the winner is a sphere (Refraction tracks the sphere a ray is inside)
    int32_t tri;
    int32_t mat;
    double t;
    V3 pos;
    V3 normal;
};

// What a descent ends on: the leaf number and its run in the triangle stream (DNodeI of a leaf
// carries first/count in the bytes an interior node uses for its split).
struct LeafRef {
    int32_t leaf;    // -1 = the point is outside the box find_point started from
    int32_t first, count;
};

// KDTree::find_point (kdtree.rs:23-33) starting at node `at`.
// `lnodes` != nullptr: this mesh's descent array is staged in LDS (wave-uniform choice) -- whole (TOP false), or,
// for a tree too large for that, only its top levels in breadth-first order (TOP true; only the descent from the
// root, at == 0, can use it): children of entry i at 2i+1 / 2i+2, an entry with axis kTopLink hands over to node
// `right` of the global pre-order array, so the first ~10 of the ~20 dependent node loads of a ray's first descent
// are LDS reads.
template <bool TOP = false>
RRT_DEV LeafRef find_point(const DNodeI* nodes, const DBox* boxes, const DNodeI* lnodes, int32_t at, V3 p) {
    const RRT_GLOBAL DBox* bx = as_global(boxes) + at;
    if (!TOP && lnodes) {
        if (!box_contains(ld3g(bx->mn), ld3g(bx->mx), p)) return LeafRef{-1, 0, 0};
        DNodeI n = lnodes[at];
        while (n.axis >= 0) {
            CNT(3);
            at = (comp(p, n.axis) >= n.split) ? n.right : at + 1;   // pre-order: left child = at + 1
            n = lnodes[at];
        }
        return LeafRef{n.right, n.run[0], n.run[1]};
    }
    const RRT_GLOBAL DNodeI* gn = as_global(nodes);
    double split;
    int32_t axis, right;
    if (TOP && lnodes) {
        if (!box_contains(ld3g(bx->mn), ld3g(bx->mx), p)) return LeafRef{-1, 0, 0};
        uint32_t i = 0;
        DNodeI n = lnodes[0];
        while (n.axis >= 0) {
            CNT(3);
            i = 2u * i + ((comp(p, n.axis) >= n.split) ? 2u : 1u);
            n = lnodes[i];
        }
        at = n.right;
        split = gn[at].split; axis = gn[at].axis; right = gn[at].right;
    } else {
        // global-memory descent: the first node is requested together with the box, not after its test
        split = gn[at].split; axis = gn[at].axis; right = gn[at].right;
        if (!box_contains(ld3g(bx->mn), ld3g(bx->mx), p)) return LeafRef{-1, 0, 0};
    }
    while (axis >= 0) {
        CNT(3);
        at = (comp(p, axis) >= split) ? right : at + 1;
        split = gn[at].split; axis = gn[at].axis; right = gn[at].right;
    }
    return LeafRef{right, __double2loint(split), __double2hiint(split)};
}

// get_plane (kdtree.rs:237-244)
RRT_DEV int get_plane(V3 p, V3 mn, V3 mx) {
    if (p.x < mn.x) return 0;
    if (p.y < mn.y) return 1;
    if (p.z < mn.z) return 2;
    if (p.x > mx.x) return 3;
    if (p.y > mx.y) return 4;
    return 5;
}

// ---------------------------------------------------------------------------
// Conservative cull.  Most triangles of a leaf are nowhere near the ray (ufo: ~94 % of the
// tests), and the exact f64 Moller-Trumbore test costs ~90 f64 instructions.  A triangle is
// skipped when the ray provably cannot be ACCEPTED by triangle.rs:25-58 evaluated in f64, so
// the result is unchanged bit for bit.  Argument (u = 2^-53, S = longest edge of the mesh,
// R >= |o - v0|): an accepted test has |a| >= 1e-7, hence its computed barycentrics and t differ
// from the exact ones of the ray/plane intersection by at most
//     du, dv <= 9u*R*S*|d|/1e-7 + 9u*S^2*|d|/1e-7  ~= 1e-8 * S*|d|*(R+S),    dt <= 1e-8 * S^2 * R
// so the exact line passes within 3*S*du of the triangle, at a parameter >= -dt.  The cull tests
// the ray against the triangle's box inflated by pad >= 8x that bound (plus the f32 conversion
// error of the origin), in f32 with outward-rounded boxes and a relative slack on the interval
// comparison; every rounding errs towards "keep".  cull_mode 2 runs both tests and counts
// disagreements (expected and tested: 0).
// ---------------------------------------------------------------------------
// The slab test itself runs on packed f32 pairs (v_pk_fma_f32): per axis i one fused multiply-add
// gives t0 = mn_i*inv_i + c0_i and t1 = mx_i*inv_i + c1_i, where the per-ray constants
//     c0_i = -(o_i + pad)*inv_i - sgn(inv_i)*K_i,     c1_i = -(o_i - pad)*inv_i + sgn(inv_i)*K_i
// are formed in f64 and rounded once.  K_i = 2^-21 (|o_i| + pad) |inv_i| is 4x the absolute error
// of that form (the f32 rounding of c and of inv against the product o_i*inv_i; the fma itself
// adds only a relative 2^-24, inside the 2^-18 relative slack), applied so that the entry time
// moves down and the exit time up.  An infinite inv_i (d_i = 0) makes t0/t1 NaN or a harmless
// infinity (never entry = +inf or exit = -inf): v_min/v_max drop NaN operands, so that axis
// simply stops constraining the interval.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct CullRay {
    f32x2 c[3];     // (c0_i, c1_i)
    float inv[3];
    float tmin;     // -(bound on dt)
};

// The mesh constants of the cull (DMesh::center .. abs_max) as plain per-lane values: wave-uniform,
// but they are only ever VALU operands, so they need no SGPRs (and no v_readfirstlane to get there).
struct CullConst {
    double center[3];
    double radius, edge_max, abs_max;
};
static_assert(offsetof(DMesh, abs_max) - offsetof(DMesh, center) == 5 * sizeof(double), "DMesh cull constants are contiguous");
RRT_DEV CullRay make_cull_ray(const CullConst& M, V3 o, V3 d, V3 inv) {
    const double dl = fabs(d.x) + fabs(d.y) + fabs(d.z);                       // >= |d|
    const double R = fabs(o.x - M.center[0]) + fabs(o.y - M.center[1]) + fabs(o.z - M.center[2]) + M.radius;
    const double S = M.edge_max;
    const double omax = fmax(fabs(o.x), fmax(fabs(o.y), fabs(o.z)));
    const double pad = (2.5e-7 * S * S * dl * (R + S) + 0x1p-20 * (omax + M.abs_max)) * 1.000001 + 1e-300;
    CullRay c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double oi = comp(o, i), ii = comp(inv, i);
        const double K = 0x1p-21 * (fabs(oi) + pad) * fabs(ii);
        const double sK = ii < 0.0 ? -K : K;
        c.c[i] = f32x2{(float)(-((oi + pad) * ii) - sK), (float)(-((oi - pad) * ii) + sK)};
        c.inv[i] = (float)ii;
    }
    c.tmin = -((float)(8e-8 * S * S * (R + S) * 1.000001) + 1e-30f);
    return c;
}

struct Box32 {
    f32x2 p[3];     // (mn_i, mx_i)
};
RRT_DEV Box32 ld_box32(const RRT_GLOBAL DBox32* b) {
    return Box32{{f32x2{b->v[0], b->v[1]}, f32x2{b->v[2], b->v[3]}, f32x2{b->v[4], b->v[5]}}};
}

// true = keep (the exact test must run).  NaN-safe: any NaN comparison falls to "keep".
RRT_DEV bool cull_keep(const Box32& b, const CullRay& c) {
    const f32x2 tx = __builtin_elementwise_fma(b.p[0], f32x2{c.inv[0], c.inv[0]}, c.c[0]);
    const f32x2 ty = __builtin_elementwise_fma(b.p[1], f32x2{c.inv[1], c.inv[1]}, c.c[1]);
    const f32x2 tz = __builtin_elementwise_fma(b.p[2], f32x2{c.inv[2], c.inv[2]}, c.c[2]);
    const float tlo = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fminf(tz.x, tz.y));
    const float thi = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fmaxf(tz.x, tz.y));
    const float hi = thi + 0x1p-18f * (fabsf(tlo) + fabsf(thi));
    return !(hi < tlo) && !(hi < c.tmin);
}

// Closest accepted triangle of a leaf.  Only what the comparisons need is kept live in the
// traversal loop (register pressure decides how many waves hide the memory latency); position
// and normal are rebuilt afterwards from t with the same expressions.
struct LeafBest {
    double dist, t;
    int32_t entry;   // index in the triangle stream, -1 = none
};

// One Triangle::collide + the closest-in-leaf bookkeeping of kdtree.rs:203-212.
// The stream inside a leaf is not in the reference's list order (see DRef): of equally distant hits the one that
// comes first in THAT order wins (kdtree.rs:207 is a strict <), decided by the entries' ranks -- a load that only
// ever happens on an exact tie.
RRT_DEV void leaf_test(const TriData& T, int32_t entry, const RRT_GLOBAL DRef* refs, V3 o, V3 d, LeafBest* b) {
    double t;
    if (tri_collide(T, o, d, &t)) {
        const V3 pos = o + d * t;                 // triangle.rs:53
        const double dist = mag2(pos - o);        // kdtree.rs:206
        bool better = dist < b->dist;
        if (dist == b->dist && b->entry >= 0) better = refs[entry].rank < refs[b->entry].rank;
        if (better) {
            b->dist = dist; b->t = t; b->entry = entry;
        }
    }
}

// KDTree::collide (kdtree.rs:192-235).  Returns the winning stream entry in out->entry (-1 = miss).
// ANY (shadow rays): the any-hit variant.  The caller only wants to know whether this collide() answers with a
// hit closer than lim2 (squared distance; kF64Max = "any hit at all"), so the walk returns at the first accepted
// triangle that is close enough -- out->dist < lim2 then -- instead of finishing the leaf for the closest one.
// A leaf whose accepted triangles are all farther still ends the walk (kdtree.rs:214: the first leaf with any hit
// answers), with out->entry >= 0 and out->dist >= lim2: "hit, but not an occluder".
template <int CULL, bool ANY = false>
RRT_DEV void mesh_collide(const DMesh& M, const CullConst& K, const DNodeI* lnodes, bool ltop, V3 o, V3 d, LeafBest* out,
                          unsigned long long* stats, double lim2 = 0.0) {
    out->entry = -1; out->dist = kF64Max; out->t = 0.0;
    CNT(1);
    const V3 inv = mk(1. / d.x, 1. / d.y, 1. / d.z);   // aabb.rs:74, same for every box test of this ray
    V3 point;
    {
        const RRT_GLOBAL DBox* rb = as_global(M.boxes);
        if (!box_collide(ld3g(rb->mn), ld3g(rb->mx), o, d, inv, &point)) return;
    }
    CNT(2);
    point = point + d * 0.01;
    // `lnodes`: this mesh's staged array -- the whole descent array, or (ltop) only the top of the tree, which
    // serves the descent from the root; rope targets then descend from global memory
    LeafRef lr;
    if (ltop) {
        lr = find_point<true>(M.nodes, M.boxes, lnodes, 0, point);
        lnodes = nullptr;
    } else {
        lr = find_point(M.nodes, M.boxes, lnodes, 0, point);
    }
    // (a ray that starts inside the root box is handed its EXIT point by aabb.rs:56-72, and that point + 0.01 d lies
    // outside the box: kdtree.rs:23-24 answers None -- every reflection ray off the mesh itself ends here, before the
    // cull constants are formed)
    if (lr.leaf < 0) return;
    CullRay cr;
    if (CULL != 0) cr = make_cull_ray(K, o, d, inv);
    const RRT_GLOBAL DLeaf* leaves = as_global(M.leaves);
    const RRT_GLOBAL DTri* stream = as_global(M.tris);
    const RRT_GLOBAL DRef* refs = as_global(M.leaf_refs);
    uint32_t visits = 0;
    // "While-while": the inner loop walks on through leaves in which this ray provably finds nothing (empty, or the
    // ray misses the leaf's content box -- the union of its entries' cull boxes, so every entry would be culled) and
    // only the leaves that pass get the three test phases below.  Written as "if (content) test" inside one loop, a
    // wave paid the phases on every trip in which ANY of its lanes had content (cfg5: nearly always, for a third of
    // the lanes); here it pays them once per leaf-with-content of its busiest lane.
    bool advance = false;
    for (;;) {
        const RRT_GLOBAL DLeaf* L;
        bool content;
        do {
            if (advance) {
                CNT(7);
                // the six ropes travel with the box (one round trip); the face picks among registers
                typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
                typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
                static_assert(sizeof(DLeaf) % 16 == 0 && offsetof(DLeaf, rope) % 16 == 0, "ropes are fetched as 16 + 8 bytes");
                L = leaves + lr.leaf;
                const V3 mn = ld3g(L->mn), mx = ld3g(L->mx);
                const i32x4 r03 = *(const RRT_GLOBAL i32x4*)&L->rope[0];
                const i32x2 r45 = *(const RRT_GLOBAL i32x2*)&L->rope[4];
                if (M.flags & kTreeStaleBoxes) point = leaf_exit_general(mn, mx, point, d, inv) + d * 0.001;
                else point = box_exit(mn, mx, point, d, inv) + d * 0.001;
                const int pl = get_plane(point, mn, mx);
                const int32_t nb = pl == 0 ? r03.x : pl == 1 ? r03.y : pl == 2 ? r03.z : pl == 3 ? r03.w : pl == 4 ? r45.x : r45.y;
                if (nb < 0) return;
                lr = find_point(M.nodes, M.boxes, lnodes, nb, point);
            }
            advance = true;
            if (lr.leaf < 0) return;
            if (++visits > kMaxLeafVisits) {
                atomicAdd(&stats[1], 1ull);
                return;
            }
            CNT(4);
            L = leaves + lr.leaf;
            content = lr.count > 0;
            if (CULL == 1 && content) content = cull_keep(ld_box32(&L->content), cr);
        } while (!content);
        CNT(9);   // leaves with content (probe build; shares the slot of shade_direct, which cfg5 probes ignore)
        bool content_keep = true;
        if (CULL == 2) content_keep = cull_keep(ld_box32(&L->content), cr);
        (void)content_keep;
        const int32_t first = lr.first, count = lr.count;
        LeafBest best;
        best.dist = kF64Max; best.t = 0.0; best.entry = -1;
        const RRT_GLOBAL DTri* tp = stream + first;
        if (CULL == 0) {
            // Software pipeline over the leaf's contiguous triangle stream, unrolled by two so the
            // prefetched triangle lands in the other register set (no copies): while A is tested B
            // is in flight and vice versa.  The prefetch may run up to two entries past the leaf:
            // the next leaf's entries, or the kStreamPad pad entries the host appends to both arrays.
            TriData A = ld_tri(tp);
            for (int32_t k = 0; k < count; k += 2) {
                const TriData B = ld_tri(tp + k + 1);
                leaf_test(A, first + k, refs, o, d, &best);
                A = ld_tri(tp + k + 2);
                if (k + 1 < count) leaf_test(B, first + k + 1, refs, o, d, &best);
                if (ANY && best.dist < lim2) break;
            }
        } else {
            // Three phases per chunk of up to 32 stream entries (8 groups of kGroup = 4).
            //  1. group cull: the union boxes of the chunk's groups, four loaded per trip before the
            //     first test, branch-free, one keep bit per group;
            //  2. entry cull: for each kept group (lanes advance by their own next kept group) its
            //     four 24-byte entry boxes, one keep bit per entry;
            //  3. exact: the f64 test (72-byte triangle) for the kept entries, in stream order (ties
            //     keep the first, kdtree.rs:207), again by each lane's own next kept entry -- a wave
            //     runs max-over-lanes(kept) tests, not one per entry that any lane kept.
            // Reads past the leaf stay inside the padded arrays; those bits are masked off.
            static_assert(kGroup == 4, "the leaf loop is written for groups of four");
            const RRT_GLOBAL DBox32* bp = as_global(M.cull) + first;
            const RRT_GLOBAL DBox32* gp = as_global(M.grp) + (first >> 2);
            for (int32_t base = 0; base < count; base += 32) {
                const int32_t n = count - base < 32 ? count - base : 32;
                const int32_t ng = (n + 3) >> 2;
                uint32_t gkeep = 0u;
                for (int32_t g = 0; g < ng; g += 4) {
                    const RRT_GLOBAL DBox32* q = gp + (base >> 2) + g;
                    const Box32 b0 = ld_box32(q), b1 = ld_box32(q + 1), b2 = ld_box32(q + 2), b3 = ld_box32(q + 3);
                    CNT(5);
                    const uint32_t m = (cull_keep(b0, cr) ? 1u : 0u) | (cull_keep(b1, cr) ? 2u : 0u) |
                                       (cull_keep(b2, cr) ? 4u : 0u) | (cull_keep(b3, cr) ? 8u : 0u);
                    gkeep |= m << g;
                }
                gkeep &= (1u << ng) - 1u;
                uint32_t keep = 0u;
                while (gkeep != 0u) {
                    CNT(11);
                    const int32_t g = __ffs((int)gkeep) - 1;
                    gkeep &= gkeep - 1u;
                    const RRT_GLOBAL DBox32* q = bp + base + 4 * g;
                    const Box32 b0 = ld_box32(q), b1 = ld_box32(q + 1), b2 = ld_box32(q + 2), b3 = ld_box32(q + 3);
                    const uint32_t m = (cull_keep(b0, cr) ? 1u : 0u) | (cull_keep(b1, cr) ? 2u : 0u) |
                                       (cull_keep(b2, cr) ? 4u : 0u) | (cull_keep(b3, cr) ? 8u : 0u);
                    keep |= m << (4 * g);
                }
                const uint32_t valid = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
                keep &= valid;
                if (CULL == 2 && !content_keep) keep = 0u;   // mode 1 would have walked past this leaf
                uint32_t todo = (CULL == 2) ? valid : keep;   // verify mode runs every entry
                while (todo != 0u) {
                    CNT(6);
                    const int32_t j = __ffs((int)todo) - 1;
                    todo &= todo - 1u;
                    const int32_t e = base + j;
                    const TriData T = ld_tri(tp + e);
                    if (CULL == 2 && ((keep >> j) & 1u) == 0u) {   // the exact test must agree this is no hit
                        double t;
                        if (tri_collide(T, o, d, &t)) atomicAdd(&stats[2], 1ull);
                    }
                    leaf_test(T, first + e, refs, o, d, &best);
                    if (ANY && best.dist < lim2) break;
                }
                if (ANY && best.dist < lim2) break;
            }
        }
        if (best.entry >= 0) {
            *out = best;
            return;
        }
    }
}

// Sphere::collide (sphere.rs:28-47): the hit parameter only; position and normal are rebuilt by
// finish_hit for the entity that wins.
RRT_DEV bool sphere_collide(const DSphere& S, V3 o, V3 d, double* tout) {
    const V3 l = ld3(S.c) - o;
    const double tca = dot(l, d);
    if (tca < 0.0) return false;
    const double d2 = mag2(l) - tca * tca;
    const double r2 = S.r2;
    if (d2 > r2) return false;
    const double thc = sqrt(r2 - d2);
    *tout = tca - thc;
    return true;
}

// EXTENSION: the sphere the ray travels inside (entered through a Refraction lobe) answers with
// its far root; sphere.rs:28-47 only ever returns the near one.
RRT_DEV bool sphere_collide_inside(const DSphere& S, V3 o, V3 d, double* tout) {
    const V3 l = ld3(S.c) - o;
    const double tca = dot(l, d);
    const double d2 = mag2(l) - tca * tca;
    const double r2 = S.r2;
    if (d2 > r2) return false;
    const double thc = sqrt(r2 - d2);
    const double t = tca + thc;
    if (!(t > 0.0)) return false;
    *tout = t;
    return true;
}

struct Closest {
    double dist, t;
    int32_t ent;     // entity id, -1 = miss
    int32_t entry;   // mesh: winning stream entry
};

// What a trace block keeps in LDS (stage_scene): the descent array of one mesh, and the world's
// entity table -- one kEntRec-dword record per entity: kind, index, then the DSphere / DMesh /
// DGroup it refers to.  Read from global memory these are two dependent scalar loads per entity and
// ray (entity -> sphere/mesh struct), ~600 cycles a wave cannot overlap with anything: 0.1 ms per
// entity on a 1080p/16spp frame.  The records are wave-uniform, so they go back into SGPRs.
#define RRT_LDS __attribute__((address_space(3)))
constexpr int32_t kEntRec = 36;        // kind, index, the struct (<= 30 dwords), then for a mesh its mesh_stage word
constexpr int32_t kEntRecStage = 32;
static_assert(sizeof(DSphere) <= 4 * (kEntRecStage - 2) && sizeof(DMesh) <= 4 * (kEntRecStage - 2) && sizeof(DGroup) <= 4 * (kEntRecStage - 2) &&
              sizeof(DTriEnt) <= 4 * (kEntRecStage - 2), "entity record too small");
// SMALL (kernel template parameter, = DScene::lds_ents): the entity table and the materials are in
// LDS too.  A compile-time choice, so that the kernels of small scenes carry no global-memory
// fallback (its five base pointers alone would sit in ten SGPRs through the whole kernel).
template <bool SMALL>
struct StagedT {
    static constexpr bool kStaged = true;
    static constexpr bool kSmall = SMALL;
    const DNodeI* nodes;             // base of the staged descent arrays (DScene::st_*), or nullptr
    const RRT_LDS uint32_t* ents;    // SMALL: entity records
    const RRT_LDS uint32_t* mats;    // SMALL: the scene's materials (sizeof(DMaterial) each)
    const RRT_LDS double* by255;     // i / 255.0 for i = 0..255 (cubemap.rs:70-74)
    const RRT_LDS double* consts;    // light, ambient, camera, sky faces (kC* below)
};
// What the one-thread-per-ray test kernels pass: nothing is staged, every branch on it folds away.
struct NoLds {
    static constexpr bool kStaged = false;
    static constexpr bool kSmall = false;
    static constexpr const DNodeI* nodes = nullptr;
    static constexpr const RRT_LDS uint32_t* ents = nullptr;
    static constexpr const RRT_LDS uint32_t* mats = nullptr;
    static constexpr const RRT_LDS double* by255 = nullptr;
    static constexpr const RRT_LDS double* consts = nullptr;
};
constexpr NoLds kNoLds{};
// The block of per-frame constants a trace block keeps in LDS.  As kernel arguments they sit in
// ~60 SGPRs for the whole kernel (the SGPR file holds ~100; the rest is spilled to VGPR lanes and
// read back with v_readlane, a VALU slot each), and the six sky faces are picked per lane with
// select chains.  From LDS they arrive where they are used, as VGPR operands: indices in doubles.
constexpr int32_t kCLightDir = 0, kCLightColor = 3, kCIntensity = 6, kCAmbient = 7, kCEye = 8, kCLensZ = 11,
                  kCLensW = 12, kCLensH = 13, kCFaces = 14 /* 6 x {ptr, w, h}: 2 doubles each */,
                  kCCover = 26 /* {ptr, n_cover, cover_stride, cover_bytes, -}: 3 doubles */,
                  kCRcpW = 29 /* RN(1 / width), RN(1 / height), camera scale M (sky_texel_fast) */, kCRcpH = 30, kCScale = 31, kCDoubles = 32;
RRT_DEV V3 ld3l(const RRT_LDS double* p) { return V3{p[0], p[1], p[2]}; }

// Per-lane read of a field of an LDS-staged struct (the record index differs per lane).
template <class T>
RRT_DEV T lds_field(const RRT_LDS uint32_t* rec, size_t byte_off) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized fields only");
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; ++i) w[i] = rec[byte_off / 4 + i];
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}

template <class T>
RRT_DEV T lds_uniform(const RRT_LDS uint32_t* p) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized structs only");
    uint32_t w[sizeof(T) / 4];
#pragma unroll
    for (uint32_t i = 0; i < sizeof(T) / 4; ++i) w[i] = __builtin_amdgcn_readfirstlane(p[i]);
    T t;
    __builtin_memcpy(&t, w, sizeof(T));
    return t;
}

// Entity::collide of a sphere or a mesh + the "closer than what we have" update that both the world
// loop (tracer.rs:183-188) and a kd leaf over entities (kdtree.rs:204-211) apply: strict <, so the
// first of equals stays.
// `rec` != nullptr: entity e's LDS record (e is wave-uniform then); nullptr: read from global memory.
// GEN: the "general scene" instantiations (kernel template parameter GROUPS: Scene<T> groups, point lights or bare
// Triangle entities present) also know entity kind 3; the kernels of the reference's own worlds carry none of it.
template <int CULL, bool REC, bool ANY = false, bool GEN = false>
RRT_DEV void entity_collide(const DScene& sc, const DNodeI* lnodes, const RRT_LDS uint32_t* rec, int32_t e, DEntity ent,
                            V3 o, V3 d, int32_t medium, Closest* best, unsigned long long* stats, double lim2 = 0.0,
                            uint32_t cov = ~0u) {
    if ((RRT_ABL & 4) && ent.kind == 0) return;
    if ((RRT_ABL & 2) && ent.kind != 0) return;
    if (GEN && ent.kind == 3) {
        // Triangle::collide (triangle.rs:25-58) of a Triangle that is an entity of its own: the test itself, nothing
        // in front of it (a mesh of one triangle would put its kd root box and the +0.001 nudge of aabb.rs:89 there)
        DTriEnt T;
        if constexpr (REC) T = lds_field<DTriEnt>(rec + 2, 0);
        else T = sc.tri_ents[ent.index];
        const TriData td{ld3(T.v0), ld3(T.e1), ld3(T.e2)};
        double t;
        if (tri_collide(td, o, d, &t)) {
            const V3 pos = o + d * t;                  // triangle.rs:53
            const double dist = mag2(pos - o);         // tracer.rs:184
            if (dist < best->dist) { best->dist = dist; best->t = t; best->ent = e; best->entry = -1; }
        }
        return;
    }
    if (ent.kind == 0) {
        double t;
        DSphere S;
        if constexpr (REC) S = lds_field<DSphere>(rec + 2, 0);   // VALU operands only: straight from LDS into VGPRs
        else S = sc.spheres[ent.index];
        const bool got = (e == medium) ? sphere_collide_inside(S, o, d, &t) : sphere_collide(S, o, d, &t);
        if (got) {
            const V3 pos = o + d * t;                  // sphere.rs:39
            const double dist = mag2(pos - o);         // tracer.rs:184
            if (dist < best->dist) { best->dist = dist; best->t = t; best->ent = e; best->entry = -1; }
        }
    } else {
        DMesh M;
        CullConst K;
        if constexpr (REC) {
            M = lds_uniform<DMesh>(rec + 2);      // the array bases become SGPRs; what is not used is never read
            K = lds_field<CullConst>(rec + 2, offsetof(DMesh, center));
        } else {
            M = sc.meshes[ent.index];
            K = CullConst{{M.center[0], M.center[1], M.center[2]}, M.radius, M.edge_max, M.abs_max};
        }
        const DNodeI* ln = nullptr;
        bool ltop = false;
        if (lnodes) {     // is this mesh's descent array (or its top) among the staged ones?
            int32_t so;     // +-(LDS node offset + 1), negative = the array is a breadth-first top; 0 = not staged
            if constexpr (REC) so = __builtin_amdgcn_readfirstlane((int)rec[kEntRecStage]);
            else so = sc.mesh_stage[ent.index];
            ltop = so < 0;
            if (so != 0) ln = lnodes + ((ltop ? -so : so) - 1);
        }
        // primary rays: the frame's coverage mask of this mesh says no triangle can accept a ray through this pixel
        // (cov is all ones for every other ray, and the test folds away)
        bool masked = false;
        if (CULL != 0 && !ANY) {
            const int32_t cs = (M.flags >> kCoverShift) & 15;
            masked = cs != 0 && ((cov >> (cs - 1)) & 1u) == 0u;
            if (CULL == 1 && masked) return;
        }
        LeafBest h;
        mesh_collide<CULL, ANY>(M, K, ln, ltop, o, d, &h, stats, lim2);
        if (CULL == 2 && masked && h.entry >= 0) atomicAdd(&stats[2], 1ull);    // verify mode: the mask must have been right
        // same |pos - origin|^2 as the leaf-level metric (kdtree.rs:206 == tracer.rs:184)
        if (h.entry >= 0 && h.dist < best->dist) {
            best->dist = h.dist; best->t = h.t; best->ent = e; best->entry = h.entry;
        }
    }
}

// Scene<T>::collide (scene.rs:50-52) = KDTree<T>::collide (kdtree.rs:192-235) with whole entities in
// the leaves: the closest member of the first leaf in which any member answers.  `out` receives
// that member (its entity id), or ent = -1.
// NEST (the instantiations that also carry the light list: a world with nested scenes takes those): a member may be a
// Scene<T> itself (scene.rs:32 is generic over any Entity, itself included) -- one level, DEPTH 0 -> 1.  Its answer is
// its own collide(), i.e. this function again on the member's tree; the outer leaf compares it like any other member's.
template <int CULL, bool ANY = false, bool NEST = false, int DEPTH = 0, class LDS = NoLds>
RRT_DEV void group_collide(const DScene& sc, LDS lds, const DGroup& G, const DNodeI* lnodes, V3 o, V3 d, int32_t medium,
                           Closest* out, unsigned long long* stats, double lim2 = 0.0, uint32_t cov = ~0u, uint64_t ebits = ~0ull) {
    out->dist = kF64Max; out->t = 0.0; out->ent = -1; out->entry = -1;
    const V3 inv = mk(1. / d.x, 1. / d.y, 1. / d.z);
    V3 point;
    {
        const RRT_GLOBAL DBox* rb = as_global(G.boxes);
        if (!box_collide(ld3g(rb->mn), ld3g(rb->mx), o, d, inv, &point)) return;
    }
    point = point + d * 0.01;
    LeafRef lr = find_point(G.nodes, G.boxes, nullptr, 0, point);
    const RRT_GLOBAL int32_t* members = as_global(G.members);
    uint32_t visits = 0;
    while (lr.leaf >= 0) {
        if (++visits > kMaxLeafVisits) {
            atomicAdd(&stats[1], 1ull);
            return;
        }
        Closest best;
        best.dist = kF64Max; best.t = 0.0; best.ent = -1; best.entry = -1;
        for (int32_t k = 0; k < lr.count; ++k) {
            const int32_t e = members[lr.first + k];
            if constexpr (LDS::kSmall) {
                // The member differs per lane in principle, but the lanes of a wave sit in the same or neighbouring
                // leaves: the wave goes through the DISTINCT members its lanes are at, one after the other, each as a
                // wave-uniform entity -- its record from the LDS table into SGPRs, its staged descent array, exactly the
                // code path of a top-level entity (from global memory, per lane, the three-entity scene of main.rs ran
                // at 11.8 Grays/s against 20.9 for the flat list).
                for (bool todo = true; todo;) {
                    const int32_t eu = __builtin_amdgcn_readfirstlane(e);
                    if (e == eu) {
                        todo = false;
                        const bool binned = CULL != 0 && !ANY && eu < 64 && ((ebits >> eu) & 1ull) == 0ull;   // see closest_hit
                        if (CULL == 1 && binned) continue;
                        const RRT_LDS uint32_t* rec = lds.ents + eu * kEntRec;
                        DEntity ent = lds_uniform<DEntity>(rec);
                        ent.kind &= ~kEntityMember;        // 0 sphere, 1 mesh, 3 bare triangle; 2 = a nested scene (NEST)
                        if constexpr (NEST && DEPTH == 0) {
                            if (ent.kind == 2) {
                                const DGroup G2 = lds_uniform<DGroup>(rec + 2);
                                Closest h;
                                group_collide<CULL, ANY, NEST, 1>(sc, lds, G2, lnodes, o, d, medium, &h, stats, lim2, cov, ebits);
                                if (CULL == 2 && binned && h.ent >= 0) atomicAdd(&stats[2], 1ull);
                                if (h.ent >= 0 && h.dist < best.dist) best = h;
                                continue;
                            }
                        }
                        entity_collide<CULL, true, ANY, true>(sc, lnodes, rec, eu, ent, o, d, medium, &best, stats, lim2, cov);
                        if (CULL == 2 && binned && best.ent == eu) atomicAdd(&stats[2], 1ull);
                    }
                }
            } else {
                DEntity ent = sc.ents[e];
                ent.kind &= ~kEntityMember;
                bool nested = false;
                if constexpr (NEST && DEPTH == 0) {
                    if (ent.kind == 2) {
                        nested = true;
                        const DGroup G2 = sc.groups[ent.index];
                        Closest h;
                        group_collide<CULL, ANY, NEST, 1>(sc, lds, G2, lnodes, o, d, medium, &h, stats, lim2, cov, ebits);
                        if (h.ent >= 0 && h.dist < best.dist) best = h;
                    }
                }
                if (!nested) entity_collide<CULL, false, ANY, true>(sc, lnodes, nullptr, e, ent, o, d, medium, &best, stats, lim2, cov);   // e differs per lane
            }
            if (ANY && best.dist < lim2) break;
        }
        if (best.ent >= 0) {
            *out = best;
            return;
        }
        const RRT_GLOBAL DLeaf* L = as_global(G.leaves) + lr.leaf;
        const V3 mn = ld3g(L->mn), mx = ld3g(L->mx);
        if (G.flags & kTreeStaleBoxes) point = leaf_exit_general(mn, mx, point, d, inv) + d * 0.001;
        else point = box_exit(mn, mx, point, d, inv) + d * 0.001;
        const int32_t nb = L->rope[get_plane(point, mn, mx)];
        if (nb < 0) return;
        lr = find_point(G.nodes, G.boxes, nullptr, nb, point);
    }
}

// The entity loop of RayTracer::cast (tracer.rs:178-190).  GROUPS: the scene holds Scene<T>
// entities (kind 2); their members carry kEntityMember and are not in world.entities.
template <int CULL, bool GROUPS, bool NEST = false, class LDS>
RRT_DEV void closest_hit(const DScene& sc, LDS lds, V3 o, V3 d, int32_t medium, Closest* best,
                         unsigned long long* stats, uint32_t cov = ~0u, uint64_t ebits = ~0ull) {
    const DNodeI* lnodes = lds.nodes;
    best->dist = kF64Max; best->t = 0.0; best->ent = -1; best->entry = -1;
    for (int32_t e = 0; e < sc.n_ents; ++e) {
        // primary rays: the tile's bin says this entity cannot answer any ray through the tile (wave-uniform: a
        // scalar branch); all ones for every other ray
        const bool binned = CULL != 0 && e < 64 && ((ebits >> e) & 1ull) == 0ull;
        if (CULL == 1 && binned) continue;
        const RRT_LDS uint32_t* rec = nullptr;
        DEntity ent;
        if constexpr (LDS::kSmall) {
            rec = lds.ents + e * kEntRec;
            ent = lds_uniform<DEntity>(rec);
        } else {
            ent = sc.ents[e];
        }
        if (GROUPS) {
            if (ent.kind & kEntityMember) continue;
            if (ent.kind == 2) {
                DGroup G;
                if constexpr (LDS::kSmall) G = lds_uniform<DGroup>(rec + 2);
                else G = sc.groups[ent.index];
                Closest h;
                group_collide<CULL, false, NEST>(sc, lds, G, lnodes, o, d, medium, &h, stats, 0.0, cov, ebits);
                if (CULL == 2 && binned && h.ent >= 0) atomicAdd(&stats[2], 1ull);    // a group is binned when all its members are
                if (h.ent >= 0 && h.dist < best->dist) *best = h;
                continue;
            }
        }
        entity_collide<CULL, LDS::kSmall, false, GROUPS>(sc, lnodes, rec, e, ent, o, d, medium, best, stats, 0.0, cov);
        if (CULL == 2 && binned && best->ent == e) atomicAdd(&stats[2], 1ull);    // verify mode: the bin must have been right
    }
}

// The any-hit question of a shadow ray: does some entity of world.entities answer the ray (o, d) with a collision
// closer than lim2 (squared distance of its position from o)?  Same entity order and Entity::collide semantics as
// closest_hit; stops at the first occluder.  LITERAL (lighting.rs:77-82 as written): returns true as soon as one
// entity does NOT collide at all ("visible"), false when every entity collides.
template <int CULL, bool GROUPS, bool LITERAL, bool NEST = false, class LDS>
RRT_DEV bool any_hit(const DScene& sc, LDS lds, V3 o, V3 d, double lim2, unsigned long long* stats) {
    const DNodeI* lnodes = lds.nodes;
    for (int32_t e = 0; e < sc.n_ents; ++e) {
        const RRT_LDS uint32_t* rec = nullptr;
        DEntity ent;
        if constexpr (LDS::kSmall) {
            rec = lds.ents + e * kEntRec;
            ent = lds_uniform<DEntity>(rec);
        } else {
            ent = sc.ents[e];
        }
        Closest h;
        h.dist = kF64Max; h.t = 0.0; h.ent = -1; h.entry = -1;
        if (GROUPS) {
            if (ent.kind & kEntityMember) continue;
            if (ent.kind == 2) {
                DGroup G;
                if constexpr (LDS::kSmall) G = lds_uniform<DGroup>(rec + 2);
                else G = sc.groups[ent.index];
                group_collide<CULL, true, NEST>(sc, lds, G, lnodes, o, d, -1, &h, stats, lim2);
                if (LITERAL) { if (h.ent < 0) return true; }
                else if (h.ent >= 0 && h.dist < lim2) return true;
                continue;
            }
        }
        entity_collide<CULL, LDS::kSmall, true, GROUPS>(sc, lnodes, rec, e, ent, o, d, -1, &h, stats, lim2);
        if (LITERAL) { if (h.ent < 0) return true; }
        else if (h.ent >= 0 && h.dist < lim2) return true;
    }
    return false;
}

// PointLight::visible.  shadow_mode 0: lighting.rs:70-84 exactly as written -- the ray starts on the surface, its
// direction (light - pos) is not normalised, and the answer is "visible" as soon as ONE entity is missed (inverted
// with respect to a shadow test; restated because it is what the code says).  shadow_mode 1 (EXTENSION, see
// rrt_set_shadow_mode): the surface must face the light, the ray starts at pos + normal*0.3, is normalised, and the
// light is visible iff nothing is hit closer than the light.
template <int CULL, bool GROUPS, bool NEST = false, class LDS>
RRT_DEV bool point_light_visible(const DScene& sc, LDS lds, V3 lpos, V3 pos, V3 normal, unsigned long long* stats) {
    if (sc.shadow_mode == 0) return any_hit<CULL, GROUPS, true, NEST>(sc, lds, pos, lpos - pos, kF64Max, stats);
    if (!(dot(normal, pos - lpos) < 0.)) return false;
    const V3 o = pos + normal * 0.3;
    const V3 v = lpos - o;
    return !any_hit<CULL, GROUPS, false, NEST>(sc, lds, o, normalize(v), mag2(v), stats);
}

// Rebuilds the ColliderResult of the winning entity: position (sphere.rs:39 / triangle.rs:53),
// normal (sphere.rs:45 / the triangle's stored normal, triangle.rs:54), material.
// With the entity table in LDS the winner's record comes from there (per-lane index): from global
// memory it is two dependent loads (entity -> its sphere / mesh struct) in front of the normal's.
template <class LDS>
RRT_DEV void finish_hit(const DScene& sc, LDS lds, V3 o, V3 d, const Closest& c, Hit* h) {
    h->hit = c.ent >= 0;
    h->tri = -1; h->mat = -1; h->t = 0.0; h->sphere = false;
    h->pos = mk(0., 0., 0.); h->normal = mk(0., 0., 0.);
    if (!h->hit) return;
    h->t = c.t;
    h->pos = o + d * c.t;
    if constexpr (LDS::kSmall) {
        const RRT_LDS uint32_t* rec = lds.ents + c.ent * kEntRec;
        const int32_t kind = (int32_t)rec[0] & ~kEntityMember;
        h->sphere = kind == 0;
        if (kind == 3) {         // a bare Triangle: its stored normal (triangle.rs:54); no triangle id (it is not part of a mesh)
            h->normal = mk(lds_field<double>(rec, 8 + offsetof(DTriEnt, normal)), lds_field<double>(rec, 8 + offsetof(DTriEnt, normal) + 8),
                           lds_field<double>(rec, 8 + offsetof(DTriEnt, normal) + 16));
            h->mat = lds_field<int32_t>(rec, 8 + offsetof(DTriEnt, material));
        } else if (h->sphere) {
            const V3 ctr = mk(lds_field<double>(rec, 8 + offsetof(DSphere, c)), lds_field<double>(rec, 8 + offsetof(DSphere, c) + 8),
                              lds_field<double>(rec, 8 + offsetof(DSphere, c) + 16));
            h->normal = normalize(h->pos - ctr);
            h->mat = lds_field<int32_t>(rec, 8 + offsetof(DSphere, material));
        } else {
            const DRef* refs = lds_field<const DRef*>(rec, 8 + offsetof(DMesh, leaf_refs));
            const double* nrm = lds_field<const double*>(rec, 8 + offsetof(DMesh, normals));
            const int32_t id = as_global(refs)[c.entry].tri;
            h->tri = id;
            h->normal = ld3g(as_global(nrm) + 3 * (size_t)id);
            h->mat = lds_field<int32_t>(rec, 8 + offsetof(DMesh, material));
        }
    } else {
        DEntity ent = sc.ents[c.ent];
        ent.kind &= ~kEntityMember;
        h->sphere = ent.kind == 0;
        if (ent.kind == 3) {
            const DTriEnt* T = sc.tri_ents + ent.index;
            h->normal = ld3(T->normal);
            h->mat = T->material;
        } else if (ent.kind == 0) {
            const DSphere* S = sc.spheres + ent.index;
            h->normal = normalize(h->pos - ld3(S->c));
            h->mat = S->material;
        } else {
            const DMesh* M = sc.meshes + ent.index;
            const int32_t id = as_global(M->leaf_refs)[c.entry].tri;
            h->tri = id;
            h->normal = ld3g(as_global(M->normals) + 3 * (size_t)id);
            h->mat = M->material;
        }
    }
}

template <class LDS>
RRT_DEV DMaterial load_material(const DScene& sc, LDS lds, int32_t mat) {
    if constexpr (LDS::kSmall) return lds_field<DMaterial>(lds.mats + mat * (int32_t)(sizeof(DMaterial) / 4), 0);
    else return sc.mats[mat];
}

// texel / 255.0 (cubemap.rs:70-74) for the 256 possible texel values, divided at compile time
// (IEEE round-to-nearest, the same quotient the device's f64 division gives): three table loads
// replace three ~30-instruction f64 divisions per sky lookup.
struct By255 {
    double v[256];
    constexpr By255() : v{} {
        for (int i = 0; i < 256; ++i) v[i] = (double)i / 255.0;
    }
};
__device__ const By255 kBy255{};

// cubemap() + CubemapBehavior::compute (cubemap.rs:84-164, 56-77)
// The texel cubemap() lands on, packed r | g << 8 | b << 16.
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
// one unaligned dword instead of three byte loads (the face buffers carry 4 spare bytes at the end)
RRT_DEV uint32_t ld_texel(const RRT_GLOBAL uint8_t* p) { return *(const RRT_GLOBAL u32_unaligned*)p & 0x00ffffffu; }
template <class LDS>
RRT_DEV const RRT_GLOBAL uint8_t* sky_texel_ptr(const DScene& sc, LDS lds, V3 dir) {
    const double x = dir.x, y = dir.y, z = dir.z;
    const double ax = fabs(x), ay = fabs(y), az = fabs(z);
    const bool xp = x > 0.0, yp = y > 0.0, zp = z > 0.0;
    double max_axis = 0.0, uc = 0.0, vc = 0.0;
    uint32_t idx = 0;
    if (xp && ax >= ay && ax >= az) { max_axis = ax; uc = -z; vc = y; idx = 0; }
    if (!xp && ax >= ay && ax >= az) { max_axis = ax; uc = z; vc = y; idx = 1; }
    if (yp && ay >= ax && ay >= az) { max_axis = ay; uc = x; vc = -z; idx = 2; }
    if (!yp && ay >= ax && ay >= az) { max_axis = ay; uc = x; vc = z; idx = 3; }
    if (zp && az >= ax && az >= ay) { max_axis = az; uc = x; vc = y; idx = 4; }
    if (!zp && az >= ax && az >= ay) { max_axis = az; uc = -x; vc = y; idx = 5; }
    const double u = 0.5 * (uc / max_axis + 1.0);
    const double v = 0.5 * (vc / max_axis + 1.0);
    uint32_t w, hgt;
    const uint8_t* face;
    if constexpr (LDS::kStaged) {
        const RRT_LDS uint32_t* f = (const RRT_LDS uint32_t*)(lds.consts + kCFaces) + 4u * idx;
        face = lds_field<const uint8_t*>(f, 0);
        w = f[2]; hgt = f[3];
    } else {
        w = sc.sky_w[idx]; hgt = sc.sky_h[idx];
        face = sc.sky[idx];
    }
    uint32_t px = sat_u32(u * (double)w), py = sat_u32(v * (double)hgt);
    if (px >= w) px = w - 1;
    if (py >= hgt) py = hgt - 1;
    return as_global(face) + ((size_t)py * w + px) * 3;
}
template <class LDS>
RRT_DEV uint32_t sky_texel(const DScene& sc, LDS lds, V3 dir) {
    if (RRT_ABL & 1) return 0u;
    return ld_texel(sky_texel_ptr(sc, lds, dir));
}
// CubemapBehavior::compute's rgb / 255.0 (cubemap.rs:70-74) through the 256-entry quotient table
template <class LDS>
RRT_DEV V3 texel_color(LDS lds, uint32_t t) {
    const uint32_t r = t & 255u, g = (t >> 8) & 255u, b = (t >> 16) & 255u;
    if constexpr (LDS::kStaged) return mk(lds.by255[r], lds.by255[g], lds.by255[b]);
    return mk(kBy255.v[r], kBy255.v[g], kBy255.v[b]);
}
template <class LDS>
RRT_DEV V3 sky_color(const DScene& sc, LDS lds, V3 dir) {
    if (RRT_ABL & 1) return dir;
    return texel_color(lds, sky_texel(sc, lds, dir));
}

// Direct terms of a hit (tracer.rs:194-202 with lambert.rs:22-45, phong.rs:25-47):
//   A  = color*ambient + Lambert*mixL      Pm = Phong*mixP
// PL: the world holds PointLights behind the directional light (light_sources order: lambert.rs:34, phong.rs:37);
// a compile-time flag, because their visible() walks the scene again (shadow rays) from inside the shading.
template <int CULL, bool GROUPS, bool PL, class LDS>
RRT_DEV void shade_direct(const DScene& sc, LDS lds, const DMaterial& m, V3 d, V3 pos, V3 normal, V3* A, V3* Pm,
                          unsigned long long* stats) {
    const V3 color = ld3(m.color);
    double ambient, intensity;
    V3 L, lcolor;
    if constexpr (LDS::kStaged) {
        ambient = lds.consts[kCAmbient]; intensity = lds.consts[kCIntensity];
        L = ld3l(lds.consts + kCLightDir); lcolor = ld3l(lds.consts + kCLightColor);
    } else {
        ambient = sc.ambient; intensity = sc.light_intensity;
        L = ld3(sc.light_dir); lcolor = ld3(sc.light_color);
    }
    V3 fin = color * ambient;
    V3 lam = mk(0., 0., 0.), ph = mk(0., 0., 0.);
    if constexpr (!PL) {
        // world.light_sources = [one DirectionalLight] (what new_empty_world builds, tracer.rs:55-65), or empty
        const bool visible = sc.has_light && (dot(normal, L) < 0.);   // lighting.rs:43-45
        if (visible) {
            double power = intensity * (m.albedo / 3.14159265358979323846) * -dot(normal, L);
            power = fmax(power, 0.);
            lam = lam + color * power;
            const V3 bis = normalize(neg(L) - d);
            const double pw = intensity * powi(fmax(dot(bis, normal), 0.), m.alpha);
            ph = ph + lcolor * pw;
        }
    } else {
        // any other list: every light in push order, each adding its Lambert term to one sum and its Phong term to
        // the other (lambert.rs:34-43, phong.rs:37-45 -- two loops over the same list, the same visible() answers)
        (void)L; (void)lcolor; (void)intensity;
        for (int32_t k = 0; k < sc.n_lights; ++k) {
            const DLight Lk = sc.lights[k];
            V3 dir;
            double lp;
            if (Lk.kind == 0) {                                           // DirectionalLight, lighting.rs:35-45
                dir = ld3(Lk.v);
                if (!(dot(normal, dir) < 0.)) continue;
                lp = Lk.a;
            } else {                                                      // PointLight, lighting.rs:60-84
                const V3 lpos = ld3(Lk.v);
                if (!point_light_visible<CULL, GROUPS, PL>(sc, lds, lpos, pos, normal, stats)) continue;
                const V3 dir0 = pos - lpos;
                const double distance2 = mag2(dir0);
                dir = normalize(dir0);
                lp = Lk.a / (Lk.b * distance2);
            }
            double power = lp * (m.albedo / 3.14159265358979323846) * -dot(normal, dir);
            power = fmax(power, 0.);
            lam = lam + color * power;
            const V3 bis = normalize(neg(dir) - d);
            const double pw = lp * powi(fmax(dot(bis, normal), 0.), m.alpha);
            ph = ph + ld3(Lk.color) * pw;
        }
    }
    *A = fin + lam * m.mixL;
    *Pm = ph * m.mixP;
}

// EXTENSION: uniform point on the unit sphere, Marsaglia's rejection in the unit disc (no trig,
// so CPU oracle and GPU agree bit for bit).  Philox counter {pixel, sample, bounce, 1 + attempt}.
RRT_DEV V3 sphere_point(const DFrame& fr, uint32_t pixel, uint32_t sample, uint32_t bounce, V3 fallback) {
    for (uint32_t j = 0; j < 4u; ++j) {
        uint32_t r[4];
        philox4x32_10(pixel, sample, bounce, 1u + j, fr.seed_lo, fr.seed_hi, r);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double x1 = (double)r[2 * h] * (1.0 / 2147483648.0) - 1.0;
            const double x2 = (double)r[2 * h + 1] * (1.0 / 2147483648.0) - 1.0;
            const double s2 = x1 * x1 + x2 * x2;
            if (s2 < 1.0) {
                const double w = 2.0 * sqrt(1.0 - s2);
                return mk(x1 * w, x2 * w, 1.0 - 2.0 * s2);
            }
        }
    }
    return fallback;
}

// Continuation ray of a hit for the material's lobe: Reflection is reflection.rs:29-33, 42-44;
// Refraction and DiffuseBounce are the extensions documented at rrt_add_material_ext.
RRT_DEV void continuation(const DFrame& fr, const DMaterial& m, V3 pos, V3 n, V3 d, bool ent_is_sphere, int32_t ent,
                          uint32_t bounce, uint32_t pixel, uint32_t sample, V3* ro, V3* rd, int32_t* medium) {
    if (m.lobe == 0) {
        *ro = pos + n * 0.3;
        *rd = normalize(d - (n * dot(n, d)) * 2.0);
        return;
    }
    const double cosi = dot(n, d);
    const bool entering = cosi < 0.0;
    const V3 nn = entering ? n : neg(n);
    if (m.lobe == 1) {
        const double cc = entering ? -cosi : cosi;
        const double eta = entering ? 1.0 / m.ior : m.ior;
        const double k = 1.0 - eta * eta * (1.0 - cc * cc);
        if (k < 0.0) {
            *ro = pos + nn * 0.3;
            *rd = normalize(d - (nn * dot(nn, d)) * 2.0);
        } else {
            *ro = pos - nn * 0.3;
            *rd = normalize(d * eta + nn * (eta * cc - sqrt(k)));
            *medium = (entering && ent_is_sphere) ? ent : -1;
        }
        return;
    }
    const V3 sp = sphere_point(fr, pixel, sample, bounce, nn);
    *ro = pos + nn * 0.3;
    *rd = normalize(nn + sp);
}

// Path-indexed work arrays hold one record per path (AoS: kRecDoubles f64 per level record and per
// queue entry, 3 per sample colour) and are addressed as (uniform base) + (32-bit byte offset of the
// lane) + immediate: one scalar base per array and one VGPR offset serve all fields of a record.
// With one array per field (SoA) every access carried its own 64-bit address and base pointer --
// 15 SGPR pairs for the stores of a continuing path, in a kernel that is short of SGPRs.
// plan_frame keeps 24 * (paths per batch) below 2^32 (sample colours; the 56-byte records: see at_rec).
typedef double f64x2 __attribute__((ext_vector_type(2)));
template <class T, class B>
RRT_DEV T& at_off(B* base, uint32_t byte_off) { return *(T*)((char*)base + byte_off); }
constexpr uint32_t kDeadSky = 0x80000000u;     // queue entry, path word: the cast that spawned this ray has mix() == +-0.0
constexpr uint32_t kSampTexel = 0x01000000u;   // samp32 tag: the low 24 bits are a sky texel (else: the colour is in samp)
// The 56-byte records (level records, queue entries) of a batch of more than 2^32 / 56 = 76.7 M paths do not fit a
// 32-bit byte offset.  Record `uidx + lidx`: uidx is the wave-uniform part of the index (the compiler keeps
// base + 56 uidx in SGPRs: scalar arithmetic), lidx the lane's (56 lidx is the 32-bit VGPR offset, as before); with
// uidx = 0 and a per-lane index the address is formed per lane in 64 bits (one v_mad_u64_u32).
constexpr uint32_t kRecBytes = (uint32_t)(kRecDoubles * 8);
template <class T, class B>
RRT_DEV T& at_rec(B* base, uint32_t uidx, uint32_t lidx, uint32_t field) {
    return *(T*)((char*)base + (uint64_t)uidx * kRecBytes + (uint32_t)(lidx * kRecBytes + field));
}
template <class T, class B>
RRT_DEV T& at_rec(B* base, uint32_t idx, uint32_t field) {
    return *(T*)((char*)base + ((uint64_t)idx * kRecBytes + field));
}
// level record {A.xyz, Pm.xyz, mixR}; UNI: the record index is uidx + lidx with uidx wave-uniform and 56 lidx < 2^32
template <bool UNI>
RRT_DEV void st_rec(double* level_base, uint32_t uidx, uint32_t lidx, V3 A, V3 Pm, double mixR) {
    if (UNI) {
        at_rec<f64x2>(level_base, uidx, lidx, 0u) = f64x2{A.x, A.y};
        at_rec<f64x2>(level_base, uidx, lidx, 16u) = f64x2{A.z, Pm.x};
        at_rec<f64x2>(level_base, uidx, lidx, 32u) = f64x2{Pm.y, Pm.z};
        at_rec<double>(level_base, uidx, lidx, 48u) = mixR;
    } else {
        at_rec<f64x2>(level_base, uidx + lidx, 0u) = f64x2{A.x, A.y};
        at_rec<f64x2>(level_base, uidx + lidx, 16u) = f64x2{A.z, Pm.x};
        at_rec<f64x2>(level_base, uidx + lidx, 32u) = f64x2{Pm.y, Pm.z};
        at_rec<double>(level_base, uidx + lidx, 48u) = mixR;
    }
}
RRT_DEV void ld_rec(double* level_base, uint32_t idx, V3* A, V3* Pm, double* mixR) {
    const f64x2 a = at_rec<f64x2>(level_base, idx, 0u), b = at_rec<f64x2>(level_base, idx, 16u),
                c = at_rec<f64x2>(level_base, idx, 32u);
    *mixR = at_rec<double>(level_base, idx, 48u);
    *A = mk(a.x, a.y, b.x);
    *Pm = mk(b.y, c.x, c.y);
}

// Queue segments.  The ray queue of a level is cut into SEGMENTS, one per block of the primary kernel: the entries
// [start, start + cap) that hold exactly the paths the block traces (its chunks x 256; the blocks' chunk counts follow
// from the launch arithmetic, so the segments tile the queue without gaps: seg_layout / seg_range).  A block appends
// the continuation rays of its paths to ITS segment -- live rays from the front, dead (mix 0.0) rays from the back,
// through one 64-bit counter in LDS (low word live, high word dead; one wave-aggregated LDS atomic per wave) -- and
// leaves the two counts in DWork::seg_n when it is done.  The bounce kernel of the next level takes the segments one
// by one: the rays of segment s are cast by one block, whose continuation rays go to segment s of the other queue
// (never more rays than came in, so they fit).  No global atomic is left on the path: the first version kept one
// counter per level in device memory, and with every wave of the chip adding to the same address -- such atomics
// are performed one after the other at the memory side -- the primary kernel of the 1080p/64spp frame spent a third
// of its time (1.25 of 3.64 ms) queueing for them.
// What a trace kernel carries through a cast to append the continuation ray with: ONE LDS address.  The block's queue
// table sits there (words): [1] n = segments of the block's region, [2..3] the 64-bit append counter, [kSegCpre + i]
// entries of the segments before i (i = 0 .. 16; [kSegCpre + 16] = the region's capacity), [kSegStart + i] first
// queue entry of segment i.  (As six values in registers -- counter address, start, capacity, two table addresses, n --
// they stayed live across the whole traversal in two kernels that spill scalars as it is.)
constexpr uint32_t kSegN = 1u, kSegCnt = 2u, kSegCpre = 70u, kSegStart = 87u;
struct SegOut {
    RRT_LDS uint32_t* tab;
};
// entry of index j of the region
RRT_DEV uint32_t seg_entry(const SegOut& seg, uint32_t j) {
    const uint32_t n = seg.tab[kSegN];
    uint32_t i = 0u;
    while (i + 1u < n && j >= seg.tab[kSegCpre + i + 1u]) ++i;
    return seg.tab[kSegStart + i] + (j - seg.tab[kSegCpre + i]);
}
RRT_DEV void queue_slots(RRT_LDS unsigned long long* counter, bool live, bool dead, uint32_t* slot_live, uint32_t* slot_dead) {
    const unsigned long long ml = __ballot(live), md = __ballot(dead);
    *slot_live = 0u; *slot_dead = 0u;
    if ((ml | md) == 0ull) return;
    uint32_t lane;         // made here, every time (hoisted out of a trace kernel's loop it is spilled to scratch)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const unsigned long long below = (1ull << lane) - 1ull;
    const int leader = __ffsll((long long)(ml | md)) - 1;
    unsigned long long base = 0ull;
    if ((int)lane == leader)
        base = __hip_atomic_fetch_add(counter, (unsigned long long)__popcll(ml) | ((unsigned long long)__popcll(md) << 32), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_WORKGROUP);
    // v_readlane with the (scalar) leader: __shfl would go through ds_bpermute and the lane's own number
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader), bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), leader);
    *slot_live = blo + (uint32_t)__popcll(ml & below);
    *slot_dead = bhi + (uint32_t)__popcll(md & below);
}

// A value every lane of the wave holds alike, but that arrived through a vector load (a count or a list entry in
// device memory, a table entry in LDS): moved into an SGPR, so that what is computed from it is scalar arithmetic and
// -- what matters in kernels that are out of registers -- does not sit in VGPRs across a whole traversal.  (Left in
// VGPRs, the unit counter's LDS address and the region's two ray counts were spilled to scratch in the bounce kernel
// and reloaded at the top of every unit, one memory round trip after the other: 14.6 k of a unit's 52 k cycles.)
RRT_DEV uint32_t wave_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
RRT_DEV unsigned long long wave_uniform64(unsigned long long v) {
    return ((unsigned long long)wave_uniform((uint32_t)(v >> 32)) << 32) | wave_uniform((uint32_t)v);
}

// How the primary kernel's launch cuts its chunks into blocks (and so the queues into segments).  host_grid = the grid
// the primary kernel was launched with (DFrame::seg_grid); with tile lists only `grid` blocks of it have work.
struct SegLayout {
    uint32_t n_work;     // tile slots the launch covers (the heavy ones with lists, else all)
    uint32_t n_chunks;   // n_work * ns
    uint32_t grid;       // blocks with work = segments
    bool by_xcd;
};
RRT_DEV SegLayout seg_layout(const DFrame& fr) {
    SegLayout L;
    L.n_work = fr.list_n ? wave_uniform(fr.list_n[0]) : fr.npx_local >> 8;
    L.n_chunks = L.n_work * fr.ns;
    L.grid = fr.seg_grid;
    if (fr.list_n) {
        uint32_t want = (L.n_chunks / fr.chunks_per_block + 2047u) / 2048u * 2048u;   // launch_primary's rule
        if (want > L.n_chunks) want = L.n_chunks;
        if (want > L.grid) want = L.grid;
        if (want >= 8u) want -= want % 8u;
        if (want == 0u) want = 1u;
        L.grid = want;
    }
    L.by_xcd = RRT_XCD_ORDER && (L.grid & 7u) == 0u && L.n_work >= 8u;
    return L;
}
// Block b of the layout: the chunk ordinals m = first, first + step, ... < limit of its list (XCD-aware order: the
// list of XCD b & 7), and its queue segment [start, start + cap) in entries.
struct SegRange {
    uint32_t first, step, limit, xcd;
    uint32_t start, cap;
};
RRT_DEV SegRange seg_range(const SegLayout& L, uint32_t ns, uint32_t b) {
    SegRange R;
    uint32_t xbase = 0u;
    if (L.by_xcd) {
        R.xcd = b & 7u;
        R.first = b >> 3;
        R.step = L.grid >> 3;
        R.limit = ((L.n_work + 7u - R.xcd) >> 3) * ns;
        // chunks of the XCDs before this one: sum over x' < xcd of ((n_work + 7 - x') >> 3) * ns
        const uint32_t q8 = L.n_work >> 3, r8 = L.n_work & 7u;
        xbase = (R.xcd * q8 + (R.xcd < r8 ? R.xcd : r8)) * ns;
    } else {
        R.xcd = 0u;
        R.first = b;
        R.step = L.grid;
        R.limit = L.n_chunks;
    }
    const uint32_t q = R.limit / R.step, r = R.limit - q * R.step;      // blocks j < r of the list take q + 1 chunks
    const uint32_t chunks = q + (R.first < r ? 1u : 0u);
    const uint32_t before = R.first * q + (R.first < r ? R.first : r);
    R.start = (xbase + before) * 256u;
    R.cap = chunks * 256u;
    return R;
}

// tracer.rs:84-110, 126-131 with the sub-pixel jitter extension.
template <class LDS>
RRT_DEV void primary_ray(const DFrame& fr, LDS lds, uint32_t x, uint32_t y, uint32_t sample, V3* o, V3* d) {
    double jx = 0.0, jy = 0.0;
    if (sample != 0u && !(RRT_ABL & 8)) {
        uint32_t r[4];
        philox4x32_10(y * fr.width + x, sample, 0u, 0u, fr.seed_lo, fr.seed_hi, r);
        jx = (double)r[0] * (1.0 / 4294967296.0);
        jy = (double)r[1] * (1.0 / 4294967296.0);
    }
    V3 eye;
    double lens_z, lens_w, lens_h;
    if constexpr (LDS::kStaged) {
        eye = ld3l(lds.consts + kCEye);
        lens_z = lds.consts[kCLensZ]; lens_w = lds.consts[kCLensW]; lens_h = lds.consts[kCLensH];
    } else {
        eye = mk(fr.cam.eye[0], fr.cam.eye[1], fr.cam.eye[2]);
        lens_z = fr.cam.lens_z; lens_w = fr.cam.lens_w; lens_h = fr.cam.lens_h;
    }
    const V3 lens_pos = eye + mk(0.0, 0.0, lens_z);
    const V3 ll = lens_pos - mk(lens_w / 2.0, lens_h / 2.0, 0.0);
    const V3 lh = mk(lens_w, 0.0, 0.0), lv = mk(0.0, lens_h, 0.0);
    const double fx = ((double)x + jx) / (double)fr.width;
    const double fy = ((double)y + jy) / (double)fr.height;
    const V3 lp = (ll + lh * fx) + lv * fy;
    *o = eye;
    *d = normalize(lp - eye);
}

// The sky texel of a primary ray WITHOUT the ray: for a primary ray that ends in the sky only the texel it lands on
// matters (its colour is texel / 255), and four primary rays of five do.  The reference's chain -- (x + jx) / W, the
// lens point, normalize (a square root and a reciprocal), two divisions uc / max, u * w truncated (tracer.rs:126-131,
// cubemap.rs:84-164, 63-74) -- costs ~280 f64 instructions a wave, most of them in five correctly rounded divisions
// and a square root whose last bits cannot matter unless the lookup lands within 1e-13 of a texel boundary.  This
// evaluates the same chain with cheap divisions (f32 reciprocal + one Newton step in f64: relative error < 2^-40) and
// without the normalisation (uc / max is scale invariant), bounds what that can move, and answers only when the
// texel is beyond doubt:
//   * fx, fy: a * RN(1/W) differs from RN(a / W) by < 2^-51 relative; the un-normalised direction v = lp - eye is
//     formed with the reference's own operations, so |vF - v| < 2^-49 M per component (M = |eye| + lens_z + lens_w +
//     lens_h); the reference's d = v * s(1 + e), |e| < 2^-51, one s > 0 for the three components.
//   * face: max = the largest |component| of vF must exceed 2^-10 M (so |vF - v| < 2^-39 max) and the other two by
//     a relative margin of 2^-30: then the same component is the largest of |d| with no tie (cubemap.rs:99-157
//     decides ties by statement order; they all go to the exact path), and its sign is the sign the reference tests.
//   * u: the true ratios of v and of vF differ by < 2^-38; the cheap division (y0 = rcp_f32(max), |y0 max - 1| <
//     2^-22; one Newton step: relative error (y0 max - 1)^2 < 2^-44) adds < 2^-43; the reference's own roundings of
//     d, the quotient, + 1.0, * 0.5 and * w add < 2^-49: |qF - q| < 2^-37, and with w <= 2^16: |tF - t| < 2^-21 texels.
//   * the answer is given when tF lies more than 2^-16 texels (thirty times that bound) inside (k, k + 1) for an
//     integer 0 <= k < w: truncation then yields k on both sides and the clamp to w - 1 is moot.
// Anything else -- a doubtful face, a texel boundary within 2^-16, a face wider than 65536, a non-finite value --
// returns false and the caller runs the exact chain.  A wave falls back about once in 10^3 (two axes, 64 lanes,
// 2^-15 of a texel each).  cull_mode 2 computes both for every primary ray that ends in the sky and counts
// disagreements in cull_violations (expected and tested: 0).
RRT_DEV bool cheap_ratio(double num, double den, double* q) {      // den > 0; |num| <= den
    const float df = (float)den;
    const double y0 = (double)__builtin_amdgcn_rcpf(df);
    const double r0 = num * y0;
    const double e = __builtin_fma(-r0, den, num);
    *q = __builtin_fma(e, y0, r0);
    return df > 0x1p-100f && df < 0x1p100f;
}
template <class LDS>
RRT_DEV bool sky_texel_fast(const DFrame& fr, LDS lds, uint32_t x, uint32_t y, uint32_t sample, const RRT_GLOBAL uint8_t** texel) {
    if constexpr (!LDS::kStaged) {
        return false;
    } else {
        if (RRT_ABL & 1) return false;
        double jx = 0.0, jy = 0.0;
        if (sample != 0u && !(RRT_ABL & 8)) {
            uint32_t r[4];
            philox4x32_10(y * fr.width + x, sample, 0u, 0u, fr.seed_lo, fr.seed_hi, r);
            jx = (double)r[0] * (1.0 / 4294967296.0);
            jy = (double)r[1] * (1.0 / 4294967296.0);
        }
        const V3 eye = ld3l(lds.consts + kCEye);
        const double lens_z = lds.consts[kCLensZ], lens_w = lds.consts[kCLensW], lens_h = lds.consts[kCLensH];
        const double M = lds.consts[kCScale];
        const double fx = ((double)x + jx) * lds.consts[kCRcpW];
        const double fy = ((double)y + jy) * lds.consts[kCRcpH];
        // tracer.rs:84-110, 128-131 with the reference's operations (primary_ray above)
        const V3 lens_pos = eye + mk(0.0, 0.0, lens_z);
        const V3 ll = lens_pos - mk(lens_w / 2.0, lens_h / 2.0, 0.0);
        const V3 lp = (ll + mk(lens_w, 0.0, 0.0) * fx) + mk(0.0, lens_h, 0.0) * fy;
        const V3 v = lp - eye;
        const double ax = fabs(v.x), ay = fabs(v.y), az = fabs(v.z);
        const double ma = fmax(ax, fmax(ay, az));
        const double lim = ma * (1.0 - 0x1p-30);
        // exactly one component above lim: the winner; (a NaN makes every comparison false: no winner)
        const bool wx = ax > lim, wy = ay > lim, wz = az > lim;
        const int winners = (wx ? 1 : 0) + (wy ? 1 : 0) + (wz ? 1 : 0);
        if (winners != 1 || !(ma > 0x1p-10 * M) || !(ma < 0x1p100)) return false;
        double uc, vc;
        uint32_t idx;
        if (wx) { const bool pos = v.x > 0.0; uc = pos ? -v.z : v.z; vc = v.y; idx = pos ? 0u : 1u; }            // cubemap.rs:99-117
        else if (wy) { const bool pos = v.y > 0.0; uc = v.x; vc = pos ? -v.z : v.z; idx = pos ? 2u : 3u; }       // :119-137
        else { const bool pos = v.z > 0.0; uc = pos ? v.x : -v.x; vc = v.y; idx = pos ? 4u : 5u; }              // :139-157
        double qu, qv;
        const bool ok = cheap_ratio(uc, ma, &qu) & cheap_ratio(vc, ma, &qv);
        const RRT_LDS uint32_t* f = (const RRT_LDS uint32_t*)(lds.consts + kCFaces) + 4u * idx;
        const uint32_t w = f[2], hgt = f[3];
        const double tu = (0.5 * (qu + 1.0)) * (double)w, tv = (0.5 * (qv + 1.0)) * (double)hgt;
        const double ku = floor(tu), kv = floor(tv);
        const double du = tu - ku, dv = tv - kv;
        // strictly inside a texel on both axes, the texel inside the face (negated comparisons: a NaN fails them)
        if (!ok || w > 65536u || hgt > 65536u || !(du > 0x1p-16 && du < 1.0 - 0x1p-16 && dv > 0x1p-16 && dv < 1.0 - 0x1p-16) ||
            !(ku >= 0.0 && ku < (double)w && kv >= 0.0 && kv < (double)hgt))
            return false;
        const uint32_t px = (uint32_t)ku, py = (uint32_t)kv;
        const uint8_t* face = lds_field<const uint8_t*>(f, 0);
        *texel = as_global(face) + ((size_t)py * w + px) * 3;
        return true;
    }
}

// Which pixel of its 32x8 tile thread t of a trace block takes.  Storage (accum, the compact tile
// buffer) stays row-major inside the tile; only the thread -> pixel assignment is swizzled, so that a
// wave covers a compact footprint: its 64 rays then mostly enter the same leaves.
//   RRT_WAVE_SHAPE 0: 32x2 strip (thread order = storage order)   1: 8x8 block   2: 16x4 block
#ifndef RRT_WAVE_SHAPE
#define RRT_WAVE_SHAPE 1
#endif
__host__ __device__ inline uint32_t tile_swz(uint32_t t) {        // thread order -> storage order
#if RRT_WAVE_SHAPE == 1
    return ((t >> 3) & 7u) * 32u + (t >> 6) * 8u + (t & 7u);
#elif RRT_WAVE_SHAPE == 2
    const uint32_t w = t >> 6;
    return ((w >> 1) * 4u + ((t >> 4) & 3u)) * 32u + (w & 1u) * 16u + (t & 15u);
#else
    return t;
#endif
}
__host__ __device__ inline uint32_t tile_unswz(uint32_t in) {     // storage order -> thread order
#if RRT_WAVE_SHAPE == 1
    const uint32_t row = in >> 5, col = in & 31u;
    return (col >> 3) * 64u + row * 8u + (col & 7u);
#elif RRT_WAVE_SHAPE == 2
    const uint32_t row = in >> 5, col = in & 31u;
    return ((row >> 2) * 2u + (col >> 4)) * 64u + (row & 3u) * 16u + (col & 15u);
#else
    return in;
#endif
}
// path-order local pixel index (what samp/records/queues are indexed by) -> storage-order index
RRT_DEV uint32_t path_to_lp(uint32_t pp) { return (pp & ~255u) | tile_swz(pp & 255u); }

// Path numbers under sample packing (DFrame::pack_log2 = p, S = 2^p).  A chunk of 256 consecutive path numbers is
// what one block iteration of the primary kernel traces, lane t = path & 255: without packing (p = 0) all 256 pixels
// of tile slot `slot` at sample `sc` (chunk = sc * n_tiles + slot).  With packing the S chunks (g S + j, slot),
// j < S, cover the same S x 256 paths differently: chunk j holds pixels t' in [j 256/S, (j+1) 256/S) of the tile at
// the S samples g S + (t & (S-1)), t' = j 256/S + (t >> p) -- a wave is 64/S neighbouring pixels at S samples, and its
// 64 records, queue entries and samples are neighbours in memory.
RRT_DEV uint32_t path_of(const DFrame& fr, uint32_t s_local, uint32_t pp) {        // pp = slot * 256 + t'
    const uint32_t p = fr.pack_log2, sm = (1u << p) - 1u, n_tiles = fr.npx_local >> 8;
    const uint32_t tp = pp & 255u, slot = pp >> 8, j = tp >> (8u - p);
    const uint32_t t = ((tp & ((1u << (8u - p)) - 1u)) << p) | (s_local & sm);
    return ((((s_local & ~sm) | j) * n_tiles + slot) << 8) | t;
}
RRT_DEV void path_decode(const DFrame& fr, uint32_t path, uint32_t* s_local, uint32_t* pp) {
    const uint32_t p = fr.pack_log2, sm = (1u << p) - 1u;
    // the divisor passes through an empty volatile asm: otherwise the compiler prepares its f32 reciprocal once per
    // kernel in VGPRs, spills both to scratch in the trace kernels, and every cast waits for their reload
    uint32_t n_tiles = fr.npx_local >> 8;
    asm volatile("" : "+s"(n_tiles));
    const uint32_t chunk = path >> 8, t = path & 255u, sc = chunk / n_tiles, slot = chunk - sc * n_tiles;
    *s_local = (sc & ~sm) | (t & sm);
    *pp = (slot << 8) | ((sc & sm) << (8u - p)) | (t >> p);
}

// Where tile slot `slot` of this rank lies in the tile grid; false = a padding slot.
RRT_DEV bool slot_pos(const DFrame& fr, uint32_t slot, uint32_t* tx, uint32_t* ty) {
    if (fr.slot_tile) {                                   // dealt mode: the table k_lists wrote
        const int32_t p = fr.slot_tile[slot];
        if (p < 0) return false;
        *ty = (uint32_t)p / fr.tiles_x;
        *tx = (uint32_t)p - *ty * fr.tiles_x;
        return true;
    }
    const uint32_t tile = slot * fr.tile_world + fr.tile_rank;
    if (tile >= fr.tiles_x * fr.tiles_y) return false;
    *ty = tile / fr.tiles_x;
    *tx = tile - *ty * fr.tiles_x;
    if (fr.tile_skew) *tx = (*tx + fr.tile_skew * *ty) % fr.tiles_x;
    return true;
}
RRT_DEV bool local_pixel(const DFrame& fr, uint32_t lp, uint32_t* x, uint32_t* y) {
    const uint32_t slot = lp >> 8, in = lp & 255u;
    uint32_t tx, ty;
    if (!slot_pos(fr, slot, &tx, &ty)) return false;
    *x = tx * RRT_TILE_W + (in & 31u);
    *y = ty * RRT_TILE_H + (in >> 5);
    return *x < fr.width && *y < fr.height;
}

// Coverage bits of the primary ray through pixel (x, y): bit k clear = no triangle of the mesh with mask slot k can
// accept this ray (see kMaxCover in rrt_device.h), so its Entity::collide is answered "miss" without being run.
template <class LDS>
RRT_DEV uint32_t cover_bits(LDS lds, uint32_t x, uint32_t y) {
    uint32_t cov = ~0u;
    if constexpr (LDS::kStaged) {
        const RRT_LDS uint32_t* cv = (const RRT_LDS uint32_t*)(lds.consts + kCCover);
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)cv[2]);
        if (n != 0u) {
            const RRT_GLOBAL uint8_t* m = as_global(lds_field<const uint8_t*>(cv, 0));
            const uint32_t stride = cv[3], bytes = cv[4];
            const uint32_t at = y * stride + (x >> 2);
#pragma unroll
            for (uint32_t k = 0; k < (uint32_t)kMaxCover; ++k) {
                if (k < n) {
                    const uint32_t c = m[k * bytes + at], all = m[k * bytes + bytes - 8u];
                    if ((c | all) == 0u) cov &= ~(1u << k);
                }
            }
        }
    }
    return cov;
}

// One cast() (tracer.rs:177-211) of path `path` at depth `bounce`.
// Appends the continuation rays of a wave's casts at depth `bounce` to the queue of depth bounce + 1 (all lanes of
// the wave call this together).  Live rays fill the queue from its front, dead (mix 0.0) rays from its back: a
// 64-ray unit of the next level is then all live or all dead.  In one mixed queue the units were mixed too (waves of
// UFO tiles and of sphere tiles append at the same time), and a unit costs what its dearest lane costs: measured
// 10.7 k cycles for an all-dead unit against 43 k for one with live rays in it.
RRT_DEV void enqueue_continuation(const DWork& wk, const SegOut& seg, uint32_t bounce, bool cont, bool zero_mix, V3 ro, V3 rd, uint32_t path,
                                  int32_t medium) {
    uint32_t slot_live, slot_dead;
    queue_slots((RRT_LDS unsigned long long*)(seg.tab + kSegCnt), cont && !zero_mix, cont && zero_mix, &slot_live, &slot_dead);
    if (cont) {
        // queue entry {o.xyz, d.xyz, (path, medium)}
        double* q = wk.q[(bounce + 1) & 1];
        const uint32_t slot = seg_entry(seg, zero_mix ? seg.tab[kSegCpre + 16u] - 1u - slot_dead : slot_live);
        at_rec<f64x2>(q, slot, 0u) = f64x2{ro.x, ro.y};
        at_rec<f64x2>(q, slot, 16u) = f64x2{ro.z, rd.x};
        at_rec<f64x2>(q, slot, 32u) = f64x2{rd.y, rd.z};
        at_rec<uint2>(q, slot, 48u) = make_uint2(path | (zero_mix ? kDeadSky : 0u), (uint32_t)medium);
    }
}

// What a cast leaves for the next level when the bounce kernel carries the path on in the same lane.
struct NextRay {
    V3 o, d;
    int32_t medium;
    bool cont, zero_mix;
};

// `next` != nullptr (bounce kernel, tail mode): the continuation ray is handed back instead of being queued.
template <bool PRIMARY, int CULL, bool GROUPS, bool PL, class LDS>
RRT_DEV void process_ray(const DScene& sc, LDS lds, const DWork& wk, const DFrame& fr, const SegOut& seg, bool active,
                         uint32_t path, V3 o, V3 d, int32_t medium, uint32_t bounce, int32_t* d_ids, double* d_t,
                         uint32_t pixel, uint32_t sample, uint32_t tag_base, bool dead_sky = false,
                         unsigned long long* tacc = nullptr, NextRay* next = nullptr, uint32_t cov = ~0u, uint64_t ebits = ~0ull,
                         uint32_t upath = 0u) {
    // upath (primary kernel): the wave-uniform part of `path` (its chunk's first path), for the record addressing
    const uint64_t P = wk.cap;
    const uint32_t tag_off = path * 4u;
    (void)tag_base;
    const bool aov = sample == 0u;
    bool cont = false, compact = false, zero_mix = false;
    V3 ro = mk(0., 0., 0.), rd = mk(0., 0., 0.);
    if (active) {
        Closest c;
        CNT(0);
        const unsigned long long tc0 = TSTAMP();
        closest_hit<CULL, GROUPS, PL>(sc, lds, o, d, medium, &c, wk.stats, cov, ebits);
        if (RRT_PROBE == 2 && tacc) *tacc += TSTAMP() - tc0;
        (void)tc0;
        Hit h;
        finish_hit(sc, lds, o, d, c, &h);
        const int32_t ent = c.ent;
        if (PRIMARY && aov) {
            if (d_ids) { d_ids[2 * (size_t)pixel] = ent; d_ids[2 * (size_t)pixel + 1] = h.tri; }
            if (d_t) d_t[pixel] = h.hit ? h.t : 0.0;
        }
        V3 col;
        uint32_t fold_from = bounce;        // records [0, fold_from) still enclose col
        if (h.hit) {
            const DMaterial m = load_material(sc, lds, h.mat);
            V3 A, Pm;
            CNT(9);
            shade_direct<CULL, GROUPS, PL>(sc, lds, m, d, h.pos, h.normal, &A, &Pm, wk.stats);
            cont = !(bounce > fr.max_bounce) && !(fr.prune && m.mixR == 0.0);
            if (cont) {
                zero_mix = m.mixR == 0.0;
                if (PRIMARY && zero_mix) {
                    // What this path's sample will be if the (dead) reflection ray ends in the sky, computed with the
                    // child's own expression on col = +0: the level-1 kernel then has nothing to read or write for
                    // it.  If the ray hits something instead, that kernel overwrites the sample.
                    const V3 c0 = (A + mk(0., 0., 0.) * m.mixR) + Pm;
                    at_off<f64x2>(wk.samp, path * 24u) = f64x2{c0.x, c0.y};
                    at_off<double>(wk.samp, path * 24u + 16u) = c0.z;
                }
                st_rec<PRIMARY>(wk.rec + (size_t)bounce * kRecDoubles * P, upath, path - upath, A, Pm, m.mixR);
                uint32_t kp = pixel, ks = sample;
                if (!PRIMARY && m.lobe == 2) {
                    // the RNG stream of the diffuse lobe is keyed by the global pixel and sample of the path
                    // (two integer divisions per lane: only paid by the rays that draw from it)
                    uint32_t s_local, pp, x = 0, y = 0;
                    path_decode(fr, path, &s_local, &pp);
                    local_pixel(fr, path_to_lp(pp), &x, &y);
                    kp = y * fr.width + x;
                    ks = fr.s0 + s_local;
                }
                continuation(fr, m, h.pos, h.normal, d, h.sphere, ent, bounce, kp, ks, &ro, &rd, &medium);
            } else {
                col = A + Pm;   // Reflection returned None (reflection.rs:26-28) or was pruned
            }
        } else if (PRIMARY) {
            // A primary ray that ends in the sky (4 of 5 do): its sample colour is texel / 255, so the three
            // texel bytes ARE the colour -- 4 bytes go to samp32 instead of 24 to samp, and k_accumulate
            // looks the quotients up.
            CNT(8);
            const RRT_GLOBAL uint8_t* tp = sky_texel_ptr(sc, lds, d);
            const uint32_t texel = (RRT_ABL & 1) ? 0u : ld_texel(tp);
            if (CULL == 2) {     // verify mode: the cheap lookup of the sky-only arm, where it answers, must name this texel
                const RRT_GLOBAL uint8_t* tf;
                if (sky_texel_fast(fr, lds, pixel % fr.width, pixel / fr.width, sample, &tf) && tf != tp) atomicAdd(&wk.stats[2], 1ull);
            }
            at_off<uint32_t>(wk.samp32, tag_off) = kSampTexel | texel;
            compact = true;
        } else {
            // A bounce ray that leaves the scene ends its path (most do).  Its innermost enclosing
            // record is requested BEFORE the sky texel: both are HBM misses on incoherent addresses,
            // and the bounce kernel lives on memory latency (22 % VALU-busy), so they should overlap.
            CNT(8);
            const uint32_t k = bounce - 1u;
            double mr;
            V3 A, Pm;
            if (dead_sky && bounce == 1u) {
                compact = true;          // the primary kernel already left this outcome in samp (see there)
            } else {
            ld_rec(wk.rec + (size_t)k * kRecDoubles * P, path, &A, &Pm, &mr);
            // dead_sky: the enclosing cast's mix() is +-0.0 (the queue entry says so), and a sky colour is
            // texel/255 >= +0, finite: its product with the mix is the same signed zero whatever the texel, so
            // the texel is not fetched.  The ray itself was cast like any other (reference ray semantics).
            col = dead_sky ? mk(0., 0., 0.) : sky_color(sc, lds, d);
            col = (A + col * mr) + Pm;
            fold_from = k;
            }
        }
        if (PRIMARY && !compact) at_off<uint32_t>(wk.samp32, tag_off) = 0u;   // "see samp" (written now or by a bounce level)
        if (!cont && !compact) {
            CNT(10);
            // fold the enclosing casts: final = (A + R*mixR) + Pm  (tracer.rs:194-202)
            for (uint32_t k = fold_from; k-- > 0;) {
                double mr;
                V3 A, Pm;
                ld_rec(wk.rec + (size_t)k * kRecDoubles * P, path, &A, &Pm, &mr);
                col = (A + col * mr) + Pm;
            }
            if (!(RRT_ABL & 16) || col.x == 12345.678) {
                at_off<f64x2>(wk.samp, path * 24u) = f64x2{col.x, col.y};
                at_off<double>(wk.samp, path * 24u + 16u) = col.z;
            }
        }
    }
    // (stats[3], the primary hits that continue with a dead (mix 0.0) ray, is the dead end of the level-1 queue: the
    // bounce kernel adds it up from the counter)
    if (!PRIMARY && next) {
        next->o = ro; next->d = rd; next->medium = medium; next->cont = cont; next->zero_mix = zero_mix;
        return;
    }
    enqueue_continuation(wk, seg, bounce, cont, zero_mix, ro, rd, path, medium);
}

// Bytes of LDS stage_scene fills (descent array, then entity records); the bounce kernel's unit
// counter sits right behind them.
__host__ __device__ inline uint32_t lds_layout_bytes(const DScene& sc) {
    return (uint32_t)sc.lds_nodes * (uint32_t)sizeof(DNodeI) +
           (sc.lds_ents ? (uint32_t)sc.n_ents * (uint32_t)kEntRec * 4u : 0u) +
           (uint32_t)sc.lds_mats * (uint32_t)sizeof(DMaterial) + 256u * 8u + (uint32_t)kCDoubles * 8u;
}

// The per-frame constants block (kC*) at `at`; thread 0 writes it, the caller's barrier publishes it.
RRT_DEV RRT_LDS double* stage_consts(const DScene& sc, const DFrame& fr, unsigned char* at) {
    RRT_LDS double* c = (RRT_LDS double*)at;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c[kCLightDir + a] = sc.light_dir[a];
            c[kCLightColor + a] = sc.light_color[a];
            c[kCEye + a] = fr.cam.eye[a];
        }
        c[kCIntensity] = sc.light_intensity; c[kCAmbient] = sc.ambient;
        c[kCLensZ] = fr.cam.lens_z; c[kCLensW] = fr.cam.lens_w; c[kCLensH] = fr.cam.lens_h;
        RRT_LDS uint32_t* f = (RRT_LDS uint32_t*)(c + kCFaces);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint64_t p = (uint64_t)sc.sky[i];
            f[4 * i] = (uint32_t)p; f[4 * i + 1] = (uint32_t)(p >> 32);
            f[4 * i + 2] = sc.sky_w[i]; f[4 * i + 3] = sc.sky_h[i];
        }
        c[kCRcpW] = 1.0 / (double)fr.width; c[kCRcpH] = 1.0 / (double)fr.height;
        c[kCScale] = fmax(fmax(fabs(fr.cam.eye[0]), fabs(fr.cam.eye[1])), fabs(fr.cam.eye[2])) + fabs(fr.cam.lens_z) + fabs(fr.cam.lens_w) + fabs(fr.cam.lens_h);
        RRT_LDS uint32_t* cv = (RRT_LDS uint32_t*)(c + kCCover);
        const uint64_t cp = (uint64_t)fr.cover;
        cv[0] = (uint32_t)cp; cv[1] = (uint32_t)(cp >> 32);
        cv[2] = fr.cover ? fr.n_cover : 0u; cv[3] = fr.cover_stride; cv[4] = fr.cover_bytes; cv[5] = 0u;
    }
    return c;
}

// Fills the block's LDS (all threads; ends with a barrier): the staged descent arrays (DScene::st_*) and,
// after them, the entity records.  Either part may be absent (see DScene::n_staged / lds_ents).
template <bool SMALL>
RRT_DEV StagedT<SMALL> stage_scene(const DScene& sc, const DFrame& fr, unsigned char* smem, bool with_nodes) {
    StagedT<SMALL> st{nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t off = 0;
    if (sc.n_staged > 0 && with_nodes) {
        // one 16-byte node per thread and trip, four trips' loads in flight before the first store
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        for (int32_t k = 0; k < sc.n_staged; ++k) {
            const RRT_GLOBAL u32x4* src = (const RRT_GLOBAL u32x4*)sc.st_src[k];
            RRT_LDS u32x4* dst = (RRT_LDS u32x4*)smem + sc.st_off[k];
            const int32_t n = sc.st_cnt[k], step = (int32_t)blockDim.x;
            for (int32_t i = threadIdx.x; i < n; i += 4 * step) {
                const int32_t i1 = i + step, i2 = i + 2 * step, i3 = i + 3 * step;
                const u32x4 v0 = src[i];
                const u32x4 v1 = src[i1 < n ? i1 : i], v2 = src[i2 < n ? i2 : i], v3 = src[i3 < n ? i3 : i];
                dst[i] = v0;
                if (i1 < n) dst[i1] = v1;
                if (i2 < n) dst[i2] = v2;
                if (i3 < n) dst[i3] = v3;
            }
        }
        st.nodes = (const DNodeI*)smem;
    }
    off = (uint32_t)sc.lds_nodes * (uint32_t)sizeof(DNodeI);
    if constexpr (SMALL) {
        RRT_LDS uint32_t* tab = (RRT_LDS uint32_t*)(smem + off);
        for (int32_t i = threadIdx.x; i < sc.n_ents * kEntRec; i += blockDim.x) {
            const int32_t e = i / kEntRec, w = i % kEntRec;
            const DEntity ent = sc.ents[e];
            const int32_t kind = ent.kind & ~kEntityMember;
            const uint32_t* src = kind == 0   ? (const uint32_t*)(sc.spheres + ent.index)
                                  : kind == 1 ? (const uint32_t*)(sc.meshes + ent.index)
                                  : kind == 3 ? (const uint32_t*)(sc.tri_ents + ent.index)
                                              : (const uint32_t*)(sc.groups + ent.index);
            const int32_t n = kind == 0 ? (int32_t)sizeof(DSphere) / 4 : kind == 1 ? (int32_t)sizeof(DMesh) / 4
                              : kind == 3 ? (int32_t)sizeof(DTriEnt) / 4 : (int32_t)sizeof(DGroup) / 4;
            uint32_t v = 0u;
            if (w == 0) v = (uint32_t)ent.kind;
            else if (w == 1) v = (uint32_t)ent.index;
            else if (w - 2 < n) v = src[w - 2];
            else if (kind == 1 && w == kEntRecStage) v = (uint32_t)sc.mesh_stage[ent.index];
            tab[i] = v;
        }
        st.ents = tab;
        off += (uint32_t)sc.n_ents * (uint32_t)kEntRec * 4u;
    }
    if constexpr (SMALL) {
        RRT_LDS uint32_t* tab = (RRT_LDS uint32_t*)(smem + off);
        const uint32_t* src = (const uint32_t*)sc.mats;
        const int32_t n = sc.lds_mats * (int32_t)(sizeof(DMaterial) / 4);
        for (int32_t i = threadIdx.x; i < n; i += blockDim.x) tab[i] = src[i];
        st.mats = tab;
        off += (uint32_t)n * 4u;
    }
    {
        // texel / 255.0 for the 256 texel values: one IEEE division per thread (the reference's own
        // expression, cubemap.rs:70-74), read back per lane at ~LDS latency instead of a global load
        RRT_LDS double* tab = (RRT_LDS double*)(smem + off);
        for (int32_t i = threadIdx.x; i < 256; i += blockDim.x) tab[i] = (double)i / 255.0;
        st.by255 = tab;
        off += 256u * 8u;
    }
    st.consts = stage_consts(sc, fr, smem + off);
    __syncthreads();
    return st;
}

// Persistent blocks: block b takes path tiles b, b + grid, ... (neighbouring tiles, which cost
// about the same, go to different blocks), so the LDS staging is paid once per block.
template <int CULL, bool GROUPS, bool SMALL, bool PL>
__global__ __launch_bounds__(256, RRT_WAVES) void k_wave_primary(RRT_SC_PARAM, DWork wk, DFrame fr, int32_t* d_ids, double* d_t) {
    RRT_SC_BIND
    const uint32_t n_tiles = fr.npx_local >> 8;
    // The slots this launch covers: with tile lists only the "heavy" ones (a primary ray through the tile can meet an
    // entity; fr.order lists them first) -- the sky-only slots are k_sky's; without lists every slot.  The host sizes
    // the grid for every slot; with lists the blocks beyond what the heavy chunks need leave at once.
    const SegLayout L = seg_layout(fr);
    if (blockIdx.x >= L.grid) return;              // block-uniform, before any barrier
    const uint32_t n_work = L.n_work;
    const SegRange R = seg_range(L, fr.ns, blockIdx.x);
    extern __shared__ __align__(16) unsigned char smem[];
    RRT_LDS uint32_t* seg_tab = (RRT_LDS uint32_t*)(smem + lds_layout_bytes(sc));
    RRT_LDS unsigned long long* seg_cnt = (RRT_LDS unsigned long long*)(seg_tab + kSegCnt);
    if (threadIdx.x == 0) {        // the block's queue table: one segment
        *seg_cnt = 0ull;
        seg_tab[kSegN] = 1u;
        seg_tab[kSegCpre] = 0u; seg_tab[kSegCpre + 1u] = R.cap; seg_tab[kSegCpre + 16u] = R.cap;
        seg_tab[kSegStart] = R.start;
    }
    const StagedT<SMALL> lds = stage_scene<SMALL>(sc, fr, smem, true);      // ends with the barrier that also publishes *seg_cnt
    const SegOut seg{seg_tab};
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  XCD x takes the list entries
    // k = x (mod 8), and its blocks walk them tile-major (all samples of a tile back to back, on
    // neighbouring blocks of the same XCD), so what one sample of a tile pulls in -- its leaves and
    // triangles, the sky texel lines behind it -- is in that L2 for the other samples.
    for (uint32_t m = R.first; m < R.limit; m += R.step) {
        uint32_t kk, s_local;                      // list entry, sample (group) of this chunk
        if (L.by_xcd) {
            const uint32_t k = m / fr.ns;
            s_local = m - k * fr.ns;
            kk = R.xcd + 8u * k;
        } else {
            s_local = m / n_work;
            kk = m - s_local * n_work;
        }
        // chunk = (sample, tile slot), or with sample packing (sample group and part of the tile, tile slot): see
        // path_of.  Wave-uniform, so the tile's position is scalar arithmetic.
        const uint32_t slot = fr.order ? wave_uniform((uint32_t)fr.order[kk]) : kk;
        const uint32_t chunk = s_local * n_tiles + slot;
        // the thread's number, opaque to the optimiser in every trip: what it would otherwise prepare from it once per
        // kernel (the lane's part of its record, queue and sample addresses, its place in the tile) are VGPRs that
        // live through the whole cast -- the record base was spilled to scratch and its reload stood, with a wait for
        // every store before it, in front of the record's stores
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const uint32_t path = chunk * 256u + tid;
        uint32_t my_t = tid;
        uint32_t tag_base = s_local * fr.npx_local + (slot << 8);
        if (fr.pack_log2) {
            const uint32_t p = fr.pack_log2, sm = (1u << p) - 1u;
            tag_base = (s_local & ~sm) * fr.npx_local + (slot << 8) + ((s_local & sm) << (8u - p));
            my_t = ((s_local & sm) << (8u - p)) | (tid >> p);
            s_local = (s_local & ~sm) | (tid & sm);
        }
        const uint32_t my_in = tile_swz(my_t);
        uint32_t tx = 0, ty = 0;
        const bool tile_ok = slot_pos(fr, slot, &tx, &ty);                 // wave-uniform: scalar arithmetic
        const uint32_t x = tx * RRT_TILE_W + (my_in & 31u), y = ty * RRT_TILE_H + (my_in >> 5);
        const bool active = tile_ok && x < fr.width && y < fr.height;
        const uint32_t sample = fr.s0 + s_local;
        V3 o = mk(0., 0., 0.), d = mk(0., 0., 1.);
        uint32_t cov = ~0u;
        // the tile's bin: a 64-bit scalar through the entity loop (parked in LDS and looked up per entity instead, to
        // spare the two SGPRs, it measured slower: cfg2 primary kernel 1.00 against 0.96 ms)
        uint64_t ebits = ~0ull;
        if (CULL != 0 && fr.tile_bins && tile_ok) ebits = wave_uniform64(fr.tile_bins[ty * fr.tiles_x + tx]);   // one word a chunk, into an SGPR pair
        if (active) {
            if (CULL != 0) cov = cover_bits(lds, x, y);       // requested before the ray is generated
            primary_ray(fr, lds, x, y, sample, &o, &d);
        }
        process_ray<true, CULL, GROUPS, PL>(sc, lds, wk, fr, seg, active, path, o, d, -1, 0u, d_ids, d_t, y * fr.width + x, sample, tag_base,
                                            false, nullptr, nullptr, cov, ebits, chunk * 256u);
    }
    // the segment's two counts, for the first bounce level
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long c = *seg_cnt;
        wk.seg_n[(size_t)kMaxSegs + blockIdx.x] = uint2_{(uint32_t)c, (uint32_t)(c >> 32)};     // parity 1 = level 1
        // the level's total, so that a launch of a level without rays leaves on one load (one atomic per BLOCK)
        const unsigned long long n = (c & 0xffffffffull) + (c >> 32);
        if (n) atomicAdd(&wk.q_count[1], n);
    }
}

// The sky-only slots of a batch (fr.order[list_n[0] .. list_n[1]): no entity can be met anywhere in the tile, four
// tiles of five on the UFO frame): every primary ray ends in the sky, so a path is its ray's texel -- sky_texel_fast,
// or the reference's own chain where that is in doubt -- and the 4-byte sample word.  A kernel of its own, because it
// needs nothing of what the wavefront kernels carry (no descent arrays in LDS, no entity records, no hit path's
// registers): a block stages 256 bytes of constants and runs at twice their occupancy, beside them on a second stream.
// Same chunk geometry and path numbers as k_wave_primary.
__global__ __launch_bounds__(256) void k_sky(DScene sc, DWork wk, DFrame fr, int32_t* d_ids, double* d_t) {
    __shared__ __align__(16) unsigned char cmem[kCDoubles * 8];
    const uint32_t n_heavy = wave_uniform(fr.list_n[0]), n_sky = wave_uniform(fr.list_n[1]) - n_heavy;
    const uint32_t n_chunks = n_sky * fr.ns;
    if (blockIdx.x >= n_chunks) return;
    StagedT<false> lds{nullptr, nullptr, nullptr, nullptr, nullptr};
    lds.consts = stage_consts(sc, fr, cmem);
    __syncthreads();
    const uint32_t n_tiles = fr.npx_local >> 8;
    // tile-major: neighbouring blocks take the samples of one tile, whose texels are neighbours in the cubemap
    for (uint32_t m = blockIdx.x; m < n_chunks; m += gridDim.x) {
        const uint32_t k = m / fr.ns;
        uint32_t s_local = m - k * fr.ns;
        const uint32_t slot = wave_uniform((uint32_t)fr.order[n_heavy + k]);
        const uint32_t path = (s_local * n_tiles + slot) * 256u + threadIdx.x;
        uint32_t my_t = threadIdx.x;
        if (fr.pack_log2) {
            const uint32_t p = fr.pack_log2, sm = (1u << p) - 1u;
            my_t = ((s_local & sm) << (8u - p)) | (threadIdx.x >> p);
            s_local = (s_local & ~sm) | (threadIdx.x & sm);
        }
        const uint32_t my_in = tile_swz(my_t);
        uint32_t tx = 0, ty = 0;
        const bool tile_ok = slot_pos(fr, slot, &tx, &ty);
        const uint32_t x = tx * RRT_TILE_W + (my_in & 31u), y = ty * RRT_TILE_H + (my_in >> 5);
        if (!(tile_ok && x < fr.width && y < fr.height)) continue;
        const uint32_t sample = fr.s0 + s_local;
        const RRT_GLOBAL uint8_t* tp;
        if (!sky_texel_fast(fr, lds, x, y, sample, &tp)) {      // the texel is in doubt: the reference's chain
            V3 o, d;
            primary_ray(fr, lds, x, y, sample, &o, &d);
            tp = sky_texel_ptr(sc, lds, d);
        }
        const uint32_t texel = (RRT_ABL & 1) ? 0u : ld_texel(tp);
        if (sample == 0u) {
            const size_t pixel = (size_t)y * fr.width + x;
            if (d_ids) { d_ids[2 * pixel] = -1; d_ids[2 * pixel + 1] = -1; }
            if (d_t) d_t[pixel] = 0.0;
        }
        at_off<uint32_t>(wk.samp32, path * 4u) = kSampTexel | texel;
    }
}

// Tile lists of a frame (DFrame::order / slot_tile / list_n), from the tile bins.  One block of 1024 threads: an
// ordered (stable) compaction by class, so that every rank computes the same lists from the same bins.
//   dealt = false: this rank's slots j (tile by the arithmetic map) in the order [heavy, sky-only, padding].
//   dealt = true (rrt_settings::tile_deal): ALL tiles of the image in the order [a mesh can be met, only spheres can,
//     sky only], each class in image order; position p of that sequence is slot p / world of rank p % world.  Every
//     rank gets its share of each class (to within one tile) instead of whatever lies under its lattice of positions
//     -- the ranks' frames then cost the same (the reference's own split, 12 contiguous pixel ranges, tracer.rs:117-158,
//     is the opposite extreme).  The rank's slots are in class order already: order = 0, 1, 2, ...
// all_heavy (cull mode 2 = verify): every valid slot counts as heavy, so that the wavefront kernels see every tile
// and check the bins.
RRT_DEV unsigned long long block_exclusive_scan(unsigned long long v, unsigned long long* wave_tot, unsigned long long* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, d, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d, 64);
        if (lane >= d) incl += ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();                       // wave_tot may still be read from the previous call
    if (lane == 63u) wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (uint32_t w = 0; w < n_waves; ++w) {
        const unsigned long long t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}
constexpr uint32_t kDealClasses = 16u;      // cost classes of the tiles in which an entity can be met (dealt mode)
__global__ __launch_bounds__(1024) void k_lists(DFrame fr, uint32_t dealt, uint32_t all_heavy, uint32_t scatter,
                                                int32_t* slot_tile, int32_t* order, uint32_t* list_n) {
    __shared__ unsigned long long wave_tot[16];
    __shared__ uint32_t px_sum;
    if (threadIdx.x == 0) px_sum = 0u;
    const uint32_t n_img = fr.tiles_x * fr.tiles_y, W = fr.tile_world, r = fr.tile_rank;
    const uint32_t n_items = dealt ? n_img : fr.n_slots;
    const uint32_t per = (n_items + blockDim.x - 1u) / blockDim.x;
    const uint32_t i0 = threadIdx.x * per, i1 = i0 + per < n_items ? i0 + per : n_items;
    // Classes, dearest first.  dealt: the tiles of the image by the cost estimate of k_bins in kDealClasses steps, then
    // (class kDealClasses) the tiles behind which there is only sky.  not dealt: this rank's slots -- 0 = an entity can
    // be met, 1 = sky only, 2 = padding slot.
    // (not dealt: ONE scan -- the padding slots, class 2, are the last slots of the rank by construction and keep
    // their places behind the valid ones)
    const uint32_t K = dealt ? kDealClasses + 1u : 2u, sky_class = dealt ? kDealClasses : 1u;
    const auto cls_of = [&](uint32_t i, uint32_t* pos) -> uint32_t {
        // dealt: item i is tile i * scatter mod n_img (scatter coprime to n_img, near 0.618 n_img): inside a class the
        // tiles then follow each other in a scattered order instead of image order -- in image order the rows of a
        // shape (the UFO is about 32 tiles wide) beat against the rounds of the deal, and the same ranks got the
        // sparse edge tiles every time (measured: 6.8 % between the ranks' bounce rays, 2.4 % with the positional map)
        uint32_t p = dealt ? (uint32_t)(((unsigned long long)i * scatter) % n_img) : i;
        if (!dealt) {                        // slot i of this rank by the arithmetic map (slot_pos without a table)
            const uint32_t tile = i * W + r;
            if (tile >= n_img) return 2u;
            const uint32_t ty = tile / fr.tiles_x;
            uint32_t tx = tile - ty * fr.tiles_x;
            if (fr.tile_skew) tx = (tx + fr.tile_skew * ty) % fr.tiles_x;
            p = ty * fr.tiles_x + tx;
        }
        *pos = p;
        const bool heavy = all_heavy || fr.tile_bins[p] != 0ull;
        if (!dealt) return heavy ? 0u : 1u;
        if (!heavy) return sky_class;
        const uint32_t cost = fr.tile_cost[p] & 127u;
        return (kDealClasses - 1u) - cost * kDealClasses / 128u;
    };
    const auto tile_px = [&](uint32_t p) -> uint32_t {
        const uint32_t ty = p / fr.tiles_x, tx = p - ty * fr.tiles_x;
        const uint32_t w = fr.width - tx * RRT_TILE_W < RRT_TILE_W ? fr.width - tx * RRT_TILE_W : RRT_TILE_W;
        const uint32_t h = fr.height - ty * RRT_TILE_H < RRT_TILE_H ? fr.height - ty * RRT_TILE_H : RRT_TILE_H;
        return w * h;
    };
    // Stable compaction by class, two classes a scan (32-bit counts in one 64-bit word).  Dealt: position p of the
    // sorted sequence belongs to round p / W, and the rounds are dealt to the ranks to and fro (round 0: ranks 0 .. W-1,
    // round 1: W-1 .. 0, ...), so that every rank takes one tile of every W neighbours in cost and the dearer and the
    // cheaper end of each round alternate: the ranks' sums differ by less than one tile's cost.  Slot = round.
    uint32_t run_base = 0u, heavy_total = 0u, my_px = 0u;
    for (uint32_t g0 = 0; g0 < K; g0 += 2u) {
        unsigned long long cnt = 0ull;
        for (uint32_t i = i0; i < i1; ++i) {
            uint32_t p;
            const uint32_t c = cls_of(i, &p);
            if (c == g0) cnt += 1ull;
            else if (c == g0 + 1u) cnt += 1ull << 32;
        }
        unsigned long long tot;
        const unsigned long long base = block_exclusive_scan(cnt, wave_tot, &tot);
        const uint32_t cb0 = run_base, cb1 = run_base + (uint32_t)tot;
        uint32_t off0 = (uint32_t)base, off1 = (uint32_t)(base >> 32);
        for (uint32_t i = i0; i < i1; ++i) {
            uint32_t p = 0;
            const uint32_t c = cls_of(i, &p);
            if (c != g0 && c != g0 + 1u) continue;
            const uint32_t pos = c == g0 ? cb0 + off0++ : cb1 + off1++;
            if (dealt) {
                const uint32_t round = pos / W, idx = pos - round * W;
                if (((round & 1u) ? W - 1u - idx : idx) == r) {
                    slot_tile[round] = (int32_t)p;
                    order[round] = (int32_t)round;
                    my_px += tile_px(p);
                }
            } else {
                order[pos] = (int32_t)i;
                if (c != 2u) my_px += tile_px(p);
            }
        }
        run_base = cb1 + (uint32_t)(tot >> 32);
        if (g0 < sky_class) heavy_total += (uint32_t)tot;
        if (g0 + 1u < sky_class) heavy_total += (uint32_t)(tot >> 32);
    }
    if (my_px) atomicAdd(&px_sum, my_px);
    // positions p < n that are this rank's (rounds dealt to and fro)
    const auto share = [&](uint32_t n) -> uint32_t {
        const uint32_t full = n / W, rem = n - full * W;
        const bool extra = rem != 0u && ((full & 1u) ? r >= W - rem : r < rem);
        return full + (extra ? 1u : 0u);
    };
    if (dealt) {
        // padding slots of this rank: the rounds it has no tile in
        for (uint32_t slot = share(n_img) + threadIdx.x; slot < fr.n_slots; slot += blockDim.x) {
            slot_tile[slot] = -1;
            order[slot] = (int32_t)slot;
        }
    } else {
        // padding slots (j * W + r >= n_img): the tail of the slot range, behind the heavy and the sky-only lists
        const uint32_t n_valid = n_img > r ? (n_img - r + W - 1u) / W : 0u;
        for (uint32_t slot = n_valid + threadIdx.x; slot < fr.n_slots; slot += blockDim.x) order[slot] = (int32_t)slot;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        list_n[0] = dealt ? share(heavy_total) : heavy_total;
        list_n[1] = dealt ? share(n_img) : (n_img > r ? (n_img - r + W - 1u) / W : 0u);      // valid slots (not dealt: j * W + r < n_img)
        list_n[2] = px_sum;
        list_n[3] = 0u;
    }
}

// Dealt mode, root side: every rank's buffer is its slots' pixels followed by its slot -> tile table.
__global__ __launch_bounds__(256) void k_assemble_dealt(const uint8_t* g, uint32_t w, uint32_t h, uint32_t world, uint32_t slots,
                                                        uint32_t tiles_x, uint8_t* out) {
    const uint32_t rs = blockIdx.x;                          // (rank, slot)
    if (rs >= world * slots) return;
    const uint32_t rank = rs / slots, slot = rs - rank * slots;
    const size_t stride = (size_t)slots * (RRT_TILE_W * RRT_TILE_H * 3 + 4);
    const uint8_t* base = g + (size_t)rank * stride;
    const int32_t tile = ((const int32_t*)(base + (size_t)slots * RRT_TILE_W * RRT_TILE_H * 3))[slot];
    if (tile < 0) return;
    const uint32_t ty = (uint32_t)tile / tiles_x, tx = (uint32_t)tile - ty * tiles_x;
    const uint32_t in = threadIdx.x, x = tx * RRT_TILE_W + (in & 31u), y = ty * RRT_TILE_H + (in >> 5);
    if (x >= w || y >= h) return;
    const uint8_t* src = base + ((size_t)slot * 256u + in) * 3;
    uint8_t* dst = out + ((size_t)y * w + x) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// A block of the bounce kernel whose segments hold at most this many rays (one unit a wave) runs in tail mode (see
// k_wave_bounce): it carries every path on to its end, and the deeper levels find its segments empty.
constexpr uint32_t kTailBlock = 256u;
constexpr uint32_t kSegPerBlock = 16u;      // segments a block of the bounce kernel takes at most (kMaxSegs / the smallest bounce grid)

// Bounce level `bounce`.  The queue comes in segments (see SegOut): block b takes the segments b, b + grid, ... one
// after the other; a segment is cut into units of 64 rays -- live units from its front, dead units from its back --
// which the block's four waves pull from a counter in LDS.  Rays of a bounce level differ in cost by
// more than an order of magnitude (a reflection that leaves the UFO at once vs. one that walks a
// dozen leaves inside it); with one fixed 64-ray slice per wave and round, a block kept its LDS and
// register slots until its slowest wave was done (measured: 2.4 of 4 waves/SIMD resident).
template <int CULL, bool GROUPS, bool SMALL, bool PL>
__global__ __launch_bounds__(256, RRT_BWAVES) void k_wave_bounce(RRT_SC_PARAM, DWork wk, DFrame fr, uint32_t bounce) {
    RRT_SC_BIND
    extern __shared__ __align__(16) unsigned char smem[];
    const SegLayout L = seg_layout(fr);
    const uint32_t n_seg = L.grid;
    const uint2_* seg_in = wk.seg_n + (size_t)(bounce & 1u) * kMaxSegs;
    uint2_* seg_out = wk.seg_n + (size_t)((bounce + 1u) & 1u) * kMaxSegs;
    // This block's segments sg_i = blockIdx.x + i * grid (at most kSegPerBlock of them: launch_bounce sizes the
    // grid) are treated as ONE region, their concatenation in the order i = 0, 1, ...: the live rays of all of them
    // form one sequence that is cut into units of 64 (then the dead rays likewise), whatever segment a ray sits in,
    // and the continuation rays go to the front (live) and the back (dead) of the concatenation of the same segments
    // in the other queue -- never more rays than came in, so they fit.  Per-segment units (the first version) left the
    // deeper levels with a handful of rays per unit (the 222 k second-level rays of the 64-spp frame in 8192 segments:
    // 27 a unit), and a barrier per segment left three waves idle at the end of every segment.
    RRT_LDS uint32_t* next_unit = (RRT_LDS uint32_t*)(smem + lds_layout_bytes(sc));
    RRT_LDS unsigned long long* s_cnt = (RRT_LDS unsigned long long*)(next_unit + 2);   // the block's append counter (low: live, high: dead)
    RRT_LDS uint32_t* s_live = next_unit + 4;                      // [kSegPerBlock] live rays of segment i
    RRT_LDS uint32_t* s_dead = s_live + kSegPerBlock;              // [kSegPerBlock] dead rays
    RRT_LDS uint32_t* s_lpre = s_dead + kSegPerBlock;              // [kSegPerBlock + 1] live rays of the segments before i
    RRT_LDS uint32_t* s_dpre = s_lpre + kSegPerBlock + 1;          // [kSegPerBlock + 1] dead rays ...
    RRT_LDS uint32_t* s_cpre = s_dpre + kSegPerBlock + 1;          // [kSegPerBlock + 1] entries (capacity) ...
    RRT_LDS uint32_t* s_start = s_cpre + kSegPerBlock + 1;         // [kSegPerBlock] first entry of segment i
    RRT_LDS uint32_t* s_cap = s_start + kSegPerBlock;              // [kSegPerBlock]
    // The blocks that work: a level with few rays is left to the first `grid` blocks of the launch (its ray count is
    // known from the level before; about fr.bounce_rays_per_block rays each, at least so many that none takes more than
    // kSegPerBlock segments, a multiple of 8 so that a segment stays on the XCD that filled it), the others leave on
    // this one load.  Every block pays the LDS staging (2.3 us of a wave slot): a rank's eighth of the 64-spp frame in
    // 3 x 2048 staged blocks held 1.67 x the wave-cycles of an eighth of the whole frame's bounce kernels.
    const unsigned long long level_rays = wave_uniform64(wk.q_count[bounce]);
    if (level_rays == 0ull) return;                // a level without rays (grid-uniform): nothing is read, nothing written
    uint32_t grid = gridDim.x;
    {
        // ... but never fewer than fill the chip once: 1024 blocks are resident at a time (4 a CU).  A thin level (it
        // runs in tail mode, below) lasts as long as its slowest wave whatever the block count -- 137 us for the
        // 55.7 k second-level rays of the 16-spp frame with 1024 as with 2048 blocks -- and fewer blocks stage less.
        // Measured and dropped: units of 16 or 32 rays for a thin level, so that all four waves of a block have work
        // (in tail mode every unit is carried to its paths' ends, one after the other: cfg1 0.097 -> 0.134 ms).
        unsigned long long want = level_rays / fr.bounce_rays_per_block + 1ull;
        if (want < 1024ull) want = 1024ull;
        const uint32_t need = (n_seg + kSegPerBlock - 1u) / kSegPerBlock;      // <= gridDim.x (launch_bounce)
        uint32_t g = want < (unsigned long long)grid ? (uint32_t)want : grid;
        if (g < need) g = need;
        g = (g + 7u) & ~7u;
        if (g < grid) grid = g;
    }
    if (blockIdx.x >= grid) return;
    const uint32_t my_nseg = wave_uniform(blockIdx.x < n_seg ? (n_seg - blockIdx.x + grid - 1u) / grid : 0u);
    if (threadIdx.x < kSegPerBlock) {
        uint2_ c{0u, 0u};
        if (threadIdx.x < my_nseg) c = seg_in[blockIdx.x + threadIdx.x * grid];
        s_live[threadIdx.x] = c.x;
        s_dead[threadIdx.x] = c.y;
    }
    __syncthreads();
    uint32_t mine = 0u, mine_dead = 0u;
    for (uint32_t i = 0; i < my_nseg; ++i) { mine += s_live[i] + s_dead[i]; mine_dead += s_dead[i]; }
    mine = wave_uniform(mine); mine_dead = wave_uniform(mine_dead);
    if (mine == 0u) {       // block-uniform: no rays here; the next level is told so, and the staging is skipped
        if (threadIdx.x < my_nseg) seg_out[blockIdx.x + threadIdx.x * grid] = uint2_{0u, 0u};
        return;
    }
    if (threadIdx.x < kSegPerBlock) {
        uint32_t st = 0u, cp = 0u;
        if (threadIdx.x < my_nseg) {
            const SegRange R = seg_range(L, fr.ns, blockIdx.x + threadIdx.x * grid);
            st = R.start; cp = R.cap;
        }
        s_start[threadIdx.x] = st;
        s_cap[threadIdx.x] = cp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (bounce == 1u) atomicAdd(&wk.stats[3], (unsigned long long)mine_dead);     // primary hits whose continuation ray is dead
        atomicAdd(&wk.stats[0], (unsigned long long)mine);
        if (bounce < 4u) atomicAdd(&wk.stats[4 + bounce], (unsigned long long)mine);
        uint32_t al = 0u, ad = 0u, ac = 0u;
        for (uint32_t i = 0; i < kSegPerBlock; ++i) {
            s_lpre[i] = al; s_dpre[i] = ad; s_cpre[i] = ac;
            al += s_live[i]; ad += s_dead[i]; ac += s_cap[i];
        }
        s_lpre[kSegPerBlock] = al; s_dpre[kSegPerBlock] = ad; s_cpre[kSegPerBlock] = ac;
        *next_unit = 0u;
        next_unit[kSegN] = my_nseg;
        *s_cnt = 0ull;
    }
    const StagedT<SMALL> lds = stage_scene<SMALL>(sc, fr, smem, true);      // ends with the barrier that also publishes the above
    const uint32_t n_live = wave_uniform(s_lpre[kSegPerBlock]), n_dead = wave_uniform(s_dpre[kSegPerBlock]);
    const uint32_t live_units = (n_live + 63u) >> 6, my_units = live_units + ((n_dead + 63u) >> 6);
    // a block with at most one unit a wave carries its paths to their ends (tail mode, below)
    // ... and so does every block of a thin level (fewer rays than 1024 blocks x 256): the level after it would be a
    // launch for the few blocks that hold more -- 28 us for 1.9 k rays on the 16-spp frame, 64 us for 7.8 k at 64 spp
    const bool tail = n_live + n_dead <= kTailBlock || level_rays < 1024ull * 256ull;
    static_assert(kSegPerBlock == 16u, "the queue table's layout (kSegCpre, kSegStart) is written for 16 segments a block");
    const SegOut seg{next_unit};       // [1] = my_nseg, [2..3] = s_cnt, [kSegCpre ..] = s_cpre, [kSegStart ..] = s_start
    double* q = wk.q[bounce & 1];
    // What the top of the unit loop needs is made anew in every trip from a scalar (the table's LDS offset) and from
    // the hardware (the lane number), behind volatile asm so that neither is hoisted: hoisted, both were VGPRs that
    // lived through process_ray, were spilled to scratch, and their reloads -- one in front of the unit grab, one
    // behind it, each a memory round trip -- stood at the head of every unit.
    const uint32_t tab_off = wave_uniform(lds_layout_bytes(sc));
    const uint32_t wave_in_block = wave_uniform(threadIdx.x >> 6);      // so that threadIdx.x need not live through the loop
    const unsigned long long tk0 = TSTAMP();
    unsigned long long acc_q = 0, acc_p = 0, acc_n = 0, acc_dp = 0, acc_dn = 0, acc_lc = 0;
    (void)tk0; (void)acc_q; (void)acc_p; (void)acc_n; (void)acc_dp; (void)acc_dn; (void)acc_lc;
    {
      for (;;) {
        const unsigned long long t0 = TSTAMP();
        (void)t0;
        uint32_t lane, off_s = tab_off;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        asm volatile("" : "+s"(off_s));
        RRT_LDS uint32_t* unit_ctr = (RRT_LDS uint32_t*)(smem + off_s);
        uint32_t u = 0u;
        if (lane == 0u) u = __hip_atomic_fetch_add(unit_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        u = (uint32_t)__builtin_amdgcn_readfirstlane((int)u);
        if (u >= my_units) break;               // wave-uniform; the counter only grows, so every wave gets here
        // unit -> rays: live units count up through the live runs of the segments (each at its segment's front), dead
        // units through the dead runs (each at its segment's back, downwards)
        const bool live_unit = u < live_units;
        const uint32_t g = (live_unit ? u : u - live_units) * 64u + lane;
        const bool active = g < (live_unit ? n_live : n_dead);
        uint32_t entry = 0u;
        if (active) {
            const RRT_LDS uint32_t* pre = live_unit ? s_lpre : s_dpre;
            uint32_t i = 0u;
            while (i + 1u < my_nseg && g >= pre[i + 1u]) ++i;
            const uint32_t off = g - pre[i];
            entry = live_unit ? s_start[i] + off : s_start[i] + s_cap[i] - 1u - off;
        }
        V3 o = mk(0., 0., 0.), d = mk(0., 0., 1.);
        uint32_t path = 0;
        bool dead_sky = false;
        int32_t medium = -1;
        if (active) {
            const f64x2 a = at_rec<f64x2>(q, entry, 0u), b = at_rec<f64x2>(q, entry, 16u), c = at_rec<f64x2>(q, entry, 32u);
            const uint2 pm = at_rec<uint2>(q, entry, 48u);
            o = mk(a.x, a.y, b.x);
            d = mk(b.y, c.x, c.y);
            path = pm.x & ~kDeadSky;
            dead_sky = (pm.x & kDeadSky) != 0u;
            medium = (int32_t)pm.y;
        }
        const unsigned long long t1 = TSTAMP();
        unsigned long long tch = 0;
        // Carry or queue.  A cast's continuation ray normally goes to the queue of the next level, where rays are
        // compacted into full waves again.  That is worth a launch only for many rays: a thin level (`tail`: fewer
        // rays than the chip has lanes) carries every path on to its end in its own lane -- a launch whose rays do not
        // fill the chip lasts as long as its slowest ray, 25-90 us each for the last levels of a frame.  All lanes of a
        // wave sit at the same depth, so the level stays wave-uniform; the rays of the carried levels are counted
        // where they are cast.  Two separate call sites on purpose: routed through the carrying loop, the thick levels
        // spilled 17 VGPRs (72 B of scratch per lane, +330 MB of HBM writes per cfg2 frame) and lost 15 %.  Also
        // measured and dropped: letting a wave of a THICK level carry on when at most 16 of its rays continue
        // (cfg2 bounce levels 0.375 -> 0.478 ms: the nearly empty waves hold their slots for two more casts).
        if (tail) {
            bool act = active;
            for (uint32_t b = bounce;;) {
                NextRay nx;
                process_ray<false, CULL, GROUPS, PL>(sc, lds, wk, fr, seg, act, path, o, d, medium, b, nullptr, nullptr, 0u, 0u, 0u, dead_sky, &tch, &nx);
                act = act && nx.cont;
                const unsigned long long going = __ballot(act);
                if (going == 0ull) break;
                ++b;
                if (lane == (uint32_t)(__ffsll((long long)going) - 1)) {
                    const unsigned long long n = (unsigned long long)__popcll(going);
                    atomicAdd(&wk.stats[0], n);
                    if (b < 4u) atomicAdd(&wk.stats[4 + b], n);
                }
                o = nx.o; d = nx.d; medium = nx.medium; dead_sky = nx.zero_mix;
            }
        } else {
            process_ray<false, CULL, GROUPS, PL>(sc, lds, wk, fr, seg, active, path, o, d, medium, bounce, nullptr, nullptr, 0u, 0u, 0u, dead_sky, &tch);
        }
        const unsigned long long t2 = TSTAMP();
        if (__ballot(active && !dead_sky) != 0ull) acc_lc += tch;
        acc_q += t1 - t0; acc_p += t2 - t1; acc_n += 1;
        if (__ballot(active && !dead_sky) == 0ull) { acc_dp += t2 - t1; acc_dn += 1; }
      }
    }
    // The region is done when all four waves are.  Its continuation rays fill the concatenation from the front (live)
    // and from the back (dead): what that leaves in segment i -- a run at its front, a run at its back -- are its two
    // counts for the next level.  (Measured and dropped: no barrier here, the wave that finishes last -- a counter in
    // LDS tells -- does this and the others leave at once: 12 % fewer wave-cycles in a rank's bounce kernels, no
    // change in frame time.)
    if (bounce == 1u) { TADD(0, acc_q); TADD(1, acc_p); TADD(2, acc_n); TADD(3, TSTAMP() - tk0); TADD(4, 1); TADD(5, acc_dp); TADD(6, acc_dn); TADD(7, acc_lc); }
    __syncthreads();
    uint32_t tid;          // wave 0's lanes = threads 0 .. 63 (kSegPerBlock <= 64)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid));
    if (wave_in_block == 0u && tid < my_nseg) {
        const unsigned long long c = *s_cnt;
        const uint32_t tl = (uint32_t)c, td = (uint32_t)(c >> 32), total = s_cpre[kSegPerBlock];
        const uint32_t c0 = s_cpre[tid], c1 = c0 + s_cap[tid];
        const uint32_t live_i = tl > c0 ? (tl < c1 ? tl - c0 : c1 - c0) : 0u;
        const uint32_t d0 = total - td;                                    // dead entries: [d0, total) of the concatenation
        const uint32_t lo = d0 > c0 ? d0 : c0;
        const uint32_t dead_i = c1 > lo ? c1 - lo : 0u;
        seg_out[blockIdx.x + tid * grid] = uint2_{live_i, dead_i};
        if (tid == 0u && tl + td != 0u) atomicAdd(&wk.q_count[bounce + 1u], (unsigned long long)tl + td);
    }
}

// mean = sum / spp ; vec_rgb (common.rs:75-81); row-major frame or this rank's compact tile buffer
RRT_DEV void resolve_pixel(const DFrame& fr, uint32_t lp, V3 sum, uint8_t* d_rgb8, double* d_color) {
    uint32_t x = 0, y = 0;
    const bool valid = local_pixel(fr, lp, &x, &y);
    const double n = (double)fr.spp;
    V3 mean = mk(0., 0., 0.);
    if (valid) mean = mk(sum.x / n, sum.y / n, sum.z / n);
    const uint32_t r = sat_u8(mean.x * 255.0), g = sat_u8(mean.y * 255.0), b = sat_u8(mean.z * 255.0);
    if (fr.tile_world == 1u) {
        if (!valid) return;
        const size_t px = (size_t)y * fr.width + x;
        if (d_rgb8) { d_rgb8[3 * px] = (uint8_t)r; d_rgb8[3 * px + 1] = (uint8_t)g; d_rgb8[3 * px + 2] = (uint8_t)b; }
        if (d_color) { d_color[3 * px] = mean.x; d_color[3 * px + 1] = mean.y; d_color[3 * px + 2] = mean.z; }
    } else if (d_rgb8) {
        d_rgb8[3 * (size_t)lp] = (uint8_t)r; d_rgb8[3 * (size_t)lp + 1] = (uint8_t)g; d_rgb8[3 * (size_t)lp + 2] = (uint8_t)b;
    }
}

// One trip of k_accumulate: samples s0 .. s0 + TRIP - 1 of pixel pp.  Their tags, then the colours the tags point to,
// are requested together (the kernel is a latency chain tag -> colour per sample otherwise); the sum stays in sample
// order.  PACKED (sample packing with at least TRIP samples a wave): the path numbers of these samples are
// consecutive (path_of) and ns is a multiple of TRIP, so a lane reads TRIP * 4 B of tags and up to TRIP * 24 B of
// colours in one piece -- with TRIP = 16 whole cache lines that no other lane shares, each fetched once.
template <uint32_t TRIP, bool PACKED>
RRT_DEV void accumulate_trip(const DWork& wk, const DFrame& fr, const double* by255, uint32_t s0, uint32_t pp, V3* acc) {
    uint32_t tag[TRIP];
    V3 c[TRIP];
    const uint32_t p0 = path_of(fr, s0, pp);
    if (PACKED) {
#pragma unroll
        for (uint32_t j = 0; j < TRIP; j += 4u) {
            const uint4 tg = *(const uint4*)(wk.samp32 + p0 + j);
            tag[j] = tg.x; tag[j + 1] = tg.y; tag[j + 2] = tg.z; tag[j + 3] = tg.w;
        }
    } else {
#pragma unroll
        for (uint32_t j = 0; j < TRIP; ++j) {
            const uint32_t s = s0 + j < fr.ns ? s0 + j : fr.ns - 1u;     // clamped: re-reads the last sample, not added
            tag[j] = wk.samp32[path_of(fr, s, pp)];
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < TRIP; ++j) {
        const uint32_t s = s0 + j < fr.ns ? s0 + j : fr.ns - 1u;
        const size_t p = 3 * (size_t)(PACKED ? p0 + j : path_of(fr, s, pp));
        c[j] = mk(0., 0., 0.);
        if (!(tag[j] & kSampTexel)) c[j] = mk(wk.samp[p], wk.samp[p + 1], wk.samp[p + 2]);
    }
#pragma unroll
    for (uint32_t j = 0; j < TRIP; ++j) {
        if (!PACKED && s0 + j >= fr.ns) break;
        const uint32_t t = tag[j];
        const V3 v = (t & kSampTexel) ? mk(by255[t & 255u], by255[(t >> 8) & 255u], by255[(t >> 16) & 255u]) : c[j];
        *acc = (fr.s0 + s0 + j == 0u) ? v : *acc + v;
    }
}

// accum[lp] (+)= sum over this batch's samples, in sample order (deterministic: the oracle's loop).
// FINAL (the frame's last batch): the sum goes straight into the pixel instead of back to accum.
template <bool FINAL>
__global__ __launch_bounds__(256) void k_accumulate(DWork wk, DFrame fr, uint8_t* d_rgb8, double* d_color, uint32_t block0) {
    __shared__ double by255[256];
    by255[threadIdx.x] = (double)threadIdx.x / 255.0;          // cubemap.rs:70-74, one IEEE division per thread
    __syncthreads();
    const uint32_t lp = (blockIdx.x + block0) * 256u + threadIdx.x;   // grid = npx_local / 256 exactly, or a band of tile rows (block0 = its first tile)
    const uint32_t N = fr.npx_local;
    const uint32_t pp = (lp & ~255u) | tile_unswz(lp & 255u);   // where the trace kernels put this pixel's samples
    V3 acc = mk(0., 0., 0.);
    if (fr.s0 != 0u) acc = mk(wk.accum[lp], wk.accum[lp + N], wk.accum[lp + 2 * (size_t)N]);
    if (fr.pack_log2 >= 4u) {
        for (uint32_t s0 = 0; s0 < fr.ns; s0 += 16u) accumulate_trip<16, true>(wk, fr, by255, s0, pp, &acc);
    } else if (fr.pack_log2 >= 2u) {
        for (uint32_t s0 = 0; s0 < fr.ns; s0 += 4u) accumulate_trip<4, true>(wk, fr, by255, s0, pp, &acc);
    } else {
        for (uint32_t s0 = 0; s0 < fr.ns; s0 += 4u) accumulate_trip<4, false>(wk, fr, by255, s0, pp, &acc);
    }
    if (FINAL) {
        resolve_pixel(fr, lp, acc, d_rgb8, d_color);
    } else {
        wk.accum[lp] = acc.x; wk.accum[lp + N] = acc.y; wk.accum[lp + 2 * (size_t)N] = acc.z;
    }
}

// k_accumulate for batches packed with at least 16 samples a wave (pack_log2 >= 4, so ns is a multiple of 16): a QUAD
// of lanes per pixel.  In a trip of 16 samples lane j of the quad reads samples 4j .. 4j + 3 -- its four tags are one
// 16-byte load, the quad's 64 bytes and (for 16-sample groups) the wave's kilobyte are contiguous -- and the running
// sum goes round the quad in sample order: lane 0 adds its four, hands the sum to lane 1 (a DPP quad broadcast, no
// LDS), and so on; the order of the additions is that of the one-lane loop, so is every bit of the result.  Four times
// the threads of the one-lane kernel, which is what a rank's thin frame needs (one wave per SIMD there otherwise).
RRT_DEV double quad_bcast(double v, int j) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int l2, h2;
    switch (j) {     // dpp_ctrl = quad_perm [j, j, j, j]
    case 0: l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x00, 0xf, 0xf, false); h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x00, 0xf, 0xf, false); break;
    case 1: l2 = __builtin_amdgcn_update_dpp(lo, lo, 0x55, 0xf, 0xf, false); h2 = __builtin_amdgcn_update_dpp(hi, hi, 0x55, 0xf, 0xf, false); break;
    case 2: l2 = __builtin_amdgcn_update_dpp(lo, lo, 0xaa, 0xf, 0xf, false); h2 = __builtin_amdgcn_update_dpp(hi, hi, 0xaa, 0xf, 0xf, false); break;
    default: l2 = __builtin_amdgcn_update_dpp(lo, lo, 0xff, 0xf, 0xf, false); h2 = __builtin_amdgcn_update_dpp(hi, hi, 0xff, 0xf, 0xf, false); break;
    }
    return __hiloint2double(h2, l2);
}

template <bool FINAL>
__global__ __launch_bounds__(256) void k_accumulate_quad(DWork wk, DFrame fr, uint8_t* d_rgb8, double* d_color, uint32_t block0) {
    __shared__ double by255[256];
    by255[threadIdx.x] = (double)threadIdx.x / 255.0;          // cubemap.rs:70-74, one IEEE division per thread
    __syncthreads();
    const uint32_t gid = (blockIdx.x + block0) * 256u + threadIdx.x;  // grid = npx_local / 64 exactly, or a band (block0 = 4 x its first tile)
    const uint32_t lp = gid >> 2, sub = gid & 3u;
    const uint32_t N = fr.npx_local;
    const uint32_t pp = (lp & ~255u) | tile_unswz(lp & 255u);
    V3 acc = mk(0., 0., 0.);
    if (fr.s0 != 0u) acc = mk(wk.accum[lp], wk.accum[lp + N], wk.accum[lp + 2 * (size_t)N]);
    // one trip = samples s0 .. s0 + 15 of the pixel (consecutive paths); tg = this lane's four tags of them
    const auto trip = [&](uint32_t s0, uint32_t p0, uint4 tg) {
        const uint32_t tag[4] = {tg.x, tg.y, tg.z, tg.w};
        V3 v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const size_t p = 3 * (size_t)(p0 + j);
            v[j] = mk(0., 0., 0.);
            if (!(tag[j] & kSampTexel)) v[j] = mk(wk.samp[p], wk.samp[p + 1], wk.samp[p + 2]);
        }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t t = tag[j];
            if (t & kSampTexel) v[j] = mk(by255[t & 255u], by255[(t >> 8) & 255u], by255[(t >> 16) & 255u]);
        }
        const bool first = fr.s0 + s0 == 0u;                   // the frame's first sample starts the sum (no 0 + v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            V3 a = acc;
            a = (first && q == 0) ? v[0] : a + v[0];
            a = a + v[1]; a = a + v[2]; a = a + v[3];
            // every lane computed "the quad's sum so far + its own four"; lane q's is the one that is in order
            acc = mk(quad_bcast(a.x, q), quad_bcast(a.y, q), quad_bcast(a.z, q));
        }
    };
    // The tags of four trips are requested together: a trip is a chain tag -> colour -> sixteen additions, and one
    // after the other the four trips of a 64-sample batch were eight dependent memory round trips per pixel; a pixel
    // behind which there is only sky (three of four) has no colours to fetch and is left with one.
    uint32_t s0 = 0u;
    for (; s0 + 64u <= fr.ns; s0 += 64u) {
        uint32_t p0[4];
        uint4 tg[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            p0[k] = path_of(fr, s0 + 16u * k, pp) + 4u * sub;
            tg[k] = *(const uint4*)(wk.samp32 + p0[k]);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) trip(s0 + 16u * k, p0[k], tg[k]);
    }
    for (; s0 < fr.ns; s0 += 16u) {
        const uint32_t p0 = path_of(fr, s0, pp) + 4u * sub;
        trip(s0, p0, *(const uint4*)(wk.samp32 + p0));
    }
    if (sub != 0u) return;
    if (FINAL) {
        resolve_pixel(fr, lp, acc, d_rgb8, d_color);
    } else {
        wk.accum[lp] = acc.x; wk.accum[lp + N] = acc.y; wk.accum[lp + 2 * (size_t)N] = acc.z;
    }
}

// Coverage mask of one mesh for one frame (see kMaxCover in rrt_device.h).  Thread = stream entry (a triangle; the
// entries a triangle is duplicated into just set the same bits again).
//   Claim used: if Triangle::collide (triangle.rs:25-58, f64) ACCEPTS a ray (o, d), the line through o along d passes
//   within `delta` of the triangle at a parameter >= -dt -- the bound behind the conservative cull (make_cull_ray:
//   delta is its `pad` without the f32 term, dt its -tmin), doubled here.  Primary rays start at the eye, so with the
//   triangle's dilated hull entirely in front of the eye plane (z - delta > 2 dt: otherwise the mask's "all" word is
//   set) that point of the line projects through the eye onto the ray's own lens point, i.e. into the ray's pixel
//   [x, x+1) x [y, y+1) (tracer.rs:126-131 + jitter in [0,1)).  A perspective image of a triangle in front of the eye
//   is the triangle of its projected corners; dilating by delta moves a projection u = x/z by at most
//   delta (1 + |u|) / (z - delta).  The mask takes the bounding rectangle of all that plus one whole pixel each side
//   (thirteen orders of magnitude above the rounding of the reference's own ray generation).
// Non-finite intermediate values set the "all" word as well.  Verified in cull mode 2 (a masked ray that the walk
// nevertheless answers counts as a cull violation).
struct CoverParams {
    double eye[3], lens_z, lens_w, lens_h;
    double delta, z_near;
    uint32_t width, height, stride, bytes;
};
// cells [c0, c1] of row y, as bytes (a cell = kCoverCell pixels of a row).  Plain stores of the value 1: every writer
// writes the same thing, so nothing is read, nothing is atomic and nothing waits.  (The first version kept a bit per
// pixel and OR-ed it in: device-scope atomics are performed at the memory side of the eight XCDs' L2s, one after the
// other per address -- 90 us for the UFO's 9 825 entries, 2.6 ms for the 2 M-triangle field.)
// A run is written as (possibly overlapping, possibly unaligned) windows of 8, 4, 2 or 1 bytes that stay inside it:
// at most two stores up to 8 cells, one per 8 cells beyond -- a wave's scattered stores are what this kernel's time is.
RRT_DEV void cover_row(uint8_t* mask, const CoverParams& p, int32_t y, int32_t c0, int32_t c1) {
    typedef uint64_t u64u __attribute__((aligned(1)));
    typedef uint32_t u32u __attribute__((aligned(1)));
    typedef uint16_t u16u __attribute__((aligned(1)));
    uint8_t* row = mask + (size_t)y * p.stride;
    const int32_t len = c1 - c0 + 1;
    if (len >= 8) {
        for (int32_t c = c0; c + 7 < c1; c += 8) *(u64u*)(row + c) = 0x0101010101010101ull;
        *(u64u*)(row + c1 - 7) = 0x0101010101010101ull;
    } else if (len >= 4) {
        *(u32u*)(row + c0) = 0x01010101u; *(u32u*)(row + c1 - 3) = 0x01010101u;
    } else if (len >= 2) {
        *(u16u*)(row + c0) = 0x0101u; *(u16u*)(row + c1 - 1) = 0x0101u;
    } else if (len == 1) {
        row[c0] = 1;
    }
}
// Several lanes per triangle, each taking every n-th row of the rectangle: with one lane per triangle the
// kernel lasted as long as its slowest lane's few hundred scattered byte stores (186 us on the UFO: 154 waves).
// (2^lanes_log2 lanes: 8 for meshes like the UFO, 1 from a million entries on, where triangles are a few pixels and the
// projection itself is the work: 0.20 against 0.72 ms on the 2 M-triangle field.)
__global__ __launch_bounds__(256) void k_cover(const DTri* tris, int32_t n, CoverParams p, uint8_t* mask, uint32_t lanes_log2) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, kCoverLanes = 1u << lanes_log2;
    const int32_t j = (int32_t)(gid >> lanes_log2), sub = (int32_t)(gid & (kCoverLanes - 1u));
    bool big = false;
    int32_t x0 = 0, x1 = -1, y0 = 0, y1 = -1;
    if (j < n) {
        const RRT_GLOBAL DTri* T = as_global(tris) + j;
        const V3 eye = mk(p.eye[0], p.eye[1], p.eye[2]);
        const V3 v0 = ld3g(T->v0) - eye, v1 = (ld3g(T->v0) + ld3g(T->e1)) - eye, v2 = (ld3g(T->v0) + ld3g(T->e2)) - eye;
        const double zmin = fmin(v0.z, fmin(v1.z, v2.z)) - p.delta;
        bool all = !(zmin > p.z_near);                      // also catches NaN
        if (!all) {
            const double u0 = v0.x / v0.z, u1 = v1.x / v1.z, u2 = v2.x / v2.z;
            const double w0 = v0.y / v0.z, w1 = v1.y / v1.z, w2 = v2.y / v2.z;
            const double ulo = fmin(u0, fmin(u1, u2)), uhi = fmax(u0, fmax(u1, u2));
            const double wlo = fmin(w0, fmin(w1, w2)), whi = fmax(w0, fmax(w1, w2));
            const double g = 2.0 * p.delta / zmin;
            const double mu = g * (1.0 + fmax(fabs(ulo), fabs(uhi))), mw = g * (1.0 + fmax(fabs(wlo), fabs(whi)));
            // lens point of direction (u, w, 1): fx = (u lens_z + lens_w / 2) / lens_w, pixel coordinate fx * width
            const double sx = p.lens_z / p.lens_w * (double)p.width, sy = p.lens_z / p.lens_h * (double)p.height;
            const double pxlo = (ulo - mu) * sx + 0.5 * (double)p.width - 1.0, pxhi = (uhi + mu) * sx + 0.5 * (double)p.width + 1.0;
            const double pylo = (wlo - mw) * sy + 0.5 * (double)p.height - 1.0, pyhi = (whi + mw) * sy + 0.5 * (double)p.height + 1.0;
            if (!(fabs(pxlo) <= 1e300 && fabs(pxhi) <= 1e300 && fabs(pylo) <= 1e300 && fabs(pyhi) <= 1e300)) {
                all = true;
            } else if (pxhi >= 0.0 && pyhi >= 0.0 && pxlo < (double)p.width && pylo < (double)p.height) {
                x0 = (int32_t)floor(fmax(pxlo, 0.0)); x1 = (int32_t)floor(fmin(pxhi, (double)p.width - 1.0));
                y0 = (int32_t)floor(fmax(pylo, 0.0)); y1 = (int32_t)floor(fmin(pyhi, (double)p.height - 1.0));
            }
        }
        if (all) mask[p.bytes - 8u] = 1;
        x0 /= kCoverCell; x1 = x1 >= 0 ? x1 / kCoverCell : -1;          // pixels -> cells
        big = x1 >= x0 && y1 >= y0 && (int64_t)(x1 - x0 + 1) * (y1 - y0 + 1) > (2048 << lanes_log2);
        if (!big)
            for (int32_t y = y0 + sub; y <= y1; y += (int32_t)kCoverLanes) cover_row(mask, p, y, x0, x1);
        big = big && sub == 0;
    }
    // rectangles of more than 2048 cells per lane (a triangle that fills a good part of the image): the wave fills them
    // together, one after the other -- lane l takes rows l / 8 (mod 8) and the 8-cell pieces l % 8 (mod 8) of a row
    // (measured: with every rectangle above 16 words on this path the UFO's masks took 27 us instead of 9)
    for (unsigned long long todo = __ballot(big); todo != 0ull; todo &= todo - 1ull) {
        const int src = __ffsll((long long)todo) - 1;
        const int32_t bx0 = __shfl(x0, src, 64), bx1 = __shfl(x1, src, 64), by0 = __shfl(y0, src, 64), by1 = __shfl(y1, src, 64);
        const int32_t l = (int32_t)(threadIdx.x & 63u);
        for (int32_t y = by0 + (l >> 3); y <= by1; y += 8)
            for (int32_t c = bx0 + 8 * (l & 7); c <= bx1; c += 64) cover_row(mask, p, y, c, c + 7 > bx1 ? bx1 : c + 7);
    }
}

// Tile bins (DFrame::tile_bins).  Thread = tile.  Sphere::collide (sphere.rs:28-47) accepts a ray when tca >= 0 and
// d2 = |l|^2 - tca^2 <= r^2; its rounding moves d2 by less than 1e-15 |l|^2, so an accepted ray's line passes within
// r' = r + 1e-12 (|l|^2 / r + r) of the centre (three orders of magnitude of slack), at a point in front of the eye.
// The rays (u, v, 1) that touch a sphere of radius r' about (cx, cy, cz), cz - r' > 0, have
// u in (cx cz -+ r' sqrt(cx^2 + cz^2 - r'^2)) / (cz^2 - r'^2) (tangent planes through the eye and the y direction) and
// v likewise; the bin takes that rectangle plus one pixel each side.  Anything doubtful sets the bit.
struct BinParams {
    double eye[3], lens_z, lens_w, lens_h, z_slack;
    uint32_t width, height, tiles_x, tiles_y;
};
__global__ __launch_bounds__(256) void k_bins(DScene sc, BinParams p, const uint8_t* cover, uint32_t n_cover, uint32_t stride,
                                              uint32_t bytes, uint64_t* bins, uint8_t* cost_out) {
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    if (tile >= p.tiles_x * p.tiles_y) return;
    const uint32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    const double x0 = (double)(tx * RRT_TILE_W), x1 = fmin(x0 + (double)RRT_TILE_W, (double)p.width);
    const double y0 = (double)(ty * RRT_TILE_H), y1 = fmin(y0 + (double)RRT_TILE_H, (double)p.height);
    const double sx = p.lens_z / p.lens_w * (double)p.width, sy = p.lens_z / p.lens_h * (double)p.height;
    uint64_t bits = ~0ull;
    // an estimate of what the tile's primary rays will cost (the dealt tile map sorts by it): cells of 4 pixels of the
    // tile that a mesh's coverage mask keeps, + cells inside the bounding rectangle of a sphere's image; 64 for an
    // entity nothing is known about
    uint32_t cost = 0u;
    for (int32_t e = 0; e < sc.n_ents && e < 64; ++e) {
        const DEntity ent = sc.ents[e];
        const int32_t kind = ent.kind & ~kEntityMember;
        bool may = true;
        uint32_t ecost = kind == 2 ? 0u : 64u;
        if (kind == 0) {
            const DSphere S = sc.spheres[ent.index];
            const double cx = S.c[0] - p.eye[0], cy = S.c[1] - p.eye[1], cz = S.c[2] - p.eye[2];
            const double r = sqrt(S.r2), l2 = cx * cx + cy * cy + cz * cz;
            const double rr = r + 1e-12 * (l2 / r + r) + p.z_slack;
            if (r > 0.0 && cz - rr > p.z_slack && fabs(l2) <= 1e300) {
                const double den = cz * cz - rr * rr;
                const double su = rr * sqrt(fmax(cx * cx + den, 0.0)), sv = rr * sqrt(fmax(cy * cy + den, 0.0));
                const double ulo = (cx * cz - su) / den, uhi = (cx * cz + su) / den;
                const double vlo = (cy * cz - sv) / den, vhi = (cy * cz + sv) / den;
                const double pxlo = ulo * sx + 0.5 * (double)p.width - 1.0, pxhi = uhi * sx + 0.5 * (double)p.width + 1.0;
                const double pylo = vlo * sy + 0.5 * (double)p.height - 1.0, pyhi = vhi * sy + 0.5 * (double)p.height + 1.0;
                // written so that a NaN anywhere keeps the bit
                if (pxhi < x0 || pxlo >= x1 || pyhi < y0 || pylo >= y1) may = false;
                else if (pxhi >= pxlo && pyhi >= pylo) {
                    const double ox = fmin(pxhi, x1) - fmax(pxlo, x0), oy = fmin(pyhi, y1) - fmax(pylo, y0);
                    ecost = (uint32_t)fmin(64.0, fmax(1.0, ceil(ox / (double)kCoverCell) * ceil(oy)));
                }
            }
        } else if (kind == 3) {
            // a bare Triangle: k_cover's projection (see there) with the triangle's own constants -- S its longer edge,
            // R >= the distance from the eye to any of its points
            const DTriEnt T = sc.tri_ents[ent.index];
            const V3 eye = mk(p.eye[0], p.eye[1], p.eye[2]);
            const V3 v0 = ld3(T.v0) - eye, v1 = (ld3(T.v0) + ld3(T.e1)) - eye, v2 = (ld3(T.v0) + ld3(T.e2)) - eye;
            const double S = sqrt(fmax(mag2(ld3(T.e1)), mag2(ld3(T.e2)))) * (1.0 + 1e-9);
            const double R = fabs(v0.x) + fabs(v0.y) + fabs(v0.z) + S;
            const double amax = fmax(fmax(fabs(T.v0[0]), fabs(T.v0[1])), fabs(T.v0[2])) + S;
            const double omax = fmax(fmax(fabs(p.eye[0]), fabs(p.eye[1])), fabs(p.eye[2]));
            const double delta = 2.0 * (2.5e-7 * S * S * 1.7320508075688774 * (R + S)) * 1.000001 + 1e-9 * (omax + amax) + 1e-300;
            const double z_near = 4.0 * (8e-8 * S * S * (R + S)) * 1.000001 + 1e-9 * (omax + amax) + 1e-300;
            const double zmin = fmin(v0.z, fmin(v1.z, v2.z)) - delta;
            if (zmin > z_near) {                                 // (false for NaN: the bit stays)
                const double u0 = v0.x / v0.z, u1 = v1.x / v1.z, u2 = v2.x / v2.z;
                const double w0 = v0.y / v0.z, w1 = v1.y / v1.z, w2 = v2.y / v2.z;
                const double ulo = fmin(u0, fmin(u1, u2)), uhi = fmax(u0, fmax(u1, u2));
                const double wlo = fmin(w0, fmin(w1, w2)), whi = fmax(w0, fmax(w1, w2));
                const double g = 2.0 * delta / zmin;
                const double mu = g * (1.0 + fmax(fabs(ulo), fabs(uhi))), mw = g * (1.0 + fmax(fabs(wlo), fabs(whi)));
                const double pxlo = (ulo - mu) * sx + 0.5 * (double)p.width - 1.0, pxhi = (uhi + mu) * sx + 0.5 * (double)p.width + 1.0;
                const double pylo = (wlo - mw) * sy + 0.5 * (double)p.height - 1.0, pyhi = (whi + mw) * sy + 0.5 * (double)p.height + 1.0;
                if (pxhi < x0 || pxlo >= x1 || pyhi < y0 || pylo >= y1) may = false;      // a NaN anywhere keeps the bit
            }
        } else if (kind == 1) {
            const int32_t cs = (sc.meshes[ent.index].flags >> kCoverShift) & 15;
            if (cs != 0 && (uint32_t)cs <= n_cover) {
                // a tile row = 32 pixels = 8 cells = one aligned 8-byte word of the mask
                const uint8_t* m = cover + (size_t)(cs - 1) * bytes;
                uint64_t any = *(const uint64_t*)(m + bytes - 8u);
                uint32_t cells = any ? 64u : 0u;
                for (uint32_t y = ty * RRT_TILE_H; y < (ty + 1u) * RRT_TILE_H && y < p.height; ++y) {
                    const uint64_t wd = *(const uint64_t*)(m + (size_t)y * stride + 8u * tx);
                    any |= wd;
                    cells += (uint32_t)__popcll(wd & 0x0101010101010101ull);     // a set cell is a byte of value 1
                }
                may = any != 0ull;
                ecost = cells < 64u ? cells : 64u;
            }
        }
        if (!may) bits &= ~(1ull << e);
        else cost += ecost;
    }
    // a Scene<T> answers only through a member (scene.rs:50-52 = the kd walk over its members): no member that can be
    // met, no group.  (Members of any group count: with several groups that only keeps a bit too many.)
    uint64_t members = 0ull, groups = 0ull;
    for (int32_t e = 0; e < sc.n_ents && e < 64; ++e) {
        const int32_t k = sc.ents[e].kind;
        if ((k & ~kEntityMember) == 2) groups |= 1ull << e;          // top-level scenes and scenes that are members of one
        else if (k & kEntityMember) members |= 1ull << e;
    }
    if ((bits & members) == 0ull && sc.n_ents <= 64) bits &= ~groups;
    if (sc.n_ents < 64) bits &= (1ull << sc.n_ents) - 1ull;      // no entity, no bit: 0 = "only sky behind this tile"
    bins[tile] = bits;
    if (cost_out) cost_out[tile] = (uint8_t)(cost < 127u ? cost : 127u);
}

// [world][slots][8][32][3] -> row-major [H][W][3]
__global__ __launch_bounds__(256) void k_assemble(const uint8_t* g, uint32_t w, uint32_t h, uint32_t world,
                                                  uint32_t slots, uint32_t tiles_x, uint32_t skew, uint8_t* out) {
    const uint32_t px = blockIdx.x * 256u + threadIdx.x;
    if (px >= w * h) return;
    const uint32_t x = px % w, y = px / w;
    const uint32_t ty = y / RRT_TILE_H, txp = x / RRT_TILE_W;
    const uint32_t tile = ty * tiles_x + (txp + tiles_x - (skew * ty) % tiles_x) % tiles_x;     // undo the row rotation
    const uint32_t rank = tile % world, slot = tile / world;
    const size_t src = (((size_t)rank * slots + slot) * 256u + (y % RRT_TILE_H) * RRT_TILE_W + (x % RRT_TILE_W)) * 3;
    out[3 * (size_t)px] = g[src]; out[3 * (size_t)px + 1] = g[src + 1]; out[3 * (size_t)px + 2] = g[src + 2];
}

template <bool GROUPS>
__global__ __launch_bounds__(256) void k_trace_rays(DScene sc, const double* rays, int64_t n, int32_t* ent,
                                                    int32_t* tri, double* t, double* pos, double* nrm,
                                                    unsigned long long* stats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
    Closest c;
    if (sc.cull_mode == 0) closest_hit<0, GROUPS, GROUPS>(sc, kNoLds, o, d, -1, &c, stats);
    else if (sc.cull_mode == 1) closest_hit<1, GROUPS, GROUPS>(sc, kNoLds, o, d, -1, &c, stats);
    else closest_hit<2, GROUPS, GROUPS>(sc, kNoLds, o, d, -1, &c, stats);
    Hit h;
    finish_hit(sc, kNoLds, o, d, c, &h);
    const int32_t e = c.ent;
    if (ent) ent[i] = e;
    if (tri) tri[i] = h.tri;
    if (t) t[i] = h.hit ? h.t : 0.0;
    if (pos) { pos[3 * i] = h.pos.x; pos[3 * i + 1] = h.pos.y; pos[3 * i + 2] = h.pos.z; }
    if (nrm) { nrm[3 * i] = h.normal.x; nrm[3 * i + 1] = h.normal.y; nrm[3 * i + 2] = h.normal.z; }
}

// cast() as one thread per ray with a private record stack (test / probe API).
template <bool GROUPS, bool PL>
__global__ __launch_bounds__(256) void k_cast_rays(DScene sc, const double* rays, int64_t n, uint32_t max_bounce,
                                                   uint32_t prune, double* color, unsigned long long* stats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    V3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
    V3 sA[kMaxLevels], sP[kMaxLevels];
    double sR[kMaxLevels];
    uint32_t bounce = 0;
    int32_t medium = -1;
    DFrame fr{};                 // RNG stream of ray i: pixel = i, sample 0, seed 0 (as the oracle's cast_rays)
    V3 col;
    for (;;) {
        Closest c;
        if (sc.cull_mode == 0) closest_hit<0, GROUPS, GROUPS>(sc, kNoLds, o, d, medium, &c, stats);
        else if (sc.cull_mode == 1) closest_hit<1, GROUPS, GROUPS>(sc, kNoLds, o, d, medium, &c, stats);
        else closest_hit<2, GROUPS, GROUPS>(sc, kNoLds, o, d, medium, &c, stats);
        Hit h;
        finish_hit(sc, kNoLds, o, d, c, &h);
        if (!h.hit) { col = sky_color(sc, kNoLds, d); break; }
        const DMaterial m = sc.mats[h.mat];
        V3 A, Pm;
        if (sc.cull_mode == 0) shade_direct<0, GROUPS, PL>(sc, kNoLds, m, d, h.pos, h.normal, &A, &Pm, stats);
        else if (sc.cull_mode == 1) shade_direct<1, GROUPS, PL>(sc, kNoLds, m, d, h.pos, h.normal, &A, &Pm, stats);
        else shade_direct<2, GROUPS, PL>(sc, kNoLds, m, d, h.pos, h.normal, &A, &Pm, stats);
        const bool cont = !(bounce > max_bounce) && !(prune && m.mixR == 0.0);
        if (!cont) { col = A + Pm; break; }
        sA[bounce] = A; sP[bounce] = Pm; sR[bounce] = m.mixR;
        V3 ro, rd;
        continuation(fr, m, h.pos, h.normal, d, h.sphere, c.ent, bounce, (uint32_t)i, 0u, &ro, &rd, &medium);
        o = ro; d = rd; ++bounce;
    }
    for (uint32_t k = bounce; k-- > 0;) col = (sA[k] + col * sR[k]) + sP[k];
    color[3 * i] = col.x; color[3 * i + 1] = col.y; color[3 * i + 2] = col.z;
}

// The any-hit walk for caller rays (shadow-ray API): out[i] = 1 iff some entity answers ray i closer than lim2[i].
template <bool GROUPS>
__global__ __launch_bounds__(256) void k_occluded_rays(DScene sc, const double* rays, const double* lim2, int64_t n,
                                                       uint8_t* out, unsigned long long* stats) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
    const double lim = lim2[i];
    bool r;
    if (sc.cull_mode == 0) r = any_hit<0, GROUPS, false, GROUPS>(sc, kNoLds, o, d, lim, stats);
    else if (sc.cull_mode == 1) r = any_hit<1, GROUPS, false, GROUPS>(sc, kNoLds, o, d, lim, stats);
    else r = any_hit<2, GROUPS, false, GROUPS>(sc, kNoLds, o, d, lim, stats);
    out[i] = r ? 1 : 0;
}

__global__ void k_selftest(int op, const double* a, const double* b, int64_t n, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    double r;
    if (op == 0) r = x / y;
    else if (op == 1) r = sqrt(x);
    else if (op == 2) r = x * y + y;
    else r = powi(x, (int)y);
    out[i] = r;
}

// Persistent grid of the trace kernels: 256 CUs x 64 blocks of 256 threads -- 16 rounds of resident
// blocks, so the hardware block scheduler evens out the tail (measured, primary kernel of a cfg2 frame:
// 4096 blocks 1.63 ms, 8192 1.51, 16384 1.47, 32768 1.51, 65528 1.64).  RRT_TRACE_GRID overrides, for tuning.
inline uint32_t trace_grid() {
    static const uint32_t g = [] {
        const char* e = getenv("RRT_TRACE_GRID");
        const long v = e ? atol(e) : 0;
        return (uint32_t)(v > 0 && v <= (long)kMaxSegs ? v : (long)kMaxSegs);   // the cap; launch_primary sizes the grid by the chunk count
    }();
    return g;
}
inline uint32_t bounce_grid() {
    static const uint32_t g = [] {
        const char* e = getenv("RRT_BOUNCE_GRID");
        const long v = e ? atol(e) : 0;
        return (uint32_t)(v > 0 && v <= 65535 ? v : 256 * 8);   // measured: 1024-3072 blocks within 1 %, 512 and 8192 slower
    }();
    return g;
}
// + the primary kernel's append counter / the bounce kernel's unit counter, append counter and segment tables
inline size_t lds_bytes(const DScene& sc) { return (size_t)lds_layout_bytes(sc) + 16 + (7 * kSegPerBlock + 3) * 4 + 4; }

}  // namespace

// Kernel variant by scene: cull mode x (general scene: Scene<T> groups or bare Triangle entities present) x (entity
// table in LDS) x (general light list: anything but the reference's single DirectionalLight).
// A world with a general light list always takes the GROUPS instantiation (it handles group-less worlds too), so the
// shadow-ray code is compiled 6 times per kernel, not 12.
#define RRT_PICK(name)                                                                                           \
    [&]() {                                                                                                      \
        const bool pl = sc.general_lights != 0, g = sc.n_groups > 0 || sc.n_tri_ents > 0 || pl, sm = sc.lds_ents != 0;  \
        const int c = sc.cull_mode;                                                                              \
        if (pl) {                                                                                                \
            if (sm) return c == 0 ? name<0, true, true, true> : c == 1 ? name<1, true, true, true> : name<2, true, true, true>;      \
            return c == 0 ? name<0, true, false, true> : c == 1 ? name<1, true, false, true> : name<2, true, false, true>;           \
        }                                                                                                        \
        if (g) {                                                                                                 \
            if (sm) return c == 0 ? name<0, true, true, false> : c == 1 ? name<1, true, true, false> : name<2, true, true, false>;   \
            return c == 0 ? name<0, true, false, false> : c == 1 ? name<1, true, false, false> : name<2, true, false, false>;        \
        }                                                                                                        \
        if (sm) return c == 0 ? name<0, false, true, false> : c == 1 ? name<1, false, true, false> : name<2, false, true, false>;    \
        return c == 0 ? name<0, false, false, false> : c == 1 ? name<1, false, false, false> : name<2, false, false, false>;         \
    }()

#if RRT_TU_HAS(0)
uint32_t primary_chunks_per_block() {
    static const uint32_t per_block = [] { const char* e = getenv("RRT_CHUNKS_PER_BLOCK"); const long v = e ? atol(e) : 0; return (uint32_t)(v > 0 ? v : 16); }();
    return per_block;
}
void launch_primary(const DScene& sc, const DScene* d_sc, const DWork& wk, const DFrame& fr, int32_t* d_ids, double* d_t, void* stream) {
    (void)d_sc;
    // About sixteen 256-path chunks per persistent block, in a grid that is a multiple of 8 (the XCD-aware order) and of
    // at least 2048 blocks (a small launch still fills the chip).  Sweeps on the full cfg2 frame (130 k chunks), one
    // frame at a time, before the sky-only arm: 4096 blocks 1.63 ms, 8192 1.51, 16384 1.47, 32768 1.51.  With that arm
    // most chunks are cheap and the per-block LDS staging weighs more: three frames in flight, 16 chunks a block give
    // 36.1 Grays/s, 8: 35.4, 32: 35.3 (one frame at a time 8 is 2 % ahead).  Height field, 259 k chunks a launch: 16384
    // blocks 57.0 ms, 32768 56.5.
    auto k = RRT_PICK(k_wave_primary);
    hipLaunchKernelGGL(k, dim3(fr.seg_grid), dim3(256), lds_bytes(sc), (hipStream_t)stream, RRT_SC_ARG(sc, d_sc), wk, fr, d_ids, d_t);      // fr.seg_grid = primary_grid(fr)
}
// The primary kernel's grid for a batch (every slot counted; with tile lists the kernel itself uses fewer blocks, see
// seg_layout).  Also the number of queue segments the bounce kernels may meet: DFrame::seg_grid.
uint32_t primary_grid(const DFrame& fr) {
    const uint64_t chunks = ((uint64_t)fr.ns * fr.npx_local) / 256;
    const uint64_t per_block = primary_chunks_per_block();
    uint64_t want = (chunks / per_block + 2047) / 2048 * 2048;      // in steps of 2048 blocks = two rounds of resident blocks
    if (want > chunks) want = chunks;
    if (want > trace_grid()) want = trace_grid();
    if (want >= 8) want -= want % 8;
    return (uint32_t)(want ? want : 1);
}
#endif
#if RRT_TU_HAS(1)
void launch_bounce(const DScene& sc, const DScene* d_sc, const DWork& wk, const DFrame& fr, uint32_t bounce, void* stream) {
    (void)d_sc;
    auto k = RRT_PICK(k_wave_bounce);
    uint32_t grid = bounce_grid();
    const uint32_t need = (fr.seg_grid + kSegPerBlock - 1u) / kSegPerBlock;      // a block takes at most kSegPerBlock segments
    if (grid < need) grid = (need + 7u) / 8u * 8u;
    static const uint32_t rpb = [] { const char* e = getenv("RRT_BOUNCE_RPB"); const long v = e ? atol(e) : 0; return (uint32_t)(v > 0 ? v : 2048); }();
    DFrame f = fr;
    f.bounce_rays_per_block = rpb;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes(sc), (hipStream_t)stream, RRT_SC_ARG(sc, d_sc), wk, f, bounce);
}
#endif
#if RRT_TU_HAS(0)
void launch_accumulate(const DWork& wk, const DFrame& fr, bool final_batch, uint8_t* d_rgb8, double* d_color, void* stream,
                       uint32_t tile0, uint32_t n_tiles) {
    // tile slots [tile0, tile0 + n_tiles) of this rank (n_tiles = 0: all of them).  rrt_render resolves the frame in
    // bands of tile rows, so that a band's pixels travel to the host while the next band is resolved.
    if (n_tiles == 0u) { tile0 = 0u; n_tiles = fr.npx_local / 256u; }
    if (fr.pack_log2 >= 4u) {
        hipLaunchKernelGGL(final_batch ? k_accumulate_quad<true> : k_accumulate_quad<false>, dim3(4u * n_tiles), dim3(256), 0,
                           (hipStream_t)stream, wk, fr, d_rgb8, d_color, 4u * tile0);
        return;
    }
    hipLaunchKernelGGL(final_batch ? k_accumulate<true> : k_accumulate<false>, dim3(n_tiles), dim3(256), 0,
                       (hipStream_t)stream, wk, fr, d_rgb8, d_color, tile0);
}
void launch_cover(const DMesh& mesh, int32_t n_entries, const DFrame& fr, uint32_t slot, void* stream) {
    if (n_entries <= 0) return;
    CoverParams p{};
    double o1 = 0.0, omax = 0.0;
    for (int a = 0; a < 3; ++a) {
        p.eye[a] = fr.cam.eye[a];
        o1 += fabs(fr.cam.eye[a] - mesh.center[a]);
        omax = fmax(omax, fabs(fr.cam.eye[a]));
    }
    p.lens_z = fr.cam.lens_z; p.lens_w = fr.cam.lens_w; p.lens_h = fr.cam.lens_h;
    // make_cull_ray's bounds for a ray from the eye (|d|_1 <= sqrt 3), doubled; + the rounding of the projection itself
    const double S = mesh.edge_max, R = o1 + mesh.radius;
    p.delta = 2.0 * (2.5e-7 * S * S * 1.7320508075688774 * (R + S)) * 1.000001 + 1e-9 * (omax + mesh.abs_max) + 1e-300;
    const double dt = 2.0 * (8e-8 * S * S * (R + S)) * 1.000001;
    p.z_near = 2.0 * dt + 1e-9 * (omax + mesh.abs_max) + 1e-300;
    p.width = fr.width; p.height = fr.height; p.stride = fr.cover_stride; p.bytes = fr.cover_bytes;
    const uint32_t lanes_log2 = n_entries >= (1 << 20) ? 0u : 3u;
    const uint64_t threads = (uint64_t)n_entries << lanes_log2;
    hipLaunchKernelGGL(k_cover, dim3((uint32_t)((threads + 255u) / 256u)), dim3(256), 0, (hipStream_t)stream, mesh.tris, n_entries, p,
                       const_cast<uint8_t*>(fr.cover) + (size_t)slot * fr.cover_bytes, lanes_log2);
}
void launch_bins(const DScene& sc, const DFrame& fr, void* stream) {
    static_assert(RRT_TILE_W == 8 * kCoverCell, "a tile row is one 8-byte word of the mask");
    BinParams p{};
    double omax = 0.0;
    for (int a = 0; a < 3; ++a) {
        p.eye[a] = fr.cam.eye[a];
        omax = fmax(omax, fabs(fr.cam.eye[a]));
    }
    p.lens_z = fr.cam.lens_z; p.lens_w = fr.cam.lens_w; p.lens_h = fr.cam.lens_h;
    p.z_slack = 1e-9 * omax + 1e-300;
    p.width = fr.width; p.height = fr.height; p.tiles_x = fr.tiles_x; p.tiles_y = fr.tiles_y;
    const uint32_t n = fr.tiles_x * fr.tiles_y;
    hipLaunchKernelGGL(k_bins, dim3((n + 255u) / 256u), dim3(256), 0, (hipStream_t)stream, sc, p, fr.cover, fr.cover ? fr.n_cover : 0u,
                       fr.cover_stride, fr.cover_bytes, const_cast<uint64_t*>(fr.tile_bins), const_cast<uint8_t*>(fr.tile_cost));
}
void launch_lists(const DFrame& fr, bool dealt, bool all_heavy, void* stream) {
    uint32_t scatter = 1u;
    if (dealt) {
        const uint32_t n = fr.tiles_x * fr.tiles_y;
        const auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
        scatter = (uint32_t)(0.6180339887498949 * (double)n) | 1u;
        while (n > 1u && gcd(scatter % n, n) != 1u) scatter += 2u;
        if (n <= 1u) scatter = 1u;
    }
    hipLaunchKernelGGL(k_lists, dim3(1), dim3(1024), 0, (hipStream_t)stream, fr, dealt ? 1u : 0u, all_heavy ? 1u : 0u, scatter,
                       const_cast<int32_t*>(fr.slot_tile), const_cast<int32_t*>(fr.order), const_cast<uint32_t*>(fr.list_n));
}
void launch_sky(const DScene& sc, const DWork& wk, const DFrame& fr, int32_t* d_ids, double* d_t, void* stream) {
    // sized for every slot being sky-only (the count is on the device): blocks beyond the work leave at once
    const uint64_t chunks = ((uint64_t)fr.ns * fr.npx_local) / 256;
    uint64_t want = (chunks + 7) / 8;
    if (want > 16384) want = 16384;
    if (want == 0) want = 1;
    hipLaunchKernelGGL(k_sky, dim3((uint32_t)want), dim3(256), 0, (hipStream_t)stream, sc, wk, fr, d_ids, d_t);
}
void launch_assemble_dealt(const uint8_t* d_gathered, uint32_t w, uint32_t h, uint32_t world, uint8_t* d_rgb8, void* stream) {
    const uint32_t tiles_x = (w + RRT_TILE_W - 1) / RRT_TILE_W, tiles_y = (h + RRT_TILE_H - 1) / RRT_TILE_H;
    const uint32_t slots = (tiles_x * tiles_y + world - 1) / world;
    hipLaunchKernelGGL(k_assemble_dealt, dim3(world * slots), dim3(256), 0, (hipStream_t)stream, d_gathered, w, h, world, slots, tiles_x, d_rgb8);
}
void launch_assemble(const uint8_t* d_gathered, uint32_t w, uint32_t h, uint32_t world, uint8_t* d_rgb8, void* stream) {
    const uint32_t tiles_x = (w + RRT_TILE_W - 1) / RRT_TILE_W, tiles_y = (h + RRT_TILE_H - 1) / RRT_TILE_H;
    const uint32_t slots = (tiles_x * tiles_y + world - 1) / world;
    const uint32_t n = w * h;
    hipLaunchKernelGGL(k_assemble, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_gathered, w, h, world,
                       slots, tiles_x, tile_skew(tiles_x, world), d_rgb8);
}
void launch_trace_rays(const DScene& sc, const double* d_rays, int64_t n, int32_t* ent, int32_t* tri, double* t,
                       double* pos, double* nrm, unsigned long long* stats, void* stream) {
    hipLaunchKernelGGL(sc.n_groups > 0 || sc.n_tri_ents > 0 ? k_trace_rays<true> : k_trace_rays<false>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sc, d_rays,
                       n, ent, tri, t, pos, nrm, stats);
}
void launch_cast_rays(const DScene& sc, const double* d_rays, int64_t n, uint32_t max_bounce, uint32_t prune,
                      double* color, unsigned long long* stats, void* stream) {
    auto k = sc.general_lights ? k_cast_rays<true, true> : sc.n_groups > 0 || sc.n_tri_ents > 0 ? k_cast_rays<true, false> : k_cast_rays<false, false>;
    hipLaunchKernelGGL(k, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sc, d_rays, n, max_bounce, prune,
                       color, stats);
}
void launch_occluded_rays(const DScene& sc, const double* d_rays, const double* d_lim2, int64_t n, uint8_t* out,
                          unsigned long long* stats, void* stream) {
    hipLaunchKernelGGL(sc.n_groups > 0 || sc.n_tri_ents > 0 ? k_occluded_rays<true> : k_occluded_rays<false>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, sc, d_rays, d_lim2, n, out, stats);
}
void launch_selftest(int op, const double* a, const double* b, int64_t n, double* out, void* stream) {
    hipLaunchKernelGGL(k_selftest, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, a, b, n,
                       out);
}
// ---------------------------------------------------------------------------
// Scratch of rrt_render_device_noalloc.  In host code `scratch` is only a stand-in for the symbol,
// not a device address: it is never passed to a kernel or a memset.  The address of each device's
// copy is looked up once (per device, with that device current) and kept; every use goes through it.
// ---------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
constexpr int kMaxDevices = 64;
static unsigned char* g_scratch_ptr[kMaxDevices];      // 0 = not resolved yet
static std::mutex g_scratch_mu;

int scratch_base(int device, unsigned char** ptr) {
    *ptr = nullptr;
    if (device < 0 || device >= kMaxDevices) return (int)hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        int cur = -1;
        hipError_t e = hipGetDevice(&cur);
        if (e != hipSuccess) return (int)e;
        if (cur != device) return (int)hipErrorInvalidDevice;       // the caller makes `device` current first
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return (int)hipErrorInvalidSymbol;
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *ptr = g_scratch_ptr[device];
    return 0;
}

void debug_phases(unsigned long long out[32]) {
#if RRT_PROBE
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), 32 * sizeof(unsigned long long));
    unsigned long long z[32] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
#else
    for (int i = 0; i < 16; ++i) out[i] = 0;
#endif
}
void philox4x32_10_host(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

#endif      // RRT_TU_HAS(0)
}  // namespace rrt
