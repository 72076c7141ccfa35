// C ABI (include/rrt.h): scene container, flattening into HBM, wavefront
// frame scheduling.  Host code only; kernels live in rrt_kernels.hip.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "host_io.hpp"
#include "kd_build.hpp"
#include "rrt_device.h"

namespace {

thread_local std::string g_err;
thread_local int g_device = 0;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

// No C++ exception may cross the C ABI (the caller may be Rust, C or ctypes): every entry point that
// can allocate is a function-try-block ending here.
int on_exception() {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(RRT_E_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(RRT_E_STATE, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(RRT_E_STATE, "internal error");
    }
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RRT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t ensure(size_t n) {
        if (n <= bytes && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    hipError_t upload(const void* src, size_t n) {
        hipError_t e = ensure(n);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpy(p, src, n, hipMemcpyHostToDevice);
    }
    template <class T>
    T* as() const { return (T*)p; }
};

struct HostMesh {
    std::vector<double> verts;       // 9 per triangle (v0, v1, v2): Triangle::translate moves these, the edges follow
    std::vector<rrt::DTri> tris;
    std::vector<double> normals;
    rrt::KdTreeFlat tree;
    int material = 0;
    bool stale_boxes = false;        // went through a literal translate: the boxes no longer match the split planes
    // device copies
    DevBuf d_nodes, d_boxes, d_leaves, d_refs, d_tris, d_normals, d_cull, d_grp, d_top;
};

// Scene<T> (scene.rs:31-47): kd-tree over member entities
struct HostGroup {
    std::vector<int32_t> members;   // entity ids, in the order given (= item numbers of the tree)
    rrt::KdTreeFlat tree;
    bool stale_boxes = false;
    DevBuf d_nodes, d_boxes, d_leaves, d_members;
};

struct SkyFace {
    std::vector<uint8_t> rgb;
    uint32_t w = 0, h = 0;
    DevBuf d;
};

struct TimedLaunch {
    hipEvent_t a = nullptr, b = nullptr;
    int kind = 0;            // 0 = primary kernel, 1 = bounce levels, 2 = sky kernel
    uint32_t launches = 1;   // kernel launches between a and b
};

struct WorkSet {
    hipStream_t stream = nullptr;
    DevBuf w_q[2], w_rec, w_samp, w_samp32, w_accum, w_zero, w_bins, w_lists, w_seg;   // w_zero: what a frame starts from zero, see zero_layout
    hipStream_t side = nullptr;                 // the sky kernel runs here, beside the wavefront kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    uint32_t spp = 0;                           // of the last frame (rrt_get_stats: paths of a dealt frame = device count x spp)
    bool dealt = false, lists = false;
    // timing and counters of the last frame rendered with this set
    std::vector<TimedLaunch> timed;
    size_t timed_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool frame_timed = false;                   // the last frame recorded events (the no-alloc entry point does not)
    const unsigned long long* stats_src = nullptr;   // device counters of the last frame (w_stats or the static scratch)
    rrt_stats stats{};
    WorkSet() = default;
    WorkSet(const WorkSet&) = delete;
    WorkSet& operator=(const WorkSet&) = delete;
    ~WorkSet() {
        for (auto& t : timed) {
            if (t.a) (void)hipEventDestroy(t.a);
            if (t.b) (void)hipEventDestroy(t.b);
        }
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
    }
};

}  // namespace

namespace rrt {
void set_last_error(const std::string& msg) { g_err = msg; }
}  // namespace rrt

struct rrt_scene {
    int device = 0;
    std::vector<rrt::DMaterial> mats;
    std::vector<rrt::DSphere> spheres;
    std::vector<std::unique_ptr<HostMesh>> meshes;
    std::vector<rrt::DMesh> dmeshes;          // host copies of what commit uploaded (device pointers + cull constants)
    std::vector<std::pair<int32_t, int32_t>> cover;   // meshes that carry a primary-ray coverage mask: (mesh, stream entries), slot = position
    std::vector<std::unique_ptr<HostGroup>> groups;
    std::vector<double> sphere_radius;   // Sphere::bounding_box needs r, the kernels only r^2
    std::vector<rrt::DEntity> ents;
    std::vector<rrt::DLight> lights;         // world.light_sources, in push order (common.rs:12)
    std::vector<rrt::DTriEnt> tri_ents;      // bare Triangle entities
    std::vector<double> tri_verts;           // their 9 vertex coordinates each (Triangle::translate moves vertices; the edges follow)
    int shadow_mode = 0;
    SkyFace sky[6];
    double ambient = 0.15;  // tracer.rs:73
    uint64_t work_budget = 48ull << 30;   // of 288 GB, per stream a scene renders on; a batch is also capped at 178.9 M paths (plan_frame)
    int cull_mode = 1;
    int timing = 2;        // rrt_set_timing: 0 no events, 1 frame begin/end, 2 + per-kernel pairs
    int sky_stream = 1;    // rrt_set_sky_stream: 1 = the sky-only kernel on a side stream, 0 = on the frame's own

    bool dirty = true;
    DevBuf d_ents, d_spheres, d_meshes, d_mats, d_groups, d_lights, d_tri_ents, d_mesh_stage;
    rrt::DScene dscene{};
    DevBuf d_dscene;                // the same block in device memory (kernels may take it by pointer: RRT_SC_BY_PTR)

    DevBuf w_ids, w_t, w_color, w_rgb;  // staging for the host-output render
    hipStream_t host_stream = nullptr, copy_stream = nullptr;    // rrt_render: its own stream + the one its D2H copies run on
    hipEvent_t band_ev[8] = {};                                  // "band k is resolved" (rrt_render into pinned memory)
    ~rrt_scene() {
        for (hipEvent_t e : band_ev) if (e) (void)hipEventDestroy(e);
        if (host_stream) (void)hipStreamDestroy(host_stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }

    // Work buffers and the record of the last frame, one set per stream the caller renders on: frames on different
    // streams may be in flight together (each has its own queues, records and samples; the scene is read-only to
    // them), frames on one stream are ordered by it and share a set.
    std::vector<std::unique_ptr<WorkSet>> sets;
    WorkSet noalloc_set;            // rrt_render_device_noalloc: no buffers of its own (the static scratch), no events
    WorkSet* last = nullptr;        // what rrt_get_stats reports
};

namespace {

constexpr size_t kMaxWorkSets = 8;

// The work set of `stream` (created on first use; at most kMaxWorkSets streams per scene).
int work_set_for(rrt_scene* s, hipStream_t stream, WorkSet** out) {
    for (auto& w : s->sets)
        if (w->stream == stream) { *out = w.get(); return RRT_OK; }
    if (s->sets.size() >= kMaxWorkSets)
        return fail(RRT_E_STATE, "a scene renders on at most " + std::to_string(kMaxWorkSets) + " different streams (one set of work buffers each)");
    s->sets.emplace_back(new WorkSet);
    s->sets.back()->stream = stream;
    *out = s->sets.back().get();
    return RRT_OK;
}

}  // namespace

namespace {

int commit(rrt_scene* s) {
    HIP_TRY(hipSetDevice(s->device));
    if (!s->dirty) return RRT_OK;
    for (int f = 0; f < 6; ++f)
        if (s->sky[f].w == 0) return fail(RRT_E_STATE, "sky face " + std::to_string(f) + " not set (rrt_set_sky_face)");
    std::vector<rrt::DMesh> dm(s->meshes.size());
    s->cover.clear();
    static const bool cover_on = [] { const char* e = getenv("RRT_COVER"); return !e || atoi(e) != 0; }();   // A/B knob
    for (size_t i = 0; i < s->meshes.size(); ++i) {
        HostMesh& m = *s->meshes[i];
        const rrt::KdTreeFlat& T = m.tree;
        const size_t nn = (size_t)T.n_nodes();
        std::vector<rrt::DNodeI> nodes(nn);
        std::vector<rrt::DBox> boxes(nn);
        std::vector<rrt::DLeaf> leaves;
        for (size_t k = 0; k < nn; ++k) {
            for (int a = 0; a < 3; ++a) {
                boxes[k].mn[a] = T.boxes[6 * k + a];
                boxes[k].mx[a] = T.boxes[6 * k + 3 + a];
            }
            nodes[k].split = T.split[k];
            nodes[k].axis = T.axis[k];
            if (T.axis[k] >= 0) {
                // pre-order: the left child directly follows its parent
                if (T.child[2 * k] != (int32_t)k + 1) return fail(RRT_E_STATE, "kd-tree is not in pre-order");
                nodes[k].right = T.child[2 * k + 1];
            } else {
                rrt::DLeaf L{};
                for (int a = 0; a < 3; ++a) {
                    L.mn[a] = boxes[k].mn[a];
                    L.mx[a] = boxes[k].mx[a];
                }
                for (int r = 0; r < 6; ++r) L.rope[r] = T.ropes[6 * k + r];
                L.first = T.child[2 * k];
                L.count = T.child[2 * k + 1];
                nodes[k].right = (int32_t)leaves.size();
                leaves.push_back(L);
            }
        }
        HIP_TRY(m.d_boxes.upload(boxes.data(), boxes.size() * sizeof(rrt::DBox)));
        // Leaf-ordered stream, every leaf's run aligned to kGroup entries and padded to a multiple of
        // it (pad = copy of the run's last entry, never tested: the kernels mask by the leaf's count).
        constexpr int32_t G = rrt::kGroup;
        std::vector<rrt::DRef> refs;
        refs.reserve(T.leaf_refs.size() + (size_t)G * leaves.size());
        std::vector<std::pair<uint32_t, int32_t>> order;      // (Morton key, position in the reference's leaf list)
        std::vector<int32_t> leaf_node(leaves.size());
        for (size_t k = 0; k < nn; ++k)
            if (nodes[k].axis < 0) leaf_node[(size_t)nodes[k].right] = (int32_t)k;
        for (size_t li = 0; li < leaves.size(); ++li) {
            rrt::DLeaf& L = leaves[li];
            const int32_t src = L.first;
            L.first = (int32_t)refs.size();
            nodes[(size_t)leaf_node[li]].run[0] = L.first;
            nodes[(size_t)leaf_node[li]].run[1] = L.count;
            // spatial order inside the leaf: Morton code of the triangle centroids within their own bounds (10 bits
            // an axis), ties in list order; the list position goes along as the rank that decides equal distances
            order.clear();
            double cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            const auto centroid = [&](int32_t tri, int a) { const rrt::DTri& t = m.tris[(size_t)tri]; return t.v0[a] + (t.e1[a] + t.e2[a]) / 3.0; };
            for (int32_t k = 0; k < L.count; ++k)
                for (int a = 0; a < 3; ++a) {
                    const double c = centroid(T.leaf_refs[(size_t)src + k], a);
                    cmn[a] = std::fmin(cmn[a], c);
                    cmx[a] = std::fmax(cmx[a], c);
                }
            for (int32_t k = 0; k < L.count; ++k) {
                uint32_t key = 0;
                for (int a = 0; a < 3; ++a) {
                    const double ext = cmx[a] - cmn[a];
                    double f = ext > 0.0 ? (centroid(T.leaf_refs[(size_t)src + k], a) - cmn[a]) / ext : 0.0;
                    if (!(f >= 0.0)) f = 0.0;
                    uint32_t q = (uint32_t)std::fmin(f * 1024.0, 1023.0);
                    for (int bit = 0; bit < 10; ++bit) key |= ((q >> bit) & 1u) << (3 * bit + a);
                }
                order.emplace_back(key, k);
            }
            std::stable_sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
            for (const auto& kv : order) refs.push_back(rrt::DRef{T.leaf_refs[(size_t)src + kv.second], kv.second});
            while (refs.size() % G) refs.push_back(refs.back());
        }
        if (refs.size() > 0x7fffffffull - 64) return fail(RRT_E_INVALID, "mesh too large (triangle stream index exceeds 31 bits)");
        HIP_TRY(m.d_nodes.upload(nodes.data(), nodes.size() * sizeof(rrt::DNodeI)));   // after the runs are final
        HIP_TRY(m.d_refs.upload(refs.data(), refs.size() * sizeof(rrt::DRef)));
        std::vector<rrt::DTri> stream(refs.size());
        for (size_t j = 0; j < stream.size(); ++j) stream[j] = m.tris[(size_t)refs[j].tri];
        {
            // the un-culled leaf loop prefetches up to kStreamPad entries past a leaf: keep them readable
            std::vector<rrt::DTri> padded(stream);
            padded.resize(stream.size() + rrt::kStreamPad, stream.empty() ? rrt::DTri{} : stream.back());
            HIP_TRY(m.d_tris.upload(padded.data(), padded.size() * sizeof(rrt::DTri)));
        }
        // outward-rounded f32 boxes of the stream entries + the cull constants
        std::vector<rrt::DBox32> cull(stream.size());
        double edge_max = 0.0;
        // bounds of the triangles themselves: the cull's distance bound must hold for where the triangles ARE,
        // which after a literal rrt_translate_entity is no longer the tree's (stale) root box
        double tmn[3] = {INFINITY, INFINITY, INFINITY}, tmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t j = 0; j < stream.size(); ++j) {
            const rrt::DTri& t = stream[j];
            double l1 = 0.0, l2 = 0.0;
            for (int a = 0; a < 3; ++a) {
                const double p0 = t.v0[a], p1 = t.v0[a] + t.e1[a], p2 = t.v0[a] + t.e2[a];
                const double lo = std::fmin(p0, std::fmin(p1, p2)), hi = std::fmax(p0, std::fmax(p1, p2));
                tmn[a] = std::fmin(tmn[a], lo);
                tmx[a] = std::fmax(tmx[a], hi);
                float flo = (float)lo, fhi = (float)hi;
                if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
                if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
                cull[j].v[2 * a] = std::nextafterf(flo, -INFINITY);   // one more ulp for the p0+e rounding
                cull[j].v[2 * a + 1] = std::nextafterf(fhi, INFINITY);
                l1 += t.e1[a] * t.e1[a];
                l2 += t.e2[a] * t.e2[a];
            }
            edge_max = std::fmax(edge_max, std::sqrt(std::fmax(l1, l2)));
        }
        // one box per group of kGroup consecutive entries: the union of its entries' boxes
        std::vector<rrt::DBox32> grp(stream.size() / G + rrt::kStreamPad);
        for (size_t g = 0; g < stream.size() / G; ++g) {
            rrt::DBox32 u = cull[g * G];
            for (int32_t k = 1; k < G; ++k)
                for (int a = 0; a < 3; ++a) {
                    u.v[2 * a] = std::fmin(u.v[2 * a], cull[g * G + k].v[2 * a]);
                    u.v[2 * a + 1] = std::fmax(u.v[2 * a + 1], cull[g * G + k].v[2 * a + 1]);
                }
            grp[g] = u;
        }
        for (size_t g = stream.size() / G; g < grp.size(); ++g) grp[g] = stream.empty() ? rrt::DBox32{} : grp[stream.size() / G - 1];
        // per leaf: the union of its entries' cull boxes (an empty leaf keeps an inverted box and is skipped by its count)
        for (rrt::DLeaf& L : leaves) {
            rrt::DBox32 u{{INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY}};
            for (int32_t k = 0; k < L.count; ++k)
                for (int a = 0; a < 3; ++a) {
                    u.v[2 * a] = std::fmin(u.v[2 * a], cull[(size_t)L.first + k].v[2 * a]);
                    u.v[2 * a + 1] = std::fmax(u.v[2 * a + 1], cull[(size_t)L.first + k].v[2 * a + 1]);
                }
            L.content = u;
        }
        HIP_TRY(m.d_leaves.upload(leaves.data(), leaves.size() * sizeof(rrt::DLeaf)));
        HIP_TRY(m.d_cull.upload(cull.data(), cull.size() * sizeof(rrt::DBox32)));
        HIP_TRY(m.d_grp.upload(grp.data(), grp.size() * sizeof(rrt::DBox32)));
        dm[i].grp = m.d_grp.as<rrt::DBox32>();
        dm[i].cull = m.d_cull.as<rrt::DBox32>();
        dm[i].edge_max = edge_max * (1.0 + 1e-9);
        {
            double r2 = 0.0, am = 0.0;
            for (int a = 0; a < 3; ++a) {
                // (one ulp of slack per bound: p0 + e is the rounded sum, not the stored vertex)
                const double lo = std::nextafter(std::fmin(tmn[a], T.boxes[a]), -INFINITY), hi = std::nextafter(std::fmax(tmx[a], T.boxes[3 + a]), INFINITY);
                dm[i].center[a] = 0.5 * (lo + hi);
                r2 += 0.25 * (hi - lo) * (hi - lo);
                am = std::fmax(am, std::fmax(std::fabs(lo), std::fabs(hi)));
            }
            dm[i].radius = std::sqrt(r2) * (1.0 + 1e-9);
            dm[i].abs_max = am;
        }
        HIP_TRY(m.d_normals.upload(m.normals.data(), m.normals.size() * sizeof(double)));
        dm[i].nodes = m.d_nodes.as<rrt::DNodeI>();
        dm[i].boxes = m.d_boxes.as<rrt::DBox>();
        dm[i].leaves = m.d_leaves.as<rrt::DLeaf>();
        dm[i].leaf_refs = m.d_refs.as<rrt::DRef>();
        dm[i].tris = m.d_tris.as<rrt::DTri>();
        dm[i].normals = m.d_normals.as<double>();
        dm[i].material = m.material;
        dm[i].flags = m.stale_boxes ? rrt::kTreeStaleBoxes : 0;
        if (cover_on && stream.size() >= (size_t)rrt::kCoverMinEntries && s->cover.size() < (size_t)rrt::kMaxCover) {
            s->cover.emplace_back((int32_t)i, (int32_t)stream.size());
            dm[i].flags |= (int32_t)s->cover.size() << rrt::kCoverShift;      // slot + 1
        }
    }
    HIP_TRY(s->d_meshes.upload(dm.data(), dm.size() * sizeof(rrt::DMesh)));
    s->dmeshes = dm;
    std::vector<rrt::DGroup> dg(s->groups.size());
    for (size_t i = 0; i < s->groups.size(); ++i) {
        HostGroup& g = *s->groups[i];
        const rrt::KdTreeFlat& T = g.tree;
        const size_t nn = (size_t)T.n_nodes();
        std::vector<rrt::DNodeI> nodes(nn);
        std::vector<rrt::DBox> boxes(nn);
        std::vector<rrt::DLeaf> leaves;
        for (size_t k = 0; k < nn; ++k) {
            for (int a = 0; a < 3; ++a) {
                boxes[k].mn[a] = T.boxes[6 * k + a];
                boxes[k].mx[a] = T.boxes[6 * k + 3 + a];
            }
            nodes[k].split = T.split[k];
            nodes[k].axis = T.axis[k];
            if (T.axis[k] >= 0) {
                if (T.child[2 * k] != (int32_t)k + 1) return fail(RRT_E_STATE, "kd-tree is not in pre-order");
                nodes[k].right = T.child[2 * k + 1];
            } else {
                rrt::DLeaf L{};
                for (int a = 0; a < 3; ++a) {
                    L.mn[a] = boxes[k].mn[a];
                    L.mx[a] = boxes[k].mx[a];
                }
                for (int r = 0; r < 6; ++r) L.rope[r] = T.ropes[6 * k + r];
                L.first = T.child[2 * k];
                L.count = T.child[2 * k + 1];
                nodes[k].run[0] = L.first;
                nodes[k].run[1] = L.count;
                nodes[k].right = (int32_t)leaves.size();
                leaves.push_back(L);
            }
        }
        std::vector<int32_t> member_ids(T.leaf_refs.size());
        for (size_t j = 0; j < member_ids.size(); ++j) member_ids[j] = g.members[(size_t)T.leaf_refs[j]];
        HIP_TRY(g.d_nodes.upload(nodes.data(), nodes.size() * sizeof(rrt::DNodeI)));
        HIP_TRY(g.d_boxes.upload(boxes.data(), boxes.size() * sizeof(rrt::DBox)));
        HIP_TRY(g.d_leaves.upload(leaves.data(), leaves.size() * sizeof(rrt::DLeaf)));
        HIP_TRY(g.d_members.upload(member_ids.data(), member_ids.size() * sizeof(int32_t)));
        dg[i].nodes = g.d_nodes.as<rrt::DNodeI>();
        dg[i].boxes = g.d_boxes.as<rrt::DBox>();
        dg[i].leaves = g.d_leaves.as<rrt::DLeaf>();
        dg[i].members = g.d_members.as<int32_t>();
        dg[i].flags = g.stale_boxes ? rrt::kTreeStaleBoxes : 0;
    }
    HIP_TRY(s->d_groups.upload(dg.data(), dg.size() * sizeof(rrt::DGroup)));
    HIP_TRY(s->d_spheres.upload(s->spheres.data(), s->spheres.size() * sizeof(rrt::DSphere)));
    HIP_TRY(s->d_mats.upload(s->mats.data(), s->mats.size() * sizeof(rrt::DMaterial)));
    HIP_TRY(s->d_ents.upload(s->ents.data(), s->ents.size() * sizeof(rrt::DEntity)));
    HIP_TRY(s->d_lights.upload(s->lights.data(), s->lights.size() * sizeof(rrt::DLight)));
    HIP_TRY(s->d_tri_ents.upload(s->tri_ents.data(), s->tri_ents.size() * sizeof(rrt::DTriEnt)));
    rrt::DScene& d = s->dscene;
    d.lights = s->d_lights.as<rrt::DLight>();
    d.n_lights = (int32_t)s->lights.size();
    // the reference's own world: light_sources = [one DirectionalLight] (tracer.rs:55-65) -- or none at all
    d.general_lights = (s->lights.empty() || (s->lights.size() == 1 && s->lights[0].kind == 0)) ? 0 : 1;
    // a world with nested scenes takes the general instantiations too (the nested walk is compiled only there)
    for (const auto& g : s->groups)
        for (int32_t m2 : g->members)
            if ((s->ents[(size_t)m2].kind & ~rrt::kEntityMember) == 2) d.general_lights = 1;
    d.tri_ents = s->d_tri_ents.as<rrt::DTriEnt>();
    d.n_tri_ents = (int32_t)s->tri_ents.size();
    d.shadow_mode = s->shadow_mode;
    d.ents = s->d_ents.as<rrt::DEntity>();
    d.spheres = s->d_spheres.as<rrt::DSphere>();
    d.meshes = s->d_meshes.as<rrt::DMesh>();
    d.mats = s->d_mats.as<rrt::DMaterial>();
    d.groups = s->d_groups.as<rrt::DGroup>();
    d.n_groups = (int32_t)s->groups.size();
    d.n_ents = (int32_t)s->ents.size();
    d.has_light = (!d.general_lights && !s->lights.empty()) ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
        d.light_dir[a] = d.has_light ? s->lights[0].v[a] : 0.0;
        d.light_color[a] = d.has_light ? s->lights[0].color[a] : 0.0;
    }
    d.light_intensity = d.has_light ? s->lights[0].a : 0.0;
    d.ambient = s->ambient;
    d.cull_mode = s->cull_mode;
    // the first mesh whose descent array fits is staged in LDS by the trace kernels
    // small scenes (the reference's are: 1-4 entities): entity table and materials live in LDS
    d.lds_ents = (!s->ents.empty() && s->ents.size() <= (size_t)rrt::kMaxLdsEnts && s->mats.size() <= (size_t)rrt::kMaxLdsMats) ? 1 : 0;
    d.lds_mats = d.lds_ents ? (int32_t)s->mats.size() : 0;
    // Which descent arrays the trace kernels keep in LDS: meshes in order, each whole if it still fits the node
    // budget, else (a tree too large for that) the breadth-first top of its tree; at most kMaxStaged arrays.
    // RRT_MAX_LDS_NODES / RRT_LDS_TOP_NODES (tests): lower the two limits, so that small meshes take the
    // large-mesh code paths (0 / 0 = nothing staged: every descent from global memory).
    int32_t budget = rrt::kMaxLdsNodes, top_nodes = rrt::kTopNodes;
    if (const char* e = getenv("RRT_MAX_LDS_NODES")) budget = std::min<int32_t>(budget, std::max(0, atoi(e)));
    if (const char* e = getenv("RRT_LDS_TOP_NODES")) top_nodes = std::min<int32_t>(top_nodes, std::max(0, atoi(e)));
    d.n_staged = 0;
    d.lds_nodes = 0;
    std::vector<int32_t> mesh_stage(std::max<size_t>(s->meshes.size(), 1), 0);
    bool is_top = false;
    int32_t whole_left = budget, total_left = rrt::kMaxLdsNodes;
    for (size_t i = 0; i < s->meshes.size() && d.n_staged < rrt::kMaxStaged; ++i) {
        HostMesh& m = *s->meshes[i];
        const int32_t nn = m.tree.n_nodes();
        const int k = d.n_staged;
        if (nn <= whole_left) {
            d.st_src[k] = m.d_nodes.as<rrt::DNodeI>();
            d.st_cnt[k] = nn;
            is_top = false;
        } else if (top_nodes >= 3 && top_nodes <= total_left) {
            // breadth-first top: entry e is interior while both its children 2e+1, 2e+2 are entries too;
            // otherwise it links to the node of the global pre-order array the descent continues from
            std::vector<rrt::DNodeI> top((size_t)top_nodes);
            std::vector<std::pair<int32_t, int32_t>> todo{{0, 0}};      // (entry, node)
            int32_t used = 0;
            while (!todo.empty()) {
                const auto [e, node] = todo.back();
                todo.pop_back();
                used = std::max(used, e + 1);
                rrt::DNodeI& t = top[(size_t)e];
                if (m.tree.axis[(size_t)node] >= 0 && 2 * e + 2 < top_nodes) {
                    t.split = m.tree.split[(size_t)node];
                    t.axis = m.tree.axis[(size_t)node];
                    t.right = 0;
                    todo.emplace_back(2 * e + 1, m.tree.child[2 * (size_t)node]);
                    todo.emplace_back(2 * e + 2, m.tree.child[2 * (size_t)node + 1]);
                } else {
                    t.split = 0.0;
                    t.axis = rrt::kTopLink;
                    t.right = node;
                }
            }
            top.resize((size_t)used);
            HIP_TRY(m.d_top.upload(top.data(), top.size() * sizeof(rrt::DNodeI)));
            d.st_src[k] = m.d_top.as<rrt::DNodeI>();
            d.st_cnt[k] = used;
            is_top = true;
        } else {
            continue;
        }
        d.st_off[k] = d.lds_nodes;
        mesh_stage[i] = is_top ? -(d.lds_nodes + 1) : d.lds_nodes + 1;
        d.lds_nodes += d.st_cnt[k];
        if (!is_top) whole_left -= d.st_cnt[k];
        total_left -= d.st_cnt[k];
        whole_left = std::min(whole_left, total_left);
        ++d.n_staged;
    }
    HIP_TRY(s->d_mesh_stage.upload(mesh_stage.data(), mesh_stage.size() * sizeof(int32_t)));
    d.mesh_stage = s->d_mesh_stage.as<int32_t>();
    for (int f = 0; f < 6; ++f) {
        // + 4 readable bytes: the kernels fetch a texel as one unaligned dword (its 3 bytes and the next one)
        HIP_TRY(s->sky[f].d.ensure(s->sky[f].rgb.size() + 4));
        HIP_TRY(hipMemcpy(s->sky[f].d.p, s->sky[f].rgb.data(), s->sky[f].rgb.size(), hipMemcpyHostToDevice));
        d.sky[f] = s->sky[f].d.as<uint8_t>();
        d.sky_w[f] = s->sky[f].w;
        d.sky_h[f] = s->sky[f].h;
    }
    {
        unsigned char* base = nullptr;   // set-up: this device's no-alloc scratch (rrt_render_device_noalloc)
        const int e = rrt::scratch_base(s->device, &base);
        if (e != 0) return fail(RRT_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString((hipError_t)e));
    }
    HIP_TRY(s->d_dscene.upload(&s->dscene, sizeof(rrt::DScene)));
    s->dirty = false;
    return RRT_OK;
}

struct FramePlan {
    rrt::DFrame fr{};
    uint32_t levels = 0;
    uint32_t batch_spp = 0;
    uint32_t n_batches = 0;
    uint64_t cap = 0;  // paths per batch
    uint64_t valid_px = 0;
    bool want_bins = false;   // this frame fills DFrame::tile_bins from the scene
    bool want_lists = false;  // ... and the tile lists (always with the bins; a dealt frame has them even without)
    bool dealt = false;       // rrt_settings::tile_deal: tiles dealt to the ranks by cost class
};

size_t bytes_per_path(uint32_t levels) {
    return (size_t)levels * 7 * 8   // recA + recP + recR
           + 3 * 8 + 4              // samp, samp32
           + 2 * (6 * 8 + 4 + 4);   // two queues
}

// `budget`: bytes the batch's work buffers may take; `fixed_per_px`: bytes per local pixel that do not
// scale with the batch (the no-alloc entry point counts its accum array here, the allocating one does
// not: accum has always been outside rrt_set_work_budget).
// How many samples of a pixel share a wave of the primary kernel (DFrame::pack_log2): the largest power of two that
// divides the batch's sample count, at most 64 = one pixel per wave (RRT_SAMPLE_PACK=<log2> lowers the cap; 0 = a wave
// is 8x8 pixels of one sample, the round-1 map).  Rays of one pixel walk the same leaves: measured on the 2 M-triangle
// field at 32 samples a batch 2.20 -> 2.98 Grays/s, on the UFO frame at 16 spp 21.1 -> 22.0.
uint32_t sample_pack_log2(uint32_t ns) {
    static const int cap = [] {
        const char* e = getenv("RRT_SAMPLE_PACK");
        int v = e ? atoi(e) : 6;
        return v < 0 ? 0 : v > 6 ? 6 : v;
    }();
    uint32_t p = 0;
    while ((int)p < cap && ns % (2u << p) == 0) ++p;
    return p;
}

int plan_frame(const rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, FramePlan* out, uint64_t budget,
               uint64_t fixed_per_px = 0) {
    if (!cam || !st) return fail(RRT_E_INVALID, "null camera/settings");
    if (st->width == 0 || st->height == 0 || st->spp == 0) return fail(RRT_E_INVALID, "width, height and spp must be > 0");
    if ((uint64_t)st->width * st->height > 0x7fffffffull) return fail(RRT_E_INVALID, "image too large");
    if (st->max_bounce > (uint32_t)rrt::kMaxLevels - 2u) return fail(RRT_E_INVALID, "max_bounce too large (<= 16)");   // no addition: it would wrap
    if (st->tile_world == 0 || st->tile_rank >= st->tile_world) return fail(RRT_E_INVALID, "bad tile_rank/tile_world");
    rrt::DFrame& fr = out->fr;
    fr.cam = *cam;
    fr.width = st->width;
    fr.height = st->height;
    fr.spp = st->spp;
    fr.max_bounce = st->max_bounce;
    fr.seed_lo = (uint32_t)st->seed;
    fr.seed_hi = (uint32_t)(st->seed >> 32);
    fr.prune = st->prune_zero_mix ? 1u : 0u;
    fr.tile_rank = st->tile_rank;
    fr.tile_world = st->tile_world;
    fr.tiles_x = (st->width + RRT_TILE_W - 1) / RRT_TILE_W;
    fr.tiles_y = (st->height + RRT_TILE_H - 1) / RRT_TILE_H;
    const uint64_t n_tiles = (uint64_t)fr.tiles_x * fr.tiles_y;
    fr.tile_skew = rrt::tile_skew(fr.tiles_x, st->tile_world);
    fr.n_slots = (uint32_t)((n_tiles + st->tile_world - 1) / st->tile_world);
    fr.npx_local = fr.n_slots * 256u;
    // level records are written by the casts that continue: depths 0 .. max_bounce (a cast at depth max_bounce + 1
    // never does, tracer.rs:205), i.e. max_bounce + 1 levels -- round 1 allocated one more, 56 B per path unused
    out->levels = st->max_bounce + 1;
    const size_t bpp = bytes_per_path(out->levels);
    const uint64_t fixed = fixed_per_px * fr.npx_local;
    uint64_t max_spp = budget > fixed ? (budget - fixed) / ((uint64_t)bpp * fr.npx_local) : 0;
    if (st->batch_spp && (fixed_per_px == 0 || st->batch_spp < max_spp)) max_spp = st->batch_spp;
    if (max_spp == 0)
        return fail(RRT_E_NOMEM, "work budget too small for one sample per pixel: " + std::to_string(fr.npx_local) +
                                     " pixels x " + std::to_string(bpp + fixed_per_px) + " B > " + std::to_string(budget) + " B");
    // keep a path's byte offset in the sample arrays (24 * path id) in 32 bits: the kernels address them that way
    // (178.9 M paths; the 56-byte records and queue entries are addressed in 64 bits, rrt_kernels.hip: at_rec)
    max_spp = std::min<uint64_t>(max_spp, (0xffffffffull / 24) / fr.npx_local);
    if (max_spp == 0) return fail(RRT_E_NOMEM, "frame too large");
    out->n_batches = (uint32_t)((st->spp + max_spp - 1) / max_spp);
    out->batch_spp = (st->spp + out->n_batches - 1) / out->n_batches;
    if (out->n_batches > 1 && !st->batch_spp) {
        // Sample packing (sample_pack_log2) wants a batch's sample count to be a multiple of a large power of two
        // (height field: 8 samples of a pixel a wave 2.70 Grays/s, 16: 3.00, 32: 3.25, 64: 3.71), and every batch has a
        // fixed cost: the largest unit of 64, 32, ... whose multiples under the budget need at most half as many
        // batches again as the fewest possible; the remainder goes last.
        const uint64_t n_min = out->n_batches;
        for (uint64_t unit = 64; unit >= 1; unit >>= 1) {
            const uint64_t b = max_spp / unit * unit;
            if (b == 0) continue;
            const uint64_t n = (st->spp + b - 1) / b;
            if (2 * n <= 3 * n_min || unit == 1) {
                out->batch_spp = (uint32_t)b;
                out->n_batches = (uint32_t)n;
                break;
            }
        }
    }
    out->cap = (uint64_t)out->batch_spp * fr.npx_local;
    // pixels this rank really owns
    uint64_t valid = 0;
    for (uint32_t slot = 0; slot < fr.n_slots; ++slot) {
        const uint64_t tile = (uint64_t)slot * st->tile_world + st->tile_rank;
        if (tile >= n_tiles) continue;
        const uint32_t ty = (uint32_t)(tile / fr.tiles_x), tx = (uint32_t)((tile % fr.tiles_x + (uint64_t)fr.tile_skew * ty) % fr.tiles_x);
        const uint32_t w = std::min<uint32_t>(RRT_TILE_W, st->width - tx * RRT_TILE_W);
        const uint32_t h = std::min<uint32_t>(RRT_TILE_H, st->height - ty * RRT_TILE_H);
        valid += (uint64_t)w * h;
    }
    out->valid_px = valid;
    return RRT_OK;
}

uint64_t cover_stride(uint32_t width) { return ((uint64_t)(width + rrt::kCoverCell - 1) / rrt::kCoverCell + 7) / 8 * 8; }
uint64_t cover_bytes(const rrt::DFrame& fr) { return (uint64_t)fr.n_cover * fr.cover_bytes; }
// What every frame starts from zero sits in ONE block, so that it is one memset (each is a ~4 us kernel in the frame's
// launch chain): the statistics, the queue counters (zeroed again in front of every further batch) and the coverage masks.
constexpr uint64_t kSegBytes = 2ull * rrt::kMaxSegs * sizeof(rrt::uint2_);      // DWork::seg_n
constexpr uint64_t kStatsBytes = 8 * 8, kQCountBytes = 2 * (rrt::kMaxLevels + 1) * 4, kListNBytes = 16;
uint64_t zero_bytes(const rrt::DFrame& fr) { return 256 + cover_bytes(fr); }
static_assert(kStatsBytes + kQCountBytes + kListNBytes <= 256, "statistics, queue counters and list counts share the block's first 256 bytes");
void zero_layout(unsigned char* base, FramePlan& pl, rrt::DWork* wk) {
    wk->stats = (unsigned long long*)base;
    wk->q_count = (unsigned long long*)(base + kStatsBytes);
    pl.fr.list_n = pl.want_lists ? (const uint32_t*)(base + kStatsBytes + kQCountBytes) : nullptr;
    pl.fr.cover = pl.fr.n_cover ? base + 256 : nullptr;
}
uint64_t bins_bytes(const rrt::DFrame& fr, bool want) { return want ? (uint64_t)fr.tiles_x * fr.tiles_y * 9 : 0; }     // bins, then a cost byte per tile
uint64_t lists_bytes(const rrt::DFrame& fr, bool want) { return want ? (uint64_t)fr.n_slots * 8 : 0; }     // order, then slot_tile
void lists_layout(unsigned char* base, FramePlan& pl) {
    pl.fr.order = pl.want_lists ? (const int32_t*)base : nullptr;
    pl.fr.slot_tile = pl.want_lists && pl.dealt ? (const int32_t*)base + pl.fr.n_slots : nullptr;
}

int ensure_work(WorkSet* s, FramePlan& pl, rrt::DWork* wk) {
    const uint64_t P = pl.cap;
    HIP_TRY(s->w_zero.ensure(zero_bytes(pl.fr)));
    zero_layout((unsigned char*)s->w_zero.p, pl, wk);
    if (pl.want_lists) {
        HIP_TRY(s->w_bins.ensure(bins_bytes(pl.fr, true)));
        HIP_TRY(s->w_lists.ensure(lists_bytes(pl.fr, true)));
    }
    pl.fr.tile_bins = pl.want_lists ? s->w_bins.as<uint64_t>() : nullptr;
    pl.fr.tile_cost = pl.want_lists ? (const uint8_t*)(pl.fr.tile_bins + (size_t)pl.fr.tiles_x * pl.fr.tiles_y) : nullptr;
    lists_layout((unsigned char*)s->w_lists.p, pl);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(s->w_q[i].ensure(P * rrt::kRecDoubles * 8));
        wk->q[i] = s->w_q[i].as<double>();
    }
    HIP_TRY(s->w_rec.ensure(P * rrt::kRecDoubles * 8 * pl.levels));
    HIP_TRY(s->w_samp.ensure(P * 3 * 8));
    HIP_TRY(s->w_samp32.ensure(P * 4));
    HIP_TRY(s->w_accum.ensure((size_t)pl.fr.npx_local * 3 * 8));
    HIP_TRY(s->w_seg.ensure(kSegBytes));
    wk->seg_n = s->w_seg.as<rrt::uint2_>();

    wk->rec = s->w_rec.as<double>();
    wk->samp = s->w_samp.as<double>();
    wk->samp32 = s->w_samp32.as<uint32_t>();
    wk->accum = s->w_accum.as<double>();
    wk->cap = P;
    return RRT_OK;
}

// The same buffers carved out of one caller- or library-provided block (the no-alloc entry point:
// the per-device static scratch array of rrt_kernels.hip).  No allocation; fails if the block is too small.
constexpr uint64_t kCarveAlign = 256;
uint64_t carve_bytes(const FramePlan& pl) {
    const uint64_t P = pl.cap;
    const uint64_t parts[] = {zero_bytes(pl.fr), P * rrt::kRecDoubles * 8, P * rrt::kRecDoubles * 8,
                              P * rrt::kRecDoubles * 8 * pl.levels, P * 3 * 8, P * 4, (uint64_t)pl.fr.npx_local * 3 * 8,
                              bins_bytes(pl.fr, pl.want_lists), lists_bytes(pl.fr, pl.want_lists), kSegBytes};
    uint64_t total = 0;
    for (uint64_t b : parts) total += (b + kCarveAlign - 1) / kCarveAlign * kCarveAlign;
    return total;
}
int carve_work(unsigned char* base, uint64_t bytes, FramePlan& pl, rrt::DWork* wk) {
    if (carve_bytes(pl) > bytes)
        return fail(RRT_E_NOMEM, "frame needs " + std::to_string(carve_bytes(pl)) + " B of scratch, the block has " + std::to_string(bytes));
    const uint64_t P = pl.cap;
    uint64_t off = 0;
    const auto take = [&](uint64_t b) {
        unsigned char* p = base + off;
        off += (b + kCarveAlign - 1) / kCarveAlign * kCarveAlign;
        return p;
    };
    zero_layout(take(zero_bytes(pl.fr)), pl, wk);
    wk->q[0] = (double*)take(P * rrt::kRecDoubles * 8);
    wk->q[1] = (double*)take(P * rrt::kRecDoubles * 8);
    wk->rec = (double*)take(P * rrt::kRecDoubles * 8 * pl.levels);
    wk->samp = (double*)take(P * 3 * 8);
    wk->samp32 = (uint32_t*)take(P * 4);
    wk->accum = (double*)take((uint64_t)pl.fr.npx_local * 3 * 8);
    pl.fr.tile_bins = pl.want_lists ? (const uint64_t*)take(bins_bytes(pl.fr, true)) : nullptr;
    pl.fr.tile_cost = pl.want_lists ? (const uint8_t*)(pl.fr.tile_bins + (size_t)pl.fr.tiles_x * pl.fr.tiles_y) : nullptr;
    lists_layout(take(lists_bytes(pl.fr, pl.want_lists)), pl);
    wk->seg_n = (rrt::uint2_*)take(kSegBytes);
    wk->cap = P;
    return RRT_OK;
}

int timed_slot(WorkSet* s, TimedLaunch** out) {
    if (s->timed_used == s->timed.size()) {
        TimedLaunch t;
        HIP_TRY(hipEventCreate(&t.a));
        HIP_TRY(hipEventCreate(&t.b));
        s->timed.push_back(t);
    }
    *out = &s->timed[s->timed_used++];
    return RRT_OK;
}

// NOALLOC (rrt_render_device_noalloc): nothing here may allocate -- no hipMalloc/hipFree, no event
// creation, no growth of a host container.  The work buffers are carved out of the device's static
// scratch array, the scene must be committed already, and the frame is not timed.
// rrt_render into PINNED host memory: the last accumulate pass is issued as bands of tile rows, and each band's RGB8
// rows are copied to the host (on copy_stream, behind an event) while the next band is resolved.
struct HostBands {
    uint8_t* rgb = nullptr;          // pinned host destination, row-major
    hipStream_t copy_stream = nullptr;
    hipEvent_t* ev = nullptr;
    uint32_t n_bands = 0;
};

template <bool NOALLOC>
int render_device(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, void* d_rgb8, void* d_ids, void* d_t,
                  void* d_color, void* stream_v, const HostBands* bands = nullptr) {
    if (!s) return fail(RRT_E_INVALID, "null scene");
    FramePlan pl;
    // Primary-ray coverage masks (kMaxCover in rrt_device.h): one bit per pixel of the whole image per masked mesh,
    // refilled every frame (they depend on the camera).  Off with the cull (cull_mode 0 is "every test as written")
    // and for a lens the projection does not cover.
    // Tile bins (DFrame::tile_bins) go with them: which entities of a small scene a tile's primary rays can meet.
    uint64_t mask_bytes = 0;
    uint32_t n_cover = 0;
    bool want_bins = false, want_lists = false;
    const auto count_masks = [&]() {      // needs the committed scene
        n_cover = 0;
        mask_bytes = 0;
        want_bins = false;
        want_lists = false;
        static const bool bins_on = [] { const char* e = getenv("RRT_BINS"); return !e || atoi(e) != 0; }();   // A/B knob
        if (cam && st && st->width && st->height && !s->dirty && s->cull_mode != 0 && cam->lens_z > 0.0 &&
            cam->lens_w > 0.0 && cam->lens_h > 0.0 && std::isfinite(cam->lens_z + cam->lens_w + cam->lens_h + cam->eye[0] + cam->eye[1] + cam->eye[2])) {
            n_cover = (uint32_t)s->cover.size();
            // (a very tall, narrow image: the mask's byte count is a 32-bit field of DFrame -- no masks then)
            if (cover_stride(st->width) * st->height + 8 > 0xffffffffull) n_cover = 0;
            mask_bytes = (uint64_t)n_cover * (cover_stride(st->width) * st->height + 8) + kCarveAlign;
            if (NOALLOC && mask_bytes > rrt::kScratchBytes / 8) { n_cover = 0; mask_bytes = 0; }     // a huge image: leave the scratch to the paths
            // bins are worth their launch once the frame has some paths (a 640x480 frame at 1 spp is four launches long)
            want_bins = bins_on && s->dscene.lds_ents != 0 && s->ents.size() >= 2 && (uint64_t)st->width * st->height * st->spp >= (1u << 20);
        }
        // a dealt frame (rrt_settings::tile_deal) always carries the tile lists: they hold its slot -> tile table
        want_lists = want_bins || (st && st->tile_deal != 0 && st->tile_world > 1);
        if (want_lists && st && st->width && st->height) {
            const uint64_t n_img = (uint64_t)((st->width + 31) / 32) * ((st->height + 7) / 8);
            mask_bytes += n_img * 9 + (n_img / (st->tile_world ? st->tile_world : 1) + 1) * 8 + 2 * kCarveAlign;
        }
    };
    if (NOALLOC) count_masks();           // (an uncommitted scene is refused below)
    // the scratch block minus the two small headers; accum (24 B per pixel) is part of it
    int rc = NOALLOC ? plan_frame(s, cam, st, &pl, rrt::kScratchBytes - 16 * kCarveAlign - mask_bytes - kSegBytes, 3 * 8) : plan_frame(s, cam, st, &pl, s->work_budget);
    if (rc) {
        if (NOALLOC && rc == RRT_E_NOMEM)
            g_err = "rrt_render_device_noalloc: frame too large for the " + std::to_string(rrt::kScratchBytes >> 20) +
                    " MiB static scratch (" + g_err + "); shard it (tile_world) or use rrt_render_device";
        return rc;
    }
    if (st->tile_world > 1 && (d_ids || d_t || d_color))
        return fail(RRT_E_INVALID, "AOV / f64 colour outputs need tile_world == 1");
    pl.fr.cover_stride = (uint32_t)cover_stride(st->width);
    pl.fr.cover_bytes = (uint32_t)std::min<uint64_t>(cover_stride(st->width) * st->height + 8, 0xffffffffull);   // n_cover is 0 when this saturates
    rrt::DWork wk{};
    hipStream_t stream = (hipStream_t)stream_v;
    WorkSet* ws = &s->noalloc_set;
    if (NOALLOC) {
        if (s->dirty) return fail(RRT_E_STATE, "rrt_render_device_noalloc needs a committed scene (rrt_scene_commit uploads, i.e. allocates)");
        HIP_TRY(hipSetDevice(s->device));
        unsigned char* base = nullptr;
        const int e = rrt::scratch_base(s->device, &base);
        if (e != 0 || !base) return fail(RRT_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString((hipError_t)e));
        pl.fr.n_cover = n_cover;
        pl.want_bins = want_bins;
        pl.want_lists = want_lists;
        pl.dealt = st->tile_deal != 0 && st->tile_world > 1;
        rc = carve_work(base, rrt::kScratchBytes, pl, &wk);
        if (rc) return rc;
    } else {
        rc = commit(s);
        if (rc) return rc;
        count_masks();
        pl.fr.n_cover = n_cover;
        pl.want_bins = want_bins;
        pl.want_lists = want_lists;
        pl.dealt = st->tile_deal != 0 && st->tile_world > 1;
        rc = work_set_for(s, stream, &ws);
        if (rc) return rc;
        rc = ensure_work(ws, pl, &wk);
        if (rc) return rc;
    }
    const bool timed = !NOALLOC && s->timing >= 1, timed_kernels = !NOALLOC && s->timing >= 2;
    if (timed && !ws->ev_begin) {
        HIP_TRY(hipEventCreate(&ws->ev_begin));
        HIP_TRY(hipEventCreate(&ws->ev_end));
    }
    ws->timed_used = 0;
    ws->frame_timed = timed;
    ws->stats_src = wk.stats;
    if (timed) HIP_TRY(hipEventRecord(ws->ev_begin, stream));
    HIP_TRY(hipMemsetAsync(wk.stats, 0, zero_bytes(pl.fr), stream));      // statistics, queue counters, coverage masks
    rrt::DFrame fr = pl.fr;
    if (fr.n_cover) {
        for (uint32_t k = 0; k < fr.n_cover; ++k)
            rrt::launch_cover(s->dmeshes[(size_t)s->cover[k].first], s->cover[k].second, fr, k, stream);
    }
    fr.chunks_per_block = rrt::primary_chunks_per_block();
    if (pl.want_bins) rrt::launch_bins(s->dscene, fr, stream);
    else if (pl.want_lists) HIP_TRY(hipMemsetAsync(const_cast<uint64_t*>(fr.tile_bins), 0xff, bins_bytes(fr, true), stream));   // dealt, no bins: every tile "can meet everything"
    if (pl.want_lists) {
        // cull mode 2 (verify): every tile goes through the wavefront kernels, which check the bins
        rrt::launch_lists(fr, pl.dealt, s->cull_mode != 1 || !pl.want_bins, stream);
    }
    // The sky-only tiles run beside the wavefront kernels on a stream of their own (RRT_SKY_FORK=0, and the entry
    // point that allocates nothing: in front of them on the same stream).
    static const bool fork_on = [] { const char* e = getenv("RRT_SKY_FORK"); return !e || atoi(e) != 0; }();
    const bool fork = pl.want_lists && !NOALLOC && fork_on && s->sky_stream != 0;
    if (fork && !ws->side) {
        HIP_TRY(hipStreamCreateWithFlags(&ws->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ws->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ws->ev_join, hipEventDisableTiming));
    }
    ws->spp = st->spp;
    ws->dealt = pl.dealt;
    ws->lists = pl.want_lists;
    for (uint32_t b = 0; b < pl.n_batches; ++b) {
        fr.s0 = b * pl.batch_spp;
        fr.ns = std::min(pl.batch_spp, st->spp - fr.s0);
        fr.pack_log2 = sample_pack_log2(fr.ns);
        fr.seg_grid = rrt::primary_grid(fr);
        if (b > 0) HIP_TRY(hipMemsetAsync(wk.q_count, 0, kQCountBytes, stream));
        TimedLaunch* tl = nullptr;
        // Where in the batch's chain the sky kernel forks off: BEHIND the primary kernel.  The sky tiles are off the
        // frame's critical path (primary kernel -> bounce levels -> accumulate) and the bounce levels leave most of the
        // chip idle (the first one in its tail, the thin ones throughout): forked in front of the primary kernel
        // (round 3's first version) the sky kernel shared the chip with it from the start and was done long before
        // those gaps came.  One frame at a time, 1080p: 16 spp 1.37 -> 1.24 ms (primary kernel 775 -> 622 us, first
        // bounce level 320 -> 343, the sky kernel inside it), 64 spp 3.84 -> 3.56 ms, a rank's eighth 0.81 -> 0.75;
        // four frames in flight 3.148 -> 3.120 ms, the eighth 0.486 -> 0.494.  Behind the first bounce level
        // (RRT_SKY_AFTER=2) the sky kernel is longer than what is left of the chain: 3.77 ms, in flight 3.24.
        // RRT_SKY_AFTER=0: in front of the primary kernel.
        static const int sky_after_env = [] { const char* e = getenv("RRT_SKY_AFTER"); return e ? atoi(e) : 1; }();
        const int sky_after = fork ? sky_after_env : 0;
        auto sky = [&]() -> int {
            // (the sky kernel writes sample words the previous batch's accumulate pass reads: it starts behind it)
            hipStream_t ss = fork ? ws->side : stream;
            TimedLaunch* ts = nullptr;
            if (fork) {
                HIP_TRY(hipEventRecord(ws->ev_fork, stream));
                HIP_TRY(hipStreamWaitEvent(ss, ws->ev_fork, 0));
            }
            if (timed_kernels) {
                const int trc = timed_slot(ws, &ts);
                if (trc) return trc;
                ts->kind = 2;
                ts->launches = 1;
                HIP_TRY(hipEventRecord(ts->a, ss));
            }
            rrt::launch_sky(s->dscene, wk, fr, (int32_t*)d_ids, (double*)d_t, ss);
            if (timed_kernels) HIP_TRY(hipEventRecord(ts->b, ss));
            if (fork) HIP_TRY(hipEventRecord(ws->ev_join, ss));
            return RRT_OK;
        };
        if (pl.want_lists && sky_after == 0) {
            rc = sky();
            if (rc) return rc;
        }
        if (timed_kernels) {
            rc = timed_slot(ws, &tl);
            if (rc) return rc;
            tl->kind = 0;
            tl->launches = 1;
            HIP_TRY(hipEventRecord(tl->a, stream));
        }
        rrt::launch_primary(s->dscene, s->d_dscene.as<rrt::DScene>(), wk, fr, (int32_t*)d_ids, (double*)d_t, stream);
        if (timed_kernels) HIP_TRY(hipEventRecord(tl->b, stream));
        if (pl.want_lists && sky_after == 1) {       // (takes a timing slot of its own: `tl` does not outlive this)
            rc = sky();
            if (rc) return rc;
        }
        if (timed_kernels) {
            // one event pair around the batch's bounce levels together (each hipEventRecord between two
            // launches costs ~3 us of idle GPU: ten events a frame were 1.5 % of it)
            rc = timed_slot(ws, &tl);
            if (rc) return rc;
            tl->kind = 1;
            tl->launches = st->max_bounce + 1;
            HIP_TRY(hipEventRecord(tl->a, stream));
        }
        for (uint32_t bounce = 1; bounce <= st->max_bounce + 1; ++bounce) {
            rrt::launch_bounce(s->dscene, s->d_dscene.as<rrt::DScene>(), wk, fr, bounce, stream);
            if (pl.want_lists && sky_after >= 2 && bounce == 1) {
                const size_t slot = timed_kernels ? (size_t)(tl - ws->timed.data()) : 0;
                rc = sky();
                if (rc) return rc;
                if (timed_kernels) tl = &ws->timed[slot];       // sky() may have grown the vector
            }
        }
        if (timed_kernels) HIP_TRY(hipEventRecord(tl->b, stream));
        if (fork) HIP_TRY(hipStreamWaitEvent(stream, ws->ev_join, 0));
        const bool last = b + 1 == pl.n_batches;
        if (last && bands && bands->n_bands > 1 && d_rgb8 && st->tile_world == 1) {
            for (uint32_t k = 0; k < bands->n_bands; ++k) {
                const uint32_t ty0 = (uint32_t)((uint64_t)fr.tiles_y * k / bands->n_bands), ty1 = (uint32_t)((uint64_t)fr.tiles_y * (k + 1) / bands->n_bands);
                if (ty1 == ty0) continue;
                rrt::launch_accumulate(wk, fr, true, (uint8_t*)d_rgb8, (double*)d_color, stream, ty0 * fr.tiles_x, (ty1 - ty0) * fr.tiles_x);
                HIP_TRY(hipEventRecord(bands->ev[k], stream));
                HIP_TRY(hipStreamWaitEvent(bands->copy_stream, bands->ev[k], 0));
                const size_t r0 = (size_t)ty0 * RRT_TILE_H, r1 = std::min<size_t>((size_t)ty1 * RRT_TILE_H, st->height);
                HIP_TRY(hipMemcpyAsync(bands->rgb + r0 * st->width * 3, (const uint8_t*)d_rgb8 + r0 * st->width * 3, (r1 - r0) * st->width * 3,
                                       hipMemcpyDeviceToHost, bands->copy_stream));
            }
        } else {
            rrt::launch_accumulate(wk, fr, last, (uint8_t*)d_rgb8, (double*)d_color, stream);   // the last one resolves
        }
    }
    if (pl.dealt && d_rgb8)       // the slot -> tile table travels behind the pixels (rrt_assemble_tiles_dealt reads it)
        HIP_TRY(hipMemcpyAsync((uint8_t*)d_rgb8 + (size_t)fr.n_slots * RRT_TILE_W * RRT_TILE_H * 3, fr.slot_tile, (size_t)fr.n_slots * 4,
                               hipMemcpyDeviceToDevice, stream));
    if (timed) HIP_TRY(hipEventRecord(ws->ev_end, stream));
    HIP_TRY(hipGetLastError());
    s->last = ws;
    ws->stats = rrt_stats{};
    ws->stats.paths = pl.valid_px * st->spp;
    ws->stats.primary_rays = ws->stats.paths;
    ws->stats.n_batches = pl.n_batches;
    return RRT_OK;
}

}  // namespace

extern "C" {

const char* rrt_last_error(void) { return g_err.c_str(); }
const char* rrt_version(void) { return "rustraytracer_amd 0.1 (gfx950)"; }

int rrt_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    unsigned char* base = nullptr;   // set-up: resolve this device's copy of the no-alloc scratch array once
    const int e = rrt::scratch_base(device, &base);
    if (e != 0) return fail(RRT_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString((hipError_t)e));
    return RRT_OK;
}

int rrt_device_image_support(void) {
    int v = 0;
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeImageSupport, g_device));
    return v;
}

void rrt_default_camera(rrt_camera* cam) {
    cam->eye[0] = cam->eye[1] = cam->eye[2] = 0.0;
    cam->lens_z = 75.0;
    cam->lens_w = 160.0;
    cam->lens_h = 90.0;
}

void rrt_default_settings(rrt_settings* st, uint32_t w, uint32_t h) {
    std::memset(st, 0, sizeof(*st));
    st->width = w;
    st->height = h;
    st->spp = 1;
    st->max_bounce = 2;
    st->tile_world = 1;
}

uint64_t rrt_tile_buffer_bytes(uint32_t width, uint32_t height, uint32_t tile_world, uint32_t tile_deal) {
    const uint64_t slots = (uint64_t)rrt_tile_slots(width, height, tile_world);
    return slots * RRT_TILE_W * RRT_TILE_H * 3 + (tile_deal && tile_world > 1 ? slots * 4 : 0);
}

rrt_scene* rrt_scene_create(void) {
    rrt_scene* s = new (std::nothrow) rrt_scene();
    if (s) s->device = g_device;
    return s;
}

void rrt_scene_destroy(rrt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    delete s;
}

int rrt_add_material_ext(rrt_scene* s, const double color[3], double albedo, double lambert_mix, double lobe_mix,
                         double phong_mix, int32_t alpha, int32_t lobe, double ior) {
    if (lobe < 0 || lobe > 2) return fail(RRT_E_INVALID, "lobe must be 0 (reflection), 1 (refraction) or 2 (diffuse)");
    if (lobe == 1 && !(ior > 0.0)) return fail(RRT_E_INVALID, "index of refraction must be > 0");
    const int id = rrt_add_material(s, color, albedo, lambert_mix, lobe_mix, phong_mix, alpha);
    if (id < 0) return id;
    s->mats[(size_t)id].lobe = lobe;
    s->mats[(size_t)id].ior = ior;
    return id;
}

int rrt_add_material(rrt_scene* s, const double color[3], double albedo, double lambert_mix, double reflective_mix,
                     double phong_mix, int32_t alpha) try {
    if (!s || !color) return fail(RRT_E_INVALID, "null argument");
    rrt::DMaterial m{};
    m.lobe = 0;
    m.ior = 1.0;
    for (int a = 0; a < 3; ++a) m.color[a] = color[a];
    m.albedo = albedo;
    m.mixL = lambert_mix;
    m.mixR = reflective_mix;
    m.mixP = phong_mix;
    m.alpha = alpha;
    s->mats.push_back(m);
    s->dirty = true;
    return (int)s->mats.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_add_sphere(rrt_scene* s, const double center[3], double radius, int material) try {
    if (!s || !center) return fail(RRT_E_INVALID, "null argument");
    if (material < 0 || material >= (int)s->mats.size()) return fail(RRT_E_INVALID, "unknown material id");
    rrt::DSphere sp{};
    for (int a = 0; a < 3; ++a) sp.c[a] = center[a];
    sp.r2 = radius * radius;  // radius.powi(2), sphere.rs:21
    sp.material = material;
    s->spheres.push_back(sp);
    s->sphere_radius.push_back(radius);
    s->ents.push_back(rrt::DEntity{0, (int32_t)s->spheres.size() - 1});
    s->dirty = true;
    return (int)s->ents.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_add_mesh(rrt_scene* s, const double* verts, const double* normals, int64_t n, int material) try {
    if (!s || !verts || !normals) return fail(RRT_E_INVALID, "null argument");
    if (n <= 0 || n > 0x3fffffff) return fail(RRT_E_INVALID, "triangle count out of range");
    if (material < 0 || material >= (int)s->mats.size()) return fail(RRT_E_INVALID, "unknown material id");
    auto m = std::make_unique<HostMesh>();
    m->material = material;
    m->tris.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const double* v = verts + 9 * i;
        rrt::DTri& t = m->tris[(size_t)i];
        for (int a = 0; a < 3; ++a) {
            t.v0[a] = v[a];
            t.e1[a] = v[3 + a] - v[a];  // triangle.rs:30
            t.e2[a] = v[6 + a] - v[a];  // triangle.rs:31
        }
    }
    m->verts.assign(verts, verts + 9 * n);
    m->normals.assign(normals, normals + 3 * n);
    m->tree = rrt::build_kd_tree(verts, n);
    s->meshes.push_back(std::move(m));
    s->ents.push_back(rrt::DEntity{1, (int32_t)s->meshes.size() - 1});
    s->dirty = true;
    return (int)s->ents.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_add_triangle(rrt_scene* s, const double v0[3], const double v1[3], const double v2[3], const double normal[3], int material) try {
    if (!s || !v0 || !v1 || !v2 || !normal) return fail(RRT_E_INVALID, "null argument");
    if (material < 0 || material >= (int)s->mats.size()) return fail(RRT_E_INVALID, "unknown material id");
    rrt::DTriEnt t{};
    for (int a = 0; a < 3; ++a) {
        t.v0[a] = v0[a];
        t.e1[a] = v1[a] - v0[a];   // triangle.rs:30
        t.e2[a] = v2[a] - v0[a];   // triangle.rs:31
        t.normal[a] = normal[a];
    }
    t.material = material;
    s->tri_ents.push_back(t);
    s->tri_verts.insert(s->tri_verts.end(), {v0[0], v0[1], v0[2], v1[0], v1[1], v1[2], v2[0], v2[1], v2[2]});
    s->ents.push_back(rrt::DEntity{3, (int32_t)s->tri_ents.size() - 1});
    s->dirty = true;
    return (int)s->ents.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_add_group(rrt_scene* s, const int32_t* members, int32_t n) try {
    if (!s || !members) return fail(RRT_E_INVALID, "null argument");
    if (n <= 0) return fail(RRT_E_INVALID, "a group needs at least one member");
    auto g = std::make_unique<HostGroup>();
    std::vector<double> mn((size_t)n * 3), mx((size_t)n * 3);
    for (int32_t i = 0; i < n; ++i) {
        const int32_t e = members[i];
        if (e < 0 || e >= (int32_t)s->ents.size()) return fail(RRT_E_INVALID, "unknown entity id in group");
        const rrt::DEntity ent = s->ents[(size_t)e];
        if (ent.kind != 0 && ent.kind != 1 && ent.kind != 2 && ent.kind != 3)
            return fail(RRT_E_INVALID, "group members must be top-level spheres, meshes, triangles or groups (not members of another group)");
        if (ent.kind == 2)     // Scene<Scene<T>> (scene.rs:32): one level of nesting
            for (int32_t m2 : s->groups[(size_t)ent.index]->members)
                if ((s->ents[(size_t)m2].kind & ~rrt::kEntityMember) == 2)
                    return fail(RRT_E_INVALID, "groups nest one level deep (a member group may not hold groups itself)");
        for (int32_t j = 0; j < i; ++j)
            if (members[j] == e) return fail(RRT_E_INVALID, "entity listed twice in group");
        if (ent.kind == 0) {   // Sphere::bounding_box, sphere.rs:49-54
            const rrt::DSphere& sp = s->spheres[(size_t)ent.index];
            const double r = s->sphere_radius[(size_t)ent.index];
            for (int a = 0; a < 3; ++a) {
                mn[3 * (size_t)i + a] = sp.c[a] - r;
                mx[3 * (size_t)i + a] = sp.c[a] + r;
            }
        } else if (ent.kind == 2) {   // Scene::bounding_box = aa_bb = from_entities(models), scene.rs:38-47,60-62: its tree's root box
            const rrt::KdTreeFlat& T = s->groups[(size_t)ent.index]->tree;
            for (int a = 0; a < 3; ++a) {
                mn[3 * (size_t)i + a] = T.boxes[(size_t)a];
                mx[3 * (size_t)i + a] = T.boxes[3 + (size_t)a];
            }
        } else if (ent.kind == 3) {   // Triangle::bounding_box, triangle.rs:60-65
            const double* p = &s->tri_verts[9 * (size_t)ent.index];
            for (int a = 0; a < 3; ++a) {
                mn[3 * (size_t)i + a] = std::fmin(p[a], std::fmin(p[3 + a], p[6 + a]));
                mx[3 * (size_t)i + a] = std::fmax(p[a], std::fmax(p[3 + a], p[6 + a]));
            }
        } else {               // Model::bounding_box = from_entities(triangles), model.rs:51,68-70
            const rrt::KdTreeFlat& T = s->meshes[(size_t)ent.index]->tree;
            for (int a = 0; a < 3; ++a) {
                mn[3 * (size_t)i + a] = T.boxes[(size_t)a];
                mx[3 * (size_t)i + a] = T.boxes[3 + (size_t)a];
            }
        }
        g->members.push_back(e);
    }
    g->tree = rrt::build_kd_tree_boxes(mn, mx, n);     // KDTree::new(models), scene.rs:43
    for (int32_t e : g->members) s->ents[(size_t)e].kind |= rrt::kEntityMember;
    s->groups.push_back(std::move(g));
    s->ents.push_back(rrt::DEntity{2, (int32_t)s->groups.size() - 1});
    s->dirty = true;
    return (int)s->ents.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_set_sky_face(rrt_scene* s, int face, const uint8_t* rgb8, uint32_t w, uint32_t h) try {
    if (!s || !rgb8) return fail(RRT_E_INVALID, "null argument");
    if (face < 0 || face > 5 || w == 0 || h == 0) return fail(RRT_E_INVALID, "bad face index or size");
    s->sky[face].rgb.assign(rgb8, rgb8 + (size_t)w * h * 3);
    s->sky[face].w = w;
    s->sky[face].h = h;
    s->dirty = true;
    return RRT_OK;
} catch (...) {
    return on_exception();
}

int rrt_add_directional_light(rrt_scene* s, const double direction[3], const double color[3], double intensity) try {
    if (!s || !direction || !color) return fail(RRT_E_INVALID, "null argument");
    if (s->lights.size() >= (size_t)rrt::kMaxLights) return fail(RRT_E_INVALID, "too many lights (<= 32)");
    // InnerSpace::normalize, lighting.rs:26 (v * (1/|v|))
    const double m2 = direction[0] * direction[0] + direction[1] * direction[1] + direction[2] * direction[2];
    const double inv = 1.0 / std::sqrt(m2);
    rrt::DLight L{};
    L.kind = 0;
    for (int a = 0; a < 3; ++a) {
        L.v[a] = direction[a] * inv;
        L.color[a] = color[a];
    }
    L.a = intensity;
    s->lights.push_back(L);
    s->dirty = true;
    return (int)s->lights.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_set_directional_light(rrt_scene* s, const double direction[3], const double color[3], double intensity) {
    if (!s || !direction || !color) return fail(RRT_E_INVALID, "null argument");
    s->lights.clear();       // world.light_sources = [this light]: what new_empty_world leaves (tracer.rs:55-65)
    const int rc = rrt_add_directional_light(s, direction, color, intensity);
    return rc < 0 ? rc : RRT_OK;
}

int rrt_add_point_light(rrt_scene* s, const double position[3], const double color[3], double brightness,
                        double attenuation) try {
    if (!s || !position || !color) return fail(RRT_E_INVALID, "null argument");
    size_t n_point = 0;
    for (const rrt::DLight& l : s->lights) n_point += l.kind == 1;
    if (n_point >= (size_t)rrt::kMaxPointLights) return fail(RRT_E_INVALID, "too many point lights (<= 16)");
    if (s->lights.size() >= (size_t)rrt::kMaxLights) return fail(RRT_E_INVALID, "too many lights (<= 32)");
    rrt::DLight L{};
    L.kind = 1;
    for (int a = 0; a < 3; ++a) {
        L.v[a] = position[a];
        L.color[a] = color[a];
    }
    L.a = brightness;
    L.b = attenuation;
    s->lights.push_back(L);
    s->dirty = true;
    return (int)s->lights.size() - 1;
} catch (...) {
    return on_exception();
}

int rrt_set_shadow_mode(rrt_scene* s, int mode) {
    if (!s || mode < 0 || mode > 1) return fail(RRT_E_INVALID, "shadow mode must be 0 (as the reference wrote it) or 1 (corrected)");
    s->shadow_mode = mode;
    s->dirty = true;
    return RRT_OK;
}

namespace {
// KDTree::translate_nodes (kdtree.rs:35-42): split planes move; boxes only with the refit extension.
void translate_tree(rrt::KdTreeFlat& T, const double v[3], bool refit) {
    for (size_t k = 0; k < T.axis.size(); ++k) {
        if (T.axis[k] >= 0) T.split[k] += v[T.axis[k]];
        if (refit)
            for (int a = 0; a < 3; ++a) {
                T.boxes[6 * k + a] += v[a];
                T.boxes[6 * k + 3 + a] += v[a];
            }
    }
}
void translate_entity(rrt_scene* s, int e, const double v[3], bool refit) {
    const rrt::DEntity ent = s->ents[(size_t)e];
    const int kind = ent.kind & ~rrt::kEntityMember;
    if (kind == 0) {                                   // sphere.rs:68-70
        for (int a = 0; a < 3; ++a) s->spheres[(size_t)ent.index].c[a] += v[a];
    } else if (kind == 1) {                            // model.rs:80-86 -> triangle.rs:79-83 + translate_nodes
        HostMesh& m = *s->meshes[(size_t)ent.index];
        for (size_t i = 0; i < m.tris.size(); ++i) {
            double* p = &m.verts[9 * i];
            for (int k = 0; k < 9; ++k) p[k] += v[k % 3];
            rrt::DTri& t = m.tris[i];
            for (int a = 0; a < 3; ++a) {
                t.v0[a] = p[a];
                t.e1[a] = p[3 + a] - p[a];             // triangle.rs:30-31 on the moved vertices
                t.e2[a] = p[6 + a] - p[a];
            }
        }
        translate_tree(m.tree, v, refit);
        if (!refit) m.stale_boxes = true;
    } else if (kind == 3) {                            // triangle.rs:79-83
        double* p = &s->tri_verts[9 * (size_t)ent.index];
        for (int k = 0; k < 9; ++k) p[k] += v[k % 3];
        rrt::DTriEnt& t = s->tri_ents[(size_t)ent.index];
        for (int a = 0; a < 3; ++a) {
            t.v0[a] = p[a];
            t.e1[a] = p[3 + a] - p[a];
            t.e2[a] = p[6 + a] - p[a];
        }
    } else {                                           // scene.rs:68-74
        HostGroup& g = *s->groups[(size_t)ent.index];
        for (int32_t mem : g.members) translate_entity(s, mem, v, refit);
        translate_tree(g.tree, v, refit);
        if (!refit) g.stale_boxes = true;
    }
}
}  // namespace

int rrt_translate_entity(rrt_scene* s, int entity, const double delta[3], int refit) try {
    if (!s || !delta) return fail(RRT_E_INVALID, "null argument");
    if (entity < 0 || entity >= (int)s->ents.size()) return fail(RRT_E_INVALID, "unknown entity id");
    translate_entity(s, entity, delta, refit != 0);
    s->dirty = true;
    return RRT_OK;
} catch (...) {
    return on_exception();
}

int rrt_set_ambient(rrt_scene* s, double ambient) {
    if (!s) return fail(RRT_E_INVALID, "null scene");
    s->ambient = ambient;
    s->dirty = true;
    return RRT_OK;
}

int rrt_set_work_budget(rrt_scene* s, uint64_t bytes) {
    if (!s) return fail(RRT_E_INVALID, "null scene");
    s->work_budget = bytes;
    return RRT_OK;
}

int rrt_set_sky_stream(rrt_scene* s, int mode) {
    if (!s || mode < 0 || mode > 1) return fail(RRT_E_INVALID, "sky stream mode must be 0 or 1");
    s->sky_stream = mode;
    return RRT_OK;
}

int rrt_set_timing(rrt_scene* s, int level) {
    if (!s || level < 0 || level > 2) return fail(RRT_E_INVALID, "timing level must be 0, 1 or 2");
    s->timing = level;
    return RRT_OK;
}

int rrt_set_cull_mode(rrt_scene* s, int mode) {
    if (!s || mode < 0 || mode > 2) return fail(RRT_E_INVALID, "cull mode must be 0, 1 or 2");
    s->cull_mode = mode;
    s->dirty = true;
    return RRT_OK;
}

int rrt_scene_commit(rrt_scene* s) try {
    if (!s) return fail(RRT_E_INVALID, "null scene");
    return commit(s);
} catch (...) {
    return on_exception();
}

uint32_t rrt_tile_skew(uint32_t width, uint32_t tile_world) {
    return rrt::tile_skew((width + RRT_TILE_W - 1) / RRT_TILE_W, tile_world);
}

int64_t rrt_tile_slots(uint32_t width, uint32_t height, uint32_t tile_world) {
    if (tile_world == 0) return 0;
    const uint64_t tx = (width + RRT_TILE_W - 1) / RRT_TILE_W, ty = (height + RRT_TILE_H - 1) / RRT_TILE_H;
    return (int64_t)((tx * ty + tile_world - 1) / tile_world);
}

int rrt_render_device(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, void* d_rgb8, void* d_ids, void* d_t,
                      void* d_color, void* stream) try {
    return render_device<false>(s, cam, st, d_rgb8, d_ids, d_t, d_color, stream);
} catch (...) {
    return on_exception();
}

int rrt_render_device_noalloc(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, void* d_rgb8, void* d_ids,
                              void* d_t, void* d_color, void* stream) try {
    return render_device<true>(s, cam, st, d_rgb8, d_ids, d_t, d_color, stream);
} catch (...) {
    return on_exception();
}

uint64_t rrt_noalloc_scratch_bytes(void) { return rrt::kScratchBytes; }

void* rrt_alloc_host_pinned(uint64_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? (size_t)bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        g_err = "hipHostMalloc failed";
        return nullptr;
    }
    return p;
}
void rrt_free_host_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}

int rrt_release_stream(rrt_scene* s, void* stream_v) try {
    if (!s) return fail(RRT_E_INVALID, "null scene");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t)stream_v;
    for (size_t i = 0; i < s->sets.size(); ++i) {
        if (s->sets[i]->stream != stream) continue;
        HIP_TRY(hipStreamSynchronize(stream));          // the set's buffers may still be in use by frames in flight
        if (s->last == s->sets[i].get()) s->last = nullptr;
        s->sets.erase(s->sets.begin() + (long)i);
        return RRT_OK;
    }
    return fail(RRT_E_INVALID, "the scene holds no work buffers for this stream");
} catch (...) {
    return on_exception();
}

int rrt_render(rrt_scene* s, const rrt_camera* cam, const rrt_settings* st, uint8_t* out_rgb8, int32_t* out_ids,
               double* out_t, double* out_color) try {
    if (!s || !st) return fail(RRT_E_INVALID, "null argument");
    if (st->tile_world != 1) return fail(RRT_E_INVALID, "rrt_render needs tile_world == 1");
    {
        FramePlan probe;   // argument errors are reported before anything touches the device
        const int prc = plan_frame(s, cam, st, &probe, s->work_budget);
        if (prc) return prc;
    }
    HIP_TRY(hipSetDevice(s->device));
    const size_t npx = (size_t)st->width * st->height;
    if (out_rgb8) HIP_TRY(s->w_rgb.ensure(npx * 3));
    if (out_ids) HIP_TRY(s->w_ids.ensure(npx * 8));
    if (out_t) HIP_TRY(s->w_t.ensure(npx * 8));
    if (out_color) HIP_TRY(s->w_color.ensure(npx * 24));
    // The frame runs on a stream of the scene's own and the call waits for THAT stream (and the copies behind it), not
    // for the device: other streams of the process go on undisturbed.
    if (!s->host_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&s->host_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        for (hipEvent_t& e : s->band_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // Is the RGB8 destination page-locked host memory (rrt_alloc_host_pinned, hipHostMalloc, a registered range)?
    // Then the copy engine writes into it directly, and the frame is resolved and shipped in bands of tile rows --
    // all but the last band's copy hide under the resolve pass.  Pageable memory: one copy after the frame (the
    // runtime pins or stages it: ~0.3 ms for a 1080p frame).
    bool pinned = false;
    if (out_rgb8) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, out_rgb8) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();        // an unknown (plain malloc) pointer: not an error
    }
    HostBands hb;
    hb.rgb = out_rgb8;
    hb.copy_stream = s->copy_stream;
    hb.ev = s->band_ev;
    hb.n_bands = pinned ? 4u : 0u;
    int rc = render_device<false>(s, cam, st, out_rgb8 ? s->w_rgb.p : nullptr, out_ids ? s->w_ids.p : nullptr,
                           out_t ? s->w_t.p : nullptr, out_color ? s->w_color.p : nullptr, s->host_stream, pinned ? &hb : nullptr);
    if (rc) {
        (void)hipStreamSynchronize(s->copy_stream);
        return rc;
    }
    if (out_rgb8 && !pinned) HIP_TRY(hipMemcpyAsync(out_rgb8, s->w_rgb.p, npx * 3, hipMemcpyDeviceToHost, s->host_stream));
    if (out_ids) HIP_TRY(hipMemcpyAsync(out_ids, s->w_ids.p, npx * 8, hipMemcpyDeviceToHost, s->host_stream));
    if (out_t) HIP_TRY(hipMemcpyAsync(out_t, s->w_t.p, npx * 8, hipMemcpyDeviceToHost, s->host_stream));
    if (out_color) HIP_TRY(hipMemcpyAsync(out_color, s->w_color.p, npx * 24, hipMemcpyDeviceToHost, s->host_stream));
    HIP_TRY(hipStreamSynchronize(s->host_stream));
    if (pinned) HIP_TRY(hipStreamSynchronize(s->copy_stream));
    return RRT_OK;
} catch (...) {
    return on_exception();
}

int rrt_assemble_tiles(const void* d_gathered, uint32_t width, uint32_t height, uint32_t tile_world, void* d_rgb8,
                       void* stream) {
    if (!d_gathered || !d_rgb8 || tile_world == 0 || width == 0 || height == 0)
        return fail(RRT_E_INVALID, "bad argument");
    rrt::launch_assemble((const uint8_t*)d_gathered, width, height, tile_world, (uint8_t*)d_rgb8, stream);
    HIP_TRY(hipGetLastError());
    return RRT_OK;
}

int rrt_assemble_tiles_dealt(const void* d_gathered, uint32_t width, uint32_t height, uint32_t tile_world, void* d_rgb8,
                             void* stream) {
    if (!d_gathered || !d_rgb8 || tile_world == 0 || width == 0 || height == 0)
        return fail(RRT_E_INVALID, "bad argument");
    if (tile_world == 1) return fail(RRT_E_INVALID, "a single rank's frame is row-major already (tile_deal needs tile_world > 1)");
    rrt::launch_assemble_dealt((const uint8_t*)d_gathered, width, height, tile_world, (uint8_t*)d_rgb8, stream);
    HIP_TRY(hipGetLastError());
    return RRT_OK;
}

static int ray_query(rrt_scene* s, const double* rays, int64_t n, bool cast, uint32_t max_bounce, uint32_t prune,
                     int32_t* ent, int32_t* tri, double* t, double* pos, double* nrm, double* color) {
    if (!s || !rays || n < 0) return fail(RRT_E_INVALID, "bad argument");
    if (cast && max_bounce > (uint32_t)rrt::kMaxLevels - 2u) return fail(RRT_E_INVALID, "max_bounce too large (<= 16)");
    if (n == 0) return RRT_OK;
    int rc = commit(s);
    if (rc) return rc;
    DevBuf d_rays, d_ent, d_tri, d_t, d_pos, d_nrm, d_col, d_stats;
    HIP_TRY(d_rays.upload(rays, (size_t)n * 48));
    HIP_TRY(d_stats.ensure(64));
    HIP_TRY(hipMemset(d_stats.p, 0, 64));
    if (cast) {
        HIP_TRY(d_col.ensure((size_t)n * 24));
        rrt::launch_cast_rays(s->dscene, d_rays.as<double>(), n, max_bounce, prune, d_col.as<double>(),
                              d_stats.as<unsigned long long>(), nullptr);
    } else {
        if (ent) HIP_TRY(d_ent.ensure((size_t)n * 4));
        if (tri) HIP_TRY(d_tri.ensure((size_t)n * 4));
        if (t) HIP_TRY(d_t.ensure((size_t)n * 8));
        if (pos) HIP_TRY(d_pos.ensure((size_t)n * 24));
        if (nrm) HIP_TRY(d_nrm.ensure((size_t)n * 24));
        rrt::launch_trace_rays(s->dscene, d_rays.as<double>(), n, ent ? d_ent.as<int32_t>() : nullptr,
                               tri ? d_tri.as<int32_t>() : nullptr, t ? d_t.as<double>() : nullptr,
                               pos ? d_pos.as<double>() : nullptr, nrm ? d_nrm.as<double>() : nullptr,
                               d_stats.as<unsigned long long>(), nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (cast) HIP_TRY(hipMemcpy(color, d_col.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (ent) HIP_TRY(hipMemcpy(ent, d_ent.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (tri) HIP_TRY(hipMemcpy(tri, d_tri.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (t) HIP_TRY(hipMemcpy(t, d_t.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (pos) HIP_TRY(hipMemcpy(pos, d_pos.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    if (nrm) HIP_TRY(hipMemcpy(nrm, d_nrm.p, (size_t)n * 24, hipMemcpyDeviceToHost));
    return RRT_OK;
}

int rrt_occluded_rays(rrt_scene* s, const double* rays, const double* max_dist2, int64_t n, uint8_t* out) try {
    if (!s || !rays || !max_dist2 || !out || n < 0) return fail(RRT_E_INVALID, "bad argument");
    if (n == 0) return RRT_OK;
    int rc = commit(s);
    if (rc) return rc;
    DevBuf d_rays, d_lim, d_out, d_stats;
    HIP_TRY(d_rays.upload(rays, (size_t)n * 48));
    HIP_TRY(d_lim.upload(max_dist2, (size_t)n * 8));
    HIP_TRY(d_out.ensure((size_t)n));
    HIP_TRY(d_stats.ensure(64));
    HIP_TRY(hipMemset(d_stats.p, 0, 64));
    rrt::launch_occluded_rays(s->dscene, d_rays.as<double>(), d_lim.as<double>(), n, d_out.as<uint8_t>(),
                              d_stats.as<unsigned long long>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_out.p, (size_t)n, hipMemcpyDeviceToHost));
    return RRT_OK;
} catch (...) {
    return on_exception();
}

int rrt_trace_rays(rrt_scene* s, const double* rays, int64_t n, int32_t* ent, int32_t* tri, double* t, double* pos,
                   double* normal) try {
    return ray_query(s, rays, n, false, 0, 0, ent, tri, t, pos, normal, nullptr);
} catch (...) {
    return on_exception();
}

int rrt_cast_rays(rrt_scene* s, const double* rays, int64_t n, uint32_t max_bounce, uint32_t prune_zero_mix,
                  double* color) try {
    if (!color) return fail(RRT_E_INVALID, "null colour output");
    return ray_query(s, rays, n, true, max_bounce, prune_zero_mix, nullptr, nullptr, nullptr, nullptr, nullptr, color);
} catch (...) {
    return on_exception();
}

int rrt_get_stats(const rrt_scene* cs, rrt_stats* out) try {
    rrt_scene* s = const_cast<rrt_scene*>(cs);
    if (!s || !out) return fail(RRT_E_INVALID, "null argument");
    if (!s->last) return fail(RRT_E_STATE, "no frame rendered yet");
    HIP_TRY(hipSetDevice(s->device));
    WorkSet* ws = s->last;     // the frame enqueued last (on whichever stream)
    float ms = 0.f;
    if (ws->frame_timed) {
        HIP_TRY(hipEventSynchronize(ws->ev_end));
        HIP_TRY(hipEventElapsedTime(&ms, ws->ev_begin, ws->ev_end));
    } else {
        HIP_TRY(hipDeviceSynchronize());   // an untimed (no-alloc) frame: wait for its stream the blunt way
    }
    ws->stats.gpu_ms = ms;
    double prim = 0.0, bnc = 0.0, sky = 0.0;
    uint32_t nprim = 0, nbnc = 0;
    for (size_t i = 0; i < ws->timed_used; ++i) {
        float k = 0.f;
        HIP_TRY(hipEventElapsedTime(&k, ws->timed[i].a, ws->timed[i].b));
        if (ws->timed[i].kind == 0) { prim += k; nprim += ws->timed[i].launches; }
        else if (ws->timed[i].kind == 1) { bnc += k; nbnc += ws->timed[i].launches; }
        else sky += k;
    }
    ws->stats.sky_kernel_ms = sky;
    ws->stats.primary_kernel_ms = prim;
    ws->stats.bounce_kernel_ms = bnc;
    ws->stats.primary_kernel_launches = nprim;
    ws->stats.bounce_kernel_launches = nbnc;
    unsigned long long dev[32] = {};        // the zero block's head: statistics, queue counters, list counts
    static_assert(sizeof(dev) == 256, "head of the zero block");
    HIP_TRY(hipMemcpy(dev, ws->stats_src, sizeof(dev), hipMemcpyDeviceToHost));
    ws->stats.heavy_tiles = ws->stats.sky_tiles = 0;
    if (ws->lists) {
        uint32_t ln[4];
        std::memcpy(ln, (const unsigned char*)dev + kStatsBytes + kQCountBytes, sizeof(ln));
        ws->stats.heavy_tiles = ln[0];
        ws->stats.sky_tiles = ln[1] - ln[0];
        if (ws->dealt) {                     // which tiles a rank owns is decided on the device: so is its pixel count
            ws->stats.paths = (uint64_t)ln[2] * ws->spp;
            ws->stats.primary_rays = ws->stats.paths;
        }
    }
    ws->stats.rays_traced = ws->stats.primary_rays + dev[0];
    ws->stats.traversal_overflow = dev[1];
    ws->stats.cull_violations = dev[2];
    ws->stats.zero_mix_paths = dev[3];
    ws->stats.level_rays[0] = ws->stats.primary_rays;
    for (int l = 1; l < 4; ++l) ws->stats.level_rays[l] = dev[4 + l];
    *out = ws->stats;
    return RRT_OK;
} catch (...) {
    return on_exception();
}

// the tree of a mesh (over its triangles) or of a group (over its members, numbered in the order given)
static const rrt::KdTreeFlat* tree_of(const rrt_scene* s, int entity) {
    if (!s || entity < 0 || entity >= (int)s->ents.size()) return nullptr;
    const rrt::DEntity ent = s->ents[(size_t)entity];
    const int kind = ent.kind & ~rrt::kEntityMember;
    if (kind == 1) return &s->meshes[(size_t)ent.index]->tree;
    if (kind == 2) return &s->groups[(size_t)ent.index]->tree;
    return nullptr;
}
int rrt_cover_mask(rrt_scene* s, const rrt_camera* cam, uint32_t width, uint32_t height, int entity, uint8_t* pixels, int* every) try {
    if (!s || !cam || !pixels || !every || width == 0 || height == 0) return fail(RRT_E_INVALID, "null argument or empty image");
    if ((uint64_t)width * height > 0x7fffffffull) return fail(RRT_E_INVALID, "image too large");
    if (cover_stride(width) * height + 8 > 0xffffffffull) return fail(RRT_E_INVALID, "image too tall for a coverage mask");
    if (entity < 0 || (size_t)entity >= s->ents.size() || (s->ents[(size_t)entity].kind & ~rrt::kEntityMember) != 1)
        return fail(RRT_E_INVALID, "entity is not a mesh");
    int rc = commit(s);
    if (rc) return rc;
    const int32_t mi = s->ents[(size_t)entity].index;
    int slot = -1;
    for (size_t k = 0; k < s->cover.size(); ++k)
        if (s->cover[k].first == mi) slot = (int)k;
    if (slot < 0 || !(cam->lens_z > 0.0 && cam->lens_w > 0.0 && cam->lens_h > 0.0))
        return fail(RRT_E_STATE, "this mesh (or this lens) has no coverage mask");
    rrt::DFrame fr{};
    fr.cam = *cam;
    fr.width = width;
    fr.height = height;
    fr.n_cover = 1;
    fr.cover_stride = (uint32_t)cover_stride(width);
    fr.cover_bytes = fr.cover_stride * height + 8;
    DevBuf buf;
    HIP_TRY(buf.ensure((size_t)fr.cover_bytes));
    fr.cover = buf.as<uint8_t>();
    HIP_TRY(hipMemsetAsync(buf.p, 0, (size_t)fr.cover_bytes, nullptr));
    rrt::launch_cover(s->dmeshes[(size_t)mi], s->cover[(size_t)slot].second, fr, 0, nullptr);
    HIP_TRY(hipGetLastError());
    std::vector<uint8_t> cells((size_t)fr.cover_bytes);
    HIP_TRY(hipMemcpy(cells.data(), buf.p, cells.size(), hipMemcpyDeviceToHost));
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) pixels[(size_t)y * width + x] = cells[(size_t)y * fr.cover_stride + x / rrt::kCoverCell] ? 1 : 0;
    *every = cells[(size_t)fr.cover_bytes - 8] ? 1 : 0;
    return RRT_OK;
} catch (...) {
    return on_exception();
}

int rrt_mesh_tree_sizes(const rrt_scene* s, int entity, int32_t* n_nodes, int32_t* n_leaf_refs) {
    if (!tree_of(s, entity)) return fail(RRT_E_INVALID, "entity is not a mesh or a group");
    const rrt::KdTreeFlat& T = *tree_of(s, entity);
    if (n_nodes) *n_nodes = T.n_nodes();
    if (n_leaf_refs) *n_leaf_refs = (int32_t)T.leaf_refs.size();
    return RRT_OK;
}

int rrt_mesh_tree_export(const rrt_scene* s, int entity, double* boxes, double* split, int32_t* axis, int32_t* child,
                         int32_t* ropes, int32_t* leaf_refs) {
    if (!tree_of(s, entity)) return fail(RRT_E_INVALID, "entity is not a mesh or a group");
    const rrt::KdTreeFlat& T = *tree_of(s, entity);
    if (boxes) std::memcpy(boxes, T.boxes.data(), T.boxes.size() * 8);
    if (split) std::memcpy(split, T.split.data(), T.split.size() * 8);
    if (axis) std::memcpy(axis, T.axis.data(), T.axis.size() * 4);
    if (child) std::memcpy(child, T.child.data(), T.child.size() * 4);
    if (ropes) std::memcpy(ropes, T.ropes.data(), T.ropes.size() * 4);
    if (leaf_refs) std::memcpy(leaf_refs, T.leaf_refs.data(), T.leaf_refs.size() * 4);
    return RRT_OK;
}

int rrt_selftest_math(int op, const double* a, const double* b, int64_t n, double* out) {
    if (!a || !b || !out || n <= 0 || op < 0 || op > 3) return fail(RRT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(g_device));
    DevBuf da, db, dout;
    HIP_TRY(da.upload(a, (size_t)n * 8));
    HIP_TRY(db.upload(b, (size_t)n * 8));
    HIP_TRY(dout.ensure((size_t)n * 8));
    rrt::launch_selftest(op, da.as<double>(), db.as<double>(), n, dout.as<double>(), nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return RRT_OK;
}

// not part of include/rrt.h: timing-probe builds only (tools/phase_probe.py)
void rrt_debug_phases(unsigned long long out[32]) { rrt::debug_phases(out); }

void rrt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    rrt::philox4x32_10_host(ctr, key, out);
}

}  // extern "C"
