#include "kd_build.hpp"

#include <cmath>
#include <limits>
#include <utility>

#include "unstable_sort.hpp"

namespace rrt {
namespace {

struct Box {
    double mn[3], mx[3];
};

struct Pending {
    std::vector<int32_t> items;
    Box box;
    int32_t depth;
    int32_t parent;  // -1 for the root
    int32_t side;    // 0 = left, 1 = right
};

inline double min3(double a, double b, double c) { return std::fmin(a, std::fmin(b, c)); }
inline double max3(double a, double b, double c) { return std::fmax(a, std::fmax(b, c)); }

}  // namespace

KdTreeFlat build_kd_tree(const double* verts, int64_t n_tris) {
    // Triangle::bounding_box (src/geometry/triangle.rs:60-65)
    std::vector<double> tmin((size_t)n_tris * 3), tmax((size_t)n_tris * 3);
    for (int64_t i = 0; i < n_tris; ++i) {
        const double* v = verts + 9 * i;
        for (int a = 0; a < 3; ++a) {
            tmin[3 * i + a] = min3(v[a], v[3 + a], v[6 + a]);
            tmax[3 * i + a] = max3(v[a], v[3 + a], v[6 + a]);
        }
    }
    return build_kd_tree_boxes(tmin, tmax, n_tris);
}

KdTreeFlat build_kd_tree_boxes(const std::vector<double>& tmin, const std::vector<double>& tmax, int64_t n_tris) {
    KdTreeFlat T;
    // AABB::from_entities (src/geometry/aabb.rs:25-39)
    Box root;
    for (int a = 0; a < 3; ++a) {
        root.mn[a] = std::numeric_limits<double>::max();
        root.mx[a] = std::numeric_limits<double>::lowest();
    }
    for (int64_t i = 0; i < n_tris; ++i)
        for (int a = 0; a < 3; ++a) {
            if (tmin[3 * i + a] < root.mn[a]) root.mn[a] = tmin[3 * i + a];
            if (tmax[3 * i + a] > root.mx[a]) root.mx[a] = tmax[3 * i + a];
        }

    auto emit = [&T](const Box& b) -> int32_t {
        const int32_t idx = T.n_nodes();
        for (int a = 0; a < 3; ++a) T.boxes.push_back(b.mn[a]);
        for (int a = 0; a < 3; ++a) T.boxes.push_back(b.mx[a]);
        T.split.push_back(0.0);
        T.axis.push_back(-1);
        T.child.push_back(0);
        T.child.push_back(0);
        for (int k = 0; k < 6; ++k) T.ropes.push_back(-1);
        return idx;
    };

    std::vector<Pending> stack;
    {
        Pending p;
        p.items.resize((size_t)n_tris);
        for (int64_t i = 0; i < n_tris; ++i) p.items[(size_t)i] = (int32_t)i;
        p.box = root;
        p.depth = 0;
        p.parent = -1;
        p.side = 0;
        stack.push_back(std::move(p));
    }
    std::vector<SortRec> keyed;
    while (!stack.empty()) {
        Pending cur = std::move(stack.back());
        stack.pop_back();
        const int32_t me = emit(cur.box);
        if (cur.parent >= 0) T.child[2 * (size_t)cur.parent + cur.side] = me;
        if (cur.depth > T.max_depth) T.max_depth = cur.depth;

        auto make_leaf = [&](const std::vector<int32_t>& items) {
            T.child[2 * (size_t)me] = (int32_t)T.leaf_refs.size();
            T.child[2 * (size_t)me + 1] = (int32_t)items.size();
            T.leaf_refs.insert(T.leaf_refs.end(), items.begin(), items.end());
            if ((int32_t)items.size() > T.max_leaf) T.max_leaf = (int32_t)items.size();
        };

        const size_t n = cur.items.size();
        if (n < 5) {
            make_leaf(cur.items);
            continue;
        }
        const int ax = cur.depth % 3;
        keyed.resize(n);
        for (size_t i = 0; i < n; ++i) keyed[i] = SortRec{tmin[3 * (size_t)cur.items[i] + ax], cur.items[i]};
        UnstableSort::run(keyed.data(), n);

        const size_t half = n / 2;
        double cut = keyed[half].key;
        if (std::fabs(cut - cur.box.mn[ax]) < 0.001 || std::fabs(cut - cur.box.mx[ax]) < 0.001)
            cut = (cur.box.mn[ax] + cur.box.mx[ax]) / 2.;

        std::vector<int32_t> lo, hi;
        lo.reserve(n);
        hi.reserve(n);
        for (size_t i = 0; i < half; ++i) lo.push_back(keyed[i].id);
        for (size_t i = half; i < n; ++i) hi.push_back(keyed[i].id);
        for (size_t i = 0; i < half; ++i)
            if (tmax[3 * (size_t)keyed[i].id + ax] >= cut) hi.push_back(keyed[i].id);
        for (size_t i = half; i < n; ++i)
            if (keyed[i].key < cut) lo.push_back(keyed[i].id);

        if (lo.size() >= n) {
            make_leaf(lo);
            continue;
        }
        if (hi.size() >= n) {
            make_leaf(hi);
            continue;
        }
        T.axis[(size_t)me] = ax;
        T.split[(size_t)me] = cut;
        Pending L, R;
        L.items = std::move(lo);
        L.box = cur.box;
        L.box.mx[ax] = cut;
        L.depth = cur.depth + 1;
        L.parent = me;
        L.side = 0;
        R.items = std::move(hi);
        R.box = cur.box;
        R.box.mn[ax] = cut;
        R.depth = cur.depth + 1;
        R.parent = me;
        R.side = 1;
        stack.push_back(std::move(R));  // left is popped first -> pre-order
        stack.push_back(std::move(L));
    }

    // make_ropes (src/geometry/kdtree.rs:139-190), top-down with an explicit stack.
    struct RopeJob {
        int32_t node;
        int32_t in[6];
    };
    std::vector<RopeJob> jobs;
    jobs.push_back(RopeJob{0, {-1, -1, -1, -1, -1, -1}});
    while (!jobs.empty()) {
        RopeJob j = jobs.back();
        jobs.pop_back();
        const size_t me = (size_t)j.node;
        if (T.axis[me] < 0) {
            for (int k = 0; k < 6; ++k) T.ropes[6 * me + k] = j.in[k];
            continue;
        }
        const double* mybox = &T.boxes[6 * me];
        for (int k = 0; k < 6; ++k) {
            int32_t t = j.in[k];
            if (t < 0) continue;
            while (T.axis[(size_t)t] >= 0) {
                const int ta = T.axis[(size_t)t];
                const double tp = T.split[(size_t)t];
                if (ta == k % 3) t = T.child[2 * (size_t)t + (k < 3 ? 1 : 0)];
                else if (tp > mybox[3 + ta]) t = T.child[2 * (size_t)t];
                else if (tp < mybox[ta]) t = T.child[2 * (size_t)t + 1];
                else break;
            }
            j.in[k] = t;
        }
        const int ax = T.axis[me];
        const int32_t lc = T.child[2 * me], rc = T.child[2 * me + 1];
        RopeJob jl{lc, {j.in[0], j.in[1], j.in[2], j.in[3], j.in[4], j.in[5]}};
        jl.in[ax + 3] = rc;
        RopeJob jr{rc, {j.in[0], j.in[1], j.in[2], j.in[3], j.in[4], j.in[5]}};
        jr.in[ax] = lc;
        jobs.push_back(jr);
        jobs.push_back(jl);
    }
    return T;
}

}  // namespace rrt
