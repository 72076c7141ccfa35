// Host-side builder of the rope-linked kd-tree the reference builds in
// KDTree::new (src/geometry/kdtree.rs:46-51): build_tree (:53-137) followed by
// make_ropes (:139-190), emitted directly as the flat pre-order arrays the HIP
// kernels traverse.  Semantics that must match the reference bit for bit:
//   * axis = depth % 3; leaf when fewer than 5 items              (:54-65)
//   * sort by bbox-min[axis] with sort_unstable_by's tie order    (:70)  -> unstable_sort.hpp
//   * split = min[axis] of item len/2; box midpoint if within 1e-3 of a box face (:69-75)
//   * left  = first half  + right-half items with min <  split
//     right = second half + left-half items with max >= split     (:77-91)
//   * if either side is not smaller than the input it becomes a leaf (:93-113)
//   * child boxes = parent box cut at split                       (:115-126)
//   * ropes: index 0..2 = -x,-y,-z faces, 3..5 = +x,+y,+z; pushed down through
//     interior targets while unambiguous (:145-161); leaves keep them (:174)
#pragma once
#include <cstdint>
#include <vector>

namespace rrt {

struct KdTreeFlat {
    std::vector<double> boxes;      // 6 per node: min xyz, max xyz
    std::vector<double> split;      // 1 per node (0 for leaves)
    std::vector<int32_t> axis;      // 1 per node, -1 = leaf
    std::vector<int32_t> child;     // 2 per node: interior {left,right}; leaf {first,count} into leaf_refs
    std::vector<int32_t> ropes;     // 6 per node, -1 = none (leaves only; interior rows are -1)
    std::vector<int32_t> leaf_refs; // triangle ids, leaf after leaf in pre-order
    int32_t n_nodes() const { return (int32_t)axis.size(); }
    int32_t max_depth = 0;
    int32_t max_leaf = 0;
};

// verts: 9 doubles per triangle.  Returns the flat tree.
KdTreeFlat build_kd_tree(const double* verts, int64_t n_tris);

// KDTree<T: Entity> is generic (kdtree.rs:45): the build consults nothing but each item's
// bounding_box().  item_min / item_max: 3 doubles per item; leaf_refs then holds item numbers.
// Used for triangles (above) and for Scene<T>'s tree over whole entities (scene.rs:39-47).
KdTreeFlat build_kd_tree_boxes(const std::vector<double>& item_min, const std::vector<double>& item_max, int64_t n_items);

}  // namespace rrt
