/* =============================================================================
 * rrt_host.h -- host-side asset helpers that sit ABOVE the tracing boundary
 * (file I/O never crosses rrt.h).  They restate the loaders the reference
 * gets from third-party crates, so a host program can be a drop-in for
 * main.rs without Rust:
 *   obj-rs 0.6.2  `load_obj`          (used at src/geometry/model.rs:27)
 *   image 0.23.14 `open().to_rgb8()`  (used at src/behavior/cubemap.rs:19-44)
 *   image 0.23.14 `ImageBuffer::save` (used at src/tracer.rs:167)
 * All return RRT_OK or a negative RRT_E_* with rrt_last_error() set.
 * ========================================================================== */
#ifndef RRT_HOST_H
#define RRT_HOST_H

#include "rrt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_E_IO (-5)      /* file missing / unreadable                         */
#define RRT_E_FORMAT (-6)  /* file is not something the reference would load    */

/* Model::new's triangle list (src/geometry/model.rs:24-45): parses the OBJ with
 * obj-rs semantics (numbers as f32; faces must be triangles carrying a normal
 * index, else error -- obj-rs refuses what it cannot index), widens to f64,
 * applies T(position)*S(scale) to positions (model.rs:29-41), averages the
 * three normalised vertex normals un-transformed (model.rs:35-42) and drops
 * the last face as `for i in (0..indices.len()-4).step_by(3)` does (model.rs:31).
 * *verts (9 doubles/triangle) and *normals (3/triangle) are malloc'ed; release
 * with rrt_host_free. */
int rrt_host_load_model(const char* obj_path, const double position[3], const double scale[3], double** verts,
                        double** normals, int64_t* n_triangles);

/* The same with options.  RRT_OBJ_TRIANGULATE (EXTENSION; obj-rs 0.6.2 has no such mode and the reference
 * would panic at model.rs:27): a face with more than three corners is taken as the triangle fan
 * (c0, ci, ci+1), so that the reference's own obj/ufo.obj (789 quads, 38 triangles, 12 n-gons, "v/vt/vn"
 * corners) loads; everything else -- f32 parse, transform, averaged normals, the dropped last triangle --
 * is as rrt_host_load_model.  flags = 0 is rrt_host_load_model. */
#define RRT_OBJ_TRIANGULATE 1u
int rrt_host_load_model_ex(const char* obj_path, const double position[3], const double scale[3], uint32_t flags,
                           double** verts, double** normals, int64_t* n_triangles);
int rrt_host_add_model_ex(rrt_scene* s, const char* obj_path, int material, const double position[3],
                          const double scale[3], uint32_t flags);

/* Model::new(path, material, position, scale) + world.entities.push
 * (src/geometry/model.rs:24-56, src/main.rs:61).  Returns the entity id. */
int rrt_host_add_model(rrt_scene* s, const char* obj_path, int material, const double position[3],
                       const double scale[3]);

/* image::open(path).to_rgb8() for PNG: every colour type (grey, RGB, palette, grey+alpha, RGBA) at every
 * bit depth the format allows (1/2/4/8/16), plain or Adam7-interlaced; alpha is dropped, palettes are
 * expanded, grey is replicated (packed grey scaled to 0..255), 16-bit samples keep their high byte
 * (image 0.23.14 downcast_channel).  The reference's own cubemaps are 8-bit RGB / RGBA / palette,
 * non-interlaced; the rest is covered by generated fixtures (tests/golden/png/).
 * *rgb8 is malloc'ed (h rows of w texels). */
int rrt_host_load_png_rgb8(const char* path, uint8_t** rgb8, uint32_t* w, uint32_t* h);

/* ImageBuffer<Rgb<u8>>::save as PNG (src/tracer.rs:167). */
int rrt_host_save_png_rgb8(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h);

/* CubemapBehavior::new(folder) (src/behavior/cubemap.rs:16-52): loads
 * left,right,front,back,up,down .png and installs them in the reference's
 * maps[] order [left,right,down,up,front,back]. */
int rrt_host_load_cubemap(rrt_scene* s, const char* folder);

void rrt_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* RRT_HOST_H */
