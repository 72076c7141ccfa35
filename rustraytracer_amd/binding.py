"""ctypes binding of librrt.so (include/rrt.h, include/rrt_host.h).

The HIP library is the only implementation: if it is missing or does not
load, importing this module raises -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRT_LIB", os.path.join(_HERE, "librrt.so"))   # RRT_LIB: timing-probe builds only


class RrtError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("lens_z", C.c_double),
                ("lens_w", C.c_double), ("lens_h", C.c_double)]


class Settings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("max_bounce", C.c_uint32), ("seed", C.c_uint64),
                ("prune_zero_mix", C.c_uint32), ("tile_rank", C.c_uint32),
                ("tile_world", C.c_uint32), ("batch_spp", C.c_uint32), ("tile_deal", C.c_uint32), ("reserved", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays_traced", C.c_uint64), ("primary_rays", C.c_uint64), ("paths", C.c_uint64),
                ("traversal_overflow", C.c_uint64), ("gpu_ms", C.c_double),
                ("primary_kernel_ms", C.c_double), ("bounce_kernel_ms", C.c_double),
                ("primary_kernel_launches", C.c_uint32), ("bounce_kernel_launches", C.c_uint32),
                ("n_batches", C.c_uint32), ("reserved", C.c_uint32),
                ("level_rays", C.c_uint64 * 4), ("cull_violations", C.c_uint64),
                ("zero_mix_paths", C.c_uint64), ("sky_kernel_ms", C.c_double),
                ("heavy_tiles", C.c_uint32), ("sky_tiles", C.c_uint32)]


# every symbol include/rrt.h and include/rrt_host.h declare
EXPORTS = [
    "rrt_last_error", "rrt_version", "rrt_set_device", "rrt_device_image_support",
    "rrt_default_camera", "rrt_default_settings", "rrt_scene_create", "rrt_scene_destroy",
    "rrt_add_material", "rrt_add_material_ext", "rrt_add_sphere", "rrt_add_mesh", "rrt_add_triangle", "rrt_add_group", "rrt_set_sky_face",
    "rrt_set_directional_light", "rrt_add_directional_light", "rrt_set_ambient", "rrt_add_point_light", "rrt_set_shadow_mode", "rrt_translate_entity",
    "rrt_occluded_rays", "rrt_set_work_budget", "rrt_set_cull_mode", "rrt_set_timing", "rrt_set_sky_stream",
    "rrt_scene_commit",
    "rrt_render", "rrt_alloc_host_pinned", "rrt_free_host_pinned", "rrt_release_stream", "rrt_render_device", "rrt_render_device_noalloc", "rrt_noalloc_scratch_bytes", "rrt_tile_slots", "rrt_tile_buffer_bytes", "rrt_tile_skew", "rrt_comm_unique_id", "rrt_comm_init", "rrt_comm_destroy", "rrt_gather_tiles", "rrt_gather_tiles_dealt", "rrt_assemble_tiles", "rrt_assemble_tiles_dealt",
    "rrt_trace_rays", "rrt_cast_rays", "rrt_get_stats", "rrt_mesh_tree_sizes",
    "rrt_mesh_tree_export", "rrt_cover_mask", "rrt_selftest_math", "rrt_philox4x32_10",
    "rrt_host_load_model", "rrt_host_add_model", "rrt_host_load_model_ex", "rrt_host_add_model_ex", "rrt_host_load_png_rgb8",
    "rrt_host_save_png_rgb8", "rrt_host_load_cubemap", "rrt_host_free",
]


def _load():
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7
    # (same SONAME as /opt/rocm's).  Whichever is loaded first serves both, and
    # torch does not find the GPU on a runtime it did not bring.  So when torch is
    # importable it is loaded first and librrt.so binds to torch's runtime; the
    # library itself has no torch dependency (RRT_NO_TORCH=1 skips this).
    if os.environ.get("RRT_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "librrt.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C rustraytracer_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, dp, i32, i64, u32, u64 = C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_uint32, C.c_uint64
    L.rrt_last_error.restype = C.c_char_p
    L.rrt_version.restype = C.c_char_p
    L.rrt_set_device.argtypes = [i32]
    L.rrt_default_camera.argtypes = [C.POINTER(Camera)]
    L.rrt_default_camera.restype = None
    L.rrt_default_settings.argtypes = [C.POINTER(Settings), u32, u32]
    L.rrt_default_settings.restype = None
    L.rrt_scene_create.restype = vp
    L.rrt_scene_destroy.argtypes = [vp]
    L.rrt_scene_destroy.restype = None
    L.rrt_add_material.argtypes = [vp, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32]
    L.rrt_add_material_ext.argtypes = [vp, dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_double]
    L.rrt_add_sphere.argtypes = [vp, dp, C.c_double, i32]
    L.rrt_add_mesh.argtypes = [vp, vp, vp, i64, i32]
    L.rrt_add_triangle.argtypes = [vp, dp, dp, dp, dp, i32]
    L.rrt_add_group.argtypes = [vp, vp, i32]
    L.rrt_set_sky_face.argtypes = [vp, i32, vp, u32, u32]
    L.rrt_set_directional_light.argtypes = [vp, dp, dp, C.c_double]
    L.rrt_add_directional_light.argtypes = [vp, dp, dp, C.c_double]
    L.rrt_set_ambient.argtypes = [vp, C.c_double]
    L.rrt_add_point_light.argtypes = [vp, dp, dp, C.c_double, C.c_double]
    L.rrt_set_shadow_mode.argtypes = [vp, i32]
    L.rrt_translate_entity.argtypes = [vp, i32, dp, i32]
    L.rrt_occluded_rays.argtypes = [vp, vp, vp, i64, vp]
    L.rrt_set_work_budget.argtypes = [vp, u64]
    L.rrt_set_cull_mode.argtypes = [vp, i32]
    L.rrt_set_timing.argtypes = [vp, i32]
    L.rrt_set_sky_stream.argtypes = [vp, i32]
    L.rrt_scene_commit.argtypes = [vp]
    L.rrt_render.argtypes = [vp, C.POINTER(Camera), C.POINTER(Settings), vp, vp, vp, vp]
    L.rrt_render_device.argtypes = [vp, C.POINTER(Camera), C.POINTER(Settings), vp, vp, vp, vp, vp]
    L.rrt_render_device_noalloc.argtypes = [vp, C.POINTER(Camera), C.POINTER(Settings), vp, vp, vp, vp, vp]
    L.rrt_noalloc_scratch_bytes.restype = u64
    L.rrt_alloc_host_pinned.argtypes = [u64]
    L.rrt_alloc_host_pinned.restype = vp
    L.rrt_free_host_pinned.argtypes = [vp]
    L.rrt_free_host_pinned.restype = None
    L.rrt_release_stream.argtypes = [vp, vp]
    L.rrt_comm_unique_id.argtypes = [vp]
    L.rrt_comm_init.argtypes = [i32, i32, vp, C.POINTER(vp)]
    L.rrt_comm_destroy.argtypes = [vp]
    L.rrt_comm_destroy.restype = None
    L.rrt_gather_tiles.argtypes = [vp, vp, vp, u32, u32, i32, vp]
    L.rrt_tile_skew.argtypes = [u32, u32]
    L.rrt_tile_skew.restype = u32
    L.rrt_tile_slots.argtypes = [u32, u32, u32]
    L.rrt_tile_slots.restype = i64
    L.rrt_assemble_tiles.argtypes = [vp, u32, u32, u32, vp, vp]
    L.rrt_assemble_tiles_dealt.argtypes = [vp, u32, u32, u32, vp, vp]
    L.rrt_gather_tiles_dealt.argtypes = [vp, vp, vp, u32, u32, i32, vp]
    L.rrt_tile_buffer_bytes.argtypes = [u32, u32, u32, u32]
    L.rrt_tile_buffer_bytes.restype = u64
    L.rrt_trace_rays.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    L.rrt_cast_rays.argtypes = [vp, vp, i64, u32, u32, vp]
    L.rrt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rrt_mesh_tree_sizes.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rrt_mesh_tree_export.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.rrt_cover_mask.argtypes = [vp, vp, C.c_uint32, C.c_uint32, i32, vp, vp]
    L.rrt_selftest_math.argtypes = [i32, vp, vp, i64, vp]
    L.rrt_philox4x32_10.argtypes = [vp, vp, vp]
    L.rrt_philox4x32_10.restype = None
    L.rrt_host_load_model.argtypes = [C.c_char_p, dp, dp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.rrt_host_add_model.argtypes = [vp, C.c_char_p, i32, dp, dp]
    L.rrt_host_load_model_ex.argtypes = [C.c_char_p, dp, dp, u32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64)]
    L.rrt_host_add_model_ex.argtypes = [vp, C.c_char_p, i32, dp, dp, u32]
    L.rrt_host_load_png_rgb8.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]
    L.rrt_host_save_png_rgb8.argtypes = [C.c_char_p, vp, u32, u32]
    L.rrt_host_load_cubemap.argtypes = [vp, C.c_char_p]
    L.rrt_host_free.argtypes = [vp]
    L.rrt_host_free.restype = None
    return L


lib = _load()


def check(rc):
    if rc < 0:
        raise RrtError("rrt error %d: %s" % (rc, lib.rrt_last_error().decode()))
    return rc
