"""ctypes mirror of include/rt_api.h and the loader for librt_amd.so.

The library is the product; this file only declares its C ABI to Python.  It
never falls back to a CPU implementation: if the shared library is missing the
import of `raytracer_amd` fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_amd.so")

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_PARSE = -2
RT_ERR_IO = -3
RT_ERR_DEVICE = -4
RT_ERR_UNSUPPORTED = -5
RT_NOALLOC_MAX_SLOTS = 84 * 4096
RT_NOALLOC_MAX_DEPTH = 64

(RT_SHAPE_SPHERE, RT_SHAPE_PLANE, RT_SHAPE_XYRECT, RT_SHAPE_XZRECT, RT_SHAPE_ZYRECT, RT_SHAPE_CUBE,
 RT_SHAPE_CYLINDER, RT_SHAPE_TORUS, RT_SHAPE_UNION, RT_SHAPE_INTERSECTION, RT_SHAPE_DIFFERENCE,
 RT_SHAPE_COMPOSITE, RT_SHAPE_MESH, RT_SHAPE_MEDIUM) = range(14)
(RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC) = range(5)
(RT_TEX_SOLID, RT_TEX_CHECKERBOARD, RT_TEX_IMAGE) = range(3)


class rt_shape_node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("child_a", C.c_int32), ("child_b", C.c_int32), ("flags", C.c_int32),
                ("params", C.c_double * 8), ("o2w", C.c_double * 16), ("w2o", C.c_double * 16)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("texture", C.c_int32), ("fuzz", C.c_double), ("ior", C.c_double),
                ("intensity", C.c_double)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_int32), ("image", C.c_int32), ("c1", C.c_double * 3), ("c2", C.c_double * 3),
                ("scale", C.c_double)]


class rt_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class rt_mesh(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("n_vertices", C.c_uint32), ("first_normal", C.c_uint32),
                ("n_normals", C.c_uint32), ("first_triangle", C.c_uint32), ("n_triangles", C.c_uint32)]


class rt_object(C.Structure):
    _fields_ = [("shape", C.c_int32), ("material", C.c_int32)]


class rt_camera(C.Structure):
    _fields_ = [("fov_factor", C.c_double), ("camera_to_world", C.c_double * 16)]


class rt_scene_desc(C.Structure):
    _fields_ = [("camera", rt_camera), ("background", C.c_double * 3),
                ("n_objects", C.c_uint32), ("objects", C.POINTER(rt_object)),
                ("n_nodes", C.c_uint32), ("nodes", C.POINTER(rt_shape_node)),
                ("n_child_list", C.c_uint32), ("child_list", C.POINTER(C.c_int32)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(rt_material)),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(rt_texture)),
                ("n_images", C.c_uint32), ("images", C.POINTER(rt_image)),
                ("n_image_bytes", C.c_uint64), ("image_data", C.POINTER(C.c_uint8)),
                ("n_meshes", C.c_uint32), ("meshes", C.POINTER(rt_mesh)),
                ("n_vertices", C.c_uint32), ("vertices", C.POINTER(C.c_double)),
                ("n_normals", C.c_uint32), ("normals", C.POINTER(C.c_double)),
                ("n_triangles", C.c_uint32), ("tri_v", C.POINTER(C.c_uint32)), ("tri_n", C.POINTER(C.c_uint32))]


class rt_options(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32), ("max_depth", C.c_uint32),
                ("bias", C.c_double), ("seed", C.c_uint64), ("tile_size", C.c_uint32), ("shard_rank", C.c_uint32),
                ("shard_count", C.c_uint32), ("flags", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary_rays", C.c_uint64), ("strata", C.c_uint32),
                ("bounce_passes", C.c_uint32), ("guards", C.c_uint32 * 5), ("interval_overflow", C.c_uint32),
                ("ms_device", C.c_double), ("ms_raygen", C.c_double), ("ms_intersect", C.c_double),
                ("ms_shade", C.c_double), ("ms_mesh", C.c_double), ("ms_tail", C.c_double),
                ("n_intersect_launches", C.c_uint32), ("n_mesh_launches", C.c_uint32),
                ("mesh_rays", C.c_uint64), ("intersect_rays", C.c_uint64), ("mesh_records", C.c_uint64),
                ("mesh_one_rays", C.c_uint64)]


class rt_hit(C.Structure):
    _fields_ = [("t", C.c_double), ("n", C.c_double * 3), ("uv", C.c_double * 2), ("object", C.c_int32),
                ("pad", C.c_int32)]


rt_progress_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double))
rt_stratum_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)
rt_snapshot_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint64)


class rt_progress(C.Structure):
    _fields_ = [("on_stratum", rt_stratum_cb), ("on_snapshot", rt_snapshot_cb), ("min_interval_ms", C.c_uint32),
                ("reserved", C.c_uint32), ("user", C.c_void_p)]


# every symbol include/rt_api.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_options), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                            C.POINTER(rt_stats), rt_progress_cb, C.c_void_p]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(rt_options), C.c_void_p, C.c_void_p, C.POINTER(rt_stats),
                                   C.c_void_p]),
    "rt_render_device_progress": (C.c_int, [C.c_void_p, C.POINTER(rt_options), C.c_void_p, C.c_void_p,
                                            C.POINTER(rt_stats), C.c_void_p, C.POINTER(rt_progress)]),
    "rt_render_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(rt_options), C.POINTER(C.c_double),
                                  C.POINTER(C.c_int32), C.POINTER(rt_stats), C.POINTER(rt_progress)]),
    "rt_render_rgb8": (C.c_int, [C.c_void_p, C.POINTER(rt_options), C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                 C.POINTER(rt_stats), C.POINTER(rt_progress)]),
    "rt_render_multi_rgb8": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(rt_options), C.POINTER(C.c_uint8),
                                       C.POINTER(C.c_int32), C.POINTER(rt_stats), C.POINTER(rt_progress)]),
    "rt_pinned_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "rt_pinned_free": (None, [C.c_void_p]),
    "rt_render_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(rt_options), C.c_void_p, C.c_void_p,
                                           C.POINTER(rt_stats), C.c_void_p]),
    "rt_shard_pixels": (C.c_uint64, [C.POINTER(rt_options)]),
    "rt_shard_map": (C.c_int, [C.POINTER(rt_options), C.POINTER(C.c_uint32)]),
    "rt_unshard_device": (C.c_int, [C.POINTER(rt_options), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "rt_unshard_rgb8_device": (C.c_int, [C.POINTER(rt_options), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_trace_rays": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(rt_hit), C.POINTER(rt_stats)]),
    "rt_tonemap": (C.c_int, [C.POINTER(C.c_double), C.c_uint64, C.c_double, C.POINTER(C.c_uint8)]),
    "rt_tonemap_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]),
    "rt_write_png": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32]),
    "rt_last_error": (C.c_char_p, []),
    "rt_sdl_parse": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(rt_scene_desc))]),
    "rt_sdl_parse_file": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32,
                                    C.POINTER(C.POINTER(rt_scene_desc))]),
    "rt_scene_desc_free": (None, [C.POINTER(rt_scene_desc)]),
}


def load(path=None):
    """Load librt_amd.so and attach the prototypes.  Raises if it is missing."""
    path = path or os.environ.get("RT_AMD_LIB") or LIB_PATH  # RT_AMD_LIB: another build of the same library (experiments)
    if not os.path.exists(path):
        raise ImportError(
            "raytracer_amd: %s not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    # PyTorch-ROCm wheels carry their own libamdhip64.so; librt_amd.so links the one under /opt/rocm.  Whichever
    # is loaded first serves the whole process, and when it is ours a later `import torch` finds "No HIP GPUs".
    # A Python process that has torch installed therefore loads torch first (hosts without torch are unaffected).
    import sys
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
