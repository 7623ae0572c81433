"""ctypes mirror of include/rt2022.h, include/rt2022_host.h and include/rt2022_debug.h.

Plumbing only: every structure here has the byte layout of its C twin (checked by
tests/test_abi.py against `rtb_abi_sizes`). Loading fails loudly when the HIP
library has not been built — there is no Python or CPU fallback for the path.
"""
import ctypes as C
import os

RT2022_ABI_VERSION = 3

RT_REF_FLIP = 0x80000000
RT_REF_KIND_SHIFT = 27
RT_REF_INDEX_MASK = 0x07FFFFFF

(RT_KIND_NODE, RT_KIND_SPHERE, RT_KIND_MOVING_SPHERE, RT_KIND_RECT, RT_KIND_BOX, RT_KIND_TRIANGLE,
 RT_KIND_RING, RT_KIND_MEDIUM, RT_KIND_TRANSLATE, RT_KIND_ROTATE_Y, RT_KIND_ZOOM, RT_KIND_LIST) = range(12)
RT_KIND_COUNT = 12
RT_RECT_XY, RT_RECT_XZ, RT_RECT_YZ = 0, 1, 2
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT, RT_MAT_ISOTROPIC = range(5)
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_NOISE, RT_TEX_IMAGE = range(4)
RT_MAX_XFORM_DEPTH = 4
RT_FLAG_COUNTERS = 0x1
RT_FLAG_KERNEL_TIMES = 0x2
RT_FLAG_ASYNC = 0x4
RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_DEVICE, RT_ERR_NOMEM = 0, -1, -2, -3, -4

KIND_NAMES = ["node", "sphere", "moving_sphere", "rect", "box", "triangle", "ring", "medium",
              "translate", "rotate_y", "zoom", "list"]


def make_ref(kind, index, flip=False):
    return ((kind << RT_REF_KIND_SHIFT) | (index & RT_REF_INDEX_MASK) | (RT_REF_FLIP if flip else 0)) & 0xFFFFFFFF


def ref_kind(ref):
    return (ref >> RT_REF_KIND_SHIFT) & 0xF


def ref_index(ref):
    return ref & RT_REF_INDEX_MASK


d3 = C.c_double * 3


class rt_bvh_node(C.Structure):
    _fields_ = [("bmin", d3), ("bmax", d3), ("left", C.c_uint32), ("right", C.c_uint32), ("_pad", C.c_uint32 * 2)]


class rt_sphere(C.Structure):
    _fields_ = [("center", d3), ("radius", C.c_double), ("mat", C.c_uint32), ("_pad", C.c_uint32)]


class rt_moving_sphere(C.Structure):
    _fields_ = [("center0", d3), ("center1", d3), ("time0", C.c_double), ("time1", C.c_double),
                ("radius", C.c_double), ("mat", C.c_uint32), ("_pad", C.c_uint32)]


class rt_rect(C.Structure):
    _fields_ = [("a0", C.c_double), ("a1", C.c_double), ("b0", C.c_double), ("b1", C.c_double), ("k", C.c_double),
                ("axis", C.c_uint32), ("mat", C.c_uint32)]


class rt_box(C.Structure):
    _fields_ = [("p0", d3), ("p1", d3), ("mat", C.c_uint32), ("_pad", C.c_uint32)]


class rt_triangle(C.Structure):
    _fields_ = [("a", d3), ("b", d3), ("c", d3), ("mat", C.c_uint32), ("_pad", C.c_uint32)]


class rt_ring(C.Structure):
    _fields_ = [("r", C.c_double), ("t", C.c_double), ("dis_min", C.c_double), ("dis_max", C.c_double),
                ("mat", C.c_uint32), ("_pad", C.c_uint32)]


class rt_medium(C.Structure):
    _fields_ = [("boundary", C.c_uint32), ("mat", C.c_uint32), ("neg_inv_density", C.c_double)]


class rt_xform(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("child", C.c_uint32), ("p", d3)]


class rt_list(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tex", C.c_uint32), ("albedo", d3), ("param", C.c_double)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("_pad", C.c_uint32),
                ("color", d3), ("scale", C.c_double)]


class rt_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class rt_perlin(C.Structure):
    _fields_ = [("randvec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


def _pool(name, ctype):
    return [("n_" + name, C.c_uint32), (name, C.POINTER(ctype))]


class rt_scene_desc(C.Structure):
    _fields_ = ([("abi_version", C.c_uint32), ("root", C.c_uint32)]
                + _pool("nodes", rt_bvh_node) + _pool("spheres", rt_sphere)
                + _pool("moving_spheres", rt_moving_sphere) + _pool("rects", rt_rect) + _pool("boxes", rt_box)
                + _pool("triangles", rt_triangle) + _pool("rings", rt_ring) + _pool("media", rt_medium)
                + _pool("xforms", rt_xform) + _pool("lists", rt_list) + _pool("list_items", C.c_uint32)
                + _pool("lights", C.c_uint32) + _pool("materials", rt_material) + _pool("textures", rt_texture)
                + _pool("images", rt_image)
                + [("image_data_bytes", C.c_uint64), ("image_data", C.POINTER(C.c_uint8))]
                + _pool("perlins", rt_perlin))


class rt_camera(C.Structure):
    _fields_ = [("origin", d3), ("lower_left_corner", d3), ("horizontal", d3), ("vertical", d3),
                ("u", d3), ("v", d3), ("w", d3), ("lens_radius", C.c_double), ("time0", C.c_double),
                ("time1", C.c_double)]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
                ("background", d3), ("t_min", C.c_double), ("seed", C.c_uint64),
                ("n_frames", C.c_uint32), ("n_rows", C.c_uint32), ("row_ids", C.c_void_p),
                ("spp_chunk", C.c_uint32), ("flags", C.c_uint32),
                ("progress_cb", C.c_void_p), ("progress_user", C.c_void_p)]


# void (*progress_cb)(void *user, uint32_t worker, uint64_t paths_done, uint64_t paths_total)
PROGRESS_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64)


class rt_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64),
                ("prim_tests", C.c_uint64 * RT_KIND_COUNT), ("light_pdf_tests", C.c_uint64),
                ("rng_draws", C.c_uint64), ("ms", C.c_double),
                ("spp_chunk", C.c_uint32), ("passes", C.c_uint32), ("pool_slots", C.c_uint64),
                ("trace_ms", C.c_double), ("shade_ms", C.c_double), ("partial_bytes", C.c_uint64)]

    def as_dict(self):
        return {"paths": self.paths, "rays": self.rays, "node_visits": self.node_visits,
                "prim_tests": list(self.prim_tests), "light_pdf_tests": self.light_pdf_tests,
                "rng_draws": self.rng_draws}


ABI_STRUCTS = [rt_bvh_node, rt_sphere, rt_moving_sphere, rt_rect, rt_box, rt_triangle, rt_ring, rt_medium, rt_xform,
               rt_list, rt_material, rt_texture, rt_image, rt_perlin, rt_scene_desc, rt_camera, rt_params, rt_stats]

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT2022_LIB") or os.path.join(_HERE, "librt2022.so")   # (RT2022_LIB: A/B builds of the same library, tools/ab.sh)

# Every symbol the three headers declare (tests/test_abi.py checks the export table).
ABI_SYMBOLS = [
    "rt_scene_create", "rt_scene_destroy", "rt_render", "rt_render_device", "rt_render_wait", "rt_write_color",
    "rt_tonemap_device", "rt_last_error", "rt_abi_version", "rt_scene_set_create", "rt_scene_set_destroy", "rt_render_multi",
    "rtb_scene_build", "rtb_scene_free", "rtb_scene_desc", "rtb_scene_default_view", "rtb_camera_new",
    "rtb_shuffled_rows", "rtb_bvh_build", "rtb_fill_image", "rtb_write_ppm", "rtb_write_jpeg", "rtb_image_load",
    "rtb_last_error", "rtb_abi_sizes",
    "rt_debug_math_device", "rt_debug_rng_device", "rt_debug_scene_info", "rt_debug_trace_variant", "rt_debug_set_tuning", "rt_debug_set_engine", "rt_debug_census", "rt_debug_pass_timing", "rt_debug_traffic_probe", "rt_debug_valu_probe", "rt_debug_set_partial_ring", "rt_debug_f32_slabs",
]

_lib = None


def lib():
    """The product library. Raises if it was not built: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "raytracer_2022_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C raytracer_2022_amd/csrc`). The path-tracing hot loop only exists as HIP kernels." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_double
    P = C.POINTER
    L.rt_scene_create.argtypes = [P(rt_scene_desc), P(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_render.argtypes = [vp, P(rt_camera), P(rt_params), P(dbl), P(rt_stats)]
    L.rt_render_device.argtypes = [vp, P(rt_camera), P(rt_params), vp, vp, P(rt_stats)]
    L.rt_render_wait.argtypes = [vp, vp]
    L.rt_write_color.argtypes = [P(dbl), i32, P(C.c_uint8)]
    L.rt_write_color.restype = None
    L.rt_tonemap_device.argtypes = [vp, u64, i32, vp, vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_abi_version.restype = u32
    L.rt_scene_set_create.argtypes = [P(rt_scene_desc), u64, P(vp)]
    L.rt_scene_set_destroy.argtypes = [vp]
    L.rt_render_multi.argtypes = [vp, P(rt_camera), P(rt_params), P(dbl), P(rt_stats)]
    L.rtb_scene_build.argtypes = [C.c_char_p, u64, C.c_char_p, i32, P(vp)]
    L.rtb_scene_free.argtypes = [vp]
    L.rtb_scene_free.restype = None
    L.rtb_scene_desc.argtypes = [vp]
    L.rtb_scene_desc.restype = P(rt_scene_desc)
    L.rtb_scene_default_view.argtypes = [vp, dbl, P(rt_camera), P(dbl)]
    L.rtb_camera_new.argtypes = [P(dbl), P(dbl), P(dbl), dbl, dbl, dbl, dbl, dbl, dbl, P(rt_camera)]
    L.rtb_shuffled_rows.argtypes = [u32, u64, P(u32)]
    L.rtb_bvh_build.argtypes = [P(u32), P(dbl), u32, u64, P(rt_bvh_node), u32]
    L.rtb_fill_image.argtypes = [P(dbl), P(u32), u32, u32, u32, i32, P(C.c_uint8)]
    L.rtb_write_ppm.argtypes = [C.c_char_p, P(C.c_uint8), u32, u32]
    L.rtb_write_jpeg.argtypes = [C.c_char_p, P(C.c_uint8), u32, u32, i32]
    L.rtb_image_load.argtypes = [C.c_char_p, P(u32), P(u32), P(C.c_uint8), u64]
    L.rtb_last_error.restype = C.c_char_p
    L.rtb_abi_sizes.argtypes = [P(u32), u32]
    L.rt_debug_math_device.argtypes = [C.c_int, P(dbl), P(dbl), P(dbl), u64]
    L.rt_debug_rng_device.argtypes = [u64, C.c_int, dbl, dbl, u64, P(u64), u64]
    L.rt_debug_scene_info.argtypes = [vp, P(u32), P(i32)]
    L.rt_debug_trace_variant.argtypes = [vp, P(u32), P(u32), P(u32), P(u32)]
    L.rt_debug_set_tuning.argtypes = [vp, u32, u32]
    L.rt_debug_set_engine.argtypes = [vp, C.c_int, C.c_int]
    L.rt_debug_census.argtypes = [vp, P(u64), P(u64)]
    L.rt_debug_pass_timing.argtypes = [vp, P(dbl)]
    L.rt_debug_traffic_probe.argtypes = [C.c_int, u64, u64, u64]
    L.rt_debug_valu_probe.argtypes = [C.c_int, u32]
    L.rt_debug_set_partial_ring.argtypes = [vp, C.c_int]
    L.rt_debug_f32_slabs.argtypes = [P(u64)]
    _lib = L
    return L


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rt2022 error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc < 0:
        raise RtError(rc, (lib().rt_last_error() or b"").decode())
    return rc
