"""Host side: scenes, cameras, row lists, image output (include/rt2022_host.h).

Mirrors what raytracer/src/main.rs:43-99 and :191-221 do around the render loop.
"""
import ctypes as C

import numpy as np

from . import _ffi as F


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


def camera_new(lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, time0, time1):
    """Camera::new, basic/camera.rs:24-62."""
    cam = F.rt_camera()
    F.check(F.lib().rtb_camera_new(_d3(lookfrom), _d3(lookat), _d3(vup), vfov, aspect_ratio, aperture, focus_dist,
                                   time0, time1, C.byref(cam)))
    return cam


def shuffled_rows(image_height, seed):
    """main.rs:93-99: Fisher-Yates shuffled line ids."""
    out = np.zeros(image_height, dtype=np.uint32)
    F.check(F.lib().rtb_shuffled_rows(image_height, seed, out.ctypes.data_as(C.POINTER(C.c_uint32))))
    return out


def write_color(rgb_sum, spp):
    """write_color, main.rs:280-299, over an (..., 3) array of sums."""
    sums = np.ascontiguousarray(rgb_sum, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((sums.shape[0], 3), dtype=np.uint8)
    L = F.lib()
    for i in range(sums.shape[0]):
        L.rt_write_color(sums[i].ctypes.data_as(C.POINTER(C.c_double)), spp,
                         out[i].ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(np.shape(rgb_sum))


def fill_image(rgb_sum, row_ids, width, height, spp):
    """main.rs:191-201: un-shuffle, flip, tone-map → (height, width, 3) uint8."""
    sums = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    rows = np.ascontiguousarray(row_ids, dtype=np.uint32)
    img = np.zeros((height, width, 3), dtype=np.uint8)
    F.check(F.lib().rtb_fill_image(sums.ctypes.data_as(C.POINTER(C.c_double)), rows.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   len(rows), width, height, spp, img.ctypes.data_as(C.POINTER(C.c_uint8))))
    return img


def load_image(path):
    """ImageTexture::new's decode step (texture/mod.rs:90-93): baseline JPEG or binary PPM → (h, w, 3) uint8, rows top-down."""
    w, h = C.c_uint32(), C.c_uint32()
    F.check(F.lib().rtb_image_load(path.encode(), C.byref(w), C.byref(h), None, 0))
    img = np.empty((h.value, w.value, 3), dtype=np.uint8)
    F.check(F.lib().rtb_image_load(path.encode(), C.byref(w), C.byref(h), img.ctypes.data_as(C.POINTER(C.c_uint8)), img.size))
    return img


def write_jpeg(path, rgb8, quality=100):
    """main.rs:213-221: the finished image as a baseline JPEG (IMAGE_QUALITY = 100 in the reference)."""
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    F.check(F.lib().rtb_write_jpeg(path.encode(), img.ctypes.data_as(C.POINTER(C.c_uint8)), img.shape[1], img.shape[0], quality))


def make_params(width, height, spp, max_depth=50, background=(0.0, 0.0, 0.0), seed=2022, n_frames=1,
                spp_chunk=0, flags=0, t_min=0.001):
    p = F.rt_params()
    p.width, p.height, p.spp, p.max_depth = width, height, spp, max_depth
    p.background = _d3(background)
    p.t_min = t_min
    p.seed = seed
    p.n_frames = n_frames
    p.spp_chunk = spp_chunk
    p.flags = flags
    return p


class HostScene:
    """scene::<name>() + BvhNode::new_list (main.rs:89-90) → flattened rt_scene_desc."""

    def __init__(self, name, seed=2022, assets_dir=None, param=0):
        self._h = C.c_void_p()
        self.name = name
        F.check(F.lib().rtb_scene_build(name.encode(), seed, (assets_dir or "").encode(), param, C.byref(self._h)))

    @property
    def desc(self):
        d = F.lib().rtb_scene_desc(self._h).contents
        d._owner = self          # the pools live inside this scene: keep it alive with the view
        return d

    def default_view(self, aspect_ratio):
        cam = F.rt_camera()
        bg = (C.c_double * 3)()
        F.check(F.lib().rtb_scene_default_view(self._h, aspect_ratio, C.byref(cam), bg))
        return cam, (bg[0], bg[1], bg[2])

    def close(self):
        if self._h:
            F.lib().rtb_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DescBuilder:
    """Assembles a flattened rt_scene_desc by hand (pool by pool). Used for small
    known-answer scenes; real scenes come from HostScene."""

    POOLS = [("nodes", F.rt_bvh_node), ("spheres", F.rt_sphere), ("moving_spheres", F.rt_moving_sphere),
             ("rects", F.rt_rect), ("boxes", F.rt_box), ("triangles", F.rt_triangle), ("rings", F.rt_ring),
             ("media", F.rt_medium), ("xforms", F.rt_xform), ("lists", F.rt_list), ("list_items", C.c_uint32),
             ("lights", C.c_uint32), ("materials", F.rt_material), ("textures", F.rt_texture),
             ("images", F.rt_image), ("perlins", F.rt_perlin)]

    def __init__(self):
        self.pools = {name: [] for name, _ in self.POOLS}
        self.image_data = bytearray()
        self.root = 0
        self._keep = []

    # -- textures / materials
    def solid(self, color):
        t = F.rt_texture(kind=F.RT_TEX_SOLID, color=_d3(color))
        self.pools["textures"].append(t)
        return len(self.pools["textures"]) - 1

    def checker(self, odd_tex, even_tex):
        self.pools["textures"].append(F.rt_texture(kind=F.RT_TEX_CHECKER, a=odd_tex, b=even_tex))
        return len(self.pools["textures"]) - 1

    def noise(self, perlin_id, scale):
        self.pools["textures"].append(F.rt_texture(kind=F.RT_TEX_NOISE, a=perlin_id, scale=scale))
        return len(self.pools["textures"]) - 1

    def perlin(self, randvec, perm_x, perm_y, perm_z):
        p = F.rt_perlin()
        for i in range(256):
            for k in range(3):
                p.randvec[i][k] = float(randvec[i][k])
            p.perm_x[i], p.perm_y[i], p.perm_z[i] = int(perm_x[i]), int(perm_y[i]), int(perm_z[i])
        self.pools["perlins"].append(p)
        return len(self.pools["perlins"]) - 1

    def image(self, rgb_bottom_up):
        """rgb_bottom_up: (h, w, 3) uint8 with row 0 = bottom row (texture/mod.rs:94-99 storage)."""
        arr = np.ascontiguousarray(rgb_bottom_up, dtype=np.uint8)
        h, w = arr.shape[0], arr.shape[1]
        self.pools["images"].append(F.rt_image(width=w, height=h, offset=len(self.image_data)))
        self.image_data += arr.tobytes()
        self.pools["textures"].append(F.rt_texture(kind=F.RT_TEX_IMAGE, a=len(self.pools["images"]) - 1))
        return len(self.pools["textures"]) - 1

    def _mat(self, kind, tex=0, albedo=(0, 0, 0), param=0.0):
        self.pools["materials"].append(F.rt_material(kind=kind, tex=tex, albedo=_d3(albedo), param=param))
        return len(self.pools["materials"]) - 1

    def lambertian(self, color=None, tex=None):
        return self._mat(F.RT_MAT_LAMBERTIAN, tex=self.solid(color) if tex is None else tex)

    def metal(self, albedo, fuzz):
        return self._mat(F.RT_MAT_METAL, albedo=albedo, param=fuzz if fuzz < 1.0 else 1.0)

    def dielectric(self, ir):
        return self._mat(F.RT_MAT_DIELECTRIC, param=ir)

    def diffuse_light(self, color=None, tex=None):
        return self._mat(F.RT_MAT_DIFFUSE_LIGHT, tex=self.solid(color) if tex is None else tex)

    def isotropic(self, color=None, tex=None):
        return self._mat(F.RT_MAT_ISOTROPIC, tex=self.solid(color) if tex is None else tex)

    # -- hittables (return refs)
    def _add(self, pool, kind, rec, flip=False):
        self.pools[pool].append(rec)
        return F.make_ref(kind, len(self.pools[pool]) - 1, flip)

    def sphere(self, center, radius, mat, flip=False):
        return self._add("spheres", F.RT_KIND_SPHERE, F.rt_sphere(center=_d3(center), radius=radius, mat=mat), flip)

    def moving_sphere(self, c0, c1, t0, t1, radius, mat, flip=False):
        return self._add("moving_spheres", F.RT_KIND_MOVING_SPHERE,
                         F.rt_moving_sphere(center0=_d3(c0), center1=_d3(c1), time0=t0, time1=t1, radius=radius, mat=mat), flip)

    def rect(self, axis, a0, a1, b0, b1, k, mat, flip=False):
        return self._add("rects", F.RT_KIND_RECT, F.rt_rect(a0=a0, a1=a1, b0=b0, b1=b1, k=k, axis=axis, mat=mat), flip)

    def box(self, p0, p1, mat, flip=False):
        return self._add("boxes", F.RT_KIND_BOX, F.rt_box(p0=_d3(p0), p1=_d3(p1), mat=mat), flip)

    def triangle(self, a, b, c, mat, flip=False):
        return self._add("triangles", F.RT_KIND_TRIANGLE, F.rt_triangle(a=_d3(a), b=_d3(b), c=_d3(c), mat=mat), flip)

    def ring(self, r, t, mat, flip=False):
        return self._add("rings", F.RT_KIND_RING,
                         F.rt_ring(r=r, t=t, dis_min=(r - t) * (r - t), dis_max=(r + t) * (r + t), mat=mat), flip)

    def medium(self, boundary_ref, density, iso_mat, flip=False):
        return self._add("media", F.RT_KIND_MEDIUM,
                         F.rt_medium(boundary=boundary_ref, mat=iso_mat, neg_inv_density=-1.0 / density), flip)

    def translate(self, child, offset, flip=False):
        return self._add("xforms", F.RT_KIND_TRANSLATE, F.rt_xform(kind=F.RT_KIND_TRANSLATE, child=child, p=_d3(offset)), flip)

    def rotate_y(self, child, sin_theta, cos_theta, flip=False):
        return self._add("xforms", F.RT_KIND_ROTATE_Y,
                         F.rt_xform(kind=F.RT_KIND_ROTATE_Y, child=child, p=_d3((sin_theta, cos_theta, 0.0))), flip)

    def zoom(self, child, rate, flip=False):
        return self._add("xforms", F.RT_KIND_ZOOM, F.rt_xform(kind=F.RT_KIND_ZOOM, child=child, p=_d3((rate, 0.0, 0.0))), flip)

    def node(self, bmin, bmax, left, right):
        return self._add("nodes", F.RT_KIND_NODE, F.rt_bvh_node(bmin=_d3(bmin), bmax=_d3(bmax), left=left, right=right))

    def list(self, refs):
        first = len(self.pools["list_items"])
        self.pools["list_items"] += [C.c_uint32(r) for r in refs]
        return self._add("lists", F.RT_KIND_LIST, F.rt_list(first=first, count=len(refs)))

    def light(self, ref):
        self.pools["lights"].append(C.c_uint32(ref))

    def set_root(self, ref):
        self.root = ref

    def desc(self):
        d = F.rt_scene_desc()
        d.abi_version = F.RT2022_ABI_VERSION
        d.root = self.root
        for name, ctype in self.POOLS:
            items = self.pools[name]
            arr = (ctype * max(1, len(items)))(*items)
            self._keep.append(arr)
            setattr(d, "n_" + name, len(items))
            setattr(d, name, C.cast(arr, C.POINTER(ctype)))
        data = (C.c_uint8 * max(1, len(self.image_data))).from_buffer_copy(bytes(self.image_data) or b"\0")
        self._keep.append(data)
        d.image_data_bytes = len(self.image_data)
        d.image_data = C.cast(data, C.POINTER(C.c_uint8))
        return d
