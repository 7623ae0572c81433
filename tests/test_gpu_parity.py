"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle and the
committed golden vectors. Bar: bit-exact — f64 pixel sums, u8 pixels, ray / node / primitive /
RNG-draw counters (the last prove that every path took the same branches)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from raytracer_2022_amd import _ffi as F

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
INDEX = json.load(open(os.path.join(HERE, "golden", "golden_index.json")))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_extension_is_loaded_and_device_visible(rt):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU: the HIP path has no fallback"
    assert os.path.exists(F.LIB_PATH)
    s = rt.HostScene("cornell_box")
    dev = rt.DeviceScene(s.desc)
    info = dev.info()
    assert info["stack_need"] >= 1 and info["grid_blocks"] > 0


@pytest.mark.parametrize("op,lo,hi", [(0, -1e4, 1e4), (1, -1e4, 1e4), (0, -1e12, 1e12), (2, -1.0, 1.0),
                                      (4, 0.0, 1.0), (4, 0.0, 1e300), (5, 0.0, 1e6), (5, 0.0, 1e-300)])
def test_device_math_is_bit_identical_to_host(rt, O, op, lo, hi):
    """rt_math.h compiled by hipcc for gfx950 == the same header compiled by g++ (sqrt and '/' included)."""
    x = np.random.default_rng(op + 11).uniform(lo, hi, 300_000)
    out = np.empty_like(x)
    F.check(rt.lib().rt_debug_math_device(op, x.ctypes.data_as(C.POINTER(C.c_double)), None,
                                          out.ctypes.data_as(C.POINTER(C.c_double)), x.size))
    assert np.array_equal(bits(out), bits(O.math_array(op, x)))


@pytest.mark.parametrize("op", [3, 6])
def test_device_binary_math_is_bit_identical(rt, O, op):
    rng = np.random.default_rng(5)
    a, b = rng.uniform(-3, 3, 300_000), rng.uniform(-3, 3, 300_000)
    out = np.empty_like(a)
    F.check(rt.lib().rt_debug_math_device(op, a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)),
                                          out.ctypes.data_as(C.POINTER(C.c_double)), a.size))
    assert np.array_equal(bits(out), bits(O.math_array(op, a, b)))


def test_device_rng_stream_matches_host(rt, O):
    n = 2048
    for mode, host in ((0, "u64"), (1, "f64"), (2, "range"), (3, "index")):
        out = (C.c_uint64 * n)()
        F.check(rt.lib().rt_debug_rng_device(777, mode, -2.5, 7.25, 11, out, n))
        if host == "u64":
            ref = (C.c_uint64 * n)(); O.lib().rto_rng_u64(777, ref, n); ref = np.array(ref[:], dtype=np.uint64)
        elif host == "f64":
            ref = (C.c_double * n)(); O.lib().rto_rng_f64(777, ref, n); ref = bits(np.array(ref[:]))
        elif host == "range":
            ref = (C.c_double * n)(); O.lib().rto_rng_range(777, -2.5, 7.25, ref, n); ref = bits(np.array(ref[:]))
        else:
            ref = (C.c_uint64 * n)(); O.lib().rto_rng_index(777, 11, ref, n); ref = np.array(ref[:], dtype=np.uint64)
        assert np.array_equal(np.array(out[:], dtype=np.uint64), ref), host


def golden_case(name, rt):
    c = INDEX[name]
    g = np.load(os.path.join(HERE, "golden", "golden_%s.npz" % name))
    s = rt.HostScene(c["scene"], seed=c["seed"], param=c["param"])
    cam, bg = s.default_view(c["width"] / c["height"])
    p = rt.make_params(c["width"], c["height"], c["spp"], c["max_depth"], bg, seed=c["seed"], n_frames=c["n_frames"],
                       spp_chunk=c["spp_chunk"])
    return c, g, s, cam, p


@pytest.mark.parametrize("engine", ["wavefront", "mega"])
@pytest.mark.parametrize("name", sorted(INDEX))
def test_hip_path_reproduces_golden(rt, name, engine):
    """Every scene builder of the reference (scene.rs) through the kernels, against the committed vectors."""
    c, g, s, cam, p = golden_case(name, rt)
    dev = rt.DeviceScene(s.desc)
    dev.set_engine(engine)
    out, st = dev.render(cam, p, g["rows"], want_stats=True)
    assert st.as_dict() == c["counters"], "the device paths took different branches than the oracle's"
    assert np.array_equal(bits(out), bits(g["rgb_sum"]))
    assert np.array_equal(rt.write_color(out, c["spp"]), g["rgb8"])
    # and without counters (the timed kernel variant)
    out2 = dev.render(cam, p, g["rows"])
    assert np.array_equal(bits(out2), bits(g["rgb_sum"]))


@pytest.mark.parametrize("scene,W,H,spp,chunk,frames", [
    ("final_scene", 96, 80, 6, 0, 1), ("final_scene", 64, 48, 9, 4, 3), ("cornell_box", 80, 80, 16, 5, 1),
    ("random_scene", 120, 80, 4, 1, 1), ("cornell_smoke", 64, 64, 6, 0, 2), ("wwscene", 96, 54, 2, 0, 1),
])
def test_hip_path_matches_oracle_on_larger_renders(rt, O, scene, W, H, spp, chunk, frames):
    s = rt.HostScene(scene, seed=31)
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=31, n_frames=frames, spp_chunk=chunk)
    rows = rt.shuffled_rows(H * frames, 4)
    ref, st_ref = O.render_cpu(s.desc, cam, p, rows, n_threads=os.cpu_count() or 4, want_stats=True)
    dev = rt.DeviceScene(s.desc)
    out, st = dev.render(cam, p, rows, want_stats=True)
    assert st.as_dict() == st_ref.as_dict()
    assert np.array_equal(bits(out), bits(ref))
    assert np.array_equal(rt.write_color(out, spp), O.write_color(ref, spp))


def test_hand_built_scene_with_every_object_kind(rt, O):
    """One of each hittable kind (incl. triangle, ring, list, zoom, flipped refs, an image texture, a sphere
    light) through both engines: the arms no named scene combines."""
    b = rt.DescBuilder()
    lam = b.lambertian((0.6, 0.5, 0.4))
    img = (np.arange(8 * 4 * 3, dtype=np.uint8).reshape(4, 8, 3) * 7) % 251
    refs = [
        b.sphere((0, -100, 0), 100.0, b.lambertian(tex=b.checker(b.solid((0.2, 0.3, 0.1)), b.solid((0.9, 0.9, 0.9))))),
        b.sphere((0, 1, 0), 1.0, b.lambertian(tex=b.image(img))),
        b.moving_sphere((2.5, 0.5, 0), (2.5, 1.0, 0), 0, 1, 0.5, b.metal((0.8, 0.7, 0.6), 0.3)),
        b.sphere((-2.5, 1, 0), 1.0, b.dielectric(1.5)),
        b.triangle((-1, 0.01, 2), (1, 0.01, 2), (0, 1.5, 2.5), lam),
        b.translate(b.ring(1.5, 0.3, lam), (0, 0.5, -3)),
        b.translate(b.rotate_y(b.zoom(b.box((-0.5, 0, -0.5), (0.5, 1, 0.5), lam), 1.5), 0.5, 0.8660254037844386), (4.5, 0, 2)),
        b.medium(b.sphere((-4, 1, 2), 1.0, b.dielectric(1.5)), 0.8, b.isotropic((0.3, 0.3, 0.9))),
        b.list([b.rect(F.RT_RECT_XZ, -1, 1, -1, 1, 6.0, b.diffuse_light((8, 8, 8)), flip=True),
                b.rect(F.RT_RECT_XY, -6, 6, 0, 4, -6.0, lam)]),
    ]
    light_sphere = b.sphere((5, 6, -2), 0.7, b.diffuse_light((20, 18, 15)))
    refs.append(light_sphere)
    b.light(light_sphere)
    b.light(F.make_ref(F.RT_KIND_RECT, 0))
    b.set_root(b.list(refs))
    d = b.desc()
    W, H, spp = 56, 40, 5
    cam = rt.camera_new((9, 4, 9), (0, 1, 0), (0, 1, 0), 35.0, W / H, 0.05, 12.0, 0.0, 1.0)
    p = rt.make_params(W, H, spp, 20, (0.05, 0.06, 0.1), seed=99)
    rows = np.arange(H, dtype=np.uint32)
    ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=8, want_stats=True)
    assert all(st_ref.prim_tests[k] > 0 for k in range(1, F.RT_KIND_COUNT)), "scene does not reach every kind"
    for engine in ("wavefront", "mega"):
        dev = rt.DeviceScene(d)
        dev.set_engine(engine)
        out, st = dev.render(cam, p, rows, want_stats=True)
        assert st.as_dict() == st_ref.as_dict(), engine
        assert np.array_equal(np.isnan(out), np.isnan(ref)) and np.array_equal(bits(out), bits(ref)), engine


def test_deep_stacks_select_the_larger_kernels(rt, O):
    """Traversal-stack variants: 22 entries (default), 30 (million-triangle meshes), 64 (anything deeper)."""
    # a quarter-million-triangle mesh under three movers: needs the 30-entry stack
    s = rt.HostScene("wwscene", seed=2022, param=2)
    dev = rt.DeviceScene(s.desc)
    assert 22 < dev.info()["stack_need"] <= 30
    W, H, spp = 64, 36, 2
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=11)
    rows = rt.shuffled_rows(H, 5)
    ref, st_ref = O.render_cpu(s.desc, cam, p, rows, n_threads=os.cpu_count() or 4, want_stats=True)
    out, st = dev.render(cam, p, rows, want_stats=True)
    assert st.as_dict() == st_ref.as_dict() and np.array_equal(bits(out), bits(ref))
    assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref))
    # a 45-deep left-leaning chain of nodes: needs the 64-entry stack
    b = rt.DescBuilder()
    lam = b.lambertian((0.6, 0.6, 0.7))
    chain = b.sphere((0, 0, -60), 1.0, lam)
    for i in range(45):
        chain = b.node((-50, -50, -100), (50, 50, 10), chain, b.sphere((-8 + 0.35 * i, 0.1 * i - 2, -20 - i), 0.8, lam))
    b.set_root(chain)
    d = b.desc()
    dev = rt.DeviceScene(d)
    assert dev.info()["stack_need"] > 30
    cam = rt.camera_new((0, 0, 10), (0, 0, -30), (0, 1, 0), 50.0, 1.5, 0.0, 10.0, 0.0, 1.0)
    p = rt.make_params(48, 32, 3, 20, (0.6, 0.7, 0.9), seed=2)
    rows = np.arange(32, dtype=np.uint32)
    ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=4, want_stats=True)
    out, st = dev.render(cam, p, rows, want_stats=True)
    assert st.as_dict() == st_ref.as_dict() and np.array_equal(bits(out), bits(ref))
    assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref))
    # deeper than 64: refused, not truncated
    for i in range(30):
        chain = b.node((-50, -50, -100), (50, 50, 10), chain, b.sphere((0, 0, -20), 0.8, lam))
    b.set_root(chain)
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(b.desc())
    assert e.value.code == F.RT_ERR_UNSUPPORTED


ASSETS = os.path.join(os.path.dirname(HERE), "assets")


@pytest.mark.parametrize("with_assets", [True, False])
def test_c5_mesh_at_its_benchmark_size(rt, O, with_assets):
    """BASELINE config 5 at the mesh size `bench.py --config c5` times: wwscene, param=3 — the reference's Shuttle.obj
    (assets/) subdivided three times, 837 056 + 2 048 triangles, with the planets' JPEG textures; and the
    1.05 M-triangle synthetic stand-in used when no assets directory is given — each in a ~20-deep BVH under
    Translate<RotateY<Zoom<…>>> (scene.rs:408-412), on a small image the oracle finishes in a second. Counters and bits,
    with and without RT_FLAG_COUNTERS, running sum and one-sample items."""
    if with_assets and not os.path.isdir(ASSETS):
        pytest.skip("assets/ not present")
    s = rt.HostScene("wwscene", seed=2022, param=3, assets_dir=ASSETS if with_assets else None)
    assert s.desc.n_triangles == (837056 + 2048 if with_assets else 1050624) and s.desc.n_nodes > 800_000
    dev = rt.DeviceScene(s.desc)
    assert 22 < dev.info()["stack_need"] <= 30
    W, H, spp = 96, 54, 2
    cam, bg = s.default_view(W / H)
    rows = rt.shuffled_rows(H, 3)
    for chunk in (0, 1):
        p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=chunk)
        ref, st_ref = O.render_cpu(s.desc, cam, p, rows, n_threads=os.cpu_count() or 4, want_stats=True)
        assert st_ref.prim_tests[F.RT_KIND_TRIANGLE] > 100_000 and st_ref.prim_tests[F.RT_KIND_ZOOM] > 0
        out, st = dev.render(cam, p, rows, want_stats=True)
        assert st.as_dict() == st_ref.as_dict(), chunk
        assert np.array_equal(bits(out), bits(ref)), chunk
        assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref)), chunk       # the timed kernel variant
        assert np.array_equal(rt.write_color(out, spp), O.write_color(ref, spp))


@pytest.mark.parametrize("scene,W,H,spp,param", [("earth", 64, 40, 4, 0), ("final_scene", 72, 72, 4, 0), ("wwscene", 80, 45, 3, 0)])
def test_scenes_with_the_committed_assets(rt, O, scene, W, H, spp, param):
    """The builders reading the reference's own input files (assets/: four JPEG textures through host/jpeg.cpp, Shuttle.obj
    through the OBJ reader) instead of the procedural stand-ins: what `bench.py` times. ImageTexture::value on real texels
    (texture/mod.rs:110-139), 13 079 real triangles (scene.rs:364-414)."""
    if not os.path.isdir(ASSETS):
        pytest.skip("assets/ not present")
    s = rt.HostScene(scene, seed=2022, assets_dir=ASSETS, param=param)
    assert s.desc.n_images >= 1 and s.desc.image_data_bytes >= 800 * 383 * 3
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022)
    rows = rt.shuffled_rows(H, 9)
    ref, st_ref = O.render_cpu(s.desc, cam, p, rows, n_threads=os.cpu_count() or 4, want_stats=True)
    for engine in ("wavefront", "mega"):
        dev = rt.DeviceScene(s.desc)
        dev.set_engine(engine)
        out, st = dev.render(cam, p, rows, want_stats=True)
        assert st.as_dict() == st_ref.as_dict(), engine
        assert np.array_equal(bits(out), bits(ref)), engine
        assert np.array_equal(rt.write_color(out, spp), O.write_color(ref, spp))


def test_node_step_variants(rt, O):
    """The traversal's short node step (min / max slabs, one exit test) is only taken where it provably
    equals AABB::hit (aabb.rs:15-32); everything else runs the literal restatement. Both against the oracle."""
    # 1. literal step forced on a scene that normally takes the short one
    c, g, s, cam, p = golden_case("final_scene", rt)
    dev = rt.DeviceScene(s.desc)
    dev.set_tuning(18 | (1 << 8) | (2 << 12) | (6 << 16) | (2 << 20) | (1 << 24) | (1 << 30))
    out, st = dev.render(cam, p, g["rows"], want_stats=True)
    assert st.as_dict() == c["counters"] and np.array_equal(bits(out), bits(g["rgb_sum"]))
    assert np.array_equal(bits(dev.render(cam, p, g["rows"])), bits(g["rgb_sum"]))

    def grid_scene(inf_root=False, inverted=False):
        b = rt.DescBuilder()
        lam = b.lambertian((0.7, 0.6, 0.5))
        met = b.metal((0.8, 0.8, 0.9), 0.0)
        # unit spheres on integer centres: box faces fall on the integer planes the camera rays start in
        leaves = []
        for i, (x, y, z) in enumerate([(1, 0, -6), (-1, 0, -6), (0, 3, -7), (0, -2, -5), (3, 1, -8), (-3, -1, -4), (0, 1, -10)]):
            sp = b.sphere((x, y, z), 1.0, met if i % 2 else lam)
            leaves.append((sp, (x - 1, y - 1, z - 1), (x + 1, y + 1, z + 1)))
        floor = b.rect(F.RT_RECT_XZ, -20, 20, -20, 20, -3.0, lam)
        leaves.append((floor, (-20, -3.0001, -20), (20, -2.9999, 20)))
        lamp = b.rect(F.RT_RECT_XZ, -2, 2, -8, -4, 6.0, b.diffuse_light((9, 9, 9)), flip=True)
        leaves.append((lamp, (-2, 5.9999, -8), (2, 6.0001, -4)))
        b.light(F.make_ref(F.RT_KIND_RECT, 1))         # (the light list holds the plain rect, as scene.rs does)
        def build(items):
            if len(items) == 1:
                r, lo, hi = items[0]
                return b.node(lo, hi, r, r), lo, hi
            h = len(items) // 2
            l, llo, lhi = build(items[:h])
            r, rlo, rhi = build(items[h:])
            lo = tuple(min(a, c_) for a, c_ in zip(llo, rlo)); hi = tuple(max(a, c_) for a, c_ in zip(lhi, rhi))
            return b.node(lo, hi, l, r), lo, hi
        root, lo, hi = build(leaves)
        if inf_root:
            root = b.node((-np.inf,) * 3, (np.inf,) * 3, root, root)
        if inverted:                                   # a box with min > max on x: never hit, by either formulation of the test
            dead = b.sphere((0, 0, -3), 0.5, lam)
            root = b.node(lo, hi, root, b.node((1, -1, -4), (-1, 1, -2), dead, dead))
        b.set_root(root)
        return b.desc()

    W, H, spp = 40, 30, 6
    rows = np.arange(H, dtype=np.uint32)
    p = rt.make_params(W, H, spp, 12, (0.3, 0.4, 0.6), seed=5)
    ordinary = rt.camera_new((0.3, 0.2, 4), (0, 0, -6), (0, 1, 0), 50.0, W / H, 0.0, 10.0, 0.0, 1.0)
    # 2. every camera ray is (0, 0, -1) from an integer point: 1/d = +-inf on two axes and 0 * inf = NaN at the
    #    box faces through the origin's coordinates — AABB::hit's comparisons with NaN, kept literally
    axis = F.rt_camera.from_buffer_copy(ordinary)
    for k, v in (("origin", (0.0, 1.0, 4.0)), ("lower_left_corner", (0.0, 1.0, 3.0)), ("horizontal", (0.0, 0.0, 0.0)),
                 ("vertical", (0.0, 0.0, 0.0))):
        setattr(axis, k, (C.c_double * 3)(*v))
    axis.lens_radius = 0.0
    for label, d, cam in (("plain boxes", grid_scene(), ordinary), ("axis-parallel rays", grid_scene(), axis),
                          ("infinite root box", grid_scene(inf_root=True), ordinary),
                          ("infinite root box, axis-parallel rays", grid_scene(inf_root=True), axis),
                          ("inverted box", grid_scene(inverted=True), ordinary)):
        ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=4, want_stats=True)
        for engine in ("wavefront", "mega"):
            dev = rt.DeviceScene(d)
            dev.set_engine(engine)
            out, st = dev.render(cam, p, rows, want_stats=True)
            assert st.as_dict() == st_ref.as_dict(), (label, engine)
            assert np.array_equal(bits(out), bits(ref)), (label, engine)
            assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref)), (label, engine)
    assert st_ref.node_visits > 0


def test_single_precision_slab_test_on_hostile_spheres(rt, O):
    """The traversal kernel of sphere-only scenes tests node boxes in single precision and hands a step to the double-precision
    test wherever the two could differ (wf_trace, t_slabs32: the error bound is derived there). Scenes built to sit on that
    edge — box faces through the ray origins, coordinates where a float's spacing exceeds the sphere, a 1e-6 sphere on a 1e6
    one, directions whose reciprocal leaves the float range — against the oracle, bit for bit, pixels and counters."""
    out5 = (C.c_uint64 * 5)()
    F.check(F.lib().rt_debug_f32_slabs(out5))
    assert out5[3] == 1, "this library is built without the single-precision slab test of the sphere-only kernel"

    def spheres_scene(items):
        b = rt.DescBuilder()
        mats = [b.lambertian((0.7, 0.6, 0.5)), b.metal((0.8, 0.8, 0.9), 0.0), b.dielectric(1.5)]
        leaves = []
        for i, (c, r) in enumerate(items):
            sp = b.sphere(c, r, mats[i % 3])
            leaves.append((sp, tuple(x - abs(r) for x in c), tuple(x + abs(r) for x in c)))

        def build(nodes):
            if len(nodes) == 1:
                r, lo, hi = nodes[0]
                return b.node(lo, hi, r, r), lo, hi
            h = len(nodes) // 2
            l, llo, lhi = build(nodes[:h])
            r, rlo, rhi = build(nodes[h:])
            lo = tuple(min(a, c_) for a, c_ in zip(llo, rlo)); hi = tuple(max(a, c_) for a, c_ in zip(lhi, rhi))
            return b.node(lo, hi, l, r), lo, hi
        root, _, _ = build(leaves)
        b.set_root(root)
        return b.desc()

    W, H, spp = 48, 32, 8
    rows = np.arange(H, dtype=np.uint32)
    p = rt.make_params(W, H, spp, 16, (0.6, 0.7, 0.9), seed=9)
    rng = np.random.default_rng(4)
    # unit spheres on integer centres seen from an integer point: box faces pass through ray origins (hits restart ON the spheres' boxes' faces at the poles)
    lattice = [((float(x), float(y), float(z)), 1.0) for x in (-2, 0, 2) for y in (-2, 0, 2) for z in (-8, -6)]
    # a cluster 1e7 away: a float there is spaced 1 apart, the spheres are 0.5 across
    far = [((1e7 + float(dx), 1e7 + float(dy), -1e7 + float(dz)), 0.25) for dx, dy, dz in rng.uniform(-3, 3, size=(14, 3))]
    # a 1e-6 sphere resting on a 1e6 one, and neighbours of ordinary size
    scales = [((0.0, -1e6, -5.0), 1e6), ((0.0, 1e-6, -5.0), 1e-6), ((0.5, 0.5, -5.0), 0.5), ((-0.7, 0.3, -4.0), 0.3)]
    cams = {
        "lattice": rt.camera_new((0.0, 0.0, 4.0), (0.0, 0.0, -7.0), (0, 1, 0), 40.0, W / H, 0.0, 10.0, 0.0, 1.0),
        "far": rt.camera_new((1e7 + 0.5, 1e7 + 1.0, -1e7 + 12.0), (1e7, 1e7, -1e7), (0, 1, 0), 40.0, W / H, 0.0, 10.0, 0.0, 1.0),
        "scales": rt.camera_new((0.0, 1e-5, -4.99), (0.0, 1e-6, -5.0), (0, 1, 0), 30.0, W / H, 0.0, 1.0, 0.0, 1.0),
    }
    # directions with one component of 1e-42 (1/d = 1e42: outside the float range, finite in double): every step of such a ray is the double-precision test's
    thin = F.rt_camera.from_buffer_copy(cams["lattice"])
    thin.origin = (C.c_double * 3)(0.25, 0.125, 4.0)
    thin.lower_left_corner = (C.c_double * 3)(0.25 - 2.0, 0.125 + 1e-42, 3.0)
    thin.horizontal = (C.c_double * 3)(4.0, 0.0, 0.0)
    thin.vertical = (C.c_double * 3)(0.0, 0.0, 0.0)
    thin.lens_radius = 0.0
    for label, items, cam in (("lattice", lattice, cams["lattice"]), ("far cluster", far, cams["far"]), ("scales", scales, cams["scales"]),
                              ("1/d beyond float", lattice, thin)):
        d = spheres_scene(items)
        ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=4, want_stats=True)
        dev = rt.DeviceScene(d)
        v = dev.trace_variant()
        assert v["spheres_in_lds"] and v["f32_slabs"] and v["nodes_in_lds"] == d.n_nodes, (label, v)     # (the instance with the single-precision test)
        out, st = dev.render(cam, p, rows, want_stats=True)
        assert st.as_dict() == st_ref.as_dict(), label
        assert np.array_equal(bits(out), bits(ref)), label
        assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref)), label       # (the timed kernel: the one that holds the test)
        assert st_ref.node_visits > 0 and st_ref.rays > W * H * spp, label
        # ... and the plain kernel of sphere scenes (what a scene too large for LDS takes): the same test on 32-byte records from L2 / HBM
        dev.set_tuning(18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24) | (1 << 28))
        v2 = dev.trace_variant()
        assert v2["nodes_in_lds"] == 0 and v2["f32_slabs"], (label, v2)
        assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref)), label


def _median_split_bvh(b, leaves):
    def build(items, axis=0):
        if len(items) == 1:
            r, lo, hi = items[0]
            return b.node(lo, hi, r, r), lo, hi
        items = sorted(items, key=lambda it: it[1][axis])
        h = len(items) // 2
        l, llo, lhi = build(items[:h], (axis + 1) % 3)
        r, rlo, rhi = build(items[h:], (axis + 1) % 3)
        lo = tuple(min(a, c_) for a, c_ in zip(llo, rlo)); hi = tuple(max(a, c_) for a, c_ in zip(lhi, rhi))
        return b.node(lo, hi, l, r), lo, hi
    return build(leaves)[0]


def test_mesh_takes_the_single_precision_records(rt, O):
    """A triangle mesh (8 191 nodes; a quarter of the triangles axis-aligned, lying IN faces of their boxes, where the float test decides
    least): the plain mesh kernel on 32-byte single-precision node records, the double-precision record for the steps it leaves
    undecided. Pixels and counters against the oracle."""
    rng = np.random.default_rng(12)
    b = rt.DescBuilder()
    mats = [b.lambertian((0.7, 0.5, 0.4)), b.metal((0.8, 0.8, 0.8), 0.1)]
    leaves = []
    n = 4096
    for i in range(n):
        c = rng.uniform(-4.0, 4.0, 3) + np.array([0.0, 0.0, -12.0])
        e1, e2 = rng.uniform(-0.25, 0.25, 3), rng.uniform(-0.25, 0.25, 3)
        if i % 4 == 0:                                   # an axis-aligned triangle in the plane z = const
            e1[2] = 0.0; e2[2] = 0.0
        pa, pb, pc = c, c + e1, c + e2
        lo = np.minimum(np.minimum(pa, pb), pc) - 1e-4; hi = np.maximum(np.maximum(pa, pb), pc) + 1e-4
        leaves.append((b.triangle(tuple(pa), tuple(pb), tuple(pc), mats[i % 2]), tuple(lo), tuple(hi)))
    b.set_root(_median_split_bvh(b, leaves))
    d = b.desc()
    assert d.n_nodes == 2 * n - 1                        # (4 095 inner nodes over a span-1 node per triangle)
    W, H, spp = 64, 48, 4
    cam = rt.camera_new((0.0, 0.5, 2.0), (0.0, 0.0, -12.0), (0, 1, 0), 40.0, W / H, 0.0, 10.0, 0.0, 1.0)
    p = rt.make_params(W, H, spp, 12, (0.7, 0.8, 1.0), seed=21)
    rows = np.arange(H, dtype=np.uint32)
    dev = rt.DeviceScene(d)
    v = dev.trace_variant()
    assert v["workgroup_threads"] == 256 and v["nodes_in_lds"] == 0 and v["f32_slabs"], v
    ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=4, want_stats=True)
    out, st = dev.render(cam, p, rows, want_stats=True)
    assert st.as_dict() == st_ref.as_dict()
    assert np.array_equal(bits(out), bits(ref))
    assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref))                   # the timed kernel
    assert st_ref.prim_tests[F.RT_KIND_TRIANGLE] > 10_000 and st_ref.rays > W * H * spp


def test_partial_node_table_of_a_scene_with_boxes(rt, O):
    """A scene with a `Boxes` in it keeps the double-precision node test; with 8 193 nodes under stacks of 16 it takes the partial-table
    kernel: the first 1 740 records (the top of the BVH: the device copy numbers the nodes breadth-first) in LDS, the rest from L2 / HBM.
    Against the plain kernels and the oracle."""
    rng = np.random.default_rng(13)
    b = rt.DescBuilder()
    mats = [b.lambertian((0.6, 0.6, 0.7)), b.metal((0.9, 0.8, 0.7), 0.0), b.dielectric(1.5)]
    leaves = []
    n = 4096
    for i in range(n):
        c = rng.uniform(-5.0, 5.0, 3) + np.array([0.0, 0.0, -14.0])
        r = float(rng.uniform(0.05, 0.2))
        leaves.append((b.sphere(tuple(c), r, mats[i % 3]), tuple(c - r), tuple(c + r)))
    leaves.append((b.box((-1.0, -1.0, -9.0), (1.0, 1.0, -8.0), mats[0]), (-1.0, -1.0, -9.0), (1.0, 1.0, -8.0)))
    b.set_root(_median_split_bvh(b, leaves))
    d = b.desc()
    W, H, spp = 64, 48, 4
    cam = rt.camera_new((0.0, 0.5, 2.0), (0.0, 0.0, -14.0), (0, 1, 0), 40.0, W / H, 0.0, 10.0, 0.0, 1.0)
    p = rt.make_params(W, H, spp, 12, (0.7, 0.8, 1.0), seed=22)
    rows = np.arange(H, dtype=np.uint32)
    dev = rt.DeviceScene(d)
    v = dev.trace_variant()
    assert v["workgroup_threads"] == 1024 and v["nodes_in_lds"] == 1740 < d.n_nodes and not v["f32_slabs"], v
    ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=4, want_stats=True)
    out, st = dev.render(cam, p, rows, want_stats=True)
    assert st.as_dict() == st_ref.as_dict() and np.array_equal(bits(out), bits(ref))
    a = dev.render(cam, p, rows)
    assert np.array_equal(bits(a), bits(ref))                                          # the timed kernel: partial table
    dev.set_tuning(18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24) | (1 << 28))
    assert dev.trace_variant()["nodes_in_lds"] == 0
    assert np.array_equal(bits(dev.render(cam, p, rows)), bits(ref))
    assert st_ref.prim_tests[F.RT_KIND_BOX] > 0


def test_rect_scenes_keep_the_double_precision_node_test(rt):
    """A rect lies in the faces of its node's box — where the single-precision test decides nothing: a FEAT-0 scene with
    rects takes the double-precision kernels, in LDS and in the plain variant alike."""
    c, g, s, cam, p = golden_case("cornell_box", rt)
    dev = rt.DeviceScene(s.desc)
    v = dev.trace_variant()
    assert v["nodes_in_lds"] == s.desc.n_nodes and not v["f32_slabs"] and not v["spheres_in_lds"], v
    a = dev.render(cam, p, g["rows"])
    dev.set_tuning(18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24) | (1 << 28))
    v = dev.trace_variant()
    assert v["nodes_in_lds"] == 0 and not v["f32_slabs"], v
    assert np.array_equal(bits(a), bits(dev.render(cam, p, g["rows"]))) and np.array_equal(bits(a), bits(g["rgb_sum"]))


def test_scheduling_knobs_never_change_results(rt, O):
    """Segments per traversal workgroup, stream groups, pacing, the timing probe: speed only (rt2022_debug.h)."""
    def word(q=18, reps=1, tail=2, segs=4, shift=2, groups=1, extra=0):
        return q | (reps << 8) | (tail << 12) | (segs << 16) | (shift << 20) | (groups << 24) | extra
    for name in ("final_scene", "cornell_smoke"):
        c, g, s, cam, p = golden_case(name, rt)
        dev = rt.DeviceScene(s.desc)
        for w in (word(segs=1), word(segs=8, groups=3), word(groups=8, shift=0), word(q=64, tail=15, extra=1 << 28),
                  word(q=1, reps=3, extra=1 << 29), word(segs=3, groups=2, extra=(1 << 28) | (1 << 30))):
            dev.set_tuning(w)
            out, st = dev.render(cam, p, g["rows"], want_stats=True)
            assert st.as_dict() == c["counters"], (name, hex(w))
            assert np.array_equal(bits(out), bits(g["rgb_sum"])), (name, hex(w))
            assert np.array_equal(bits(dev.render(cam, p, g["rows"])), bits(g["rgb_sum"])), (name, hex(w))
            if w & (1 << 29):
                t = dev.pass_timing()
                assert t["passes"] > 0 and 0 < t["wave_life_ms"] <= t["span_ms"] and t["wave_dry_ms"] <= t["wave_life_ms"]


def test_one_sample_per_item_is_the_running_sum(rt, O):
    """spp_chunk = 1 adds 0 + L0 + L1 + ... like spp_chunk = 0 (main.rs:144-151): same bits, finest work items."""
    for scene, W, H, spp in (("cornell_box", 40, 40, 7), ("final_scene", 48, 32, 5)):
        s = rt.HostScene(scene, seed=9)
        cam, bg = s.default_view(W / H)
        rows = rt.shuffled_rows(H, 2)
        dev = rt.DeviceScene(s.desc)
        outs = [dev.render(cam, rt.make_params(W, H, spp, 50, bg, seed=9, spp_chunk=k), rows) for k in (0, 1)]
        assert np.array_equal(bits(outs[0]), bits(outs[1])), scene
        ref = O.render_cpu(s.desc, cam, rt.make_params(W, H, spp, 50, bg, seed=9, spp_chunk=1), rows, n_threads=4)
        assert np.array_equal(bits(outs[1]), bits(ref)), scene


def test_medium_with_a_composite_boundary(rt, O):
    """ConstantMedium<H> for any H (constantmedium.rs:14-22): a boundary that is a BVH of a box and a sphere.
    The two boundary queries run through the ordinary traversal arms with their own t window."""
    b = rt.DescBuilder()
    glass = b.dielectric(1.5)
    shell = b.node((-3, -3, -8), (3, 3, -2), b.box((-2, -1, -7), (0.5, 1, -4), glass), b.sphere((1.0, 0, -5), 1.5, glass))
    fog = b.medium(shell, 0.9, b.isotropic((0.8, 0.5, 0.3)))
    floor_ = b.rect(F.RT_RECT_XZ, -20, 20, -20, 20, -1.5, b.lambertian((0.5, 0.5, 0.5)))
    light = b.rect(F.RT_RECT_XZ, -2, 2, -7, -3, 6.0, b.diffuse_light((9, 9, 9)), flip=True)
    b.set_root(b.list([fog, floor_, light]))
    b.light(F.make_ref(F.RT_KIND_RECT, 1))
    d = b.desc()
    W, H, spp = 48, 36, 6
    cam = rt.camera_new((0, 1, 4), (0, 0, -5), (0, 1, 0), 45.0, W / H, 0.0, 9.0, 0.0, 1.0)
    p = rt.make_params(W, H, spp, 30, (0.1, 0.1, 0.15), seed=8)
    rows = np.arange(H, dtype=np.uint32)
    ref, st_ref = O.render_cpu(d, cam, p, rows, n_threads=8, want_stats=True)
    assert st_ref.prim_tests[F.RT_KIND_MEDIUM] > 0 and st_ref.prim_tests[F.RT_KIND_BOX] > 0
    dev = rt.DeviceScene(d)
    out, st = dev.render(cam, p, rows, want_stats=True)
    assert st.as_dict() == st_ref.as_dict()
    assert np.array_equal(bits(out), bits(ref))
    with pytest.raises(rt.RtError) as e:                      # the A/B megakernel cannot take this scene
        dev.set_engine("mega")
    assert e.value.code == F.RT_ERR_UNSUPPORTED
    # media do not nest inside a boundary
    b2 = rt.DescBuilder()
    inner = b2.medium(b2.sphere((0, 0, 0), 1.0, b2.dielectric(1.5)), 0.5, b2.isotropic((1, 1, 1)))
    b2.set_root(b2.medium(inner, 0.5, b2.isotropic((1, 1, 1))))
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(b2.desc())
    assert e.value.code == F.RT_ERR_UNSUPPORTED


def test_edge_cases_empty_and_degenerate(rt, O):
    s = rt.HostScene("cornell_box")
    cam, bg = s.default_view(1.0)
    dev = rt.DeviceScene(s.desc)
    # no rows
    out = dev.render(cam, rt.make_params(16, 16, 2, 50, bg), np.zeros(0, dtype=np.uint32))
    assert out.shape == (0, 16, 3)
    rows = np.array([3, 3, 0], dtype=np.uint32)                      # a row may be asked twice
    for spp, depth in ((0, 50), (3, 0), (1, 1)):
        p = rt.make_params(16, 16, spp, depth, bg, seed=5)
        ref = O.render_cpu(s.desc, cam, p, rows)
        got = dev.render(cam, p, rows)
        assert np.array_equal(bits(got), bits(ref)), (spp, depth)
        if spp == 0 or depth == 0:
            assert not got.any()
    # 1-pixel-wide image: (W-1) == 0 divides to inf/NaN exactly like main.rs:147
    p = rt.make_params(1, 4, 2, 5, bg, seed=5)
    r4 = np.arange(4, dtype=np.uint32)
    a, b2 = dev.render(cam, p, r4), O.render_cpu(s.desc, cam, p, r4)
    assert np.array_equal(np.isnan(a), np.isnan(b2)) and np.array_equal(bits(a), bits(b2))
    # a bad row id is an error, not a fault
    with pytest.raises(rt.RtError):
        dev.render(cam, rt.make_params(16, 16, 1, 5, bg), np.array([16], dtype=np.uint32))


def test_nan_pixels_survive_like_the_reference(rt, O):
    """pdf_val == 0 -> NaN in the sum -> write_color scrubs the channel (main.rs:266-271, 284-292)."""
    b = rt.DescBuilder()
    floor_ = b.rect(F.RT_RECT_XZ, -5, 5, -5, 5, 0.0, b.lambertian((0.7, 0.7, 0.7)))
    b.set_root(floor_)
    # A FlipFace'd light in the light list answers pdf_value = 0 and random = (1,0,0) (trait defaults,
    # mod.rs:62-67): along the floor, so cosine = 0, both pdfs 0 and the sample is 0/0.
    b.light(b.rect(F.RT_RECT_XZ, -1, 1, -1, 1, 3.0, b.diffuse_light((5, 5, 5)), flip=True))
    d = b.desc()
    cam = rt.camera_new((0, 3, 6), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 10.0, 0.0, 1.0)
    p = rt.make_params(24, 24, 8, 10, (0.2, 0.2, 0.2), seed=3)
    rows = np.arange(24, dtype=np.uint32)
    ref = O.render_cpu(d, cam, p, rows)
    out = rt.DeviceScene(d).render(cam, p, rows)
    assert np.isnan(ref).any(), "the construction should produce NaN samples"
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    m = ~np.isnan(ref)
    assert np.array_equal(bits(out[m]), bits(ref[m]))
    assert np.array_equal(rt.write_color(out, 8), O.write_color(ref, 8))


def test_device_buffers_and_tonemap(rt, O):
    """rt_render_device + rt_tonemap_device with torch-owned HBM buffers on torch's stream."""
    import torch
    s = rt.HostScene("final_scene", seed=2022)
    W = H = 40
    cam, bg = s.default_view(1.0)
    p = rt.make_params(W, H, 3, 50, bg, seed=2022)
    rows = rt.shuffled_rows(H, 1)
    dev = rt.DeviceScene(s.desc)
    d_rows = torch.from_numpy(rows.view(np.int32)).cuda()
    d_out = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
    d_u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    st = F.rt_stats()
    dev.render_device(cam, p, d_rows.data_ptr(), H, d_out.data_ptr(), stream, st)
    F.check(rt.lib().rt_tonemap_device(C.c_void_p(d_out.data_ptr()), H * W, 3, C.c_void_p(d_u8.data_ptr()), C.c_void_p(stream)))
    dev.wait(stream)
    ref = O.render_cpu(s.desc, cam, p, rows)
    assert np.array_equal(bits(d_out.cpu().numpy()), bits(ref))
    assert np.array_equal(d_u8.cpu().numpy(), O.write_color(ref, 3))
    assert st.ms > 0


def test_asynchronous_calls_on_two_streams(rt, O):
    """RT_FLAG_ASYNC: rt_render_device returns at once and a host thread of the library drives the passes; two frames in flight
    on two streams of ONE scene (each with its own pool), a third queued behind the first, all equal to the oracle bit for bit;
    an engine-side error (a row id out of range, found by the device-side check) surfaces at rt_render_wait."""
    import torch
    s = rt.HostScene("cornell_smoke", seed=2022)
    W, H, spp = 48, 40, 4
    cam, bg = s.default_view(W / H)
    rows = rt.shuffled_rows(H, 3)
    dev = rt.DeviceScene(s.desc)
    d_rows = torch.from_numpy(rows.view(np.int32)).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    seeds = (11, 12, 13)
    outs = [torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda") for _ in seeds]
    stats = [F.rt_stats() for _ in seeds]
    params = [rt.make_params(W, H, spp, 50, bg, seed=sd, spp_chunk=1) for sd in seeds]
    torch.cuda.synchronize()
    for p in params:
        p.flags |= F.RT_FLAG_COUNTERS
    dev.render_device(cam, params[0], d_rows.data_ptr(), H, outs[0].data_ptr(), streams[0].cuda_stream, stats[0], asynchronous=True)
    dev.render_device(cam, params[1], d_rows.data_ptr(), H, outs[1].data_ptr(), streams[1].cuda_stream, stats[1], asynchronous=True)
    # a second call on a stream that still has one in flight waits for it first (one call at a time per scene and stream)
    dev.render_device(cam, params[2], d_rows.data_ptr(), H, outs[2].data_ptr(), streams[0].cuda_stream, stats[2], asynchronous=True)
    dev.wait(streams[1].cuda_stream)
    dev.wait(streams[0].cuda_stream)
    for i, sd in enumerate(seeds):
        ref, st_ref = O.render_cpu(s.desc, cam, params[i], rows, n_threads=4, want_stats=True)
        assert np.array_equal(bits(outs[i].cpu().numpy()), bits(ref)), sd
        assert stats[i].as_dict() == st_ref.as_dict(), sd         # (the first call's were filled when the third call's start waited for it)
    # the synchronous form still works on the same workspace, and gives the same bits
    again = torch.full((H, W, 3), float("nan"), dtype=torch.float64, device="cuda")
    dev.render_device(cam, params[1], d_rows.data_ptr(), H, again.data_ptr(), streams[1].cuda_stream, None)
    dev.wait(streams[1].cuda_stream)
    assert np.array_equal(bits(again.cpu().numpy()), bits(outs[1].cpu().numpy()))
    # an error found by the engine's thread is reported by the wait
    bad = torch.from_numpy(np.full(H, H * 7, dtype=np.int32)).cuda()
    dev.render_device(cam, params[0], bad.data_ptr(), H, again.data_ptr(), streams[0].cuda_stream, None, asynchronous=True)
    with pytest.raises(F.RtError) as e:
        dev.wait(streams[0].cuda_stream)
    assert "row id" in str(e.value)
    dev.render_device(cam, params[0], d_rows.data_ptr(), H, again.data_ptr(), streams[0].cuda_stream, None, asynchronous=True)   # and the stream is usable again
    dev.wait(streams[0].cuda_stream)
    assert np.array_equal(bits(again.cpu().numpy()), bits(outs[0].cpu().numpy()))


def test_full_size_properties_of_the_headline_config(rt, O):
    """At BASELINE's full image size (800x800, book-2 final scene) the oracle cannot render the frame, so:
    (1) rows rendered separately == rows rendered together (sharding invariance, multi-GPU);
    (2) re-running gives bit-identical sums (determinism) and both engines agree;
    (3) a few full-width rows at reduced spp still match the oracle exactly."""
    W = H = 800
    s = rt.HostScene("final_scene", seed=2022)
    cam, bg = s.default_view(1.0)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(W, H, 8, 50, bg, seed=2022, spp_chunk=2)
    rows = rt.shuffled_rows(H, 2022)
    full = dev.render(cam, p, rows)
    again = dev.render(cam, p, rows)
    assert np.array_equal(bits(full), bits(again))
    part = dev.render(cam, p, rows[100:164])
    assert np.array_equal(bits(part), bits(full[100:164]))
    dev.set_engine("mega")
    mega = dev.render(cam, p, rows[300:340])
    assert np.array_equal(bits(mega), bits(full[300:340]))
    ref = O.render_cpu(s.desc, cam, p, rows[:6], n_threads=6)
    assert np.array_equal(bits(full[:6]), bits(ref))
    img = rt.fill_image(full, rows, W, H, 8)
    assert img.shape == (H, W, 3) and img.max() == 255 and img.mean() > 5


def test_one_call_over_a_device_mask(rt, O):
    """rt_render_multi (SURVEY.md §8b "Threading": one call drives every device of the mask): rows dealt cyclically
    over the devices, shares rendered concurrently, output in row_ids order — the same bits as rt_render whatever the
    mask. One GPU here exercises the dealing and the reassembly; with more visible the mask grows."""
    import torch
    s = rt.HostScene("cornell_smoke", seed=4)
    W, H, spp = 48, 40, 5
    cam, bg = s.default_view(W / H)
    p = rt.make_params(W, H, spp, 50, bg, seed=4, spp_chunk=2)
    rows = rt.shuffled_rows(H, 6)
    ref, st_ref = O.render_cpu(s.desc, cam, p, rows, n_threads=4, want_stats=True)
    ndev = torch.cuda.device_count()
    for mask in sorted({1, (1 << min(ndev, 2)) - 1, (1 << ndev) - 1}):
        group = rt.DeviceSceneSet(s.desc, mask)
        out, st = group.render(cam, p, rows, want_stats=True)
        assert st.as_dict() == st_ref.as_dict(), mask
        assert np.array_equal(bits(out), bits(ref)), mask
        assert np.array_equal(bits(group.render(cam, p, rows[:7])), bits(ref[:7])), mask      # fewer rows than a full deal
        assert group.render(cam, p, rows[:0]).shape == (0, W, 3)
        group.close()
    with pytest.raises(rt.RtError) as e:                              # a device this process cannot see
        rt.DeviceSceneSet(s.desc, 1 << ndev)
    assert e.value.code == F.RT_ERR_DEVICE
    with pytest.raises(rt.RtError):
        rt.DeviceSceneSet(s.desc, 0)


def test_node_table_variant_gives_the_same_bits(rt):
    """The traversal variant that keeps the BVH's node table in LDS (1024-thread workgroups) against the plain kernels
    (tuning bit 28): same pixels bit for bit, whether the whole table fits or only its first records."""
    default = 18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24)
    for name, expect_table in (("final_scene", True), ("random_scene", True), ("cornell_smoke", True), ("two_perlin_spheres", True)):
        c, g, s, cam, p = golden_case(name, rt)
        dev = rt.DeviceScene(s.desc)
        v = dev.trace_variant()
        assert (v["nodes_in_lds"] == s.desc.n_nodes and v["workgroup_threads"] == 1024) == expect_table, (name, v)
        assert v["spheres_in_lds"] == (name in ("random_scene", "two_perlin_spheres")), (name, v)    # (the sphere-only scenes of the four)
        a = dev.render(cam, p, g["rows"])
        dev.set_tuning(default | (1 << 28))
        v2 = dev.trace_variant()
        assert v2["nodes_in_lds"] == 0 and v2["workgroup_threads"] == 256, (name, v2)
        b = dev.render(cam, p, g["rows"])
        assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(g["rgb_sum"])), name
    # A sphere-only scene too large for the all-in-LDS instance whose table still fits (601 to 1 740 nodes): single-precision records in LDS
    mid = rt.HostScene("random_scene", seed=5, param=12)             # (25 x 25 grid: ~740 nodes)
    dev = rt.DeviceScene(mid.desc)
    v = dev.trace_variant()
    assert v["workgroup_threads"] == 1024 and v["nodes_in_lds"] == mid.desc.n_nodes and v["f32_slabs"] and not v["spheres_in_lds"], v
    cam, bg = mid.default_view(16 / 9)
    p = rt.make_params(64, 36, 3, 50, bg, seed=5)
    rows = np.arange(36, dtype=np.uint32)
    from oracle import oracle_ffi as O_
    assert np.array_equal(bits(dev.render(cam, p, rows)), bits(O_.render_cpu(mid.desc, cam, p, rows, n_threads=4)))
    # A sphere scene whose node table does not fit takes the plain kernel — single-precision 32-byte node records from L2 / HBM, five
    # waves per SIMD — not the partial table (measured: 10 % slower there).
    big = rt.HostScene("random_scene", seed=5, param=40)             # (81 x 81 grid: ~13 K nodes)
    dev = rt.DeviceScene(big.desc)
    v = dev.trace_variant()
    assert v["workgroup_threads"] == 256 and v["nodes_in_lds"] == 0 and v["f32_slabs"] and v["stack_entries"] >= dev.info()["stack_need"], v
    cam, bg = big.default_view(16 / 9)
    p = rt.make_params(64, 36, 2, 50, bg, seed=5)
    rows = np.arange(36, dtype=np.uint32)
    a = dev.render(cam, p, rows)
    assert np.array_equal(bits(a), bits(O_.render_cpu(big.desc, cam, p, rows, n_threads=4)))
    # ... and a BVH too deep for stacks of 16 entries takes the plain kernels
    deep = rt.HostScene("wwscene", seed=5, param=1)
    assert rt.DeviceScene(deep.desc).trace_variant()["nodes_in_lds"] == 0


def test_kernel_times_and_run_report(rt):
    """rt_stats says how the call ran (chunk, passes, pool) and, on request, the device time of the two kernels."""
    s = rt.HostScene("final_scene", seed=2022)
    W = H = 96
    cam, bg = s.default_view(1.0)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(W, H, 6, 50, bg, seed=1, spp_chunk=0, flags=F.RT_FLAG_KERNEL_TIMES)
    rows = np.arange(H, dtype=np.uint32)
    pr = F.rt_params.from_buffer_copy(p); pr.n_rows = len(rows); pr.row_ids = rows.ctypes.data
    out = np.empty((H, W, 3)); st = F.rt_stats()
    F.check(rt.lib().rt_render(dev._h, C.byref(cam), C.byref(pr), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)))
    assert st.spp_chunk == 6 and st.passes > 5 and st.pool_slots >= W * H // 64 * 64
    assert 0 < st.trace_ms < st.ms and 0 < st.shade_ms < st.ms and st.trace_ms + st.shade_ms <= st.ms * 1.05
    p.flags = 0
    st2 = F.rt_stats()
    pr = F.rt_params.from_buffer_copy(p); pr.n_rows = len(rows); pr.row_ids = rows.ctypes.data
    out2 = np.empty((H, W, 3))
    F.check(rt.lib().rt_render(dev._h, C.byref(cam), C.byref(pr), out2.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st2)))
    assert st2.trace_ms == 0 and st2.shade_ms == 0 and np.array_equal(bits(out), bits(out2))


def test_device_resident_row_ids_are_checked(rt):
    import torch
    s = rt.HostScene("cornell_box")
    cam, bg = s.default_view(1.0)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(16, 16, 1, 5, bg)
    d_rows = torch.tensor([0, 5, 16], dtype=torch.int32, device="cuda")      # 16 is one past the last row
    d_out = torch.zeros((3, 16, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(rt.RtError) as e:
        dev.render_device(cam, p, d_rows.data_ptr(), 3, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert e.value.code == F.RT_ERR_INVALID
    d_rows[2] = 15
    dev.render_device(cam, p, d_rows.data_ptr(), 3, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    dev.wait(torch.cuda.current_stream().cuda_stream)
    assert torch.isfinite(d_out).all()


@pytest.mark.parametrize("scene,W,H,param,assets", [
    ("random_scene", 1200, 800, 0, False),        # BASELINE config 2
    ("cornell_box", 600, 600, 0, False),          # BASELINE config 4
    ("wwscene", 1920, 1080, 3, True),             # BASELINE config 5: Shuttle.obj x 3 subdivisions, real textures
])
def test_full_size_properties_of_the_other_configs(rt, O, scene, W, H, param, assets):
    """BASELINE configs 2, 4 and 5 at their stated image sizes (reduced spp — Mrays/s and every per-ray decision are
    spp-independent): rows rendered apart == rows rendered together (what the multi-GPU sharding relies on), a rerun
    gives the same bits, and a few full-width rows match the oracle exactly, counters included."""
    if assets and not os.path.isdir(ASSETS):
        pytest.skip("assets/ not present")
    s = rt.HostScene(scene, seed=2022, param=param, assets_dir=ASSETS if assets else None)
    cam, bg = s.default_view(W / H)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(W, H, 3, 50, bg, seed=2022, spp_chunk=1)
    rows = rt.shuffled_rows(H, 2022)[:96]
    full = dev.render(cam, p, rows)
    assert np.array_equal(bits(full), bits(dev.render(cam, p, rows)))
    part = dev.render(cam, p, rows[40:56])
    assert np.array_equal(bits(part), bits(full[40:56]))
    ref, st_ref = O.render_cpu(s.desc, cam, p, rows[:4], n_threads=4, want_stats=True)
    out, st = dev.render(cam, p, rows[:4], want_stats=True)
    assert st.as_dict() == st_ref.as_dict()
    assert np.array_equal(bits(out), bits(ref)) and np.array_equal(bits(full[:4]), bits(ref))


def test_c1_at_its_stated_size(rt, O):
    """BASELINE config 1 — book-1 final scene, 400x225, 100 spp, depth 50, 'CPU reference path only' — at its own size:
    the oracle renders a sample of full-width rows at the full 100 spp (the whole frame is bench.py --config c1's CPU leg)
    and the HIP path gives the same bits and counters; the whole frame on the HIP path is deterministic and its rows do not
    depend on what else is in the call."""
    W, H, spp = 400, 225, 100
    s = rt.HostScene("random_scene", seed=2022)
    cam, bg = s.default_view(W / H)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
    rows = rt.shuffled_rows(H, 2022)
    full, st_full = dev.render(cam, p, rows, want_stats=True)
    assert st_full.paths == W * H * spp
    ref, st_ref = O.render_cpu(s.desc, cam, p, rows[:10], n_threads=8, want_stats=True)
    out, st = dev.render(cam, p, rows[:10], want_stats=True)
    assert st.as_dict() == st_ref.as_dict()
    assert np.array_equal(bits(out), bits(ref)) and np.array_equal(bits(full[:10]), bits(ref))
    assert np.array_equal(rt.write_color(full[:10], spp), O.write_color(ref, spp))
    p0 = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=0)        # the reference's own loop order, one item per pixel
    assert np.array_equal(bits(dev.render(cam, p0, rows[100:110])), bits(full[100:110]))


def test_frames_of_a_strip_are_the_frames_rendered_alone(rt, O):
    """bench.py hands the library several frames per call (an n_frames strip: row id g = frame g / height, frames differ in
    their RNG key only) so that the pool drains once per call. Every frame of the strip must be the frame rendered by itself,
    bit for bit, and a strip frame must equal the oracle's."""
    W, H, spp, nf = 96, 64, 6, 3
    s = rt.HostScene("final_scene", seed=2022)
    cam, bg = s.default_view(W / H)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, n_frames=nf, spp_chunk=1)
    from raytracer_2022_amd import film
    strip_rows = film.strip_rows(H, nf, 2022)                         # all rows of all frames, shuffled together
    strip = dev.render(cam, p, strip_rows)
    for f in range(nf):
        mine = np.flatnonzero(strip_rows // H == f)
        alone = dev.render(cam, p, strip_rows[mine])                  # the same rows in a call of their own
        assert np.array_equal(bits(alone), bits(strip[mine]))
    f1 = np.flatnonzero(strip_rows // H == 1)[:8]
    ref = O.render_cpu(s.desc, cam, p, strip_rows[f1], n_threads=4)
    assert np.array_equal(bits(strip[f1]), bits(ref))
    # and frame 0 of a strip is the single-frame render (n_frames = 1) of the same seed
    p1 = rt.make_params(W, H, spp, 50, bg, seed=2022, n_frames=1, spp_chunk=1)
    f0 = np.flatnonzero(strip_rows // H == 0)
    assert np.array_equal(bits(dev.render(cam, p1, strip_rows[f0])), bits(strip[f0]))


@pytest.mark.parametrize("planes", [1, 2, 8, 20, 32])
def test_partial_sum_ring_gives_the_same_bits(rt, O, planes):
    """Ring of partial-sum planes (one-sample work items, r3): sample c of a pixel goes to plane c mod R, the host adds finished
    planes to the output in sample order while the frame runs, and work items beyond R planes wait. One plane makes every sample
    wait for its predecessor (the worst case for the bookkeeping: starved slots, the claim limit, the oldest item in flight);
    the sums must be the reference's running sums bit for bit whatever R is."""
    W, H, spp = 64, 48, 40
    s = rt.HostScene("final_scene", seed=2022)
    cam, bg = s.default_view(W / H)
    dev = rt.DeviceScene(s.desc)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
    rows = rt.shuffled_rows(H, 2022)
    dev.set_partial_ring(-1)
    plain, st0 = dev.render(cam, p, rows, want_stats=True)
    assert st0.partial_bytes == W * H * spp * 24
    dev.set_partial_ring(planes)
    ring, st1 = dev.render(cam, p, rows, want_stats=True)
    assert st1.partial_bytes == W * H * planes * 24
    assert st1.as_dict() == st0.as_dict()
    assert np.array_equal(bits(ring), bits(plain))
    again = dev.render(cam, p, rows)                                   # (the timed build of the kernels, not the counting one)
    assert np.array_equal(bits(again), bits(plain))
    ref = O.render_cpu(s.desc, cam, p, rows[:6], n_threads=4)
    assert np.array_equal(bits(ring[:6]), bits(ref))
    # a ring larger than spp is no ring; spp_chunk = 0 and k > 1 never use one
    dev.set_partial_ring(1024)
    _, st2 = dev.render(cam, p, rows, want_stats=True)
    assert st2.partial_bytes == W * H * spp * 24
    p4 = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=4)
    dev.set_partial_ring(2)
    out4, st4 = dev.render(cam, p4, rows, want_stats=True)
    assert st4.partial_bytes == W * H * 10 * 24
    assert np.array_equal(bits(out4[:4]), bits(O.render_cpu(s.desc, cam, p4, rows[:4], n_threads=4)))


def test_partial_sum_ring_on_a_strip_with_a_small_pool(rt, O):
    """The ring under pressure: a three-frame strip, a pool of one segment (4096 slots for 27 648 pixels: many pool fills per
    plane) and two planes — claims stall and resume all through the frame."""
    W, H, spp, nf = 96, 96, 12, 3
    s = rt.HostScene("cornell_box", seed=5)
    cam, bg = s.default_view(1.0)
    dev = rt.DeviceScene(s.desc)
    from raytracer_2022_amd import film
    rows = film.strip_rows(H, nf, 5)
    p = rt.make_params(W, H, spp, 50, bg, seed=5, n_frames=nf, spp_chunk=1)
    dev.set_partial_ring(-1)
    plain = dev.render(cam, p, rows)
    dev.set_engine("wavefront", 1)                                     # one segment of 4096 path slots
    dev.set_partial_ring(2)
    ring = dev.render(cam, p, rows)
    assert np.array_equal(bits(ring), bits(plain))
