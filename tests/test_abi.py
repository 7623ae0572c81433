"""The C-ABI shared library loads, exports every symbol the headers declare, and its
structures have the layout the bindings assume. No compute calls (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("rt2022.h", "rt2022_host.h", "rt2022_debug.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(rtb?_[a-z0-9_]+)\s*\(", text):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol(rt):
    from raytracer_2022_amd import _ffi as F
    lib = rt.lib()
    declared = declared_symbols()
    assert declared, "no declarations parsed"
    assert declared == set(F.ABI_SYMBOLS), "bindings and headers disagree: %s" % (declared ^ set(F.ABI_SYMBOLS))
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_abi_version_and_struct_layout(rt):
    from raytracer_2022_amd import _ffi as F
    lib = rt.lib()
    assert lib.rt_abi_version() == F.RT2022_ABI_VERSION == 3
    out = (C.c_uint32 * 64)()
    n = lib.rtb_abi_sizes(out, 64)
    assert n == len(F.ABI_STRUCTS)
    assert [out[i] for i in range(n)] == [C.sizeof(t) for t in F.ABI_STRUCTS]
    # SURVEY.md §8(d) algorithmic record sizes
    assert C.sizeof(F.rt_bvh_node) == 64 and C.sizeof(F.rt_sphere) == 40 and C.sizeof(F.rt_moving_sphere) == 80
    assert C.sizeof(F.rt_rect) == 48 and C.sizeof(F.rt_box) == 56 and C.sizeof(F.rt_triangle) == 80
    assert C.sizeof(F.rt_camera) == 192


def test_ref_encoding(rt):
    from raytracer_2022_amd import _ffi as F
    r = F.make_ref(F.RT_KIND_RECT, 12345, flip=True)
    assert F.ref_kind(r) == F.RT_KIND_RECT and F.ref_index(r) == 12345 and r & F.RT_REF_FLIP
    assert F.make_ref(F.RT_KIND_NODE, 0) == 0


def test_scene_validation_errors_are_reported_not_crashed(rt):
    """rt_scene_create validates before it touches the device: malformed scenes give RT_ERR_INVALID
    (the reference would index out of bounds / panic)."""
    from raytracer_2022_amd import _ffi as F
    b = rt.DescBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    b.set_root(b.sphere((0, 0, 0), 1.0, m))
    d = b.desc()
    d.root = F.make_ref(F.RT_KIND_SPHERE, 7)                    # index out of range
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(d)
    assert e.value.code == F.RT_ERR_INVALID and "out of range" in str(e.value)

    b = rt.DescBuilder()
    b.set_root(b.sphere((0, 0, 0), 1.0, 3))                      # material index out of range
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(b.desc())
    assert e.value.code == F.RT_ERR_INVALID

    b = rt.DescBuilder()
    m = b.lambertian((0.5, 0.5, 0.5))
    sph = b.sphere((0, 0, 0), 1.0, m)
    n0 = b.node((-1, -1, -1), (1, 1, 1), sph, sph)
    b.pools["nodes"][0].left = n0                                # a node that is its own child
    b.set_root(n0)
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(b.desc())
    assert e.value.code == F.RT_ERR_INVALID and "cycle" in str(e.value)

    b = rt.DescBuilder()
    lam = b.lambertian((0.5, 0.5, 0.5))
    b.set_root(b.medium(b.sphere((0, 0, 0), 1.0, lam), 0.1, lam))  # phase function must be Isotropic
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(b.desc())
    assert e.value.code == F.RT_ERR_INVALID

    d = rt.DescBuilder().desc()
    d.abi_version = 99
    with pytest.raises(rt.RtError):
        rt.DeviceScene(d)


@pytest.mark.skipif(__import__("tests.conftest", fromlist=["has_gpu"]).has_gpu(), reason="needs a machine without a GPU")
def test_no_gpu_is_a_loud_error_not_a_fallback(rt):
    """Without a HIP device the product refuses to run: there is no CPU path behind rt_render."""
    from raytracer_2022_amd import _ffi as F
    s = rt.HostScene("cornell_box")
    with pytest.raises(rt.RtError) as e:
        rt.DeviceScene(s.desc)
    assert e.value.code == F.RT_ERR_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_2022_amd/ may import, link or load it."""
    pkg = os.path.join(ROOT, "raytracer_2022_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "librt_oracle" not in text and "oracle_ffi" not in text and "rt_oracle" not in text, os.path.join(dirpath, f)
