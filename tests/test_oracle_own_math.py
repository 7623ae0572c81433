"""The oracle against itself with NOTHING shared with the product.

Every HIP-vs-oracle test compares two programs that include the same raytracer_2022_amd/csrc/rt_math.h (vectors, reflect /
refract, Onb, the RNG's conversions, float-to-integer casts, the five transcendentals): a slip in that header would be
invisible to them ("common mode"). `make -C oracle own` builds rt_oracle.cpp against oracle/rto_math.h instead —
the oracle's own, separately written restatement of the same reference lines (basic/vec.rs:24-46,119-128, basic/onb.rs:22-36,
basic/ray.rs:4-20) and of rand 0.8.5's conversions, with the platform libm for the transcendentals. This file holds that
build against the -DRTO_LIBM build (rt_math.h's vectors / RNG / casts, libm transcendentals): they must agree BIT FOR BIT — pixel
sums, u8 pixels and every counter — on all nine scene builders, which pins rt_math.h's non-transcendental half by a second
statement; the transcendental half is what tests/test_libm_sensitivity.py measures.

PARITY UNPINNED all the same: both statements are this repo's; the reference cannot be built here and is unseeded."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_the_own_build_shares_nothing(O):
    L = O.lib_own()
    assert L.rto_uses_own_math() == 1 and L.rto_uses_libm() == 1
    assert O.lib().rto_uses_own_math() == 0 and O.lib_libm().rto_uses_own_math() == 0
    # its source never reaches into the product's tree
    src = open(os.path.join(ROOT, "oracle", "rto_math.h")).read()
    assert "#include \"../" not in src and "raytracer_2022_amd" not in src.split("#ifndef RTO_MATH_H")[1]


def test_rng_streams_and_conversions_agree(O):
    """SplitMix64 words, Standard f64, UniformFloat / UniformInt::sample_single: rt_math.h (shifts, a 32-bit-halves high product)
    against rto_math.h (ldexp, a 128-bit product), word for word."""
    A, B = O.lib(), O.lib_own()
    n = 5000
    for state in (0, 1, 2022, 0xFFFFFFFFFFFFFFFF, 0x9E3779B97F4A7C15):
        ua, ub = np.empty(n, np.uint64), np.empty(n, np.uint64)
        A.rto_rng_u64(state, ua.ctypes.data_as(C.POINTER(C.c_uint64)), n); B.rto_rng_u64(state, ub.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        assert np.array_equal(ua, ub)
        fa, fb = np.empty(n), np.empty(n)
        A.rto_rng_f64(state, fa.ctypes.data_as(C.POINTER(C.c_double)), n); B.rto_rng_f64(state, fb.ctypes.data_as(C.POINTER(C.c_double)), n)
        assert np.array_equal(bits(fa), bits(fb)) and fa.min() >= 0.0 and fa.max() < 1.0
        for lo, hi in ((-1.0, 1.0), (0.0, 1.0), (123.0, 423.0), (-1e-300, 1e-300), (1.0, 1.0 + 2.0 ** -50)):
            A.rto_rng_range(state, lo, hi, fa.ctypes.data_as(C.POINTER(C.c_double)), n); B.rto_rng_range(state, lo, hi, fb.ctypes.data_as(C.POINTER(C.c_double)), n)
            assert np.array_equal(bits(fa), bits(fb)) and fa.min() >= lo and fa.max() < hi
        for bound in (1, 2, 3, 7, 1000, 2 ** 31 + 1, 2 ** 63 + 5, 2 ** 64 - 1):
            A.rto_rng_index(state, bound, ua.ctypes.data_as(C.POINTER(C.c_uint64)), n); B.rto_rng_index(state, bound, ub.ctypes.data_as(C.POINTER(C.c_uint64)), n)
            assert np.array_equal(ua, ub) and int(ua.max()) < bound
    for key in ((2022, 0, 0, 0), (2022, 3, 639999, 999), (0, 0, 2 ** 40, 2 ** 31), (2 ** 64 - 1, 2 ** 32 - 1, 2 ** 63, 2 ** 32 - 1)):
        assert A.rto_path_key(*key) == B.rto_path_key(*key)


SCENES = [("random_scene", 0), ("two_spheres", 0), ("two_perlin_spheres", 0), ("earth", 0), ("simple_light", 0), ("cornell_box", 0),
          ("cornell_smoke", 0), ("final_scene", 0), ("wwscene", 1)]


@pytest.mark.parametrize("scene,param", SCENES)
def test_own_math_build_equals_the_shared_math_build(rt, O, scene, param):
    """Every scene builder of the reference (scene.rs:22-571), rendered by the two builds: same bits, same counters."""
    assets = os.path.join(ROOT, "assets")
    s = rt.HostScene(scene, seed=7, param=param, assets_dir=assets if os.path.isdir(assets) else None)
    W, H, spp = 40, 28, 6
    cam, bg = s.default_view(W / H)
    rows = rt.shuffled_rows(H, 7)
    for chunk in (0, 2):
        p = rt.make_params(W, H, spp, 50, bg, seed=7, spp_chunk=chunk)
        a, sa = O.render_cpu(s.desc, cam, p, rows, n_threads=4, want_stats=True, libm=True)
        b, sb = O.render_cpu(s.desc, cam, p, rows, n_threads=4, want_stats=True, own=True)
        assert sa.as_dict() == sb.as_dict(), (scene, chunk)
        assert np.array_equal(bits(a), bits(b)), (scene, chunk)
        assert sa.rays > W * H * spp // 2 and sa.rng_draws > 0
    assert np.array_equal(O.write_color(a, spp), O.write_color(b, spp))
