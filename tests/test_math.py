"""rt_math.h on the host: the shared f64 arithmetic (own sin/cos/acos/atan2/log, the path RNG and
its rand-0.8.5-style conversions). Known answers + agreement with libm to <= 1 ulp."""
import ctypes as C
import math

import numpy as np
import pytest


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    spacing = np.spacing(np.maximum(np.abs(b), 1e-300))
    return np.abs(a - b) / spacing


@pytest.mark.parametrize("op,fn,lo,hi", [
    ("RTO_SIN", np.sin, -50.0, 50.0), ("RTO_COS", np.cos, -50.0, 50.0),
    ("RTO_SIN", np.sin, -1.0e5, 1.0e5), ("RTO_COS", np.cos, -1.0e5, 1.0e5),
    ("RTO_ACOS", np.arccos, -1.0, 1.0), ("RTO_LOG", np.log, 1e-300, 10.0),
])
def test_transcendentals_within_one_ulp_of_libm(O, op, fn, lo, hi):
    rng = np.random.default_rng(42)
    x = rng.uniform(lo, hi, 200_000)
    got = O.math_array(getattr(O, op), x)
    assert ulp_diff(got, fn(x)).max() <= 1.0


def test_atan2_within_one_ulp_and_quadrants(O):
    rng = np.random.default_rng(7)
    y = rng.uniform(-3, 3, 200_000)
    x = rng.uniform(-3, 3, 200_000)
    got = O.math_array(O.RTO_ATAN2, y, x)
    assert ulp_diff(got, np.arctan2(y, x)).max() <= 1.0
    L = O.lib()
    assert L.rto_math(O.RTO_ATAN2, 0.0, 1.0) == 0.0
    assert L.rto_math(O.RTO_ATAN2, 0.0, -1.0) == pytest.approx(math.pi, abs=1e-15)
    assert L.rto_math(O.RTO_ATAN2, 1.0, 0.0) == pytest.approx(math.pi / 2, abs=1e-15)
    assert L.rto_math(O.RTO_ATAN2, -1.0, 0.0) == pytest.approx(-math.pi / 2, abs=1e-15)


def test_special_values(O):
    L = O.lib()
    assert L.rto_math(O.RTO_SIN, 0.0, 0) == 0.0 and L.rto_math(O.RTO_COS, 0.0, 0) == 1.0
    assert L.rto_math(O.RTO_ACOS, 1.0, 0) == 0.0
    assert L.rto_math(O.RTO_ACOS, -1.0, 0) == pytest.approx(math.pi, abs=1e-15)
    assert math.isnan(L.rto_math(O.RTO_ACOS, 1.5, 0))
    assert L.rto_math(O.RTO_LOG, 1.0, 0) == 0.0
    assert L.rto_math(O.RTO_LOG, 0.0, 0) == -math.inf
    assert math.isnan(L.rto_math(O.RTO_LOG, -1.0, 0))
    assert L.rto_math(O.RTO_LOG, math.e, 0) == 1.0            # rnd.log(E) divides by exactly 1
    assert math.isnan(L.rto_math(O.RTO_SIN, math.inf, 0))
    # large arguments stay bounded and close to libm
    for x in (1.0e7, 3.3e9, -7.7e12, 1.0e15):
        assert abs(L.rto_math(O.RTO_SIN, x, 0) - math.sin(x)) < 1e-9


def test_rng_stream_known_answers(O):
    """SplitMix64 stream from state 0: published test vector of the algorithm."""
    out = (C.c_uint64 * 3)()
    O.lib().rto_rng_u64(0, out, 3)
    assert [hex(v) for v in out] == ["0xe220a8397b1dcdaf", "0x6e789e6aa1b965f4", "0x6c45d188009454f"]


def test_rng_conversions_follow_rand_0_8_5(O):
    n = 4096
    u = (C.c_uint64 * n)()
    O.lib().rto_rng_u64(123, u, n)
    u = np.array(u[:], dtype=np.uint64)
    f = (C.c_double * n)()
    O.lib().rto_rng_f64(123, f, n)
    # Standard f64: (u64 >> 11) * 2^-53
    assert np.array_equal(np.array(f[:]), (u >> np.uint64(11)).astype(np.float64) * 2.0 ** -53)
    r = (C.c_double * n)()
    O.lib().rto_rng_range(123, -1.0, 1.0, r, n)
    # UniformFloat::sample_single: ((u64 >> 12 | exp 0) - 1.0) * scale + low
    v12 = ((u >> np.uint64(12)) | np.uint64(0x3FF0000000000000)).view(np.float64)
    assert np.array_equal(np.array(r[:]), (v12 - 1.0) * 2.0 + (-1.0))
    assert min(r) >= -1.0 and max(r) < 1.0
    idx = (C.c_uint64 * n)()
    O.lib().rto_rng_index(123, 7, idx, n)
    idx = np.array(idx[:])
    assert idx.min() == 0 and idx.max() == 6
    # widening multiply: hi word of u * n whenever the low word is inside the zone
    zone = ((7 << 61) - 1)
    exp, k = [], 0
    for want in range(64):
        while True:
            m = int(u[k]) * 7
            k += 1
            if (m & 0xFFFFFFFFFFFFFFFF) <= zone:
                exp.append(m >> 64)
                break
    assert list(idx[:64]) == exp


def test_path_key_separates_streams(O):
    L = O.lib()
    keys = {L.rto_path_key(2022, f, p, s) for f in range(3) for p in range(50) for s in range(50)}
    assert len(keys) == 3 * 50 * 50
    assert L.rto_path_key(2022, 0, 5, 9) != L.rto_path_key(2023, 0, 5, 9)
