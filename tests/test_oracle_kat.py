"""Known-answer tests that pin the CPU oracle to the reference's behaviour, line by line.

The reference ships no tests, fixtures or seedable RNG, so these are
authored against the cited lines of /root/reference/raytracer/src (SURVEY.md §8c list).
Each test states the reference rule it checks. Everything runs through the oracle's
C hooks on hand-built flattened scenes (DescBuilder).
"""
import ctypes as C
import math

import numpy as np
import pytest

from raytracer_2022_amd import _ffi as F

INF = float("inf")


def one(rt, make):
    """Scene with a single object as root; returns (desc, ref)."""
    b = rt.DescBuilder()
    ref = make(b)
    b.set_root(ref)
    return b.desc(), ref, b


# ---------------------------------------------------------------- spheres ---
def test_sphere_roots_normal_uv(rt, O):
    """Sphere::hit sphere.rs:39-66: nearer root first; normal = (p-c)/r; uv = get_sphere_uv (sphere.rs:30-34)."""
    d, ref, _ = one(rt, lambda b: b.sphere((0, 0, -5), 1.0, b.lambertian((1, 1, 1))))
    r = O.hit(d, ref, (0, 0, 0), (0, 0, -1))
    assert r.hit and r.t == 4.0 and list(r.p) == [0, 0, -4] and list(r.normal) == [0, 0, 1] and r.front_face == 1
    # outward normal (0,0,1): theta = acos(0) = pi/2, phi = atan2(-1, 0) + pi = pi/2
    assert r.v == pytest.approx(0.5, abs=1e-15) and r.u == pytest.approx(0.25, abs=1e-15)
    # from inside: first root negative -> second root, normal flipped against the ray (mod.rs:49-56)
    r = O.hit(d, ref, (0, 0, -5), (0, 0, -1))
    assert r.hit and r.t == 1.0 and r.front_face == 0 and list(r.normal) == [0, 0, 1]
    # direction is not normalised: t scales
    r = O.hit(d, ref, (0, 0, 0), (0, 0, -2))
    assert r.t == 2.0


def test_sphere_root_interval_is_inclusive(rt, O):
    """sphere.rs:52: reject only root < t_min or t_max < root — both ends inclusive (right wins BVH ties)."""
    d, ref, _ = one(rt, lambda b: b.sphere((0, 0, -5), 1.0, b.lambertian((1, 1, 1))))
    assert O.hit(d, ref, (0, 0, 0), (0, 0, -1), t_min=4.0, t_max=4.0).hit
    assert O.hit(d, ref, (0, 0, 0), (0, 0, -1), t_min=0.001, t_max=4.0).t == 4.0
    # nearer root excluded by t_min -> farther root
    assert O.hit(d, ref, (0, 0, 0), (0, 0, -1), t_min=4.5).t == 6.0
    assert not O.hit(d, ref, (0, 0, 0), (0, 0, -1), t_min=6.5).hit
    assert not O.hit(d, ref, (0, 0, 0), (0, 0, -1), t_max=3.999).hit
    # tangent ray: discriminant 0 still hits; miss beyond
    assert O.hit(d, ref, (1, 0, 0), (0, 0, -1)).hit
    assert not O.hit(d, ref, (1.0000001, 0, 0), (0, 0, -1)).hit


def test_sphere_uv_poles_and_seam(rt, O):
    d, ref, _ = one(rt, lambda b: b.sphere((0, 0, 0), 1.0, b.lambertian((1, 1, 1))))
    top = O.hit(d, ref, (0, 5, 0), (0, -1, 0))       # normal (0,1,0): theta = acos(-1) = pi -> v = 1
    assert top.v == pytest.approx(1.0, abs=1e-15)
    bottom = O.hit(d, ref, (0, -5, 0), (0, 1, 0))    # normal (0,-1,0): theta = 0 -> v = 0
    assert bottom.v == 0.0
    px = O.hit(d, ref, (5, 0, 0), (-1, 0, 0))        # normal (1,0,0): phi = atan2(-0, 1) + pi = pi -> u = 0.5
    assert px.u == pytest.approx(0.5, abs=1e-15)
    nx = O.hit(d, ref, (-5, 0, 0), (1, 0, 0))        # normal (-1,0,0): atan2(-0,-1) = -pi -> phi = 0 -> u = 0
    assert nx.u == pytest.approx(0.0, abs=1e-15)


def test_moving_sphere_center_interpolates(rt, O):
    """MovingSphere::center sphere.rs:124-127 = c0 + (c1-c0)*((tm-t0)/(t1-t0))."""
    d, ref, _ = one(rt, lambda b: b.moving_sphere((0, 0, -5), (0, 2, -5), 0.0, 1.0, 1.0, b.lambertian((1, 1, 1))))
    assert O.hit(d, ref, (0, 0, 0), (0, 0, -1), tm=0.0).t == 4.0
    assert O.hit(d, ref, (0, 1, 0), (0, 0, -1), tm=0.5).t == 4.0
    assert not O.hit(d, ref, (0, 0, 0), (0, 0, -1), tm=1.0).hit


# ------------------------------------------------------------------ rects ---
@pytest.mark.parametrize("axis,orig,dirn,normal", [
    (F.RT_RECT_XY, (0.5, 0.25, 2.0), (0, 0, -1), [0, 0, 1]),
    (F.RT_RECT_XZ, (0.5, 2.0, 0.25), (0, -1, 0), [0, 1, 0]),
    (F.RT_RECT_YZ, (2.0, 0.5, 0.25), (-1, 0, 0), [1, 0, 0]),
])
def test_rect_hit_uv_and_edges(rt, O, axis, orig, dirn, normal):
    """aarect.rs:46-72 (+XZ/YZ twins): t = (k-o)/d, bounds inclusive, u=(a-a0)/(a1-a0), v likewise."""
    d, ref, _ = one(rt, lambda b: b.rect(axis, 0.0, 1.0, 0.0, 0.5, 0.0, b.lambertian((1, 1, 1))))
    r = O.hit(d, ref, orig, dirn)
    assert r.hit and r.t == 2.0 and r.u == 0.5 and r.v == 0.5 and list(r.normal) == normal and r.front_face == 1
    back = O.hit(d, ref, tuple(-o if i == [2, 1, 0][axis] else o for i, o in enumerate(orig)), tuple(-x for x in dirn))
    assert back.hit and back.front_face == 0 and list(back.normal) == [-n for n in normal]
    # exactly on the edge a = a1 is inside; just beyond is out
    edge = list(orig)
    a_index = [0, 0, 1][axis]
    edge[a_index] = 1.0
    assert O.hit(d, ref, edge, dirn).hit
    edge[a_index] = 1.0000001
    assert not O.hit(d, ref, edge, dirn).hit
    # t == t_max accepted (aarect.rs:48 rejects only t > t_max)
    assert O.hit(d, ref, orig, dirn, t_max=2.0).hit and not O.hit(d, ref, orig, dirn, t_max=1.999).hit


def test_boxes_side_order_and_ties(rt, O):
    """Boxes boxes.rs:24-66 + HittableList::hit mod.rs:90-100: six rects, later side wins a tie."""
    d, ref, _ = one(rt, lambda b: b.box((0, 0, 0), (1, 2, 3), b.lambertian((1, 1, 1))))
    r = O.hit(d, ref, (0.5, 1.0, 10.0), (0, 0, -1))            # enters through z = 3 (side 0), leaves z = 0
    assert r.t == 7.0 and list(r.normal) == [0, 0, 1]
    r = O.hit(d, ref, (0.5, 1.0, 1.5), (0, 0, -1))             # from inside: hits z = 0 from its +z side
    assert r.t == 1.5 and list(r.normal) == [0, 0, 1] and r.front_face == 1
    # corner ray hitting the x = 1 and y = 2 faces at the same t: YZ sides come after XZ -> x face wins
    r = O.hit(d, ref, (2.0, 3.0, 1.0), (-1, -1, 0))
    assert r.t == 1.0 and list(r.normal) == [1, 0, 0]


# -------------------------------------------------------- triangle / ring ---
def test_triangle_inside_edges_and_uv(rt, O):
    """Triangle::hit triangle.rs:51-77: plane t, three same-side tests with >= 0, (u,v) from x/y only."""
    d, ref, _ = one(rt, lambda b: b.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), b.lambertian((1, 1, 1))))
    r = O.hit(d, ref, (0.25, 0.25, 1.0), (0, 0, -1))
    assert r.hit and r.t == 1.0 and list(r.normal) == [0, 0, 1] and r.u == 0.25 and r.v == 0.25
    assert O.hit(d, ref, (0.5, 0.5, 1.0), (0, 0, -1)).hit       # on the hypotenuse: inside (>= 0)
    assert O.hit(d, ref, (0.0, 0.0, 1.0), (0, 0, -1)).hit       # on a vertex
    assert not O.hit(d, ref, (0.51, 0.51, 1.0), (0, 0, -1)).hit
    assert not O.hit(d, ref, (0.25, 0.25, 1.0), (1, 0, 0)).hit  # parallel: t = x/0 -> inf or NaN -> rejected


def test_ring_radii(rt, O):
    """Ring::hit ring.rs:36-53: plane y = 0 about the origin, (r-t)^2 <= x^2+z^2 <= (r+t)^2."""
    d, ref, _ = one(rt, lambda b: b.ring(10.0, 1.0, b.lambertian((1, 1, 1))))
    assert O.hit(d, ref, (10.0, 5.0, 0.0), (0, -1, 0)).t == 5.0
    assert O.hit(d, ref, (9.0, 5.0, 0.0), (0, -1, 0)).hit and O.hit(d, ref, (11.0, 5.0, 0.0), (0, -1, 0)).hit
    assert not O.hit(d, ref, (8.99, 5.0, 0.0), (0, -1, 0)).hit and not O.hit(d, ref, (11.01, 5.0, 0.0), (0, -1, 0)).hit
    assert O.hit(d, ref, (10.0, -5.0, 0.0), (0, 1, 0)).front_face == 0


# ----------------------------------------------------------------- movers ---
def test_translate_roundtrip_and_front_face_quirk(rt, O):
    """Translate::hit mod.rs:165-174 re-runs set_face_normal on the already forwarded normal,
    so front_face comes out true even for a hit from inside."""
    b = rt.DescBuilder()
    sph = b.sphere((0, 0, 0), 1.0, b.lambertian((1, 1, 1)))
    tr = b.translate(sph, (0, 0, -5))
    b.set_root(tr)
    d = b.desc()
    r = O.hit(d, tr, (0, 0, 0), (0, 0, -1))
    assert r.t == 4.0 and list(r.p) == [0, 0, -4] and list(r.normal) == [0, 0, 1] and r.front_face == 1
    inside = O.hit(d, tr, (0, 0, -5), (0, 0, -1))
    assert inside.t == 1.0 and list(inside.normal) == [0, 0, 1] and inside.front_face == 1   # plain sphere says 0
    assert O.hit(d, sph, (0, 0, 0), (0, 0, -1), t_min=0.5).front_face == 0


def test_rotate_y_matches_rotation_matrix(rt, O):
    """RotateY::hit mod.rs:235-264."""
    ang = math.radians(30.0)
    b = rt.DescBuilder()
    rect = b.rect(F.RT_RECT_XY, -1.0, 1.0, -1.0, 1.0, 0.0, b.lambertian((1, 1, 1)))
    rot = b.rotate_y(rect, math.sin(ang), math.cos(ang))
    b.set_root(rot)
    d = b.desc()
    # a ray along the rotated normal through the rotated centre
    n = np.array([math.sin(ang), 0.0, math.cos(ang)])
    r = O.hit(d, rot, tuple(3 * n), tuple(-n))
    assert r.hit and r.t == pytest.approx(3.0, abs=1e-12)
    assert np.allclose(r.normal, n, atol=1e-12) and np.allclose(r.p, [0, 0, 0], atol=1e-12)


def test_zoom_scales_origin_only(rt, O):
    """Zoom::hit mod.rs:321-330: orig/rate, dir and t untouched, p*rate — literally."""
    b = rt.DescBuilder()
    sph = b.sphere((0, 0, 0), 1.0, b.lambertian((1, 1, 1)))
    z = b.zoom(sph, 2.0)
    b.set_root(z)
    d = b.desc()
    r = O.hit(d, z, (0, 0, 10), (0, 0, -1))
    assert r.t == 4.0                                   # object-space t from z = 5, not the world distance 8
    assert list(r.p) == [0, 0, 2.0]


def test_flip_face_only_toggles_the_flag(rt, O):
    """FlipFace::hit mod.rs:281-288 (and DiffuseLight emits only when front_face, material/mod.rs:174-180)."""
    b = rt.DescBuilder()
    light = b.diffuse_light((7, 7, 7))
    plain = b.rect(F.RT_RECT_XZ, -1, 1, -1, 1, 5.0, light)
    flipped = plain | F.RT_REF_FLIP
    b.set_root(plain)
    d = b.desc()
    # seen from below, a ceiling XZRect (outward normal +y) is back-facing: that is why the reference
    # wraps its ceiling lights in FlipFace (scene.rs:172, 209, 289)
    up = O.hit(d, plain, (0, 0, 0), (0, 1, 0))
    assert up.front_face == 0 and list(up.normal) == [0, -1, 0]
    fl = O.hit(d, flipped, (0, 0, 0), (0, 1, 0))
    assert fl.front_face == 1 and list(fl.normal) == [0, -1, 0]      # the flag flips, the normal does not
    assert list(O.ray_color(d, (0, 0, 0), (0, 1, 0))) == [0, 0, 0]
    b.set_root(flipped)
    assert list(O.ray_color(b.desc(), (0, 0, 0), (0, 1, 0))) == [7, 7, 7]


# ------------------------------------------------------------------ medium ---
def test_medium_draws_and_clipping(rt, O):
    """ConstantMedium::hit constantmedium.rs:49-83: no draw unless both boundary hits exist and the clipped
    interval is non-empty; one draw otherwise; record = (p, normal (1,0,0), front_face, u=v=0)."""
    b = rt.DescBuilder()
    bound = b.sphere((0, 0, -5), 1.0, b.dielectric(1.5))
    med = b.medium(bound, 1000.0, b.isotropic((1, 1, 1)))      # dense: always scatters inside
    b.set_root(med)
    d = b.desc()
    r = O.hit(d, med, (0, 0, 0), (0, 0, -1))
    assert r.hit and r.rng_draws == 1 and 4.0 <= r.t <= 6.0
    assert list(r.normal) == [1, 0, 0] and r.front_face == 1 and r.u == 0.0 and r.v == 0.0
    assert r.p[2] == -r.t
    miss = O.hit(d, med, (5, 0, 0), (0, 0, -1))
    assert not miss.hit and miss.rng_draws == 0               # boundary never hit
    clipped = O.hit(d, med, (0, 0, 0), (0, 0, -1), t_max=3.0)
    assert not clipped.hit and clipped.rng_draws == 0         # interval empty after clamping to t_max
    # origin inside the boundary: rec1.t < 0 clamps to t_min then max(.,0)
    ins = O.hit(d, med, (0, 0, -5), (0, 0, -1))
    assert ins.hit and ins.rng_draws == 1 and 0.001 <= ins.t <= 1.0
    # thin medium: hit_distance = -ln(rnd)/density usually exceeds the chord -> miss but the draw is spent
    b2 = rt.DescBuilder()
    bound2 = b2.sphere((0, 0, -5), 1.0, b2.dielectric(1.5))
    med2 = b2.medium(bound2, 1e-9, b2.isotropic((1, 1, 1)))
    b2.set_root(med2)
    thin = O.hit(b2.desc(), med2, (0, 0, 0), (0, 0, -1))
    assert not thin.hit and thin.rng_draws == 1


def test_medium_hit_distance_formula(rt, O):
    """hit_distance = neg_inv_density * ln(rnd) with rnd = gen::<f64>() (constantmedium.rs:60-61)."""
    b = rt.DescBuilder()
    med = b.medium(b.sphere((0, 0, -5), 1.0, b.dielectric(1.5)), 2.0, b.isotropic((1, 1, 1)))
    b.set_root(med)
    d = b.desc()
    state = 99
    f = (C.c_double * 1)()
    O.lib().rto_rng_f64(state, f, 1)
    expect = 4.0 + (-1.0 / 2.0) * math.log(f[0])
    r = O.hit(d, med, (0, 0, 0), (0, 0, -1), rng_state=state)
    if expect <= 6.0:
        assert r.hit and r.t == pytest.approx(expect, rel=1e-15)
    else:
        assert not r.hit


# --------------------------------------------------------------------- BVH ---
def test_bvh_closest_hit_and_right_wins_ties(rt, O):
    """BvhNode::hit bvh/mod.rs:86-101: right searched with t_max = recl.t; an equal t on the right replaces the left hit."""
    b = rt.DescBuilder()
    m0, m1 = b.lambertian((1, 0, 0)), b.lambertian((0, 1, 0))
    s0 = b.sphere((0, 0, -5), 1.0, m0)
    s1 = b.sphere((0, 0, -5), 1.0, m1)            # identical geometry, different material
    n = b.node((-1, -1, -6), (1, 1, -4), s0, s1)
    b.set_root(n)
    d = b.desc()
    st = F.rt_stats()
    r = O.hit(d, n, (0, 0, 0), (0, 0, -1), stats=st)
    assert r.hit and r.t == 4.0 and r.mat == m1   # the right child wins the tie
    assert st.node_visits == 1 and st.prim_tests[F.RT_KIND_SPHERE] == 2
    # box missed -> children never tested
    st = F.rt_stats()
    assert not O.hit(d, n, (5, 0, 0), (0, 0, -1), stats=st).hit
    assert st.node_visits == 1 and st.prim_tests[F.RT_KIND_SPHERE] == 0


def test_aabb_slab_edge_cases(rt, O):
    """AABB::hit aabb.rs:15-32: zero direction component (inv_d = inf), negative inv_d swap, t_max <= t_min -> miss."""
    b = rt.DescBuilder()
    sph = b.sphere((0, 0, -5), 0.5, b.lambertian((1, 1, 1)))
    n = b.node((-1, -1, -6), (1, 1, -4), sph, sph)
    b.set_root(n)
    d = b.desc()
    st = F.rt_stats()
    assert O.hit(d, n, (0, 0, 0), (0, 0, -1), stats=st).hit            # dir.x = dir.y = 0 inside the slabs
    assert not O.hit(d, n, (2, 0, 0), (0, 0, -1)).hit                   # outside the x slab with dir.x = 0
    assert O.hit(d, n, (0, 0, -10), (0, 0, 1)).t == 4.5                 # negative-direction twin (swap branch)
    st = F.rt_stats()
    assert not O.hit(d, n, (0, 0, 0), (0, 0, -1), t_max=3.9, stats=st).hit
    assert st.prim_tests[F.RT_KIND_SPHERE] == 0                         # box entry 4.0 >= t_max: pruned at the node


def test_span1_leaf_is_tested_twice(rt, O):
    """bvh/mod.rs:44-47: a one-object node stores it as left AND right, so a medium there draws twice."""
    b = rt.DescBuilder()
    med = b.medium(b.sphere((0, 0, -5), 1.0, b.dielectric(1.5)), 1e-9, b.isotropic((1, 1, 1)))
    n = b.node((-1, -1, -6), (1, 1, -4), med, med)
    b.set_root(n)
    st = F.rt_stats()
    r = O.hit(b.desc(), n, (0, 0, 0), (0, 0, -1), stats=st)
    assert r.rng_draws == 2 and st.prim_tests[F.RT_KIND_MEDIUM] == 2 and st.prim_tests[F.RT_KIND_SPHERE] == 4


# ------------------------------------------------------ materials / pdfs ---
def rec_at(O, p=(0, 0, 0), normal=(0, 1, 0), front=True, u=0.0, v=0.0, t=1.0):
    from oracle.oracle_ffi import rto_hit_record
    r = rto_hit_record()
    r.hit = 1
    r.front_face = 1 if front else 0
    for i in range(3):
        r.p[i] = p[i]
        r.normal[i] = normal[i]
    r.t, r.u, r.v = t, u, v
    return r


def scatter(O, d, mat, ray, rec, state=5):
    out_ray = (C.c_double * 7)()
    att = (C.c_double * 3)()
    emit = (C.c_double * 3)()
    kind = O.lib().rto_scatter(C.byref(d), mat, (C.c_double * 7)(*ray), C.byref(rec), state, out_ray, att, emit)
    return kind, list(out_ray), list(att), list(emit)


def test_metal_reflects_unit_direction_time_zero(rt, O):
    """Metal::scatter material/mod.rs:85-96: reflect(unit(dir), n) + fuzz*random_in_unit_sphere; ray time = 0.;
    the sphere is sampled even when fuzz == 0."""
    b = rt.DescBuilder()
    m = b.metal((0.8, 0.6, 0.4), 0.0)
    b.set_root(b.sphere((0, 0, 0), 1, m))
    d = b.desc()
    s = math.sqrt(0.5)
    kind, ray, att, emit = scatter(O, d, m, (0, 1, 0, 2, -2, 0, 0.7), rec_at(O))
    assert kind == 1 and att == [0.8, 0.6, 0.4] and emit == [0, 0, 0]
    assert ray[3:6] == pytest.approx([s, s, 0.0], abs=1e-15) and ray[6] == 0.0   # time dropped to 0
    assert b.pools["materials"][b.metal((1, 1, 1), 3.0)].param == 1.0          # fuzz clamped, mod.rs:79


def test_dielectric_reflect_refract(rt, O):
    """Dielectric::scatter material/mod.rs:120-147 + reflectance :112-116: always one draw; att = 1."""
    b = rt.DescBuilder()
    m = b.dielectric(1.5)
    b.set_root(b.sphere((0, 0, 0), 1, m))
    d = b.desc()
    # normal incidence, front face: eta = 1/1.5, R0 = 0.04 -> nearly always refracts straight through
    refr = 0
    for state in range(200):
        kind, ray, att, _ = scatter(O, d, m, (0, 1, 0, 0, -1, 0, 0.3), rec_at(O), state)
        assert kind == 1 and att == [1, 1, 1] and ray[6] == 0.3
        if ray[4] < 0:
            refr += 1
            assert ray[3:6] == pytest.approx([0, -1, 0], abs=1e-15)
        else:
            assert ray[3:6] == pytest.approx([0, 1, 0], abs=1e-15)
    assert 180 <= refr <= 200                                   # ~96 % refraction
    # total internal reflection from inside at a grazing angle (eta = 1.5, sin > 1/1.5)
    g = (math.sin(1.2), -math.cos(1.2), 0.0)
    kind, ray, _, _ = scatter(O, d, m, (0, 1, 0) + g + (0.0,), rec_at(O, front=False))
    assert ray[3:6] == pytest.approx([g[0], -g[1], 0.0], abs=1e-15)


def test_lambertian_and_light_records(rt, O):
    b = rt.DescBuilder()
    lam = b.lambertian((0.1, 0.2, 0.3))
    light = b.diffuse_light((4, 5, 6))
    iso = b.isotropic((0.5, 0.5, 0.5))
    b.set_root(b.sphere((0, 0, 0), 1, lam))
    d = b.desc()
    kind, _, att, emit = scatter(O, d, lam, (0, 1, 0, 0, -1, 0, 0), rec_at(O))
    assert kind == 2 and att == [0.1, 0.2, 0.3] and emit == [0, 0, 0]
    kind, _, _, emit = scatter(O, d, light, (0, 1, 0, 0, -1, 0, 0), rec_at(O, front=True))
    assert kind == 0 and emit == [4, 5, 6]
    kind, _, _, emit = scatter(O, d, light, (0, 1, 0, 0, -1, 0, 0), rec_at(O, front=False))
    assert kind == 0 and emit == [0, 0, 0]
    kind, ray, att, _ = scatter(O, d, iso, (0, 1, 0, 0, -1, 0, 0.4), rec_at(O))
    assert kind == 1 and att == [0.5, 0.5, 0.5] and ray[6] == 0.4
    assert np.linalg.norm(ray[3:6]) < 1.0                       # un-normalised point in the unit sphere


def test_rect_light_pdf_and_sampling(rt, O):
    """XZRect::pdf_value / random aarect.rs:157-176; HittableList mean + index draw mod.rs:121-132."""
    b = rt.DescBuilder()
    light = b.rect(F.RT_RECT_XZ, -1.0, 1.0, -1.0, 1.0, 5.0, b.diffuse_light((1, 1, 1)))
    b.set_root(light)
    b.light(light)
    d = b.desc()
    o = (C.c_double * 3)(0, 0, 0)
    v = (C.c_double * 3)(0, 2, 0)
    # straight up: d^2 = t^2 |v|^2 = 25, cos = 1, area = 4 -> 6.25
    assert O.lib().rto_lights_pdf_value(C.byref(d), o, v) == 6.25
    assert O.lib().rto_lights_pdf_value(C.byref(d), o, (C.c_double * 3)(1, 0.1, 0)) == 0.0
    out = (C.c_double * 3)()
    for state in range(50):
        n = O.lib().rto_lights_random(C.byref(d), o, state, out)
        # index draw (rand's zone test rejects half the words when the list has one light) + 2 coordinates
        assert n >= 3 and out[1] == 5.0 and -1 <= out[0] < 1 and -1 <= out[2] < 1
    # Monte-Carlo normalisation: E[1/pdf over directions sampled from the light] ~ solid angle consistency
    acc = 0.0
    N = 4000
    for state in range(N):
        O.lib().rto_lights_random(C.byref(d), o, 1000 + state, out)
        p = O.lib().rto_lights_pdf_value(C.byref(d), o, out)
        acc += 1.0 / p
    solid = acc / N
    assert solid == pytest.approx(4 * math.atan(1 / (5 * math.sqrt(27))) * 1.0, rel=0.05)   # solid angle of a 2x2 square at h = 5


def test_sphere_light_pdf(rt, O):
    """Sphere::pdf_value sphere.rs:75-83 = 1 / (2 pi (1 - cos_max))."""
    b = rt.DescBuilder()
    s = b.sphere((0, 10, 0), 2.0, b.diffuse_light((1, 1, 1)))
    b.set_root(s)
    b.light(s)
    d = b.desc()
    o = (C.c_double * 3)(0, 0, 0)
    cos_max = math.sqrt(1 - 4 / 100)
    assert O.lib().rto_lights_pdf_value(C.byref(d), o, (C.c_double * 3)(0, 1, 0)) == pytest.approx(1 / (2 * math.pi * (1 - cos_max)), rel=1e-15)
    assert O.lib().rto_lights_pdf_value(C.byref(d), o, (C.c_double * 3)(1, 0, 0)) == 0.0
    out = (C.c_double * 3)()
    for state in range(100):
        assert O.lib().rto_lights_random(C.byref(d), o, state, out) >= 3        # index (>= 1 word) + r1 + r2
        assert O.lib().rto_lights_pdf_value(C.byref(d), o, out) > 0             # every sample points at the sphere


def test_wrapped_or_flipped_lights_fall_back_to_trait_defaults(rt, O):
    """hittable/mod.rs:62-67: wrappers do not forward pdf_value / random -> 0 and (1,0,0)."""
    b = rt.DescBuilder()
    rect = b.rect(F.RT_RECT_XZ, -1.0, 1.0, -1.0, 1.0, 5.0, b.diffuse_light((1, 1, 1)))
    b.set_root(rect)
    b.light(rect | F.RT_REF_FLIP)
    d = b.desc()
    o = (C.c_double * 3)(0, 0, 0)
    assert O.lib().rto_lights_pdf_value(C.byref(d), o, (C.c_double * 3)(0, 1, 0)) == 0.0
    out = (C.c_double * 3)()
    O.lib().rto_lights_random(C.byref(d), o, 1, out)
    assert list(out) == [1.0, 0.0, 0.0]


# --------------------------------------------------------------- textures ---
def test_checker_and_image_textures(rt, O):
    """CheckerTexture texture/mod.rs:51-60 (sin(10x) sin(10y) sin(10z) < 0 -> odd);
    ImageTexture::value :110-139 (truncation, clamp to W-1/H-1, scale 1/255.999, bottom-up rows)."""
    b = rt.DescBuilder()
    odd, even = b.solid((1, 0, 0)), b.solid((0, 1, 0))
    chk = b.checker(odd, even)
    img = np.zeros((2, 4, 3), dtype=np.uint8)
    img[0, :, 0] = [10, 20, 30, 40]          # storage row 0 = bottom row
    img[1, :, 0] = [50, 60, 70, 255]
    tex = b.image(img)
    b.set_root(b.sphere((0, 0, 0), 1, b.lambertian(tex=chk)))
    d = b.desc()
    out = (C.c_double * 3)()

    def val(t, u, v, p=(0, 0, 0)):
        O.lib().rto_texture_value(C.byref(d), t, u, v, (C.c_double * 3)(*p), out)
        return list(out)

    assert val(chk, 0, 0, (0.1, 0.1, 0.1)) == [0, 1, 0]        # all sines positive -> even
    assert val(chk, 0, 0, (-0.1, 0.1, 0.1)) == [1, 0, 0]       # one negative -> odd
    assert val(chk, 0, 0, (0.0, 0.1, 0.1)) == [0, 1, 0]        # product 0 is not < 0 -> even
    s = 1 / 255.999
    assert val(tex, 0.0, 0.0)[0] == 10 * s
    assert val(tex, 0.26, 0.0)[0] == 20 * s                    # i = trunc(0.26 * 4) = 1
    assert val(tex, 1.0, 1.0)[0] == 255 * s                    # u = 1 -> i = 4 -> clamped to 3; v = 1 -> j clamped to 1
    assert val(tex, -3.0, 7.0)[0] == 50 * s                    # clamp(u,0,1), clamp(v,0,1)
    assert val(tex, float("nan"), 0.0)[0] == 10 * s            # NaN as usize = 0


def test_perlin_with_known_tables(rt, O):
    """Perlin::noise / trilinear_interp / turb perlin.rs:52-112 incl. the double Hermite smoothing and
    `&255` on negative lattice coordinates."""
    rng = np.random.default_rng(3)
    vec = rng.uniform(-1, 1, (256, 3))
    vec /= np.linalg.norm(vec, axis=1)[:, None]
    px, py, pz = (rng.permutation(256) for _ in range(3))
    b = rt.DescBuilder()
    pid = b.perlin(vec, px, py, pz)
    pl = b.pools["perlins"][pid]

    def ref_noise(p):
        f = np.floor(p)
        u, v, w = p - f
        u, v, w = (x * x * (3 - 2 * x) for x in (u, v, w))                    # first smoothing
        i, j, k = (int(x) for x in f)
        uu, vv, ww = (x * x * (3 - 2 * x) for x in (u, v, w))                 # second smoothing
        acc = 0.0
        for a in range(2):
            for bb in range(2):
                for c in range(2):
                    g = vec[px[(i + a) & 255] ^ py[(j + bb) & 255] ^ pz[(k + c) & 255]]
                    wv = np.array([u - a, v - bb, w - c])                      # once-smoothed u,v,w in the weight
                    acc += g.dot(wv) * (a * uu + (1 - a) * (1 - uu)) * (bb * vv + (1 - bb) * (1 - vv)) * (c * ww + (1 - c) * (1 - ww))
        return acc

    for p in [(0.3, 0.7, 0.2), (-0.3, 5.25, -17.75), (255.5, -256.5, 1000.125), (3.0, 4.0, 5.0)]:
        got = O.lib().rto_perlin_noise(C.byref(pl), (C.c_double * 3)(*p))
        assert got == pytest.approx(ref_noise(np.array(p)), abs=1e-14)
    p = np.array((1.3, -2.7, 0.9))
    turb = abs(sum(0.5 ** i * ref_noise(p * 2 ** i) for i in range(7)))
    assert O.lib().rto_perlin_turb(C.byref(pl), (C.c_double * 3)(*p), 7) == pytest.approx(turb, abs=1e-13)


# ------------------------------------------------------ camera / integrator ---
def test_camera_new_and_get_ray(rt, O):
    """Camera::new camera.rs:24-62 and get_ray :64-73 (the disk loop runs even at aperture 0, then one time draw)."""
    cam = rt.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 2.0, 0.0, 1.0, 0.0, 1.0)
    assert list(cam.horizontal) == pytest.approx([4.0, 0, 0], abs=1e-14)      # vfov 90 -> h = 1 -> width = 2*2
    assert list(cam.vertical) == pytest.approx([0, 2.0, 0], abs=1e-14)
    assert list(cam.lower_left_corner) == pytest.approx([-2, -1, -1], abs=1e-14)
    assert cam.lens_radius == 0.0
    ray = (C.c_double * 7)()
    draws = O.lib().rto_get_ray(C.byref(cam), 0.5, 0.5, 11, ray)
    assert draws >= 3 and (draws - 1) % 2 == 0                                # pairs for the disk + 1 for the time
    assert list(ray)[:6] == pytest.approx([0, 0, 0, 0, 0, -1], abs=1e-14) and 0.0 <= ray[6] < 1.0
    cam2 = rt.camera_new((0, 0, 0), (0, 0, -1), (0, 1, 0), 90.0, 2.0, 2.0, 1.0, 0.0, 1.0)
    O.lib().rto_get_ray(C.byref(cam2), 0.5, 0.5, 11, ray)
    off = np.array(ray[:3])
    assert 0 < np.linalg.norm(off) < 1.0 and list(np.array(ray[3:6]) + off) == pytest.approx([0, 0, -1], abs=1e-14)


def test_ray_color_miss_depth_and_light(rt, O):
    """ray_color main.rs:233-278: depth <= 0 -> black; miss -> background; light -> emitted."""
    b = rt.DescBuilder()
    light = b.rect(F.RT_RECT_XZ, -1, 1, -1, 1, 5.0, b.diffuse_light((3, 2, 1)))
    b.set_root(light)
    b.light(light)
    d = b.desc()
    assert list(O.ray_color(d, (0, 0, 0), (0, -1, 0), background=(0.1, 0.2, 0.3))) == [0.1, 0.2, 0.3]
    assert list(O.ray_color(d, (0, 0, 0), (0, -1, 0), background=(0.1, 0.2, 0.3), depth=0)) == [0, 0, 0]
    assert list(O.ray_color(d, (0, 10, 0), (0, -1, 0))) == [3, 2, 1]


def test_white_furnace(rt, O):
    """Closed diffuse sphere of albedo a, seen from inside, background irrelevant; only the depth cut loses
    energy: with an emissive environment replaced by 'every path ends black', a furnace test needs an emitter.
    Here: Lambertian sphere (albedo a) around the origin inside a huge emissive sphere of radiance 1 that the
    inner sphere hides -> radiance must be 0; without the inner sphere it is exactly the emitter's 1."""
    b = rt.DescBuilder()
    env = b.sphere((0, 0, 0), 100.0, b.diffuse_light((1, 1, 1)), flip=True)   # emits towards the inside
    b.set_root(env)
    d = b.desc()
    for state in range(5):
        assert list(O.ray_color(d, (0, 0, 0), (0.3, 0.4, -0.5), rng_state=state)) == [1, 1, 1]
    # a Lambertian ball in a uniform radiance-1 environment reflects albedo * 1 (cosine-only mode, pdf cancels)
    b = rt.DescBuilder()
    env = b.sphere((0, 0, 0), 100.0, b.diffuse_light((1, 1, 1)), flip=True)
    ball = b.sphere((0, 0, -5), 1.0, b.lambertian((0.5, 0.25, 0.125)))
    n = b.node((-100, -100, -100), (100, 100, 100), env, ball)
    b.set_root(n)
    d = b.desc()
    acc = np.zeros(3)
    N = 400
    for state in range(N):
        acc += O.ray_color(d, (0, 0, 0), (0, 0, -1), rng_state=state)
    # first bounce returns albedo exactly (spdf / pdf == 1 up to rounding) unless the bounce re-hits the ball
    assert acc / N == pytest.approx([0.5, 0.25, 0.125], rel=1e-12)


def test_write_color(rt, O):
    """write_color main.rs:280-299."""
    def wc(v, spp=4):
        return list(O.write_color(np.array([v, v, v]), spp)[0:3])
    assert wc(0.0) == [0, 0, 0]
    assert wc(float("nan")) == [0, 0, 0]                     # NaN scrubbed on the summed pixel
    assert wc(4.0) == [255, 255, 255]                        # sqrt(1) -> clamp 0.999 -> floor(255.74)
    assert wc(1e9) == [255, 255, 255] and wc(float("inf")) == [255, 255, 255]
    assert wc(1.0) == [127, 127, 127]                        # sqrt(0.25) = 0.5 -> floor(127.9995)
    assert wc(-1.0) == [0, 0, 0]                             # sqrt(negative) = NaN -> clamp keeps NaN -> as u8 = 0
    assert list(rt.write_color(np.array([1.0, 4.0, float("nan")]), 4)) == [127, 255, 0]
