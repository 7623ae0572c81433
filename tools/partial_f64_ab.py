"""The partial node table (first 1 740 double-precision records in LDS, 1024-thread workgroups) against the plain kernels on a scene that keeps the
double-precision node test and does not fit LDS whole: N random spheres and one Boxes (so: FEAT volumes), BVH by median split.

    python3 tools/partial_f64_ab.py [n_spheres W H spp]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt
from raytracer_2022_amd import _ffi as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 600
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 64
rng = np.random.default_rng(13)
b = rt.DescBuilder()
mats = [b.lambertian((0.6, 0.6, 0.7)), b.metal((0.9, 0.8, 0.7), 0.0), b.dielectric(1.5)]
leaves = []
for i in range(n):
    c = rng.uniform(-5.0, 5.0, 3) + np.array([0.0, 0.0, -14.0])
    r = float(rng.uniform(0.05, 0.2))
    leaves.append((b.sphere(tuple(c), r, mats[i % 3]), tuple(c - r), tuple(c + r)))
leaves.append((b.box((-1.0, -1.0, -9.0), (1.0, 1.0, -8.0), mats[0]), (-1.0, -1.0, -9.0), (1.0, 1.0, -8.0)))

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
b.set_root(build(leaves)[0])
d = b.desc()
cam = rt.camera_new((0.0, 0.5, 2.0), (0.0, 0.0, -14.0), (0, 1, 0), 40.0, W / H, 0.0, 10.0, 0.0, 1.0)
rows = np.arange(H, dtype=np.uint32)
dev = rt.DeviceScene(d)
pc = rt.make_params(W, H, 4, 50, (0.7, 0.8, 1.0), seed=3, spp_chunk=1)
_, st = dev.render(cam, pc, rows, want_stats=True)
rays = st.rays / 4 * spp
p = rt.make_params(W, H, spp, 50, (0.7, 0.8, 1.0), seed=3, spp_chunk=1)
base = 18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20)
for rep in range(2):
    for label, word in (("partial table", base), ("plain kernels", base | (1 << 28))):
        dev.set_tuning(word)
        v = dev.trace_variant()
        dev.render(cam, pc, rows)
        t0 = time.perf_counter(); dev.render(cam, p, rows); dt = time.perf_counter() - t0
        print("%d nodes, %s %s: %.1f ms, %.0f Mrays/s" % (d.n_nodes, label, v, dt * 1e3, rays / dt / 1e6), flush=True)
