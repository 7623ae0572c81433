"""Sphere-only scenes between the all-in-LDS instance (up to 600 nodes) and the table's capacity (1 740 nodes): the double-precision
whole-table kernel against the plain single-precision one (tuning bit 28).   python3 tools/mid_spheres_ab.py [k ...]   (grid half-size)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt
W, H, spp = 1200, 800, 100
base = 18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20)
for k in [int(a) for a in sys.argv[1:]] or [12, 14]:
    s = rt.HostScene("random_scene", seed=2022, param=k)
    cam, bg = s.default_view(W / H)
    rows = np.arange(H, dtype=np.uint32)
    dev = rt.DeviceScene(s.desc)
    pc = rt.make_params(W, H, 4, 50, bg, seed=2022, spp_chunk=1)
    _, st = dev.render(cam, pc, rows, want_stats=True)
    rays = st.rays / 4 * spp
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
    for rep in range(2):
        for label, word in (("table", base), ("plain", base | (1 << 28))):
            dev.set_tuning(word)
            v = dev.trace_variant()
            dev.render(cam, pc, rows)
            t0 = time.perf_counter(); dev.render(cam, p, rows); dt = time.perf_counter() - t0
            print("k %d: %d spheres, %d nodes, %s %s: %.1f ms, %.0f Mrays/s" % (k, s.desc.n_spheres + s.desc.n_moving_spheres, s.desc.n_nodes, label, v, dt * 1e3, rays / dt / 1e6), flush=True)
