"""How often the single-precision slab test of the traversal kernel hands a node step to the double-precision one.

    RT2022_LIB=$PWD/raytracer_2022_amd/variants_lean/C_f32_census.so python3 tools/f32_census.py [scene W H spp [param]] ...

Needs a library built with -DRT2022_F32_CENSUS (make EXTRA='-DRT2022_F32_CENSUS -DRT2022_F32_SLABS=2': the test in every
instance that holds the whole node table, not only the sphere-only one): rt_debug_f32_slabs then returns the node steps of
the fast path that took the single-precision test and those it left undecided, over the timed (counter-free) render made here.
"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt
from raytracer_2022_amd import _ffi as F

CASES = [("final_scene", 800, 800, 20, 0), ("random_scene", 1200, 800, 20, 0), ("cornell_box", 600, 600, 20, 0), ("cornell_smoke", 600, 600, 20, 0)]
if len(sys.argv) > 4:
    CASES = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 0)]
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")
for name, W, H, spp, param in CASES:
    s = rt.HostScene(name, seed=2022, param=param, assets_dir=assets if os.path.isdir(assets) else None)
    cam, bg = s.default_view(W / H)
    rows = np.arange(H, dtype=np.uint32)
    dev = rt.DeviceScene(s.desc)
    v = (C.c_uint64 * 5)()
    F.check(F.lib().rt_debug_f32_slabs(v))                    # (clears the counters)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
    dev.render(cam, p, rows, want_stats=False)
    F.check(F.lib().rt_debug_f32_slabs(v))
    tv = dev.trace_variant()
    if not v[3]:
        print(name, "this library has no single-precision slab test"); continue
    if not v[2]:
        print(name, "this library does not count (build with EXTRA=-DRT2022_F32_CENSUS)"); continue
    print("%-14s %dx%dx%d  nodes in LDS %d  fast-path node steps %d  undecided %d  = 1 in %.0f   verdicts differing from the double-precision test: %d" % (
        name, W, H, spp, tv["nodes_in_lds"], v[0], v[1], v[0] / max(v[1], 1), v[4]), flush=True)
