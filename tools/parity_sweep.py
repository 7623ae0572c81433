"""Wider parity sweep than the test suite has time for: the TIMED kernels (no counters: node-table variants, sphere
tables, plain kernels with bit 28) against the CPU oracle, bit for bit, over scenes x sizes x seeds x samples-per-item."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt
from oracle import oracle_ffi as O

bits = lambda a: np.ascontiguousarray(a).view(np.uint64)
default = 18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24)
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'assets')
cases = [("final_scene", 0, 160, 120, 6), ("random_scene", 0, 160, 100, 5), ("random_scene", 30, 128, 80, 3), ("cornell_box", 0, 120, 120, 8),
         ("cornell_smoke", 0, 96, 96, 6), ("two_perlin_spheres", 0, 96, 64, 6), ("earth", 0, 96, 64, 6), ("simple_light", 0, 96, 64, 8),
         ("wwscene", 1, 96, 54, 2)]
bad = 0
for name, param, W, H, spp in cases:
    for seed in (3, 11):
        s = rt.HostScene(name, seed=seed, param=param, assets_dir=assets if os.path.isdir(assets) else None)
        cam, bg = s.default_view(W / H)
        rows = rt.shuffled_rows(H, seed)
        dev = rt.DeviceScene(s.desc)
        for chunk in (0, 1, 3):
            p = rt.make_params(W, H, spp, 50, bg, seed=seed, spp_chunk=chunk)
            ref = O.render_cpu(s.desc, cam, p, rows, n_threads=os.cpu_count() or 8)
            for tune in (default, default | (1 << 28)):
                dev.set_tuning(tune)
                out = dev.render(cam, p, rows)
                ok = np.array_equal(bits(out), bits(ref))
                bad += 0 if ok else 1
                print("%-18s param %2d seed %2d chunk %d %s %s" % (name, param, seed, chunk, dev.trace_variant(), "ok" if ok else "MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
