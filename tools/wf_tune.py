"""Wavefront engine tuning probe: Mrays/s for pool sizes / chunk sizes / node quorum."""
import sys, os, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import raytracer_2022_amd as rt
from raytracer_2022_amd import _ffi as F

def run(dev, cam, p, rows, W):
    pr = F.rt_params.from_buffer_copy(p); pr.n_rows = len(rows); pr.row_ids = rows.ctypes.data
    o = np.empty((len(rows), W, 3)); st = F.rt_stats()
    F.check(F.lib().rt_render(dev._h, C.byref(cam), C.byref(pr), o.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)))
    return st.ms

name = sys.argv[1] if len(sys.argv) > 1 else 'final_scene'
W = H = 800; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 20
s = rt.HostScene(name, seed=2022)
cam, bg = s.default_view(W / H)
rows = np.arange(H, dtype=np.uint32)
dev = rt.DeviceScene(s.desc)
p0 = rt.make_params(W, H, min(spp, 20), 50, bg, seed=2022, spp_chunk=1)
dev.set_tuning()
out, st = dev.render(cam, p0, rows, want_stats=True)
rays = st.rays * (spp / min(spp, 20))      # (estimate: the counter pass runs at most 20 spp)
print('rays', rays, flush=True)
for k, v in dev.census().items():
    print('  census %-10s rounds %12d lanes %14d util %.3f' % (k, v[0], v[1], v[2]))
def Q(q=18, reps=1, tail=2, segs=6, shift=2, groups=1):
    return q | (reps << 8) | (tail << 12) | (segs << 16) | (shift << 20) | (groups << 24)
dev.set_tuning(Q() | (1 << 29))
pp = rt.make_params(W, H, min(spp, 200), 50, bg, seed=2022, spp_chunk=1)
ms = run(dev, cam, pp, rows, W)
t = dev.pass_timing()
print('pass timing (probe run %.1f ms): %s' % (ms, t))
print('  mean wave lifetime / pass span = %.3f   dry tail / lifetime = %.3f' % (t['wave_life_ms'] / t['span_ms'], t['wave_dry_ms'] / t['wave_life_ms']), flush=True)
# (segments in the pool, tuning word): pool sizes around the default, then quorum and list-class granularity
cfgs = [('wavefront', b, 1, Q(), 0, None) for b in (0, 3840, 5120, 10240)] + [('wavefront', 0, 1, Q(q=q, shift=sh), 0, None) for q, sh in ((12, 2), (24, 2), (18, 1), (18, 3))]
for eng, blocks, chunk, q, wts, pc in cfgs:
    dev.set_engine(eng, blocks); dev.set_tuning(q, wts)
    p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=chunk)
    run(dev, cam, p, rows, W)
    ms = run(dev, cam, p, rows, W)
    print(f'{eng} blocks={blocks} chunk={chunk} quorum={q} pace={pc}: {ms:.1f} ms  {rays / ms / 1e3:.1f} Mrays/s', flush=True)
