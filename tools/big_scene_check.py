"""Large-scene check: wwscene with ~1.05 M triangles — parity at a small image, then a timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt
from oracle import oracle_ffi as O
s = rt.HostScene('wwscene', seed=2022, param=3)
d = s.desc
print('nodes', d.n_nodes, 'tris', d.n_triangles, flush=True)
dev = rt.DeviceScene(d)
print(dev.info(), flush=True)
W, H, spp = 96, 54, 2
cam, bg = s.default_view(W / H)
p = rt.make_params(W, H, spp, 50, bg, seed=2022)
rows = rt.shuffled_rows(H, 3)
ref, sr = O.render_cpu(d, cam, p, rows, n_threads=16, want_stats=True)
out, st = dev.render(cam, p, rows, want_stats=True)
print('parity: counters', st.as_dict() == sr.as_dict(), 'bits', int((out.view(np.uint64) != ref.view(np.uint64)).sum()), flush=True)
W, H, spp = 960, 540, 16
cam, bg = s.default_view(W / H)
p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
rows = np.arange(H, dtype=np.uint32)
out, st = dev.render(cam, p, rows, want_stats=True)
t = time.time(); out = dev.render(cam, p, rows); dt = time.time() - t
print('timing %dx%dx%d: rays %d  %.1f ms  %.1f Mrays/s  nodes/ray %.1f tri/ray %.1f' % (W, H, spp, st.rays, dt * 1e3, st.rays / dt / 1e6, st.node_visits / st.rays, st.prim_tests[5] / st.rays), flush=True)
