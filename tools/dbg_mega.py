import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt
from oracle import oracle_ffi as O
W, H, spp = 32, 18, 1
s = rt.HostScene('wwscene', seed=2022)
cam, bg = s.default_view(W / H)
p = rt.make_params(W, H, spp, 50, bg, seed=2022)
rows = np.arange(H, dtype=np.uint32)
ref, st_ref = O.render_cpu(s.desc, cam, p, rows, n_threads=8, want_stats=True)
print('ref ', st_ref.as_dict())
for eng in ('wavefront', 'mega'):
    dev = rt.DeviceScene(s.desc); dev.set_engine(eng)
    out, st = dev.render(cam, p, rows, want_stats=True)
    print(eng, st.as_dict(), dev.info())
    bad = np.argwhere((out.view(np.uint64) != ref.view(np.uint64)).any(axis=2))
    print(eng, 'bad pixels', len(bad), bad[:10].tolist())
