"""Scheduler census of the traversal kernel on a scene (counter build): rounds, lanes, lane utilisation per operation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt

name = sys.argv[1] if len(sys.argv) > 1 else 'final_scene'
W = int(sys.argv[2]) if len(sys.argv) > 2 else 800
H = int(sys.argv[3]) if len(sys.argv) > 3 else 800
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 50
param = int(sys.argv[5]) if len(sys.argv) > 5 else 0
assets = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'assets')
s = rt.HostScene(name, seed=2022, param=param, assets_dir=assets if os.path.isdir(assets) else None)
cam, bg = s.default_view(W / H)
rows = np.arange(H, dtype=np.uint32)
dev = rt.DeviceScene(s.desc)
p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
out, st = dev.render(cam, p, rows, want_stats=True)
d = st.as_dict()
print(name, W, H, spp, 'rays', d['rays'], 'nodes/ray %.2f' % (d['node_visits'] / d['rays']), 'prims', d['prim_tests'], flush=True)
tot_r = 0
for k, v in dev.census().items():
    print('  %-10s rounds %13d lanes %15d lanes/round %6.2f util %.3f' % (k, v[0], v[1], v[1] / max(v[0], 1), v[2]))
    tot_r += v[0]
print('  total rounds', tot_r, 'rounds per ray %.2f' % (tot_r * 64 / d['rays']))
