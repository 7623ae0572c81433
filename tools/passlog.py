"""Pass-timing probe of the wavefront engine on the headline scene (RT2022_PASS_LOG=1 prints one line per
traversal pass: span, mean wave lifetime / span, share of a wave's life after the chunk counter ran out, rays)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_2022_amd as rt

W = H = 800
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 400
s = rt.HostScene('final_scene', seed=2022)
cam, bg = s.default_view(1.0)
rows = rt.shuffled_rows(H, 1)
dev = rt.DeviceScene(s.desc)
dev.set_tuning((18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24)) | (1 << 29))      # the defaults + the probe
p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
out = dev.render(cam, p, rows)
t = dev.pass_timing()
print(t)
print('mean wave lifetime / pass span = %.3f   after the counter ran out / lifetime = %.3f' % (t['wave_life_ms'] / t['span_ms'], t['wave_dry_ms'] / t['wave_life_ms']))
