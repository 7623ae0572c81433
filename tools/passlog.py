import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes as C
import raytracer_2022_amd as rt
from raytracer_2022_amd import _ffi as F
W = H = 800; spp = 400
s = rt.HostScene('final_scene', seed=2022)
cam, bg = s.default_view(1.0)
rows = rt.shuffled_rows(H, 1)
dev = rt.DeviceScene(s.desc)
dev.set_tuning((18 | (1 << 8) | (2 << 12) | (2 << 16) | (2 << 20) | (1 << 24)) | (1 << 29))
p = rt.make_params(W, H, spp, 50, bg, seed=2022, spp_chunk=1)
out = dev.render(cam, p, rows)
print(dev.pass_timing())
