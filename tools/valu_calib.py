"""One saturating vector-instruction kernel per mode (rt_debug_valu_probe); run under rocprofv3 by tools/valu_calib.sh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_2022_amd import _ffi as F
ITERS = 12500          # x 64 vector instructions per lane
for mode in range(5):
    F.check(F.lib().rt_debug_valu_probe(mode, ITERS))
print("ok")
