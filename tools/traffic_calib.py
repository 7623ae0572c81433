"""One probe dispatch per access pattern (rt_debug_traffic_probe); run under rocprofv3 by tools/traffic_calib.sh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_2022_amd import _ffi as F
GIB = 1 << 30
buf = 8 * GIB
n = 48 * 1000 * 1000
for mode in range(7):
    F.check(F.lib().rt_debug_traffic_probe(mode, buf, n, 2022))
print("ok")
