"""Calibration for the lane-utilisation formula: a fully-active f64 kernel (math probe) under rocprofv3 PMC."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from raytracer_2022_amd import _ffi as F
a = np.random.default_rng(0).uniform(-10, 10, 1 << 22)
out = np.empty_like(a)
for _ in range(3):
    F.check(F.lib().rt_debug_math_device(0, a.ctypes.data_as(C.POINTER(C.c_double)), None, out.ctypes.data_as(C.POINTER(C.c_double)), a.size))
print('ok')
