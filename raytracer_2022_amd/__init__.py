"""raytracer_2022_amd — MI355X-native path-tracing hot loop of Jerx2y/Raytracer-2022.

The product is `librt2022.so` (hand-written HIP kernels for gfx950 behind the C ABI
of include/rt2022.h, plus the C++ host mirror of the reference's scene-builder
API). This package is the thin Python plumbing over it: ctypes bindings, numpy /
torch buffer handling. Nothing here computes pixels.
"""
from . import _ffi
from ._ffi import RtError, lib, make_ref, ref_index, ref_kind  # noqa: F401
from .host import (DescBuilder, HostScene, camera_new, fill_image, load_image, make_params, shuffled_rows,  # noqa: F401
                   write_color, write_jpeg)
from .device import DeviceScene, DeviceSceneSet  # noqa: F401

__all__ = ["DescBuilder", "HostScene", "DeviceScene", "DeviceSceneSet", "camera_new", "fill_image", "make_params", "shuffled_rows",
           "write_color", "write_jpeg", "load_image", "RtError", "lib", "make_ref", "ref_kind", "ref_index"]
