"""Device side: a scene resident in HBM and the render calls (include/rt2022.h)."""
import ctypes as C

import numpy as np

from . import _ffi as F


class DeviceScene:
    """rt_scene: the flattened scene copied into HBM on the current HIP device."""

    def __init__(self, desc):
        self._h = C.c_void_p()
        F.check(F.lib().rt_scene_create(C.byref(desc), C.byref(self._h)))

    def info(self):
        need, blocks = C.c_uint32(), C.c_int32()
        F.check(F.lib().rt_debug_scene_info(self._h, C.byref(need), C.byref(blocks)))
        return {"stack_need": need.value, "grid_blocks": blocks.value}

    def trace_variant(self):
        """The traversal kernel variant timed renders of this scene take (rt_debug_trace_variant)."""
        wg, st, nc, sp = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        F.check(F.lib().rt_debug_trace_variant(self._h, C.byref(wg), C.byref(st), C.byref(nc), C.byref(sp)))
        return {"workgroup_threads": wg.value, "stack_entries": st.value, "nodes_in_lds": nc.value, "spheres_in_lds": bool(sp.value & 1),
                "f32_slabs": bool(sp.value & 2)}

    def set_tuning(self, node_quorum=18 | (1 << 8) | (2 << 12) | (8 << 16) | (2 << 20) | (1 << 24), vote_weights=0):
        F.check(F.lib().rt_debug_set_tuning(self._h, node_quorum, vote_weights))

    def set_engine(self, engine, max_pool_blocks=0):
        """engine: "wavefront" (default) or "mega"."""
        F.check(F.lib().rt_debug_set_engine(self._h, {"mega": 0, "wavefront": 1}[engine], max_pool_blocks))

    def set_partial_ring(self, planes):
        """Planes of the partial-sum ring: 0 automatic, -1 never, n > 0 force (rt_debug_set_partial_ring)."""
        F.check(F.lib().rt_debug_set_partial_ring(self._h, planes))

    def pass_timing(self):
        """Probe of the last render made with tuning bit 29: dict of sums over the traversal passes (ms)."""
        o = (C.c_double * 5)()
        F.check(F.lib().rt_debug_pass_timing(self._h, o))
        return {"span_ms": o[0] / 1e5, "wave_life_ms": o[1] / 1e5, "wave_dry_ms": o[2] / 1e5, "passes": int(o[3]), "waves": int(o[4])}

    def census(self):
        """Scheduler census of the last counter run: {label: (rounds, lanes, utilisation)}."""
        r, l = (C.c_uint64 * 9)(), (C.c_uint64 * 9)()
        F.check(F.lib().rt_debug_census(self._h, r, l))
        names = ["node", "sphere", "rect", "box", "medium", "misc", "ctx", "done", "node_fast"]
        return {n: (r[i], l[i], (l[i] / (64.0 * r[i])) if r[i] else 0.0) for i, n in enumerate(names)}

    def render(self, cam, params, row_ids, want_stats=False, progress=None):
        """rt_render with host buffers → (n_rows, width, 3) float64 sums [, rt_stats].
        progress: a callable (worker, paths_done, paths_total), rt_params.progress_cb."""
        rows = np.ascontiguousarray(row_ids, dtype=np.uint32)
        p = F.rt_params.from_buffer_copy(params)
        p.n_rows = len(rows)
        p.row_ids = rows.ctypes.data
        if want_stats:
            p.flags |= F.RT_FLAG_COUNTERS
        if progress is not None:
            cb = F.PROGRESS_CB(lambda user, worker, done, total: progress(worker, done, total))      # (kept alive until the call returns)
            p.progress_cb = C.cast(cb, C.c_void_p).value
        out = np.empty((len(rows), p.width, 3), dtype=np.float64)
        st = F.rt_stats()
        F.check(F.lib().rt_render(self._h, C.byref(cam), C.byref(p), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)))
        return (out, st) if want_stats else out

    def render_device(self, cam, params, d_row_ids_ptr, n_rows, d_out_ptr, stream_ptr=None, stats=None, asynchronous=False):
        """rt_render_device: device pointers in, enqueue on `stream_ptr` (hipStream_t as int). asynchronous=True sets
        RT_FLAG_ASYNC: the call returns at once, a host thread of the library drives the passes, wait() joins it."""
        p = F.rt_params.from_buffer_copy(params)
        p.n_rows = n_rows
        p.row_ids = d_row_ids_ptr
        if asynchronous:
            p.flags |= F.RT_FLAG_ASYNC
        F.check(F.lib().rt_render_device(self._h, C.byref(cam), C.byref(p), C.c_void_p(d_out_ptr),
                                         C.c_void_p(stream_ptr or 0), C.byref(stats) if stats is not None else None))

    def wait(self, stream_ptr=None):
        F.check(F.lib().rt_render_wait(self._h, C.c_void_p(stream_ptr or 0)))

    def close(self):
        if self._h:
            F.lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceSceneSet:
    """rt_scene_set: a copy of the scene on every device of `device_mask`, rendered by ONE call (rt_render_multi)."""

    def __init__(self, desc, device_mask=1):
        self._h = C.c_void_p()
        F.check(F.lib().rt_scene_set_create(C.byref(desc), device_mask, C.byref(self._h)))

    def render(self, cam, params, row_ids, want_stats=False, progress=None):
        rows = np.ascontiguousarray(row_ids, dtype=np.uint32)
        p = F.rt_params.from_buffer_copy(params)
        p.n_rows = len(rows)
        p.row_ids = rows.ctypes.data
        if want_stats:
            p.flags |= F.RT_FLAG_COUNTERS
        if progress is not None:                     # (called from the devices' host threads, concurrently: ctypes takes the GIL for each call)
            cb = F.PROGRESS_CB(lambda user, worker, done, total: progress(worker, done, total))
            p.progress_cb = C.cast(cb, C.c_void_p).value
        out = np.empty((len(rows), p.width, 3), dtype=np.float64)
        st = F.rt_stats()
        F.check(F.lib().rt_render_multi(self._h, C.byref(cam), C.byref(p), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st)))
        return (out, st) if want_stats else out

    def close(self):
        if self._h:
            F.lib().rt_scene_set_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
