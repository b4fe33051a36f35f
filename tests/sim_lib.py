"""ctypes loader of tests/cpu_sim/libkernel_logic.so: the kernel headers' per-ray logic
compiled for the host so that it can be checked where no GPU exists.  Test-only."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import abi

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_sim")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", DIR], check=True, capture_output=True)
        L = C.CDLL(os.path.join(DIR, "libkernel_logic.so"))
        L.sim_scene_create.restype = C.c_void_p
        L.sim_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_uint32]
        L.sim_scene_destroy.argtypes = [C.c_void_p]
        L.sim_scene_nodes.argtypes = [C.c_void_p]
        L.sim_scene_depth.argtypes = [C.c_void_p]
        L.sim_trace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float,
                                      C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        L.sim_render.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.c_int, C.c_int, C.c_uint32,
                                 C.c_void_p, C.POINTER(C.c_ulonglong)]
        L.sim_set_stack_mode.argtypes = [C.c_int]
        L.sim_schedule_certificate.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float,
                                               C.POINTER(C.c_ulonglong)]
        _lib = L
    return _lib


def set_stack_mode(mode):
    """0: 32-bit stack entries in one column; 1: 16-bit entries (small scenes); 2: 32-bit entries
    that spill to a second column after four (the kernels' variant for deep trees in HBM)."""
    lib().sim_set_stack_mode(mode)


class SimScene:
    def __init__(self, arrays, flags=0):
        self.arrays = arrays
        d = arrays.desc()
        self.handle = lib().sim_scene_create(C.byref(d), flags)
        if not self.handle:
            raise RuntimeError("flatten failed")

    def __del__(self):
        if getattr(self, "handle", None):
            lib().sim_scene_destroy(self.handle)
            self.handle = None

    @property
    def n_nodes(self):
        return lib().sim_scene_nodes(self.handle)

    @property
    def depth(self):
        return lib().sim_scene_depth(self.handle)

    def trace(self, rays, tmin=0.001, tmax=np.inf):
        rays = np.ascontiguousarray(rays, abi.RAY_DT)
        hits = np.zeros(len(rays), abi.HIT_DT)
        b, p = C.c_ulonglong(), C.c_ulonglong()
        lib().sim_trace_batch(self.handle, rays.ctypes.data, len(rays), tmin, tmax, hits.ctypes.data,
                              C.byref(b), C.byref(p))
        return hits, b.value, p.value

    def schedule_certificate(self, rays, tmin=0.001, tmax=np.inf):
        """(rays whose closest hit some schedule of the decoupled kernels could lose, the same count
        without the culling slack, rays that differ from the whole-ray traversal, leaves reached)."""
        rays = np.ascontiguousarray(rays, abi.RAY_DT)
        out = (C.c_ulonglong * 4)()
        rc = lib().sim_schedule_certificate(self.handle, rays.ctypes.data, len(rays), tmin, tmax, out)
        if rc != 0:
            raise RuntimeError("one-level BVH scenes only")
        return tuple(int(v) for v in out)

    def render(self, cam, sample_count, sample_begin=0, seed=0):
        H, W = cam.image_height(), cam.image_width
        accum = np.zeros((H, W, 3), np.float32)
        rays = C.c_ulonglong()
        lib().sim_render(self.handle, C.byref(cam), sample_begin, sample_count, seed, accum.ctypes.data,
                         C.byref(rays))
        return accum, rays.value
