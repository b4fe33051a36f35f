"""ctypes binding of raytracing_amd/lib/librt_amd.so (C ABI: include/rt_amd.h).

This is the only way Python reaches the renderer; there is no Python or CPU
implementation behind it.  A missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

# RT_AMD_LIB: load another build of the same library (tuning experiments)
LIB_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib",
                                                       "librt_amd.so")

# every symbol include/rt_amd.h declares
EXPORTS = ["rt_last_error", "rt_abi_version", "rt_device_count", "rt_camera_initialize",
           "rt_scene_create", "rt_scene_destroy", "rt_scene_get_info", "rt_scene_read_nodes", "rt_trace_batch",
           "rt_trace_batch_device", "rt_trace_batch_sched", "rt_trace_batch_sched_device", "rt_closest_hits", "rt_closest_hits_device", "rt_render", "rt_render_device", "rt_render_device_static", "rt_tonemap_u8", "rt_tonemap_u8_device",
           "rt_film_create", "rt_film_destroy", "rt_film_render", "rt_film_snapshot", "rt_film_wait",
           "rt_film_upload", "rt_multi_create", "rt_multi_destroy", "rt_multi_render", "rt_profile_begin", "rt_profile_end", "rt_scene_file_save", "rt_scene_file_load",
           "rt_scene_file_free"]


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"librt_amd: error {code}: {message}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(abi.RT_ERR_IO, f"{LIB_PATH} is missing: build it with "
                          "`make -C raytracing_amd/csrc` (or __graft_entry__.build())")
        L = C.CDLL(LIB_PATH)
        L.rt_last_error.restype = C.c_char_p
        L.rt_camera_initialize.argtypes = [C.POINTER(abi.CameraDesc), C.POINTER(abi.CameraFrame)]
        L.rt_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int, C.c_uint32,
                                      C.POINTER(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_scene_get_info.argtypes = [C.c_void_p, C.POINTER(abi.SceneInfo)]
        L.rt_scene_read_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.rt_trace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                     C.c_void_p]
        L.rt_trace_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                            C.c_void_p, C.c_void_p]
        L.rt_trace_batch_sched.argtypes = L.rt_trace_batch.argtypes
        L.rt_trace_batch_sched_device.argtypes = L.rt_trace_batch_device.argtypes
        L.rt_closest_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.rt_closest_hits_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.RenderOpts),
                                C.c_void_p, C.POINTER(abi.RenderStats)]
        L.rt_render_device.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc),
                                       C.POINTER(abi.RenderOpts), C.c_void_p, C.c_void_p,
                                       C.POINTER(abi.RenderStats)]
        L.rt_render_device_static.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc),
                                              C.POINTER(abi.RenderOpts), C.c_void_p, C.c_void_p]
        L.rt_tonemap_u8.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        L.rt_tonemap_u8_device.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]
        if hasattr(L, "rt_film_create"):  # (older builds of the library, loaded through RT_AMD_LIB for A/B runs, lack it)
            L.rt_film_create.argtypes = [C.c_int, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
            L.rt_film_destroy.argtypes = [C.c_void_p]
            L.rt_film_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.RenderOpts)]
            L.rt_film_snapshot.argtypes = [C.c_void_p, C.c_int]
            L.rt_film_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
            L.rt_film_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.rt_multi_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_int32), C.c_int32, C.c_uint32,
                                      C.POINTER(C.c_void_p)]
        L.rt_multi_destroy.argtypes = [C.c_void_p]
        L.rt_multi_render.argtypes = [C.c_void_p, C.POINTER(abi.CameraDesc), C.POINTER(abi.RenderOpts), C.c_void_p,
                                      C.POINTER(abi.RenderStats)]
        L.rt_profile_begin.argtypes = [C.c_void_p]
        L.rt_profile_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.rt_scene_file_save.argtypes = [C.POINTER(abi.SceneDesc), C.c_char_p]
        L.rt_scene_file_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(abi.SceneDesc))]
        L.rt_scene_file_free.argtypes = [C.POINTER(abi.SceneDesc)]
        _lib = L
    return _lib


def check(rc):
    if rc != abi.RT_OK:
        raise RtError(rc, lib().rt_last_error().decode())


def device_count():
    return lib().rt_device_count()


def camera_initialize(cam):
    f = abi.CameraFrame()
    check(lib().rt_camera_initialize(C.byref(cam), C.byref(f)))
    return f


def tonemap_u8(accum, scale):
    accum = np.ascontiguousarray(accum, np.float32)
    out = np.zeros(accum.shape, np.uint8)
    check(lib().rt_tonemap_u8(accum.ctypes.data, accum.size // 3, scale, out.ctypes.data))
    return out


def tonemap_u8_device(device, d_accum_ptr, n_pixels, scale, d_rgb8_ptr, stream=0):
    """Device pointers in, device bytes out (e.g. torch tensors' data_ptr())."""
    check(lib().rt_tonemap_u8_device(device, C.c_void_p(d_accum_ptr), n_pixels, scale, C.c_void_p(d_rgb8_ptr),
                                     C.c_void_p(stream)))


class Scene:
    """Device-resident scene (rt_scene_create .. rt_scene_destroy)."""

    def __init__(self, arrays: abi.SceneArrays, device=0, flags=0):
        self.arrays = arrays
        d = arrays.desc()
        h = C.c_void_p()
        check(lib().rt_scene_create(C.byref(d), device, flags, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if getattr(self, "handle", None):
            lib().rt_scene_destroy(self.handle)
            self.handle = None

    __del__ = close

    def info(self):
        i = abi.SceneInfo()
        check(lib().rt_scene_get_info(self.handle, C.byref(i)))
        return i

    def nodes(self):
        """The nodes of the 4-wide BVH as an (n_nodes, 32) float32 array in the kernels' plane layout:
        words 8a + k and 8a + 4 + k are the lower and the upper plane of child k's box along axis a
        (a = 0, 1, 2); the four child references and the child count are the bit patterns of
        [:, 24:29].  node_boxes() decodes the planes."""
        out = np.zeros((self.info().n_nodes, 32), np.float32)
        check(lib().rt_scene_read_nodes(self.handle, out.ctypes.data, out.nbytes))
        return out

    def node_boxes(self):
        """(lo, hi, refs, n_children): lo / hi are (n_nodes, 4, 3) arrays with the corners of the four
        child boxes of every node (an unused child has an inverted box), refs (n_nodes, 4) int32."""
        n = self.nodes()
        planes = n[:, :24].reshape(-1, 3, 2, 4)  # [node, axis, lo | hi, child]
        lo = np.ascontiguousarray(planes[:, :, 0, :].transpose(0, 2, 1))
        hi = np.ascontiguousarray(planes[:, :, 1, :].transpose(0, 2, 1))
        ints = n.view(np.int32)
        return lo, hi, ints[:, 24:28].copy(), ints[:, 28].copy()

    def trace(self, rays, tmin=0.001, tmax=np.inf, sched=False):
        """Closest hits of a ray batch; sched=True: through the render kernel's own traversal."""
        rays = np.ascontiguousarray(rays, abi.RAY_DT)
        hits = np.zeros(len(rays), abi.HIT_DT)
        fn = lib().rt_trace_batch_sched if sched else lib().rt_trace_batch
        check(fn(self.handle, rays.ctypes.data, len(rays), tmin, tmax, hits.ctypes.data))
        return hits

    def closest_hits(self, rays, tmin=0.001, tmax=np.inf):
        """rt_closest_hits: (prim[n] int32, -1 for a miss; t[n] float32) of a ray batch."""
        rays = np.ascontiguousarray(rays, abi.RAY_DT)
        prim = np.zeros(len(rays), np.int32)
        t = np.zeros(len(rays), np.float32)
        check(lib().rt_closest_hits(self.handle, rays.ctypes.data, len(rays), tmin, tmax, prim.ctypes.data, t.ctypes.data))
        return prim, t

    def closest_hits_device(self, d_rays_ptr, n, d_prim_ptr, d_t_ptr, stream=0, tmin=0.001, tmax=np.inf):
        check(lib().rt_closest_hits_device(self.handle, C.c_void_p(d_rays_ptr), n, tmin, tmax, C.c_void_p(d_prim_ptr),
                                           C.c_void_p(d_t_ptr), C.c_void_p(stream)))

    def trace_device(self, d_rays_ptr, n, d_hits_ptr, stream=0, tmin=0.001, tmax=np.inf, sched=False):
        """The same on device buffers (n rt_ray records in, n rt_hit records out), enqueued on `stream`."""
        fn = lib().rt_trace_batch_sched_device if sched else lib().rt_trace_batch_device
        check(fn(self.handle, C.c_void_p(d_rays_ptr), n, tmin, tmax, C.c_void_p(d_hits_ptr), C.c_void_p(stream)))

    def render(self, cam, sample_count, sample_begin=0, seed=0, counters=False, accum=None,
               mode=abi.RT_MODE_MEGAKERNEL):
        """Host-buffer render: returns (accum[H,W,3] fp32 sums, RenderStats)."""
        H, W = cam.image_height(), cam.image_width
        if accum is None:
            accum = np.zeros((H, W, 3), np.float32)
        o = abi.RenderOpts(sample_begin, sample_count, seed, mode,
                           abi.RT_RENDER_COUNTERS if counters else 0)
        st = abi.RenderStats()
        check(lib().rt_render(self.handle, C.byref(cam), C.byref(o), accum.ctypes.data, C.byref(st)))
        return accum, st

    def render_device(self, cam, sample_count, d_accum_ptr, stream=0, sample_begin=0, seed=0,
                      counters=False, want_stats=False, mode=abi.RT_MODE_MEGAKERNEL, overlap=True):
        """Device-buffer render: d_accum_ptr is a device address (e.g. tensor.data_ptr()).
        overlap=False keeps the launches on `stream` alone (RT_RENDER_NO_OVERLAP)."""
        o = abi.RenderOpts(sample_begin, sample_count, seed, mode,
                           (abi.RT_RENDER_COUNTERS if counters else 0) |
                           (0 if overlap else abi.RT_RENDER_NO_OVERLAP))
        st = abi.RenderStats() if (want_stats or counters) else None
        check(lib().rt_render_device(self.handle, C.byref(cam), C.byref(o), C.c_void_p(d_accum_ptr),
                                     C.c_void_p(stream), C.byref(st) if st is not None else None))
        return st

    def render_device_static(self, cam, sample_count, d_accum_ptr, stream=0, sample_begin=0, seed=0):
        """rt_render_device_static: the same render without any device allocation (scratch in the
        library's static block; image size limited to RT_STATIC_MAX_PIXELS)."""
        o = abi.RenderOpts(sample_begin, sample_count, seed, abi.RT_MODE_MEGAKERNEL, 0)
        check(lib().rt_render_device_static(self.handle, C.byref(cam), C.byref(o), C.c_void_p(d_accum_ptr),
                                            C.c_void_p(stream)))

    def profile_begin(self):
        check(lib().rt_profile_begin(self.handle))

    def profile_end(self):
        ms, n = C.c_double(), C.c_int()
        check(lib().rt_profile_end(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class MultiScene:
    """Replicas of a scene on several GPUs of one node (rt_multi_create .. rt_multi_destroy)."""

    def __init__(self, arrays: abi.SceneArrays, devices, flags=0):
        self.arrays = arrays
        d = arrays.desc()
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        check(lib().rt_multi_create(C.byref(d), devs, len(devices), flags, C.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            lib().rt_multi_destroy(self.handle)
            self.handle = None

    __del__ = close

    def render(self, cam, sample_count, sample_begin=0, seed=0, accum=None, want_stats=True):
        """Samples [sample_begin, +sample_count) split over the devices; returns (accum, stats)."""
        H, W = cam.image_height(), cam.image_width
        if accum is None:
            accum = np.zeros((H, W, 3), np.float32)
        o = abi.RenderOpts(sample_begin, sample_count, seed, abi.RT_MODE_MEGAKERNEL, 0)
        st = abi.RenderStats() if want_stats else None
        check(lib().rt_multi_render(self.handle, C.byref(cam), C.byref(o), accum.ctypes.data,
                                    C.byref(st) if st is not None else None))
        return accum, st


class Film:
    """Device-resident accumulation image rendered in passes (rt_film_create .. rt_film_destroy)."""

    def __init__(self, width, height, device=0):
        h = C.c_void_p()
        check(lib().rt_film_create(device, width, height, C.byref(h)))
        self.handle, self.width, self.height = h, width, height

    def close(self):
        if self.handle:
            lib().rt_film_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def render(self, scene, cam, sample_count, sample_begin, seed=0):
        o = abi.RenderOpts(sample_begin, sample_count, seed, abi.RT_MODE_MEGAKERNEL, 0)
        check(lib().rt_film_render(scene.handle, self.handle, C.byref(cam), C.byref(o)))

    def snapshot(self, with_accum=False):
        check(lib().rt_film_snapshot(self.handle, 1 if with_accum else 0))

    def wait(self, want_accum=False):
        """Returns (rgb8[H,W,3], accum[H,W,3] or None, samples) of the last snapshot."""
        rgb = np.zeros((self.height, self.width, 3), np.uint8)
        acc = np.zeros((self.height, self.width, 3), np.float32) if want_accum else None
        n = C.c_int64()
        check(lib().rt_film_wait(self.handle, rgb.ctypes.data, acc.ctypes.data if want_accum else None, C.byref(n)))
        return rgb, acc, n.value

    def upload(self, accum, samples):
        if accum is None:
            check(lib().rt_film_upload(self.handle, None, 0))
            return
        accum = np.ascontiguousarray(accum, np.float32)
        if accum.shape != (self.height, self.width, 3):
            raise ValueError("accum shape does not match the film")
        check(lib().rt_film_upload(self.handle, accum.ctypes.data, samples))
