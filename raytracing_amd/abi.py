"""ctypes mirror of include/rt_amd.h (plain structs and constants, no behaviour).

Used by the ctypes binding of librt_amd.so (raytracing_amd/binding.py) and by the
tests, which also hand the same structs to the CPU oracle.
"""
import ctypes as C

import numpy as np

RT_AMD_ABI_VERSION = 4

RT_OK, RT_ERR_ARG, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_IO = 0, -1, -2, -3, -4
RT_PRIM_SPHERE, RT_PRIM_QUAD = 0, 1
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT = 0, 1, 2, 3
RT_TEX_SOLID, RT_TEX_CHECKER, RT_TEX_IMAGE, RT_TEX_NOISE = 0, 1, 2, 3
RT_XF_TRANSLATE, RT_XF_ROTATE_Y, RT_XF_ROTATE_XYZ, RT_XF_SCALE = 0, 1, 2, 3
RT_GROUP_LIST, RT_GROUP_BVH = 0, 1
RT_PRIM_FLAG_MOVING = 1
RT_MODE_MEGAKERNEL, RT_MODE_WAVEFRONT = 0, 1
RT_RENDER_COUNTERS = 1
RT_RENDER_NO_OVERLAP = 2
RT_STATIC_MAX_PIXELS = 4 << 20  # rt_render_device_static: largest image (pixels)
RT_SCENE_FORCE_BVH, RT_SCENE_FORCE_LINEAR, RT_SCENE_FORCE_GLOBAL, RT_SCENE_BUILD_DEVICE = 1, 2, 4, 8

f3 = C.c_float * 3

# numpy dtypes of the array records (all naturally aligned, no padding surprises)
PRIM_DT = np.dtype([("type", "<i4"), ("material", "<i4"), ("xform", "<i4"), ("flags", "<i4"),
                    ("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3), ("d", "<f4", 3)])
MATERIAL_DT = np.dtype([("type", "<i4"), ("texture", "<i4"), ("albedo", "<f4", 3), ("param", "<f4")])
TEXTURE_DT = np.dtype([("type", "<i4"), ("even", "<i4"), ("odd", "<i4"), ("perlin", "<i4"),
                       ("image", "<i4"), ("color", "<f4", 3), ("scale", "<f4")])
PERLIN_DT = np.dtype([("randvec", "<f4", (256, 3)), ("perm_x", "<i4", 256)])
IMAGE_DT = np.dtype([("width", "<i4"), ("height", "<i4"), ("offset", "<u8")])
XFORM_OP_DT = np.dtype([("type", "<i4"), ("p", "<f4", 3)])
CHAIN_DT = np.dtype([("first_op", "<i4"), ("n_ops", "<i4")])
GROUP_DT = np.dtype([("kind", "<i4"), ("first_prim", "<i4"), ("n_prims", "<i4"), ("reserved", "<i4")])
INSTANCE_DT = np.dtype([("first_prim", "<i4"), ("n_prims", "<i4"), ("xform", "<i4"), ("reserved", "<i4")])
RAY_DT = np.dtype([("orig", "<f4", 3), ("dir", "<f4", 3), ("tm", "<f4")])
HIT_DT = np.dtype([("prim", "<i4"), ("front_face", "<i4"), ("t", "<f4"), ("p", "<f4", 3),
                   ("normal", "<f4", 3), ("u", "<f4"), ("v", "<f4")])

assert PRIM_DT.itemsize == 64 and MATERIAL_DT.itemsize == 24 and TEXTURE_DT.itemsize == 36
assert PERLIN_DT.itemsize == 4096 and IMAGE_DT.itemsize == 16 and RAY_DT.itemsize == 28
assert HIT_DT.itemsize == 44


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_prims", C.c_int32), ("prims", C.c_void_p),
                ("n_groups", C.c_int32), ("groups", C.c_void_p),
                ("n_materials", C.c_int32), ("materials", C.c_void_p),
                ("n_textures", C.c_int32), ("textures", C.c_void_p),
                ("n_perlins", C.c_int32), ("perlins", C.c_void_p),
                ("n_images", C.c_int32), ("images", C.c_void_p),
                ("n_texel_bytes", C.c_uint64), ("texels", C.c_void_p),
                ("n_xform_ops", C.c_int32), ("xform_ops", C.c_void_p),
                ("n_chains", C.c_int32), ("chains", C.c_void_p),
                ("n_instances", C.c_int32), ("instances", C.c_void_p)]


class CameraDesc(C.Structure):
    _fields_ = [("aspect_ratio", C.c_float), ("image_width", C.c_int32),
                ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
                ("background", f3), ("vfov", C.c_float),
                ("lookfrom", f3), ("lookat", f3), ("vup", f3),
                ("defocus_angle", C.c_float), ("focus_dist", C.c_float)]

    def copy(self):
        out = CameraDesc()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(CameraDesc))
        return out

    def image_height(self):
        """camera.h:77-78, in fp32 like the reference."""
        h = int(np.float32(self.image_width) / np.float32(self.aspect_ratio))
        return max(1, h)


class CameraFrame(C.Structure):
    _fields_ = [("image_width", C.c_int32), ("image_height", C.c_int32),
                ("pixel_samples_scale", C.c_float),
                ("center", f3), ("pixel00_loc", f3), ("pixel_delta_u", f3), ("pixel_delta_v", f3),
                ("u", f3), ("v", f3), ("w", f3), ("defocus_disk_u", f3), ("defocus_disk_v", f3)]


class RenderOpts(C.Structure):
    _fields_ = [("sample_begin", C.c_int32), ("sample_count", C.c_int32), ("seed", C.c_uint32),
                ("mode", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32 * 3)]


class RenderStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("box_tests", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("sphere_hits", C.c_uint64),
                ("quad_tests", C.c_uint64), ("shades", C.c_uint64 * 4),
                ("noise_evals", C.c_uint64), ("image_evals", C.c_uint64),
                ("checker_evals", C.c_uint64), ("sched", C.c_uint64 * 6), ("kernel_ms", C.c_double),
                ("cycles", C.c_uint64 * 6), ("tail_us", C.c_double * 3), ("coherence", C.c_uint64 * 2)]


class SceneInfo(C.Structure):
    _fields_ = [("n_prims", C.c_int32), ("n_spheres", C.c_int32), ("n_quads", C.c_int32),
                ("n_nodes", C.c_int32), ("bvh_depth", C.c_int32), ("lds_resident", C.c_int32),
                ("device_bytes", C.c_uint64), ("build_ms", C.c_float), ("built_on_device", C.c_int32),
                ("n_upfront", C.c_int32)]


class SceneArrays:
    """A scene as numpy record arrays; `desc()` yields an rt_scene_desc over them.

    The arrays are kept alive by this object, so keep it alive while the
    descriptor is in use.
    """

    FIELDS = [("prims", PRIM_DT), ("groups", GROUP_DT), ("materials", MATERIAL_DT),
              ("textures", TEXTURE_DT), ("perlins", PERLIN_DT), ("images", IMAGE_DT),
              ("texels", np.dtype("u1")), ("xform_ops", XFORM_OP_DT), ("chains", CHAIN_DT),
              ("instances", INSTANCE_DT)]

    def __init__(self, **arrays):
        for name, dt in self.FIELDS:
            a = arrays.get(name)
            a = np.zeros(0, dt) if a is None else np.ascontiguousarray(a, dtype=dt)
            setattr(self, name, a)
        if len(self.groups) == 0 and len(self.prims) > 0:
            self.groups = np.array([(RT_GROUP_LIST, 0, len(self.prims), 0)], GROUP_DT)

    def desc(self):
        d = SceneDesc()
        d.abi_version = RT_AMD_ABI_VERSION

        def ptr(a):
            return a.ctypes.data if a.size else None

        d.n_prims, d.prims = len(self.prims), ptr(self.prims)
        d.n_groups, d.groups = len(self.groups), ptr(self.groups)
        d.n_materials, d.materials = len(self.materials), ptr(self.materials)
        d.n_textures, d.textures = len(self.textures), ptr(self.textures)
        d.n_perlins, d.perlins = len(self.perlins), ptr(self.perlins)
        d.n_images, d.images = len(self.images), ptr(self.images)
        d.n_texel_bytes, d.texels = self.texels.size, ptr(self.texels)
        d.n_xform_ops, d.xform_ops = len(self.xform_ops), ptr(self.xform_ops)
        d.n_chains, d.chains = len(self.chains), ptr(self.chains)
        d.n_instances, d.instances = len(self.instances), ptr(self.instances)
        return d

    @classmethod
    def load(cls, path):
        """Reads a .rtscene file (layout: include/rt_scene_io.h)."""
        with open(path, "rb") as f:
            blob = f.read()
        if blob[:8] != b"RTSCENE1":
            raise ValueError(f"{path}: not a .rtscene file")
        counts = np.frombuffer(blob, "<u8", 10, 8)
        off = 8 + 80
        arrays = {}
        for (name, dt), n in zip(cls.FIELDS, counts[:10]):
            nbytes = int(n) * dt.itemsize
            arrays[name] = np.frombuffer(blob, dt, int(n), off).copy()
            off += (nbytes + 7) & ~7
        return cls(**arrays)

    def save(self, path):
        counts = np.zeros(10, "<u8")
        blobs = []
        for i, (name, dt) in enumerate(self.FIELDS):
            a = getattr(self, name)
            counts[i] = len(a)
            b = a.tobytes()
            blobs.append(b + b"\0" * (-len(b) % 8))
        with open(path, "wb") as f:
            f.write(b"RTSCENE1" + counts.tobytes() + b"".join(blobs))


def load_camera_file(path):
    """Reads a .cam fixture: rt_camera_desc followed by rt_camera_frame."""
    with open(path, "rb") as f:
        blob = f.read()
    cam = CameraDesc.from_buffer_copy(blob[:C.sizeof(CameraDesc)])
    frame = CameraFrame.from_buffer_copy(blob[C.sizeof(CameraDesc):])
    return cam, frame
