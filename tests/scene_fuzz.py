"""Random scenes for parity fuzzing: every primitive kind, material, texture kind and wrapper
the path supports, in one scene (test infrastructure; shared by CPU and GPU tests)."""
import numpy as np

from raytracing_amd import abi


def random_scene(seed, n_prims=70):
    rng = np.random.default_rng(seed)
    # ---- textures: solids, noise (two tables), an image, checkers over any of them ----------
    tex = []

    def add_tex(**kw):
        t = np.zeros(1, abi.TEXTURE_DT)[0]
        t["even"] = t["odd"] = t["perlin"] = t["image"] = -1
        for k, v in kw.items():
            t[k] = v
        tex.append(t)
        return len(tex) - 1

    solids = [add_tex(type=abi.RT_TEX_SOLID, color=tuple(rng.uniform(0.05, 0.95, 3))) for _ in range(6)]
    noises = [add_tex(type=abi.RT_TEX_NOISE, perlin=k, scale=float(rng.uniform(0.5, 6.0))) for k in range(2)]
    image = add_tex(type=abi.RT_TEX_IMAGE, image=0)
    missing_image = add_tex(type=abi.RT_TEX_IMAGE, image=-1)  # texture.h:62: renders (0, 0, 1)
    checks = [add_tex(type=abi.RT_TEX_CHECKER, even=solids[0], odd=solids[1], scale=float(1 / rng.uniform(0.3, 3))),
              add_tex(type=abi.RT_TEX_CHECKER, even=noises[0], odd=image, scale=float(1 / rng.uniform(0.3, 3)))]
    checks.append(add_tex(type=abi.RT_TEX_CHECKER, even=checks[0], odd=solids[2], scale=float(1 / rng.uniform(0.3, 3))))
    all_tex = solids + noises + [image, missing_image] + checks
    perlins = np.zeros(2, abi.PERLIN_DT)
    for k in range(2):
        v = rng.uniform(-1, 1, (256, 3))
        perlins[k]["randvec"] = v / np.linalg.norm(v, axis=1, keepdims=True)
        perlins[k]["perm_x"] = rng.permutation(256)
    img_w, img_h = 16, 8
    texels = rng.integers(0, 256, img_w * img_h * 3).astype(np.uint8)
    images = np.array([(img_w, img_h, 0)], abi.IMAGE_DT)
    # ---- materials ------------------------------------------------------------------------------
    mats = []
    for _ in range(18):
        m = np.zeros(1, abi.MATERIAL_DT)[0]
        kind = rng.integers(0, 10)
        if kind < 5:
            m["type"], m["texture"] = abi.RT_MAT_LAMBERTIAN, all_tex[rng.integers(0, len(all_tex))]
        elif kind < 7:
            m["type"], m["albedo"], m["param"] = abi.RT_MAT_METAL, tuple(rng.uniform(0.4, 1, 3)), rng.uniform(0, 0.6)
        elif kind < 9:
            m["type"], m["param"] = abi.RT_MAT_DIELECTRIC, rng.choice([1.5, 1.33, 0.75, 2.4])
        else:
            m["type"], m["texture"] = abi.RT_MAT_DIFFUSE_LIGHT, all_tex[rng.integers(0, len(all_tex))]
        mats.append(m)
    # ---- primitives: spheres (some moving) and quads, most inside wrapper chains -----------------
    prims = np.zeros(n_prims, abi.PRIM_DT)
    ops, chains = [], []
    for i in range(n_prims):
        prims[i]["material"] = rng.integers(0, len(mats))
        c = rng.uniform(-1, 1, 3)
        if i % 3 == 2:
            prims[i]["type"] = abi.RT_PRIM_QUAD
            prims[i]["a"], prims[i]["b"], prims[i]["c"] = c, rng.uniform(-2, 2, 3), rng.uniform(-2, 2, 3)
        else:
            prims[i]["type"] = abi.RT_PRIM_SPHERE
            prims[i]["a"], prims[i]["c"] = c, (rng.uniform(0.3, 1.2), 0, 0)
            if i % 4 == 0:
                prims[i]["flags"] = abi.RT_PRIM_FLAG_MOVING
                prims[i]["b"] = rng.uniform(-0.4, 0.4, 3)
                prims[i]["d"] = prims[i]["a"] + prims[i]["b"]
        if i == 0:  # the ground: one big static sphere
            prims[i]["type"], prims[i]["flags"], prims[i]["xform"] = abi.RT_PRIM_SPHERE, 0, -1
            prims[i]["a"], prims[i]["b"], prims[i]["c"] = (0, -1010, 0), (0, 0, 0), (1000, 0, 0)
            continue
        if i % 5 == 4:
            prims[i]["xform"] = -1
            prims[i]["a"] += rng.uniform(-9, 9, 3).astype(np.float32)
            continue
        first = len(ops)
        kind = i % 4
        if kind == 0:
            ops.append((abi.RT_XF_ROTATE_Y, (rng.uniform(-180, 180), 0, 0)))
        elif kind == 1:
            ops.append((abi.RT_XF_ROTATE_XYZ, tuple(rng.uniform(-90, 90, 3))))
            ops.append((abi.RT_XF_SCALE, tuple(rng.uniform(0.5, 2.0, 3))))
        elif kind == 2:
            ops.append((abi.RT_XF_SCALE, (1.5, 1.5, 1.5)))
        else:
            ops.append((abi.RT_XF_ROTATE_XYZ, tuple(rng.uniform(-90, 90, 3))))
            ops.append((abi.RT_XF_ROTATE_Y, (rng.uniform(-180, 180), 0, 0)))
        ops.append((abi.RT_XF_TRANSLATE, tuple(rng.uniform(-9, 9, 3))))
        chains.append((first, len(ops) - first))
        prims[i]["xform"] = len(chains) - 1
    arrays = abi.SceneArrays(prims=prims, materials=np.array(mats, abi.MATERIAL_DT), textures=np.array(tex, abi.TEXTURE_DT),
                             perlins=perlins, images=images, texels=texels,
                             xform_ops=np.array(ops, abi.XFORM_OP_DT), chains=np.array(chains, abi.CHAIN_DT))
    cam = abi.CameraDesc(1.25, 80, 4, 12, (0.5, 0.6, 0.8), 55.0, (3, 6, 26), (0, 0, 0), (0, 1, 0),
                         float(rng.choice([0.0, 1.2])), 24.0)
    return arrays, cam


def random_rays(seed, n):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, abi.RAY_DT)
    o = rng.normal(size=(n, 3))
    o *= rng.uniform(15, 35, size=(n, 1)) / np.linalg.norm(o, axis=1, keepdims=True)
    o[:, 1] = np.abs(o[:, 1])
    rays["orig"], rays["dir"], rays["tm"] = o, rng.uniform(-10, 10, (n, 3)) - o, rng.uniform(0, 1, n)
    return rays
