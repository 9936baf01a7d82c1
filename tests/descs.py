"""Build small rt_scene_desc values from Python for unit-level tests."""
import ctypes as C

import numpy as np

from raytracer_amd import _capi

IDENT = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]
KINDS = {"sphere": _capi.RT_SHAPE_SPHERE, "plane": _capi.RT_SHAPE_PLANE, "xyrect": _capi.RT_SHAPE_XYRECT,
         "xzrect": _capi.RT_SHAPE_XZRECT, "zyrect": _capi.RT_SHAPE_ZYRECT, "cube": _capi.RT_SHAPE_CUBE,
         "cylinder": _capi.RT_SHAPE_CYLINDER, "torus": _capi.RT_SHAPE_TORUS}


class Desc:
    """Keeps the ctypes arrays alive next to the rt_scene_desc that points at them."""

    def __init__(self, shapes):
        """shapes: list of {"kind", "params", optional "flags", optional "o2w" (16 doubles, row-major,
        row-vector convention; its inverse becomes w2o)}; one white lambertian object per shape."""
        n = len(shapes)
        self.nodes = (_capi.rt_shape_node * n)()
        self.objects = (_capi.rt_object * n)()
        for i, s in enumerate(shapes):
            nd = self.nodes[i]
            nd.kind = KINDS[s["kind"]]
            nd.child_a = nd.child_b = -1
            nd.flags = s.get("flags", 0)
            for k, v in enumerate(s["params"]):
                nd.params[k] = float(v)
            o2w = s.get("o2w", IDENT)
            w2o = IDENT
            if "o2w" in s:
                inv = np.linalg.inv(np.array(o2w, dtype=float).reshape(4, 4))
                inv[:, 3] = [0.0, 0.0, 0.0, 1.0]          # affine: the last column is exact
                w2o = inv.reshape(16).tolist()
            for k in range(16):
                nd.o2w[k] = float(o2w[k])
                nd.w2o[k] = float(w2o[k])
            self.objects[i].shape = i
            self.objects[i].material = 0
        self.materials = (_capi.rt_material * 1)()
        self.materials[0].kind = _capi.RT_MAT_LAMBERTIAN
        self.materials[0].texture = 0
        self.textures = (_capi.rt_texture * 1)()
        self.textures[0].kind = _capi.RT_TEX_SOLID
        self.textures[0].image = -1
        for k in range(3):
            self.textures[0].c1[k] = 1.0
        d = _capi.rt_scene_desc()
        d.camera.fov_factor = 0.5773502691896257
        for k in range(16):
            d.camera.camera_to_world[k] = IDENT[k]
        d.n_objects = n
        d.objects = C.cast(self.objects, C.POINTER(_capi.rt_object))
        d.n_nodes = n
        d.nodes = C.cast(self.nodes, C.POINTER(_capi.rt_shape_node))
        d.n_materials = 1
        d.materials = C.cast(self.materials, C.POINTER(_capi.rt_material))
        d.n_textures = 1
        d.textures = C.cast(self.textures, C.POINTER(_capi.rt_texture))
        self.desc = d

    @property
    def ptr(self):
        return C.pointer(self.desc)


class FakeScene:
    """Duck-types raytracer_amd.sdl.Scene for DeviceScene()."""

    def __init__(self, d):
        self._d = d
        self.desc = d.ptr
