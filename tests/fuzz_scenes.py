"""Random scene descriptions in the reference's SDL (sdl_grammar.rs), for fuzzing the HIP path against the oracle.

Every construct of the grammar can appear: the planar shapes, the solids, CSG trees of solids with
transforms at every level, meshes with transforms, homogenous media, all five materials and all three
textures.  `scene_text(seed)` is deterministic in its seed."""
import numpy as np


def _f(x):
    return repr(round(float(x), 4))


def _vec(v):
    return "<" + ", ".join(_f(x) for x in v) + ">"


def _transform(rng, spread, linear=True):
    ops = []
    if linear and rng.random() < 0.5:
        ops.append("scale " + _vec(rng.uniform(0.6, 1.6, 3)))
    for ax in "xyz":
        if linear and rng.random() < 0.4:
            ops.append("rotate_%s %s" % (ax, _f(rng.uniform(-180, 180))))
    ops.append("translate " + _vec(rng.uniform(-spread, spread, 3) + np.array([0.0, spread, 0.0])))
    return "transform { " + " ".join(ops) + " }"


def _texture(rng):
    k = rng.integers(0, 4)
    if k == 0:
        return "texture { pattern { checkerboard color rgb %s color rgb %s %s } }" % (
            _vec(rng.uniform(0, 1, 3)), _vec(rng.uniform(0, 1, 3)), _f(rng.uniform(0.5, 6)))
    if k == 1:
        return 'texture { image "earth.ppm" %s }' % _f(rng.uniform(0.2, 2))
    return "texture { solid color rgb %s }" % _vec(rng.uniform(0.1, 1, 3))


def _material(rng, medium=False):
    if medium:
        return "material { isotropic %s }" % _texture(rng)
    k = rng.integers(0, 8)
    if k <= 2:
        return "material { lambertian %s }" % _texture(rng)
    if k <= 4:
        return "material { metal fuzz %s %s }" % (_f(rng.choice([0.0, 0.05, 0.3])), _texture(rng))
    if k == 5:
        return "material { dielectric ior %s }" % _f(rng.uniform(1.2, 1.8))
    if k == 6:
        return "material { dielectric ior %s fuzz %s }" % (_f(rng.uniform(1.2, 1.8)), _f(rng.uniform(0.01, 0.2)))
    return "material { diffuse_light intensity %s %s }" % (_f(rng.uniform(1, 6)), _texture(rng))


def _solid(rng, depth, spread, tx=True, torus=True):
    """one `solid_shape` (sdl_grammar.rs:148-156): sphere, cylinder, torus, cube or a CSG of two solids"""
    t = (" " + _transform(rng, spread)) if tx else ""
    k = rng.integers(0, 8 if depth > 0 else 5)
    if k == 0 or (k == 3 and not torus):
        return "sphere { origin %s radius %s%s }" % (_vec(rng.uniform(-0.3, 0.3, 3)), _f(rng.uniform(0.4, 1.0)), t)
    if k == 1:
        return "cylinder { radius %s height %s%s }" % (_f(rng.uniform(0.3, 0.8)), _f(rng.uniform(0.6, 1.6)), t)
    if k == 2 or k == 4:
        lo = rng.uniform(-0.9, -0.2, 3)
        return "cube { %s %s%s }" % (_vec(lo), _vec(lo + rng.uniform(0.5, 1.6, 3)), t)
    if k == 3:
        return "torus { radius %s radius %s%s }" % (_f(rng.uniform(0.6, 1.0)), _f(rng.uniform(0.12, 0.3)), t)
    op = ["union", "intersection", "difference"][k - 5]
    # (A union or difference returns one operand's intervals without its own to_world when the other operand is
    # missed, csg.rs:33-38,216-221: with a rotation or scale here the shape shows up in two places.  Both sides
    # reproduce that.)
    # children sit close to each other so that the operation has something to do
    a = _solid(rng, depth - 1, 0.35, tx=rng.random() < 0.7, torus=torus)
    b = _solid(rng, depth - 1, 0.35, tx=rng.random() < 0.7, torus=torus)
    return "%s { %s %s%s }" % (op, a, b, t)


def _shape(rng, spread, meshes, torus):
    k = rng.integers(0, 10)
    if k == 0:
        kind = ["xyrect", "xzrect", "zyrect"][rng.integers(0, 3)]
        return "%s { origin %s width %s height %s%s %s }" % (
            kind, _vec(rng.uniform(-0.3, 0.3, 3)), _f(rng.uniform(0.8, 2.5)), _f(rng.uniform(0.8, 2.5)),
            " reverse" if rng.random() < 0.3 else "", _transform(rng, spread)), False
    if k == 1 and meshes:
        name = meshes[rng.integers(0, len(meshes))]
        return 'mesh { file "%s" %s }' % (name, _transform(rng, spread)), False
    if k == 2:
        inner = _solid(rng, 1, spread, tx=False, torus=False)
        return "homogenous_medium { density %s %s %s }" % (_f(rng.uniform(0.3, 3)), inner, _transform(rng, spread)), True
    return _solid(rng, 2 if rng.random() < 0.8 else 3, spread, torus=torus), False


def scene_text(seed, n_objects=10, meshes=("two_parts.obj", "LinkedTorus.obj"), torus=True, spread=2.2):
    rng = np.random.default_rng(seed)
    out = []
    if rng.random() < 0.5:
        out.append("options { background color rgb %s }" % _vec(rng.uniform(0, 1, 3)))
    cam = np.array([rng.uniform(-1, 1), spread + rng.uniform(-0.5, 1.5), 3.2 * spread])
    out.append("camera { origin %s look_at %s fov %s }" % (_vec(cam), _vec([0, spread, 0]), _f(rng.uniform(45, 75))))
    out.append('object "light" { sphere { origin %s radius 3 } material { diffuse_light intensity 4 '
               'texture { solid color white } } }' % _vec([3 * spread, 4 * spread, 2 * spread]))
    if rng.random() < 0.8:
        nrm = np.array([0.0, 1.0, 0.0]) + (rng.normal(size=3) * 0.1 if rng.random() < 0.5 else 0.0)
        out.append('object "floor" { plane { origin <0, 0, 0> normal %s } %s }' % (_vec(nrm * rng.uniform(0.5, 2)),
                                                                                  _material(rng)))
    for i in range(n_objects):
        shape, medium = _shape(rng, spread, list(meshes), torus)
        out.append('object "o%d" { %s %s }' % (i, shape, _material(rng, medium)))
    return "\n".join(out) + "\n"
