"""Independent checks of the parts of the path that the reference's own unit tests do not pin
(camera, trace/cast, to_world under transforms, CSG, cylinder, mesh, fog, materials, textures).

Nothing here restates the reference's algorithms: every expectation is computed from geometry or physics in
plain numpy (closed-form quadrics, point-in-solid classification, a brute-force triangle loop, Snell and
Fresnel, Beer-Lambert, the furnace identity) and compared with what the scene parser + flattening conventions +
CPU oracle produce for the same SDL text.  Because the expectation never sees rt_scene_desc, a wrong rotation
sign, transform order or matrix inverse in the host plumbing shows up here (VERDICT r1 "common-mode plumbing").
The GPU path is compared with the oracle elsewhere (tests/test_gpu_parity.py); its test_analytic_cases_on_device
runs the ray-batch cases of this file through rt_trace_rays on the device as well."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
import parity
from raytracer_amd import _capi

MAT = ' material { lambertian texture { solid color white } } }\n'
CAM = "camera { origin <0, 0, 9> look_at <0, 0, 0> }\n"


# ---------------------------------------------------------------- conventions, stated independently
# Row vectors: p' = p @ M; a transform list composes left to right (matrix.rs:29-75, sdl.rs:94-96).
def T(x, y, z):
    m = np.eye(4)
    m[3, :3] = [x, y, z]
    return m


def S(x, y, z):
    return np.diag([x, y, z, 1.0])


def RX(deg):
    a = np.radians(deg)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    return m


def RY(deg):
    a = np.radians(deg)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    return m


def RZ(deg):
    a = np.radians(deg)
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
    return m


def chain(*ms):
    out = np.eye(4)
    for m in ms:
        out = out @ m
    return out


def to_object(M, o, d):
    """world ray -> object ray with the SAME parameter t (direction left un-normalised)"""
    Mi = np.linalg.inv(M)
    oo = np.c_[o, np.ones(len(o))] @ Mi
    return oo[:, :3], d @ Mi[:3, :3], Mi


def normal_to_world(Mi, n):
    nw = n @ Mi[:3, :3].T
    return nw / np.linalg.norm(nw, axis=1, keepdims=True)


def rays_towards(n, seed, center, r_from, r_to):
    rng = np.random.default_rng(seed)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = np.asarray(center, float) + u * r_from
    tgt = np.asarray(center, float) + rng.uniform(-r_to, r_to, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


ON_DEVICE = [False]          # test_gpu_parity.test_analytic_cases_on_device flips this


def trace(rt, text, o, d, base=None):
    opts = rt.Options(width=16, height=16)
    sc = rt.parse(opts, text, base or parity.ASSETS)
    if ON_DEVICE[0]:
        hits, _ = rt.DeviceScene(sc).trace_rays(o, d, seed=4)
    else:
        hits, _ = oracle_lib.trace_rays(sc.desc, o, d, seed=4)
    return hits, sc


# ---------------------------------------------------------------- quadrics under a full transform list
def test_transformed_sphere_closed_form(rt):
    """sphere.rs + Ray::to_object + Intersection::to_world (system.rs:106-110,190-199) + the parser's transform
    folding (sdl.rs:44-47,94-96) + the matrix inverse, against the quadratic solved in numpy."""
    c, r = np.array([0.3, -0.2, 0.1]), 0.8
    text = CAM + ('object "s" { sphere { origin <0.3, -0.2, 0.1> radius 0.8 transform { scale <1.5, 0.7, 1.2> rotate_x 25 '
                  'rotate_y -40 rotate_z 70 translate <0.5, 1, -2> } }' + MAT)
    M = chain(S(1.5, 0.7, 1.2), RX(25), RY(-40), RZ(70), T(0.5, 1, -2))
    o, d = rays_towards(20000, 1, [0.5, 1, -2], 6.0, 1.3)
    hits, _ = trace(rt, text, o, d)
    oo, dd, Mi = to_object(M, o, d)
    l = oo - c
    a, b, cc = (dd * dd).sum(1), 2 * (dd * l).sum(1), (l * l).sum(1) - r * r
    disc = b * b - 4 * a * cc
    clear = np.abs(disc) > 1e-6 * b * b                       # grazing rays: either answer is fine
    hit = disc > 0
    t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), -1.0)
    assert hit.mean() > 0.3
    assert np.array_equal((hits["object"] == 0)[clear], hit[clear])
    ok = hit & clear
    assert np.max(np.abs(hits["t"][ok] - t[ok])) < 1e-9
    n = normal_to_world(Mi, (oo + dd * t[:, None] - c)[ok])
    assert np.max(np.abs(hits["n"][ok] - n)) < 1e-9
    # the sphere's uv comes from the object-space normal (sphere.rs:27-38)
    no = (oo + dd * t[:, None] - c)[ok] / r
    u = (1 - np.arctan2(no[:, 2], no[:, 0]) / np.pi) / 2
    v = np.arccos(np.clip(no[:, 1], -1, 1)) / np.pi
    assert np.max(np.abs(hits["uv"][ok] - np.c_[u, v])) < 1e-7


def _cylinder_numpy(oo, dd, r, h):
    """nearest t >= 0 on the closed cylinder x^2 + z^2 = r^2, |y| <= h/2, and its object-space normal"""
    n_rays = len(oo)
    best_t = np.full(n_rays, np.inf)
    best_n = np.zeros((n_rays, 3))
    a = dd[:, 0] ** 2 + dd[:, 2] ** 2
    b = 2 * (oo[:, 0] * dd[:, 0] + oo[:, 2] * dd[:, 2])
    c = oo[:, 0] ** 2 + oo[:, 2] ** 2 - r * r
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore", divide="ignore"):
        for sgn in (-1.0, 1.0):
            t = (-b + sgn * np.sqrt(np.maximum(disc, 0))) / (2 * a)
            y = oo[:, 1] + dd[:, 1] * t
            good = (disc > 0) & (t >= 0) & (np.abs(y) <= h / 2) & (t < best_t)
            p = oo + dd * t[:, None]
            best_t = np.where(good, t, best_t)
            best_n = np.where(good[:, None], np.c_[p[:, 0], np.zeros(n_rays), p[:, 2]] / r, best_n)
        for cap in (-h / 2, h / 2):
            t = (cap - oo[:, 1]) / dd[:, 1]
            p = oo + dd * t[:, None]
            good = (t >= 0) & (p[:, 0] ** 2 + p[:, 2] ** 2 <= r * r) & (t < best_t)
            best_t = np.where(good, t, best_t)
            best_n = np.where(good[:, None], np.array([0.0, np.sign(cap), 0.0]), best_n)
    return best_t, best_n, disc


def test_transformed_cylinder_closed_form(rt):
    """cylinder.rs:53-144 (quadratic in xz, nine-way case split on the caps) against the closed cylinder in numpy."""
    r, h = 0.7, 1.6
    text = CAM + ('object "c" { cylinder { radius 0.7 height 1.6 transform { rotate_z 30 scale <1.2, 0.8, 1> rotate_x -55 '
                  'translate <-0.4, 0.3, 0.6> } }' + MAT)
    M = chain(RZ(30), S(1.2, 0.8, 1), RX(-55), T(-0.4, 0.3, 0.6))
    o, d = rays_towards(30000, 2, [-0.4, 0.3, 0.6], 6.0, 1.2)
    hits, _ = trace(rt, text, o, d)
    oo, dd, Mi = to_object(M, o, d)
    t, n, disc = _cylinder_numpy(oo, dd, r, h)
    hit = np.isfinite(t)
    # rays that graze the side or pass within 1e-6 of a cap rim may go either way
    p = oo + dd * np.where(hit, t, 0.0)[:, None]
    rim = np.abs(np.hypot(p[:, 0], p[:, 2]) - r) < 1e-6
    edge = hit & rim & (np.abs(np.abs(p[:, 1]) - h / 2) < 1e-6)
    clear = (np.abs(disc) > 1e-9) & ~edge
    got_hit = hits["object"] == 0
    assert hit.mean() > 0.3
    assert np.mean(got_hit[clear] == hit[clear]) > 0.9999
    ok = hit & got_hit & clear
    assert np.mean(np.abs(hits["t"][ok] - t[ok]) < 1e-9) > 0.9999
    nw = normal_to_world(Mi, n[ok])
    assert np.mean(np.all(np.abs(hits["n"][ok] - nw) < 1e-8, axis=1)) > 0.9995       # rim pixels take either normal


def test_transformed_cube_and_rects(rt):
    """cube.rs (six rectangle tests, exit normal negated) and the three axis rectangles (plane.rs:111-244) under
    rotations: nearest face of the box / the rectangle by the slab method in numpy."""
    lo, hi = np.array([-0.6, -0.3, -0.9]), np.array([0.4, 0.8, 0.5])
    text = CAM + ('object "b" { cube { <-0.6, -0.3, -0.9> <0.4, 0.8, 0.5> transform { rotate_y 33 rotate_x 12 translate <1, 0, 0> } }' + MAT +
                  'object "r" { xzrect { origin <0, 0, 0> width 2 height 1 transform { rotate_z 20 translate <-2, 0.5, 0> } }' + MAT)
    Mb = chain(RY(33), RX(12), T(1, 0, 0))
    Mr = chain(RZ(20), T(-2, 0.5, 0))
    o, d = rays_towards(30000, 3, [-0.3, 0.3, 0], 7.0, 2.2)
    hits, _ = trace(rt, text, o, d)
    # box: slabs
    oo, dd, Mi = to_object(Mb, o, d)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (lo - oo) / dd, (hi - oo) / dd
    tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
    t_in, t_out = tn.max(1), tf.min(1)
    box_hit = (t_in <= t_out) & (t_in >= 0)                    # origins are outside: the entry face
    axis = tn.argmax(1)
    nb = np.zeros((len(o), 3))
    nb[np.arange(len(o)), axis] = -np.sign(dd[np.arange(len(o)), axis])
    t_box = np.where(box_hit, t_in, np.inf)
    # rectangle in the plane y = 0 of its object space, |x| <= 1, |z| <= 0.5
    ro, rd, Ri = to_object(Mr, o, d)
    with np.errstate(divide="ignore", invalid="ignore"):
        tr = -ro[:, 1] / rd[:, 1]
    pr = ro + rd * tr[:, None]
    rect_hit = (tr >= 0) & (np.abs(pr[:, 0]) <= 1) & (np.abs(pr[:, 2]) <= 0.5)
    t_rect = np.where(rect_hit, tr, np.inf)
    want_obj = np.where(np.minimum(t_box, t_rect) == np.inf, -1, np.where(t_box <= t_rect, 0, 1))
    want_t = np.minimum(t_box, t_rect)
    near_edge = np.zeros(len(o), bool)
    pb = oo + dd * np.where(box_hit, t_in, 0)[:, None]
    for k in range(3):
        near_edge |= (np.abs(tn[:, k] - t_in) < 1e-7) & (k != axis)          # box edges
        near_edge |= np.abs(t_in - t_out) < 1e-7
    near_edge |= (np.abs(np.abs(pr[:, 0]) - 1) < 1e-7) | (np.abs(np.abs(pr[:, 2]) - 0.5) < 1e-7)
    clear = ~near_edge
    assert (want_obj == 0).mean() > 0.1 and (want_obj == 1).mean() > 0.03
    assert np.mean(hits["object"][clear] == want_obj[clear]) > 0.9999
    ok = clear & (want_obj >= 0) & (hits["object"] == want_obj)
    assert np.max(np.abs(hits["t"][ok] - want_t[ok])) < 1e-9
    b = ok & (want_obj == 0)
    assert np.max(np.abs(hits["n"][b] - normal_to_world(Mi, nb[b]))) < 1e-9
    r = ok & (want_obj == 1)
    nr = normal_to_world(Ri, np.tile([0.0, 1.0, 0.0], (r.sum(), 1)))
    nr *= -np.sign((nr * d[r]).sum(1))[:, None]                               # planes face the ray (plane.rs:67-70)
    assert np.max(np.abs(hits["n"][r] - nr)) < 1e-9


# ---------------------------------------------------------------- CSG by point classification
def _inside_sphere(p, c, r):
    return ((p - c) ** 2).sum(-1) < r * r


def _inside_cube(p, lo, hi):
    return np.all((p > lo) & (p < hi), axis=-1)


def test_csg_first_hit_by_point_classification(rt):
    """csg.rs:28-280: union, intersection, difference and a nested difference of spheres and cubes.  Sample points
    along each ray are classified inside/outside the solid from the primitives' definitions; the first sample
    inside (refined by bisection) must be where the oracle reports its first hit, and the normal there must be
    the gradient of the primitive whose surface is met, facing the ray."""
    # four solids, well apart along x; node transforms are translations (no skipped to_world here: that quirk has
    # its own test), primitives carry translations and a scale
    text = CAM + (
        'object "u" { union { sphere { origin <0, 0, 0> radius 0.8 } cube { <0.3, -0.4, -0.4> <1.3, 0.4, 0.4> } '
        'transform { translate <-4.5, 0, 0> } }' + MAT +
        'object "i" { intersection { sphere { origin <0, 0, 0> radius 0.8 } cube { <0.3, -0.4, -0.4> <1.3, 0.4, 0.4> } '
        'transform { translate <-1.5, 0, 0> } }' + MAT +
        'object "d" { difference { cube { <-0.7, -0.7, -0.7> <0.7, 0.7, 0.7> } sphere { origin <0, 0, 0> radius 0.9 } '
        'transform { translate <1.5, 0, 0> } }' + MAT +
        'object "n" { difference { difference { intersection { cube { <-1, 0, -1> <1, 2, 1> } sphere { origin <0, 0, 0> radius 1 '
        'transform { scale <1.4, 1.4, 1.4> translate <0, 1, 0> } } } sphere { origin <0, 1, 0> radius 1.35 } } '
        'cube { <-0.2, -0.2, -2> <0.2, 0.2, 2> transform { translate <0, 1, 0> } } transform { translate <4.5, -1, 0> } }' + MAT)

    def inside(p):
        q = p - [-4.5, 0, 0]
        u = _inside_sphere(q, 0, 0.8) | _inside_cube(q, [0.3, -0.4, -0.4], [1.3, 0.4, 0.4])
        q = p - [-1.5, 0, 0]
        i = _inside_sphere(q, 0, 0.8) & _inside_cube(q, [0.3, -0.4, -0.4], [1.3, 0.4, 0.4])
        q = p - [1.5, 0, 0]
        dd = _inside_cube(q, [-0.7] * 3, [0.7] * 3) & ~_inside_sphere(q, 0, 0.9)
        q = p - [4.5, -1, 0]
        n = (_inside_cube(q, [-1, 0, -1], [1, 2, 1]) & _inside_sphere(q, [0, 1, 0], 1.4) & ~_inside_sphere(q, [0, 1, 0], 1.35)
             & ~_inside_cube(q, [-0.2, 0.8, -2], [0.2, 1.2, 2]))
        return np.stack([u, i, dd, n], -1)

    rng = np.random.default_rng(5)
    n_rays = 6000
    u = rng.normal(size=(n_rays, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = u * 9.0
    tgt = np.c_[rng.choice([-4.5, -1.5, 1.5, 4.5], n_rays) + rng.uniform(-1.2, 1.2, n_rays), rng.uniform(-1.2, 1.2, n_rays),
                rng.uniform(-1.2, 1.2, n_rays)]
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits, _ = trace(rt, text, o, d)

    ts = np.linspace(0.0, 18.0, 6001)
    dt = ts[1] - ts[0]
    want_t = np.full(n_rays, np.inf)
    want_obj = np.full(n_rays, -1)
    for k0 in range(0, n_rays, 500):
        sl = slice(k0, k0 + 500)
        p = o[sl, None, :] + d[sl, None, :] * ts[None, :, None]
        ins = inside(p)                                          # rays x samples x 4
        any_in = ins.any(-1)
        first = any_in.argmax(1)
        has = any_in.any(1)
        idx = np.arange(len(first))
        lo_t, hi_t = ts[np.maximum(first - 1, 0)], ts[first]
        obj = ins[idx, first].argmax(-1)
        for _ in range(40):                                      # bisection on the classification
            mid = 0.5 * (lo_t + hi_t)
            m_in = inside(o[sl] + d[sl] * mid[:, None])[idx, obj]
            hi_t = np.where(m_in, mid, hi_t)
            lo_t = np.where(m_in, lo_t, mid)
        want_t[sl] = np.where(has, hi_t, np.inf)
        want_obj[sl] = np.where(has, obj, -1)
    got_obj, got_t = hits["object"], np.where(hits["object"] >= 0, hits["t"], np.inf)
    for k in range(4):
        assert (want_obj == k).mean() > 0.02
    # a feature thinner than the sampling step can be stepped over: allow a few in a thousand
    both = (want_obj >= 0) & (got_obj >= 0)
    assert np.mean(got_obj == want_obj) > 0.995, np.mean(got_obj == want_obj)
    agree = both & (got_obj == want_obj)
    close = np.abs(got_t[agree] - want_t[agree]) < 1e-6
    assert close.mean() > 0.995, close.mean()
    # normals: unit length, facing the ray, and the gradient of a sphere (radial) or a cube (an axis) at the point
    n = hits["n"][agree][close]
    dd = d[agree][close]
    assert np.max(np.abs(np.linalg.norm(n, axis=1) - 1)) < 1e-12
    assert np.mean((n * dd).sum(1) < 0) > 0.999
    p = (o[agree] + d[agree] * got_t[agree][:, None])[close]
    centres = {0: [(-4.5, 0, 0)], 1: [(-1.5, 0, 0)], 2: [(1.5, 0, 0)], 3: [(4.5, 0, 0)]}
    ob = got_obj[agree][close]
    axis_aligned = np.isclose(np.abs(n).max(1), 1.0, atol=1e-9)
    radial = np.zeros(len(n), bool)
    for k, cs in centres.items():
        for c in cs:
            v = p - np.array(c)
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            radial |= (ob == k) & (np.abs(np.abs((v * n).sum(1)) - 1) < 1e-7)
    assert np.mean(axis_aligned | radial) > 0.999


# ---------------------------------------------------------------- mesh by brute force
def _load_obj(path):
    vs, ns, tris = [], [], []
    with open(path) as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] == "v":
                vs.append([float(x) for x in w[1:4]])
            elif w[0] == "vn":
                ns.append([float(x) for x in w[1:4]])
            elif w[0] == "f":
                idx = [(int(c.split("/")[0]) - 1, int(c.split("/")[2]) - 1) for c in w[1:]]
                for k in range(1, len(idx) - 1):                 # fan
                    tris.append([idx[0], idx[k], idx[k + 1]])
    tris = np.array(tris)
    return np.array(vs), np.array(ns), tris[:, :, 0], tris[:, :, 1]


def test_mesh_scene_against_brute_force_moller_trumbore(rt):
    """mesh.rs:54-113 + composite.rs + the OBJ path (sdl.rs:55-92, fan triangulation) + the mesh's translate, on
    the shipped mesh.scene: an OBJ reader and a Moeller-Trumbore loop written here (back faces culled,
    0 <= u, v, u + v <= 1, t >= 0, smooth normals), nearest of {light sphere, floor plane, mesh}."""
    text = open(parity.scene_path("mesh")).read()
    o, d = rays_towards(6000, 7, [0, 1, 0], 4.5, 1.4)
    o[:, 1] = np.abs(o[:, 1]) + 0.2                                # above the floor
    d = np.array([0, 1, 0]) + np.random.default_rng(8).uniform(-1.3, 1.3, (len(o), 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hits, sc = trace(rt, text, o, d)
    V, N, tv, tn = _load_obj(os.path.join(parity.ASSETS, "LinkedTorus.obj"))
    assert len(tv) == 2304 and sc.desc.contents.n_triangles == 2304
    V = V + [0, 1, 0]                                              # mesh { ... transform { translate <0, 1, 0> } }
    v0, e1, e2 = V[tv[:, 0]], V[tv[:, 1]] - V[tv[:, 0]], V[tv[:, 2]] - V[tv[:, 0]]
    t_mesh = np.full(len(o), np.inf)
    n_mesh = np.zeros((len(o), 3))
    for k0 in range(0, len(o), 250):
        sl = slice(k0, k0 + 250)
        D = d[sl, None, :]
        pvec = np.cross(D, e2[None])
        det = (e1[None] * pvec).sum(-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tvec = o[sl, None, :] - v0[None]
            u = (tvec * pvec).sum(-1) * inv
            qvec = np.cross(tvec, e1[None])
            v = (D * qvec).sum(-1) * inv
            t = (e2[None] * qvec).sum(-1) * inv
        ok = (det > 1e-12) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t >= 0)
        t = np.where(ok, t, np.inf)
        best = t.argmin(1)
        idx = np.arange(t.shape[0])
        t_mesh[sl] = t[idx, best]
        uu, vv = u[idx, best], v[idx, best]
        nn = (1 - uu - vv)[:, None] * N[tn[best, 0]] + uu[:, None] * N[tn[best, 1]] + vv[:, None] * N[tn[best, 2]]
        n_mesh[sl] = nn / np.linalg.norm(nn, axis=1, keepdims=True)
    # the other two objects
    with np.errstate(divide="ignore", invalid="ignore"):
        t_plane = -o[:, 1] / d[:, 1]
    t_plane = np.where(t_plane >= 0, t_plane, np.inf)
    l = o - 7.0
    b, c = 2 * (d * l).sum(1), (l * l).sum(1) - 25.0
    disc = b * b - 4 * c
    t_sph = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / 2, np.inf)
    t_sph = np.where(t_sph >= 0, t_sph, np.inf)
    all_t = np.c_[t_sph, t_plane, t_mesh]
    want_obj = np.where(np.isfinite(all_t.min(1)), all_t.argmin(1), -1)
    assert (want_obj == 2).mean() > 0.25
    assert np.mean(hits["object"] == want_obj) > 0.9995          # triangle edges seen edge-on may differ
    m = (want_obj == 2) & (hits["object"] == 2)
    assert np.mean(np.abs(hits["t"][m] - t_mesh[m]) < 1e-9) > 0.9995
    assert np.mean(np.all(np.abs(hits["n"][m] - n_mesh[m]) < 1e-7, axis=1)) > 0.999


# ---------------------------------------------------------------- materials and textures
def test_fresnel_and_refract_physics():
    """dielectric.rs:68-108: normal incidence ((n-1)/(n+1))^2 from either side, Snell's law for the refracted
    direction (unit length, in the plane of incidence, on the far side), total internal reflection, grazing -> 1,
    energy split between 0 and 1; direction.rs:75-77 reflect."""
    n = np.array([0.0, 1.0, 0.0])
    for ior in (1.5, 1.33, 2.4):
        r0 = ((ior - 1) / (ior + 1)) ** 2
        assert abs(oracle_lib.fresnel([0, -1, 0], n, ior) - r0) < 1e-15          # from outside
        assert abs(oracle_lib.fresnel([0, 1, 0], n, ior) - r0) < 1e-15           # from inside
        assert np.allclose(oracle_lib.refract([0, -1, 0], n, ior), [0, -1, 0], atol=1e-15)
        for deg in (5, 30, 60, 85):
            a = np.radians(deg)
            inc = np.array([np.sin(a), -np.cos(a), 0.0])                         # entering
            out = oracle_lib.refract(inc, n, ior)
            assert abs(np.linalg.norm(out) - 1) < 1e-12 and abs(out[2]) < 1e-15 and out[1] < 0
            assert abs(out[0] - np.sin(a) / ior) < 1e-12                         # sin(theta_t) = sin(theta_i) / ior
            kr = oracle_lib.fresnel(inc, n, ior)
            ct = np.sqrt(1 - (np.sin(a) / ior) ** 2)
            rs = (np.cos(a) - ior * ct) / (np.cos(a) + ior * ct)
            rp = (ior * np.cos(a) - ct) / (ior * np.cos(a) + ct)
            assert abs(kr - 0.5 * (rs * rs + rp * rp)) < 1e-12 and r0 - 1e-12 <= kr < 1
            inc_in = np.array([np.sin(a), np.cos(a), 0.0])                       # leaving: normal along the ray
            crit = np.degrees(np.arcsin(1 / ior))
            out = oracle_lib.refract(inc_in, n, ior)
            if deg > crit:
                assert np.all(out == 0) and oracle_lib.fresnel(inc_in, n, ior) == 1.0     # total internal reflection
            else:
                assert abs(out[0] - np.sin(a) * ior) < 1e-12 and out[1] > 0
        assert oracle_lib.fresnel([1, -1e-9, 0] / np.linalg.norm([1, -1e-9, 0]), n, ior) > 0.999999
    assert np.allclose(oracle_lib.reflect([1, -1, 0], n), [1, 1, 0])
    assert np.allclose(oracle_lib.reflect([0.3, 0.4, -0.5], [0, 0, 1]), [0.3, 0.4, 0.5])


def test_checkerboard_table_and_image_lookup_wrap():
    """texture.rs:54-94: the checkerboard is an XOR of which half of the unit cell u*scale and v*scale fall in,
    mirrored for negative coordinates (fract keeps the sign); the image lookup is nearest-texel on
    u * scale * (W - 1), wraps with %, saturates negatives to texel 0, row 0 on top."""
    c1, c2 = (1.0, 0.5, 0.25), (0.0, 0.25, 0.5)
    K = _capi.RT_TEX_CHECKERBOARD
    for scale in (1.0, 2.0, 0.5):
        for u in (0.1, 0.4, 0.6, 0.9, 1.2, 1.7, -0.1, -0.4, -0.6, -0.9, 3.25):
            for v in (0.2, 0.7, -0.2, -0.7, 5.6):
                fu, fv = np.fmod(u * scale, 1.0), np.fmod(v * scale, 1.0)
                half_u = (abs(fu) > 0.5) != (fu < 0)
                half_v = (abs(fv) > 0.5) != (fv < 0)
                want = c2 if (half_u != half_v) else c1
                assert np.allclose(oracle_lib.texture_color(K, u, v, c1, c2, scale), want), (scale, u, v)
    # same cell parity as a plain floor-based checkerboard of period 1/scale with squares of half a period
    for u, v in np.random.default_rng(2).uniform(-4, 4, (200, 2)):
        want = c2 if (int(np.floor(2 * u)) + int(np.floor(2 * v))) % 2 else c1
        assert np.allclose(oracle_lib.texture_color(K, u, v, c1, c2, 1.0), want)
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)             # H = 5, W = 7
    K = _capi.RT_TEX_IMAGE
    for u, v, scale in ((0.0, 0.0, 1.0), (1.0, 1.0, 1.0), (0.5, 0.5, 1.0), (0.99, 0.01, 1.0), (0.3, 0.8, 2.0),
                        (1.6, 2.3, 1.0), (-0.4, 0.5, 1.0), (0.5, -3.0, 1.0), (0.75, 0.25, 3.5)):
        x = int(max(u * scale * 6, 0.0)) % 7
        y = int(max(v * scale * 4, 0.0)) % 5
        assert np.allclose(oracle_lib.texture_color(K, u, v, scale=scale, image=img), img[y, x] / 255.0), (u, v, scale)


# ---------------------------------------------------------------- whole estimator: furnace identities
def test_convex_objects_in_a_uniform_environment(rt):
    """Furnace: a convex Lambertian object of albedo rho under a uniform background B returns exactly rho * B
    for every path (the scattered ray leaves along the outward normal's hemisphere and sees only background),
    whatever the random numbers: checks the outward normals, the bias offset and emit/attenuation bookkeeping
    of cast/hit_color (system.rs:112-142, lambertian.rs) for sphere, cube, cylinder, rectangle and plane."""
    shapes = {"sphere": "sphere { origin <0, 0, 0> radius 1 transform { scale <1, 0.6, 1.3> rotate_z 30 } }",
              "cube": "cube { <-0.8, -0.5, -0.7> <0.8, 0.5, 0.7> transform { rotate_y 25 rotate_x 40 } }",
              "cylinder": "cylinder { radius 0.8 height 1.4 transform { rotate_x 60 rotate_y 15 } }",
              "plane": "plane { origin <0, -1, 0> normal <0.1, 1, 0.2> }"}
    B, rho = np.array([0.9, 0.7, 0.5]), np.array([0.8, 0.6, 0.4])
    for name, shape in shapes.items():
        text = ("options { background color rgb <0.9, 0.7, 0.5> }\ncamera { origin <0, 0.5, 3.2> look_at <0, 0, 0> }\n"
                'object "x" { %s material { lambertian texture { solid color rgb <0.8, 0.6, 0.4> } } }\n' % shape)
        o = rt.Options(width=96, height=72, samples=4, seed=3)
        sc = rt.parse(o, text, parity.ASSETS)
        rgb, ids, st = oracle_lib.render(sc.desc, o.to_c(), threads=8)
        on = ids == 0
        assert on.mean() > 0.1, name
        # every stratum of a pixel may hit or miss the object; pixels whose 4 strata all hit it sum to 4 * rho * B
        full = np.all(np.isclose(rgb, 4 * rho * B, rtol=1e-12), axis=-1)
        miss = np.all(np.isclose(rgb, 4 * B, rtol=1e-12), axis=-1)
        mixed = ~(full | miss)
        # mixed pixels are silhouette pixels: k strata on the object, 4 - k off it
        k = (rgb[..., 0] - 4 * B[0]) / ((rho[0] - 1) * B[0])
        assert np.all(np.abs(k[mixed] - np.round(k[mixed])) < 1e-9), name
        assert full.sum() > 0.8 * on.sum() and st.rays > 0, name


def test_closed_box_depth_cap(rt):
    """A closed Lambertian box (six rectangles, the camera and a cube inside), no light: no path can escape, so
    every path ends at the depth cap and sees the background through all its attenuations: rho^max_depth * B
    (system.rs:112-120).  One leaked ray (a hole, an inward normal, a bias of the wrong sign) would add a full B."""
    rho, B = 0.9, 2.0
    wall = ' material { lambertian texture { solid color rgb <0.9, 0.9, 0.9> } } }\n'
    text = ("options { background color rgb <2, 2, 2> }\ncamera { origin <0, 0, 0.5> look_at <0, 0, -1> }\n"
            'object { xyrect { origin <0, 0, -2> width 4 height 4 }' + wall +
            'object { xyrect { origin <0, 0, 2> width 4 height 4 }' + wall +
            'object { xzrect { origin <0, -2, 0> width 4 height 4 }' + wall +
            'object { xzrect { origin <0, 2, 0> width 4 height 4 }' + wall +
            'object { zyrect { origin <-2, 0, 0> width 4 height 4 }' + wall +
            'object { zyrect { origin <2, 0, 0> width 4 height 4 }' + wall +
            'object { cube { <-0.5, -2, -1.5> <0.5, -0.8, -0.5> transform { rotate_y 20 } }' + wall)
    for depth in (1, 2, 7, 50):
        o = rt.Options(width=48, height=36, samples=1, seed=5, max_depth=depth)
        sc = rt.parse(o, text, parity.ASSETS)
        rgb, ids, st = oracle_lib.render(sc.desc, o.to_c(), threads=8)
        assert (ids >= 0).all()
        assert np.allclose(rgb, rho ** depth * B, rtol=1e-12), depth
        assert st.rays == 48 * 36 * depth


def test_fog_free_path_statistics(rt):
    """homogenous_medium.rs:29-57: inside a medium of density rho the free path is exponential: a ray with L of
    medium ahead scatters with probability 1 - exp(-rho L), at mean distance 1/rho - L e^(-rho L) / (1 - e^(-rho L))
    (Beer-Lambert), measured over 200 000 independent rays (each ray draws with its own key)."""
    rho, L = 0.2, 10.0
    text = CAM + ('object "fog" { homogenous_medium { density 0.2 cube { <-10, -10, -10> <10, 10, 10> } } '
                  "material { isotropic texture { solid color white } } }\n")
    n = 200000
    o = np.zeros((n, 3))
    d = np.tile([1.0, 0.0, 0.0], (n, 1))
    hits, _ = trace(rt, text, o, d)
    hit = hits["object"] == 0
    p = 1 - np.exp(-rho * L)
    assert abs(hit.mean() - p) < 4 * np.sqrt(p * (1 - p) / n), (hit.mean(), p)
    t = hits["t"][hit]
    assert t.min() >= 0 and t.max() < L
    mean = 1 / rho - L * np.exp(-rho * L) / p
    var = 1 / rho ** 2                                                  # upper bound of the truncated variance
    assert abs(t.mean() - mean) < 4 * np.sqrt(var / hit.sum()), (t.mean(), mean)
    # a ray that starts outside sees only the part inside: from x = -15 the medium spans t in [5, 25]
    o2 = np.tile([-15.0, 0.0, 0.0], (n, 1))
    hits2, _ = trace(rt, text, o2, d)
    h2 = hits2["object"] == 0
    p2 = 1 - np.exp(-rho * 20.0)
    assert abs(h2.mean() - p2) < 4 * np.sqrt(p2 * (1 - p2) / n)
    assert hits2["t"][h2].min() >= 5.0 and hits2["t"][h2].max() < 25.0


def test_lambertian_scatter_is_cosine_weighted(rt):
    """lambertian.rs:20-31 + direction.rs:29-39: normalize(n + S), S uniform on the unit sphere, is the
    cosine-weighted hemisphere: E[cos] = 2/3, E[cos^2] = 1/2, azimuth uniform.  Read off a two-segment render:
    a floor under a sky whose colour is a step function of direction is not available in the grammar, so use the
    furnace with an occluder: a light disc cannot be placed either; instead sample S through the oracle's RNG
    directly: a one-bounce path from a floor point to a large emissive ceiling rectangle of solid angle fraction f
    (cosine-weighted form factor) returns intensity * F, with F the analytic form factor of a parallel rectangle."""
    # floor point under the centre of a parallel a x b rectangle at height h: form factor (differential area to
    # parallel rectangle through its corner, four quadrants)
    h, a, b = 1.0, 3.0, 2.0

    def corner_ff(x, y, z):
        X, Y = x / z, y / z
        return (1 / (2 * np.pi)) * (X / np.sqrt(1 + X * X) * np.arctan(Y / np.sqrt(1 + X * X)) +
                                    Y / np.sqrt(1 + Y * Y) * np.arctan(X / np.sqrt(1 + Y * Y)))
    F = 4 * corner_ff(a / 2, b / 2, h)
    text = ("camera { origin <0, 0.5, 0.0001> look_at <0, 0, 0> fov 0.001 }\n"
            'object "floor" { plane { origin <0, 0, 0> normal <0, 1, 0> } material { lambertian texture { solid color white } } }\n'
            'object "lamp" { xzrect { origin <0, 1, 0> width 3 height 2 } material { diffuse_light intensity 5 texture { solid color white } } }\n')
    o = rt.Options(width=256, height=256, samples=16, seed=11, max_depth=2)
    sc = rt.parse(o, text, parity.ASSETS)
    rgb, ids, st = oracle_lib.render(sc.desc, o.to_c(), threads=8)
    assert (ids == 0).all()                                             # the narrow camera sees the floor below the lamp's centre
    est = rgb[..., 0].mean() / 16 / 5.0                                 # mean over 1 M paths of [scattered ray meets the lamp]
    sigma = np.sqrt(F * (1 - F) / (256 * 256 * 16))
    assert abs(est - F) < 4 * sigma, (est, F, sigma)


# ---------------------------------------------------------------- host plumbing against independent numbers
def _probe():
    """a tiny C++ probe around the product's host_math.h (built here with g++)"""
    src = os.path.join(parity.ROOT, "tests", "helpers", "host_math_probe.cpp")
    out = os.path.join(parity.ROOT, "tests", "helpers", "_host_math_probe.so")
    hdr = os.path.join(parity.ROOT, "raytracer_amd", "csrc", "host_math.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", out])
    lib = C.CDLL(out)
    dp = C.POINTER(C.c_double)
    lib.probe_inverse.argtypes = [dp, dp]
    lib.probe_mul.argtypes = [dp, dp, dp]
    lib.probe_make.argtypes = [C.c_int, dp, dp]
    return lib


def test_host_math_inverse_against_the_reference_vector_and_the_oracle(rt):
    """The product's own Gauss-Jordan (raytracer_amd/csrc/host_math.h, used by the parser for every w2o) on the
    reference's inverse vector (matrix.rs:257-275), on its transform constructors, and bit for bit against the
    oracle's separate copy on every node matrix of the nine shipped scenes."""
    lib = _probe()
    dp = C.POINTER(C.c_double)

    def inv(m):
        a = np.ascontiguousarray(m, dtype=np.float64).reshape(16)
        out = np.zeros(16)
        lib.probe_inverse(a.ctypes.data_as(dp), out.ctypes.data_as(dp))
        return out.reshape(4, 4)
    m = np.array([[1, 3, 2, 4], [4, 2, 3, 5], [5, 4, 3, 1], [3, 1, 2, 4]], dtype=float)
    want = np.array([[0, -20, 4, 24], [6, -20, 4, 18], [-9, 64, -8, -69], [3, -12, 0, 15]]) / 12.0
    assert np.allclose(inv(m), want, atol=1e-12)
    for kind, args, M in ((0, [1, 2, 3], T(1, 2, 3)), (1, [2, 3, 4], S(2, 3, 4)), (2, [30], RX(30)), (3, [-75], RY(-75)),
                          (4, [110], RZ(110))):
        a = np.array(args + [0, 0], dtype=float)
        out = np.zeros(16)
        lib.probe_make(kind, a.ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert np.allclose(out.reshape(4, 4), M, atol=1e-15), kind
    n_checked = 0
    for name in parity.SCENES:
        o = rt.Options(width=64, height=48)
        sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
        d = sc.desc.contents
        for k in range(d.n_nodes):
            o2w = np.array(d.nodes[k].o2w[:]).reshape(4, 4)
            w2o = np.array(d.nodes[k].w2o[:]).reshape(4, 4)
            mine = inv(o2w)
            theirs = np.zeros(16)
            oracle_lib.lib().oracle_matrix_inverse(np.ascontiguousarray(o2w.reshape(16)).ctypes.data_as(dp), theirs.ctypes.data_as(dp))
            assert np.array_equal(mine, theirs.reshape(4, 4)) and np.array_equal(mine, w2o), (name, k)
            assert np.allclose(o2w @ w2o, np.eye(4), atol=1e-9)
            n_checked += 1
    assert n_checked > 60


def test_parser_transform_folding_against_numpy(rt):
    """sdl.rs:44-47,94-96 + object.rs:21-26: `transform { a b c }` folds left to right into o2w = A*B*C, and w2o
    is its inverse; a nested CSG child keeps its own.  Expected matrices from the numpy constructors above."""
    text = CAM + ('object "a" { difference { sphere { origin <0, 0, 0> radius 1 transform { scale <2, 1, 0.5> rotate_y 90 translate <1, 2, 3> } } '
                  "cube { <0, 0, 0> <1, 1, 1> transform { rotate_x 45 } } transform { rotate_z -30 scale <1, 2, 1> translate <0, -1, 0> } }" + MAT)
    o = rt.Options(width=64, height=48)
    sc = rt.parse(o, text, parity.ASSETS)
    d = sc.desc.contents
    by_kind = {d.nodes[k].kind: k for k in range(d.n_nodes)}
    want = {_capi.RT_SHAPE_SPHERE: chain(S(2, 1, 0.5), RY(90), T(1, 2, 3)), _capi.RT_SHAPE_CUBE: RX(45),
            _capi.RT_SHAPE_DIFFERENCE: chain(RZ(-30), S(1, 2, 1), T(0, -1, 0))}
    for kind, M in want.items():
        nd = d.nodes[by_kind[kind]]
        assert np.allclose(np.array(nd.o2w[:]).reshape(4, 4), M, atol=1e-15), kind
        assert np.allclose(np.array(nd.w2o[:]).reshape(4, 4), np.linalg.inv(M), atol=1e-14), kind
    # rotate_y 90 takes +x to -z (matrix.rs:278-477's handedness), scale first then translate
    p = np.array([1.0, 0, 0, 1]) @ want[_capi.RT_SHAPE_SPHERE]
    assert np.allclose(p[:3], [1, 2, 3 - 2])


def test_camera_against_pinhole_geometry(rt):
    """Camera::new / pixel_ray (system.rs:37-66) for a level camera (no pitch: the un-normalised x axis quirk is
    absent): the ray through pixel (x, y) passes through the point of the image plane that a textbook pinhole
    with vertical field of view `fov` and pixel centres at integer coordinates assigns to it."""
    W, H, fov = 640, 480, 50.0
    eye, at = np.array([1.0, 2.0, 3.0]), np.array([4.0, 2.0, -1.0])
    fwd = (at - eye) / np.linalg.norm(at - eye)
    right = np.cross(fwd, [0, 1, 0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    dp = C.POINTER(C.c_double)
    for x, y in ((0, 0), (320, 240), (639, 479), (100.25, 33.5)):
        out = np.zeros(6)
        oracle_lib.lib().oracle_camera_ray(W, H, fov, eye.ctypes.data_as(dp), at.ctypes.data_as(dp), x, y, None, None,
                                           out.ctypes.data_as(dp))
        sx = (2 * x / W - 1) * np.tan(np.radians(fov) / 2) * W / H
        sy = (1 - 2 * y / H) * np.tan(np.radians(fov) / 2)
        want = fwd + sx * right + sy * up
        want /= np.linalg.norm(want)
        assert np.allclose(out[:3], eye) and np.allclose(out[3:], want, atol=1e-12), (x, y)
