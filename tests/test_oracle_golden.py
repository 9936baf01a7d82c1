"""The CPU oracle against every known-answer vector the reference's own unit tests hold
(tests/golden/reference_unit_vectors.json).  CPU only."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import descs
import oracle_lib
from oracle_lib import _dp, arr

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_unit_vectors.json")) as f:
    GOLD = json.load(f)
TOL = GOLD["tolerance"]


@pytest.mark.parametrize("case", GOLD["shape_intervals"], ids=lambda c: c["src"])
def test_shape_intervals(case):
    d = descs.Desc([case["shape"]])
    n, iv = oracle_lib.shape_intervals(d.ptr, 0, case["o"], case["d"])
    assert n * 2 == len(case["t"])
    flat = iv.reshape(-1, 6)
    assert np.allclose(flat[:, 0], case["t"], atol=TOL, rtol=0)
    assert np.allclose(flat[:, 1:4], case["n"], atol=TOL, rtol=0)


def test_torus_oblique_geometric_answer():
    c = GOLD["torus_oblique"]
    o = np.array(c["o"], dtype=float)
    dvec = np.array(c["toward"], dtype=float) - o
    dvec /= np.linalg.norm(dvec)
    d = descs.Desc([c["shape"]])
    n, iv = oracle_lib.shape_intervals(d.ptr, 0, o, dvec)
    assert n == 1
    assert np.allclose(iv.reshape(-1, 6)[:, 0], c["t"], atol=TOL, rtol=0)
    # normals: towards the ray origin on entry, away on exit
    assert np.allclose(iv[0, 0, 1:4], -dvec, atol=TOL) and np.allclose(iv[0, 1, 1:4], dvec, atol=TOL)


@pytest.mark.parametrize("case", GOLD["plane_intersect"], ids=lambda c: c["src"])
def test_plane_intersect(case):
    d = descs.Desc([{"kind": "plane", "params": case["origin"] + case["normal"]}])
    rc, out = oracle_lib.shape_intersect(d.ptr, 0, case["o"], case["d"])
    assert (rc == 1) == case["hit"]
    if case["hit"]:
        assert abs(out[0] - case["t"]) < TOL
        if "n" in case:
            assert np.allclose(out[1:4], case["n"], atol=TOL, rtol=0)


def _cmul(a, b):
    return complex(a) * complex(b)


def _poly_from_roots(m, roots):
    """coefficients, highest degree first, of m * prod (x - r)"""
    coef = np.array([complex(*m)])
    for r in roots:
        coef = np.convolve(coef, np.array([1.0, -complex(*r)]))
    return coef


@pytest.mark.parametrize("degree", ["quadratic", "cubic", "quartic"])
def test_algebra_known_roots(degree):
    L = oracle_lib.lib()
    fn = {"quadratic": L.oracle_solve_quadratic, "cubic": L.oracle_solve_cubic, "quartic": L.oracle_solve_quartic}[degree]
    tol = GOLD["algebra"]["root_tolerance"]
    for case in GOLD["algebra"][degree]:
        coef = _poly_from_roots(case["m"], case["roots"])
        flat = arr(*[v for c in coef for v in (c.real, c.imag)])
        out = np.zeros(8)
        n = fn(_dp(flat), _dp(out))
        assert n == case["expected_count"], case
        found = [complex(out[2 * k], out[2 * k + 1]) for k in range(n)]
        used = [False] * n
        # check_roots (algebra.rs:162-185): every found root matches a distinct known one
        for k in range(n):
            known = complex(*case["roots"][k])
            ok = False
            for j in range(n):
                if not used[j] and abs((known - found[j]).real) < tol and abs((known - found[j]).imag) < tol:
                    used[j] = True
                    ok = True
                    break
            assert ok, (case, found)


def test_quartic_f64_real_roots():
    out = np.zeros(4)
    coef = np.real(_poly_from_roots([1.0, 0.0], [[2, 0], [3, 0], [4, 0], [5, 0]]))
    n = oracle_lib.lib().oracle_solve_quartic_f64(*[float(c) for c in coef], _dp(out))
    assert n == 4 and np.allclose(sorted(out), [2, 3, 4, 5], atol=1e-8)


def _matrix(kind, args):
    code = {"identity": 0, "translation": 1, "scaling": 2, "rotation_x": 3, "rotation_y": 4, "rotation_z": 5}[kind]
    out = np.zeros(16)
    oracle_lib.lib().oracle_matrix_make(code, _dp(arr(*(list(args) + [0, 0, 0]))), _dp(out))
    return out


def test_matrix_inverse():
    L = oracle_lib.lib()
    ident = _matrix("identity", [])
    out = np.zeros(16)
    L.oracle_matrix_inverse(_dp(ident), _dp(out))
    assert np.array_equal(out, ident)
    g = GOLD["matrix"]["inverse"]
    m = arr(*np.array(g["m"], dtype=float).ravel())
    L.oracle_matrix_inverse(_dp(m), _dp(out))
    assert np.allclose(out, np.array(g["expected_times_12"], dtype=float).ravel() / 12.0, atol=TOL, rtol=0)
    prod = np.zeros(16)
    L.oracle_matrix_mul(_dp(m), _dp(out.copy()), _dp(prod))
    assert np.allclose(prod, ident, atol=TOL, rtol=0)


@pytest.mark.parametrize("case", GOLD["matrix"]["transforms"], ids=lambda c: c["src"])
def test_matrix_transforms(case):
    L = oracle_lib.lib()
    m = _matrix("identity", [])
    first = True
    for kind, args in case["chain"]:
        nxt = _matrix(kind, args)
        if first:
            m, first = nxt, False
        else:
            out = np.zeros(16)
            L.oracle_matrix_mul(_dp(m), _dp(nxt), _dp(out))
            m = out
    res = np.zeros(3)
    fn = L.oracle_point_mul_matrix if case["kind"] == "point" else L.oracle_direction_mul_matrix
    fn(_dp(arr(*case["v"])), _dp(m), _dp(res))
    assert np.allclose(res, case["expected"], atol=TOL, rtol=0)


@pytest.mark.parametrize("case", GOLD["color"]["cases"], ids=lambda c: c["op"])
def test_color_ops(case):
    op = {"add": 0, "add_assign": 0, "mul_color": 1, "mul_f64": 2, "div_f64": 3}[case["op"]]
    out = np.zeros(3)
    b = case["b"] + [0, 0]
    oracle_lib.lib().oracle_color_op(op, _dp(arr(*case["a"])), _dp(arr(*b)), _dp(out))
    assert np.all(np.abs(out - np.array(case["expected"])) <= 2.220446049250313e-16)


@pytest.mark.parametrize("case", GOLD["philox4x32_10"]["cases"], ids=lambda c: c["out"][0])
def test_philox_known_answers(case):
    ctr = (C.c_uint32 * 4)(*[int(x, 16) for x in case["ctr"]])
    key = (C.c_uint32 * 2)(*[int(x, 16) for x in case["key"]])
    out = (C.c_uint32 * 4)()
    oracle_lib.lib().oracle_philox(ctr, key, out)
    assert [("%08x" % v) for v in out] == case["out"]


def test_camera_quirks():
    """system.rs:37-66: pixel centre at integer coordinates, x axis not re-normalised."""
    L = oracle_lib.lib()
    c2w = np.zeros(16)
    fov = np.zeros(1)
    ray = np.zeros(6)
    L.oracle_camera_ray(640.0, 480.0, 60.0, _dp(arr(-2, 3, 4)), _dp(arr(0, 1, 0)), 320.0, 240.0, _dp(c2w), _dp(fov), _dp(ray))
    assert abs(fov[0] - math.tan(math.radians(30.0))) < 1e-15
    z = np.array([-2.0, 2.0, 4.0])
    z /= np.linalg.norm(z)
    x = np.cross([0, 1, 0], z)
    assert np.allclose(c2w[0:3], x) and np.linalg.norm(x) < 1.0  # pitched camera: |xaxis| < 1
    assert np.allclose(ray[0:3], [-2, 3, 4])
    assert np.allclose(ray[3:6], -z, atol=1e-12)  # (320, 240) is the exact image centre
