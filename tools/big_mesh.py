"""A large generated mesh (bumpy sphere, NU x NV quads) through both mesh structures and the oracle:
    python tools/big_mesh.py [NU NV]      (default 500 250 = 250 000 triangles)"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, parity
import raytracer_amd as rt

NU = int(sys.argv[1]) if len(sys.argv) > 1 else 500
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 250


def write_obj(path):
    u = np.linspace(0, 2 * np.pi, NU, endpoint=False)
    v = np.linspace(0.02, np.pi - 0.02, NV + 1)
    uu, vv = np.meshgrid(u, v)
    r = 1.0 + 0.05 * np.sin(9 * uu) * np.sin(7 * vv)
    p = np.stack([r * np.sin(vv) * np.cos(uu), r * np.cos(vv), r * np.sin(vv) * np.sin(uu)], -1).reshape(-1, 3)
    n = p / np.linalg.norm(p, axis=1, keepdims=True)
    with open(path, "w") as f:
        f.write("o bumpy\n")
        for q in p:
            f.write("v %.9f %.9f %.9f\n" % tuple(q))
        for q in n:
            f.write("vn %.9f %.9f %.9f\n" % tuple(q))
        for j in range(NV):
            for i in range(NU):
                a = j * NU + i + 1; b = j * NU + (i + 1) % NU + 1; c = a + NU; d = b + NU
                f.write("f %d//%d %d//%d %d//%d\nf %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c, b, b, d, d, c, c))


tmp = tempfile.mkdtemp()
write_obj(os.path.join(tmp, "bumpy.obj"))
text = """camera { origin <0, 1.2, 3.2> look_at <0, 1, 0> }
object "light" { sphere { origin <6, 8, 6> radius 4 } material { diffuse_light intensity 3 texture { solid color white } } }
object "floor" { plane { origin <0, 0, 0> normal up } material { lambertian texture { solid color rgb <0.5, 0.6, 0.4> } } }
object "bumpy" { mesh { file "bumpy.obj" transform { rotate_x 30 translate <0, 1.1, 0> } } material { metal fuzz 0.05 texture { solid color rgb <0.8, 0.7, 0.5> } } }
"""
o = rt.Options(width=64, height=48, samples=1, seed=2, max_depth=4)
ref = None
for accel in ("grid", "kd", "auto"):
    if accel == "auto":
        os.environ.pop("RT_ACCEL", None)
    else:
        os.environ["RT_ACCEL"] = accel
    t0 = time.time()
    sc = rt.parse(o, text, tmp)
    t1 = time.time()
    dev = rt.DeviceScene(sc)
    t2 = time.time()
    rgb, ids, st = dev.render(o)
    print(accel, "tris", sc.desc.contents.n_triangles if hasattr(sc.desc.contents, "n_triangles") else "?",
          "parse %.2fs build %.2fs" % (t1 - t0, t2 - t1), "rays", st.rays, flush=True)
    if ref is None:
        t3 = time.time()
        ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
        print("oracle %.1fs rays %d" % (time.time() - t3, rst.rays), flush=True)
    m = parity.compare(rgb, ids, ref, rids, st.strata)
    print(accel, m, "guards", list(st.guards), list(rst.guards), flush=True)
    big = rt.Options(width=1920, height=1080, samples=1, seed=2, stage_times=True)
    dev.render(big, want_ids=False)
    t4 = time.time()
    _, _, bst = dev.render(big, want_ids=False)
    dt = time.time() - t4
    print(accel, "1080p: %.1f ms, %.0f Mrays/s (host clock, includes read-back)" % (dt * 1e3, bst.rays / dt / 1e6),
          "stage ms", round(bst.ms_intersect, 2), round(bst.ms_mesh, 2), round(bst.ms_shade, 2), round(bst.ms_tail, 2), flush=True)
