"""Mesh walk stress: every shipped mesh under random object transforms, rays from everywhere (far away, inside the
bounds, grazing the grid planes, axis-parallel), long queues so that the grid / kd walks (not the wave-per-ray
kernel) take them; device hits against the oracle's, ray for ray.  python tools/mesh_stress.py [--configs 30] [--rays 60000]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib, parity  # noqa: E402
import raytracer_amd as rt  # noqa: E402



def run(configs, rays):
  import types
  a = types.SimpleNamespace(configs=configs, rays=rays)
  bad = 0
  for cfg in range(a.configs):
      rng = np.random.default_rng(1000 + cfg)
      mesh = ["LinkedTorus.obj", "Monkey.obj", "two_parts.obj"][cfg % 3]
      tx = []
      if cfg >= 3:
          if rng.random() < 0.7: tx.append("scale <%.3f, %.3f, %.3f>" % tuple(rng.uniform(0.3, 3.0, 3)))
          for ax in "xyz":
              if rng.random() < 0.6: tx.append("rotate_%s %.2f" % (ax, rng.uniform(-180, 180)))
          tx.append("translate <%.3f, %.3f, %.3f>" % tuple(rng.uniform(-2, 2, 3) * (100.0 if cfg % 7 == 0 else 1.0)))
      text = ('camera { origin <0, 0, 9> look_at <0, 0, 0> }\n'
              'object "m" { mesh { file "%s" transform { %s } } material { lambertian texture { solid color white } } }\n'
              % (mesh, " ".join(tx) if tx else "translate <0, 0, 0>"))
      o = rt.Options(width=16, height=16)
      sc = rt.parse(o, text, parity.ASSETS)
      # world position of the mesh: transform of its object-space bounds' centre, from the desc
      d = sc.desc.contents
      node = [d.nodes[k] for k in range(d.n_nodes) if d.nodes[k].kind == 11][0]
      M = np.array(node.o2w[:]).reshape(4, 4)
      V = np.ctypeslib.as_array(d.vertices, shape=(d.n_vertices, 3))
      Vw = np.c_[V, np.ones(len(V))] @ M
      c, ext = Vw[:, :3].mean(0), np.ptp(Vw[:, :3], axis=0).max()
      n = a.rays
      kind = rng.integers(0, 5, n)
      org = c + rng.normal(size=(n, 3)) * ext * np.where(kind == 0, 1000.0, np.where(kind == 1, 0.3, 1.5))[:, None]
      tgt = c + rng.normal(size=(n, 3)) * ext * 0.35
      dirs = tgt - org
      ax = kind == 3                                           # axis-parallel rays
      dirs[ax] = np.eye(3)[rng.integers(0, 3, ax.sum())] * rng.choice([-1.0, 1.0], ax.sum())[:, None]
      dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
      got, _ = rt.DeviceScene(sc).trace_rays(org, dirs, seed=4)
      ref, _ = oracle_lib.trace_rays(sc.desc, org, dirs, seed=4)
      same = got["object"] == ref["object"]
      hit = same & (ref["object"] >= 0)
      t_exact = np.mean(got["t"][hit] == ref["t"][hit]) if hit.any() else 1.0
      t_close = np.mean(np.abs(got["t"][hit] - ref["t"][hit]) <= 1e-9 * np.maximum(1, np.abs(ref["t"][hit]))) if hit.any() else 1.0
      n_close = np.mean(np.all(np.abs(got["n"][hit] - ref["n"][hit]) < 1e-9, axis=1)) if hit.any() else 1.0
      ok = same.all() and t_close == 1.0 and n_close == 1.0
      bad += 0 if ok else 1
      print("cfg %2d %-16s hits %.3f  same object %.6f  t bit-equal %.6f  t within 1e-9 %.6f  n within 1e-9 %.6f  %s  [%s]"
            % (cfg, mesh, (ref["object"] >= 0).mean(), same.mean(), t_exact, t_close, n_close, "ok" if ok else "DIFFERS", " ".join(tx)), flush=True)
  print("%d of %d configurations differ" % (bad, a.configs))
  return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=30)
    ap.add_argument("--rays", type=int, default=60000)
    args = ap.parse_args()
    sys.exit(1 if run(args.configs, args.rays) else 0)
