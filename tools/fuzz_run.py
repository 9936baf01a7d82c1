"""Render random scenes (tests/fuzz_scenes.py) on the GPU and with the oracle; print the parity metrics per seed.

    python tools/fuzz_run.py --seeds 1-40 [--width 160 --height 120 --samples 4] [--no-torus] [--dump DIR]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fuzz_scenes  # noqa: E402
import oracle_lib  # noqa: E402
import parity  # noqa: E402
import raytracer_amd as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1-20")
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--objects", type=int, default=10)
    ap.add_argument("--no-torus", action="store_true")
    ap.add_argument("--dump", default=None)
    ap.add_argument("--rays", type=int, default=0, help="also trace this many random rays per scene (origins anywhere, also inside solids)")
    a = ap.parse_args()
    lo, _, hi = a.seeds.partition("-")
    seeds = range(int(lo), int(hi or lo) + 1)
    bad = 0
    for seed in seeds:
        text = fuzz_scenes.scene_text(seed, n_objects=a.objects, torus=not a.no_torus)
        o = rt.Options(width=a.width, height=a.height, samples=a.samples, seed=seed)
        try:
            sc = rt.parse(o, text, parity.ASSETS)
        except Exception as e:  # a generator bug, not a parity failure
            print("seed %d: parse error %s" % (seed, e), flush=True)
            bad += 1
            continue
        rgb, ids, st = rt.render(o, sc)
        ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
        m = parity.compare(rgb, ids, ref, rids, st.strata)
        ok = (m["id_match"] >= 0.999 and m["mean_err_255"] <= 1.0 and m["px_close"] >= 0.999
              and list(st.guards) == list(rst.guards) and abs(int(st.rays) - int(rst.rays)) <= max(2, 1e-3 * rst.rays))
        ray_note = ""
        if a.rays:
            rng = np.random.default_rng(1000 + seed)
            ro = np.array([0.0, 2.2, 0.0]) + rng.normal(size=(a.rays, 3)) * 2.5
            tg = np.array([0.0, 2.2, 0.0]) + rng.normal(size=(a.rays, 3)) * 1.5
            rd = tg - ro
            rd /= np.linalg.norm(rd, axis=1, keepdims=True)
            dev = rt.DeviceScene(sc)
            got, gst = dev.trace_rays(ro, rd, seed=seed)
            want, wst = oracle_lib.trace_rays(sc.desc, ro, rd, seed=seed)
            dev.close()
            same = got["object"] == want["object"]
            hit = same & (want["object"] >= 0)
            scale = np.maximum(1.0, np.abs(want["t"][hit]))
            t_ok = float(np.mean(np.abs(got["t"][hit] - want["t"][hit]) / scale < 1e-9)) if hit.any() else 1.0
            n_ok = float(np.mean(np.all(np.abs(got["n"][hit] - want["n"][hit]) < 1e-7, axis=1))) if hit.any() else 1.0
            uv_ok = float(np.mean(np.all(np.abs(got["uv"][hit] - want["uv"][hit]) < 1e-6, axis=1))) if hit.any() else 1.0
            ray_ok = same.mean() >= 0.999 and t_ok >= 0.999 and n_ok >= 0.999 and uv_ok >= 0.995 and list(gst.guards) == list(wst.guards)
            ray_note = " | rays obj %.5f t %.5f n %.5f uv %.5f guards %s/%s" % (same.mean(), t_ok, n_ok, uv_ok, list(gst.guards), list(wst.guards))
            ok = ok and ray_ok
        print("seed %3d %s ids %.5f err %.4f close %.5f rays %d/%d guards %s/%s objs %d" % (
            seed, "ok " if ok else "BAD", m["id_match"], m["mean_err_255"], m["px_close"], st.rays, rst.rays,
            list(st.guards), list(rst.guards), len(np.unique(rids))) + ray_note, flush=True)
        if not ok:
            bad += 1
            if a.dump:
                os.makedirs(a.dump, exist_ok=True)
                with open(os.path.join(a.dump, "fuzz_%d.scene" % seed), "w") as f:
                    f.write(text)
    print("bad: %d of %d" % (bad, len(seeds)))


if __name__ == "__main__":
    main()
