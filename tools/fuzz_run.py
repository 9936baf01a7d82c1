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
        print("seed %3d %s ids %.5f err %.4f close %.5f rays %d/%d guards %s/%s objs %d" % (
            seed, "ok " if ok else "BAD", m["id_match"], m["mean_err_255"], m["px_close"], st.rays, rst.rays,
            list(st.guards), list(rst.guards), len(np.unique(rids))), flush=True)
        if not ok:
            bad += 1
            if a.dump:
                os.makedirs(a.dump, exist_ok=True)
                with open(os.path.join(a.dump, "fuzz_%d.scene" % seed), "w") as f:
                    f.write(text)
    print("bad: %d of %d" % (bad, len(seeds)))


if __name__ == "__main__":
    main()
