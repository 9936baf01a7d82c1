"""Where a random scene (tests/fuzz_scenes.py) differs between the GPU and the oracle: primary ids, then the
first hits of the rays of the differing pixels.

    python tools/fuzz_diag.py SEED [--no-torus] [--width 160 --height 120 --samples 4]
"""
import argparse
import ctypes as C
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
    ap.add_argument("seed", type=int)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--height", type=int, default=120)
    ap.add_argument("--samples", type=int, default=4)
    ap.add_argument("--objects", type=int, default=10)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-torus", action="store_true")
    a = ap.parse_args()
    text = fuzz_scenes.scene_text(a.seed, n_objects=a.objects, torus=not a.no_torus)
    objs = [ln for ln in text.split("\n") if ln.startswith("object")]
    o = rt.Options(width=a.width, height=a.height, samples=a.samples, seed=a.seed, max_depth=a.depth)
    sc = rt.parse(o, text, parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    print("rays", st.rays, rst.rays)
    bad_id = np.argwhere(ids != rids)
    bad_px = np.argwhere(np.any(np.abs(rgb - ref) > 1e-9 * np.maximum(1, np.abs(ref)), axis=-1))
    print("id mismatches", len(bad_id), "pixel mismatches", len(bad_px))
    for y, x in bad_id[:8]:
        print("  id px", (int(y), int(x)), "gpu", ids[y, x], "ref", rids[y, x])
    for y, x in bad_px[:8]:
        print("  px", (int(y), int(x)), "gpu", rgb[y, x], "ref", ref[y, x], "ids", ids[y, x], rids[y, x])
    # camera rays through the centres of the suspicious pixels (Camera::primary_ray, system.rs:57-66)
    cam = sc.desc.contents.camera
    m = np.array(list(cam.camera_to_world)).reshape(4, 4)
    px = np.concatenate([bad_id, bad_px])[:16]
    if len(px) == 0:
        return
    aspect = a.width / a.height
    def rays(offsets):
        ro, rd = [], []
        for y, x in px:
            for oy, ox in offsets:
                cx = (2.0 * (x + ox) / a.width - 1.0) * aspect * cam.fov_factor
                cy = (1.0 - 2.0 * (y + oy) / a.height) * cam.fov_factor
                org = np.array([0, 0, 0, 1.0]) @ m
                tgt = np.array([cx, cy, -1.0, 1.0]) @ m
                d = (tgt - org)[:3]
                ro.append(org[:3])
                rd.append(d / np.linalg.norm(d))
        return np.array(ro), np.array(rd)

    # a bundle of rays through each suspicious pixel: which of them disagree?
    grid = [((i + 0.5) / 24, (j + 0.5) / 24) for i in range(24) for j in range(24)]
    ro, rd = rays(grid)
    dev0 = rt.DeviceScene(sc)
    got, _ = dev0.trace_rays(ro, rd, seed=4)
    want, _ = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    diff = np.flatnonzero(got["object"] != want["object"])
    print("bundle rays disagreeing:", len(diff), "of", len(ro))
    for k in diff[:6]:
        print("   o", ro[k].tolist(), "d", rd[k].tolist(), "gpu", got["object"][k], got["t"][k], "ref", want["object"][k], want["t"][k])
    if len(diff):
        np.savez(os.path.join(ROOT, "gpurun_out", "diag_rays_%d.npz" % a.seed), o=ro[diff], d=rd[diff],
                 gpu_obj=got["object"][diff], gpu_t=got["t"][diff], ref_obj=want["object"][diff], ref_t=want["t"][diff])
    ro, rd = rays([(0.5, 0.5)])
    dev = rt.DeviceScene(sc)
    got, _ = dev.trace_rays(ro, rd, seed=4)
    want, _ = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    for k in range(len(px)):
        print("  centre ray px", tuple(int(v) for v in px[k]), "gpu obj %d t %.12g" % (got["object"][k], got["t"][k]),
              "ref obj %d t %.12g" % (want["object"][k], want["t"][k]))
    for k in sorted(set(ids[tuple(px.T)].tolist() + rids[tuple(px.T)].tolist())):
        if k >= 0:
            print("object", k, objs[k])


if __name__ == "__main__":
    main()
