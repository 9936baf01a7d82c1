"""Shared helpers for comparing the HIP path with the CPU oracle."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "tests", "assets")
SCENES = ["three-spheres", "cornell_box", "mesh", "torus", "csg", "cylinder", "transformations", "room", "fog_light"]


def scene_path(name):
    return os.path.join(ASSETS, "scenes", name + ".scene")


def compare(rgb_gpu, ids_gpu, rgb_ref, ids_ref, strata):
    """north_star metrics: primary-hit id agreement and mean per-channel error of
    the tone-mapped image (in units of 1/255), plus tighter diagnostics."""
    import oracle_lib
    id_match = float(np.mean(ids_gpu == ids_ref))
    img_g = oracle_lib.tonemap(rgb_gpu, strata + 1).astype(np.float64)
    img_r = oracle_lib.tonemap(rgb_ref, strata + 1).astype(np.float64)
    mean_err_255 = float(np.mean(np.abs(img_g - img_r)))
    denom = np.maximum(np.abs(rgb_ref), 1e-300)
    rel = np.abs(rgb_gpu - rgb_ref) / denom
    rel = np.where((rgb_ref == 0) & (rgb_gpu == 0), 0.0, rel)
    px_close = np.all((rel < 1e-9) | (np.abs(rgb_gpu - rgb_ref) < 1e-12), axis=-1)
    return {"id_match": id_match, "mean_err_255": mean_err_255, "px_close": float(np.mean(px_close)),
            "max_abs": float(np.max(np.abs(rgb_gpu - rgb_ref)))}
