"""Per-level phase times of k_tail_grid on the bench frame: python tools/tailg_prof.py  (library built with -DRT_TAILG_PROF,
selected with RT_AMD_LIB; the kernel's waves mark the 100 MHz wall clock between phases, the library prints the table)."""
import dataclasses, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import system
scene_name = sys.argv[1] if len(sys.argv) > 1 else "mesh"
o = rt.Options(width=3840, height=2160, samples=1, seed=1)
scene = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/%s.scene" % scene_name), os.path.join(ROOT, "tests/assets"))
dev = rt.DeviceScene(scene)
buf = torch.zeros(3840 * 2160 * 3, dtype=torch.float64, device="cuda")
for k in range(3):
    print("frame", k, file=sys.stderr, flush=True)
    dev.render_shard(dataclasses.replace(o, seed=1 + k), 0, 1, buf.data_ptr(), None, None, row_major=True)
