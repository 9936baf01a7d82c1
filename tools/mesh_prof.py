"""k_mesh phase counters (library built with -DRT_MESH_PROF) per depth limit: python tools/mesh_prof.py
RT_AMD_LIB must point at the profiling build; the counters go to stderr, one pair of lines per frame."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import raytracer_amd as rt, parity
for depth in (1, 2, 3, 50):
    o = rt.Options(width=3840, height=2160, samples=1, seed=1, max_depth=depth)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    dev = rt.DeviceScene(sc)
    buf = torch.zeros(3840 * 2160 * 3, dtype=torch.float64, device="cuda")
    sys.stderr.write("max_depth %d\n" % depth); sys.stderr.flush()
    dev.render_shard(o, 0, 1, buf.data_ptr(), None, None, row_major=True)
