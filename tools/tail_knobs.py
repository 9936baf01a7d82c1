"""Lone-frame time of the whole 4K bench frame and of shard 0 of 8 under the tail's knobs (same process, interleaved repeats):
python tools/tail_knobs.py   (RT_AMD_LIB selects the build; RT_TAIL_PATHS values from TAIL_PATHS)"""
import dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import system
o = rt.Options(width=3840, height=2160, samples=1, seed=1)
scene = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/mesh.scene"), os.path.join(ROOT, "tests/assets"))
fracs = [v for v in os.environ.get("TAIL_PATHS", "140000").split(",")]
dev = rt.DeviceScene(scene)
stream = torch.cuda.Stream()
for n in (1, 8):
    buf = torch.zeros(system.shard_pixels(o, 0, n) * 3, dtype=torch.float64, device="cuda")
    res = {f: [] for f in fracs}
    for rep in range(3):
        for f in fracs:
            os.environ["RT_TAIL_PATHS"] = f
            for k in range(3):
                dev.render_shard(dataclasses.replace(o, seed=1 + k), 0, n, buf.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 12
            for k in range(K):
                dev.render_shard(dataclasses.replace(o, seed=10 + k), 0, n, buf.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize(); res[f].append((time.perf_counter() - t0) / K * 1e3)
    print("shard 0 of %d alone: " % n + "  ".join("%s paths: %.3f ms (%s)" % (f, min(v), " ".join("%.3f" % x for x in v)) for f, v in res.items()), flush=True)
