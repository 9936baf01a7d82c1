"""Shard 0 of N rendered with F frames in flight (F host threads, each with its own scene copy and stream) on one
GPU: what a rank of an N-GPU job gains from overlapping consecutive frames.  python tools/shard_overlap.py"""
import dataclasses, os, sys, time
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("SHARD_QUEUES", "8")  # one hardware queue per stream in use (before the HIP runtime starts)
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import system
o = rt.Options(width=3840, height=2160, samples=1, seed=1)
scene = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/mesh.scene"), os.path.join(ROOT, "tests/assets"))
for n in [int(v) for v in os.environ.get("SHARD_N", "1,2,4,8").split(",")]:
    for F in [int(v) for v in os.environ.get("SHARD_F", "1,3,4").split(",")]:
        devs = [rt.DeviceScene(scene) for _ in range(F)]
        streams = [torch.cuda.Stream() for _ in range(F)]
        bufs = [torch.zeros(system.shard_pixels(o, 0, n) * 3, dtype=torch.float64, device="cuda") for _ in range(F)]
        def frame(k):
            f = k % F
            ok = dataclasses.replace(o, seed=1 + k)
            return devs[f].render_shard(ok, 0, n, bufs[f].data_ptr(), None, streams[f].cuda_stream).rays
        with ThreadPoolExecutor(F) as pool:
            list(pool.map(lambda f: [frame(f), frame(f + F)], range(F)))  # warm every slot (a scene serves one thread at a time)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            K = 24
            # each slot renders its frames in order; frames of different slots overlap
            def run_slot(f):
                return sum(frame(k) for k in range(f, K, F))
            rays = sum(pool.map(run_slot, range(F)))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print("shard 0 of %d, %d in flight: %.3f ms per frame, %.0f Mrays/s on this GPU" % (n, F, dt * 1e3, rays / K / dt / 1e6), flush=True)
