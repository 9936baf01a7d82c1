"""Render time of one shard of N on one GPU (what each rank of an N-GPU job does): python tools/shard_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import system
o = rt.Options(width=3840, height=2160, samples=1, seed=1, stage_times=True)
scene = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/mesh.scene"), os.path.join(ROOT, "tests/assets"))
dev = rt.DeviceScene(scene)
st = torch.cuda.current_stream().cuda_stream
for n in (1, 2, 4, 8):
    buf = torch.zeros(system.shard_pixels(o, 0, n) * 3, dtype=torch.float64, device="cuda")
    for k in range(3):
        o.seed = 50 + k
        dev.render_shard(o, 0, n, buf.data_ptr(), None, st)
    torch.cuda.synchronize(); t0 = time.perf_counter(); rays = 0; stages = [0, 0, 0, 0]
    for k in range(10):
        o.seed = 1 + k
        s = dev.render_shard(o, 0, n, buf.data_ptr(), None, st)
        rays += s.rays; stages = [a + b for a, b in zip(stages, (s.ms_intersect, s.ms_mesh, s.ms_shade, s.ms_tail))]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("shard 0 of %d: %.3f ms  (%.0f Mrays/s on this GPU; x%d = %.0f if every rank took as long)  stages ms %s" %
          (n, dt * 1e3, rays / 10 / dt / 1e6, n, rays / 10 / dt / 1e6 * n, [round(x / 10, 2) for x in stages]), flush=True)
