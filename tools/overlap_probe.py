"""Does running K sub-frames concurrently (K streams, K host threads) beat one frame after the other?
python tools/overlap_probe.py [K]   (an experiment for DESIGN.md, not part of the product path)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import raytracer_amd as rt
from raytracer_amd import system
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
o = rt.Options(width=3840, height=2160, samples=1, seed=1)
scene = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/mesh.scene"), os.path.join(ROOT, "tests/assets"))
devs = [rt.DeviceScene(scene) for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]
bufs = [torch.zeros(system.shard_pixels(o, r, K) * 3, dtype=torch.float64, device="cuda") for r in range(K)]
one = rt.DeviceScene(scene)
buf1 = torch.zeros(system.shard_pixels(o, 0, 1) * 3, dtype=torch.float64, device="cuda")
def whole(seed):
    o.seed = seed
    return one.render_shard(o, 0, 1, buf1.data_ptr(), None, torch.cuda.current_stream().cuda_stream).rays
def split(seed):
    rays = [0] * K
    def work(r):
        oo = rt.Options(width=3840, height=2160, samples=1, seed=seed)
        rays[r] = devs[r].render_shard(oo, r, K, bufs[r].data_ptr(), None, streams[r].cuda_stream).rays
    th = [threading.Thread(target=work, args=(r,)) for r in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    return sum(rays)
for f, name in ((whole, "one frame, one stream"), (split, "%d sub-frames, %d streams" % (K, K))):
    for s in range(2): f(100 + s)
    torch.cuda.synchronize(); t0 = time.perf_counter(); rays = 0
    for s in range(8): rays += f(s + 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s %.3f ms/frame  %.0f Mrays/s" % (name, dt / 8 * 1e3, rays / dt / 1e6), flush=True)
