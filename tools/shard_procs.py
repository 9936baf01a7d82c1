"""Shard 0 of N, F frames in flight per process, P processes on the same GPU (each with its own HIP runtime): is a
rank's frame rate bound by the GPU or by one process's launch path?  python tools/shard_procs.py N F P [K]"""
import os, subprocess, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    n, F, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sys.path.insert(0, ROOT)
    import dataclasses
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import raytracer_amd as rt
    from raytracer_amd import system
    o = rt.Options(width=3840, height=2160, samples=1, seed=1)
    scene = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/mesh.scene"), os.path.join(ROOT, "tests/assets"))
    devs = [rt.DeviceScene(scene) for _ in range(F)]
    streams = [torch.cuda.Stream() for _ in range(F)]
    bufs = [torch.zeros(system.shard_pixels(o, 0, n) * 3, dtype=torch.float64, device="cuda") for _ in range(F)]

    def frame(k):
        f = k % F
        return devs[f].render_shard(dataclasses.replace(o, seed=1 + k), 0, n, bufs[f].data_ptr(), None, streams[f].cuda_stream).rays

    with ThreadPoolExecutor(F) as pool:
        list(pool.map(lambda f: [frame(f), frame(f + F)], range(F)))
        torch.cuda.synchronize()
        # start together: wait for the go file
        while not os.path.exists(sys.argv[5]):
            time.sleep(0.001)
        t0 = time.perf_counter()
        list(pool.map(lambda f: sum(frame(k) for k in range(f, K, F)), range(F)))
        torch.cuda.synchronize()
        print("%.4f" % ((time.perf_counter() - t0) / K * 1e3), flush=True)
    sys.exit(0)
n, F, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 48
go = "/tmp/shard_procs_go_%d" % os.getpid()
procs = [subprocess.Popen([sys.executable, __file__, "worker", str(n), str(F), str(K), go], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(P)]
time.sleep(25.0)  # every worker has warmed up by then
open(go, "w").close()
ms = [float(p.communicate(timeout=200)[0].strip().splitlines()[-1]) for p in procs]
os.remove(go)
print("shard 0 of %d, %d processes x %d in flight: %.3f ms per frame and process, %.3f ms per frame overall" % (n, P, F, sum(ms) / P, sum(ms) / P / P))
