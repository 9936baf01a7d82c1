import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt, parity
o = rt.Options(width=16, height=16)
sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
dev = rt.DeviceScene(sc)
rng = np.random.default_rng(5)
for n in (64, 64, 4096, 262144):
    ro = np.array([0, 1.0, 0]) + rng.normal(size=(n, 3)) * 2.5
    tg = np.array([0, 1.0, 0]) + rng.normal(size=(n, 3)) * 0.6
    rd = tg - ro; rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    t0 = time.perf_counter(); got, st = dev.trace_rays(ro, rd); t1 = time.perf_counter()
    print(n, "rays: hits", int((got["object"] == 2).sum()), "wall %.3f ms" % ((t1 - t0) * 1e3), flush=True)
