"""rt_render (the boundary that hands the frame over in a host buffer: f64 RGB sums + ids, 28 bytes per pixel over
PCIe) timed on the bench workload, beside bench.py's device-resident figure.  python tools/host_boundary_rate.py [--steps 10]"""
import argparse, dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as rt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--scene", default="mesh")
    a = ap.parse_args()
    o = rt.Options(width=a.width, height=a.height, samples=1, seed=1)
    sc = rt.parse_file(o, os.path.join(ROOT, "tests", "assets", "scenes", a.scene + ".scene"), os.path.join(ROOT, "tests", "assets"))
    ds = rt.DeviceScene(sc)
    for want_ids in (True, False):
        ds.render(o, want_ids=want_ids)  # warm: workspace, output buffers
        rays = 0
        t0 = time.perf_counter()
        for k in range(a.steps):
            res = ds.render(dataclasses.replace(o, seed=1 + k), want_ids=want_ids)
            rays += res[2].rays
        dt = time.perf_counter() - t0
        print("rt_render to host memory, ids %s: %.2f ms per frame, %.0f Mrays/s (%d frames)" % (
            "with" if want_ids else "without", dt / a.steps * 1e3, rays / dt / 1e6, a.steps))
    # the same entry point with the caller's buffers pinned and reused (what a host that renders many frames would do)
    import ctypes as C
    import torch
    from raytracer_amd import _capi
    rgb = torch.zeros((a.height, a.width, 3), dtype=torch.float64).pin_memory()
    ids = torch.zeros((a.height, a.width), dtype=torch.int32).pin_memory()
    stats = _capi.rt_stats()

    def one(k):
        oc = dataclasses.replace(o, seed=1 + k).to_c()
        rc = ds._lib.rt_render(ds._h, C.byref(oc), C.cast(rgb.data_ptr(), C.POINTER(C.c_double)),
                               C.cast(ids.data_ptr(), C.POINTER(C.c_int32)), C.byref(stats), _capi.rt_progress_cb(), None)
        assert rc == 0, rc
        return stats.rays

    one(0)
    rays = 0
    t0 = time.perf_counter()
    for k in range(a.steps):
        rays += one(k)
    dt = time.perf_counter() - t0
    print("rt_render to pinned host memory, reused, ids with: %.2f ms per frame, %.0f Mrays/s (%d frames)" % (
        dt / a.steps * 1e3, rays / dt / 1e6, a.steps))
    # the frame as an image (rt_render_rgb8: tone-mapped on the GPU, 3 bytes per pixel over PCIe), into pinned memory
    img = torch.zeros((a.height, a.width, 3), dtype=torch.uint8).pin_memory()

    def one8(k):
        oc = dataclasses.replace(o, seed=1 + k).to_c()
        rc = ds._lib.rt_render_rgb8(ds._h, C.byref(oc), C.cast(img.data_ptr(), C.POINTER(C.c_uint8)), None, C.byref(stats), None)
        assert rc == 0, rc
        return stats.rays

    one8(0)
    rays = 0
    t0 = time.perf_counter()
    for k in range(a.steps):
        rays += one8(k)
    dt = time.perf_counter() - t0
    print("rt_render_rgb8 to pinned host memory, reused: %.2f ms per frame, %.0f Mrays/s (%d frames)" % (
        dt / a.steps * 1e3, rays / dt / 1e6, a.steps))
    # the Python wrapper with its own page-locked buffers (DeviceScene.render(reuse=True), render_rgb8(reuse=True))
    for name, call in (("DeviceScene.render(reuse=True), ids with", lambda k: ds.render(dataclasses.replace(o, seed=1 + k), reuse=True)),
                       ("DeviceScene.render(reuse=True), ids without", lambda k: ds.render(dataclasses.replace(o, seed=1 + k), want_ids=False, reuse=True)),
                       ("DeviceScene.render_rgb8(reuse=True)", lambda k: ds.render_rgb8(dataclasses.replace(o, seed=1 + k), reuse=True)),
                       ("DeviceScene.render_rgb8()", lambda k: ds.render_rgb8(dataclasses.replace(o, seed=1 + k)))):
        call(0)
        rays = 0
        t0 = time.perf_counter()
        for k in range(a.steps):
            rays += call(k)[2].rays
        dt = time.perf_counter() - t0
        print("%s: %.2f ms per frame, %.0f Mrays/s (%d frames)" % (name, dt / a.steps * 1e3, rays / dt / 1e6, a.steps))


if __name__ == "__main__":
    main()
