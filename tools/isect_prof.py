"""Where k_intersect's waves spend their time (clock64 marks, summed over all waves of one 4K mesh.scene frame).
The marks are not in the product: apply tools/isect_prof.patch, build a variant, run, drop the patch again:

    git apply tools/isect_prof.patch && tools/build_variant.sh iprof "-DRT_ISECT_PROF=1" && git checkout raytracer_amd/csrc
    gpurun -- 'RT_AMD_LIB=$PWD/variants/librt_iprof.so python tools/isect_prof.py'

Marks 5-7 and 9-11 sit in code that k_shade / k_tail share (shade_one, intersect_analytic) and count their waves too."""
import ctypes as C
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raytracer_amd as rt  # noqa: E402

o = rt.Options(width=3840, height=2160, samples=1, seed=1)
sc = rt.parse_file(o, os.path.join(ROOT, "tests/assets/scenes/mesh.scene"), os.path.join(ROOT, "tests/assets"))
ds = rt.DeviceScene(sc)
if not hasattr(ds._lib, "rt_debug_prof"):
    raise SystemExit("this library was built without -DRT_ISECT_PROF=1 (see the docstring)")
out = torch.zeros(3840 * 2160 * 3, dtype=torch.float64, device="cuda")
buf = (C.c_ulonglong * 16)()
ds.render_shard(o, 0, 1, out.data_ptr(), None, None, row_major=True)  # warm
ds._lib.rt_debug_prof(buf)  # reads and clears
st = ds.render_shard(dataclasses.replace(o, seed=2), 0, 1, out.data_ptr(), None, None, row_major=True)
ds._lib.rt_debug_prof(buf)
v = list(buf)
names = {0: "fetch / primary ray", 1: "intersect_analytic", 9: "  object loop (all kernels)", 10: "  mesh wanted (all kernels)",
         11: "  normal + uv (all kernels)", 8: "(gap before shade)", 2: "shade_one (k_intersect)", 3: "add to the pixel",
         12: "append: ballots, atomics, stages + flushes", 4: "stores of rays appended from registers", 5: "  uniform_sphere (all kernels)", 6: "  normalize (all kernels)", 7: "  texture (all kernels)"}
tot = sum(v[k] for k in (0, 1, 8, 2, 3, 12, 4))
print("rays", st.rays)
for k in (0, 1, 9, 10, 11, 8, 2, 5, 6, 7, 3, 12, 4):
    print("%-32s %14d cycles  %5.1f %% of k_intersect's loop" % (names[k], v[k], 100.0 * v[k] / tot))
