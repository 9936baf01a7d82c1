"""Render every scene on the GPU and with the oracle and print the parity metrics."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt
import oracle_lib, parity

W = int(sys.argv[1]) if len(sys.argv) > 1 else 160
H = int(sys.argv[2]) if len(sys.argv) > 2 else 120
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
names = sys.argv[4].split(",") if len(sys.argv) > 4 else parity.SCENES
for name in names:
    o = rt.Options(width=W, height=H, samples=spp)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    t0 = time.time()
    dev = rt.DeviceScene(sc)
    rgb, ids, st = dev.render(o)
    t1 = time.time()
    rgb2, ids2, st2 = dev.render(o)
    t2 = time.time()
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    t3 = time.time()
    m = parity.compare(rgb, ids, ref, rids, st.strata)
    print("%-16s ids %.6f err255 %.5f close %.6f maxabs %.3g | rays gpu %d ref %d | guards %s/%s ovf %d | gpu %.1f ms (dev %.2f ms) cpu %.0f ms | rerun-identical %s" % (
        name, m["id_match"], m["mean_err_255"], m["px_close"], m["max_abs"], st.rays, rst.rays, list(st.guards), list(rst.guards),
        st.interval_overflow, (t2 - t1) * 1e3, st2.ms_device, (t3 - t2) * 1e3, bool(np.array_equal(rgb, rgb2))), flush=True)
