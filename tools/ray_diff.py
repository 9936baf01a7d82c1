import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt, oracle_lib, parity
from test_gpu_parity import _random_rays
name = sys.argv[1]; center = [float(x) for x in sys.argv[2].split(",")]; radius = float(sys.argv[3])
o = rt.Options(width=16, height=16)
sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
ro, rd = _random_rays(20000, 11, np.array(center), radius)
dev = rt.DeviceScene(sc)
got, gst = dev.trace_rays(ro, rd, seed=4)
ref, rst = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
bad = np.nonzero(got["object"] != ref["object"])[0]
print("mismatches", len(bad), "guards", list(gst.guards), list(rst.guards))
for i in bad[:12]:
    print(i, "o", ro[i], "d", rd[i], "| gpu obj %d t %.9g | ref obj %d t %.9g n %s" % (got["object"][i], got["t"][i], ref["object"][i], ref["t"][i], ref["n"][i]))
hit = (got["object"] == ref["object"]) & (ref["object"] >= 0)
dt = np.abs(got["t"][hit] - ref["t"][hit]); print("max dt", dt.max(), "max dn", np.abs(got["n"][hit]-ref["n"][hit]).max(), "max duv", np.abs(got["uv"][hit]-ref["uv"][hit]).max())
