import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt, oracle_lib, parity
name = sys.argv[1]; W = int(sys.argv[2]); H = int(sys.argv[3]); depth = int(sys.argv[4]) if len(sys.argv) > 4 else 50
for md in range(1, depth + 1):
    o = rt.Options(width=W, height=H, samples=int(os.environ.get("SPP", "1")), seed=int(os.environ.get("SEED", "1")), max_depth=md)
    sc = rt.parse_file(o, name if name.endswith(".scene") else parity.scene_path(name), parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    bad = np.argwhere(np.any(np.abs(rgb - ref) > 1e-9 * np.maximum(1, np.abs(ref)), axis=-1))
    print("max_depth", md, "rays", st.rays, rst.rays, "bad px", len(bad), [tuple(b) for b in bad[:6]], flush=True)
    if len(bad) or st.rays != rst.rays:
        for y, x in bad[:4]:
            print("   px", (y, x), "gpu", rgb[y, x], "ref", ref[y, x], "ids", ids[y, x], rids[y, x])
        break
