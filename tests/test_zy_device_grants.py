"""What the device's runtime grants the library (collected late on purpose: a refusal here costs speed, not correctness,
and the round's end runs pytest -x)."""
import os

import pytest

import parity

pytestmark = pytest.mark.gpu


def test_lds_interval_lists_are_granted_on_this_device():
    """The CSG / fog kernels with their interval lists in LDS need more than the 64 KB a workgroup gets by default
    (hipFuncAttributeMaxDynamicSharedMemorySize; allow_large_lds, rt_kernels.hip).  A refusal is not an error - the
    scene renders the same frame with the lists in scratch memory, a sixth slower - so nothing else notices one: the
    runtime of some boxes refuses the device's limit itself for a kernel that also has static LDS, which round 3's
    first builds asked for.  The library's trace (RT_DEBUG=1) says what happened."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
            "import raytracer_amd as rt, parity\n"
            "for name in ('csg', 'room', 'fog_light'):\n"
            "    o = rt.Options(width=128, height=96, samples=1, seed=3)\n"
            "    rt.DeviceScene(rt.parse_file(o, parity.scene_path(name), parity.ASSETS)).render(o)\n" % (parity.ROOT, parity.ROOT))
    env = dict(os.environ, RT_DEBUG="1")
    env.pop("RT_LDS_LISTS", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("-> in LDS") == 3, r.stderr[-2000:]
    assert "refused" not in r.stderr and "allowed" in r.stderr, r.stderr[-2000:]
