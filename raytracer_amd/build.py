"""Build recipe for librt_amd.so: hipcc, gfx950 only, in-tree.

    python raytracer_amd/build.py            # build if sources are newer than the library
    python raytracer_amd/build.py --force    # (run by path: importing the package needs the library)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt_amd.so")
SOURCES = ["rt_kernels.hip", "rt_flatten.cpp", "host_scene.cpp", "host_jpeg.cpp", "rt_common.cpp"]
HEADERS = ["rt_device.h", "rt_dev_math.h", "rt_dev_shapes.h", "rt_flatten.h", "rt_shard.h", "host_math.h", "host_jpeg.h",
           "rt_k_common.h", "rt_queue.h", "rt_k_material.h", "rt_k_intersect.h", "rt_k_mesh.h", "rt_k_mesh_grid_round.inc", "rt_k_mesh_grid_step.inc", "rt_k_mesh_pool.h", "rt_k_shade.h", "rt_k_tail.h",
           "../../include/rt_api.h", "../../include/rt_rng.h"]

# -ffp-contract: "off" keeps a*b+c as two roundings like the reference's Rust
# code; "fast" lets the device fuse into v_fma_f64.  See DESIGN.md (numerics).
FP_CONTRACT = os.environ.get("RT_FP_CONTRACT", "off")


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS + ["../build.py"]:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def check_arch():
    """The library must carry a gfx950 code object and nothing else."""
    with open(LIB, "rb") as f:
        blob = f.read()
    if b"amdgcn-amd-amdhsa--gfx950" not in blob:
        raise RuntimeError("librt_amd.so holds no gfx950 code object")
    for other in (b"--gfx906", b"--gfx90a", b"--gfx942"):
        if other in blob:
            raise RuntimeError("librt_amd.so holds a code object for %s" % other.decode())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    common = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=" + FP_CONTRACT, "-fno-math-errno", "-Wall",
              "-Wno-unused-function"] + os.environ.get("RT_EXTRA_FLAGS", "").split()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        if src.endswith(".hip"):
            # one file per hipcc call: in a mixed `-x hip a.hip -x c++ b.cpp` command line hipcc
            # falls back to its default offload arch instead of the one asked for
            cmd = [hipcc()] + common + ["--offload-arch=gfx950", "-c", "-x", "hip", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        else:
            cmd = [hipcc()] + common + ["-c", "-x", "c++", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-lz", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    check_arch()
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
