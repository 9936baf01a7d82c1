// build.rs — compiles raytracer_amd/csrc with hipcc for gfx950 and links the result.
//
// Mirrors raytracer_amd/build.py (the recipe this repository uses and tests): one hipcc call per file, because
// a mixed `-x hip a.hip -x c++ b.cpp` command line loses --offload-arch; -ffp-contract=off because rustc never
// fuses a*b+c and the device code must round like the host (numerics).
//
// Environment: HIPCC (default /opt/rocm/bin/hipcc), RT_AMD_CSRC (default ../raytracer_amd/csrc).
use std::env;
use std::path::{Path, PathBuf};
use std::process::Command;

const SOURCES: [(&str, &str); 5] = [
    ("rt_kernels.hip", "hip"),
    ("rt_flatten.cpp", "c++"),
    ("host_scene.cpp", "c++"),
    ("host_jpeg.cpp", "c++"),
    ("rt_common.cpp", "c++"),
];

fn run(cmd: &mut Command) {
    let status = cmd.status().unwrap_or_else(|e| panic!("cannot run {:?}: {}", cmd, e));
    assert!(status.success(), "{:?} failed with {}", cmd, status);
}

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let csrc = env::var("RT_AMD_CSRC").unwrap_or_else(|_| "../raytracer_amd/csrc".to_string());
    let csrc = Path::new(&csrc);

    let mut objects = Vec::new();
    for (src, lang) in SOURCES.iter() {
        let obj = out.join(format!("{}.o", src));
        let mut c = Command::new(&hipcc);
        c.args(["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-math-errno"]);
        if *lang == "hip" {
            c.arg("--offload-arch=gfx950");
        }
        c.args(["-c", "-x", lang]).arg(csrc.join(src)).arg("-o").arg(&obj);
        run(&mut c);
        objects.push(obj);
        println!("cargo:rerun-if-changed={}", csrc.join(src).display());
    }
    // every file raytracer_amd/build.py lists under HEADERS (tests/test_host_logic.py compares the two lists)
    for header in ["rt_device.h", "rt_dev_math.h", "rt_dev_shapes.h", "rt_k_common.h", "rt_queue.h", "rt_k_material.h",
                   "rt_k_intersect.h", "rt_k_mesh.h", "rt_k_mesh_grid_round.inc", "rt_k_mesh_grid_step.inc", "rt_k_mesh_pool.h",
                   "rt_k_shade.h", "rt_k_tail.h",
                   "rt_flatten.h", "rt_shard.h", "host_math.h", "host_jpeg.h", "../../include/rt_api.h",
                   "../../include/rt_rng.h"] {
        println!("cargo:rerun-if-changed={}", csrc.join(header).display());
    }
    let lib = out.join("librt_amd.so");
    let mut link = Command::new(&hipcc);
    link.args(["--offload-arch=gfx950", "-shared", "-fPIC"]).args(&objects).args(["-lz", "-o"]).arg(&lib);
    run(&mut link);

    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    // the HIP runtime comes in through librt_amd.so's own DT_NEEDED entries
    println!("cargo:rerun-if-env-changed=HIPCC");
    println!("cargo:rerun-if-env-changed=RT_AMD_CSRC");
}
