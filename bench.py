#!/usr/bin/env python3
"""bench.py — Mrays/s (primary + secondary) at 4K on mesh.scene, 1..N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one complete frame of BASELINE.json's headline workload
(tests/assets/scenes/mesh.scene, 3840x2160, samples=1): every rank renders its
interleaved 64x64 tiles with the HIP wavefront tracer (scene already resident
in HBM), rank 0 gathers the finished tiles over RCCL and scatters them into the
final image.  The frame's seed changes every step, so nothing can be cached.
value = rays traced by all ranks in the K timed steps / max-over-ranks time.

Extra objects on the JSON line (see DESIGN.md "Measurement"):
  roofline      the dominant kernel by GPU time (k_mesh, the grid walk of the rays queued for
                the mesh): algorithmic bytes per queued ray / its HIP-event time; the other
                kernels of the frame are listed under "kernels" the same way (k_intersect: nearest
                analytic hit + shading of the rays with no mesh to visit; k_shade: the mesh records)
  frame_hbm     SURVEY 8(d)'s whole-frame figure: (264 B/ray + 24 B/pixel-sample) / frame time
  cpu_baseline  the CPU oracle (a restatement of the reference, `kind: port`) on a
                bounded sample of the same scene; N=1, rank 0 only
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# One hardware queue per stream that has work at the same time: the frames in flight each own a stream, torch's
# stream carries the gather, and HIP's default of 4 queues per process (which the GPU box also exports) would make the
# fourth frame's stream share one (measured: 4 frames in flight 2.59 ms per frame on 4 queues, 2.31 on 8; 3 frames 2.42
# on either).  Read by the HIP runtime when it starts, so it is set before anything touches the GPU.  A larger value
# from the caller stays; BENCH_HW_QUEUES=<n> asks for exactly n (tools/flight_sweep.sh's 4-queue leg); the profiling
# scripts export the same 8 before rocprofv3, whose preloaded library may start the runtime before this line runs.
# The value in effect is recorded in the JSON line's config.
if os.environ.get("BENCH_HW_QUEUES"):
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ["BENCH_HW_QUEUES"]
elif int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, SAMPLES = 3840, 2160, 1
SCENE = "mesh"
BYTES_PER_RAY_INTERSECT = 80 + 80   # ray record read; scattered ray written (or a 132 B mesh record, or 24 B of frame buffer)
BYTES_PER_RAY_MESH = 48 + 12 + 52   # k_mesh: mesh record by position (origin + direction, recorded hit t + object) read; hit record written
BYTES_PER_RAY_MESH_ONE = 48 + 12 + 12  # k_mesh_one: the same read; t and triangle written (k_shade merges them into the hit)
BYTES_PER_RAY_SHADE = 132 + 80      # mesh record (ray + hit) read, scattered ray written (or 24 B of frame buffer updated)
BYTES_PER_RAY_SHADE_ONE = 132 + 12 + 80  # after k_mesh_one: its t and triangle read as well
BYTES_PER_RAY_TAIL = 80 + 24          # the one-launch tail: a path record read when it takes over; 24 B of frame buffer when the path ends (per segment: an upper bound)
BYTES_PER_RAY_FRAME = 2 * 80 + 2 * 52
BYTES_PER_PIXEL_SAMPLE = 24
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 78.6              # FP64 vector peak


def kernel_sources_sha():
    """Fingerprint of the device code (every file under raytracer_amd/csrc and include/)."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "raytracer_amd", "csrc"), os.path.join(ROOT, "include")):
        for name in sorted(os.listdir(d)):
            if name.endswith((".h", ".hip", ".cpp", ".inc")):
                with open(os.path.join(d, name), "rb") as f:
                    h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def host_cores():
    """Threads for the all-cores leg, and what the machine shows.  A one-GPU lease of the GPU box owns a share of the
    host (the harness exports it as OMP_NUM_THREADS = 16); os.cpu_count() shows the whole machine there."""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = visible
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or affinity
    return min(share, affinity), {"cpus_visible": visible, "affinity": affinity,
                                   "omp_num_threads": os.environ.get("OMP_NUM_THREADS")}


def native_oracle():
    """Rebuild the oracle here with -march=native (the reference's target-cpu=native); falls back to the shipped
    x86-64-v3 build when no compiler is at hand."""
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    odir = os.path.join(ROOT, "oracle")
    if shutil.which("g++") and shutil.which("make"):
        try:
            subprocess.check_call(["make", "-s", "-B", "-C", odir, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  timeout=240)
            oracle_lib.use_library(os.path.join(odir, "liboracle_native.so"))
            return oracle_lib, "g++ -O3 -march=native -flto, built on this host"
        except Exception as e:  # noqa: BLE001
            note = "native build failed (%s); " % type(e).__name__
    else:
        note = "no g++ on this host; "
    return oracle_lib, note + "shipped build, g++ -O3 -march=x86-64-v3 -flto"


def cpu_baseline(scene):
    """Time the oracle on the host cores on a bounded sample of the bench workload:
    the whole 4K frame row-parallel over this lease's cores (reported as cpu_baseline) and a
    1/16-size frame on one thread = the reference's own threading at samples=1."""
    oracle_lib, build_note = native_oracle()
    import raytracer_amd as rt
    path = os.path.join(ROOT, "tests", "assets", "scenes", scene + ".scene")
    assets = os.path.join(ROOT, "tests", "assets")
    cores, seen = host_cores()
    w, h = WIDTH, HEIGHT
    o = rt.Options(width=w, height=h, samples=SAMPLES, seed=1)
    sc = rt.parse_file(o, path, assets)
    t0 = time.perf_counter()
    _, _, st = oracle_lib.render(sc.desc, o.to_c(), threads=cores, want_ids=False)
    t_all = time.perf_counter() - t0
    w1, h1 = WIDTH // 4, HEIGHT // 4
    o1 = rt.Options(width=w1, height=h1, samples=SAMPLES, seed=1)
    sc1 = rt.parse_file(o1, path, assets)
    t0 = time.perf_counter()
    _, _, st1 = oracle_lib.render(sc1.desc, o1.to_c(), threads=1, want_ids=False)
    t_one = time.perf_counter() - t0
    return ({"value": st.rays / t_all / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "host": seen,
             "build": build_note,
             "sample": "%s.scene %dx%d samples=%d seed=1 (one whole bench frame, %d rays), oracle row-parallel over %d "
                       "threads = every core of this lease (the machine shows %d CPUs, the lease's share is %d), %.1f s"
                       % (scene, w, h, SAMPLES, st.rays, cores, seen["cpus_visible"], cores, t_all)},
            {"value": st1.rays / t_one / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
             "sample": "%s.scene %dx%d samples=%d seed=1, one thread (the reference parallelises over strata only), %.1f s"
                       % (scene, w1, h1, SAMPLES, t_one)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--samples", type=int, default=SAMPLES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="warmup and the timed region only (the command the rocprofv3 summaries under profiles/ are taken of)")
    ap.add_argument("--in-flight", type=int, default=4, help="independent frames a rank keeps in flight (1: one after the other)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo (with --same-device) rehearses the multi-rank path on a one-GPU box: shards are gathered through host memory")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--gather", default="rgb8", choices=["rgb8", "f64"],
                    help="what crosses the links in the exchange step of a multi-GPU frame: the shard tone-mapped where it was rendered "
                         "(main.rs:201-222 is per pixel: 3 bytes per pixel, SURVEY 8e) or its f64 sums (24 bytes per pixel)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import raytracer_amd as rt
    from raytracer_amd import system

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    opts = rt.Options(width=args.width, height=args.height, samples=args.samples, seed=1)
    scene = rt.parse_file(opts, os.path.join(ROOT, "tests", "assets", "scenes", args.scene + ".scene"),
                          os.path.join(ROOT, "tests", "assets"))
    # Frames in flight: consecutive steps are independent frames, so a rank keeps F of them going, each from its
    # own host thread on its own stream with its own resident copy of the scene and its own output buffer.  One
    # frame alone leaves the GPU idle through its late bounce passes (a few thousand rays each), its tail kernel
    # and the host's queue-length read-backs; the next frame's full passes fill those gaps.  Gathers stay in step
    # order on the main thread (a collective must be issued in the same order on every rank).
    import dataclasses
    from concurrent.futures import ThreadPoolExecutor
    F = max(1, args.in_flight)
    n_slots = [system.shard_pixels(opts, r, world) for r in range(world)]
    max_slots = max(n_slots)
    # the exchange step's buffers: f64 sums or tone-mapped bytes (--gather)
    gdtype = torch.uint8 if args.gather == "rgb8" else torch.float64
    full = torch.zeros(args.height * args.width * 3, dtype=gdtype, device=dev) if (rank == 0 and world > 1) else None
    gathered = [torch.zeros(max_slots * 3, dtype=gdtype, device=dev) for _ in range(world)] \
        if (rank == 0 and world > 1) else None
    sqrt_spp = int(args.samples ** 0.5)
    divisor = float(sqrt_spp * sqrt_spp + 1)  # the reference's own count: strata + 1 (main.rs:175-222)
    slots = []
    for f in range(F):  # scene upload and buffers: outside the timed region
        out = torch.zeros((max_slots if world > 1 else args.height * args.width) * 3, dtype=torch.float64, device=dev)
        out8 = torch.zeros(max_slots * 3, dtype=torch.uint8, device=dev) if (world > 1 and args.gather == "rgb8") else None
        slots.append({"scene": rt.DeviceScene(scene), "stream": torch.cuda.Stream(device=dev), "out": out, "out8": out8})
    main_stream = torch.cuda.current_stream().cuda_stream
    pool = ThreadPoolExecutor(F)

    def render(k, instrumented):
        slot = slots[k % F]
        o = dataclasses.replace(opts, seed=1 + k, stage_times=instrumented)
        torch.cuda.set_device(local_rank)
        # a single GPU writes the row-major image directly (RT_FLAG_ROW_MAJOR): nothing to reassemble
        st = slot["scene"].render_shard(o, rank, world, slot["out"].data_ptr(), None, slot["stream"].cuda_stream,
                                        row_major=(world == 1))
        if slot["out8"] is not None:
            # the output stage's pixel math on this rank's tiles, on the frame's own stream: the exchange then moves bytes
            system.tonemap_device(slot["out"].data_ptr(), n_slots[rank], divisor, slot["out8"].data_ptr(), slot["stream"].cuda_stream)
            slot["stream"].synchronize()
        return st

    def exchange(k):
        """the one exchange step of frame k: every rank's tiles to rank 0, scattered into the row-major image"""
        shard = slots[k % F]["out8"] if args.gather == "rgb8" else slots[k % F]["out"]
        if args.backend == "nccl":
            dist.gather(shard, gathered, dst=0)
        else:
            host = shard.cpu()
            parts = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, parts, dst=0)
            if rank == 0:
                for r in range(world):
                    gathered[r].copy_(parts[r])
        if rank == 0:
            for r in range(world):
                if args.gather == "rgb8":
                    system.unshard_rgb8(opts, r, world, gathered[r].data_ptr(), full.data_ptr(), stream=main_stream)
                else:
                    system.unshard(opts, r, world, gathered[r].data_ptr(), full.data_ptr(), stream=main_stream)
        torch.cuda.current_stream().synchronize()  # the slot's buffer is free for frame k + F

    def run(first, count, instrumented):
        """steps first .. first + count - 1, F of them in flight; returns their rt_stats in step order"""
        futs = {k: pool.submit(render, k, instrumented) for k in range(first, first + min(F, count))}
        stats = []
        for k in range(first, first + count):
            stats.append(futs.pop(k).result())
            if world > 1:
                exchange(k)
            if k + F < first + count:
                futs[k + F] = pool.submit(render, k + F, instrumented)
        return stats

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs with the per-kernel HIP events on (RT_FLAG_STAGE_TIMES): the roofline figure must
    # come from events on the launch stream inside it.  They cost about 3 % of the frame; `uninstrumented`
    # repeats the same steps without them.
    # every slot renders at least once before the clock starts (its workspace is allocated on its first frame)
    run(0, max(args.warmup, F), True)
    barrier()
    t0 = time.perf_counter()
    stats = run(args.warmup, args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    rays = sum(st.rays for st in stats)
    if args.timed_only:
        dt_plain = dt
    else:
        run(0, min(F, args.steps), False)
        barrier()
        t1 = time.perf_counter()
        run(args.warmup, args.steps, False)
        barrier()
        dt_plain = time.perf_counter() - t1
    # one frame at a time (no overlap): the latency of a frame, and per-kernel times that no other frame shares
    if F > 1 and not args.timed_only:
        F_saved, F = F, 1
        run(0, 1, True)
        barrier()
        t2 = time.perf_counter()
        alone = run(args.warmup, args.steps, True)
        barrier()
        dt_alone = time.perf_counter() - t2
        F = F_saved
    else:
        alone, dt_alone = stats, dt
    pool.shutdown()

    if world > 1:
        t = torch.tensor([dt, float(rays), dt_plain, dt_alone], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        dt_plain = float(tmax[2])
        dt_alone = float(tmax[3])
        rays_total = float(tsum[1])
    else:
        rays_total = float(rays)

    if rank == 0:
        value = rays_total / dt / 1e6
        ms_step = dt / args.steps * 1e3

        def line(kernel, rays_k, bytes_per_ray, ms, launches):
            ach = rays_k * bytes_per_ray / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_ray": bytes_per_ray, "rays_per_launch":
                    rays_k / launches if launches else None, "avg_launch_ms": ms / launches if launches else None,
                    "launches": launches, "ms_per_step": ms / args.steps}

        def kernel_lines(sts):
            """the frame's kernels from a leg's rt_stats (HIP events on the launch stream, summed per stage)"""
            a = dict(ms_int=sum(st.ms_intersect for st in sts), ms_mesh=sum(st.ms_mesh for st in sts),
                     ms_shade=sum(st.ms_shade for st in sts), ms_tail=sum(st.ms_tail for st in sts),
                     n_int=sum(st.n_intersect_launches for st in sts), n_mesh=sum(st.n_mesh_launches for st in sts),
                     int_rays=sum(st.intersect_rays for st in sts), mesh_rays=sum(st.mesh_rays for st in sts),
                     mesh_records=sum(st.mesh_records for st in sts), mesh_one_rays=sum(st.mesh_one_rays for st in sts),
                     rays=sum(st.rays for st in sts))
            # which mesh kernel walked the grids (rt_stats.mesh_one_rays): k_mesh_one writes 12 bytes per ray, k_mesh 52
            one = a["mesh_rays"] > 0 and a["mesh_one_rays"] == a["mesh_rays"]
            mesh_kernel, mesh_bytes = ("k_mesh_one", BYTES_PER_RAY_MESH_ONE) if one else ("k_mesh", BYTES_PER_RAY_MESH)
            tail_rays = a["rays"] - a["int_rays"]  # segments the one-launch tail traced (the passes counted the others)
            ks = [line("k_intersect", a["int_rays"], BYTES_PER_RAY_INTERSECT, a["ms_int"], a["n_int"]),
                  line(mesh_kernel, a["mesh_rays"], mesh_bytes, a["ms_mesh"], a["n_mesh"]),
                  line("k_shade", a["mesh_records"], BYTES_PER_RAY_SHADE_ONE if one else BYTES_PER_RAY_SHADE, a["ms_shade"], a["n_int"]),
                  line("k_tail", tail_rays, BYTES_PER_RAY_TAIL, a["ms_tail"], sum(st.strata for st in sts))]  # one tail per stratum
            return ks

        # The roofline object: the frame's dominant kernel by GPU time, from the leg that renders ONE FRAME AT A TIME,
        # where a launch has the GPU to itself and its HIP-event duration is its own (with frames in flight a launch
        # shares the GPU with the other frames' kernels and its event time is not evidence of anything).
        ks_alone = kernel_lines(alone)
        roof = dict(max(ks_alone, key=lambda k: k["ms_per_step"]))
        # HBM bytes per launch from the PMC passes over the same command one frame at a time (tools/pmc.sh,
        # tools/traffic.py).  The file records the kernel sources it was measured on; after any change to them the
        # figure is withheld (null) rather than quoted stale.
        traffic, traffic_note = None, "no PMC measurement committed"
        tj = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        fresh = False
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                fresh = tj.get("sources_sha256") == kernel_sources_sha()
                if fresh:
                    traffic = tj.get(roof["kernel"] + "_hbm_bytes_per_launch") or tj.get(roof["kernel"] + "_grid_hbm_bytes_per_launch")
                    traffic_note = "profiles/traffic.json, measured on these kernel sources"
                else:
                    traffic_note = "profiles/traffic.json was measured on other kernel sources: withheld"
            except Exception:
                traffic = None
        # The other side of the roofline (FP64 VALU, 78.6 TFLOP/s vector peak = 1024 SIMDs x one 64-lane
        # instruction per 4 cycles): wave-level VALU instructions per frame from the SQ pass of the same profile, over
        # the time of a frame alone.  Instructions, not flops: a bound from above on both counts.
        valu = None
        frame_alone_s = dt_alone / args.steps
        try:
            if fresh:
                per = {k[:-len("_valu_wave_instructions_per_frame")]: v for k, v in tj.items() if k.endswith("_valu_wave_instructions_per_frame")}
                if per and args.scene == SCENE and args.width == WIDTH and args.height == HEIGHT and args.samples == SAMPLES:
                    insts = sum(per.values()) / max(1, world)   # per GPU: the frame's rays are split over the ranks
                    lanes = sum(v * tj.get(k + "_valu_lanes_per_instruction", 0.0) for k, v in per.items()) / max(1, world)
                    peak_issue = VALU_PEAK_TFLOPS * 1e12 / 128.0  # wave instructions per second (an FMA on 64 lanes = 128 flop)
                    valu = {"bound": "fp64 valu issue", "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "wave_instructions_per_frame": per,
                            "issue_slots_used_frames_in_flight": insts / (dt / args.steps) / peak_issue,
                            "issue_slots_used_one_frame_at_a_time": insts / frame_alone_s / peak_issue,
                            "tflops_if_every_lane_instruction_were_an_fma": lanes * 2.0 / (dt / args.steps) / 1e12,
                            "frac": lanes * 2.0 / (dt / args.steps) / 1e12 / VALU_PEAK_TFLOPS,
                            "lanes_per_instruction": {k: tj.get(k + "_valu_lanes_per_instruction") for k in per},
                            "source": "profiles/traffic.json (SQ_INSTS_VALU, SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of the same command, one frame at a time)"}
                    dom = roof["kernel"]
                    if dom in per and roof["ms_per_step"] > 0:
                        roof["limited_by"] = ("fp64 valu issue: %.0f M wave instructions per frame in %.3f ms = %.0f %% of the chip's 614 G/s issue "
                                              "peak at %.1f of 64 lanes per instruction" % (
                                                  per[dom] / 1e6, roof["ms_per_step"], 100.0 * per[dom] / (roof["ms_per_step"] * 1e-3) / peak_issue,
                                                  tj.get(dom + "_valu_lanes_per_instruction", 0.0)))
        except Exception:
            valu = None
        roof["traffic"] = traffic
        roof["traffic_source"] = traffic_note
        roof["measured"] = ("HIP events on the launch stream inside the one-frame-at-a-time leg of this run (%d steps): a launch has the GPU to "
                            "itself there.  achieved = algorithmic_bytes_per_ray x rays_per_launch / avg_launch_ms; the same kernel's average in "
                            "profiles/r03g_alone_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py --in-flight 1; k_intersect is listed there as its two instantiations, camera rays and ray queues: total time over total calls) must agree" % args.steps)
        roof["kernel_note"] = ("the frame's dominant kernel by GPU time; k_intersect: nearest analytic hit and the shading of every ray with no "
                               "mesh to visit; the mesh stage is k_mesh_one in a scene with one mesh job (rt_k_mesh_pool.h), k_mesh otherwise")
        rays_alone = sum(st.rays for st in alone) * (world if world > 1 else 1)  # (rank 0's share times ranks: an estimate for N > 1)
        frame_bytes = rays_total * BYTES_PER_RAY_FRAME + args.width * args.height * args.samples * BYTES_PER_PIXEL_SAMPLE * args.steps
        frame_bytes_alone = rays_alone * BYTES_PER_RAY_FRAME + args.width * args.height * args.samples * BYTES_PER_PIXEL_SAMPLE * args.steps
        ks_flight = kernel_lines(stats)
        out = {
            "metric": "Mrays/sec (primary+secondary) at 4K, mesh.scene",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # BASELINE's metric also asks for the frame time: one frame at a time, first launch to finished frame
            "frame_time_ms": dt_alone / args.steps * 1e3,
            "config": {"workload": "%s.scene %dx%d samples=%d max_depth=50, seed=1+step" %
                                   (args.scene, args.width, args.height, args.samples),
                       "tile": 64, "sharding": "tile%d" % world, "gather": (args.gather if world > 1 else None), "rays_per_frame": rays_total / args.steps,
                       "frames_in_flight": F, "gpu_max_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)},
            "roofline": roof,
            "roofline_valu": valu,
            "kernels": [k for k in ks_alone if k["kernel"] != roof["kernel"]],
            # the same K steps again without the per-kernel events of the timed region (what rt_render costs a caller
            # that does not ask for stage times)
            "uninstrumented": {"value": rays_total / dt_plain / 1e6, "unit": "Mrays/s", "ms_per_step": dt_plain / args.steps * 1e3},
            # the same K steps one frame after the other: the latency of a frame; the roofline figures above come from this leg
            "one_frame_at_a_time": {"ms_per_step": dt_alone / args.steps * 1e3, "value": rays_alone / dt_alone / 1e6, "unit": "Mrays/s",
                                    "kernel_ms_per_step": {k["kernel"]: k["ms_per_step"] for k in ks_alone}},
            # HIP-event stage times of the timed region.  With frames in flight they overlap other frames' kernels and
            # sum to more than ms_per_step: kept for comparison between runs, not as durations
            "overlapped_stage_ms_per_step": {k["kernel"]: k["ms_per_step"] for k in ks_flight} if F > 1 else None,
            "frame_hbm": {"bytes_per_ray": BYTES_PER_RAY_FRAME, "bytes_per_pixel_sample": BYTES_PER_PIXEL_SAMPLE,
                          "achieved": frame_bytes / dt / 1e9 / world, "unit": "GB/s per GPU",
                          "frac": frame_bytes / dt / 1e9 / world / HBM_PEAK_GBS,
                          "one_frame_at_a_time_frac": frame_bytes_alone / dt_alone / 1e9 / world / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            allc, one = cpu_baseline(args.scene)
            out["cpu_baseline"] = allc
            out["cpu_baseline_reference_threading"] = one
            out["speedup_vs_cpu_all_cores"] = value / allc["value"]
            out["speedup_vs_cpu_reference_threading"] = value / one["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
