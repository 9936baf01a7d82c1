"""Render driver: mirror of reference src/system.rs (`Options`, `RenderProgress`,
`render`) over librt_amd.so, plus the per-GPU entry points used for tile
sharding across ranks."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi


ROW_MAJOR = 1  # RT_FLAG_ROW_MAJOR
STAGE_TIMES = 2  # RT_FLAG_STAGE_TIMES: per-kernel times in rt_stats (costs about 3 % of a frame)


@dataclass
class Options:
    """system.rs:18-26; defaults are main.rs:50-88.  `seed` keys the
    counter-based random numbers that replace ThreadRng (include/rt_rng.h)."""
    num_threads: int = 1  # kept for interface parity; the GPU path ignores it
    width: int = 1024
    height: int = 768
    bias: float = 1e-4
    max_depth: int = 50
    samples: int = 1
    seed: int = 1
    tile_size: int = 64
    stage_times: bool = False  # RT_FLAG_STAGE_TIMES: per-kernel times in rt_stats (about 3 % of a 4K frame)

    def to_c(self, shard_rank=0, shard_count=1, flags=0):
        if self.stage_times:
            flags |= STAGE_TIMES
        return _capi.rt_options(width=self.width, height=self.height, samples=self.samples, max_depth=self.max_depth,
                                bias=self.bias, seed=self.seed, tile_size=self.tile_size, shard_rank=shard_rank,
                                shard_count=shard_count, flags=flags)




class RenderProgress:
    """system.rs:223-227.  `renderbuf` is the running SUM over finished strata,
    float64 [H, W, 3]."""

    def render_started(self, options):
        pass

    def sample_finished(self, options, renderbuf):
        pass

    def render_finished(self, options, renderbuf):
        pass


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.rt_last_error().decode("utf-8", "replace")))


class DeviceScene:
    """A scene resident on the current HIP device (rt_scene_create)."""

    def __init__(self, scene):
        from . import lib
        self._lib = lib
        self._h = C.c_void_p()
        self._pins = {}  # page-locked output buffers kept for render(reuse=True): key -> (pointer, array)
        _check(lib, lib.rt_scene_create(scene.desc, C.byref(self._h)), "rt_scene_create")

    def close(self):
        if self._h:
            self._lib.rt_scene_destroy(self._h)
            self._h = C.c_void_p()
        for p, _ in self._pins.values():
            self._lib.rt_pinned_free(p)
        self._pins = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pinned(self, key, shape, dtype):
        """A page-locked numpy array this DeviceScene owns and reuses (rt_pinned_alloc): the copy out of the GPU then
        runs at the link's speed.  Its contents are replaced by the next render that asks for the same key."""
        shape = tuple(int(v) for v in shape)
        have = self._pins.get(key)
        if have is not None and have[1].shape == shape and have[1].dtype == np.dtype(dtype):
            return have[1]
        if have is not None:
            self._lib.rt_pinned_free(have[0])
            del self._pins[key]
        n_bytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        _check(self._lib, self._lib.rt_pinned_alloc(n_bytes, C.byref(p)), "rt_pinned_alloc")
        arr = np.frombuffer((C.c_uint8 * n_bytes).from_address(p.value), dtype=dtype).reshape(shape)
        self._pins[key] = (p, arr)
        return arr

    def render(self, options, progress=None, want_ids=True, reuse=False):
        """Whole image on this GPU -> (rgb_sum [H,W,3] f64, ids [H,W] i32 | None, rt_stats).
        reuse=True: the arrays are page-locked buffers this DeviceScene keeps and hands out again on its next
        render(reuse=True) - a 4K frame's 232 MB then cross PCIe in 4 ms instead of 20 (copy what must outlive
        the next call)."""
        o = options.to_c()
        H, W = options.height, options.width
        if reuse:
            rgb = self._pinned("rgb", (H, W, 3), np.float64)
            ids = self._pinned("ids", (H, W), np.int32) if want_ids else None
        else:
            rgb = np.zeros((H, W, 3), dtype=np.float64)
            ids = np.zeros((H, W), dtype=np.int32) if want_ids else None
        stats = _capi.rt_stats()

        def _cb(user, done, total, ptr):
            buf = np.ctypeslib.as_array(ptr, shape=(H, W, 3))
            progress.sample_finished(options, buf)
            return 0

        # no progress object, no callback: the frame then stays on the GPU until it is done
        cb = _capi.rt_progress_cb(_cb) if progress is not None else _capi.rt_progress_cb()
        if progress is not None:
            progress.render_started(options)
        rc = self._lib.rt_render(self._h, C.byref(o), rgb.ctypes.data_as(C.POINTER(C.c_double)),
                                 ids.ctypes.data_as(C.POINTER(C.c_int32)) if want_ids else None, C.byref(stats), cb, None)
        _check(self._lib, rc, "rt_render")
        if progress is not None:
            progress.render_finished(options, rgb)
        return rgb, ids, stats

    def render_rgb8(self, options, progress=None, want_ids=False, reuse=False):
        """system::render + write_render_result_to_file's pixel math (main.rs:201-222) -> (rgb8 [H,W,3] u8 tone-mapped
        on the GPU by strata + 1, ids | None, rt_stats): 3 bytes per pixel leave the GPU instead of 24.
        `progress`: a DeviceProgress (row-major RGB8 snapshots); reuse as in render()."""
        o = options.to_c()
        H, W = options.height, options.width
        if reuse:
            img = self._pinned("rgb8", (H, W, 3), np.uint8)
            ids = self._pinned("ids", (H, W), np.int32) if want_ids else None
        else:
            img = np.zeros((H, W, 3), dtype=np.uint8)
            ids = np.zeros((H, W), dtype=np.int32) if want_ids else None
        stats = _capi.rt_stats()
        p = progress.to_c() if progress is not None else None
        rc = self._lib.rt_render_rgb8(self._h, C.byref(o), img.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      ids.ctypes.data_as(C.POINTER(C.c_int32)) if want_ids else None, C.byref(stats),
                                      C.byref(p) if p is not None else None)
        _check(self._lib, rc, "rt_render_rgb8")
        return img, ids, stats

    def render_shard(self, options, shard_rank, shard_count, d_rgb_ptr, d_ids_ptr=None, stream=None, row_major=False,
                     stage_times=False, progress=None):
        """One shard of the tiles into caller-owned device buffers (rt_render_device).  `row_major`
        (a single shard only): the buffers are width * height elements indexed y * width + x.
        `stage_times`: fill the per-kernel times of the returned rt_stats.  `progress`: a DeviceProgress
        (rt_render_device_progress)."""
        o = options.to_c(shard_rank, shard_count, (ROW_MAJOR if row_major else 0) | (STAGE_TIMES if stage_times else 0))
        stats = _capi.rt_stats()
        if progress is None:
            rc = self._lib.rt_render_device(self._h, C.byref(o), C.c_void_p(d_rgb_ptr),
                                            C.c_void_p(d_ids_ptr) if d_ids_ptr else None, C.byref(stats),
                                            C.c_void_p(stream) if stream else None)
        else:
            p = progress.to_c()
            rc = self._lib.rt_render_device_progress(self._h, C.byref(o), C.c_void_p(d_rgb_ptr),
                                                     C.c_void_p(d_ids_ptr) if d_ids_ptr else None, C.byref(stats),
                                                     C.c_void_p(stream) if stream else None, C.byref(p))
        _check(self._lib, rc, "rt_render_device")
        return stats

    def trace_rays(self, origins, directions, seed=1):
        """Ray::trace for a batch of rays -> structured array (t, n, uv, object), rt_stats."""
        o = np.ascontiguousarray(origins, dtype=np.float64)
        d = np.ascontiguousarray(directions, dtype=np.float64)
        n = o.shape[0]
        hits = (_capi.rt_hit * max(n, 1))()
        stats = _capi.rt_stats()
        rc = self._lib.rt_trace_rays(self._h, seed, n, o.ctypes.data_as(C.POINTER(C.c_double)),
                                     d.ctypes.data_as(C.POINTER(C.c_double)), hits, C.byref(stats))
        _check(self._lib, rc, "rt_trace_rays")
        dt = np.dtype([("t", "f8"), ("n", "f8", 3), ("uv", "f8", 2), ("object", "i4"), ("pad", "i4")])
        return np.frombuffer(hits, dtype=dt, count=n).copy(), stats


class DeviceProgress:
    """rt_progress: RenderProgress for frames that stay on the device.  `on_stratum(done, total)` after every
    stratum; `on_snapshot(done, total, rgb8)` at most every `min_interval` seconds (5 in main.rs:178) and not
    after the last stratum, with the running sum tone-mapped by done + 1 as write_render_result_to_file does
    (main.rs:174-185,219-222): rgb8 is a uint8 array [n_pixels, 3] in the layout of the render's outputs.
    A callback that returns a true value stops the render."""

    def __init__(self, on_stratum=None, on_snapshot=None, min_interval=5.0):
        self.on_stratum, self.on_snapshot, self.min_interval = on_stratum, on_snapshot, min_interval
        self._keep = None

    def to_c(self):
        p = _capi.rt_progress()
        keep = []
        if self.on_stratum is not None:
            f = _capi.rt_stratum_cb(lambda user, done, total: 1 if self.on_stratum(done, total) else 0)
            p.on_stratum = f
            keep.append(f)
        if self.on_snapshot is not None:
            def _snap(user, done, total, ptr, n):
                img = np.ctypeslib.as_array(ptr, shape=(int(n), 3))
                return 1 if self.on_snapshot(done, total, img) else 0
            f = _capi.rt_snapshot_cb(_snap)
            p.on_snapshot = f
            keep.append(f)
        p.min_interval_ms = int(round(self.min_interval * 1000))
        self._keep = keep  # the ctypes thunks must outlive the call
        return p


class FileRenderProgress(DeviceProgress):
    """CliRenderProgress (main.rs:133-199) without the progress bar: counts strata, rewrites `filename` (PNG)
    from a snapshot at most every 5 s while the render runs; `finish(rgb_sum, strata)` writes the final image
    the way render_finished does (divisor strata + 1, the reference's own count)."""

    def __init__(self, filename, width, height, min_interval=5.0):
        super().__init__(self._count, self._snapshot, min_interval)
        self.filename, self.width, self.height = str(filename), width, height
        self.num_samples = 0
        self.files_written = 0

    def _count(self, done, total):
        self.num_samples = done
        return 0

    def _snapshot(self, done, total, rgb8):
        write_png(self.filename, rgb8.reshape(self.height, self.width, 3))
        self.files_written += 1
        return 0

    def finish(self, rgb_sum, strata):
        write_png(self.filename, tonemap(rgb_sum, strata + 1))
        self.files_written += 1


def render_multi(options, device_scenes, progress=None, want_ids=True, rgb8=False):
    """system::render over several GPUs of this process (rt_render_multi): device_scenes[k] is the scene
    created with GPU k current -> (rgb_sum [H,W,3], ids [H,W] | None, rt_stats).  `progress`: DeviceProgress;
    its snapshots are row-major images.  rgb8=True (rt_render_multi_rgb8): the gathered frame is tone-mapped on
    the first GPU by strata + 1 and comes back as uint8 [H,W,3]."""
    from . import lib
    o = options.to_c()
    H, W = options.height, options.width
    rgb = np.zeros((H, W, 3), dtype=np.uint8 if rgb8 else np.float64)
    ids = np.zeros((H, W), dtype=np.int32) if want_ids else None
    stats = _capi.rt_stats()
    handles = (C.c_void_p * len(device_scenes))(*[d._h for d in device_scenes])
    p = progress.to_c() if progress is not None else None
    fn, ctype = (lib.rt_render_multi_rgb8, C.c_uint8) if rgb8 else (lib.rt_render_multi, C.c_double)
    rc = fn(handles, len(device_scenes), C.byref(o), rgb.ctypes.data_as(C.POINTER(ctype)),
            ids.ctypes.data_as(C.POINTER(C.c_int32)) if want_ids else None, C.byref(stats),
            C.byref(p) if p is not None else None)
    _check(lib, rc, "rt_render_multi")
    return rgb, ids, stats


def shard_pixels(options, shard_rank, shard_count):
    from . import lib
    o = options.to_c(shard_rank, shard_count)
    return int(lib.rt_shard_pixels(C.byref(o)))


def shard_map(options, shard_rank, shard_count):
    """pixel index (y * width + x) of every slot of a shard; -1 marks padding."""
    from . import lib
    o = options.to_c(shard_rank, shard_count)
    n = int(lib.rt_shard_pixels(C.byref(o)))
    m = np.zeros(n, dtype=np.uint32)
    _check(lib, lib.rt_shard_map(C.byref(o), m.ctypes.data_as(C.POINTER(C.c_uint32))), "rt_shard_map")
    out = m.astype(np.int64)
    out[m == 0xFFFFFFFF] = -1
    return out


def gather_frame(options, shard, rank, world, full=None, maps=None):
    """The one exchange step of a sharded frame: every rank hands its finished tiles
    (`shard`: torch tensor [n_slots_max * 3], padded to the largest shard) to rank 0,
    which scatters them into the row-major image `full` [H * W * 3].  RCCL on GPUs
    (`nccl` backend), gloo in the CPU tests.  Returns `full` on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world > 1:
        bufs = [torch.empty_like(shard) for _ in range(world)] if rank == 0 else None
        dist.gather(shard, bufs, dst=0)
    else:
        bufs = [shard]
    if rank != 0:
        return None
    if full is None:
        full = torch.zeros(options.height * options.width * 3, dtype=shard.dtype, device=shard.device)
    img = full.view(-1, 3)
    for r in range(world):
        m = maps[r] if maps is not None else torch.from_numpy(shard_map(options, r, world)).to(shard.device)
        valid = m >= 0
        img[m[valid]] = bufs[r].view(-1, 3)[: m.numel()][valid]
    return full


def unshard(options, shard_rank, shard_count, d_shard_rgb, d_full_rgb, d_shard_ids=None, d_full_ids=None, stream=None):
    from . import lib
    o = options.to_c(shard_rank, shard_count)
    rc = lib.rt_unshard_device(C.byref(o), shard_rank, C.c_void_p(d_shard_rgb), C.c_void_p(d_full_rgb),
                               C.c_void_p(d_shard_ids) if d_shard_ids else None,
                               C.c_void_p(d_full_ids) if d_full_ids else None, C.c_void_p(stream) if stream else None)
    _check(lib, rc, "rt_unshard_device")


def unshard_rgb8(options, shard_rank, shard_count, d_shard_rgb8, d_full_rgb8, stream=None):
    """The same for a shard tone-mapped where it was rendered (rt_unshard_rgb8_device): 3 bytes per slot."""
    from . import lib
    o = options.to_c(shard_rank, shard_count)
    rc = lib.rt_unshard_rgb8_device(C.byref(o), shard_rank, C.c_void_p(d_shard_rgb8), C.c_void_p(d_full_rgb8),
                                    C.c_void_p(stream) if stream else None)
    _check(lib, rc, "rt_unshard_rgb8_device")


def render(options, scene, progress=None):
    """system::render(options, scene, &mut progress) (system.rs:263-307) on one GPU."""
    dev = DeviceScene(scene)
    try:
        return dev.render(options, progress)
    finally:
        dev.close()


def tonemap(rgb_sum, divisor):
    """main.rs:201-222; the reference passes divisor = strata + 1 (its Q1 quirk)."""
    from . import lib
    a = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    out = np.zeros(a.shape, dtype=np.uint8)
    rc = lib.rt_tonemap(a.ctypes.data_as(C.POINTER(C.c_double)), a.size // 3, float(divisor),
                        out.ctypes.data_as(C.POINTER(C.c_uint8)))
    _check(lib, rc, "rt_tonemap")
    return out


def tonemap_device(d_rgb_sum_ptr, n_pixels, divisor, d_rgb8_ptr, stream=None):
    """The same pixel math as a kernel: device pointers (ints) in and out."""
    from . import lib
    _check(lib, lib.rt_tonemap_device(d_rgb_sum_ptr, int(n_pixels), float(divisor), d_rgb8_ptr, stream), "rt_tonemap_device")


def write_png(path, rgb8):
    """write_render_result_to_file's file half (main.rs:219-227): rgb8 is an (H, W, 3) uint8 array."""
    from . import lib
    a = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("write_png wants an (H, W, 3) uint8 array")
    rc = lib.rt_write_png(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0])
    _check(lib, rc, "rt_write_png")
