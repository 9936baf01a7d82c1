"""Parity of the HIP path (through the C ABI) with the CPU oracle, on a real MI355X.

Tolerances (BASELINE.json north_star): primary-ray hit-object ids equal on >= 99.9 % of
pixels; mean per-channel error of the tone-mapped image <= 1/255.  Both sides draw the
same counter-based random numbers (include/rt_rng.h), so in practice almost every pixel
follows the same path and the tests also assert a much tighter diagnostic: >= 99.9 % of
pixels agree to 1e-9 relative and the traced-ray counts agree to 0.1 %."""
import json
import os

import numpy as np
import pytest

import descs
import fuzz_scenes
import oracle_lib
import parity

pytestmark = pytest.mark.gpu

ID_MATCH_MIN = 0.999        # north_star
MEAN_ERR_MAX = 1.0          # in units of 1/255, north_star
CLOSE_MIN = 0.999           # pixels within 1e-9 relative of the oracle (diagnostic, tighter)
HIT_TOL = 1e-9              # ray-batch hits: t, n, uv

with open(os.path.join(parity.ROOT, "tests", "golden", "reference_unit_vectors.json")) as f:
    GOLD = json.load(f)


def _check(m, st, rst):
    assert m["id_match"] >= ID_MATCH_MIN, m
    assert m["mean_err_255"] <= MEAN_ERR_MAX, m
    assert m["px_close"] >= CLOSE_MIN, m
    assert abs(int(st.rays) - int(rst.rays)) <= max(2, 1e-3 * rst.rays), (st.rays, rst.rays)
    assert list(st.guards) == list(rst.guards)
    assert st.interval_overflow == 0


@pytest.mark.parametrize("name", parity.SCENES)
def test_scene_matches_oracle(rt, name):
    o = rt.Options(width=200, height=150, samples=4, seed=3)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert st.strata == 4 and np.isfinite(rgb).all()
    _check(parity.compare(rgb, ids, ref, rids, st.strata), st, rst)


def test_csg_without_torus_is_exact(rt):
    """csg.scene with its torus replaced by an out-of-reach sphere (same CSG tree shape):
    every path must follow the oracle's, ray for ray.  The few forking pixels of the real
    csg.scene (about 1 in 25 000) come from the quartic's complex cube root at grazing torus
    rays, where the device and host libm differ by ulps (parity notes)."""
    o = rt.Options(width=320, height=240, samples=1, seed=1)
    sc = rt.parse_file(o, parity.scene_path("csg_notorus"), parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert st.rays == rst.rays
    assert np.array_equal(ids, rids)
    assert np.allclose(rgb, ref, rtol=1e-9, atol=1e-12)


def test_several_meshes_and_mesh_objects(rt):
    """Two mesh objects, one of them an OBJ with two `o` groups (a Composite of two Meshes that
    overlap), with transforms, a metal and a glass material around them: the (object, mesh) job
    order of k_mesh, the early-out bounds between meshes and the tail kernel with several meshes."""
    o = rt.Options(width=240, height=180, samples=4, seed=11)
    sc = rt.parse_file(o, parity.scene_path("meshes2"), parity.ASSETS)
    assert sc.desc.contents.n_meshes == 3
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert {1, 2, 3, 4} <= set(np.unique(rids).tolist())      # floor, both mesh objects, the glass ball
    _check(parity.compare(rgb, ids, ref, rids, st.strata), st, rst)


def test_row_major_output_equals_unsharded(rt):
    """RT_FLAG_ROW_MAJOR (one shard): the kernels index the outputs by pixel; same image as shard
    layout + unshard, bit for bit, also when the image does not fill its last tiles."""
    import torch
    from raytracer_amd import system
    o = rt.Options(width=300, height=170, samples=4, seed=21)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    dev = rt.DeviceScene(sc)
    whole, wids, wst = dev.render(o)                     # rt_render: row-major internally
    n = system.shard_pixels(o, 0, 1)
    assert n > 300 * 170                                 # padding slots exist
    buf = torch.zeros(n * 3, dtype=torch.float64, device="cuda")
    ibuf = torch.zeros(n, dtype=torch.int32, device="cuda")
    st = dev.render_shard(o, 0, 1, buf.data_ptr(), ibuf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    full = torch.zeros(170 * 300 * 3, dtype=torch.float64, device="cuda")
    fids = torch.zeros(170 * 300, dtype=torch.int32, device="cuda")
    system.unshard(o, 0, 1, buf.data_ptr(), full.data_ptr(), ibuf.data_ptr(), fids.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy().reshape(170, 300, 3), whole)
    assert np.array_equal(fids.cpu().numpy().reshape(170, 300), wids)
    assert st.rays == wst.rays
    with pytest.raises(RuntimeError):                    # the flag is for a single shard
        dev.render_shard(o, 0, 2, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream, row_major=True)


@pytest.mark.parametrize("name", ["mesh", "meshes2"])
def test_kd_tree_walk_matches_oracle_and_grid(rt, monkeypatch, name):
    """The other mesh structure of k_mesh (RT_ACCEL=kd: SAH kd-trees, LDS stacks): same image as the
    oracle, and bit for bit the image of the grid walk - both only choose which triangles get the
    reference's test."""
    o = rt.Options(width=240, height=180, samples=4, seed=13)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    monkeypatch.setenv("RT_ACCEL", "grid")
    g_rgb, g_ids, g_st = rt.DeviceScene(sc).render(o)
    monkeypatch.setenv("RT_ACCEL", "kd")
    dev = rt.DeviceScene(sc)
    rgb, ids, st = dev.render(o)
    assert np.array_equal(rgb, g_rgb) and np.array_equal(ids, g_ids) and st.rays == g_st.rays
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    _check(parity.compare(rgb, ids, ref, rids, st.strata), st, rst)
    ro, rd = _random_rays(90000, 8, np.array([0.0, 1.0, 0.0]), 2.5)
    got, _ = dev.trace_rays(ro, rd, seed=4)
    want, _ = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    same = got["object"] == want["object"]
    assert same.mean() >= ID_MATCH_MIN
    hit = same & (want["object"] >= 0)
    assert np.mean(np.abs(got["t"][hit] - want["t"][hit]) / np.maximum(1.0, np.abs(want["t"][hit])) < HIT_TOL) >= ID_MATCH_MIN


def test_crowded_mesh_teapot_in_a_stadium(rt, tmp_path, monkeypatch):
    """A mesh whose triangles crowd into a corner of its bounds (1500 small ones in a 0.3 box on a
    60 x 60 ground quad): the uniform grid puts hundreds of them into single cells, the scene then
    defaults to the kd-tree; both structures must still give the oracle's image."""
    rng = np.random.default_rng(3)
    lines = ["o Stadium", "v -30 0 -30", "v 30 0 -30", "v 30 0 30", "v -30 0 30", "vn 0 1 0"]
    faces = ["f 1//1 3//1 2//1", "f 1//1 4//1 3//1"]                  # wound to face +y
    nv = 4
    for _ in range(1500):
        c = np.array([0.0, 0.4, 0.0]) + rng.uniform(-0.15, 0.15, 3)
        p = c + rng.uniform(-0.03, 0.03, (3, 3))
        for q in p:
            lines.append("v %.6f %.6f %.6f" % tuple(q))
        faces.append("f %d//1 %d//1 %d//1" % (nv + 1, nv + 2, nv + 3))
        nv += 3
    (tmp_path / "stadium.obj").write_text("\n".join(lines + ["s 1"] + faces) + "\n")
    text = """options { background color rgb <0.4, 0.5, 0.7> }
camera { origin <0.6, 0.9, 1.2> look_at <0, 0.4, 0> }
object "light" { sphere { origin <4, 6, 3> radius 2 } material { diffuse_light intensity 4 texture { solid color white } } }
object "stadium" { mesh { file "stadium.obj" } material { lambertian texture { solid color rgb <0.8, 0.6, 0.4> } } }
"""
    o = rt.Options(width=200, height=150, samples=1, seed=2)
    sc = rt.parse(o, text, str(tmp_path))
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert (rids == 1).mean() > 0.5
    images = []
    for accel in (None, "grid", "kd"):
        if accel:
            monkeypatch.setenv("RT_ACCEL", accel)
        rgb, ids, st = rt.DeviceScene(sc).render(o)
        _check(parity.compare(rgb, ids, ref, rids, st.strata), st, rst)
        images.append(rgb)
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[1], images[2])


def test_many_mesh_objects(rt):
    """18 mesh objects of two meshes each = 36 (object, mesh) job records: more than k_mesh keeps in LDS,
    and the grid cell budget is split 36 ways."""
    text = ["options { background color rgb <0.5, 0.6, 0.8> }", "camera { origin <0, 6, 12> look_at <0, 1, 0> }",
            'object "floor" { plane { origin <0, 0, 0> normal <0, 1, 0> } material { lambertian texture { solid color rgb <0.6, 0.6, 0.6> } } }']
    for i in range(18):
        x, z = (i % 6) * 2.4 - 6.0, (i // 6) * 3.0 - 3.0
        text.append('object { mesh { file "two_parts.obj" transform { rotate_y %d translate <%.1f, 1.1, %.1f> } } '
                    "material { lambertian texture { solid color rgb <0.9, %.2f, 0.3> } } }" % (17 * i, x, z, 0.2 + 0.04 * i))
    o = rt.Options(width=240, height=160, samples=1, seed=5, max_depth=6)
    sc = rt.parse(o, "\n".join(text), parity.ASSETS)
    assert sc.desc.contents.n_meshes == 36
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert len(np.unique(rids)) >= 18
    dev = rt.DeviceScene(sc)
    rgb, ids, st = dev.render(o)
    _check(parity.compare(rgb, ids, ref, rids, 1), st, rst)
    ro, rd = _random_rays(120000, 3, np.array([0.0, 1.0, 0.0]), 5.0)      # long queue: the grid walk
    got, _ = dev.trace_rays(ro, rd, seed=4)
    want, _ = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    assert (got["object"] == want["object"]).mean() >= ID_MATCH_MIN


def test_scene_too_large_for_lds_tables(rt):
    """120 objects: the scene tables (objects, leaves) exceed the 32 KB the kernels stage in LDS, so
    they are read from memory; same paths either way."""
    rng = np.random.default_rng(9)
    shapes = [{"kind": "plane", "params": [0, -1.2, 0, 0, 1, 0]}]
    for i in range(119):
        c = rng.uniform([-3, -1, -9], [3, 2, -4])
        if i % 3:
            shapes.append({"kind": "sphere", "params": [*c, 0.25]})
        else:
            shapes.append({"kind": "cube", "params": [*(c - 0.2), *(c + 0.2)]})
    d = descs.Desc(shapes)
    for k in range(3):
        d.desc.background[k] = [0.6, 0.7, 0.9][k]
    o = rt.Options(width=160, height=120, samples=1, seed=6, max_depth=8)
    sc = descs.FakeScene(d)
    rgb, ids, st = rt.DeviceScene(sc).render(o)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert len(np.unique(rids)) > 30
    _check(parity.compare(rgb, ids, ref, rids, 1), st, rst)


def test_stats_account_for_every_stage(rt):
    """rt_stats: per-kernel times from HIP events (RT_FLAG_STAGE_TIMES), ray and mesh-queue counts; the tail
    kernel's rays are counted.  Without the flag no stage event is recorded and the times stay 0."""
    o = rt.Options(width=256, height=192, samples=1, seed=4)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    rgb0, ids0, st0 = rt.render(o, sc)
    assert st0.ms_intersect == 0 and st0.ms_mesh == 0 and st0.ms_shade == 0 and st0.ms_tail == 0 and st0.ms_device > 0
    o.stage_times = True
    rgb, ids, st = rt.render(o, sc)
    assert np.array_equal(rgb, rgb0) and np.array_equal(ids, ids0) and st.rays == st0.rays
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert st.rays == rst.rays and st.primary_rays == 256 * 192
    assert st.bounce_passes >= 3 and st.n_intersect_launches >= 1
    assert 0 < st.intersect_rays <= st.rays            # rays the tail kernel traced are in rays only
    assert st.mesh_rays <= st.intersect_rays
    assert st.ms_intersect > 0 and st.ms_shade > 0 and st.ms_mesh > 0 and st.ms_tail > 0
    assert st.ms_device >= st.ms_intersect + st.ms_mesh + st.ms_shade + st.ms_tail - 1e-3


def test_plumbing_config_three_spheres_640x480(rt):
    """BASELINE configs[0]: three-spheres at 640x480, samples=1, seed 1."""
    o = rt.Options(width=640, height=480, samples=1, seed=1)
    sc = rt.parse_file(o, parity.scene_path("three-spheres"), parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    _check(parity.compare(rgb, ids, ref, rids, 1), st, rst)


def test_non_square_sample_count_quirk(rt):
    """samples=5 renders floor(sqrt(5))^2 = 4 strata with recip 1/sqrt(5)."""
    o = rt.Options(width=96, height=64, samples=5, seed=2)
    sc = rt.parse_file(o, parity.scene_path("cylinder"), parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=8)
    assert st.strata == 4 and rst.strata == 4
    _check(parity.compare(rgb, ids, ref, rids, 4), st, rst)


def test_depth_limits(rt):
    """max_depth 0 -> background everywhere; max_depth 1 -> only lights are visible."""
    sc_text = open(parity.scene_path("cornell_box")).read().replace("background color black", "background color rgb <0.1, 0.2, 0.3>")
    for depth in (0, 1, 3):
        o = rt.Options(width=80, height=60, samples=1, seed=5, max_depth=depth)
        sc = rt.parse(o, sc_text, parity.ASSETS)
        rgb, ids, st = rt.render(o, sc)
        ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=4)
        assert np.allclose(rgb, ref, rtol=1e-12, atol=0), depth
        assert st.rays == rst.rays
        if depth == 0:
            assert np.allclose(rgb, [0.1, 0.2, 0.3]) and st.rays == 0


def test_full_size_band_mesh_4k(rt):
    """BASELINE configs[2] at its full size: a band of rows of the 4K frame against the oracle."""
    o = rt.Options(width=3840, height=2160, samples=1, seed=1)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    dev = rt.DeviceScene(sc)
    rgb, ids, st = dev.render(o)
    y0, y1 = 1000, 1024
    ref, rids, rst = oracle_lib.render_rows(sc.desc, o.to_c(), y0, y1, threads=16)
    m = parity.compare(rgb[y0:y1], ids[y0:y1], ref, rids, 1)
    assert m["id_match"] >= ID_MATCH_MIN and m["mean_err_255"] <= MEAN_ERR_MAX and m["px_close"] >= CLOSE_MIN, m
    # size-independent properties of the whole frame
    assert np.isfinite(rgb).all() and (rgb >= 0).all()
    assert rgb.max() <= 3.0 + 1e-12           # attenuations <= 1, the only light has intensity 3
    assert set(np.unique(ids)) <= {-1, 0, 1, 2}
    assert st.primary_rays == 3840 * 2160 and st.rays >= st.primary_rays
    # same seed -> bit-identical frame; sharded rendering -> the same pixels
    rgb2, ids2, st2 = dev.render(o)
    assert np.array_equal(rgb, rgb2) and np.array_equal(ids, ids2) and st.rays == st2.rays


def _render_sharded(dev, o, count):
    """The frame as `count` tile shards rendered one after the other on this GPU (as `count` GPUs would render
    them), scattered into the row-major image with rt_unshard_device -> (rgb [H,W,3], ids [H,W], rays)."""
    import torch
    from raytracer_amd import system
    H, W = o.height, o.width
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros(H * W * 3, dtype=torch.float64, device="cuda")
    fids = torch.full((H * W,), -5, dtype=torch.int32, device="cuda")
    n_max = max(system.shard_pixels(o, r, count) for r in range(count))
    buf = torch.zeros(n_max * 3, dtype=torch.float64, device="cuda")
    ibuf = torch.zeros(n_max, dtype=torch.int32, device="cuda")
    rays = 0
    for r in range(count):
        st = dev.render_shard(o, r, count, buf.data_ptr(), ibuf.data_ptr(), stream)
        rays += st.rays
        system.unshard(o, r, count, buf.data_ptr(), full.data_ptr(), ibuf.data_ptr(), fids.data_ptr(), stream)
    torch.cuda.synchronize()
    return full.cpu().numpy().reshape(H, W, 3), fids.cpu().numpy().reshape(H, W), rays


# BASELINE.json configs[1..4] at their full sizes: (scene, width, height, samples, first row of the band, rows,
# compare the whole frame's ray count with the oracle's, also render as 8 shards)
FULL_SIZE = [("cornell_box", 1920, 1080, 1, 500, 24, True, False),
             ("mesh", 3840, 2160, 1, 1000, 24, True, True),
             ("torus", 3840, 2160, 1, 1068, 24, True, False),
             ("csg", 3840, 2160, 1, 1068, 24, True, False),
             ("fog_light", 7680, 4320, 16, 2600, 16, False, True)]


@pytest.mark.parametrize("name,w,h,samples,y0,rows,count_rays,sharded", FULL_SIZE, ids=[c[0] for c in FULL_SIZE])
def test_full_size_band(rt, name, w, h, samples, y0, rows, count_rays, sharded):
    """Every BASELINE configuration at its full size (system.rs:244-253,276-286): a band of rows, all strata
    summed, against the oracle (ids, tone-mapped error, every pixel to 1e-9); the whole frame's ray count
    against the oracle's where the oracle finishes the frame in seconds (not 8K x 16: there the per-pixel
    agreement to 1e-9 of the band is what says that the paths are the same); the frame rendered as 8 tile
    shards equal to the unsharded frame, bit for bit (8K x 16 = 33 M slots: slot and offset arithmetic)."""
    o = rt.Options(width=w, height=h, samples=samples, seed=1)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    dev = rt.DeviceScene(sc)
    rgb, ids, st = dev.render(o)
    assert st.strata == samples and st.primary_rays == w * h * samples
    assert np.isfinite(rgb).all() and (rgb >= 0).all()
    y1 = y0 + rows
    ref, rids, rst = oracle_lib.render_rows_strata(sc.desc, o.to_c(), y0, y1, threads=16)
    m = parity.compare(rgb[y0:y1], ids[y0:y1], ref, rids, st.strata)
    assert m["id_match"] >= ID_MATCH_MIN and m["mean_err_255"] <= MEAN_ERR_MAX and m["px_close"] >= CLOSE_MIN, m
    assert len(np.unique(rids)) >= 2                       # the band crosses geometry, not only background
    assert st.interval_overflow == 0
    if count_rays:
        _, _, fst = oracle_lib.render(sc.desc, o.to_c(), threads=16, want_ids=False)
        assert abs(int(st.rays) - int(fst.rays)) <= 1e-3 * fst.rays, (st.rays, fst.rays)
        assert list(st.guards) == list(fst.guards)
    if sharded:
        srgb, sids, srays = _render_sharded(dev, o, 8)
        assert np.array_equal(srgb, rgb) and np.array_equal(sids, ids) and srays == st.rays


def test_sharded_render_equals_whole(rt):
    """Tiles rendered as 3 shards (as 3 GPUs would) reassemble to the single-GPU frame, bit for bit."""
    import torch
    from raytracer_amd import system
    o = rt.Options(width=300, height=200, samples=4, seed=9, tile_size=32)
    sc = rt.parse_file(o, parity.scene_path("csg"), parity.ASSETS)
    dev = rt.DeviceScene(sc)
    whole, wids, wst = dev.render(o)
    full = torch.zeros(200 * 300 * 3, dtype=torch.float64, device="cuda")
    fids = torch.full((200 * 300,), -5, dtype=torch.int32, device="cuda")
    full8 = torch.zeros(200 * 300 * 3, dtype=torch.uint8, device="cuda")
    rays = 0
    cs = torch.cuda.current_stream().cuda_stream
    for r in range(3):
        n = system.shard_pixels(o, r, 3)
        buf = torch.zeros(n * 3, dtype=torch.float64, device="cuda")
        ibuf = torch.zeros(n, dtype=torch.int32, device="cuda")
        st = dev.render_shard(o, r, 3, buf.data_ptr(), ibuf.data_ptr(), cs)
        rays += st.rays
        system.unshard(o, r, 3, buf.data_ptr(), full.data_ptr(), ibuf.data_ptr(), fids.data_ptr(), cs)
        # the same shard tone-mapped where it lies (what bench.py sends over the links: 3 bytes per slot), then scattered
        buf8 = torch.zeros(n * 3, dtype=torch.uint8, device="cuda")
        system.tonemap_device(buf.data_ptr(), n, st.strata + 1, buf8.data_ptr(), cs)
        system.unshard_rgb8(o, r, 3, buf8.data_ptr(), full8.data_ptr(), cs)
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy().reshape(200, 300, 3), whole)
    assert np.array_equal(fids.cpu().numpy().reshape(200, 300), wids)
    assert rays == wst.rays
    assert np.array_equal(full8.cpu().numpy().reshape(200, 300, 3), rt.tonemap(whole, wst.strata + 1))
    # and the torch-side scatter used after the RCCL gather gives the same image
    buf = torch.zeros(system.shard_pixels(o, 0, 1) * 3, dtype=torch.float64, device="cuda")
    dev.render_shard(o, 0, 1, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    img = system.gather_frame(o, buf, 0, 1)
    assert np.array_equal(img.cpu().numpy().reshape(200, 300, 3), whole)


def test_progress_callbacks(rt):
    """RenderProgress protocol (system.rs:223-227): started once, one call per stratum with
    the running sum, finished once."""
    calls = []

    class P(rt.RenderProgress):
        def render_started(self, options):
            calls.append(("start", None))

        def sample_finished(self, options, buf):
            calls.append(("sample", float(buf.sum())))

        def render_finished(self, options, buf):
            calls.append(("finish", float(buf.sum())))

    o = rt.Options(width=64, height=48, samples=9, seed=1)
    sc = rt.parse_file(o, parity.scene_path("transformations"), parity.ASSETS)
    rt.render(o, sc, P())
    kinds = [c[0] for c in calls]
    assert kinds == ["start"] + ["sample"] * 9 + ["finish"]
    sums = [c[1] for c in calls[1:-1]]
    assert all(b >= a for a, b in zip(sums, sums[1:])) and abs(sums[-1] - calls[-1][1]) < 1e-9


def _random_rays(n, seed, center, radius):
    rng = np.random.default_rng(seed)
    o = center + rng.normal(size=(n, 3)) * radius
    t = center + rng.normal(size=(n, 3)) * radius * 0.3
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


@pytest.mark.parametrize("name,center,radius", [("three-spheres", [0, 1.5, -4], 3.0), ("csg", [0, 1, 0], 3.0),
                                                ("torus", [0, 1.1, 0], 2.0), ("cylinder", [0, 0.7, 0], 2.5),
                                                ("mesh", [0, 1, 0], 3.0), ("room", [0, 3, 0], 4.0),
                                                ("fog_light", [0, 2, -4], 3.0), ("cornell_box", [278, 273, 200], 250.0)])
def test_ray_batches_match_oracle(rt, name, center, radius):
    """Ray::trace on arbitrary rays, including rays that start inside shapes and rays with t < 0 roots."""
    o = rt.Options(width=16, height=16)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    ro, rd = _random_rays(20000, 11, np.array(center, dtype=float), radius)
    dev = rt.DeviceScene(sc)
    got, gst = dev.trace_rays(ro, rd, seed=4)
    ref, rst = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    same_obj = got["object"] == ref["object"]
    # Coincident surfaces (the cylinder's cap and the cube's base lie in the floor plane and are
    # seen by rays from below it): two objects at the same t, and the winner hangs on the last
    # bit of t.  Such a ray counts as agreeing when both sides hit something at the same distance.
    tie = (~same_obj) & (got["object"] >= 0) & (ref["object"] >= 0) & \
        (np.abs(got["t"] - ref["t"]) <= HIT_TOL * np.maximum(1.0, np.abs(ref["t"])))
    assert (same_obj | tie).mean() >= ID_MATCH_MIN, (same_obj.mean(), tie.mean())
    hit = same_obj & (ref["object"] >= 0)
    scale = np.maximum(1.0, np.abs(ref["t"][hit]))
    assert np.mean(np.abs(got["t"][hit] - ref["t"][hit]) / scale < HIT_TOL) >= ID_MATCH_MIN
    assert np.mean(np.all(np.abs(got["n"][hit] - ref["n"][hit]) < 1e-7, axis=1)) >= ID_MATCH_MIN
    assert np.mean(np.all(np.abs(got["uv"][hit] - ref["uv"][hit]) < 1e-6, axis=1)) >= 0.995  # atan2 seams
    assert list(gst.guards) == list(rst.guards)


@pytest.mark.parametrize("n_rays", [8000, 90000])     # wave-per-ray kernel; grid walk
def test_ray_batches_several_meshes(rt, n_rays):
    """Three meshes in two objects with their own frames, rays from everywhere around and inside them."""
    o = rt.Options(width=16, height=16)
    sc = rt.parse_file(o, parity.scene_path("meshes2"), parity.ASSETS)
    ro, rd = _random_rays(n_rays, 5, np.array([0.0, 1.0, 0.3]), 2.5)
    dev = rt.DeviceScene(sc)
    got, gst = dev.trace_rays(ro, rd, seed=4)
    ref, rst = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    assert (ref["object"] == 2).mean() > 0.05 and (ref["object"] == 4).mean() > 0.02    # both mesh objects are hit
    same_obj = got["object"] == ref["object"]
    assert same_obj.mean() >= ID_MATCH_MIN, same_obj.mean()
    hit = same_obj & (ref["object"] >= 0)
    scale = np.maximum(1.0, np.abs(ref["t"][hit]))
    assert np.mean(np.abs(got["t"][hit] - ref["t"][hit]) / scale < HIT_TOL) >= ID_MATCH_MIN
    assert np.mean(np.all(np.abs(got["n"][hit] - ref["n"][hit]) < 1e-7, axis=1)) >= ID_MATCH_MIN
    mesh_hit = hit & ((ref["object"] == 2) | (ref["object"] == 4))
    assert np.mean(np.all(np.abs(got["uv"][mesh_hit] - ref["uv"][mesh_hit]) < 1e-7, axis=1)) >= ID_MATCH_MIN
    # every ray of the reference has a unit direction; others are refused rather than traced differently
    with pytest.raises(RuntimeError):
        dev.trace_rays(ro[:4], rd[:4] * 2.0, seed=4)


def _random_affine(rng):
    """rotation * non-uniform scale, then a translation; row-vector convention (point * M)."""
    a, b, c = rng.uniform(0, 2 * np.pi, 3)
    rx = np.array([[1, 0, 0], [0, np.cos(a), np.sin(a)], [0, -np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, -np.sin(b)], [0, 1, 0], [np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), np.sin(c), 0], [-np.sin(c), np.cos(c), 0], [0, 0, 1]])
    m = np.eye(4)
    m[:3, :3] = np.diag(rng.uniform(0.5, 1.8, 3)) @ rx @ ry @ rz
    m[3, :3] = rng.uniform(-2.5, 2.5, 3)
    return m.reshape(16).tolist()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_transformed_primitives(rt, seed):
    """Fuzz: every primitive kind under random rotations, non-uniform scales and translations (the composed
    world->leaf / leaf->world maps, `Intersection::to_world`'s t and normal), 24 objects, 60 000 rays."""
    rng = np.random.default_rng(seed)
    shapes = []
    for i in range(24):
        kind = ["sphere", "cube", "cylinder", "torus", "xyrect", "xzrect", "zyrect", "plane"][i % 8]
        params = {"sphere": [*rng.uniform(-0.5, 0.5, 3), rng.uniform(0.3, 0.9)],
                  "cube": [*rng.uniform(-0.8, -0.2, 3), *rng.uniform(0.2, 0.8, 3)],
                  "cylinder": [rng.uniform(0.3, 0.8), rng.uniform(0.5, 1.5)],
                  "torus": [rng.uniform(0.6, 1.0), rng.uniform(0.1, 0.3)],
                  "xyrect": [*rng.uniform(-0.3, 0.3, 3), rng.uniform(0.5, 2), rng.uniform(0.5, 2)],
                  "xzrect": [*rng.uniform(-0.3, 0.3, 3), rng.uniform(0.5, 2), rng.uniform(0.5, 2)],
                  "zyrect": [*rng.uniform(-0.3, 0.3, 3), rng.uniform(0.5, 2), rng.uniform(0.5, 2)],
                  "plane": [0, -4.0 - i, 0, *rng.normal(size=3)]}[kind]
        shape = {"kind": kind, "params": params}
        if kind != "plane" and i >= 8:                      # the first of each kind stays untransformed
            shape["o2w"] = _random_affine(rng)
        shapes.append(shape)
    d = descs.Desc(shapes)
    dev = rt.DeviceScene(descs.FakeScene(d))
    ro, rd = _random_rays(60000, 100 + seed, np.array([0.0, 0.0, 0.0]), 3.5)
    got, gst = dev.trace_rays(ro, rd, seed=4)
    ref, rst = oracle_lib.trace_rays(d.ptr, ro, rd, seed=4)
    assert (ref["object"] >= 0).mean() > 0.5 and len(np.unique(ref["object"])) >= 20
    same = got["object"] == ref["object"]
    assert same.mean() >= ID_MATCH_MIN, same.mean()
    hit = same & (ref["object"] >= 0)
    scale = np.maximum(1.0, np.abs(ref["t"][hit]))
    assert np.mean(np.abs(got["t"][hit] - ref["t"][hit]) / scale < 1e-8) >= ID_MATCH_MIN
    assert np.mean(np.all(np.abs(got["n"][hit] - ref["n"][hit]) < 1e-6, axis=1)) >= ID_MATCH_MIN
    assert list(gst.guards) == list(rst.guards)


@pytest.mark.parametrize("seed,torus", [(s, True) for s in range(1, 9)] + [(s, False) for s in range(60, 80)])
def test_random_scenes(rt, seed, torus):
    """Fuzz over the whole scene grammar (tests/fuzz_scenes.py): CSG trees with transforms at every level, meshes,
    media, every material and texture.  Without tori every path must follow the oracle's (same ray count, every
    pixel within 1e-9); with tori the quartic's libm differences may fork a path in 25 000."""
    text = fuzz_scenes.scene_text(seed, torus=torus)
    o = rt.Options(width=160, height=120, samples=4, seed=seed)
    sc = rt.parse(o, text, parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    m = parity.compare(rgb, ids, ref, rids, st.strata)
    _check(m, st, rst)
    if not torus:
        assert st.rays == rst.rays and m["px_close"] == 1.0 and m["id_match"] == 1.0, (m, st.rays, rst.rays)


def test_difference_of_a_two_interval_union(rt):
    """Regression (found by the fuzz above, seeds 64 and 76 without tori): `difference { union { a b } c }` where
    the ray meets a and b in two separate intervals, c swallows the first and trims the second.  The device lost
    the trimmed interval: the loop that merges two interval lists carried two boolean flags round a loop with two
    dozen exits and came out of the compiler wrong (numerics and parity notes); csg_apply now keeps
    integer indices instead."""
    text = """camera { origin <0.0467, 1.8314, 7.04> look_at <0.9663, 1.5929, 2.0043> fov 20 }
object "x" { difference {
    union { cube { <-0.5779, -0.3604, -0.5133> <0.1462, 1.1246, 0.9012>
                   transform { rotate_x 170.207 translate <-0.0036, 0.6639, 0.1757> } }
            sphere { origin <0.0321, -0.2402, 0.1231> radius 0.9806
                     transform { rotate_y 114.9958 translate <0.3196, 0.5964, -0.339> } }
            transform { translate <-0.107, 0.31, -0.2615> } }
    cube { <-0.2696, -0.8113, -0.3193> <0.7057, -0.2483, 0.4454> transform { translate <-0.092, 0.3407, 0.0238> } }
    transform { translate <0.9663, 1.5929, 2.0043> } }
  material { lambertian texture { solid color white } } }
"""
    o = rt.Options(width=16, height=16)
    sc = rt.parse(o, text, parity.ASSETS)
    # the ray that failed, and a cone of rays from the same eye point over the whole object
    ro = np.tile(np.array([0.0467, 1.8314, 7.04]), (40001, 1))
    rng = np.random.default_rng(7)
    tgt = np.array([0.9663, 1.5929, 2.0043]) + rng.uniform(-1.6, 1.6, (40001, 3))
    rd = tgt - ro
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[0] = [0.18580452142331003, -0.07441346100994915, -0.9797649292759849]
    rd[0] /= np.linalg.norm(rd[0])
    dev = rt.DeviceScene(sc)
    got, _ = dev.trace_rays(ro, rd, seed=4)
    ref, _ = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    assert ref["object"][0] == 0 and abs(ref["t"][0] - 5.441305195461109) < 1e-9
    assert got["object"][0] == 0 and abs(got["t"][0] - ref["t"][0]) < 1e-9
    assert np.array_equal(got["object"], ref["object"])
    hit = ref["object"] >= 0
    assert 0.2 < hit.mean() < 0.95
    assert np.max(np.abs(got["t"][hit] - ref["t"][hit])) < 1e-9
    assert np.max(np.abs(got["n"][hit] - ref["n"][hit])) < 1e-9


def test_union_and_difference_skip_their_own_to_world(rt):
    """csg.rs:33-38,216-221: a union or difference hands back one operand's intervals untouched when the ray
    misses the other operand, i.e. without its own to_world.  With a rotation or scale on that node the hit keeps
    the node's space (t scaled, normal short of one map) and the shape shows up displaced for those rays.  The
    device reproduces it (DEV_PROG_SKIP); here two levels of it, under an intersection and beside a difference."""
    tx = "transform { scale <1.5, 0.7, 1.2> rotate_y 35 rotate_z 20 translate <0.3, 0.2, -0.4> }"
    inner = "union { sphere { origin <-1.2, 0, 0> radius 0.8 } cube { <0.6, -0.5, -0.5> <1.6, 0.5, 0.5> } %s }"
    text = ("camera { origin <0, 0, 9> look_at <0, 0, 0> }\n"
            'object "a" { difference { %s cylinder { radius 0.3 height 4 transform { translate <2.5, 0, 0> } } '
            "transform { scale <0.8, 1.3, 1> rotate_x 25 } } material { lambertian texture { solid color white } } }\n"
            'object "b" { intersection { %s sphere { origin <0, 0, 0> radius 5 } transform { translate <0, 3.5, 0> } } '
            "material { lambertian texture { solid color white } } }\n") % (inner % tx, inner % tx)
    flat = ("camera { origin <0, 0, 9> look_at <0, 0, 0> }\n"       # the same solids with every map on the leaves
            'object "s" { sphere { origin <-1.2, 0, 0> radius 0.8 transform { %s scale <0.8, 1.3, 1> rotate_x 25 } } '
            "material { lambertian texture { solid color white } } }\n") % tx[len("transform { "):-2]
    o = rt.Options(width=16, height=16)
    sc = rt.parse(o, text, parity.ASSETS)
    ro, rd = _random_rays(60000, 21, np.array([0.0, 1.5, 0.0]), 4.0)
    dev = rt.DeviceScene(sc)
    got, gst = dev.trace_rays(ro, rd, seed=4)
    ref, rst = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    same = got["object"] == ref["object"]
    assert same.mean() >= ID_MATCH_MIN, same.mean()
    assert (ref["object"] == 0).mean() > 0.05 and (ref["object"] == 1).mean() > 0.05
    hit = same & (ref["object"] >= 0)
    scale = np.maximum(1.0, np.abs(ref["t"][hit]))
    assert np.mean(np.abs(got["t"][hit] - ref["t"][hit]) / scale < HIT_TOL) >= ID_MATCH_MIN
    assert np.mean(np.all(np.abs(got["n"][hit] - ref["n"][hit]) < 1e-7, axis=1)) >= ID_MATCH_MIN
    assert list(gst.guards) == list(rst.guards)
    # and the quirk is what is being compared: the sphere alone, with the composed maps on the leaf, is elsewhere
    fsc = rt.parse(o, flat, parity.ASSETS)
    geo, _ = oracle_lib.trace_rays(fsc.desc, ro, rd, seed=4)
    both = (ref["object"] == 0) & (geo["object"] == 0)
    assert both.sum() > 500 and np.mean(np.abs(ref["t"][both] - geo["t"][both]) > 1e-3) > 0.5


def test_tonemap_kernel_equals_host_pixel_math(rt, tmp_path):
    """rt_tonemap_device against rt_tonemap and the oracle (main.rs:201-217), incl. NaN, negative and over-range
    sums, sizes that are no multiple of 4 and an output that is not 4-byte aligned; then a rendered frame to PNG."""
    import torch
    rng = np.random.default_rng(3)
    for n_px in (1, 5, 1000, 1920 * 1080 + 1):
        vals = rng.uniform(-0.2, 3.0, (n_px, 3))
        vals[rng.integers(0, n_px, 3), rng.integers(0, 3, 3)] = np.nan
        vals[0, 0] = 1e300
        d_in = torch.from_numpy(vals).cuda()
        d_out = torch.zeros(n_px * 3 + 1, dtype=torch.uint8, device="cuda")
        for off in (0, 1):
            rt.tonemap_device(d_in.data_ptr(), n_px, 2.0, d_out.data_ptr() + off)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()[off:off + n_px * 3].reshape(n_px, 3)
            assert np.array_equal(got, rt.tonemap(vals, 2.0))
            assert np.array_equal(got, oracle_lib.tonemap(vals, 2.0))
    from PIL import Image
    o = rt.Options(width=96, height=64, samples=4, seed=2)
    sc = rt.parse_file(o, parity.scene_path("three-spheres"), parity.ASSETS)
    rgb, ids, st = rt.render(o, sc)
    ref, _, _ = oracle_lib.render(sc.desc, o.to_c(), threads=8)
    d_rgb = torch.from_numpy(rgb).cuda()
    d_img = torch.zeros((64, 96, 3), dtype=torch.uint8, device="cuda")
    rt.tonemap_device(d_rgb.data_ptr(), 96 * 64, st.strata + 1, d_img.data_ptr())
    torch.cuda.synchronize()
    rt.write_png(tmp_path / "out.png", d_img.cpu().numpy())
    with Image.open(tmp_path / "out.png") as im:
        img = np.asarray(im).astype(int)
    want = oracle_lib.tonemap(ref, st.strata + 1).astype(int)
    assert np.abs(img - want).max() <= 1 and np.mean(img != want) < 1e-3


def test_nested_csg_chains(rt):
    """Right-nested chains (`union { a union { b union { c ... } } }`) keep one more interval list alive per level,
    and a union of n solids in a row can hold n intervals (more: csg.rs:78-79 pushes the interval in hand again).
    Six levels of unions, a difference and an intersection with such a chain as second operand: within the
    device's 8 lists of 12 intervals.  Deeper nesting is refused when the scene is created, not miscounted."""
    def sphere(k):
        return "sphere { origin <%g, %g, 0> radius 0.45 }" % (0.6 * k - 1.5, 0.15 * (k % 3))

    def chain(op, k, n):
        return sphere(k) if k == n - 1 else "%s { %s %s }" % (op, sphere(k), chain(op, k + 1, n))

    mat = " material { lambertian texture { solid color white } } }\n"
    cam = "camera { origin <0, 0, 6> look_at <0, 0, 0> }\n"
    text = (cam + 'object "u" { ' + chain("union", 0, 7) + mat +
            'object "d" { difference { cube { <-2, -0.6, -0.4> <2.2, 0.6, 0.4> } ' + chain("union", 0, 6) +
            " transform { translate <0, 1.6, 0> } }" + mat +
            'object "i" { intersection { cube { <-2, -0.3, -0.3> <2.2, 0.5, 0.3> } ' + chain("union", 0, 6) +
            " transform { rotate_z 10 translate <0, -1.6, 0> } }" + mat)
    o = rt.Options(width=16, height=16)
    sc = rt.parse(o, text, parity.ASSETS)
    ro, rd = _random_rays(60000, 31, np.array([0.0, 0.0, 0.0]), 3.0)
    dev = rt.DeviceScene(sc)
    got, gst = dev.trace_rays(ro, rd, seed=4)
    ref, rst = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
    assert gst.interval_overflow == 0
    for k in range(3):
        assert (ref["object"] == k).mean() > 0.03
    assert np.array_equal(got["object"], ref["object"])
    hit = ref["object"] >= 0
    assert np.max(np.abs(got["t"][hit] - ref["t"][hit])) < 1e-9
    assert np.max(np.abs(got["n"][hit] - ref["n"][hit])) < 1e-9
    deep = rt.parse(o, cam + 'object "b" { ' + chain("union", 0, 10) + mat, parity.ASSETS)
    with pytest.raises(RuntimeError, match="interval lists at once"):
        rt.DeviceScene(deep)

    def left(k):        # the same ten solids nested on the first operand need two lists
        return sphere(0) if k == 0 else "union { %s %s }" % (left(k - 1), sphere(k))
    rt.DeviceScene(rt.parse(o, cam + 'object "b" { ' + left(9) + mat, parity.ASSETS)).close()


def test_reference_known_answers_on_device(rt):
    """The reference's own unit-test vectors, first hit through rt_trace_rays."""
    for case in GOLD["shape_intervals"]:
        d = descs.Desc([case["shape"]])
        dev = rt.DeviceScene(descs.FakeScene(d))
        got, _ = dev.trace_rays(np.array([case["o"]], dtype=float), np.array([case["d"]], dtype=float))
        ts, ns = case["t"], case["n"]
        if case["shape"]["kind"] == "cube":
            k = 0  # first_intersection: the entry, dropped by Object::intersect when negative
            expect_hit = ts[0] >= 0
        else:
            pos = [i for i, t in enumerate(ts) if not (t < 0)]
            expect_hit = bool(pos)
            k = pos[0] if pos else 0
        if ts[k] == 0.0 or abs(ts[k]) < 1e-9:
            continue  # t = 0 within rounding: the sign, and with it the hit, is not pinned
        assert (got["object"][0] >= 0) == expect_hit, case
        if expect_hit:
            assert abs(got["t"][0] - ts[k]) < 1e-6 and np.allclose(got["n"][0], ns[k], atol=1e-6), case
    for case in GOLD["plane_intersect"]:
        d = descs.Desc([{"kind": "plane", "params": case["origin"] + case["normal"]}])
        dev = rt.DeviceScene(descs.FakeScene(d))
        got, _ = dev.trace_rays(np.array([case["o"]], dtype=float), np.array([case["d"]], dtype=float))
        expect = case["hit"] and case["t"] >= 0
        assert (got["object"][0] >= 0) == expect
        if expect:
            assert abs(got["t"][0] - case["t"]) < 1e-6 and np.allclose(got["n"][0], case["n"], atol=1e-6)


def test_edge_cases(rt):
    """Empty batch, a 1x1 image, an image smaller than a tile, axis-parallel rays."""
    o = rt.Options(width=16, height=16)
    sc = rt.parse_file(o, parity.scene_path("cylinder"), parity.ASSETS)
    dev = rt.DeviceScene(sc)
    got, _ = dev.trace_rays(np.zeros((0, 3)), np.zeros((0, 3)))
    assert len(got) == 0
    for w, h in ((1, 1), (5, 3), (65, 9)):
        oo = rt.Options(width=w, height=h, samples=1, seed=1)
        s2 = rt.parse_file(oo, parity.scene_path("cylinder"), parity.ASSETS)
        rgb, ids, st = rt.render(oo, s2)
        ref, rids, rst = oracle_lib.render(s2.desc, oo.to_c(), threads=1)
        assert np.array_equal(ids, rids) and np.allclose(rgb, ref, rtol=1e-9, atol=1e-12)
    # vertical rays make the cylinder's quadratic degenerate (a = 0): the reference panics
    # (cylinder.rs:136); both sides count the guard and treat the object as missed
    ro = np.array([[-0.5, 5.0, 0.0], [0.3, 5.0, 0.2]])
    rd = np.array([[0.0, -1.0, 0.0], [0.0, -1.0, 0.0]])
    got, gst = dev.trace_rays(ro, rd)
    ref, rst = oracle_lib.trace_rays(sc.desc, ro, rd)
    assert list(got["object"]) == list(ref["object"]) and list(gst.guards) == list(rst.guards)


@pytest.mark.parametrize("case", ["test_transformed_sphere_closed_form", "test_transformed_cylinder_closed_form",
                                  "test_transformed_cube_and_rects", "test_csg_first_hit_by_point_classification",
                                  "test_mesh_scene_against_brute_force_moller_trumbore", "test_fog_free_path_statistics"])
def test_analytic_cases_on_device(rt, case):
    """The numpy-only expectations of tests/test_analytic.py (closed-form quadrics, CSG by point classification,
    brute-force triangles, Beer-Lambert) against the device's rt_trace_rays instead of the oracle."""
    import test_analytic
    test_analytic.ON_DEVICE[0] = True
    try:
        getattr(test_analytic, case)(rt)
    finally:
        test_analytic.ON_DEVICE[0] = False


def test_render_multi_equals_whole_and_reports_progress(rt, tmp_path):
    """rt_render_multi (tiles over the GPUs of one process, gathered to the first): with every "device" this
    one GPU the gathered frame equals the single-GPU frame bit for bit, for 1, 2, 3 and 8 shards; progress:
    on_stratum once per stratum in order, snapshots only between strata, tone-mapped by done + 1 (main.rs:174-185),
    assembled row-major from all shards; stopping from the callback ends the render after that stratum."""
    o = rt.Options(width=300, height=200, samples=9, seed=4, tile_size=32)
    sc = rt.parse_file(o, parity.scene_path("meshes2"), parity.ASSETS)
    devs = [rt.DeviceScene(sc) for _ in range(8)]
    whole, wids, wst = devs[0].render(o)
    for n in (1, 2, 3, 8):
        rgb, ids, st = rt.render_multi(o, devs[:n])
        assert np.array_equal(rgb, whole) and np.array_equal(ids, wids), n
        assert st.rays == wst.rays and st.strata == 9 and st.primary_rays == wst.primary_rays
    # the same frame as an image: tone-mapped on the first GPU by strata + 1 (main.rs:201-222), 3 bytes per pixel out
    img8, ids8, st8 = rt.render_multi(o, devs[:3], rgb8=True)
    assert np.array_equal(img8, rt.tonemap(whole, 10)) and np.array_equal(ids8, wids) and st8.rays == wst.rays
    for n in (1, 2, 8):  # (every GPU tone-maps its own tiles before the push; progress snapshots use the same byte buffers)
        shots8 = []
        img8n, ids8n, _ = rt.render_multi(o, devs[:n], rt.DeviceProgress(on_snapshot=lambda d, t, img: shots8.append(d) and None, min_interval=0.0),
                                          rgb8=True)
        assert np.array_equal(img8n, img8) and np.array_equal(ids8n, wids) and shots8 == list(range(1, 9)), n
    one8, oid8, ost8 = devs[0].render_rgb8(o, want_ids=True)
    assert np.array_equal(one8, img8) and np.array_equal(oid8, wids) and ost8.rays == wst.rays
    # page-locked buffers the DeviceScene keeps: the same frame, the same arrays handed out again
    r1, i1, _ = devs[0].render(o, reuse=True)
    assert np.array_equal(r1, whole) and np.array_equal(i1, wids)
    r2, _, _ = devs[0].render(rt.Options(width=300, height=200, samples=9, seed=5, tile_size=32), reuse=True)
    assert r2 is r1 and not np.array_equal(r2, whole)
    # one scene serves one render at a time: the same rt_scene twice is refused
    with pytest.raises(RuntimeError, match="same rt_scene"):
        rt.render_multi(o, [devs[0], devs[1], devs[0]])
    # more shards than tiles: some devices own nothing
    small = rt.Options(width=40, height=30, samples=4, seed=4, tile_size=32)
    ssc = rt.parse_file(small, parity.scene_path("meshes2"), parity.ASSETS)
    sdevs = [rt.DeviceScene(ssc) for _ in range(5)]
    w2, i2, _ = sdevs[0].render(small)
    strata_seen = []
    r2, j2, s2 = rt.render_multi(small, sdevs, rt.DeviceProgress(on_stratum=lambda d, t: strata_seen.append((d, t)) and None,
                                                                 on_snapshot=lambda d, t, img: None, min_interval=0.0))
    assert np.array_equal(r2, w2) and np.array_equal(j2, i2) and strata_seen == [(1, 4), (2, 4), (3, 4), (4, 4)]
    # progress: every stratum reported; with min_interval 0 a snapshot after every stratum but the last
    calls, snaps = [], []
    prog = rt.DeviceProgress(on_stratum=lambda d, t: calls.append((d, t)) and None,
                             on_snapshot=lambda d, t, img: snaps.append((d, img.copy())) and None, min_interval=0.0)
    rgb, ids, st = rt.render_multi(o, devs[:3], prog)
    assert calls == [(k, 9) for k in range(1, 10)] and [d for d, _ in snaps] == list(range(1, 9))
    assert np.array_equal(rgb, whole)
    # a snapshot is the running sum after `done` strata, divided by done + 1: compare with a render of 4 of the
    # 9 strata (same strata: recip_sqrt_spp and the stratum keys belong to samples=9, so stop a 9-sample render)
    part, _, pst = rt.render_multi(o, devs[:3], rt.DeviceProgress(on_stratum=lambda d, t: d >= 4))
    assert pst.strata == 4
    want = rt.tonemap(part, 5).reshape(-1, 3)
    assert np.array_equal(snaps[3][1], want)
    # the 5-second rule: a render that takes less than min_interval never takes a snapshot
    none = []
    rt.render_multi(o, devs[:2], rt.DeviceProgress(on_snapshot=lambda d, t, img: none.append(d) and None, min_interval=5.0))
    assert none == []
    # one GPU, device buffers: the same protocol through rt_render_device_progress, snapshot in shard layout
    import torch
    from raytracer_amd import system
    n = system.shard_pixels(o, 1, 3)
    buf = torch.zeros(n * 3, dtype=torch.float64, device="cuda")
    shots = []
    devs[0].render_shard(o, 1, 3, buf.data_ptr(), None, torch.cuda.current_stream().cuda_stream,
                         progress=rt.DeviceProgress(on_snapshot=lambda d, t, img: shots.append((d, img.copy())) and None,
                                                    min_interval=0.0))
    assert [d for d, _ in shots] == list(range(1, 9)) and shots[0][1].shape == (n, 3)
    m = system.shard_map(o, 1, 3)
    full4 = want.reshape(-1, 3)
    assert np.array_equal(shots[3][1][m >= 0], full4[m[m >= 0]])
    # FileRenderProgress: the reference's CliRenderProgress without the bar
    from PIL import Image
    fp = rt.FileRenderProgress(tmp_path / "out.png", 300, 200, min_interval=0.0)
    rgb, _, st = rt.render_multi(o, devs[:2], fp)
    fp.finish(rgb, st.strata)
    assert fp.num_samples == 9 and fp.files_written == 9
    with Image.open(tmp_path / "out.png") as im:
        assert np.array_equal(np.asarray(im), rt.tonemap(whole, 10))


def test_closed_mesh_with_large_grid_tables(rt, tmp_path):
    """A closed surface of 16 000 triangles: occupancy words, distance nibbles, job records and the cell starts
    of its grid together pass 64 KB, the most a workgroup may be given on some devices.  rt_scene_create checks
    the tables against the device's limit and leaves the cell starts in memory when they do not fit (or
    refuses the scene by name); either way the image is the oracle's."""
    n_lat, n_lon = 90, 90
    lines = ["o Ball"]
    for i in range(n_lat + 1):
        th = np.pi * i / n_lat
        for j in range(n_lon):
            ph = 2 * np.pi * j / n_lon
            p = (np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph))
            lines.append("v %.9f %.9f %.9f" % p)
            lines.append("vn %.9f %.9f %.9f" % p)
    faces = []
    for i in range(n_lat):
        for j in range(n_lon):
            a, b = i * n_lon + j + 1, i * n_lon + (j + 1) % n_lon + 1
            c, d = a + n_lon, b + n_lon
            faces.append("f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c))       # wound to face outwards
            faces.append("f %d//%d %d//%d %d//%d" % (b, b, d, d, c, c))
    (tmp_path / "ball.obj").write_text("\n".join(lines + ["s 1"] + faces) + "\n")
    text = """options { background color rgb <0.5, 0.6, 0.8> }
camera { origin <1.5, 1.2, 2.6> look_at <0, 0, 0> }
object "floor" { plane { origin <0, -1, 0> normal <0, 1, 0> } material { lambertian texture { solid color rgb <0.7, 0.7, 0.6> } } }
object "ball" { mesh { file "ball.obj" } material { metal fuzz 0.05 texture { solid color rgb <0.9, 0.8, 0.7> } } }
"""
    o = rt.Options(width=160, height=120, samples=1, seed=2, max_depth=6)
    sc = rt.parse(o, text, str(tmp_path))
    assert sc.desc.contents.n_triangles == 2 * n_lat * n_lon
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert (rids == 1).mean() > 0.2
    rgb, ids, st = rt.DeviceScene(sc).render(o)
    _check(parity.compare(rgb, ids, ref, rids, 1), st, rst)


def test_structure_is_chosen_per_mesh(rt, tmp_path, monkeypatch):
    """A crowded mesh (the stadium of test_crowded_mesh_teapot_in_a_stadium: kd-tree) beside an ordinary one
    (LinkedTorus.obj: grid) in one scene: each is walked through its own structure, k_mesh<grid> and k_mesh<kd>
    run one after the other over the same records, and the image is the oracle's - and bit for bit the image
    with either structure forced on every mesh."""
    import shutil
    rng = np.random.default_rng(3)
    lines = ["o Stadium", "v -30 0 -30", "v 30 0 -30", "v 30 0 30", "v -30 0 30", "vn 0 1 0"]
    faces = ["f 1//1 3//1 2//1", "f 1//1 4//1 3//1"]
    nv = 4
    for _ in range(1500):
        c = np.array([0.0, 0.4, 0.0]) + rng.uniform(-0.15, 0.15, 3)
        p = c + rng.uniform(-0.03, 0.03, (3, 3))
        for q in p:
            lines.append("v %.6f %.6f %.6f" % tuple(q))
        faces.append("f %d//1 %d//1 %d//1" % (nv + 1, nv + 2, nv + 3))
        nv += 3
    (tmp_path / "stadium.obj").write_text("\n".join(lines + ["s 1"] + faces) + "\n")
    shutil.copy(os.path.join(parity.ASSETS, "LinkedTorus.obj"), tmp_path / "LinkedTorus.obj")
    text = """options { background color rgb <0.4, 0.5, 0.7> }
camera { origin <1.8, 1.6, 3.4> look_at <0.6, 0.5, 0> }
object "light" { sphere { origin <4, 6, 3> radius 2 } material { diffuse_light intensity 4 texture { solid color white } } }
object "stadium" { mesh { file "stadium.obj" } material { lambertian texture { solid color rgb <0.8, 0.6, 0.4> } } }
object "tori" { mesh { file "LinkedTorus.obj" transform { scale <0.4, 0.4, 0.4> translate <1.3, 0.6, 0> } } material { metal fuzz 0.1 texture { solid color rgb <0.9, 0.9, 0.7> } } }
"""
    o = rt.Options(width=240, height=180, samples=1, seed=2)
    sc = rt.parse(o, text, str(tmp_path))
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert (rids == 1).mean() > 0.3 and (rids == 2).mean() > 0.015
    images = []
    for accel in (None, "grid", "kd"):
        if accel:
            monkeypatch.setenv("RT_ACCEL", accel)
        dev = rt.DeviceScene(sc)
        rgb, ids, st = dev.render(o)
        _check(parity.compare(rgb, ids, ref, rids, st.strata), st, rst)
        images.append(rgb)
        ro, rd = _random_rays(90000, 8, np.array([0.6, 0.5, 0.0]), 2.0)     # long queue: the walks, not the wave-per-ray kernel
        got, _ = dev.trace_rays(ro, rd, seed=4)
        want, _ = oracle_lib.trace_rays(sc.desc, ro, rd, seed=4)
        assert (got["object"] == want["object"]).mean() >= ID_MATCH_MIN
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[1], images[2])


def test_complex_v2_hand_translation(rt):
    """complex_v2.scene: the reference's complex.scene translated by hand into the grammar the reference parses
    today (the original is in an obsolete dialect that the reference itself rejects; the file's header lists the
    choices made).  Not a reference input - a stress case: a closed room (paths run to the depth cap), a mesh, an
    image texture, metal and glass, 14 objects."""
    o = rt.Options(width=200, height=150, samples=4, seed=3)
    sc = rt.parse_file(o, parity.scene_path("complex_v2"), parity.ASSETS)
    assert sc.n_objects == 14
    rgb, ids, st = rt.render(o, sc)
    ref, rids, rst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert {8, 9, 10, 11, 12, 13} <= set(np.unique(rids).tolist())     # earth, tori, the three spheres, the glass ball
    _check(parity.compare(rgb, ids, ref, rids, st.strata), st, rst)


def test_mesh_walk_stress(rt):
    """tools/mesh_stress.py: every shipped mesh under random object transforms (non-uniform scales, rotations,
    far translations), 40 000 rays per configuration from far away, from inside the bounds, axis-parallel - long
    queues, so the grid walk with its f32 pre-test takes them: every ray hits the oracle's object at the oracle's
    t (1e-9) with the oracle's normal.  A pre-test that ever rejected a hit would show here."""
    import sys
    sys.path.insert(0, os.path.join(parity.ROOT, "tools"))
    import mesh_stress
    assert mesh_stress.run(9, 40000) == 0


def test_mesh_kernels_agree(rt, monkeypatch):
    """k_mesh_one (one grid job in the scene: PREPARE / FINISH batches around the walk) and k_mesh (RT_MESH_POOL=0)
    write the same records: images, ids and ray counts bit for bit, on mesh.scene (long and short queues: 640x360
    goes from the grid walk through the wave-per-ray kernel to the tail) and on traced rays."""
    o = rt.Options(width=640, height=360, samples=4, seed=9)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    ro, rd = _random_rays(120000, 5, np.array([0.0, 1.0, 0.0]), 2.5)
    out = []
    for pool in ("1", "0"):
        monkeypatch.setenv("RT_MESH_POOL", pool)
        dev = rt.DeviceScene(sc)
        rgb, ids, st = dev.render(o)
        got, _ = dev.trace_rays(ro, rd, seed=4)
        out.append((rgb, ids, st.rays, got))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    for k in ("object", "t", "n", "uv"):
        assert np.array_equal(out[0][3][k], out[1][3][k]), k


def test_interval_lists_in_lds_and_in_scratch_agree(rt, monkeypatch):
    """The CSG / fog variants of k_intersect and k_tail keep a ray's interval lists in LDS when the scene's bounds on them
    fit (rt_scene_create; LdsLists, rt_dev_shapes.h) and in scratch memory otherwise (RT_LDS_LISTS=0 forces that): the
    same interval algebra on another store, so frames, ids and ray counts are equal bit for bit - on the shipped CSG and
    fog scenes, on fuzz scenes with transforms at every level of their trees, and on traced ray batches."""
    cases = [(parity.scene_path(n), None) for n in ("csg", "room", "fog_light", "csg_notorus")]
    cases += [(None, fuzz_scenes.scene_text(seed, torus=(seed % 2 == 0))) for seed in (61, 64, 70, 76, 78)]
    for path, text in cases:
        o = rt.Options(width=200, height=150, samples=4, seed=5)
        sc = rt.parse_file(o, path, parity.ASSETS) if path else rt.parse(o, text, parity.ASSETS)
        ro, rd = _random_rays(20000, 6, np.array([0.0, 1.0, 0.0]), 3.0)
        out = []
        for lds in ("1", "0"):
            monkeypatch.setenv("RT_LDS_LISTS", lds)
            dev = rt.DeviceScene(sc)
            rgb, ids, st = dev.render(o)
            got, _ = dev.trace_rays(ro, rd, seed=4)
            out.append((rgb, ids, st.rays, list(st.guards), st.interval_overflow, got))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), path or text[:80]
        assert out[0][2:5] == out[1][2:5] and out[0][4] == 0
        for k in ("object", "t", "n", "uv"):
            assert np.array_equal(out[0][5][k], out[1][5][k]), k


def test_queue_regions_that_fill_up_pass_their_entries_on(rt, monkeypatch):
    """The ray queues and the mesh record queue lie in eight regions with a counter word each (QRegions, rt_queue.h);
    appenders go round them, and one that meets a full region moves on to the next.  With RT_Q_SKEW=1 every append - per
    workgroup in the camera pass and k_shade, per wave (staged in LDS or from the registers) in the bounce passes -
    tries region 0 first, so the regions fill one after the other and most appends meet full ones, many of them split
    over two regions.  The order of a queue does not reach the image: frames, ids, ray counts and traced hits are equal
    bit for bit, at a frame size that fills the regions (a fresh scene's regions hold an eighth of its slots each)."""
    cases = [(parity.scene_path(n), None) for n in ("mesh", "cornell_box", "csg", "fog_light")]
    cases += [(None, fuzz_scenes.scene_text(seed, torus=False)) for seed in (3, 17)]
    for path, text in cases:
        o = rt.Options(width=333, height=211, samples=4, seed=9)
        sc = rt.parse_file(o, path, parity.ASSETS) if path else rt.parse(o, text, parity.ASSETS)
        ro, rd = _random_rays(30000, 8, np.array([0.0, 1.0, 0.0]), 3.0)
        out = []
        for skew in ("0", "1"):
            monkeypatch.setenv("RT_Q_SKEW", skew)
            dev = rt.DeviceScene(sc)
            rgb, ids, st = dev.render(o)
            got, _ = dev.trace_rays(ro, rd, seed=4)
            out.append((rgb, ids, st.rays, list(st.guards), st.mesh_records, got))
        monkeypatch.setenv("RT_Q_SKEW", "0")
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]), path or text[:80]
        assert out[0][2:5] == out[1][2:5], (path, out[0][2:5], out[1][2:5])
        for k in ("object", "t", "n", "uv"):
            assert np.array_equal(out[0][5][k], out[1][5][k]), k


def test_mesh_part_boxes_only_drop_rays_that_miss(rt, monkeypatch):
    """DevMesh::part (rt_flatten.cpp): k_intersect queues a ray for the mesh walk only if it enters one of up to four boxes
    that hold every triangle between them.  A ray it drops has no triangle to meet, so the frame with the boxes equals
    the frame without (RT_NO_MESH_PARTS=1) bit for bit - also under a rotated and scaled mesh object - while fewer rays
    go through the mesh queue."""
    texts = [open(parity.scene_path("mesh")).read(),
             'camera { origin <1.5, 2.5, 5> look_at <0, 0.8, 0> }\n'
             'object "light" { sphere { origin <4, 6, 3> radius 2 } material { diffuse_light intensity 4 texture { solid color white } } }\n'
             'object "floor" { plane { origin <0, -0.6, 0> normal <0, 1, 0> } material { lambertian texture { pattern { checkerboard color rgb <1, 1, 1> color rgb <0.2, 0.2, 0.6> 1 } } } }\n'
             'object "a" { mesh { file "LinkedTorus.obj" transform { scale <1.3, 0.7, 1.1> rotate_x 35 rotate_y 20 translate <0.2, 0.9, -0.3> } } material { metal fuzz 0.1 texture { solid color rgb <0.8, 0.8, 0.3> } } }\n',
             open(parity.scene_path("mesh")).read().replace("LinkedTorus.obj", "Monkey.obj")]
    for text in texts:
        o = rt.Options(width=320, height=200, samples=4, seed=11, stage_times=True)
        sc = rt.parse(o, text, parity.ASSETS)
        out = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("RT_NO_MESH_PARTS", "1")
            else:
                monkeypatch.delenv("RT_NO_MESH_PARTS", raising=False)
            dev = rt.DeviceScene(sc)
            rgb, ids, st = dev.render(o)
            out.append((rgb, ids, st.rays, st.mesh_records))
        assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
        assert out[0][3] <= out[1][3]
    assert out[0][3] > 0


def test_stats_name_the_mesh_kernel(rt, monkeypatch):
    """rt_stats.mesh_one_rays: all of mesh_rays in a scene with one mesh job (k_mesh_one walked the grid), none when the
    scene has two jobs (two mesh objects: k_mesh) or when RT_MESH_POOL=0 keeps k_mesh; mesh_records counts every ray that
    went through the mesh queue either way."""
    o = rt.Options(width=320, height=200, samples=1, seed=3, stage_times=True)  # the per-kernel fields of rt_stats
    one = rt.DeviceScene(rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS))
    _, _, st = one.render(o)
    assert st.mesh_rays > 0 and st.mesh_one_rays == st.mesh_rays and st.mesh_records >= st.mesh_rays
    text = ('camera { origin <0, 1, 6> look_at <0, 0.5, 0> }\n'
            'object "light" { sphere { origin <4, 6, 3> radius 2 } material { diffuse_light intensity 4 texture { solid color white } } }\n'
            'object "a" { mesh { file "LinkedTorus.obj" transform { translate <-1.5, 0, 0> } } material { lambertian texture { solid color white } } }\n'
            'object "b" { mesh { file "LinkedTorus.obj" transform { translate <1.5, 0, 0> } } material { lambertian texture { solid color white } } }\n')
    two = rt.DeviceScene(rt.parse(o, text, parity.ASSETS))
    _, _, st2 = two.render(o)
    assert st2.mesh_rays > 0 and st2.mesh_one_rays == 0
    monkeypatch.setenv("RT_MESH_POOL", "0")
    off = rt.DeviceScene(rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS))
    _, _, st3 = off.render(o)
    assert st3.mesh_rays == st.mesh_rays and st3.mesh_one_rays == 0 and st3.rays == st.rays


def test_tail_takeover_is_invisible(rt, monkeypatch):
    """Where the per-depth passes hand over to the one-launch tail changes nothing: k_tail_grid (one lane per path,
    grid walk or the wave on one ray) taking over after the first pass, at its default threshold, or never (all
    passes), and k_tail (RT_TAIL_GRID=0, one wave per path) give the same frame bit for bit and the same ray count;
    so does a second render of the same shape, whose passes are queued from the first one's counts without waiting
    for a queue length (the pass hint), and one with the hint switched off."""
    o = rt.Options(width=640, height=360, samples=1, seed=11)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    monkeypatch.setenv("RT_TAIL_MAX", "0")  # no tail: every level is a pass
    dev = rt.DeviceScene(sc)
    want, wids, wst = dev.render(o)
    assert wst.bounce_passes > 8
    for tail_max, hint in (("100000000", "1"), ("100000000", "1"), ("3000", "0"), (None, "1"), (None, "1"), (None, "0")):
        if tail_max is None:
            monkeypatch.delenv("RT_TAIL_MAX", raising=False)
        else:
            monkeypatch.setenv("RT_TAIL_MAX", tail_max)
        monkeypatch.setenv("RT_PASS_HINT", hint)
        rgb, ids, st = dev.render(o)
        assert np.array_equal(rgb, want) and np.array_equal(ids, wids), (tail_max, hint)
        assert st.rays == wst.rays and st.bounce_passes == wst.bounce_passes, (tail_max, hint)
    # another seed through the hinted passes: equal to a scene that follows the counts and uses k_tail
    o2 = rt.Options(width=640, height=360, samples=4, seed=12)
    rgb2, ids2, st2 = dev.render(o2)
    monkeypatch.setenv("RT_TAIL_GRID", "0")
    monkeypatch.setenv("RT_PASS_HINT", "0")
    old = rt.DeviceScene(sc)
    rgb3, ids3, st3 = old.render(o2)
    assert np.array_equal(rgb2, rgb3) and np.array_equal(ids2, ids3) and st2.rays == st3.rays
    # a transformed mesh object (rotation + non-uniform scale: the job's maps are not the identity)
    text = open(parity.scene_path("mesh")).read().replace("translate <0, 1, 0>", "rotate_x 20 rotate_y 30 scale <1.2, 0.8, 1> translate <0, 1, 0>")
    assert "rotate_x 20" in text
    sc4 = rt.parse(o, text, parity.ASSETS)
    r4, i4, s4 = old.__class__(sc4).render(o)
    monkeypatch.delenv("RT_TAIL_GRID")
    monkeypatch.setenv("RT_TAIL_MAX", "100000000")
    r5, i5, s5 = rt.DeviceScene(sc4).render(o)
    assert np.array_equal(r4, r5) and np.array_equal(i4, i5) and s4.rays == s5.rays


@pytest.mark.parametrize("name", ["csg", "room", "fog_light", "cylinder"])
def test_short_interval_lists_change_nothing(rt, monkeypatch, name):
    """Scenes whose CSG trees provably never hold more than DEV_SMALL_INTERVALS intervals in a list (the flattener
    bounds it from the tree) can run kernels with short lists in scratch memory (a build with -DRT_CSG_SMALL_KERNELS=1;
    RT_CSG_SMALL=0 keeps the full-size ones; the default build has only those and this test then compares like with
    like).  Same frame bit for bit, same rays, no list ever full."""
    o = rt.Options(width=200, height=150, samples=4, seed=6)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    rgb, ids, st = rt.DeviceScene(sc).render(o)
    monkeypatch.setenv("RT_CSG_SMALL", "0")
    rgb2, ids2, st2 = rt.DeviceScene(sc).render(o)
    assert np.array_equal(rgb, rgb2) and np.array_equal(ids, ids2)
    assert st.rays == st2.rays and st.interval_overflow == 0 and st2.interval_overflow == 0
    assert list(st.guards) == list(st2.guards)


def test_wide_union_keeps_full_size_lists(rt):
    """A union of six spheres along the ray holds six intervals: beyond the short lists, so the scene runs the
    full-size kernels and nothing overflows (the oracle sees every sphere)."""
    spheres = ["sphere { origin <%g, 1, 0> radius 0.4 }" % (x * 1.0 - 2.5) for x in range(6)]
    shape = spheres[0]
    for sph in spheres[1:]:
        shape = "union { %s %s }" % (shape, sph)
    text = ("camera { origin <-6, 1, 0> look_at <0, 1, 0> }\n"
            "object { sphere { origin <0, 9, 0> radius 3 } material { diffuse_light intensity 4 texture { solid color white } } }\n"
            "object { %s material { dielectric ior 1.5 } }\n" % shape)
    o = rt.Options(width=160, height=120, samples=1, seed=2)
    sc = rt.parse(o, text, parity.ASSETS)
    rgb, ids, st = rt.DeviceScene(sc).render(o)
    want, wids, wst = oracle_lib.render(sc.desc, o.to_c(), threads=16)
    assert st.interval_overflow == 0
    _check(parity.compare(rgb, ids, want, wids, st.strata), st, wst)
    assert (ids == 1).mean() > 0.01  # the row of spheres is in view, one behind the other
