"""Host-side logic of the product that needs no GPU: the C ABI surface, the scene
parser (sdl_grammar.rs restated), the OBJ loader, shard geometry, tone mapping."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib
import parity
from raytracer_amd import _capi

ROOT = parity.ROOT


def test_library_exports_every_declared_symbol(rt):
    """Every function include/rt_api.h declares is exported by librt_amd.so and bound in _capi."""
    with open(os.path.join(ROOT, "include", "rt_api.h")) as f:
        text = f.read()
    declared = set(re.findall(r"^\s*(?:const\s+)?[a-z_0-9]+\s*\*?\s*(rt_[a-z_0-9]+)\s*\(", text, re.M))
    declared -= {"rt_progress_cb"}
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    for name in declared:
        assert getattr(rt.lib, name) is not None


def test_struct_sizes_match_header():
    assert C.sizeof(_capi.rt_options) == 48
    assert C.sizeof(_capi.rt_shape_node) == 16 + 8 * 8 + 2 * 128
    assert C.sizeof(_capi.rt_hit) == 56
    assert C.sizeof(_capi.rt_stats) == 136


def _c_layout():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import layout_table
    return layout_table, layout_table.layout()


def test_layout_contract_ctypes_rust_and_docs():
    """The binding contract of the boundary: what the C compiler makes of every struct in include/rt_api.h
    (tests/helpers/layout_probe.c: sizeof, offsetof) equals (1) the ctypes mirror this repository exercises,
    (2) the #[repr(C)] declarations of the Rust crate under ffi/ (parsed here and laid out by the C rules, since
    no Rust toolchain exists in this image), (3) the table printed in INTEGRATION.md."""
    layout_table, lay = _c_layout()
    # (1) ctypes
    for name, s in lay.items():
        ct = getattr(_capi, name)
        assert C.sizeof(ct) == s["size"], name
        assert [f[0] for f in ct._fields_] == list(s["fields"]), name
        for f, (off, size) in s["fields"].items():
            d = getattr(ct, f)
            assert (d.offset, d.size) == (off, size), (name, f)
    # (2) Rust: #[repr(C)] pub struct NAME { pub field: TYPE, ... }
    with open(os.path.join(ROOT, "ffi", "src", "lib.rs")) as f:
        rust = f.read()
    prim = {"i32": (4, 4), "u32": (4, 4), "u64": (8, 8), "f64": (8, 8), "u8": (1, 1)}
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", rust, re.S):
        structs[m.group(1)] = re.findall(r"pub (\w+): ([^,\n]+),", m.group(2))

    def size_align(ty):
        ty = ty.strip()
        if ty in prim:
            return prim[ty]
        if ty.startswith("*") or ty in ("rt_stratum_cb", "rt_snapshot_cb", "rt_progress_cb"):
            return 8, 8                                      # pointers and Option<extern "C" fn>
        arr = re.match(r"\[(\w+); (\d+)\]", ty)
        if arr:
            sz, al = size_align(arr.group(1))
            return sz * int(arr.group(2)), al
        fields = structs[ty]
        off, al = 0, 1
        for _, t in fields:
            s, a = size_align(t)
            off = (off + a - 1) // a * a + s
            al = max(al, a)
        return (off + al - 1) // al * al, al

    for name, s in lay.items():
        assert name in structs, "ffi/src/lib.rs lacks %s" % name
        assert [f for f, _ in structs[name]] == list(s["fields"]), name
        off = 0
        for f, t in structs[name]:
            sz, al = size_align(t)
            off = (off + al - 1) // al * al
            assert [off, sz] == s["fields"][f], (name, f, off, sz)
            off += sz
        assert size_align(name)[0] == s["size"], name
    # every function of the header is declared in the crate's extern block
    for fn in _capi.SIGNATURES:
        assert re.search(r"pub fn %s\(" % fn, rust), fn
    # (3) the table in INTEGRATION.md
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    assert layout_table.table(lay) in doc, "INTEGRATION.md's layout table is stale: python tools/layout_table.py"


def test_build_recipes_list_every_source():
    """ffi/build.rs (never compiled here: no Rust toolchain) must rebuild on exactly the files raytracer_amd/build.py
    builds from, and both lists must hold every file under csrc/ - a device header missing from either would leave a
    stale library behind an edit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_build_recipe", os.path.join(ROOT, "raytracer_amd", "build.py"))
    recipe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(recipe)
    with open(os.path.join(ROOT, "ffi", "build.rs")) as f:
        rs = f.read()
    rs_sources = re.findall(r'\("([\w.]+)", "(?:hip|c\+\+)"\)', rs)
    block = re.search(r"for header in \[(.*?)\]", rs, re.S).group(1)
    rs_headers = re.findall(r'"([^"]+)"', block)
    assert sorted(rs_sources) == sorted(recipe.SOURCES)
    assert sorted(rs_headers) == sorted(recipe.HEADERS), set(rs_headers) ^ set(recipe.HEADERS)
    on_disk = {n for n in os.listdir(os.path.join(ROOT, "raytracer_amd", "csrc")) if n.endswith((".h", ".hip", ".cpp", ".inc"))}
    listed = set(recipe.SOURCES) | {h for h in recipe.HEADERS if not h.startswith("../")}
    assert on_disk == listed, on_disk ^ listed


# objects per scene, in file order
EXPECTED_OBJECTS = {"three-spheres": 11, "cornell_box": 8, "mesh": 3, "torus": 3, "csg": 3, "cylinder": 4,
                    "transformations": 3, "room": 5, "fog_light": 5}


@pytest.mark.parametrize("name", parity.SCENES)
def test_parse_shipped_scenes(rt, name):
    o = rt.Options(width=64, height=48)
    sc = rt.parse_file(o, parity.scene_path(name), parity.ASSETS)
    assert sc.n_objects == EXPECTED_OBJECTS[name]
    d = sc.desc.contents
    for k in range(d.n_objects):
        assert 0 <= d.objects[k].shape < d.n_nodes and 0 <= d.objects[k].material < d.n_materials


def test_complex_scene_is_rejected_like_the_reference(rt):
    """scenes/complex.scene uses an obsolete dialect (`location`, ...): sdl_grammar.rs:37-40 cannot parse it."""
    o = rt.Options(width=64, height=48)
    with pytest.raises(rt.sdl.ParseError):
        rt.parse_file(o, parity.scene_path("complex"), parity.ASSETS)


def test_mesh_scene_contents(rt):
    o = rt.Options(width=64, height=48)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)  # keep the owner alive
    d = sc.desc.contents
    assert d.n_triangles == 2304 and d.n_vertices == 1152 and d.n_normals == 1054 and d.n_meshes == 1
    tv = np.ctypeslib.as_array(d.tri_v, shape=(2304 * 3,))
    assert tv.min() == 0 and tv.max() == 1151
    # first face `f 1//1 13//2 14//3 2//4` fans into (0,12,13) and (0,13,1)
    assert list(tv[:6]) == [0, 12, 13, 0, 13, 1]
    root = d.nodes[d.objects[2].shape]
    assert root.kind == _capi.RT_SHAPE_COMPOSITE and root.child_b == 1
    assert root.o2w[13] == 1.0  # translate <0, 1, 0>


def test_monkey_obj_triangulation(rt, tmp_path):
    scene = 'camera { origin <0,0,5> look_at <0,0,0> }\nobject { mesh { file "Monkey.obj" } material { lambertian texture { solid color white } } }\n'
    sc = rt.parse(rt.Options(width=8, height=8), scene, parity.ASSETS)
    d = sc.desc.contents
    assert d.n_triangles == 468 * 2 + 32 and d.n_vertices == 507


def test_camera_and_defaults(rt):
    o = rt.Options(width=640, height=480)
    sc = rt.parse_file(o, parity.scene_path("cornell_box"), parity.ASSETS)
    d = sc.desc.contents
    assert abs(d.camera.fov_factor - np.tan(np.radians(20.0))) < 1e-15  # fov 40
    assert list(d.camera.camera_to_world)[12:15] == [278.0, 273.0, -800.0]
    sc2 = rt.parse_file(o, parity.scene_path("torus"), parity.ASSETS)
    d2 = sc2.desc.contents
    assert abs(d2.camera.fov_factor - np.tan(np.radians(30.0))) < 1e-15  # default fov 60 (sdl_grammar.rs:39)
    assert tuple(d2.background) == (0.0, 0.0, 0.0)


CAM = "camera { origin <0,0,5> look_at <0,0,0> }\n"
MAT = " material { lambertian texture { solid color white } } "


def test_grammar_details(rt):
    o = rt.Options(width=8, height=8)
    # short vectors, named directions, colour spelling, comments, negative and exponent floats
    text = ("options { background colour rgb <0.25> }\n" + CAM +
            "// c\nobject \"a\" { /* b */ plane { origin <1> normal up }" + MAT + "}\n"
            "object { sphere { origin <-1.5, 2.0e1, 3> radius 2 transform { scale <2> rotate_y 90 translate left } }" + MAT + "}\n")
    sc = rt.parse(o, text, ".")
    d = sc.desc.contents
    assert tuple(d.background) == (0.25, 0.25, 0.25)
    assert list(d.nodes[0].params)[:6] == [1.0, 1.0, 1.0, 0.0, 1.0, 0.0]
    assert list(d.nodes[1].params)[:4] == [-1.5, 20.0, 3.0, 2.0]
    # scale then rotate then translate, folded left to right (sdl.rs:94-96)
    m = np.array(list(d.nodes[1].o2w)).reshape(4, 4)
    p = np.array([1.0, 0, 0, 1.0]) @ m
    assert np.allclose(p[:3], [-1.0, 0.0, -2.0], atol=1e-12)
    w2o = np.array(list(d.nodes[1].w2o)).reshape(4, 4)
    assert np.allclose(m @ w2o, np.eye(4), atol=1e-12)


@pytest.mark.parametrize("text", [
    CAM,                                                            # no objects
    "  options { background color black }\n" + CAM + "object { sphere { }" + MAT + "}",  # options only at offset 0
    CAM + "object { sphere { radius .5 }" + MAT + "}",             # `.5` is not a float literal
    CAM + "object { sphere { radius 1e3 }" + MAT + "}",            # exponent needs a fraction
    CAM + "object { torus { radius 1 }" + MAT + "}",               # torus needs both radii
    CAM + "object { sphere { radius 1 origin <0> }" + MAT + "}",   # field order is fixed
    CAM + "object { sphere { } material { plastic } }",
    CAM + "object { sphere { }" + MAT + "} trailing",
    CAM + "/* a * b */ object { sphere { }" + MAT + "}",           # block comments cannot hold '*'
])
def test_grammar_rejections(rt, text):
    with pytest.raises(rt.sdl.ParseError):
        rt.parse(rt.Options(width=8, height=8), text, ".")


def test_missing_assets_raise(rt):
    o = rt.Options(width=8, height=8)
    with pytest.raises(FileNotFoundError):
        rt.parse(o, CAM + 'object { mesh { file "nope.obj" }' + MAT + "}", parity.ASSETS)
    with pytest.raises(FileNotFoundError):
        rt.parse(o, CAM + 'object { sphere { } material { lambertian texture { image "nope.jpg" 1.0 } } }', parity.ASSETS)


def test_earth_texture_bytes(rt):
    o = rt.Options(width=8, height=8)
    sc = rt.parse_file(o, parity.scene_path("three-spheres"), parity.ASSETS)
    d = sc.desc.contents
    assert d.n_images == 1 and d.images[0].width == 756 and d.images[0].height == 378
    assert [d.image_data[k] for k in range(3)] == [205, 206, 224]


@pytest.mark.parametrize("w,h,tile,count", [(100, 70, 16, 3), (3840, 2160, 64, 8), (64, 64, 64, 2), (33, 9, 8, 5)])
def test_shards_partition_the_image(rt, w, h, tile, count):
    from raytracer_amd import system
    o = rt.Options(width=w, height=h, tile_size=tile)
    seen = np.zeros(w * h, dtype=np.int32)
    sizes = []
    for r in range(count):
        m = system.shard_map(o, r, count)
        assert len(m) == system.shard_pixels(o, r, count)
        sizes.append(len(m))
        np.add.at(seen, m[m >= 0], 1)
    assert (seen == 1).all()
    assert max(sizes) - min(sizes) <= tile * tile  # interleaved tiles: at most one tile apart
    # one wave = one 8x8 block
    m0 = system.shard_map(o, 0, count)
    blk = m0[:64]
    if (blk >= 0).all():
        ys, xs = blk // w, blk % w
        assert xs.max() - xs.min() == 7 and ys.max() - ys.min() == 7


def test_tonemap_matches_reference_pixel_math(rt):
    """main.rs:201-222 incl. its divide-by-(N+1) and saturating casts."""
    vals = np.array([[0.0, 0.5, 2.0], [8.0, -1.0, np.nan], [1e-9, 1.0, 0.25]])
    out = rt.tonemap(vals, 2.0)
    ref = oracle_lib.tonemap(vals, 2.0)
    assert np.array_equal(out, ref)
    assert list(out[0]) == [0, int(np.sqrt(0.25) * 255.0), 255]
    assert list(out[1]) == [255, 255, 255] or list(out[1])[0] == 255


def _texture_bytes(rt, base_dir, name):
    text = ('camera { origin <0,0,5> look_at <0,0,0> }\nobject "x" { sphere { origin <0,0,0> radius 1 } '
            'material { lambertian texture { image "%s" 1.0 } } }\n' % name)
    sc = rt.parse(rt.Options(width=8, height=8), text, str(base_dir))
    d = sc.desc.contents
    im = d.images[0]
    buf = np.ctypeslib.as_array(d.image_data, (im.offset + im.width * im.height * 3,))[im.offset:]
    return buf.reshape(im.height, im.width, 3).astype(int)


def test_jpeg_textures_decode_like_libjpeg(rt, tmp_path):
    """`image "x.jpg"` (sdl.rs:52, image 0.14 / jpeg-decoder 0.1.22 in the reference): host_jpeg.cpp decodes
    baseline files.  Decoders differ in the rounding of IDCT, upsampling and colour conversion, so the check is
    against PIL (libjpeg) within 3 levels and 0.2 on average; texel parity with the reference is unpinned."""
    from PIL import Image
    src = os.path.join(parity.ASSETS, "jpeg", "earth.jpg")           # the reference's own texture
    ref = np.asarray(Image.open(src).convert("RGB")).astype(int)
    got = _texture_bytes(rt, os.path.dirname(src), "earth.jpg")
    assert got.shape == ref.shape == (378, 756, 3)
    d = np.abs(got - ref)
    assert d.max() <= 3 and d.mean() < 0.1
    # 4:4:4, 4:2:2, 4:2:0, odd sizes, restart intervals, greyscale
    cases = [(64, 48, 0, 90, {}), (67, 45, 1, 85, {}), (123, 77, 2, 75, {}), (33, 17, 2, 95, {}),
             (200, 100, 2, 50, {"restart_marker_blocks": 7}), (16, 16, 0, 100, {}), (1, 1, 2, 80, {})]
    for w, h, sub, q, kw in cases:
        im = Image.fromarray(ref[:h * 3:3, :w * 3:3][:h, :w].astype(np.uint8))
        try:
            im.save(tmp_path / "t.jpg", quality=q, subsampling=sub, **kw)
        except TypeError:
            im.save(tmp_path / "t.jpg", quality=q, subsampling=sub)
        want = np.asarray(Image.open(tmp_path / "t.jpg").convert("RGB")).astype(int)
        d = np.abs(_texture_bytes(rt, tmp_path, "t.jpg") - want)
        assert d.max() <= 3 and d.mean() < 0.2, (w, h, sub, q, d.max(), d.mean())
    Image.fromarray(ref[:40, :50].astype(np.uint8)).convert("L").save(tmp_path / "g.jpg", quality=90)
    want = np.asarray(Image.open(tmp_path / "g.jpg").convert("RGB")).astype(int)
    assert np.abs(_texture_bytes(rt, tmp_path, "g.jpg") - want).max() <= 1
    # what it does not cover is refused with a reason, like the reference's expect("could not open image file")
    Image.fromarray(ref[:40, :50].astype(np.uint8)).save(tmp_path / "p.jpg", quality=90, progressive=True)
    with pytest.raises(Exception, match="progressive"):
        _texture_bytes(rt, tmp_path, "p.jpg")
    data = open(src, "rb").read()
    open(tmp_path / "cut.jpg", "wb").write(data[:300])
    with pytest.raises(Exception, match="could not decode image file"):
        _texture_bytes(rt, tmp_path, "cut.jpg")
    open(tmp_path / "half.jpg", "wb").write(data[:len(data) // 2])      # the entropy-coded data ends early
    with pytest.raises(Exception, match="restart marker"):
        _texture_bytes(rt, tmp_path, "half.jpg")


def test_png_file_holds_the_pixels(rt, tmp_path):
    """write_render_result_to_file's file half (main.rs:219-227): an 8-bit RGB PNG, rows top to bottom."""
    from PIL import Image
    rng = np.random.default_rng(5)
    for w, h in ((1, 1), (37, 11), (640, 480)):
        px = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = tmp_path / ("t%dx%d.png" % (w, h))
        rt.write_png(path, px)
        with Image.open(path) as im:
            assert im.mode == "RGB" and im.size == (w, h)
            assert np.array_equal(np.asarray(im), px)
    with pytest.raises(RuntimeError, match="cannot open"):
        rt.write_png(tmp_path / "no_such_dir" / "x.png", px)
    with pytest.raises(ValueError):
        rt.write_png(tmp_path / "x.png", px[:, :, 0])


def test_no_gpu_is_a_loud_error(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = rt.parse_file(rt.Options(width=8, height=8), parity.scene_path("torus"), parity.ASSETS)
    with pytest.raises(RuntimeError, match="no HIP device"):
        rt.render(rt.Options(width=8, height=8), sc)


def test_unsupported_and_invalid_descriptions(rt):
    """Errors come back as status codes + rt_last_error, never as a crash."""
    import descs
    d = descs.Desc([{"kind": "sphere", "params": [0, 0, 0, 1]}])
    d.objects[0].material = 7
    h = C.c_void_p()
    rc = rt.lib.rt_scene_create(d.ptr, C.byref(h))
    assert rc == _capi.RT_ERR_INVALID and b"material" in rt.lib.rt_last_error()
    assert rt.lib.rt_scene_create(None, C.byref(h)) == _capi.RT_ERR_INVALID


def _copy_desc(sc):
    """A shallow copy of a parsed scene's rt_scene_desc whose mesh table is a private copy (so a test can damage it)."""
    d = _capi.rt_scene_desc()
    C.memmove(C.byref(d), sc.desc, C.sizeof(d))
    meshes = (_capi.rt_mesh * d.n_meshes)()
    C.memmove(meshes, d.meshes, C.sizeof(meshes))
    d.meshes = C.cast(meshes, C.POINTER(_capi.rt_mesh))
    return d, meshes


def test_scene_create_rejects_mesh_ranges_outside_the_pools(rt):
    """rt_scene_create validates what a host flatten pass hands over before anything is read through it: mesh
    vertex / normal / triangle ranges inside the pools, composite children inside child_list, null arrays with
    non-zero counts, the packed hit id's limits.  Bad descriptions come back as RT_ERR_INVALID / UNSUPPORTED with
    a message - never a crash, never an exception across the C ABI.  (No GPU needed: validation comes first; a
    valid description then fails with RT_ERR_DEVICE here.)"""
    o = rt.Options(width=64, height=48)
    sc = rt.parse_file(o, parity.scene_path("mesh"), parity.ASSETS)
    h = C.c_void_p()

    def create(d):
        rc = rt.lib.rt_scene_create(C.byref(d), C.byref(h))
        if rc == 0:
            rt.lib.rt_scene_destroy(h)
        return rc, rt.lib.rt_last_error().decode()

    d, meshes = _copy_desc(sc)
    rc, msg = create(d)
    assert rc in (0, _capi.RT_ERR_DEVICE), msg                     # the untouched copy is fine
    for field, pool in (("n_vertices", "vertex"), ("n_normals", "normal"), ("n_triangles", "triangle")):
        d, meshes = _copy_desc(sc)
        setattr(meshes[0], field, getattr(meshes[0], field) + 10 ** 6)
        rc, msg = create(d)
        assert rc == _capi.RT_ERR_INVALID and pool in msg, (field, rc, msg)
    for field in ("first_vertex", "first_normal", "first_triangle"):
        d, meshes = _copy_desc(sc)
        setattr(meshes[0], field, 0xFFFFFFF0)                      # first + count wraps in 32 bits
        rc, msg = create(d)
        assert rc == _capi.RT_ERR_INVALID, (field, rc, msg)
    d, meshes = _copy_desc(sc)
    d.n_child_list = 0                                             # the composite's children are gone
    rc, msg = create(d)
    assert rc == _capi.RT_ERR_INVALID and "child" in msg
    d, meshes = _copy_desc(sc)
    d.normals = None                                               # a count without an array
    rc, msg = create(d)
    assert rc == _capi.RT_ERR_INVALID and "normals" in msg
    d, meshes = _copy_desc(sc)
    d.n_triangles = d.n_triangles - 1                              # the pool is shorter than the mesh says
    rc, msg = create(d)
    assert rc == _capi.RT_ERR_INVALID and "triangle" in msg


def test_options_limits(rt):
    """check_options: the random-number key holds 16 bits of stratum and of depth, 32 of pixel."""
    def rc_of(**kw):
        o = rt.Options(width=64, height=64, **kw).to_c()
        return rt.lib.rt_unshard_device(C.byref(o), 0, None, None, None, None, None)
    ok = (0, _capi.RT_ERR_DEVICE)                  # the options pass; without a GPU the launch behind them fails
    assert rc_of(samples=65536) in ok and rc_of(samples=65537) in ok      # floor(sqrt)^2 = 65536 strata: the last that fit
    assert rc_of(samples=257 * 257) == _capi.RT_ERR_INVALID
    assert b"strata" in rt.lib.rt_last_error()
    assert rc_of(max_depth=65535) in ok and rc_of(max_depth=65536) == _capi.RT_ERR_INVALID


def test_greyscale_jpeg_with_sampling_factors(rt, tmp_path):
    """A one-component JPEG whose frame header says 2x2 sampling is still one 8x8 block per MCU in raster order
    (T.81 A.2.2: a scan with one component is not interleaved).  PIL writes 1x1; the SOF0 sampling byte is
    patched to 0x22 here, which changes nothing for a conforming decoder."""
    from PIL import Image
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:45, 0:70]
    img = (127 + 80 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + rng.normal(0, 3, (45, 70))).clip(0, 255).astype(np.uint8)
    Image.fromarray(img, "L").save(tmp_path / "grey.jpg", quality=92)
    raw = bytearray((tmp_path / "grey.jpg").read_bytes())
    k = raw.find(b"\xff\xc0")
    assert k > 0 and raw[k + 9] == 1 and raw[k + 11] == 0x11          # SOF0, one component, sampling 1x1
    raw[k + 11] = 0x22
    (tmp_path / "grey22.jpg").write_bytes(bytes(raw))
    with Image.open(tmp_path / "grey22.jpg") as im:
        want = np.asarray(im.convert("RGB")).astype(int)
    text = ('camera { origin <0, 0, 5> look_at <0, 0, 0> }\nobject { sphere { origin <0, 0, 0> radius 1 } '
            'material { lambertian texture { image "grey22.jpg" 1.0 } } }\n')
    o = rt.Options(width=16, height=16)
    sc = rt.parse(o, text, str(tmp_path))
    d = sc.desc.contents
    assert d.n_images == 1 and d.images[0].width == 70 and d.images[0].height == 45
    got = np.ctypeslib.as_array(d.image_data, shape=(d.n_image_bytes,))[:70 * 45 * 3].reshape(45, 70, 3).astype(int)
    assert np.abs(got - want).max() <= 3 and np.abs(got - want).mean() < 0.5


# ---------------------------------------------------------------- the f32 pre-test of the mesh kernels, on the host
def _pretest_probe():
    """tests/helpers/pretest_probe.cpp: the product's flattener and f64 triangle test as host code, around a host copy
    of the pre-test's f32 arithmetic (built here with g++, -ffp-contract=off)."""
    import subprocess
    src = os.path.join(parity.ROOT, "tests", "helpers", "pretest_probe.cpp")
    out = os.path.join(parity.ROOT, "tests", "helpers", "_pretest_probe.so")
    csrc = os.path.join(parity.ROOT, "raytracer_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("rt_flatten.cpp", "rt_flatten.h", "rt_device.h", "rt_dev_shapes.h", "rt_dev_math.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                               "-I" + os.path.join(parity.ROOT, "tools", "host_shim"), "-I" + csrc,
                               "-I" + os.path.join(parity.ROOT, "include"), src, os.path.join(csrc, "rt_flatten.cpp"), "-o", out])
    lib = C.CDLL(out)
    lib.probe_pretest.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.probe_pretest.restype = C.c_int
    return lib


def test_f32_pretest_never_rejects_a_hit(rt, tmp_path):
    """Stage 1 of the mesh kernels' triangle test (rt_k_mesh_grid_round.inc) may reject a triangle only when the
    reference's f64 test (stage 2, tri_hit_v) would reject it too.  The probe runs both on the host, from the
    flattener's own records, for random rays through every occupied grid cell: the shipped meshes, and LinkedTorus.obj
    with its vertices scaled by 1e-3 and 1e3 and moved 1e4 away from the origin (the records are relative to the cell
    centre and the error bound to the cell's size, so none of these may matter) - origins in the cell, 10 and 1000
    units back, a quarter of the rays entering through a face, a quarter aimed at an edge or a corner of one of the
    cell's triangles (where the f64 test decides by the last bit: with the error bound set to zero the probe counts
    a thousand missed hits on LinkedTorus.obj), some axis-parallel.  No hit is missed in up to 2 M
    triangle tests per mesh, and the pre-test passes little more than the hits (it is worth having)."""
    lib = _pretest_probe()
    o = rt.Options(width=8, height=8)
    src = open(os.path.join(parity.ASSETS, "LinkedTorus.obj")).read().splitlines()

    def variant(name, scale, shift):
        with open(tmp_path / name, "w") as f:
            for line in src:
                if line.startswith("v "):
                    x, y, z = [float(v) for v in line.split()[1:4]]
                    line = "v %.17g %.17g %.17g" % (x * scale[0] + shift[0], y * scale[1] + shift[1], z * scale[2] + shift[2])
                f.write(line + "\n")

    variant("small.obj", (1e-3, 2e-3, 1.5e-3), (0.0, 0.0, 0.0))
    variant("large.obj", (1e3, 3e2, 2e3), (0.0, 0.0, 0.0))
    variant("far.obj", (1.0, 1.0, 1.0), (1e4, -2e4, 5e3))
    variant("flat.obj", (1.0, 1e-4, 1.0), (3.0, 4.0, 5.0))
    cases = [(parity.ASSETS, "LinkedTorus.obj"), (parity.ASSETS, "Monkey.obj"), (parity.ASSETS, "two_parts.obj"),
             (str(tmp_path), "small.obj"), (str(tmp_path), "large.obj"), (str(tmp_path), "far.obj"), (str(tmp_path), "flat.obj")]
    for k, (base, mesh) in enumerate(cases):
        text = ('camera { origin <0, 0, 9> look_at <0, 0, 0> }\n'
                'object "m" { mesh { file "%s" transform { rotate_y 30 translate <0, 1, 0> } } '
                'material { lambertian texture { solid color white } } }\n' % mesh)
        sc = rt.parse(o, text, base)
        out = (C.c_uint64 * 8)()
        assert lib.probe_pretest(C.cast(sc.desc, C.c_void_p), 48, 100 + k, out) == 0
        cells, tested, hits, passed, missed, jobs, aimed, _ = [int(v) for v in out]
        print(mesh, list(out))
        assert jobs >= 1 and cells > 20 and tested > (100000 if cells > 200 else 5000) and hits > tested // 200, (mesh, list(out))
        assert missed == 0 and aimed >= 20, (mesh, list(out))
        assert passed < 2.0 * hits, (mesh, list(out))


def test_flattener_bounds_and_part_boxes(rt):
    """What rt_flatten.cpp decides about bounds (host code, probed through tests/helpers/pretest_probe.cpp):
    * an expensive object carries the world bounds of the primitives a ray MUST meet - a difference its first
      operand's, an intersection the smaller operand's, a union both (csg.scene's object: the cube, not the hull of
      cube, spheres and torus); bit 1 of has_bounds (no hit lies before the bounds) only without a union / difference
      that has a rotation or scale of its own;
    * a mesh carries up to four part boxes that hold every triangle between them, with less surface than its bounds,
      and world-space boxes that contain them under the object's frame;
    * the scene's bounds on its interval lists (what decides whether they live in LDS)."""
    lib = _pretest_probe()
    lib.probe_flat_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    lib.probe_flat_bounds.restype = C.c_int
    o = rt.Options(width=8, height=8)

    def probe(sc):
        iout, dout = (C.c_uint64 * 64)(), (C.c_double * 512)()
        assert lib.probe_flat_bounds(C.cast(sc.desc, C.c_void_p), iout, dout) == 0
        return [int(v) for v in iout], np.array(dout[:])

    def close(a, b, tol=1e-5):
        return np.allclose(a, b, rtol=0, atol=tol)

    # csg.scene: light, floor, thing = ((cube ^ sphere) - sphere) - torus
    i, d = probe(rt.parse_file(o, parity.scene_path("csg"), parity.ASSETS))
    assert i[0] == 3 and i[2] == 4 and i[3] == 2
    assert i[8 + 2] == 3 and close(d[12:18], [-1, 0, -1, 1, 2, 1])
    # room.scene: sky = sphere - sphere, earth, walls = (cube - cube) - cube, window, ball
    i, d = probe(rt.parse_file(o, parity.scene_path("room"), parity.ASSETS))
    assert i[2] == 3 and i[3] == 2
    assert i[8] == 3 and close(d[0:6], [-1000.1, -1900.1, -1000.1, 1000.1, 100.1, 1000.1], 1e-3)
    assert i[8 + 2] == 3 and close(d[12:18], [-5.1, -0.1, -5.1, 5.1, 8.1, 5.1])
    assert i[8 + 1] == 0 and i[8 + 3] == 0 and i[8 + 4] == 0  # plane, cube, sphere: cheap objects carry none
    i, d = probe(rt.parse_file(o, parity.scene_path("fog_light"), parity.ASSETS))
    assert i[2] == 2 and i[3] == 2
    text = ('camera { origin <0, 1, 6> look_at <0, 1, 0> }\n'
            'object "a" { difference { sphere { origin <0, 1, 0> radius 1 } cube { <-2, 1, -2> <2, 3, 2> } transform { scale <2, 1, 1> } } material { lambertian texture { solid color white } } }\n'
            'object "b" { union { sphere { origin <3, 1, 0> radius 1 } sphere { origin <5, 1, 0> radius 0.5 } } material { lambertian texture { solid color white } } }\n'
            'object "c" { intersection { sphere { origin <0, 5, 0> radius 3 } cube { <-1, 4, -1> <1, 6, 1> } } material { lambertian texture { solid color white } } }\n'
)
    i, d = probe(rt.parse(o, text, parity.ASSETS))
    assert i[8] == 1 and close(d[0:6], [-2, 0, -1, 2, 2, 1])        # scaled difference: its sphere, and hits may lie before it
    assert i[8 + 1] == 3 and close(d[6:12], [2, 0, -1, 5.5, 2, 1])  # union: both
    assert i[8 + 2] == 3 and close(d[12:18], [-1, 4, -1, 1, 6, 1])  # intersection: the cube (less surface than the sphere)
    # (a plane inside a CSG tree - no bounds then - is not in the grammar: sdl_grammar.rs allows solids only)
    # meshes
    for mesh, xf in (("LinkedTorus.obj", "translate <0, 1, 0>"), ("Monkey.obj", "scale <1.3, 0.7, 1.1> rotate_x 35 rotate_y 20 translate <0.2, 0.9, -0.3>"),
                     ("two_parts.obj", "rotate_z 40 translate <1, 2, 3>")):
        text = ('camera { origin <0, 0, 9> look_at <0, 0, 0> }\n'
                'object "m" { mesh { file "%s" transform { %s } } material { lambertian texture { solid color white } } }\n' % (mesh, xf))
        sc = rt.parse(o, text, parity.ASSETS)
        i, d = probe(sc)
        m = d[128:128 + 60]
        n_parts, outside = i[32], i[33]
        area = lambda b: (lambda e: e[0] * e[1] + e[1] * e[2] + e[2] * e[0])(b[3:6] - b[0:3])
        if mesh == "LinkedTorus.obj":
            assert 2 <= n_parts <= 4
        if n_parts:
            assert outside == 0, mesh
            parts = [m[6 + 6 * p:12 + 6 * p] for p in range(n_parts)]
            assert sum(area(p) for p in parts) < 0.85 * area(m[0:6])
            for p in parts:
                assert np.all(p[0:3] >= m[0:3] - 1e-6) and np.all(p[3:6] <= m[3:6] + 1e-6)
        # world boxes contain the object-space boxes' corners under the object's frame (row-vector o2w of the root node)
        o2w = np.array(sc.desc.contents.nodes[sc.desc.contents.objects[0].shape].o2w[:]).reshape(4, 4)
        for ob, wb in [(m[0:6], m[30:36])] + [(m[6 + 6 * p:12 + 6 * p], m[36 + 6 * p:42 + 6 * p]) for p in range(n_parts)]:
            for c in range(8):
                q = np.array([ob[3 * ((c >> k) & 1) + k] for k in range(3)] + [1.0])
                w = q @ o2w
                assert np.all(w[:3] >= wb[0:3]) and np.all(w[:3] <= wb[3:6]), (mesh, c)


def test_interval_algebra_is_the_same_through_both_list_stores():
    """csg_apply (with its sort, copy and push helpers, rt_dev_shapes.h) is written once against the member names of
    IvListT (a list per array: scratch memory on the device) and LdsList (a strided view with the interval counts packed
    into one word: LDS on the device).  tests/helpers/lds_list_probe.cpp runs it on the host through both, over every
    ordering of up to 3 x 3 intervals on a grid and 600 000 random pairs of lists (touching end points, full lists whose
    results overflow), all three operations, capacities 12 and 4: same verdict, same guard and overflow counts, same
    intervals and tags, and the neighbouring lanes' words untouched."""
    import subprocess
    src = os.path.join(parity.ROOT, "tests", "helpers", "lds_list_probe.cpp")
    out = os.path.join(parity.ROOT, "tests", "helpers", "_lds_list_probe.so")
    csrc = os.path.join(parity.ROOT, "raytracer_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("rt_device.h", "rt_dev_shapes.h", "rt_dev_math.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                               "-I" + os.path.join(parity.ROOT, "tools", "host_shim"), "-I" + csrc,
                               "-I" + os.path.join(parity.ROOT, "include"), src, "-o", out])
    lib = C.CDLL(out)
    lib.probe_list_stores.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.probe_list_stores.restype = C.c_int
    cases = C.c_uint64()
    assert lib.probe_list_stores(11, 300000, C.byref(cases)) == 0
    assert cases.value > 600000


def test_queue_regions_place_and_map_every_entry():
    """The ray queues and the mesh record queue lie in eight regions with a counter word each (rt_queue.h): appenders
    reserve room through region_reserve - going round the regions, on to the next when one is full - and readers map
    0 .. total-1 to positions through QMap.  tests/helpers/queue_probe.cpp runs that code on the host over random region
    sizes, fill levels (to the last entry), append sizes and starting regions (all at region 0 as with RT_Q_SKEW=1):
    every entry gets a position of its own inside a region, none is left without one while the regions have room, the
    map is one to one onto the appended positions in rising order, its per-block scalar path agrees with the per-entry
    one, and a queue filled by position maps every index to itself."""
    import subprocess
    src = os.path.join(parity.ROOT, "tests", "helpers", "queue_probe.cpp")
    out = os.path.join(parity.ROOT, "tests", "helpers", "_queue_probe.so")
    csrc = os.path.join(parity.ROOT, "raytracer_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("rt_queue.h", "rt_dev_math.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC",
                               "-I" + os.path.join(parity.ROOT, "tools", "host_shim"), "-I" + csrc,
                               "-I" + os.path.join(parity.ROOT, "include"), src, "-o", out])
    lib = C.CDLL(out)
    lib.probe_queue_regions.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.probe_queue_regions.restype = C.c_int
    entries = C.c_uint64()
    assert lib.probe_queue_regions(7, 4000, C.byref(entries)) == 0
    assert entries.value > 2000000

