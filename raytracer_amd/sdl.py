"""Scene description language front end: mirror of reference src/sdl.rs.

`parse(options, text)` has the reference's signature (sdl.rs:36-38): the
options supply width/height to Camera::new.  Parsing itself happens in
librt_amd.so (csrc/host_scene.cpp)."""
import ctypes as C
import os

from . import _capi


class ParseError(ValueError):
    pass


class Scene:
    """Owns an rt_scene_desc produced by the parser (sdl.rs:18-22)."""

    def __init__(self, desc_ptr, lib):
        self._ptr = desc_ptr
        self._lib = lib

    @property
    def desc(self):
        return self._ptr

    @property
    def n_objects(self):
        return self._ptr.contents.n_objects

    @property
    def background_color(self):
        return tuple(self._ptr.contents.background)

    def close(self):
        if self._ptr is not None:
            self._lib.rt_scene_desc_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _raise(lib, rc):
    msg = lib.rt_last_error().decode("utf-8", "replace")
    if rc == _capi.RT_ERR_IO:
        raise FileNotFoundError(msg)
    raise ParseError(msg)


def parse(options, text, base_dir="."):
    """sdl::parse(&options, &text).  `base_dir` stands for the reference's CWD
    when resolving `mesh { file .. }` and `image ".."` paths."""
    from . import lib
    out = C.POINTER(_capi.rt_scene_desc)()
    rc = lib.rt_sdl_parse(text.encode("utf-8"), os.fspath(base_dir).encode(), int(options.width), int(options.height),
                          C.byref(out))
    if rc != 0:
        _raise(lib, rc)
    return Scene(out, lib)


def parse_file(options, path, base_dir=None):
    """main.rs:95-100: read the scene file and parse it."""
    from . import lib
    if base_dir is None:
        base_dir = os.path.dirname(os.path.abspath(path))
    out = C.POINTER(_capi.rt_scene_desc)()
    rc = lib.rt_sdl_parse_file(os.fspath(path).encode(), os.fspath(base_dir).encode(), int(options.width),
                               int(options.height), C.byref(out))
    if rc != 0:
        _raise(lib, rc)
    return Scene(out, lib)
