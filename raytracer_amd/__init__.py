"""raytracer_amd — MI355X-native rendering core for the doxxx/raytracer path tracer.

The package is a thin host-side mirror of the reference's `sdl::parse` /
`system::render` interface over the C ABI of librt_amd.so (include/rt_api.h).
There is no CPU fallback: importing the package without the built library
raises ImportError, and rendering without a GPU raises RuntimeError.
"""
from . import _capi

lib = _capi.load()

from . import sdl, system  # noqa: E402
from .sdl import Scene, parse, parse_file  # noqa: E402
from .system import (Options, RenderProgress, render, DeviceScene, DeviceProgress, FileRenderProgress,  # noqa: E402
                     render_multi, tonemap, tonemap_device, write_png)

__all__ = ["lib", "sdl", "system", "Scene", "parse", "parse_file", "Options", "RenderProgress", "render",
           "DeviceScene", "DeviceProgress", "FileRenderProgress", "render_multi", "tonemap", "tonemap_device", "write_png"]
