"""maray_amd — MI355X (gfx950) per-pixel expression evaluator for maray scenes.

A thin ctypes binding over libmaray_hip.so (C ABI in include/maray_hip.h).
The compute path is the HIP library; there is no Python or CPU fallback: every
render call raises MarayError when the library or a gfx950 device is missing.
"""
from .api import (BACKEND_AUTO, BACKEND_JIT, BACKEND_TAPE, BACKEND_TAPE_SMEM, Context, MarayError, PinnedRaster, Scene, Tape, device_count, gen,
                  gen_cache_clear, gen_cache_info, gen_to_image, image_read, lib, lib_path, png_read, png_write, version)

__all__ = ['Scene', 'Tape', 'Context', 'PinnedRaster', 'MarayError', 'gen', 'gen_to_image', 'gen_cache_clear', 'gen_cache_info', 'device_count', 'lib', 'lib_path',
           'png_read', 'png_write', 'image_read', 'version', 'BACKEND_TAPE', 'BACKEND_TAPE_SMEM', 'BACKEND_JIT', 'BACKEND_AUTO']
