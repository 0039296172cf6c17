"""Texture FILE decoding for the importers (obj_io, gltf_io, crts_io): four 8-bit channels, rows top to bottom -- what
the reference's `stbi_load(..., 4)` hands back before its optional vertical flip (util/material.cpp:5-17; tinygltf and
the CRTS reader ask stb_image for four components too).

JPEG goes through the native decoder of libcrt_scene_io.so (csrc/jpeg_reader.cpp: stb_image's inverse DCT, chroma
up-sampling and YCbCr -> RGB arithmetic restated, bit for bit -- libjpeg-based decoders differ from it by up to 2/255);
everything else through Pillow, which matches stb_image bit for bit on PNG and TGA (tests/test_texture_decoders.py,
tests/test_importers_pinned.py). Without the native library (no C++ compiler on the host) JPEG falls back to Pillow, with
a warning: the documented <= 2/255 deviation then applies.
"""
import ctypes as C
import io
import subprocess
import sys

import numpy as np

_lib = None
_warned = False


def _native():
    global _lib
    if _lib is None:
        from . import build
        L = C.CDLL(build.build_scene_io())
        L.crt_image_decode_jpeg.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_uint8))]
        L.crt_image_decode_jpeg.restype = C.c_int
        L.crt_image_free.argtypes = [C.POINTER(C.c_uint8)]
        L.crt_image_free.restype = None
        L.crt_image_error.restype = C.c_char_p
        _lib = L
    return _lib


def decode_jpeg_rgba(data: bytes) -> np.ndarray:
    """(height, width, 4) uint8 of a JPEG file's bytes, decoded with the reference decoder's arithmetic."""
    L = _native()
    w, h, p = C.c_int32(), C.c_int32(), C.POINTER(C.c_uint8)()
    if L.crt_image_decode_jpeg(data, len(data), C.byref(w), C.byref(h), C.byref(p)) != 0:
        raise ValueError("JPEG: " + L.crt_image_error().decode())
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        L.crt_image_free(p)


def decode_rgba(data: bytes, pil_check=None) -> np.ndarray:
    """(height, width, 4) uint8 of an image file's bytes. pil_check(pil_image): the importer's own format checks."""
    global _warned
    if pil_check is None and data[:2] == b"\xff\xd8":
        try:
            return decode_jpeg_rgba(data)
        except (OSError, subprocess.CalledProcessError) as e:  # no native library on this host
            if not _warned:
                _warned = True
                print(f"[chameleonrt_amd.image_io] native JPEG decoder unavailable ({e}); Pillow decodes JPEG within 2/255 of the "
                      "reference's stb_image, not bit for bit", file=sys.stderr)
    from PIL import Image as PILImage
    pil = PILImage.open(io.BytesIO(data))
    if pil_check is not None:
        pil_check(pil)
        if data[:2] == b"\xff\xd8":
            return decode_rgba(data)
    return np.asarray(pil.convert("RGBA"), dtype=np.uint8).copy()
