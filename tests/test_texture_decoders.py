"""CPU: texture FILE decoding against the reference's decoder.

The reference reads every texture file with its vendored stb_image (`Image::Image(file, ...)`, util/material.cpp:5-17:
stbi_set_flip_vertically_on_load(1), four channels forced); chameleonrt_amd's importers read them with Pillow
(obj_io._load_texture, gltf_io). tests/golden/scenes/decoders/ holds an OBJ whose materials name the formats real Sponza /
San Miguel assets ship besides PNG: JPEG (4:2:0, 4:4:4, greyscale, progressive) and TGA (32-bit RLE, 24-bit bottom-up and
top-down, 8-bit grey); tests/golden/refdecoders_obj.npz is what the reference's importer, compiled from where it lies
(oracle/_ref/libref_scene.so, tests/golden/make_scene_golden.py --decoders), makes of them.

The bar: PNG (tests/test_importers_pinned.py) and TGA bit for bit. JPEG is a DOCUMENTED DEVIATION:
stb_image and libjpeg differ in the inverse DCT's rounding and in chroma up-sampling, so decoded texels may differ by
at most 2 / 255 on a few per cent of the samples -- asserted here so that the deviation cannot grow unnoticed.
"""
import os

import numpy as np
import pytest

from chameleonrt_amd.obj_io import load_obj
from tests import ref_scene_lib as R
from tests.golden.make_scene_golden import DECODER_IMAGES

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "golden", "scenes", "decoders", "t.obj")


def _pairs(ref):
    mine = R.flatten(load_obj(OBJ))
    assert int(ref["counts"][4]) == len(DECODER_IMAGES) == int(mine["counts"][4])
    for i, name in enumerate(DECODER_IMAGES):
        assert np.array_equal(ref[f"tex{i}_info"], mine[f"tex{i}_info"]), name  # width, height, 4 channels, sRGB
        yield name, np.asarray(ref[f"tex{i}_data"]).astype(int), np.asarray(mine[f"tex{i}_data"]).astype(int)


def _check(ref):
    for name, a, b in _pairs(ref):
        d = np.abs(a - b)
        if name.endswith(".tga"):
            assert d.max() == 0, f"{name}: TGA decoding differs from stb_image's"
        else:
            assert d.max() <= 2 and d.mean() <= 0.12 and (d > 0).mean() <= 0.10, (name, int(d.max()), float(d.mean()), float((d > 0).mean()))
            assert (a.reshape(-1, 4)[:, 3] == 255).all() and (b.reshape(-1, 4)[:, 3] == 255).all()


def test_tga_bit_exact_and_jpeg_within_two_lsb_of_the_reference_dump():
    _check(dict(np.load(os.path.join(HERE, "golden", "refdecoders_obj.npz"))))


@pytest.mark.skipif(not R.available(), reason="oracle/_ref/libref_scene.so is built where /root/reference exists")
def test_the_dump_is_what_the_reference_decodes_now():
    live = R.load(OBJ)
    gold = dict(np.load(os.path.join(HERE, "golden", "refdecoders_obj.npz")))
    for k in gold:
        assert np.array_equal(np.asarray(live[k]), np.asarray(gold[k])), k
    _check(live)
