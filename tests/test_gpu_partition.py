"""GPU: image-tile partition invariants. The RNG is keyed by the GLOBAL pixel id,
so rendering with the framebuffer split over `world` ranks and assembling the tiles must give
exactly the image a single rank renders. Both ranks run on the one GPU a test box has."""
import ctypes as C

import numpy as np
import pytest

from chameleonrt_amd import scenes
from chameleonrt_amd.render_hip import RenderHIP
from tests.parity import camera_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_render_equals_single(world, hip_lib):
    hip = C.CDLL("libamdhip64.so")  # resolves to the runtime the core already uses
    sc = scenes.instanced_grove()
    w, h = 300, 200  # 5 x 4 tiles, clipped edges, tile count not divisible by 3
    e, d, u, fovy = camera_of(sc)
    ref = RenderHIP()
    ref.initialize(w, h)
    ref.set_scene(sc)
    ranks = [RenderHIP(rank=k, world=world) for k in range(world)]
    for r in ranks:
        r.initialize(w, h)
        r.set_scene(sc)
    total = 0
    for f in range(2):
        st = ref.render(e, d, u, fovy, f == 0, True)
        parts = [r.render(e, d, u, fovy, f == 0, False) for r in ranks]
        total = sum(int(p.rays) for p in parts)
        assert total == int(st.rays)
    acc = np.zeros((h, w, 3), np.float32)
    for r in ranks:
        acc += np.nan_to_num(r.accum())  # each rank fills only its own tiles, others are 0
    assert np.array_equal(np.nan_to_num(ref.accum()), acc)
    # gather the compact RGBA8 tile buffers as RCCL would (here: device-to-device copies) and un-permute
    _, nbytes = ranks[0].tile_buffer()
    gathered = C.c_void_p()
    assert hip.hipMalloc(C.byref(gathered), C.c_size_t(world * nbytes)) == 0
    for k, r in enumerate(ranks):
        ptr, nb = r.tile_buffer()
        assert nb == nbytes
        rc = hip.hipMemcpy(C.c_void_p(gathered.value + k * nbytes), C.c_void_p(ptr), C.c_size_t(nbytes), 3)
        assert rc == 0
    assert hip.hipDeviceSynchronize() == 0
    ranks[0].assemble_tiles(gathered.value, world, readback=True)
    assert np.array_equal(ranks[0].img, ref.img)
    hip.hipFree(gathered)
    for r in ranks + [ref]:
        r.close()
