"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/crt_hip.h
declares; without a GPU the product path fails loudly (no CPU fallback, no oracle behind it)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from chameleonrt_amd import build, core
    build.build()
    return core.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "crt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_hip_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from chameleonrt_amd import core
    assert _declared_symbols() == sorted(core.EXPORTS)


def test_flag_values_of_the_header_and_the_binding_agree():
    """Context flags are part of the boundary: the header's enum == chameleonrt_amd.core's constants."""
    from chameleonrt_amd import core
    src = open(os.path.join(ROOT, "include", "crt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    flags = {k: int(v) for k, v in re.findall(r"\bCRT_HIP_FLAG_([A-Z_]+)\s*=\s*(\d+)", src)}
    assert flags == {"NONE": 0, "COUNTERS": core.FLAG_COUNTERS, "TIMING": core.FLAG_TIMING,
                     "ELIDE_UNUSED_SHADOW_RAYS": core.FLAG_ELIDE_UNUSED_SHADOW_RAYS, "REFINE_IN_BACKGROUND": core.FLAG_REFINE_IN_BACKGROUND}
    assert len(set(flags.values())) == len(flags) and all(v & (v - 1) == 0 for v in flags.values())  # distinct single bits


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.crt_hip_abi_version() == 4


def test_scene_io_library_exports_its_header():
    """include/crt_scene_io.h (the harness's streaming OBJ reader and JPEG decoder) == what libcrt_scene_io.so exports."""
    import ctypes as C
    from chameleonrt_amd import build
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "crt_scene_io.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(crt_(?:obj|image)_[a-z_]+)\s*\(", src)))
    assert "crt_image_decode_jpeg" in names and "crt_obj_parse" in names
    assert len(names) >= 9
    L = C.CDLL(build.build_scene_io())
    for n in names:
        assert hasattr(L, n), n


def test_header_is_plain_c():
    """The boundary must be consumable from C (cgo/JNI/ctypes style bindings)."""
    src = '#include "crt_hip.h"\n#include "crt_kat.h"\n#include "crt_scene_io.h"\nint main(void){return crt_hip_abi_version()==CRT_HIP_ABI_VERSION?0:1;}\n'
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c", "-"], input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_render_stats_layout_matches_the_binding(tmp_path):
    """crt_render_stats as a C compiler lays it out == the ctypes mirror (size and the offset of its last field)."""
    import ctypes as C
    from chameleonrt_amd import core
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "crt_hip.h"\nint main(void){printf("%zu %zu %zu\\n", sizeof(crt_render_stats), '
           'offsetof(crt_render_stats, closest_ms_bounce), offsetof(crt_render_stats, shadow_slots));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", str(exe)], input=src.encode(), check=True)
    size, off_ms, off_last = (int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert size == C.sizeof(core.RenderStats)
    assert off_ms == core.RenderStats.closest_ms_bounce.offset and off_last == core.RenderStats.shadow_slots.offset


def test_product_never_links_the_oracle(lib):
    from chameleonrt_amd import core
    out = subprocess.run(["ldd", core.LIB_PATH], capture_output=True, text=True).stdout
    assert "liborc" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "chameleonrt_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "crt_oracle" not in txt and "oracle_lib" not in txt and "liborc" not in txt, f


def test_no_device_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from chameleonrt_amd import core
    from chameleonrt_amd.render_hip import RenderHIP
    assert lib.crt_hip_device_count() == 0
    with pytest.raises(core.CoreError, match="no HIP device"):
        RenderHIP()


def test_kernels_are_compiled_for_gfx950(lib):
    from chameleonrt_amd import core
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", core.LIB_PATH],
                         capture_output=True, text=True)
    blob = open(core.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_trace_closest" in blob and b"k_shade" in blob, out.stderr
