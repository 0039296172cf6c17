"""CPU: the drop-in boundary is compiled against the REFERENCE's own headers, not a stand-in.

backends/hip/render_hip.{h,cpp}, render_hip_plugin.cpp and render_hip_gl.{h,cpp} include util/render_backend.h,
scene.h, render_plugin.h, display/gldisplay.h, glad and imgui.h from where they lie under /root/reference
(oracle/Makefile: `_ref/libcrt_hip.so`, `_ref/crt_bench`, `boundary_check`); the only stand-ins are for what the
reference itself fetches from outside its tree: GLM (oracle/ref_shim_scene) and <SDL.h> (oracle/ref_shim_app).
The executed half is tests/test_gpu_cpp_boundary.py. Contract: util/render_backend.h:12-32, util/render_plugin.h:23-63,
util/display/gldisplay.h:35-37 (GLNativeRenderer).
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
have_ref = os.path.exists(os.path.join(REFERENCE, "util", "render_plugin.h"))
needs_ref = pytest.mark.skipif(not have_ref, reason="the reference tree exists in the development container only")


def test_no_stand_in_header_is_left():
    assert not os.path.exists(os.path.join(ROOT, "backends", "hip", "standin"))
    for f in os.listdir(os.path.join(ROOT, "backends", "hip")):
        if f.endswith((".h", ".cpp")):
            assert "STANDIN" not in open(os.path.join(ROOT, "backends", "hip", f)).read(), f


@needs_ref
def test_plugin_and_gl_interop_type_check_against_the_real_headers():
    """render_hip_plugin.cpp (with and without CRT_HIP_GL_INTEROP) and render_hip_gl.cpp: -fsyntax-only, 0 errors."""
    p = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "boundary_check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "boundary ok" in p.stdout


@needs_ref
def test_plugin_exports_the_one_c_symbol_and_only_lazy_function_holes():
    """libcrt_hip.so exports `populate_plugin_functions` (util/render_plugin.h:55-63, the only C-linkage symbol of a
    plugin); what it leaves undefined beyond its link libraries are FUNCTIONS of the reference's display / imgui libraries
    (resolved lazily, render_plugin.cpp:37 dlopens RTLD_LAZY) -- no data symbol (vtable, typeinfo) that would stop dlopen."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/libcrt_hip.so"])
    so = os.path.join(ROOT, "oracle", "_ref", "libcrt_hip.so")
    nm = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout.splitlines()
    assert any(ln.split()[-2:] == ["T", "populate_plugin_functions"] for ln in nm)
    ours = [ln.split()[-1] for ln in nm if " U " in ln and ("GLDisplay" in ln or "ImGui" in ln)]
    assert sorted(ours) == ["_ZN5ImGui17SetCurrentContextEP12ImGuiContext", "_ZN9GLDisplayC1EP10SDL_Window"], ours


@needs_ref
def test_reference_loader_opens_the_plugin_and_reaches_the_backend():
    """The reference's RenderPlugin (util/render_plugin.cpp, compiled in place into oracle/_ref/crt_bench) dlopens the plugin,
    finds populate_plugin_functions and calls make_renderer; without a GPU that ends in RenderHIP's own refusal -- which
    proves the call crossed loader -> function table -> backend constructor -> C-ABI (there is no CPU fallback)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/crt_bench"])
    from chameleonrt_amd import core
    if core.load().crt_hip_device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_cpp_boundary.py runs the full path")
    p = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "crt_bench"), "-benchmark-frames", "1"], capture_output=True, text=True)
    assert p.returncode == 1
    assert "RenderHIP: no HIP device (this backend has no CPU fallback)" in p.stderr, p.stderr


def test_cmake_recipe_compiles_the_core_with_the_flags_of_build_py():
    """backends/hip/CMakeLists.txt (what a maintainer of the reference adds) builds libcrt_hip_core.so with the
    same compiler flags and sources as chameleonrt_amd/build.py -- in particular -ffp-contract=off / -fno-fast-math (parity) and
    -fno-slp-vectorize (without it the traversal kernels spill at their 7-wave launch bounds)."""
    import re

    from chameleonrt_amd import build
    text = open(os.path.join(ROOT, "backends", "hip", "CMakeLists.txt")).read()
    cmd = re.search(r"COMMAND \$\{HIPCC\}(.*?)-o \$\{CRT_HIP_CORE\}", text, re.S).group(1)
    flags = [f for f in build.FLAGS if f not in ("-Wall", "-Wno-unused-function")]
    for f in flags:
        assert f in cmd.split(), f
    for src in build.SOURCES:
        assert f"/chameleonrt_amd/csrc/{src}" in cmd, src
