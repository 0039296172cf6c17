"""Build the HIP core (libcrt_hip_core.so) for gfx950, in-tree.

    python -m chameleonrt_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU
box with the working tree. Parity builds use -ffp-contract=off and no fast-math so that
+ - * / sqrt are bit-reproducible against the CPU oracle (DESIGN.md "Numerics").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcrt_hip_core.so")
SOURCES = ["kernels.hip", "crt_core.cpp", "scene_prepare.cpp", "bvh_builder.cpp", "bvh_device.hip"]
HEADERS = ["crt_types.h", "pt_device.h", "traverse.h", "slab.h", "wavefront.h", "kernels.h", "scene_prepare.h", "host_parallel.h", "bvh_builder.h", "bvh_device.h", "lbvh.h", "leaf_slots.h", "presplit.h",
           os.path.join("..", "..", "include", "crt_hip.h"), os.path.join("..", "..", "include", "crt_kat.h")]
# -fno-slp-vectorize: the SLP vectoriser pairs fp32 operations into gfx950's packed instructions (v_pk_fma_f32, v_pk_mul_f32,
# v_pk_add_f32: ~3 800 of them in these kernels). They compute the same bits, but they need aligned register pairs -- the
# traversal kernels took 80 VGPRs with them and take 65-72 without, which is the difference between 6 and 7 waves per SIMD --
# and they are no faster here: C4 60.6 -> 58.1 ms with the flag alone, 56.9 ms with the seventh wave
# (profiles/r04_issue_bound_ab.txt).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-pthread", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, defines=(), out=None):
    """defines/out: build a tuning variant (tools/variants.py) next to the production library."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = out or LIB
    extra = [d if d.startswith("-") else "-D" + d for d in defines]  # (a variant may also name a raw compiler flag)
    cmd = [hipcc] + FLAGS + extra + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


# SPEED MODE: the same sources with fast-math -- approximate division / sqrt / transcendentals, FMA contraction
# -- as the reference ships its own Embree backend (backends/embree/CMakeLists.txt:12: ispc --opt=fast-math). Opt-in
# (CRT_HIP_SPEED=1, chameleonrt_amd/core.py); every parity statement of this repository is about the default build.
FAST_LIB = os.path.join(HERE, "libcrt_hip_core_fast.so")
FAST_ONLY = ["kernels.hip"]  # the frame's kernels; the host code and the BVH builders (conservative quantisation) stay IEEE
FAST_FLAGS = ["-ffast-math", "-fno-finite-math-only", "-ffp-contract=fast", "-fgpu-approx-transcendentals", "-DCRT_SPEED_MODE=1"]


def build_fast(force=False):
    if not force and os.path.exists(FAST_LIB):
        t = os.path.getmtime(FAST_LIB)
        deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
        if all(os.path.getmtime(d) <= t for d in deps):
            return FAST_LIB
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [f for f in FLAGS if f != "-shared"]
    precise = base + ["-DCRT_SPEED_MODE=1"]
    fast = [f for f in base if f not in ("-ffp-contract=off", "-fno-fast-math")] + FAST_FLAGS
    with tempfile.TemporaryDirectory() as tmp:
        objs = []
        for src in SOURCES:
            obj = os.path.join(tmp, src + ".o")
            subprocess.check_call([hipcc] + (fast if src in FAST_ONLY else precise) + ["-c", os.path.join(CSRC, src), "-o", obj])
            objs.append(obj)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", FAST_LIB])
    return FAST_LIB


IO_LIB = os.path.join(HERE, "libcrt_scene_io.so")
IO_SOURCES = ["obj_reader.cpp", "jpeg_reader.cpp"]


def build_scene_io(force=False):
    """The harness's scene-file readers (include/crt_scene_io.h): plain C++17, no HIP, no GPU."""
    deps = [os.path.join(CSRC, f) for f in IO_SOURCES] + [os.path.join(HERE, "..", "include", "crt_scene_io.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(IO_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(IO_LIB) for d in deps):
        return IO_LIB
    subprocess.check_call([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off"] +
                          [os.path.join(CSRC, f) for f in IO_SOURCES] + ["-o", IO_LIB])
    return IO_LIB


if __name__ == "__main__":
    build_scene_io(force="--force" in sys.argv)
    build_fast(force="--force" in sys.argv)
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
