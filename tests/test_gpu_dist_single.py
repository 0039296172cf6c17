"""GPU: the multi-GPU bench plumbing on the one GPU a test box has -- NCCL/RCCL process group of
size 1 under torchrun, zero-copy torch view of the core's tile buffer, gather + K8 assemble.
(The real N > 1 collective cannot run here; its host logic is covered by the gloo test.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--workload", "C1", "--cpu-seconds", "0", "--no-roofline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and "RCCL gather" in out["config"]["parallelism"]


def test_bench_two_ranks_on_one_gpu():
    """The N > 1 path of bench.py with two ranks sharing this box's one GPU (CRT_BENCH_SHARE_GPU=1: gloo with host
    staging instead of RCCL, which refuses two ranks per device): rank 0 prepares the scene once and the other rank
    loads it from /dev/shm, each renders its half of the tiles, the gather of frame f overlaps frame f+1, rank 0
    assembles; the strong-scaling headline and the weak-scaling figure come out of the same run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CRT_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--workload", "C1", "--cpu-seconds", "0", "--also", "320x200x3"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["spp_per_frame"] == 1 and out["weak_scaling"]["spp_per_frame"] == 2
    assert out["weak_scaling"]["value"] > 0
    # Cornell 512 x 512, 1 spp: every pixel-sample traces at least its primary ray, whatever the split
    assert out["config"]["rays_per_step"] >= 512 * 512
    assert "roofline" in out and out["roofline"]["traffic"] is None  # counter passes are a single-GPU leg
    # what the N > 1 line says about itself: how every rank cut its frames, what the un-hidden collective costs, and the
    # second configuration of the same scene (at --gpus 8 on C4 that is BASELINE.json's C5)
    assert out["config"]["pass_lanes"] == [1, 1] and out["config"]["passes_per_frame"] == 1
    assert out["gather"]["isolated_ms"] > 0 and out["gather"]["bytes_per_rank"] == 32 * 4096 * 4
    assert out["c5"]["spp_per_frame"] == 3 and out["c5"]["value"] > 0 and out["c5"]["rays_per_step"] >= 320 * 200 * 3


def _bench(extra, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def _device_count():
    sys.path.insert(0, ROOT)
    from chameleonrt_amd import core
    return core.load().crt_hip_device_count()


COMMON = ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-pmc", "--no-roofline", "--no-other-schedule", "--no-speed-mode",
          "--workload", "C1", "--also", "none"]


def test_bench_gpus2_without_a_launcher_over_rccl(tmp_path):
    """`python bench.py --gpus 2` exactly as the driver types it, on a box with >= 2 devices: the command starts its own two
    ranks, they gather over RCCL, and the assembled image equals the N = 1 image bit for bit. (Skipped on the one-GPU test
    box; the two tests below run the same code there.)"""
    import numpy as np
    if _device_count() < 2:
        pytest.skip("needs >= 2 devices")
    one, two = str(tmp_path / "n1.npy"), str(tmp_path / "n2.npy")
    o1 = _bench(["--gpus", "1", "--schedule", "overlap", "--dump-image", one] + COMMON)
    o2 = _bench(["--gpus", "2", "--dump-image", two] + COMMON)
    assert o1["n_gpus"] == 1 and o2["n_gpus"] == 2 and "requested_gpus" not in o2
    assert o2["config"]["rays_per_step"] == o1["config"]["rays_per_step"]
    assert o2["gather"]["isolated_ms"] > 0
    assert np.array_equal(np.load(one), np.load(two))


def test_bench_gpus2_without_a_launcher_degrades_to_the_devices_present():
    """The same command on a box with ONE device: rc 0, one rank, and the line says what was asked for and what ran."""
    if _device_count() != 1:
        pytest.skip("a one-device box is what this is about")
    out = _bench(["--gpus", "2"] + COMMON)
    assert out["n_gpus"] == 1 and out["requested_gpus"] == 2 and "needs 2 devices, this box has 1" in out["degraded"]
    assert out["value"] > 0 and out["config"]["schedule"] == "serial"


def test_bench_self_launched_ranks_assemble_the_single_rank_image(tmp_path):
    """Self-launch end to end on one device (CRT_BENCH_SHARE_GPU=1: both ranks on device 0, gloo with host staging instead of
    RCCL): `python bench.py --gpus 2` spawns its two ranks, and the image rank 0 assembles from the gathered tile buffers
    is the N = 1 image bit for bit."""
    import numpy as np
    one, two = str(tmp_path / "n1.npy"), str(tmp_path / "n2.npy")
    o1 = _bench(["--gpus", "1", "--schedule", "overlap", "--dump-image", one] + COMMON)
    o2 = _bench(["--gpus", "2", "--dump-image", two] + COMMON, env_extra={"CRT_BENCH_SHARE_GPU": "1"})
    assert o1["n_gpus"] == 1 and o2["n_gpus"] == 2
    assert o2["config"]["rays_per_step"] == o1["config"]["rays_per_step"]
    a, b = np.load(one), np.load(two)
    assert a.shape == (512, 512) and (a >> 24 == 255).all()
    assert np.array_equal(a, b)


def test_gathered_image_equals_direct_image(hip_lib):
    """world = 1 through the gather/assemble path gives the image render() itself produces."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from chameleonrt_amd import scenes
    from chameleonrt_amd.render_hip import RenderHIP
    from tests.parity import camera_of
    sc = scenes.cornell(spp=1)
    r = RenderHIP()
    r.initialize(200, 136)
    r.set_scene(sc)
    e, d, u, fovy = camera_of(sc)
    r.render(e, d, u, fovy, True, True)
    direct = r.img.copy()
    ptr, nbytes = r.tile_buffer()
    view = bench.wrap_device_buffer(ptr, nbytes)
    assert view.is_cuda and view.numel() * 4 == nbytes
    gathered = view.clone()
    torch.cuda.synchronize()
    r.assemble_tiles(gathered.data_ptr(), 1, readback=True)
    assert np.array_equal(r.img, direct)
    r.close()
