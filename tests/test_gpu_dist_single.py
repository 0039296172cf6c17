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
