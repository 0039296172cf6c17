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
