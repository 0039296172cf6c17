"""CPU: `python bench.py --gpus N` outside a launcher starts its own ranks (the driver's N > 1 command must
not die on a missing torchrun). No GPU needed: the launcher call and the device count are stand-ins here; the real
launch is exercised by tests/test_gpu_dist_single.py on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_the_drivers_form():
    cmd = bench.launch_command(["--gpus", "8", "--steps", "5", "--warmup=2", "--workload", "C4"], requested=8, ranks=8, port=29500)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29500"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    tail = cmd[script + 1:]
    assert tail == ["--gpus", "8", "--requested-gpus", "8", "--steps", "5", "--warmup=2", "--workload", "C4"]


def test_launch_command_degrades_gpus_and_keeps_the_request():
    cmd = bench.launch_command(["--steps", "5", "--gpus=8", "--no-pmc"], requested=8, ranks=4, port=1)
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--requested-gpus", "8", "--steps", "5", "--no-pmc"]
    # the ranks parse what they are handed
    a = bench.parse(tail)
    assert a.gpus == 4 and a.requested_gpus == 8 and a.steps == 5 and a.no_pmc


@pytest.mark.parametrize("have,asked,ranks", [(8, 8, 8), (4, 8, 4), (2, 2, 2), (3, 2, 2)])
def test_self_launch_starts_min_of_request_and_devices(monkeypatch, have, asked, ranks):
    calls = []
    monkeypatch.setattr(bench, "devices_present", lambda: have)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.delenv("CRT_BENCH_SHARE_GPU", raising=False)
    argv = ["--gpus", str(asked), "--steps", "2"]
    args = bench.parse(argv)
    rc = bench.self_launch(args, argv)
    assert rc == 7  # the launcher's exit code is the command's (torch.distributed.run: non-zero if any rank failed)
    (cmd, env), = calls
    assert f"--nproc-per-node={ranks}" in cmd
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail[:4] == ["--gpus", str(ranks), "--requested-gpus", str(asked)]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_self_launch_on_a_one_device_box_runs_one_rank_in_process(monkeypatch):
    monkeypatch.setattr(bench, "devices_present", lambda: 1)
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("nothing to launch with one device"))
    monkeypatch.delenv("CRT_BENCH_SHARE_GPU", raising=False)
    args = bench.parse(["--gpus", "8"])
    assert bench.self_launch(args, ["--gpus", "8"]) is None
    assert args.gpus == 1 and args.requested_gpus == 8  # the line then carries requested_gpus / degraded


def test_self_launch_without_a_device_fails_loudly(monkeypatch):
    monkeypatch.setattr(bench, "devices_present", lambda: 0)
    args = bench.parse(["--gpus", "2"])
    with pytest.raises(SystemExit, match="needs a GPU"):
        bench.self_launch(args, ["--gpus", "2"])


def test_gpus_n_on_this_gpu_less_box_exits_nonzero_with_the_reason():
    """End to end, no stand-ins: the command the driver uses. Here there is no device, so it must say so -- not the old
    'launch N>1 with torchrun' refusal."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    from chameleonrt_amd import core
    if core.load().crt_hip_device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_dist_single.py covers the real launch")
    assert p.returncode != 0 and "needs a GPU" in p.stderr and "torch.distributed.run" not in p.stderr
