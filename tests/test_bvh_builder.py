"""CPU: the product's host-side BVH builder and node quantiser (chameleonrt_amd/csrc/bvh_builder.cpp),
checked natively (g++, no GPU): partition of the items, containment of every subtree in its child
box, leaf size, depth, and conservativeness of the 16-bit fixed-point boxes as the kernels
dequantise them."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chameleonrt_amd", "csrc")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = os.path.join(tmp_path_factory.mktemp("bvh"), "bvh_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-ffp-contract=off", "-I", CSRC,
                           os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(CSRC, "bvh_builder.cpp"),
                           "-o", out])
    return out


@pytest.mark.parametrize("n,threads,seed,mode", [(1, 1, 1, 0), (2, 1, 1, 0), (5, 2, 2, 0), (1000, 1, 3, 0),
                                                  (200000, 4, 4, 0), (100000, 4, 5, 1), (30000, 2, 6, 2)])
def test_builder_and_quantiser(exe, n, threads, seed, mode):
    p = subprocess.run([exe, str(n), str(threads), str(seed), str(mode)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "errors 0" in p.stdout


def test_thread_count_does_not_change_the_tree(exe):
    outs = [subprocess.run([exe, "50000", str(t), "7", "0"], capture_output=True, text=True, timeout=600).stdout
            for t in (1, 3, 8)]
    assert outs[0] == outs[1] == outs[2]


def test_optimal_collapse_is_not_worse_than_the_greedy_one(exe):
    """The 4-wide nodes come from a dynamic program that minimises the summed surface area of the
    wide nodes (= expected node visits of a random ray); the earlier greedy collapse of the same
    binary tree must never beat it, and both must be valid trees."""
    import os
    import re
    cost = {}
    for mode in ("dp", "greedy"):
        env = dict(os.environ, CRT_BVH_COLLAPSE=mode)
        p = subprocess.run([exe, "60000", "4", "11", "0"], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0 and "errors 0" in p.stdout, p.stdout + p.stderr
        cost[mode] = float(re.search(r"sah_nodes ([0-9.]+)", p.stdout).group(1))
        nodes = int(re.search(r"nodes (\d+) fill", p.stdout).group(1))
        cost[mode + "_nodes"] = nodes
    assert cost["dp"] <= cost["greedy"] * (1 + 1e-6)
    assert cost["dp_nodes"] <= cost["greedy_nodes"]
