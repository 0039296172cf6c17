"""Every rank's share of a frame, timed on ONE GPU (no multi-GPU box is available to the builder).

    python tools/gpu_rank_shares.py WORKLOAD N[,N...] [FRAMES] [WxHxSPP]

For each N and each rank r of N: a context with crt_hip_set_partition(r, N) renders FRAMES frames of the workload (default
schedule = overlapped, pass lanes chosen by the library's own trial, like a rank of `bench.py --gpus N`); the share's time is
the median of the last three frames' render_time_ms. A frame of N GPUs ends with its SLOWEST rank, so the projected
strong-scaling factor is whole frame / max over ranks -- compute only, before the gather (<= 4 MiB per rank, overlapped
with the next frame). Not a scaling curve: one GPU, one rank at a time, nothing competing for the host.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from chameleonrt_amd import core, scenes  # noqa: E402
from chameleonrt_amd.camera import camera_of  # noqa: E402
from chameleonrt_amd.render_hip import PreparedScene, RenderHIP  # noqa: E402


def share_ms(ps, cam, w, h, rank, world, frames):
    r = RenderHIP(flags=core.FLAG_TIMING, rank=rank, world=world)
    r.initialize(w, h)
    r.set_prepared_scene(ps)
    e, d, u, fovy = cam
    ms, rays, lanes = [], 0, 1
    for f in range(frames):
        st = r.render(e, d, u, fovy, f == 0, False)
        ms.append(st.render_time_ms)
        rays, lanes = int(st.rays), int(st.pass_lanes)
    r.close()
    return float(np.median(ms[-3:])), rays, lanes


def main():
    workload = sys.argv[1]
    worlds = [int(x) for x in sys.argv[2].replace("+", ",").split(",")]
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    t = time.time()
    sc, w, h, spp = scenes.make_workload(workload)
    if len(sys.argv) > 4:
        w, h, spp = (int(x) for x in sys.argv[4].lower().split("x"))
    sc.samples_per_pixel = spp
    cam = camera_of(sc)
    ps = PreparedScene(sc, n_threads=len(os.sched_getaffinity(0)))
    print(f"{workload} {w}x{h} {spp} spp: scene + host set_scene {time.time() - t:.1f} s; tile dealing: {os.environ.get('CRT_HIP_TILE_DEAL', 'default')}", flush=True)
    whole, rays_whole, lanes = share_ms(ps, cam, w, h, 0, 1, frames)
    print(f"  whole frame (N = 1): {whole:.2f} ms, {rays_whole} rays, {lanes} pass lane(s)", flush=True)
    for n in worlds:
        rows = [share_ms(ps, cam, w, h, r, n, frames) for r in range(n)]
        ms = np.array([x[0] for x in rows])
        rays = np.array([x[1] for x in rows], np.int64)
        print(f"  N = {n}: per rank ms " + " ".join(f"{x:.2f}" for x in ms) + f" | lanes {[x[2] for x in rows]}")
        print(f"         max {ms.max():.2f}  mean {ms.mean():.2f}  min {ms.min():.2f}  max/mean {ms.max() / ms.mean():.3f}"
              f"  | rays max/mean {rays.max() / rays.mean():.3f}, sum {rays.sum()} ({'==' if rays.sum() == rays_whole else '!='} whole frame)"
              f"  | projected {whole / ms.max():.2f}x from the slowest rank ({whole / ms.mean():.2f}x from the mean)", flush=True)
    ps.close()


if __name__ == "__main__":
    main()
