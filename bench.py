#!/usr/bin/env python
"""bench.py -- MRay/s (REPORT_RAY_STATS semantics) of the HIP wavefront path tracer.

    python bench.py --gpus N --steps K --warmup W [--workload C4] [--scaling strong|weak]

With N > 1 and no launcher around it (no WORLD_SIZE in the environment) the command starts its own ranks: it re-executes
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port <free>`
with G = min(N, devices present), forwards its arguments, lets rank 0 print the ONE JSON line and returns the launcher's
exit code (non-zero if any rank failed). On a box with fewer than N devices the line says so (`requested_gpus`,
`degraded`) and `n_gpus` is what actually ran. Under a launcher (the driver's torchrun form) it is one rank, as before.

A "step" is one frame: one pass of the hot path over every pixel-sample of the workload
(`-benchmark-frames` protocol of the reference, main.cpp:293-345: fixed camera, camera_changed only on
frame 0). Default workload: the north-star target, the San-Miguel-like scene at 1920x1080, 16 spp
(BASELINE.json C4; it fits one GPU). Inputs (scene, BVH, textures) are resident in HBM before the
timed region: rank 0 generates the scene and runs the host half of set_scene ONCE (BVH build,
texture linearisation), the prepared arrays reach the other ranks through /dev/shm.

N > 1: one process per GPU (torch.distributed / RCCL), the framebuffer is split into 64x64 tiles
(tile_id % N == rank), every step ends with the gather of the compact RGBA8 tile buffers to rank 0
and the K8 un-permute; the gather of frame f overlaps the tracing of frame f+1 (the tile buffer is
double-buffered). The headline is STRONG scaling -- the same frame, total work fixed, which is what
north_star's ">= 6x at 8 GPUs" asks; the weak-scaling figure (spp x N, BASELINE.json's own C4/C5
pattern) is measured right after it and reported under "weak_scaling".

Rank 0 prints ONE JSON line; DESIGN.md "Measurement" explains every field of `roofline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLOCK_GHZ, N_CUS, SIMDS = 2.4, 256, 4  # MI355X_MICROARCH.md chip-level parameters
NODE_BYTES, SLOT_BYTES = 48, 64  # bytes a visit fetches: 48 of a packed BVH4 node's 64-byte record / a leaf slot of one or two triangles
QUEUE_BYTES_CLOSEST = 24 + 32   # o,d read + the 32-byte hit record {t,u,v,tri | normal,material} written per ray
QUEUE_BYTES_SHADOW = 28 + 16    # o,d,tmax + the 16-byte {c, path} record read per ray
MAX_PATH_DEPTH = 5
# CU-cycles of the vector-memory front end per divergent visit when the working set is L2-resident / in HBM, all lanes on (tools/
# node_bytes_microbench.hip, profiles/r04_node_bytes_microbench.txt): the two-point LINEAR blend of rounds 4-5, kept for continuity
NODE_VISIT_CYCLES_L2, NODE_VISIT_CYCLES_HBM = 2.06, 10.25
SLOT_VISIT_CYCLES_L2, SLOT_VISIT_CYCLES_HBM = 2.77, 10.64
# Round 6: the MEASURED cost of a divergent visit as a function of the L2 hit rate -- (hit rate, CU-cycles per lane-visit) of dependent
# chains of uniformly random records whose working set goes from 1 MB (inside L2) through the Infinity Cache's reach to 1 GB (HBM), 7
# waves per SIMD, at 100 % and at 66 % of the lanes per step (tools/miss_cost_microbench.hip, profiles/r06_miss_cost_microbench.txt, which
# also has the latency of the L2's memory-side reads per row). A visit's cost is NOT a linear blend of a hit price and a miss price: a
# wave's step waits for its slowest lane, so the first few per cent of misses cost most of a miss -- the linear blend was optimistic at
# 90 % hits (C4) and, pricing every miss at the HBM figure, pessimistic at 50 % hits served by the Infinity Cache (C3).
NODE_VISIT_CURVE = {100: [(0.999, 2.20), (0.998, 2.30), (0.942, 3.03), (0.732, 4.28), (0.427, 5.98), (0.258, 9.85), (0.168, 9.10), (0.107, 9.90), (0.066, 10.35)],
                    66: [(1.000, 2.67), (0.998, 2.75), (0.531, 4.98), (0.281, 7.29), (0.156, 8.89), (0.091, 9.96), (0.059, 10.47), (0.042, 10.71), (0.032, 10.69)]}
SLOT_VISIT_CURVE = {100: [(0.999, 2.86), (0.999, 2.94), (0.823, 4.18), (0.584, 5.05), (0.390, 6.63), (0.250, 8.42), (0.155, 9.66), (0.100, 10.37), (0.057, 10.66)],
                    66: [(1.000, 3.32), (0.998, 3.35), (0.544, 4.99), (0.272, 7.14), (0.139, 8.69), (0.071, 9.76), (0.037, 10.26), (0.020, 10.46), (0.007, 10.69)]}


def visit_cycles(curve, hit):
    """Piecewise-linear reading of a measured (hit rate -> cycles per visit) curve, made monotone first (the cost never falls as the hit
    rate falls: one row of the 100 %-lane node sweep, 67 MB, measured above its neighbours)."""
    pts = sorted(curve, key=lambda p: -p[0])  # from all hits down
    xs, ys, worst = [], [], 0.0
    for h, c in pts:
        worst = max(worst, c)
        if xs and h == xs[-1]:
            ys[-1] = max(ys[-1], worst)
            continue
        xs.append(h)
        ys.append(worst)
    if hit >= xs[0]:
        return ys[0]
    for k in range(1, len(xs)):
        if hit >= xs[k]:
            t = (xs[k - 1] - hit) / (xs[k - 1] - xs[k])
            return ys[k - 1] + t * (ys[k] - ys[k - 1])
    return ys[-1]

def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C4", help="BASELINE.json config: C1..C5 (C4F: C4 flattened, no glass)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="0 = skip the CPU baseline and the in-run parity check; otherwise both run on the FIXED tile sample of "
                         "the workload (sample_tiles): the same tiles whatever the box, so the figure is comparable between runs")
    ap.add_argument("--schedule", default=None, choices=["serial", "overlap"],
                    help="serial: the 17 launches of a frame one after the other (default at N = 1: every kernel's event span is its "
                         "own execution time, which the roofline needs); overlap: the occlusion launch of bounce b next to the "
                         "closest-hit launch of bounce b+1 on a second stream (the library's default; default here at N > 1, where "
                         "launch tails are a larger share of the frame)")
    ap.add_argument("--levels", default=None, choices=["two", "world"],
                    help="scenes with several instances: a top-level tree over the instances (two) or one tree in world space "
                         "over per-instance triangle records (world); default: the library's choice (world while the instanced "
                         "triangles fit its memory budget)")
    ap.add_argument("--no-cache", action="store_true", help="do not keep / reuse the prepared scene under /dev/shm between runs")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-schedule", action="store_true", help="N = 1: do not also time the other launch schedule")
    ap.add_argument("--no-speed-mode", action="store_true", help="N = 1: do not also time the opt-in fast-math build")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (traffic = null)")
    ap.add_argument("--keep-pmc", default=None, help="directory to keep the raw per-kernel counter sums in")
    ap.add_argument("--requested-gpus", type=int, default=None, help=argparse.SUPPRESS)  # set by self_launch for its ranks
    ap.add_argument("--dump-image", default=None, metavar="PATH",
                    help="rank 0: np.save the last timed frame's assembled RGBA8 image (uint32, height x width) here (tests: the "
                         "gathered N > 1 image must equal the N = 1 image bit for bit)")
    ap.add_argument("--also", default=None, metavar="WxHxSPP",
                    help="N > 1: after the headline, time the same scene at another resolution / spp too (reported as `c5`); default "
                         "at --gpus 8 on C4: 3840x2160x64 = BASELINE.json's own 8-GPU configuration C5; 'none' switches it off")
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(argv, requested, ranks, port):
    """The command line `python bench.py --gpus <requested> ...` turns into when it starts its own `ranks` processes:
    the driver's own launcher form, argv forwarded with --gpus set to what runs and the request kept in --requested-gpus."""
    fwd, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a in ("--gpus", "--requested-gpus"):
            skip = True
            continue
        if a.startswith("--gpus=") or a.startswith("--requested-gpus="):
            continue
        fwd.append(a)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(ranks), "--requested-gpus", str(requested)] + fwd


def devices_present():
    """HIP devices on this box, asked of the product library (hipGetDeviceCount: no context is created in the parent)."""
    from chameleonrt_amd import core
    return int(core.load().crt_hip_device_count())


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 outside any launcher. Returns None if this process should go on as the only
    rank (one usable device: the degraded N = 1 run), otherwise the exit code of the launcher it ran."""
    import subprocess
    share = os.environ.get("CRT_BENCH_SHARE_GPU") == "1"  # tests: N ranks on one device over gloo
    have = devices_present()
    if have < 1:
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    ranks = args.gpus if share else min(args.gpus, have)
    args.requested_gpus = args.gpus
    if ranks == 1:
        print(f"[bench] --gpus {args.gpus} asked for, {have} device(s) present: running 1 rank in this process", file=sys.stderr)
        args.gpus = 1
        return None
    if ranks < args.gpus:
        print(f"[bench] --gpus {args.gpus} asked for, {have} device(s) present: launching {ranks} ranks", file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // ranks)))
    return subprocess.call(launch_command(argv, args.gpus, ranks, free_port()), env=env)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def sample_tiles(width, height, spp):
    """The fixed sample of 64x64 tiles the CPU baseline is timed on and the in-run parity check compares: every k-th tile
    of the image (row-major tile ids, offset k // 2), k chosen from the workload alone so that the sample holds about
    4.8 M pixel-samples (C4: every 7th of 510 tiles = 73 tiles, ~17 M rays, 15-20 s of the scalar restatement on 16 cores)."""
    ntx, nty = (width + 63) // 64, (height + 63) // 64
    ntiles = ntx * nty
    stride = max(1, int(round(ntiles * 4096 * spp / 4.8e6)))
    return list(range(stride // 2, ntiles, stride)), ntiles


def tile_pixel_mask(width, height, tiles):
    import numpy as np
    ntx = (width + 63) // 64
    m = np.zeros((height, width), bool)
    for t in tiles:
        tx, ty = (t % ntx) * 64, (t // ntx) * 64
        m[ty:ty + 64, tx:tx + 64] = True
    return m


def wrap_device_buffer(ptr, nbytes):
    """Zero-copy torch view of the core's compact tile buffer (plumbing for the collective)."""
    import torch

    class _Dev:
        __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<i4", "data": (ptr, False), "version": 2}

    return torch.as_tensor(_Dev(), device=torch.device("cuda", torch.cuda.current_device()))


class FrameLoop:
    """render -> (N > 1) gather + assemble, with the gather of frame f overlapping frame f+1."""

    def __init__(self, r, cam, dist, rank, world, host_staged=False):
        self.r, self.cam, self.dist, self.rank, self.world = r, cam, dist, rank, world
        self.host_staged = host_staged  # gloo (tests): the tile buffers travel through host memory
        from chameleonrt_amd import multi_gpu
        self.multi_gpu = multi_gpu
        self.views = {}
        self.pending = None  # (work handle, gathered tensor) of the previous frame
        self.in_flight = False  # N > 1: a frame has been enqueued and not collected yet

    @property
    def lanes_tunable(self):
        """Does the library try this rank's frames with one pass lane and with two? (crt_core.cpp lanes_tunable: 0.5 .. 8 Mi paths)"""
        r = self.r
        local_paths = -(-(-(-r.width // 64) * -(-r.height // 64)) // self.world) * 4096 * r.samples_per_pixel
        return (1 << 19) <= local_paths <= (8 << 20)

    def _finish_pending(self):
        if self.pending is not None:
            work, gathered = self.pending
            work.wait()
            if self.rank == 0:
                if self.host_staged:
                    gathered = gathered.cuda()
                self.r.assemble_tiles(gathered.data_ptr(), self.world, readback=False)
            self.pending = None

    def step(self, frame):
        """One step. N = 1: render and return the frame's statistics. N > 1: PIPELINED -- frame f is enqueued before frame
        f-1 is collected (crt_hip_render_begin / _end), so the GPU goes from one frame into the next without waiting for
        the host, and the gather + K8 assemble of frame f-1 run next to frame f; returns the statistics of the frame that
        completed in this step (frame f-1; None in the first step after a drain) -- drain() returns the last frame's."""
        eye, cdir, up, fovy = self.cam
        if not self.dist:
            return self.r.render(eye, cdir, up, fovy, frame == 0, False)
        self.r.render_begin(eye, cdir, up, fovy, frame == 0, False)
        ptr, nbytes = self.r.tile_buffer()  # frame f's compact tile buffer (alternates between two buffers per frame)
        view = self.views.get(ptr)
        if view is None:
            view = self.views[ptr] = wrap_device_buffer(ptr, nbytes)
        done = self.r.render_end() if self.in_flight else None  # frame f-1: the host waits for it with frame f queued behind it
        self._finish_pending()  # gather(f-1) is waited for ON THE STREAM, its K8 assemble is enqueued behind frame f
        gathered, work = self.multi_gpu.gather_tile_buffers(view.cpu() if self.host_staged else view, async_op=True)
        self.pending = (work, gathered)
        self.in_flight = True
        return done

    def drain(self):
        """Collect what is in flight: the last frame's statistics (or None), its gather and assemble."""
        done = None
        if self.in_flight:
            done = self.r.render_end()
            self.in_flight = False
        self._finish_pending()
        return done


def timed_frames(loop, args, dist, first_frame):
    """W warm-up + exactly K timed steps, barrier + synchronize on both sides; sums over the K steps."""
    import torch
    acc = dict(rays=0, closest_rays=0, shadow_rays=0, closest_ms=0.0, shadow_ms=0.0, shade_ms=0.0, raygen_ms=0.0, accumulate_ms=0.0)
    for k in ("closest_rays_bounce", "shadow_rays_bounce", "closest_ms_bounce", "shadow_ms_bounce", "shade_ms_bounce"):
        acc[k] = [0.0] * MAX_PATH_DEPTH
    def account(st):
        if st is None:
            return
        acc["rays"] += st.rays
        acc["closest_rays"] += st.closest_rays
        acc["shadow_rays"] += st.shadow_rays
        acc["shadow_rays_elided"] = acc.get("shadow_rays_elided", 0) + st.shadow_rays_elided
        acc["closest_ms"] += st.closest_ms
        acc["shadow_ms"] += st.shadow_ms
        acc["shade_ms"] += st.shade_ms
        acc["raygen_ms"] += st.raygen_ms
        acc["accumulate_ms"] += st.accumulate_ms
        for name in ("closest_rays_bounce", "shadow_rays_bounce", "closest_ms_bounce", "shadow_ms_bounce", "shade_ms_bounce"):
            arr = getattr(st, name)
            for b in range(MAX_PATH_DEPTH):
                acc[name][b] += arr[b]
        acc["pass_lanes"], acc["passes"] = int(st.pass_lanes), int(st.passes)

    if os.environ.get("CRT_HIP_OVERLAP") != "0":
        # set-up, not warm-up: with the overlapped schedule the library tries frames of up to 8 Mi paths with one pass lane
        # and with two (six frames after a configuration) and keeps the faster; let it decide
        # before the W warm-up steps, whatever W is. Each of these restarts the accumulation, like step 0 below.
        # Larger frames are never cut: nothing to decide, no set-up frames.
        if loop.lanes_tunable:
            for _ in range(6):
                loop.step(0)
    for f in range(args.warmup):
        loop.step(first_frame + f)
    loop.drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for k in range(args.steps):
        account(loop.step(first_frame + args.warmup + k))  # (N > 1: the statistics of the frame before, see FrameLoop.step)
    account(loop.drain())  # the last frame's completion, gather + assemble belong to the timed region
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    total_rays = acc["rays"]
    if dist:
        dev = "cpu" if dist.get_backend() == "gloo" else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        c = torch.tensor([float(acc["rays"])], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        elapsed, total_rays = float(t.item()), int(c.item())
    return elapsed, total_rays, acc


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        rc = self_launch(args, sys.argv[1:])
        if rc is not None:
            sys.exit(rc)
    import numpy as np
    import torch
    from chameleonrt_amd import core, scenes
    from chameleonrt_amd.camera import camera_of
    from chameleonrt_amd.render_hip import PreparedScene, RenderHIP

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # under a launcher the launcher's world size is what runs; the line reports it as n_gpus
        if args.requested_gpus is None:
            args.requested_gpus = args.gpus
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # CRT_BENCH_SHARE_GPU=1 (tests only: a one-GPU box): every rank renders on device 0 and the collective goes over
    # gloo with host staging -- RCCL refuses two ranks on one device. It exercises everything of the N > 1 path but
    # the transport: one scene preparation per node, the tile partition, both scaling legs, gather and K8.
    share_gpu = os.environ.get("CRT_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torchrun the process group is used even for N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.schedule is None:
        args.schedule = "serial" if world == 1 else "overlap"
    os.environ["CRT_HIP_OVERLAP"] = "1" if args.schedule == "overlap" else "0"  # read when a context is created
    if args.levels:
        os.environ["CRT_HIP_LEVELS"] = args.levels  # read when a scene is prepared
    gen, kw, width, height, base_spp = scenes.WORKLOADS[args.workload]
    spp = base_spp * (world if args.scaling == "weak" else 1)
    # ---- scene: generated and prepared ONCE per node (rank 0), shared through /dev/shm; and once per BOX: the prepared
    # arrays stay there as a cache keyed by the workload and the library build, so that the back-to-back runs of a scaling
    # series (N = 1, 2, 4, 8) do not each spend ~20 s on scene generation and the BVH build while N - 1 ranks wait at the
    # barrier (--no-cache: always regenerate; caches older than two hours are swept) ----------------
    # tmpfs if it has room for a prepared San-Miguel-class scene (~2 GB: nodes, leaf slots, 8-bit texels); every
    # rank evaluates the same rule on the same node, so they agree on the path
    shm = "/tmp"
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > (6 << 30):
            shm = "/dev/shm"
    except OSError:
        pass
    build_on_device = os.environ.get("CRT_HIP_BUILD") == "device"  # the BVH by the device builder (bvh_device.hip) instead of host SAH
    import hashlib
    lib_stat = os.stat(core.LIB_PATH)
    key = hashlib.sha1(repr((args.workload, sorted(kw.items()), args.levels, build_on_device, os.environ.get("CRT_SCENE_DIR"),
                             [os.environ.get(k) for k in ("CRT_PAIR_MAX_RATIO", "CRT_BVH_MAX_LEAF", "CRT_BVH_BUILDER", "CRT_HIP_NO_GRAFT", "CRT_BVH_SPLITS", "CRT_BVH_REINSERT")],
                             lib_stat.st_size, int(lib_stat.st_mtime))).encode()).hexdigest()[:12]
    prepared_path, meta_path = f"{shm}/crt_prepared_{args.workload}_{key}.bin", f"{shm}/crt_prepared_{args.workload}_{key}.json"
    scene = None
    t_gen = t_prep = 0.0
    cache_hit = False
    meta = None

    def the_scene():
        """The Scene object itself (the oracle needs it; a cache hit of the prepared arrays does not)."""
        nonlocal scene, t_gen
        if scene is None:
            t0 = time.time()
            scene, _, _, _ = scenes.make_workload(args.workload)  # the real asset under $CRT_SCENE_DIR if there is one, else the stand-in
            scene.samples_per_pixel = spp
            t_gen = time.time() - t0
        return scene

    if rank == 0:
        for f in os.listdir(shm):  # sweep stale caches
            if f.startswith("crt_prepared_"):
                try:
                    if time.time() - os.path.getmtime(os.path.join(shm, f)) > 7200:
                        os.remove(os.path.join(shm, f))
                except OSError:
                    pass
        if not args.no_cache and os.path.exists(prepared_path) and os.path.exists(meta_path):
            try:
                ps = PreparedScene(path=prepared_path)
                with open(meta_path) as f:
                    meta = json.load(f)
                ps.set_samples_per_pixel(spp)
                cache_hit = True
            except Exception:  # a truncated or foreign file: rebuild
                meta = None
        if not cache_hit:
            sc = the_scene()
            t0 = time.time()
            ps = PreparedScene(sc, n_threads=usable_cores(), build_device=local_rank if build_on_device else -1)
            t_prep = time.time() - t0
            e0, d0, u0, f0 = camera_of(sc)
            meta = {"width": width, "height": height, "eye": [float(x) for x in e0], "dir": [float(x) for x in d0],
                    "up": [float(x) for x in u0], "fovy": float(f0), "name": sc.name, "triangles": sc.total_tris(),
                    "instances": len(sc.instances), "textures": len(sc.textures), "materials": len(sc.materials),
                    "scene_gen_s": round(t_gen, 2), "set_scene_host_s": round(t_prep, 2)}
            if world > 1 or not args.no_cache or not (args.no_pmc or args.no_roofline):
                ps.save(prepared_path + ".tmp")
                os.replace(prepared_path + ".tmp", prepared_path)
                with open(meta_path, "w") as f:
                    json.dump(meta, f)
        bvh_levels = ps.levels()
    if dist:
        dist.barrier()
    if rank != 0:
        ps = PreparedScene(path=prepared_path)
        ps.set_samples_per_pixel(spp)
        with open(meta_path) as f:
            meta = json.load(f)
    eye, cdir, up = (np.array(meta[k], np.float32) for k in ("eye", "dir", "up"))
    fovy = meta["fovy"]

    # a dedicated (non-default) torch stream: the legacy default stream serialises against every
    # other stream of the process, which RCCL's internal streams do not like
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    r = RenderHIP(device=local_rank, flags=core.FLAG_TIMING, rank=rank, world=world, stream=stream.cuda_stream)
    r.initialize(width, height)
    t0 = time.time()
    r.set_prepared_scene(ps)
    t_upload = time.time() - t0

    # keep the Python cyclic GC (tens of ms per full collection with torch imported) out of the
    # timed region; the frame loop itself allocates almost nothing
    import gc
    gc.collect()
    gc.disable()
    loop = FrameLoop(r, (eye, cdir, up, fovy), dist, rank, world, host_staged=share_gpu)
    elapsed, total_rays, acc = timed_frames(loop, args, dist, 0)

    out = None
    if rank == 0:
        out = {
            "metric": "MRay/s (REPORT_RAY_STATS)", "value": round(total_rays / elapsed / 1e6, 2), "unit": "MRay/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
            "data": "real" if meta["name"].startswith("real:") else "synthetic",
            "config": {"workload": f"{args.workload} {meta['name']} {width}x{height}" +
                                   ("" if meta["name"].startswith("real:") else " (synthetic stand-in, SURVEY 8d)"),
                       "spp_per_frame": spp, "triangles": meta["triangles"], "instances": meta["instances"],
                       "textures": meta["textures"], "materials": meta["materials"],
                       "pixel_samples_per_step": width * height * spp, "rays_per_step": total_rays // args.steps,
                       "schedule": args.schedule,
                       "acceleration_structure": {0: "one BVH4 (single instance)", 1: "two-level: top-level BVH4 over instances + one BVH4 per mesh",
                                                  2: "world tree: one BVH4 in world space over per-instance triangle records"}[bvh_levels],
                       "parallelism": f"image tiles 64x64 round-robin over {world} GPU(s)" +
                                      (" + RCCL gather to rank 0 every step, overlapped with the next frame" if dist else ""),
                       "scene_gen_s": meta["scene_gen_s"], "set_scene_host_s": meta["set_scene_host_s"],
                       "bvh_builder": "device linear BVH (CRT_HIP_BUILD=device)" if build_on_device else
                                      "host binned SAH" + (lambda p: f" + {p} passes of insertion-based re-optimisation (Bittner 2013)" if p > 0 else "")(
                                          int(os.environ.get("CRT_BVH_REINSERT", "2") or 0)),
                       "prepared_scene_cache": "hit: the prepared arrays of an earlier run on this box (scene_gen_s / set_scene_host_s are that run's)"
                                               if cache_hit else "miss",
                       "set_scene_upload_s": round(t_upload, 2), "host_build_threads": usable_cores()},
        }
        if args.requested_gpus is not None and args.requested_gpus != world:
            out["requested_gpus"] = args.requested_gpus
            out["degraded"] = (f"--gpus {args.requested_gpus} needs {args.requested_gpus} devices, this box has "
                               f"{devices_present()}: ran {world} rank(s); n_gpus and value are what ran")

    if args.dump_image:
        # the frame the timed loop ended on, through the same hand-off at every N: the compact tile buffers of that frame are
        # gathered (N = 1: taken as they are), K8 assembles them on rank 0 and the row-major image is read back
        ptr, nbytes = r.tile_buffer()
        view = loop.views.get(ptr)
        if view is None:
            view = wrap_device_buffer(ptr, nbytes)
        if world > 1:
            g = loop.multi_gpu.gather_tile_buffers(view.cpu() if share_gpu else view)
        else:
            g = view.clone()
        torch.cuda.synchronize()
        if rank == 0:
            r.assemble_tiles((g.cuda() if (share_gpu and world > 1) else g).data_ptr(), world, readback=True)
            np.save(args.dump_image, np.array(r.img, copy=True).view(np.uint32).reshape(height, width))

    # ---- N = 1: the same frames with the other launch schedule (the library's default is the overlapped one; the
    # headline times the serial one because only then is a kernel's event span its own execution time) ----
    if world == 1 and rank == 0 and not args.no_other_schedule:
        other_sched = "overlap" if args.schedule == "serial" else "serial"
        os.environ["CRT_HIP_OVERLAP"] = "1" if other_sched == "overlap" else "0"
        r2 = RenderHIP(device=local_rank, flags=core.FLAG_TIMING, stream=stream.cuda_stream)
        r2.initialize(width, height)
        r2.set_prepared_scene(ps)
        e2, rays2, _ = timed_frames(FrameLoop(r2, (eye, cdir, up, fovy), None, 0, 1), args, None, 0)
        r2.close()
        os.environ["CRT_HIP_OVERLAP"] = "1" if args.schedule == "overlap" else "0"
        out["schedules"] = {args.schedule: {"ms_per_step": out["ms_per_step"], "value": out["value"]},
                            other_sched: {"ms_per_step": round(e2 / args.steps * 1e3, 4), "value": round(rays2 / e2 / 1e6, 2)},
                            "note": "serial: the 17 launches of a frame one after the other (headline: per-kernel spans are execution "
                                    "times); overlap: occlusion(b) next to closest-hit(b+1) on a second stream, the library's default"}
        ov = out["schedules"]["overlap"]
        out["strong_scaling_base"] = {"schedule": "overlap", "ms_per_step": ov["ms_per_step"], "value": ov["value"],
                                      "note": "bench.py --gpus N (N > 1) times the overlapped schedule by default (config.schedule says which): "
                                              "a strong-scaling ratio divides its value by THIS one, not by the serial headline of N = 1 "
                                              "(or pass --schedule to both)"}

    # ---- N = 1: the opt-in elision of occlusion rays whose result cannot reach the image (CRT_HIP_FLAG_ELIDE_UNUSED_SHADOW_RAYS,
    # include/crt_hip.h): same frames, bit-identical images and ray statistics (tests/test_gpu_elide.py). The HEADLINE traces
    # every ray the reference traces; this leg says what the reference's dead rays cost, with the rate over TRACED rays ----
    if world == 1 and rank == 0 and not args.no_other_schedule:
        r3 = RenderHIP(device=local_rank, flags=core.FLAG_TIMING | core.FLAG_ELIDE_UNUSED_SHADOW_RAYS, stream=stream.cuda_stream)
        r3.initialize(width, height)
        r3.set_prepared_scene(ps)
        e3, rays3, acc3 = timed_frames(FrameLoop(r3, (eye, cdir, up, fovy), None, 0, 1), args, None, 0)
        r3.close()
        elided = int(acc3.get("shadow_rays_elided", 0))
        out["elide_unused_shadow_rays"] = {
            "ms_per_step": round(e3 / args.steps * 1e3, 4), "schedule": args.schedule,
            "rays_reference_semantics_per_step": rays3 // args.steps, "rays_elided_per_step": elided // args.steps,
            "share_of_occlusion_rays_elided": round(elided / max(1, elided + int(acc3["shadow_rays"])), 4),
            "MRay_per_s_traced_only": round((rays3 - elided) / e3 / 1e6, 2),
            "note": "opt-in (off in the headline): the reference traces the light-sample occlusion ray of every hit and uses its answer only if "
                    "light_pdf >= EPSILON && bsdf_pdf >= EPSILON (render_embree.ispc:131-153); rays whose contribution is an exact zero "
                    "whatever they hit are counted but not traced. Accumulated radiance, RGBA8 and per-pixel ray counts are bit-identical "
                    "to the default path's. The rate counts traced rays only"}

    # ---- N = 1: the opt-in SPEED MODE build (fast-math, as the reference builds its own ISPC kernels: backends/embree/
    # CMakeLists.txt:12) on the same prepared scene, in a child process; reported next to the headline, never as it ----
    if world == 1 and rank == 0 and not args.no_speed_mode and os.path.exists(prepared_path):
        from chameleonrt_amd import build as crt_build, pmc as pmc_mod
        if os.path.exists(crt_build.FAST_LIB):
            sm = pmc_mod.time_frames(prepared_path, meta_path, frames=10, env_overrides={"CRT_HIP_SPEED": "1"})
            sm["note"] = ("fast-math build of the same sources (approximate division / sqrt / transcendentals, FMA contraction), "
                          f"{args.schedule} schedule; NOT the build the parity tests and the headline are about")
            out["speed_mode"] = sm

    # how each rank's frames were cut (the library's one-lane / two-lane trial is per context: say what it chose)
    lanes_all = [acc.get("pass_lanes", 1)]
    if dist:
        gathered_lanes = [None] * world
        dist.all_gather_object(gathered_lanes, acc.get("pass_lanes", 1))
        lanes_all = gathered_lanes
    if rank == 0:
        out["config"]["passes_per_frame"] = acc.get("passes", 1)
        out["config"]["pass_lanes"] = lanes_all if world > 1 else lanes_all[0]

    # ---- N > 1: what the per-step collective costs when nothing hides it (in the timed loop the gather of frame f runs
    # next to the tracing of frame f+1): K isolated gather + K8 assemble rounds of the last frame's tile buffers ----
    if world > 1:
        ptr, nbytes = r.tile_buffer()
        view = loop.views.get(ptr)
        if view is None:
            view = wrap_device_buffer(ptr, nbytes)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_iso = 10
        for _ in range(n_iso):
            g = loop.multi_gpu.gather_tile_buffers(view.cpu() if share_gpu else view)
            if rank == 0:
                r.assemble_tiles((g.cuda() if share_gpu else g).data_ptr(), world, readback=False)
            torch.cuda.synchronize()
        iso_ms = (time.perf_counter() - t0) / n_iso * 1e3
        if rank == 0:
            out["gather"] = {"isolated_ms": round(iso_ms, 4), "bytes_per_rank": int(nbytes),
                             "note": "dist.gather of the compact RGBA8 tile buffers to rank 0 + kernel K8, host-synchronised, NOT "
                                     "overlapped; inside the timed steps it runs next to the next frame's tracing"}

    # ---- N > 1: the weak-scaling figure next to the strong-scaling headline (or the other way round) ----
    if world > 1:
        other = "weak" if args.scaling == "strong" else "strong"
        spp2 = base_spp * (world if other == "weak" else 1)
        ps.set_samples_per_pixel(spp2)
        r.set_prepared_scene(ps)
        e2, rays2, _ = timed_frames(loop, args, dist, 0)
        if rank == 0:
            out[f"{other}_scaling"] = {"value": round(rays2 / e2 / 1e6, 2), "unit": "MRay/s", "spp_per_frame": spp2,
                                       "ms_per_step": round(e2 / args.steps * 1e3, 4), "steps": args.steps}
        ps.set_samples_per_pixel(spp)

    # ---- N = 8 on C4: BASELINE.json's own 8-GPU configuration next to the strong-scaling headline: C5 = the same scene at
    # 3840x2160, 64 spp (16x the pixel-samples of C4: each rank renders twice a single GPU's C4 frame per step) ----
    also = args.also
    if also is None and world == 8 and args.workload == "C4":
        _, kw5, w5, h5, spp5 = scenes.WORKLOADS["C5"]
        assert kw5 == kw
        also = f"{w5}x{h5}x{spp5}"
    if world > 1 and also and also != "none":
        w5, h5, spp5 = (int(x) for x in also.lower().split("x"))
        ps.set_samples_per_pixel(spp5)
        r.initialize(w5, h5)
        r.set_prepared_scene(ps)
        loop.views.clear()  # the tile buffers were re-allocated
        import copy
        a5 = copy.copy(args)
        a5.steps, a5.warmup = min(args.steps, 4), 1
        e5, rays5, acc5 = timed_frames(loop, a5, dist, 0)
        if rank == 0:
            out["c5"] = {"workload": f"{'C5' if (w5, h5, spp5) == scenes.WORKLOADS['C5'][2:] and args.workload == 'C4' else args.workload} {meta['name']} {w5}x{h5}", "spp_per_frame": spp5, "value": round(rays5 / e5 / 1e6, 2),
                         "unit": "MRay/s", "ms_per_step": round(e5 / a5.steps * 1e3, 4), "steps": a5.steps, "warmup": a5.warmup,
                         "rays_per_step": rays5 // a5.steps, "passes_per_frame_rank0": acc5.get("passes"),
                         "note": "BASELINE.json configs[4]: tile split over 8 GPUs + RCCL gather of the 4K framebuffer every step"}
        ps.set_samples_per_pixel(spp)
        r.initialize(width, height)
        r.set_prepared_scene(ps)
        loop.views.clear()

    # ---- roofline of the traversal kernels (rank 0's share; identical code on every rank) ----
    if rank == 0 and not args.no_roofline:
        # (1) algorithmic bytes: nodes / triangles counted by the instrumented build of the same kernels
        ri = RenderHIP(device=local_rank, flags=core.FLAG_COUNTERS, rank=rank, world=world, stream=stream.cuda_stream)
        ri.initialize(width, height)
        ri.set_prepared_scene(ps)
        cn = ct = sn = stt = cr = sr = csl = ssl = 0
        n_probe = min(3, args.warmup + args.steps)
        for f in range(n_probe):  # same frames -> same rays as the timed run (deterministic)
            s2 = ri.render(eye, cdir, up, fovy, f == 0, False)
            cn, ct, sn, stt = cn + s2.closest_nodes, ct + s2.closest_tris, sn + s2.shadow_nodes, stt + s2.shadow_tris
            cr, sr = cr + s2.closest_rays, sr + s2.shadow_rays
            csl, ssl = csl + s2.closest_slots, ssl + s2.shadow_slots
        ri.close()
        bytes_closest = QUEUE_BYTES_CLOSEST + NODE_BYTES * cn / cr + SLOT_BYTES * csl / cr
        bytes_shadow = QUEUE_BYTES_SHADOW + NODE_BYTES * sn / max(1, sr) + SLOT_BYTES * ssl / max(1, sr)
        visits = {"closest": {"nodes_per_ray": round(cn / cr, 2), "leaf_slots_per_ray": round(csl / cr, 2), "triangles_per_ray": round(ct / cr, 2)},
                  "shadow": {"nodes_per_ray": round(sn / max(1, sr), 2), "leaf_slots_per_ray": round(ssl / max(1, sr), 2),
                             "triangles_per_ray": round(stt / max(1, sr), 2)}}
        launches = MAX_PATH_DEPTH * args.steps
        # (2) hardware counters of the same frames, measured now (separate rocprofv3 passes of a child process)
        pmc = {}
        if not args.no_pmc and world == 1:
            from chameleonrt_amd import pmc as pmc_mod
            pmc = pmc_mod.measure(prepared_path, meta_path, frames=3, keep_dir=args.keep_pmc)

        def counter(pass_name, kernel, name):
            d = pmc.get(pass_name, {})
            k = d.get(kernel)
            return (k.get(name), k.get("calls")) if isinstance(k, dict) and name in k else (None, None)

        def k_total_us(kernel):
            k = pmc.get("sq", {}).get(kernel)
            return k.get("total_us") if isinstance(k, dict) else None

        def roof(name, b_per_ray, n_rays, ms):
            avg_ms = ms / launches
            algorithmic = b_per_ray * n_rays / (ms * 1e-3) / 1e9
            out_k = {"kernel": name, "avg_launch_ms": round(avg_ms, 4), "rays_per_launch": n_rays // launches}
            # measured HBM-side traffic per launch: (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE reports half of the
            # bytes on gfx950 (MI355X_MICROARCH.md, HBM section); averaged over the launches of the counter pass
            fs, nf = counter("fetch", name, "FETCH_SIZE")
            ws, nw = counter("write", name, "WRITE_SIZE")
            traffic = None
            if fs is not None and ws is not None and nf and nw:
                traffic = (2.0 * fs / nf + ws / nw) * 1024.0
            # VALU: share of the kernel's SIMD-cycles in which a VALU instruction was executing
            va, _ = counter("sq", name, "SQ_ACTIVE_INST_VALU")
            busy, _ = counter("sq", name, "SQ_BUSY_CYCLES")
            insts, _ = counter("sq", name, "SQ_INSTS_VALU")
            thr, _ = counter("sq", name, "SQ_THREAD_CYCLES_VALU")
            wave_cyc, _ = counter("sq", name, "SQ_WAVE_CYCLES")
            wait_any, _ = counter("sq", name, "SQ_WAIT_ANY")
            wait_inst, _ = counter("sq", name, "SQ_WAIT_INST_ANY")
            out_k["hbm_measured"] = None if traffic is None else {
                "bytes_per_launch": round(traffic), "achieved": round(traffic / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            out_k["algorithmic"] = {"bytes_per_ray": round(b_per_ray, 1), "GB_per_s": round(algorithmic, 1),
                                    "over_hbm_peak": round(algorithmic / HBM_PEAK_GBS, 4),
                                    "note": "every node / triangle touch priced as an HBM access (SURVEY 8d); the caches absorb "
                                            "most of it, so this is demand, not HBM traffic, and not a roofline fraction"}
            if insts and thr and wave_cyc:
                out_k["valu"] = {"lanes_active_per_instruction": round(thr / insts, 1),
                                 "wave_cycles_waiting": round((wait_any or 0) / wave_cyc, 3),
                                 "wave_cycles_issue_stalled": round((wait_inst or 0) / wave_cyc, 3)}
                gui, _ = counter("sq", name, "GRBM_GUI_ACTIVE")
                if gui:
                    # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md: v_fma_f32 wave64 =
                    # 2 cyc); GRBM_GUI_ACTIVE is summed over the 8 XCDs. A lower bound of pipe occupancy: quarter-rate
                    # instructions (rcp, sqrt, 64-bit) hold the pipe longer.
                    cycles = gui / 8.0
                    out_k["valu"]["pipe_busy_frac"] = round(insts * 2.0 / (N_CUS * SIMDS * cycles), 4)
                    out_k["valu"]["clock_GHz"] = round(cycles / (k_total_us(name) * 1e-6) / 1e9, 3) if k_total_us(name) else None
            # vector-memory front end: share of the kernel's cycles in which a CU's texture addresser (TA) was busy --
            # the binding resource of this kernel (tools/ta_quad_microbench.hip prices a random
            # 64-byte node visit at 2.8 TA-bound CU-cycles when it hits L2 and ~10 when it misses)
            ta, _ = counter("mem", name, "TA_TA_BUSY_sum")
            gui_m, _ = counter("mem", name, "GRBM_GUI_ACTIVE")
            hit, _ = counter("mem", name, "TCC_HIT_sum")
            miss, _ = counter("mem", name, "TCC_MISS_sum")
            if ta and gui_m:
                out_k["vmem_front_end"] = {"ta_busy_frac": round(ta / (N_CUS * gui_m / 8.0), 4),
                                           "l2_hit_rate": round(hit / (hit + miss), 4) if hit is not None and miss else None}
            lv, _ = counter("ea", name, "TCC_EA0_RDREQ_LEVEL_sum")
            rq, _ = counter("ea", name, "TCC_EA0_RDREQ_sum")
            if lv and rq:
                # average latency of the L2's memory-side reads (L2 clocks): which regime the kernel's L2 misses are served in
                out_k["ea_read_latency_l2_clocks"] = round(lv / rq)
            out_k["traffic"] = None if traffic is None else round(traffic)
            return out_k

        rc = roof("k_trace_closest", bytes_closest, acc["closest_rays"], acc["closest_ms"])
        rs = roof("k_trace_shadow", bytes_shadow, acc["shadow_rays"], acc["shadow_ms"])
        rc["line_visits_per_ray"] = (cn + csl) / max(1, cr)
        rs["line_visits_per_ray"] = (sn + ssl) / max(1, sr)
        rc["node_share"], rs["node_share"] = cn / max(1, cn + csl), sn / max(1, sn + ssl)
        if args.schedule == "serial":
            dom, other_k = (rc, rs) if acc["closest_ms"] >= acc["shadow_ms"] else (rs, rc)
        else:  # the occlusion spans include queueing: the closest-hit kernel, whose spans are clean, is the one reported first
            dom, other_k = rc, rs

        def contract(k):
            """The contract's roofline object for one kernel: the HBM axis on MEASURED traffic (never above 1)."""
            h = k["hbm_measured"]
            base = {"kernel": k["kernel"], "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                    "achieved": h["achieved"] if h else None, "frac": h["frac"] if h else None, "traffic": k["traffic"],
                    "avg_launch_ms": k["avg_launch_ms"], "rays_per_launch": k["rays_per_launch"],
                    "algorithmic": k["algorithmic"]}
            if "valu" in k:
                base["valu"] = k["valu"]
            if "vmem_front_end" in k:
                # what actually binds the kernel, next to the contract's HBM axis: never above 1 by construction
                base["binding"] = {"resource": "per-CU vector-memory front end (TA busy cycles / kernel cycles)",
                                   "frac": k["vmem_front_end"]["ta_busy_frac"], "l2_hit_rate": k["vmem_front_end"]["l2_hit_rate"]}
            # the bound argued for in DESIGN.md section 6, checkable from this line alone: CU-cycles the launch spends per
            # 64-byte line visit (nodes + leaf slots, counted by the instrumented kernels) against what the microbenchmark
            # says such a visit costs the front end at this run's L2 hit rate
            v = k.get("line_visits_per_ray")
            if v:
                clock = (k.get("valu") or {}).get("clock_GHz") or CLOCK_GHZ
                cyc = N_CUS * clock * 1e9 * k["avg_launch_ms"] * 1e-3 / max(1, k["rays_per_launch"]) / v
                hit = (k.get("vmem_front_end") or {}).get("l2_hit_rate")
                b = base.setdefault("binding", {})
                b["line_visits_per_ray"] = round(v, 2)
                b["cycles_per_line_visit"] = round(cyc, 2)
                if hit is not None:
                    ns = k.get("node_share", 1.0)
                    ceil = {lanes: ns * visit_cycles(NODE_VISIT_CURVE[lanes], hit) + (1.0 - ns) * visit_cycles(SLOT_VISIT_CURVE[lanes], hit) for lanes in (100, 66)}
                    linear = (ns * (hit * NODE_VISIT_CYCLES_L2 + (1.0 - hit) * NODE_VISIT_CYCLES_HBM) +
                              (1.0 - ns) * (hit * SLOT_VISIT_CYCLES_L2 + (1.0 - hit) * SLOT_VISIT_CYCLES_HBM))
                    b["node_share_of_visits"] = round(ns, 3)
                    b["ceiling_cycles"] = round(ceil[100], 2)
                    b["frac_of_front_end_ceiling"] = round(ceil[100] / cyc, 3)
                    b["ceiling_cycles_at_66pct_lanes"] = round(ceil[66], 2)
                    b["frac_of_ceiling_at_66pct_lanes"] = round(ceil[66] / cyc, 3)
                    b["ceiling_cycles_linear_blend_r05"] = round(linear, 2)
                    b["frac_of_linear_blend_r05"] = round(linear / cyc, 3)
                    lat = k.get("ea_read_latency_l2_clocks")
                    if lat is not None:
                        b["l2_miss_read_latency_l2_clocks"] = lat
                        b["l2_misses_served_by"] = ("mostly the Infinity Cache (the microbenchmark's 8 ... 34 MB rows measure 630 ... 1 020)" if lat < 1030
                                                    else "mostly HBM (the microbenchmark's rows beyond the Infinity Cache measure 1 050 ... 1 350)")
                    b["ceiling_note"] = ("ceiling_cycles: what dependent chains of UNIFORMLY RANDOM node / leaf-slot visits cost the CU at this run's L2 hit rate with "
                                         "every lane on (tools/miss_cost_microbench.hip, profiles/r06_miss_cost_microbench.txt: measured per working set from 1 MB to "
                                         "1 GB, interpolated in the hit rate, blended by node share); _at_66pct_lanes: the same with 66 % of the lanes per step (the "
                                         "kernels run 33 - 42 of 64). A fraction >= 1 says the kernel beats uniformly random chains at its hit rate -- its lanes' "
                                         "misses are correlated (neighbouring rays miss together) -- i.e. it is AT the memory system's pace and only a higher hit rate "
                                         "or fewer visits would speed it up. _linear_blend_r05: rounds 4-5's two-point blend (L2 price x hits + HBM price x misses), "
                                         "kept for continuity; it has no Infinity-Cache regime. The kernels also answer to the instructions they issue per step "
                                         "(profiles/r04_issue_bound_ab.txt)")
            return base

        out["roofline"] = contract(dom)
        out["roofline_other"] = contract(other_k)
        out["roofline_note"] = ("bound: the contract's axis for this path is HBM; achieved / frac / traffic are the MEASURED HBM-side "
                                "bytes of this run's counter passes (rocprofv3 child processes on the same prepared scene), not the "
                                "algorithmic demand, which the caches absorb. What binds the traversal kernels is `binding`: the per-CU "
                                "vector-memory front end (divergent 16-byte lane requests), then the waits it causes (`valu`); DESIGN.md section 6.")
        if pmc and any("error" in v for v in pmc.values() if isinstance(v, dict)):
            out["pmc_errors"] = {k: v["error"] for k, v in pmc.items() if isinstance(v, dict) and "error" in v}
        # (overlapped schedule: the occlusion launch of bounce b runs next to the closest-hit launch of bounce b+1 on a
        # second stream; its event span then includes the time its blocks wait for CUs, so trace_shadow is an upper
        # bound and the sum over the kinds exceeds ms_per_step)
        out["visits_per_ray"] = visits  # counted by the instrumented kernels (== the oracle's walk of the same arrays, tests/)
        out["kernel_ms_per_step"] = {"schedule": args.schedule,
                                     "trace_closest": round(acc["closest_ms"] / args.steps, 4),
                                     "trace_shadow": round(acc["shadow_ms"] / args.steps, 4),
                                     "raygen+shade+accumulate": round(acc["shade_ms"] / args.steps, 4)}
        per = lambda name: [round(x / args.steps, 4) for x in acc[name]]
        out["kernel_ms_per_bounce"] = {"trace_closest": per("closest_ms_bounce"), "trace_shadow": per("shadow_ms_bounce"),
                                       "shade": per("shade_ms_bounce"), "raygen": round(acc["raygen_ms"] / args.steps, 4),
                                       "accumulate": round(acc["accumulate_ms"] / args.steps, 4),
                                       "closest_rays": [int(x / args.steps) for x in acc["closest_rays_bounce"]],
                                       "shadow_rays": [int(x / args.steps) for x in acc["shadow_rays_bounce"]]}
    # ---- CPU baseline + parity, on the FIXED tile sample of this workload: the oracle restatement renders frame 0 of
    # those tiles on this box's host cores (timed: the reported baseline, not a target), and the very same pixels of the
    # frame the HIP path renders of the SAME workload as configured -- full resolution, spp, textures -- are compared
    # with it under the image tolerance of tests/parity.py (render_embree.ispc:198-355 on BASELINE.json's config) ----
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        from tests.oracle_lib import OracleRenderer
        from tests.parity import compare_images
        cores = usable_cores()
        tiles, ntiles = sample_tiles(width, height, spp)
        o = OracleRenderer(the_scene(), width, height, cores)
        stc = o.render_tiles(eye, cdir, up, fovy, True, tiles)
        out["cpu_baseline"] = {"value": round(stc.rays_per_second / 1e6, 3), "unit": "MRay/s", "cores": cores,
                               "kind": "port",
                               "sample": f"CPU restatement (scalar C++, own BVH2 -- NOT Embree/ISPC/TBB, which cannot be built here): "
                                         f"frame 0 of the same workload, the fixed sample of {len(tiles)} of {ntiles} 64x64 tiles "
                                         f"(every {tiles[1] - tiles[0] if len(tiles) > 1 else 1}th), {stc.rays} rays in {stc.render_time_ms / 1e3:.1f} s"}
        st0 = r.render(eye, cdir, up, fovy, True, True)  # frame 0 again (camera_changed resets the accumulation)
        mask = tile_pixel_mask(width, height, tiles)
        g, c = r.accum()[mask][:, None, :], o.accum()[mask][:, None, :]
        diverged, mean_rel = compare_images(g, c)
        gc, cc = r.ray_counts()[mask].astype(np.int64), o.ray_counts()[mask].astype(np.int64)
        g8 = r.img.view(np.uint8).reshape(height, width, 4)[mask].astype(np.int32)
        c8 = o.framebuffer().view(np.uint8).reshape(height, width, 4)[mask].astype(np.int32)
        out["parity"] = {"tiles": len(tiles), "pixels": int(mask.sum()), "diverged": round(diverged, 6), "mean_rel": float(f"{mean_rel:.3e}"),
                         "ray_count_mismatch": round(float((gc != cc).mean()), 6),
                         "rays_gpu": int(gc.sum()), "rays_cpu": int(cc.sum()),
                         "rgba8_over_1lsb": round(float((np.abs(g8 - c8) > 1).any(axis=1).mean()), 6),
                         "tolerance": "|gpu - cpu| <= 1e-4 + 1e-3 |cpu| per channel; diverged = fraction of sampled pixels outside it "
                                      "(bar: <= 1e-3), mean_rel over the others (bar: <= 1e-4); ray counts differ only on diverged pixels",
                         "ok": bool(diverged <= 1e-3 and mean_rel <= 1e-4)}
        del st0
    r.close()
    ps.close()
    if dist:
        dist.barrier()
    if rank == 0:
        if args.no_cache:
            for p in (prepared_path, meta_path):
                if os.path.exists(p):
                    os.remove(p)
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
