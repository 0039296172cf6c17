"""Frame time by pass lanes and by the timing flag (event records around every launch), small frames.  GPU aid.

    python tools/gpu_lanes_check.py [C2] [frames]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chameleonrt_amd import scenes, core
from chameleonrt_amd.camera import look_at
from chameleonrt_amd.render_hip import RenderHIP

which = sys.argv[1] if len(sys.argv) > 1 else "C2"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 14
sc, w, h, spp = scenes.make_workload(which)
cam = sc.cameras[0]
e, d, u = look_at(cam.position, cam.center, cam.up)
# "busy": as inside bench.py -- torch imported, a torch stream handed to the renderer, another renderer alive next to it
busy = os.environ.get("LANES_CHECK_BUSY") == "1"
keep = None
stream = None
if busy:
    import torch
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    os.environ["CRT_HIP_OVERLAP"] = "0"
    keep = RenderHIP(flags=core.FLAG_TIMING, stream=stream.cuda_stream)
    keep.initialize(w, h)
    keep.set_scene(sc)
    keep.render(e, d, u, cam.fov_y, True, False)
    os.environ["CRT_HIP_OVERLAP"] = "1"
for lanes, aux in (("1", "1"), ("2", "1"), ("2", "0")):
    for flags in (0, core.FLAG_TIMING):
        os.environ["CRT_HIP_LANES"] = lanes
        os.environ["CRT_HIP_LANE_AUX"] = aux
        lanes = f"{lanes} aux {aux} busy {int(busy)} hwq {os.environ.get('GPU_MAX_HW_QUEUES', 'default')}"
        r = RenderHIP(flags=flags, **({"stream": stream.cuda_stream} if busy else {}))
        r.initialize(w, h)
        r.set_scene(sc)
        ms, wall = [], []
        for f in range(nframes):
            t = time.time()
            st = r.render(e, d, u, cam.fov_y, f == 0, False)
            wall.append((time.time() - t) * 1e3)
            ms.append(st.render_time_ms)
        print(f"{which} lanes {lanes or 'default'} flags {flags}: render_time {np.mean(ms[4:]):.3f} ms (min {np.min(ms[4:]):.3f}, max {np.max(ms[4:]):.3f}), wall {np.mean(wall[4:]):.3f} ms", flush=True)
        r.close()
