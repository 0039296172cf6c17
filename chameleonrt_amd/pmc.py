"""Hardware-counter passes over the traversal kernels, run from inside bench.py.

`measure()` re-runs a few frames of the SAME prepared scene (shared through /dev/shm) in a child
process under `rocprofv3 --pmc ...` -- one pass per counter group, never combined with a trace
domain other than the kernel trace -- and returns per-kernel counter sums and dispatch counts.
bench.py turns them into `roofline.traffic` (HBM-side bytes per launch, measured in the run, not
read from a committed file) and the VALU figures of the binding-resource line.

    python -m chameleonrt_amd.pmc <prepared.bin> <meta.json> <frames>      (the child)
"""
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# counter groups that fit one pass each (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ, 4 TCC where
# FETCH_SIZE costs 3 and WRITE_SIZE 2, 2 GRBM)
PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
           "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    # the per-CU vector-memory front end (texture addresser): busy cycles summed over the 256 CUs, and the L2's hit rate
    "mem": ["TA_TA_BUSY_sum", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"],
    # how far away an L2 miss is served from: LEVEL / RDREQ = the average latency of the L2's memory-side reads in L2 clocks (short:
    # the Infinity Cache; long: HBM). bench.py prices the misses of a traversal kernel with it (tools/miss_cost_microbench.hip)
    "ea": ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "GRBM_GUI_ACTIVE"],
}


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) if m else name[:60]


def read_db(path):
    """{kernel: {"calls": n, "total_us": t, counter: sum, ...}} from a rocprofv3 rocpd database."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    out = {}
    for name, s, e in cur.execute("select name, start, end from kernels").fetchall():
        k = out.setdefault(short(name), {"calls": 0, "total_us": 0.0})
        k["calls"] += 1
        k["total_us"] += (e - s) / 1e3
    try:
        q = cur.execute("select kernel_name, counter_name, sum(value) from counters_collection "
                        "group by kernel_name, counter_name").fetchall()
        for kn, cn, v in q:
            k = out.setdefault(short(kn), {"calls": 0, "total_us": 0.0})
            k[cn] = k.get(cn, 0.0) + float(v)
    except sqlite3.Error:
        pass
    db.close()
    return out


def measure(prepared_path, meta_path, frames=3, passes=("fetch", "write", "sq", "mem", "ea"), timeout=240, keep_dir=None):
    """Returns {pass: {kernel: {...}}}; a pass that fails or times out is reported as {"error": ...}."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    res = {}
    if not os.path.exists(rocprof):
        return {p: {"error": "rocprofv3 not found"} for p in passes}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for p in passes:
        d = tempfile.mkdtemp(prefix=f"crt_pmc_{p}_", dir="/tmp")
        cmd = [rocprof, "--pmc", *PASSES[p], "--kernel-trace", "-d", d, "-o", p, "--",
               sys.executable, "-m", "chameleonrt_amd.pmc", prepared_path, meta_path, str(frames)]
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            if r.returncode != 0 or not dbs:
                res[p] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
            else:
                res[p] = read_db(dbs[0])
        except (subprocess.TimeoutExpired, OSError) as ex:
            res[p] = {"error": str(ex)[:300]}
        if keep_dir:
            os.makedirs(keep_dir, exist_ok=True)
            with open(os.path.join(keep_dir, f"pmc_{p}.json"), "w") as f:
                json.dump(res[p], f, indent=1)
        shutil.rmtree(d, ignore_errors=True)
    return res


def time_frames(prepared_path, meta_path, frames=10, env_overrides=None, timeout=240):
    """ms per frame and rays per frame of a few frames of the prepared scene rendered by a CHILD process (another build
    of the library -- CRT_HIP_SPEED=1 -- or other environment settings than this process was started with)."""
    env = dict(os.environ, TMPDIR="/tmp", **(env_overrides or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run([sys.executable, "-m", "chameleonrt_amd.pmc", prepared_path, meta_path, str(frames), "--time"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
    except (subprocess.TimeoutExpired, OSError) as ex:
        return {"error": str(ex)[:300]}


def _child(prepared_path, meta_path, frames, timed=False):
    import numpy as np
    from chameleonrt_amd.render_hip import PreparedScene, RenderHIP
    with open(meta_path) as f:
        m = json.load(f)
    ps = PreparedScene(path=prepared_path)
    r = RenderHIP()
    r.initialize(m["width"], m["height"])
    r.set_prepared_scene(ps)
    ps.close()
    e, d, u = (np.array(m[k], np.float32) for k in ("eye", "dir", "up"))
    ms, rays = [], 0
    for f in range(frames):
        st = r.render(e, d, u, m["fovy"], f == 0, False)
        ms.append(st.render_time_ms)
        rays = int(st.rays)
    if timed:  # the first two frames are warm-up
        t = sum(ms[2:]) / max(1, len(ms) - 2)
        print(json.dumps({"ms_per_frame": round(t, 4), "rays_per_frame": rays, "MRay_per_s": round(rays / t / 1e3, 2), "name": r.name()}))
    r.close()


if __name__ == "__main__":
    _child(sys.argv[1], sys.argv[2], int(sys.argv[3]), "--time" in sys.argv)
