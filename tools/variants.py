"""Build tuning variants of the HIP core and time them back to back on the GPU box.

    python tools/variants.py build   name:DEF=VAL,DEF=VAL ...     (here, cross-compile)
    python tools/variants.py run WORKLOAD name ...                  (on the GPU box)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "chameleonrt_amd", "variants")

def build(specs):
    from chameleonrt_amd import build as b
    os.makedirs(VDIR, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, defs = spec.partition(":")
        defines = [d for d in defs.split(",") if d]
        out = os.path.join(VDIR, f"libcrt_{name}.so")
        b.build(defines=defines, out=out)
        print("built", out, defines)

def run(workload, names, frames="6"):
    for name in names:
        env = dict(os.environ)
        if name != "prod":
            env["CRT_HIP_LIB"] = os.path.join(VDIR, f"libcrt_{name}.so")
        print("==", name, flush=True)
        subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gpu_frames.py"), workload, "2", frames], env=env)

if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2], sys.argv[3:])
