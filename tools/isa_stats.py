"""Register / scratch / LDS budget of every kernel of the HIP core, from the compiler's own metadata (no GPU needed).

    python tools/isa_stats.py [-DNAME=VALUE ...] [--keep out.s]

Compiles chameleonrt_amd/csrc/kernels.hip for gfx950 with the product's flags (device side only) and prints, per
kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane, LDS bytes, and the waves per SIMD the register budget allows
(512 VGPRs per lane-slot of a SIMD, allocated in blocks of 8)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chameleonrt_amd import build as b


def stats(defines=(), source="kernels.hip", keep=None):
    out = keep or tempfile.mktemp(suffix=".s")
    flags = [f for f in b.FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only"] + [d if d.startswith("-") else "-D" + d for d in defines] + \
          [os.path.join(b.CSRC, source), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    text = open(out).read()
    if not keep:
        os.remove(out)
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                         r"\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", text, re.S):
        agpr, lds, name, scratch, sgpr, vgpr = m.groups()
        rows.append(dict(name=name, vgpr=int(vgpr), agpr=int(agpr), sgpr=int(sgpr), scratch=int(scratch), lds=int(lds)))
    return rows


def demangle(name):
    try:
        return subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], text=True).strip()
    except Exception:
        return name


if __name__ == "__main__":
    defs = [a[2:] if a.startswith("-D") else a for a in sys.argv[1:] if a.startswith("-D") or a.startswith("-f") or a.startswith("-m")]
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    for r in stats(defs, keep=keep):
        total = r["vgpr"] + r["agpr"]
        alloc = (total + 7) // 8 * 8
        waves = min(8, 512 // max(alloc, 1))
        n = demangle(r["name"]).replace("void crt::", "").split("(")[0]
        print(f"{n:52s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} scratch {r['scratch']:4d} B  lds {r['lds']:6d} B  waves/SIMD {waves}")
