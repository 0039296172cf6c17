#!/bin/bash
# GPU side of tools/node_replay_microbench.hip: replay the recorded visit traces (build/trace_<W>.bin, tools/make_visit_trace.py) with
# 3-request and 2-request node fetches, then once more under rocprofv3 for the L2 hit rate of the replay itself (to compare with the
# production kernels' on the same workload), and list the memory-side counters this rocprofv3 offers.
#   gpurun -- 'bash tools/replay_session.sh TAG W [W ...]'
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
rocprofv3 --list-avail 2>/dev/null | grep -i -E "mall|umc|dram|hbm|EA_RDREQ|EA0_RDREQ|TCC_REQ|TCC_HIT|TCC_MISS" | head -60 > "$OUT/counters_avail.txt"
for W in "$@"; do
  timeout 300 ./build/node_replay_microbench build/trace_$W.bin 8 > "$OUT/replay_$W.txt" 2>&1
  cat "$OUT/replay_$W.txt"
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$OUT/pmc_$W" -o replay -f csv -- ./build/node_replay_microbench build/trace_$W.bin 8 > /dev/null 2>&1
  python - "$OUT/pmc_$W" "$OUT/replay_$W.txt" <<'PY'
import csv, glob, sys, collections
rows = collections.OrderedDict()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Kernel_Name"][:40])
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
with open(sys.argv[2], "a") as out:
    for (d, name), c in rows.items():
        h, m = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
        line = f"  dispatch {d} {name}: L2 hit rate {h / max(1.0, h + m):.3f} ({h:.3g} hits, {m:.3g} misses)"
        print(line); out.write(line + "\n")
PY
  rm -rf "$OUT/pmc_$W"
done
