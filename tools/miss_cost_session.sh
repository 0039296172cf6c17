#!/bin/bash
# GPU side of tools/miss_cost_microbench.hip: every configuration once plain (timing) and once under rocprofv3 for the L2 hit rate and the
# average latency of the L2's memory-side reads.   gpurun -- 'bash tools/miss_cost_session.sh TAG'
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
: > "$OUT/miss_cost.txt"
for LOADS in 3 4; do for LANES in 100 66; do for LOG in 14 16 17 18 19 20 21 22 24; do
  LINE=$(./build/miss_cost_microbench $LOG $LOADS $LANES)
  D=$(mktemp -d /tmp/mc_XXXX)
  timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d "$D" -o mc -f csv -- ./build/miss_cost_microbench $LOG $LOADS $LANES > /dev/null 2>&1
  CNT=$(python - "$D" <<'PY'
import csv, glob, sys
last = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_miss_walk" in r["Kernel_Name"]:
            last.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
if last:
    c = last[max(last)]
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    lv, rq = c.get("TCC_EA0_RDREQ_LEVEL_sum", 0.0), c.get("TCC_EA0_RDREQ_sum", 0.0)
    print(f"L2 hit {h / max(1.0, h + m):.3f}  EA read latency {lv / max(1.0, rq):.0f} L2 clocks  ({rq:.3g} memory-side reads)")
else:
    print("no counters")
PY
)
  rm -rf "$D"
  echo "$LINE | $CNT" | tee -a "$OUT/miss_cost.txt"
done; done; done
