#!/bin/bash
# One GPU session = one gpurun call running a list of steps (replaces round 1-2's per-session one-off scripts).
#
#   gpurun --timeout 900 -- 'bash tools/gpu_session.sh TAG step [step ...]'
#
# Everything a step writes goes to gpurun_out/TAG/ (merged back into the working tree by gpurun); the summaries worth
# keeping are copied into profiles/ by hand. Steps:
#   tests[:EXPR[:ENV=V,...]]  pytest -m gpu (optionally -k EXPR, with environment settings) -> pytest_<n>.log
#   bench:W[:ARGS[:ENV=V,...]]  python bench.py --workload W ARGS (',' separates ARGS)   -> bench_W.json / .err
#   ab:W:V1,V2[:N]      frames of workload W with library variants (tools/variants.py; `prod` = the product) -> ab_W.log
#   phase:W             wave-phase profile of the instrumented kernels (CRT_HIP_DEBUG) -> phase_W.log
#   trace[:W]           rocprofv3 --kernel-trace --stats of the bench command    -> kernel_stats_W.md
#   micro:NAME          build/NAME (a tools/*.hip microbenchmark built beforehand) -> NAME.txt
#   frames:W:ENV=V,...  tools/gpu_frames.py W with environment settings          -> frames_W_<n>.log
#   py:NAME[:ARGS[:ENV=V,...]]  python tools/NAME.py ARGS (with environment settings)   -> NAME_<n>.log
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
n=0
for step in "$@"; do
  n=$((n + 1))
  IFS=: read -r kind a b c <<< "$step"
  echo "=== [$n] $step"
  case $kind in
    tests)
      if [ -n "$a" ]; then K=(-k "$a"); else K=(); fi
      ( [ -n "$b" ] && export ${b//,/ }; timeout -k 5 1500 python -m pytest tests -m gpu -x -q -s "${K[@]}" ) > "$OUT/pytest_$n.log" 2>&1
      grep -E "passed|failed|error|diverged|deepest" "$OUT/pytest_$n.log" | tail -12 ;;
    bench)
      ARGS=${b//,/ }
      ( [ -n "$c" ] && export ${c//,/ }; time timeout -k 5 900 python bench.py --workload "$a" $ARGS ) > "$OUT/bench_$a.json" 2> "$OUT/bench_$a.err"
      tail -3 "$OUT/bench_$a.err" | head -1
      python - "$OUT/bench_$a.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print("no JSON line:", e); sys.exit(0)
print(d["config"]["workload"][:40], d["value"], "MRay/s", d["ms_per_step"], "ms", d.get("kernel_ms_per_step"))
print("  schedules", d.get("schedules", {}).get("overlap"), "| per bounce", json.dumps(d.get("kernel_ms_per_bounce")))
for k in (d.get("roofline"), d.get("roofline_other")):
    if k:
        print("  ", k["kernel"], "hbm", k["achieved"], k["frac"], "traffic", k["traffic"], "binding", k.get("binding"), "valu", k.get("valu"))
print("  cpu", d.get("cpu_baseline", {}).get("value"), "parity", d.get("parity"), d.get("pmc_errors"))
PY
      ;;
    ab)
      python tools/variants.py run "$a" ${b//,/ } > "$OUT/ab_$a.log" 2>&1
      grep -E "^==|frame [45]:" "$OUT/ab_$a.log" | cut -c1-200 ;;
    phase)
      CRT_HIP_DEBUG=1 CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py "$a" 3 3 > "$OUT/phase_$a.log" 2>&1
      grep -E "frame 2" "$OUT/phase_$a.log" | grep -E "refill|inner|leaf|retire|tail|bounce" | cut -c1-200 ;;
    trace)
      W=${a:-C4}
      timeout -k 5 600 rocprofv3 --kernel-trace --stats -d "$OUT" -o trace_$W -- python bench.py --workload $W --steps 8 --warmup 2 \
        --cpu-seconds 0 --no-pmc --no-other-schedule > "$OUT/trace_$W.log" 2>&1 || echo "kernel trace failed"
      DB=$(find "$OUT" -name "trace_${W}_results.db" | head -1)
      python tools/rocpd_summary.py "$DB" "$OUT/kernel_stats_$W.md" | head -16; rm -f "$DB" ;;
    micro)
      timeout 300 "./build/$a" > "$OUT/$a.txt" 2>&1; cat "$OUT/$a.txt" ;;
    frames)
      ( export ${b//,/ }; timeout 300 python tools/gpu_frames.py "$a" 2 6 ) > "$OUT/frames_${a}_$n.log" 2>&1
      grep -E "set_scene|frame 5" "$OUT/frames_${a}_$n.log" | cut -c1-220 ;;
    py)
      ( [ -n "$c" ] && export ${c//,/ }; timeout 600 python "tools/$a.py" ${b//,/ } ) > "$OUT/${a}_$n.log" 2>&1; tail -60 "$OUT/${a}_$n.log" | cut -c1-220 ;;
    *) echo "unknown step $step" ;;
  esac
done
