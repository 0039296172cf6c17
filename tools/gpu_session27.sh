#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s27; mkdir -p $OUT
CRT_HIP_DEBUG=1 CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py C4 3 3 > $OUT/c4_phase.log 2>&1
grep -E "frame 2|refill|inner|leaf|retire|tail" $OUT/c4_phase.log | tail -24
timeout 600 python bench.py --workload C5 --steps 2 --warmup 1 --no-pmc --cpu-seconds 0 > $OUT/bench_C5.json 2> $OUT/bench_C5.err; cut -c1-400 $OUT/bench_C5.json
