#!/bin/bash
# round-2 GPU session 2: full parity suite, the new default bench line (C4, in-run PMC), C4F for A/B, per-phase wave profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s2; mkdir -p $OUT
( time timeout -k 5 1500 python -m pytest tests -m gpu -q --durations=12 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -22 $OUT/pytest.log
df -h /dev/shm /tmp | tail -2
( time timeout -k 5 900 python bench.py --steps 10 --warmup 3 --keep-pmc $OUT/pmc ) > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -3 $OUT/bench_c4.err; cat $OUT/bench_c4.json
( time timeout -k 5 600 python bench.py --workload C4F --steps 10 --warmup 3 --cpu-seconds 0 --keep-pmc $OUT/pmc_c4f ) > $OUT/bench_c4f.json 2> $OUT/bench_c4f.err; tail -3 $OUT/bench_c4f.err; cat $OUT/bench_c4f.json
for W in C4F C4 C2; do echo "== $W"; CRT_HIP_DEBUG=1 timeout 300 python tools/gpu_frames.py $W 1 2 2>&1 | grep -E "waves|launch|bounce|frame 1:" | grep -v "frame 0" ; done > $OUT/phases.log 2>&1
cat $OUT/phases.log
