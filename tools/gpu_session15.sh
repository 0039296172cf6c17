#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s15; mkdir -p $OUT
( timeout -k 5 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for W in C4 C4F C3 C2; do for O in 1 0; do echo "== $W overlap $O"; CRT_HIP_OVERLAP=$O timeout 300 python tools/gpu_frames.py $W 2 6; done; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
