#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s8; mkdir -p $OUT
( timeout -k 5 1200 python -m pytest tests -m gpu -q -s ) > $OUT/pytest.log 2>&1; grep -E "deepest|tris; device|passed|failed|Error|FAILED" $OUT/pytest.log | tail -14
for W in C4 C4F C3 C2; do echo "== $W"; timeout 300 python tools/gpu_frames.py $W 2 6; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
