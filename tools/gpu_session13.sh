#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s13; mkdir -p $OUT
( timeout -k 5 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_scale.py tests/test_gpu_image.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for W in C4; do echo "== $W"; timeout 300 python tools/gpu_frames.py $W 2 6;  timeout 300 python tools/gpu_frames.py $W 3 3; done > $OUT/ab.log 2>&1
grep -E "^==|frame [25]" $OUT/ab.log
