#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s17; mkdir -p $OUT
( timeout -k 5 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_scale.py tests/test_gpu_image.py tests/test_gpu_edge_cases.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for W in C4 C4F; do echo "== $W"; CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py $W 2 6; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
