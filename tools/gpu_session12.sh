#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s12; mkdir -p $OUT
( timeout -k 5 900 python -m pytest tests/test_gpu_kat.py tests/test_gpu_image.py tests/test_gpu_scale.py tests/test_gpu_device_build.py tests/test_gpu_zz_reference_golden.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for W in C4 C4F C2 C3; do echo "== $W"; timeout 300 python tools/gpu_frames.py $W 2 6; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
