#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s4; mkdir -p $OUT
( timeout -k 5 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_scale.py tests/test_gpu_image.py tests/test_gpu_partition.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for W in C2 C4F C4 C3; do
  for V in prod b7; do
    echo "== $W $V";
    if [ $V = prod ]; then timeout 300 python tools/gpu_frames.py $W 2 6; else CRT_HIP_LIB=chameleonrt_amd/variants/libcrt_$V.so timeout 300 python tools/gpu_frames.py $W 2 6; fi
  done
done > $OUT/ab.log 2>&1
grep -E "^==|frame [45]" $OUT/ab.log
