#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s6; mkdir -p $OUT
( timeout -k 5 900 python -m pytest tests/test_gpu_device_build.py tests/test_gpu_edge_cases.py -m gpu -q -s ) > $OUT/pytest.log 2>&1; grep -E "tris; device|passed|failed|Error" $OUT/pytest.log | tail -12
for W in C4F C4 C3 C2; do
  for B in host device; do
    echo "== $W $B"; CRT_HIP_DEBUG=1 CRT_HIP_BUILD=$B timeout 300 python tools/gpu_frames.py $W 3 4 2>&1 | grep -E "set_scene (leaf|[0-9])|frame 3:"
  done
done > $OUT/build.log 2>&1
cat $OUT/build.log
