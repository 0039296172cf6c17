#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s9; mkdir -p $OUT
for W in C4; do
  for V in prod t21s14 t5s15; do
    echo "== $W $V";
    if [ $V = prod ]; then timeout 300 python tools/gpu_frames.py $W 2 6; else CRT_HIP_LIB=chameleonrt_amd/variants/libcrt_$V.so timeout 300 python tools/gpu_frames.py $W 2 6; fi
  done
done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
