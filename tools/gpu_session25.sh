#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s25; mkdir -p $OUT
for V in prod t21 t5; do echo "== C4 $V"; if [ $V = prod ]; then CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py C4 2 6; else CRT_HIP_LIB=chameleonrt_amd/variants/libcrt_$V.so CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py C4 2 6; fi; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
