#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s20; mkdir -p $OUT
for W in C4 C2; do for V in prod notex norad nouv; do echo "== $W $V"; if [ $V = prod ]; then CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py $W 2 6; else CRT_HIP_OVERLAP=0 CRT_HIP_LIB=chameleonrt_amd/variants/libcrt_$V.so timeout 300 python tools/gpu_frames.py $W 2 6; fi; done; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
