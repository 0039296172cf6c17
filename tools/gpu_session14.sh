#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s14; mkdir -p $OUT
for P in 0/1 0/2 0/4 0/8 3/8; do echo "== C4 part $P"; CRT_PART=$P timeout 300 python tools/gpu_frames.py C4 2 6; done > $OUT/part.log 2>&1
grep -E "^==|frame [5]" $OUT/part.log
echo "== overlap"; CRT_PART=0/8 CRT_HIP_OVERLAP=1 timeout 300 python tools/gpu_frames.py C4 0 6 2>&1 | grep "frame 5"
CRT_PART=0/1 CRT_HIP_OVERLAP=1 timeout 300 python tools/gpu_frames.py C4 0 6 2>&1 | grep "frame 5"
CRT_PART=0/8 timeout 300 python tools/gpu_frames.py C4 0 6 2>&1 | grep "frame 5"
CRT_PART=0/1 timeout 300 python tools/gpu_frames.py C4 0 6 2>&1 | grep "frame 5"
