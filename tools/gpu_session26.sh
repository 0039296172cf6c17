#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s26; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kat.py tests/test_gpu_image.py tests/test_gpu_scale.py tests/test_gpu_zz_reference_golden.py tests/test_gpu_edge_cases.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for W in C3 C4 C2; do for V in prod base; do echo "== $W $V"; if [ $V = prod ]; then CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py $W 2 6; else CRT_HIP_LIB=chameleonrt_amd/variants/libcrt_$V.so CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py $W 2 6; fi; done; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
