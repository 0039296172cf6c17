#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s24; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_traversal.py tests/test_gpu_scale.py tests/test_gpu_image.py tests/test_gpu_zz_reference_golden.py tests/test_gpu_partition.py tests/test_gpu_kat.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for V in graft nograft; do echo "== C4 $V"; if [ $V = graft ]; then CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py C4 2 6; else CRT_HIP_NO_GRAFT=1 CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py C4 2 6; fi; done > $OUT/ab.log 2>&1
grep -E "^==|frame [5]" $OUT/ab.log
