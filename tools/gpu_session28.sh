#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s28; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_device_build.py -m gpu -x -q -s > $OUT/pytest.log 2>&1; grep -E "passed|failed|device build|Error|error" $OUT/pytest.log | tail -8
echo "== C4 device build"; CRT_HIP_BUILD=device CRT_HIP_OVERLAP=0 timeout 300 python tools/gpu_frames.py C4 2 5 > $OUT/c4_dev.log 2>&1; grep -E "set_scene|frame 4" $OUT/c4_dev.log
