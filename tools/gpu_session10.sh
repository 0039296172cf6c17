#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s10; mkdir -p $OUT
( time timeout -k 5 1500 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest.log 2>&1; tail -16 $OUT/pytest.log
python __graft_entry__.py --smoke 2>&1 | tail -2
