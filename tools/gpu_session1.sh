#!/bin/bash
# round-2 GPU session 1: parity suite (incl. at-scale cases), A/B of traversal variants, direct SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/s1; mkdir -p $OUT
( time timeout -k 5 1200 python -m pytest tests -m gpu -x -q --durations=12 ) > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
for W in C2 C4F C4; do
  for V in prod co1; do
    echo "== $W $V"; 
    if [ $V = prod ]; then timeout 300 python tools/gpu_frames.py $W 2 6; else CRT_HIP_LIB=chameleonrt_amd/variants/libcrt_$V.so timeout 300 python tools/gpu_frames.py $W 2 6; fi
  done
done > $OUT/ab.log 2>&1
grep -E "^==|frame [345]|set_scene|scene gen" $OUT/ab.log
run() { name=$1; shift; timeout -k 5 240 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python tools/gpu_frames.py C4F 0 3 > $OUT/$name.log 2>&1 || echo "$name: failed/timeout"; python tools/rocpd_summary.py $OUT/${name}_results.db $OUT/$name.md > /dev/null 2>&1; rm -f $OUT/${name}_results.db; }
run sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sqb SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CU_CYCLES
grep -E "k_trace|k_shade" $OUT/sqa.md | head -40
