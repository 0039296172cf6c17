#!/bin/bash
# Memory-pipeline PMC passes (TLB / L1->L2 latency / SQ levels). Every rocprofv3 run is wrapped in
# `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and then hang.
W=${1:-C2}; OUT=gpurun_out/${2:-pmc_mem}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python tools/gpu_frames.py $W 0 3 > $OUT/$name.log 2>&1 || echo "$name: failed/timeout"; python tools/rocpd_summary.py $OUT/${name}_results.db $OUT/$name.md > /dev/null 2>&1; rm -f $OUT/${name}_results.db; }
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run pend TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum
run sq3 SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU
for f in tlb lat pend sq3; do echo "== $f"; grep -E "k_trace_closest|k_trace_shadow_a|k_shade" $OUT/$f.md | grep -v "| 15 | [0-9.]* | [0-9.]* |"; done
