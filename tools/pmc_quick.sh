#!/bin/bash
# Three quick PMC passes (lane utilisation, waits, TA busy); every run under `timeout`.
W=${1:-C2}; OUT=gpurun_out/${2:-pmc_quick}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python tools/gpu_frames.py $W 0 3 > $OUT/$name.log 2>&1 || echo "$name: failed/timeout"; python tools/rocpd_summary.py $OUT/${name}_results.db $OUT/$name.md > /dev/null 2>&1; rm -f $OUT/${name}_results.db; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run mem TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum
run fetch FETCH_SIZE
for f in sq1 mem fetch; do echo "== $f"; grep -E "k_trace|k_shade" $OUT/$f.md; done
