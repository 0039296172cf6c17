#!/bin/bash
# PMC passes over a short run of the frame loop (separate rocprofv3 invocations, counters only).
# usage: tools/pmc_passes.sh <workload> <outdir-under-gpurun_out>
W=${1:-C2}; OUT=gpurun_out/${2:-pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python tools/gpu_frames.py $W 0 3 > $OUT/$name.log 2>&1; python tools/rocpd_summary.py $OUT/${name}_results.db $OUT/$name.md > /dev/null 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
rocprofv3 -L > $OUT/counters_list.txt 2>&1
ls $OUT
