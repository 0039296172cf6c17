#!/bin/bash
# Round-end evidence for a workload: kernel-trace stats of bench.py + FETCH/WRITE PMC passes.
W=${1:-C2}; TAG=${2:-r01}; OUT=gpurun_out/final_$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $OUT
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python bench.py --workload $W --steps 8 --warmup 2 --cpu-seconds 0 > $OUT/bench_prof.log 2>&1 || echo "kernel trace failed"
python tools/rocpd_summary.py $OUT/bench_results.db $OUT/bench_kernel_stats.md > /dev/null 2>&1; rm -f $OUT/bench_results.db
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT -o $name -- python tools/gpu_frames.py $W 0 3 > $OUT/$name.log 2>&1 || echo "$name: failed/timeout"; python tools/rocpd_summary.py $OUT/${name}_results.db $OUT/$name.md > /dev/null 2>&1; rm -f $OUT/${name}_results.db; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run l2 TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum
tail -1 $OUT/bench_prof.log
