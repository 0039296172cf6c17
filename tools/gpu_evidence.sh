#!/bin/bash
# Round evidence: the bench line of every workload (in-run counter passes kept), and the rocprofv3 kernel-trace
# summary of the default bench command. Usage: bash tools/gpu_evidence.sh r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
for W in C4 C4F C3 C2; do
  ( time timeout -k 5 900 python bench.py --workload $W --steps 16 --warmup 3 --keep-pmc $OUT/pmc_$W ) > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  tail -2 $OUT/bench_$W.err | head -1; python -c "
import json,sys
d=json.load(open('$OUT/bench_$W.json'))
r=d['roofline']; o=d['roofline_other']
print('$W', d['value'], 'MRay/s', d['ms_per_step'], 'ms', d['kernel_ms_per_step'])
for k in (r,o): print('  ', k['kernel'], 'hbm', k['achieved'], k['frac'], 'binding', k.get('binding'), 'valu', k.get('valu'), 'alg', k['algorithmic']['GB_per_s'], k['algorithmic']['bytes_per_ray'])
print('  cpu', d.get('cpu_baseline',{}).get('value'), d.get('pmc_errors'))
"
done
timeout -k 5 600 rocprofv3 --kernel-trace --stats -d $OUT -o bench_trace -- python bench.py --steps 8 --warmup 2 --cpu-seconds 0 --no-pmc > $OUT/bench_trace.log 2>&1 || echo "kernel trace failed"
DB=$(find $OUT -name "bench_trace_results.db" | head -1); python tools/rocpd_summary.py $DB $OUT/bench_kernel_stats.md | head -14; rm -f $DB
tail -1 $OUT/bench_trace.log | cut -c1-300
