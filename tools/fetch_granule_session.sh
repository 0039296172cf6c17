cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r4s13
for cfg in "1 0" "32 0" "32 16" "32 8" "16 0"; do
  tag=$(echo $cfg | tr ' ' _)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r4s13/$tag -o g -- ./build/fetch_granule_microbench $cfg > gpurun_out/r4s13/run_$tag.log 2>&1
  python - "$tag" <<PY
import sqlite3,glob,sys
tag=sys.argv[1]
for db in glob.glob(f"gpurun_out/r4s13/{tag}/**/*.db", recursive=True):
    con=sqlite3.connect(db)
    t=[r[0] for r in con.execute("select name from sqlite_master where type='table'") if "pmc_event" in r[0]][0]
    vals=[r[4] for r in con.execute(f"select * from {t}") if r[4]>1000]
    print(tag, "FETCH_SIZE KiB per launch", vals, "-> requests (x64 B) per visit", [round(v*1024/64/67108864,3) for v in vals])
PY
  grep visits gpurun_out/r4s13/run_$tag.log | tail -1
done
