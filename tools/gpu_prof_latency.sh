#!/bin/bash
# rocprofv3 kernel trace of the small-batch pass (tools/bench_latency.py --no-graph, one batch size), kernels by (name, grid): where a B = 1 pass spends
# its time.  LIB=<name> profiles lab_so/libvqs_<name>.so instead of the product library.  Output: gpurun_out/${TAG}/latency_b${B}_by_grid.txt
B=${B:-1}; OUT=gpurun_out/${TAG:-prof_latency}; mkdir -p $OUT
REPO=$(pwd); export PYTHONUNBUFFERED=1
[ -n "$LIB" ] && export VQS_LIB_PATH=$REPO/lab_so/libvqs_$LIB.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/$OUT/trace -o lat -- python $REPO/tools/bench_latency.py --batches $B --reps 20 --no-graph > $REPO/$OUT/latency_under_rocprof.log 2>&1
echo "rocprof exit $?"; cd $REPO
python - $OUT $B <<'PY'
import sqlite3, sys, glob
P, B = sys.argv[1], sys.argv[2]
db = glob.glob(P + '/trace/**/*.db', recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels where name like '%vqs%' group by name, grid_x order by sum(duration) desc limit 45").fetchall()
with open(f"{P}/latency_b{B}_by_grid.txt", 'w') as f:
    for r in rows:
        f.write(f"{r[0][:58]:58s} grid {r[1]:8d} n={r[2]:5d} avg {r[3]/1e3:9.1f} us total {r[4]/1e6:8.1f} ms\n")
print(open(f"{P}/latency_b{B}_by_grid.txt").read())
PY
rm -rf $OUT/trace
