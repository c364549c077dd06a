#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries land in gpurun_out/prof/.
mkdir -p gpurun_out/prof
export PYTHONUNBUFFERED=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --cpu-pairs 0 > $REPO/gpurun_out/prof/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"
cd $REPO
find gpurun_out/prof -type f | head -20
# keep only the small summaries (kernel trace csv can be large)
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
python tools/rocpd_summary.py gpurun_out/prof/bench_results.db gpurun_out/prof/summary "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --cpu-pairs 0" > /dev/null 2>&1
python - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/prof/bench_results.db')
rows = con.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels where name like '%vqs%' group by name, grid_x order by sum(duration) desc limit 40").fetchall()
with open('gpurun_out/prof/by_grid.txt', 'w') as f:
    for r in rows:
        f.write(f"{r[0][:60]:60s} grid {r[1]:8d} n={r[2]:5d} avg {r[3]/1e3:9.1f} us total {r[4]/1e6:8.1f} ms\n")
PY
rm -f gpurun_out/prof/*.db
