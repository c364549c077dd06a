#!/bin/bash
# rocprofv3 kernel stats of small-batch passes (tools/bench_latency.py, B given as $1, default 1)
B=${1:-1}; P=gpurun_out/prof_latency_b$B; mkdir -p $P; REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/$P -o lat -- python $REPO/tools/bench_latency.py --batches $B --reps 20 > $REPO/$P/log.txt 2>&1
cd $REPO
python tools/rocpd_summary.py $P/lat_results.db $P/summary "rocprofv3 --kernel-trace --stats -- python tools/bench_latency.py --batches $B --reps 20" > /dev/null 2>&1
python - $P <<'PY'
import sqlite3, sys
P = sys.argv[1]
con = sqlite3.connect(P + '/lat_results.db')
rows = con.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels where name like '%vqs%' group by name, grid_x order by sum(duration) desc limit 30").fetchall()
with open(P + '/by_grid.txt', 'w') as f:
    for r in rows:
        f.write(f"{r[0][:58]:58s} grid {r[1]:8d} n={r[2]:5d} avg {r[3]/1e3:8.1f} us total {r[4]/1e6:7.1f} ms\n")
PY
rm -f $P/*.db
tail -1 $P/log.txt | cut -c1-200; head -24 $P/by_grid.txt
