#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command; summaries land in gpurun_out/prof_<xl|xxl>/.
# MODEL=clip-flant5-xxl (default, the metric's model) | clip-flant5-xl
MODEL=${MODEL:-clip-flant5-xxl}; TAG=${MODEL#clip-flant5-}; P=gpurun_out/prof_$TAG
mkdir -p $P
export PYTHONUNBUFFERED=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $REPO/$P -o bench -- python $REPO/bench.py --model $MODEL --steps 3 --warmup 1 --cpu-pairs 0 --also none > $REPO/$P/bench_under_rocprof.log 2>&1
echo "rocprof exit $?"
cd $REPO

# keep only the small summaries (kernel trace csv can be large)
find $P -name "*kernel_trace*" -size +20M -delete
python tools/rocpd_summary.py $P/bench_results.db $P/summary "rocprofv3 --kernel-trace --stats -- python bench.py --model $MODEL --steps 3 --warmup 1 --cpu-pairs 0 --also none" > /dev/null 2>&1
python - $P <<'PY'
import sqlite3, sys
P = sys.argv[1]
con = sqlite3.connect(P + '/bench_results.db')
rows = con.execute("select name, grid_x, count(*), avg(duration), sum(duration) from kernels where name like '%vqs%' group by name, grid_x order by sum(duration) desc limit 40").fetchall()
with open(P + '/by_grid.txt', 'w') as f:
    for r in rows:
        f.write(f"{r[0][:60]:60s} grid {r[1]:8d} n={r[2]:5d} avg {r[3]/1e3:9.1f} us total {r[4]/1e6:8.1f} ms\n")
PY
rm -f $P/*.db
