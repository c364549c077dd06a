#!/usr/bin/env python3
"""Per-kernel PMC averages from rocprofv3 rocpd databases: pmc_summary.py <dir with pass*_results.db> [name filter]"""
import glob, sqlite3, sys, collections
d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
res = collections.OrderedDict()
for db in sorted(glob.glob(d + "/*_results.db")):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    q = "select kernel_name, counter_name, grid_size, count(*), avg(value), avg(end-start) from counters_collection group by kernel_name, counter_name, grid_size" if "grid_size" in cols else None
    try:
        rows = con.execute(q).fetchall()
    except Exception as e:
        print(db, "ERR", e, cols); continue
    for kn, cn, gs, n, v, dur in rows:
        if filt not in kn: continue
        key = (kn.split("(")[0][-40:], gs)
        res.setdefault(key, {})[cn] = (v, n, dur)
for key, c in res.items():
    print(key)
    for cn, (v, n, dur) in c.items():
        print(f"    {cn:34s} {v:16.1f}  (n={n}, avg dur {dur/1e3:.1f} us)")
