#!/usr/bin/env python3
"""Per-kernel PMC averages from rocprofv3 rocpd databases: pmc_summary.py <dir with pass*_results.db> [name filter]"""
import glob, os, sqlite3, sys, collections
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
        if not any(f in kn for f in filt.split("|")): continue          # "a|b": either substring
        key = (kn.split("(")[0][-40:], gs)
        res.setdefault(key, {})[cn] = (v, n, dur)
if "--each" in sys.argv:      # every dispatch in launch order (shapes that share a kernel name and a grid: tools/lab_gemm_wo.py)
    for db in sorted(glob.glob(d + "/*_results.db")):
        con = sqlite3.connect(db)
        try:
            rows = con.execute("select dispatch_id, kernel_name, counter_name, value, end-start from counters_collection order by dispatch_id").fetchall()
        except Exception as e:
            print(db, "ERR", e); continue
        print("==", os.path.basename(db) if "os" in dir() else db)
        for did, kn, cn, v, dur in rows:
            if any(f in kn for f in filt.split("|")):
                print(f"  {did:5d} {kn.split('(')[0][-44:]:44s} {cn:30s} {v:16.1f} {dur/1e3:10.1f} us")
for key, c in res.items():
    print(key)
    for cn, (v, n, dur) in c.items():
        print(f"    {cn:34s} {v:16.1f}  (n={n}, avg dur {dur/1e3:.1f} us)")

# HBM-side traffic of the GEMM launches, per launch (bench.py reports it as roofline.traffic):
#   FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B => x2
#   (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is taken as reported (uncalibrated there).
import json, os
seen_f16 = False
tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
cnt = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
for key, c in res.items():
    if "gemm_bf16" not in key[0] and "gemm_f16" not in key[0]:     # the fp16 tower's launches are GEMM launches of the step too
        continue
    seen_f16 = seen_f16 or "gemm_f16" in key[0]
    for cn in tot:
        if cn in c:
            v, n, dur = c[cn]
            tot[cn] += v * n
            cnt[cn] += n
if cnt["FETCH_SIZE"] and cnt["WRITE_SIZE"]:
    fetch = 2.0 * 1024.0 * tot["FETCH_SIZE"] / cnt["FETCH_SIZE"]
    write = 1024.0 * tot["WRITE_SIZE"] / cnt["WRITE_SIZE"]
    out = {"source": "rocprofv3 --kernel-trace --pmc (separate FETCH_SIZE / WRITE_SIZE passes) over bench.py --steps 1 --warmup 1",
           "kernels": "vqs::gemm_bf16_* + vqs::gemm_f16_* + vqs::gemm_f16b_* (all GEMM launches of the run)", "launches_per_pass": cnt["FETCH_SIZE"],
           "vit_fp16": seen_f16,           # were fp16 GEMM launches (the fp16 vision tower / encoder attention side) part of the collection?
           "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
           "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 128-B requests tallied at 64 B); WRITE_SIZE KiB x 1024 as reported"}
    try:                                    # stamp with the kernel sources the counters were taken on (bench.py checks it)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["csrc_sha256_16"] = bench.csrc_hash()
        out["gemm_kernels_patterns"] = ["gemm_bf16_", "gemm_f16"]      # gemm_f16_quad (fp16 -> fp16) and gemm_f16b_quad (fp16 -> bf16) too
        out["gemm_kernels_sha256_16"] = bench.gemm_kernels_hash(patterns=tuple(out["gemm_kernels_patterns"]))   # machine code + descriptors of the GEMM kernels alone
        out["device_code_sha256_16"] = bench.device_code_hash()       # the .hip_fatbin section of the library the passes ran
    except Exception as e:
        out["csrc_sha256_16"] = None
        print("no source stamp:", e)
    with open(os.path.join(d, "gemm_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("gemm traffic per launch: fetch %.1f MB + write %.1f MB" % (fetch / 1e6, write / 1e6))
