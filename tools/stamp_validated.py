#!/usr/bin/env python3
"""Write profiles/validated_device_code.json for the library as built now: the device-code hashes tests/test_build_invariants.py compares
with, the hipcc version they hold for, and the GPU records that validated this build (arguments: paths under profiles/, optionally followed
by a note in parentheses; --not-run TEST ... names what was not re-run on this code)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from test_build_invariants import _hipcc_version  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("records", nargs="+")
ap.add_argument("--not-run", nargs="*", default=[])
a = ap.parse_args()
so = os.path.join(ROOT, "t2v_metrics_amd", "libvqs_hip.so")
rec = {"device_code_sha256_16": bench.device_code_hash(so), "gemm_kernels_sha256_16": bench.gemm_kernels_hash(so), "hipcc_version": _hipcc_version(),
       "what": "sha256 of the .hip_fatbin section (bench.device_code_hash) / of the GEMM kernels' machine code (bench.gemm_kernels_hash) of the library the GPU "
               "records below were made with; the build is deterministic on one toolchain (hipcc_version), so a tree that builds to these hashes there runs exactly "
               "the validated device code",
       "validated_by": a.records, "not_run_on_this_code": a.not_run}
for r in a.records:
    assert os.path.exists(os.path.join(ROOT, r.split(" ")[0])), r
json.dump(rec, open(os.path.join(ROOT, "profiles", "validated_device_code.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
